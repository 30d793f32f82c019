// comm.hip -- the transport of the share-vector open inside the library: RCCL over xGMI, one communicator per context.
//
// Replaces (reference): MpcSerNet::broadcast over MPCNetConnection::broadcast_bytes for the vector opens of
// AdditiveFieldShare::batch_open (mpc-algebra/src/channel.rs:12-28, mpc-net/src/multi.rs:469-525,
// mpc-algebra/src/share/additive.rs:124-131): every party ends up with sum_p v_p mod r.
// A host in the reference's language needs nothing but this C ABI: the leader obtains a 128-byte id
// (zk_comm_unique_id), ships it to the other parties over the TCP mesh it already has, every party calls
// zk_comm_init, and the opens run GPU to GPU.
//
// Pattern (same as mpc.py::DistNet.open_sum): for P >= 3 an all-to-all of slices (grouped send / recv), a local sum of
// the P slices, an all-gather of the summed slices: 2 x 32 n bytes in per GPU for any P; for P <= 2 one all-gather and a
// local sum.  RCCL is loaded with dlopen on first use, so the library carries no link-time dependency on it (a process
// that also hosts PyTorch already has a copy mapped).
#include "../../include/zkmpc_hip.h"
#include "ctx.hpp"
#include "internal.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};

// Loaded once per process, by whichever thread comes first (a function-local static is initialised exactly once even when
// several parties' threads arrive together, the LocalTestNet shape of mpc-net/src/multi.rs:419-443); immutable afterwards.
// WHICH copy: a process that hosts PyTorch already has one mapped (torch/lib/librccl.so, soname librccl.so.1) and a second
// one (/opt/rocm/lib) would bring its own bootstrap threads, IPC handle caches and a second set of kernels for the same
// devices.  So the rule is "exactly one RCCL per process": ZK_RCCL_LIB if set; otherwise the copy that is ALREADY mapped
// (RTLD_NOLOAD by soname, which matches torch's whatever its path); only when none is mapped is one loaded by name.
// zk_comm_info reports the path that was bound.
Rccl load_rccl() {
    Rccl r;
    if (const char* forced = getenv("ZK_RCCL_LIB")) r.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) {
        for (const char* nm : {"librccl.so.1", "librccl.so"})
            if ((r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) break;
    }
    if (!r.lib) {
        for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"})
            if ((r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (r.lib) {
#define ZK_SYM(f) *(void**)(&r.f) = dlsym(r.lib, "nccl" #f)
        ZK_SYM(GetUniqueId); ZK_SYM(CommInitRank); ZK_SYM(CommDestroy); ZK_SYM(AllGather); ZK_SYM(Send); ZK_SYM(Recv);
        ZK_SYM(GroupStart); ZK_SYM(GroupEnd); ZK_SYM(GetErrorString); ZK_SYM(CommCount); ZK_SYM(CommUserRank); ZK_SYM(GetVersion);
#undef ZK_SYM
    }
    return r;
}
const Rccl* rccl(std::string* err) {
    static const Rccl r = load_rccl();
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) {
        if (err) *err = "RCCL not available (dlopen librccl.so failed or symbols missing; set ZK_RCCL_LIB)";
        return nullptr;
    }
    return &r;
}

struct Comm { ncclComm_t comm = nullptr; int rank = 0, n = 1; int pattern = 0; };   // pattern: zk_comm_set_open_pattern

#define ZK_NCCL(ctx, R, expr)                                                                   \
    do {                                                                                          \
        ncclResult_t _e = (expr);                                                                 \
        if (_e != ncclSuccess) {                                                                  \
            (ctx)->last_error = std::string(#expr " -> ") + ((R)->GetErrorString ? (R)->GetErrorString(_e) : "rccl error"); \
            return ZK_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

}  // namespace

extern "C" int zk_comm_unique_id(uint8_t out[128]) {
    ZK_API_BEGIN_NOCTX
    if (!out) return ZK_ERR_ARG;
    const Rccl* R = rccl(nullptr);
    if (!R) return ZK_ERR_STATE;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return ZK_ERR_HIP;
    static_assert(sizeof id == 128, "ncclUniqueId is 128 bytes");
    memcpy(out, &id, 128);
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_comm_init(zk_ctx* ctx, const uint8_t id_bytes[128], int rank, int n_parties) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !id_bytes || n_parties < 1 || rank < 0 || rank >= n_parties) return ZK_ERR_ARG;
    if (ctx->comm) ZK_FAIL(ctx, ZK_ERR_STATE, "zk_comm_init: this context already has a communicator");
    const Rccl* R = rccl(&ctx->last_error);
    if (!R) return ZK_ERR_STATE;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, 128);
    Comm* c = new Comm();
    c->rank = rank;
    c->n = n_parties;
    ncclResult_t e = R->CommInitRank(&c->comm, n_parties, id, rank);
    if (e != ncclSuccess) {
        delete c;
        ctx->last_error = std::string("ncclCommInitRank -> ") + (R->GetErrorString ? R->GetErrorString(e) : "rccl error");
        return ZK_ERR_HIP;
    }
    ctx->comm = c;
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_comm_destroy(zk_ctx* ctx) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    if (!ctx->comm) return ZK_OK;
    Comm* c = (Comm*)ctx->comm;
    const Rccl* R = rccl(nullptr);
    (void)hipStreamSynchronize(ctx->stream);
    if (R && R->CommDestroy && c->comm) (void)R->CommDestroy(c->comm);
    delete c;
    ctx->comm = nullptr;
    return ZK_OK;
    ZK_API_END
}

// What the communicator itself says (ncclCommCount / ncclCommUserRank, not the arguments of zk_comm_init), the device it lives
// on, RCCL's version code and the path of the library copy that was bound (dladdr of ncclGetUniqueId).  Any out pointer may be
// NULL.  Without a communicator: n_ranks = 0, rank = -1, and the library fields are still filled in when RCCL can be loaded.
extern "C" int zk_comm_info(zk_ctx* ctx, int* n_ranks, int* rank, int* device, int* rccl_version, char* lib_path, size_t lib_path_cap) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    if (n_ranks) *n_ranks = 0;
    if (rank) *rank = -1;
    if (device) *device = ctx->device;
    if (rccl_version) *rccl_version = 0;
    if (lib_path && lib_path_cap) lib_path[0] = 0;
    const Rccl* R = rccl(&ctx->last_error);
    if (!R) return ZK_ERR_STATE;
    if (rccl_version && R->GetVersion) (void)R->GetVersion(rccl_version);
    if (lib_path && lib_path_cap) {
        Dl_info di;
        if (dladdr((void*)R->GetUniqueId, &di) && di.dli_fname) { strncpy(lib_path, di.dli_fname, lib_path_cap - 1); lib_path[lib_path_cap - 1] = 0; }
    }
    if (ctx->comm) {
        Comm* c = (Comm*)ctx->comm;
        int v = c->n;
        if (n_ranks) { if (R->CommCount) ZK_NCCL(ctx, R, R->CommCount(c->comm, &v)); *n_ranks = v; }
        v = c->rank;
        if (rank) { if (R->CommUserRank) ZK_NCCL(ctx, R, R->CommUserRank(c->comm, &v)); *rank = v; }
    }
    return ZK_OK;
    ZK_API_END
}

// 0: by party count (all-gather for two parties, all-to-all of slices for three or more); 1: all-gather; 2: all-to-all.
// Every party of the communicator must make the same call.
extern "C" int zk_comm_set_open_pattern(zk_ctx* ctx, int pattern) {
    ZK_API_BEGIN(ctx)
    if (!ctx || pattern < 0 || pattern > 2) return ZK_ERR_ARG;
    if (!ctx->comm) ZK_FAIL(ctx, ZK_ERR_STATE, "zk_comm_set_open_pattern: no communicator (zk_comm_init)");
    ((Comm*)ctx->comm)->pattern = pattern;
    return ZK_OK;
    ZK_API_END
}

// out[i] = sum over parties of v[i] mod r, on every party (v, out: n field elements on the device; out may alias v)
extern "C" int zk_open_sum_fr_dev(zk_ctx* ctx, const void* v_dev, size_t n, void* out_dev) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (n && (!v_dev || !out_dev))) return ZK_ERR_ARG;
    if (!ctx->comm) ZK_FAIL(ctx, ZK_ERR_STATE, "zk_open_sum_fr_dev: no communicator (zk_comm_init)");
    if (n == 0) return ZK_OK;
    Comm* c = (Comm*)ctx->comm;
    const Rccl* R = rccl(&ctx->last_error);
    if (!R) return ZK_ERR_STATE;
    const int P = c->n;
    hipStream_t st = ctx->stream;
    // the exchange pattern is a property of the communicator (party count, or zk_comm_set_open_pattern on every party):
    // parties that disagreed about it would wait for each other in different collectives
    const bool a2a = c->pattern ? c->pattern == 2 : P >= 3;
    if (!a2a) {
        void* recv;
        ZK_TRY(zk_scratch(ctx, "open_recv", (size_t)P * n * 32, &recv));
        ZK_NCCL(ctx, R, R->AllGather(v_dev, recv, n * 4, ncclUint64, c->comm, st));
        return zk_fr_sum_parties_dev(ctx, recv, (size_t)P, n, out_dev);
    }
    const size_t chunk = (n + P - 1) / P;
    char *send, *recv, *part, *full;
    ZK_TRY(zk_scratch(ctx, "open_pad", (size_t)P * chunk * 32, (void**)&send));
    ZK_TRY(zk_scratch(ctx, "open_recv", (size_t)P * chunk * 32, (void**)&recv));
    ZK_TRY(zk_scratch(ctx, "open_part", chunk * 32, (void**)&part));
    ZK_TRY(zk_scratch(ctx, "open_full", (size_t)P * chunk * 32, (void**)&full));
    ZK_HIP(ctx, hipMemcpyAsync(send, v_dev, n * 32, hipMemcpyDeviceToDevice, st));
    if ((size_t)P * chunk > n) ZK_HIP(ctx, hipMemsetAsync(send + n * 32, 0, ((size_t)P * chunk - n) * 32, st));
    ZK_NCCL(ctx, R, R->GroupStart());
    ncclResult_t ge = ncclSuccess;
    for (int p = 0; p < P && ge == ncclSuccess; p++) {
        ge = R->Send(send + (size_t)p * chunk * 32, chunk * 4, ncclUint64, p, c->comm, st);                  // slice p goes to party p
        if (ge == ncclSuccess) ge = R->Recv(recv + (size_t)p * chunk * 32, chunk * 4, ncclUint64, p, c->comm, st);   // my slice of party p
    }
    const ncclResult_t ge_end = R->GroupEnd();     // always closed: an open group would hang every later collective on this communicator
    if (ge != ncclSuccess || ge_end != ncclSuccess) {
        ctx->last_error = std::string("zk_open_sum_fr_dev: ncclSend/ncclRecv group -> ") +
                          (R->GetErrorString ? R->GetErrorString(ge != ncclSuccess ? ge : ge_end) : "rccl error");
        return ZK_ERR_HIP;
    }
    ZK_TRY(zk_fr_sum_parties_dev(ctx, recv, (size_t)P, chunk, part));
    ZK_NCCL(ctx, R, R->AllGather(part, full, chunk * 4, ncclUint64, c->comm, st));
    ZK_HIP(ctx, hipMemcpyAsync(out_dev, full, n * 32, hipMemcpyDeviceToDevice, st));
    return ZK_OK;
    ZK_API_END
}
