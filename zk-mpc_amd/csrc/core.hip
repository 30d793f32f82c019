// core.hip -- context, device memory, scratch arena, timers.
#include "../../include/zkmpc_hip.h"
#include "ctx.hpp"
#include "internal.hpp"
#include <atomic>
#include <functional>
#include <stdexcept>
#include <string.h>

extern "C" int zk_version(void) { ZK_API_BEGIN_NOCTX return 1; ZK_API_END }

extern "C" int zk_selftest_exception_barrier(int kind) {
    ZK_API_BEGIN_NOCTX
    if (kind == 0) throw std::bad_alloc();
    if (kind == 1) throw std::runtime_error("selftest");
    if (kind == 2) throw 42;
    if (kind == 3) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
    if (kind == 4) {       // the helper pool: nested waits (a task that waits for tasks) finish, results and exceptions arrive
        ZkHostPool pool;
        auto outer = pool.submit([&] {
            int s = 0;
            std::vector<std::future<int>> in;
            for (int i = 0; i < 8; i++) in.push_back(pool.submit([i] { return i; }));
            for (auto& f : in) s += f.get();
            return s;
        });
        auto bad = pool.submit([]() -> int { throw std::runtime_error("task"); });
        if (outer.get() != 28) return ZK_ERR_STATE;
        try { (void)bad.get(); return ZK_ERR_STATE; } catch (const std::runtime_error&) {}
        const size_t after_first = pool.threads();
        for (int r = 0; r < 4; r++) if (pool.submit([r] { return r; }).get() != r) return ZK_ERR_STATE;
        return pool.threads() == after_first ? ZK_OK : ZK_ERR_STATE;      // steady state: no new threads
    }
    return ZK_ERR_ARG;
    ZK_API_END
}

// Equal stream priorities: giving the accumulate stream the lowest and every other stream the highest priority was measured on
// MI355X / ROCm 7 and changes nothing (30.5 vs 30.0 ms per proof in round 1; an ungated witness map is still starved).
hipError_t zk_stream_create(hipStream_t* st, bool) { return hipStreamCreateWithFlags(st, hipStreamNonBlocking); }

extern "C" int zk_ctx_create(int device, int party_id, int n_parties, zk_ctx** out) {
    ZK_API_BEGIN_NOCTX
    if (!out || n_parties < 1 || party_id < 0 || party_id >= n_parties) return ZK_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ZK_ERR_HIP;  // no CPU fallback
    if (device < 0 || device >= ndev) return ZK_ERR_ARG;
    zk_ctx* c = new zk_ctx();
    c->device = device;
    c->party_id = party_id;
    c->n_parties = n_parties;
    if (hipSetDevice(device) != hipSuccess || zk_stream_create(&c->stream, true) != hipSuccess) {
        delete c;
        return ZK_ERR_HIP;
    }
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
    // the bucket sort and the transforms stage up to 144 KiB per workgroup in LDS: CDNA4 (gfx950, 160 KiB).  On anything smaller
    // every MSM of 2^16 digits and every transform would fail at its first launch; say so here instead
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0 && lds < 147456) {
        fprintf(stderr, "libzkmpc_hip: device %d offers %d bytes of LDS per workgroup; this library is written for gfx950 (160 KiB)\n", device, lds);
        (void)hipStreamDestroy(c->stream);
        delete c;
        return ZK_ERR_HIP;
    }
    *out = c;
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_ctx_destroy(zk_ctx* ctx) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)zk_comm_destroy(ctx);
    for (auto& kv : ctx->slots)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& kv : ctx->pinned)
        if (kv.second.p) (void)hipHostFree(kv.second.p);
    zk_presort_free(ctx);
    zk_msm_spec_free(ctx);
    zk_bases_cache_free(ctx);
    for (auto& kv : ctx->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    ctx->graphs.clear();
    zk_xfer_free(ctx);
    zk_domains_free(ctx);
    for (auto st : ctx->aux) (void)hipStreamDestroy(st);
    if (ctx->acc_stream) (void)hipStreamDestroy(ctx->acc_stream);
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
    if (ctx->next_z_ready) (void)hipEventDestroy(ctx->next_z_ready);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return ZK_OK;
    ZK_API_END
}

extern "C" const char* zk_last_error(zk_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

extern "C" int zk_ctx_sync(zk_ctx* ctx) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
    ZK_API_END
}

extern "C" void* zk_ctx_stream(zk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int zk_dev_alloc(zk_ctx* ctx, size_t bytes, void** dev_out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !dev_out) return ZK_ERR_ARG;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    ZK_HIP(ctx, hipMalloc(dev_out, bytes ? bytes : 16));
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_dev_free(zk_ctx* ctx, void* dev) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZK_HIP(ctx, hipFree(dev));
    return ZK_OK;
    ZK_API_END
}

// Page-locked host memory for buffers that cross the boundary often (an assignment vector handed to zk_groth16_prove:
// 32 MiB at 2^20 variables copies in ~0.6 ms from pinned memory, several ms from pageable memory).
extern "C" int zk_host_alloc(zk_ctx* ctx, size_t bytes, void** host_out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !host_out) return ZK_ERR_ARG;
    ZK_HIP(ctx, hipSetDevice(ctx->device));
    ZK_HIP(ctx, hipHostMalloc(host_out, bytes ? bytes : 16, hipHostMallocDefault));
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_host_free(zk_ctx* ctx, void* host) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    if (host) ZK_HIP(ctx, hipHostFree(host));
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_memcpy_h2d(zk_ctx* ctx, void* dev, const void* host, size_t bytes) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ZK_TRY(zk_xfer_h2d(ctx, dev, host, bytes, zk_host_is_pinned(host)));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_memcpy_d2h(zk_ctx* ctx, void* host, const void* dev, size_t bytes) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ZK_TRY(zk_xfer_d2h(ctx, host, dev, bytes, zk_host_is_pinned(host)));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_memcpy_d2d(zk_ctx* ctx, void* dst, const void* src, size_t bytes) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (bytes && (!dst || !src))) return ZK_ERR_ARG;
    if (bytes) ZK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_dev_zero(zk_ctx* ctx, void* dev, size_t bytes) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (bytes && !dev)) return ZK_ERR_ARG;
    if (bytes) ZK_HIP(ctx, hipMemsetAsync(dev, 0, bytes, ctx->stream));
    return ZK_OK;
    ZK_API_END
}

int zk_scratch(zk_ctx* ctx, const char* name, size_t bytes, void** out) {
    auto& s = ctx->slots[name];
    if (s.bytes < bytes) {
        if (s.p) {
            ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ZK_HIP(ctx, hipFree(s.p));
            s.p = nullptr;
            s.bytes = 0;
        }
        size_t want = bytes + bytes / 8 + 256;
        ZK_HIP(ctx, hipMalloc(&s.p, want));
        s.bytes = want;
        ctx->scratch_gen++;                            // (captured graphs hold scratch addresses: zk_graph_run)
    }
    *out = s.p;
    return ZK_OK;
}

// Run `enqueue` -- kernel launches and memsets on `st` ONLY: no allocation, no host wait, no other stream -- and from the third use
// with the same key on replay it as ONE graph launch.  First use: plain (scratch slots and function attributes come into being);
// second use: captured while it is enqueued; a key must cover every argument of every launch (ctx->scratch_gen covers the scratch
// addresses).  With the phase timers on, or ZK_GRAPHS=0, always plain.  A sort of a 2^14-scalar MSM is 13 launches of 5 - 15 us
// of device time each: the host's ~5 us per launch was what a small Marlin round waited for.
int zk_graph_run(zk_ctx* ctx, const std::string& key, hipStream_t st, const std::function<int()>& enqueue) {
    static const bool off = getenv("ZK_GRAPHS") && atoi(getenv("ZK_GRAPHS")) == 0;
    if (off || ctx->profiling) return enqueue();
    // entries of an older scratch generation can never match again (the generation is part of every key): they go as soon as the
    // generation moves, each executable behind the stream it last ran on (they used to pile up to 512 and were then
    // destroyed all at once, in flight or not)
    auto retire = [&](bool all) {
        for (auto it = ctx->graphs.begin(); it != ctx->graphs.end();) {
            if (all || it->second.gen != ctx->scratch_gen) {
                if (it->second.exec) {
                    if (it->second.last) (void)hipStreamSynchronize(it->second.last);
                    (void)hipGraphExecDestroy(it->second.exec);
                }
                it = ctx->graphs.erase(it);
            } else {
                ++it;
            }
        }
    };
    if (ctx->graphs_gen != ctx->scratch_gen) { retire(false); ctx->graphs_gen = ctx->scratch_gen; }
    if (ctx->graphs.size() > 512) retire(true);        // (keys that stopped matching for other reasons: other sizes)
    auto& e = ctx->graphs[key];
    e.gen = ctx->scratch_gen;
    if (e.exec) {
        ZK_HIP(ctx, hipGraphLaunch(e.exec, st));
        e.last = st;
        return ZK_OK;
    }
    if (e.seen++ == 0) return enqueue();
    const uint64_t gen = ctx->scratch_gen;
    if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); return enqueue(); }
    const int rc = enqueue();
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(st, &g);
    if (rc != ZK_OK || ce != hipSuccess || !g || gen != ctx->scratch_gen) {
        // nothing ran (a capture only records): not capturable as it stands -- enqueue it for real and stop trying for this key
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        e.seen = -(1 << 30);
        if (rc != ZK_OK) return rc;
        return enqueue();
    }
    hipGraphExec_t x = nullptr;
    const hipError_t ie = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess || !x) { (void)hipGetLastError(); e.seen = -(1 << 30); return enqueue(); }
    e.exec = x;
    ZK_HIP(ctx, hipGraphLaunch(e.exec, st));
    e.last = st;
    return ZK_OK;
}

int zk_scratch_zeroed(zk_ctx* ctx, const char* name, size_t bytes, void** out) {
    // (by size, not by address: a slot that grows is freed and allocated again, and the allocator may hand the old address back)
    const auto it = ctx->slots.find(name);
    const size_t before = it == ctx->slots.end() ? 0 : it->second.bytes;
    ZK_TRY(zk_scratch(ctx, name, bytes, out));
    if (ctx->slots[name].bytes != before) {
        ZK_HIP(ctx, hipMemsetAsync(*out, 0, ctx->slots[name].bytes, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ZK_OK;
}

extern "C" int zk_set_profiling(zk_ctx* ctx, int on) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ctx->profiling = on != 0;
    ctx->timers.clear();
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_last_timers(zk_ctx* ctx, char* names, size_t name_stride, float* ms, int* counts, int max_entries) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    int k = 0;
    for (auto& kv : ctx->timers) {
        if (k >= max_entries) break;
        strncpy(names + k * name_stride, kv.first.c_str(), name_stride - 1);
        names[k * name_stride + name_stride - 1] = 0;
        ms[k] = kv.second.ms;
        if (counts) counts[k] = kv.second.count;
        k++;
    }
    ctx->timers.clear();
    return k;
    ZK_API_END
}
