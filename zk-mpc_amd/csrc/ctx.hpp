// ctx.hpp -- per-party context of libzkmpc_hip: device, stream, error string, device arena,
// NTT domain cache.  One context per MPC party / GPU; nothing is process-global, so several
// parties can live in one process (the reference's LocalTestNet does that,
// mpc-net/src/multi.rs:419-443).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#define ZK_OK 0
#define ZK_ERR_HIP -1
#define ZK_ERR_ARG -2
#define ZK_ERR_NOMEM -3
#define ZK_ERR_STATE -4

struct zk_domain;  // ntt.hip

struct zk_ctx {
    int device = 0;
    int party_id = 0;
    int n_parties = 1;
    int n_cu = 256;                 // compute units of the device
    hipStream_t stream = nullptr;
    std::vector<hipStream_t> aux;   // high-priority helper streams for concurrent MSMs (created on first use)
    hipStream_t acc_stream = nullptr;  // stream carrying the accumulate kernels back to back
    std::string last_error;
    // grow-only scratch arena: named slots, each re-used across calls (no hipMalloc on the hot path)
    struct Slot { void* p = nullptr; size_t bytes = 0; };
    std::map<std::string, Slot> slots;
    std::map<int, Slot> pinned;     // pinned host staging per MSM slot
    std::map<uint32_t, zk_domain*> domains;  // keyed by log2(size)
    std::map<std::string, int> flags;         // one-time per-context setup markers
    void* comm = nullptr;                     // RCCL communicator of this party (comm.hip), created by zk_comm_init
    void* presort = nullptr;                  // groth16.hip: a sort of z[1..] enqueued ahead of zk_groth16_msms_dev (ZkPresort)
    const void* next_z = nullptr;             // groth16.hip: zk_groth16_hint_next_dev
    // groth16.hip: zk_groth16_hint_next (host-slice form): the announced assignment is uploaded on its own stream into the
    // idle one of two device slots while the current proof runs; next_z_ready is recorded behind that copy
    hipStream_t copy_stream = nullptr;
    hipEvent_t next_z_ready = nullptr;
    const void* next_z_host = nullptr;        // the host buffer that was announced (matched by address by zk_groth16_prove)
    void* next_z_dev = nullptr;               // where its copy lives
    int z_slot = 0;                           // which of the two "prove_z" slots the CURRENT proof reads
    std::mutex mu;
    // timing of the most recent instrumented call (ms), filled when ZK_PROFILE env or explicit request
    struct Timer { float ms = 0; int count = 0; };
    std::map<std::string, Timer> timers;
    bool profiling = false;
};

#define ZK_HIP(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            char _b[512];                                                                         \
            snprintf(_b, sizeof _b, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            (ctx)->last_error = _b;                                                               \
            return ZK_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define ZK_FAIL(ctx, code, msg)         \
    do {                                \
        (ctx)->last_error = (msg);      \
        return (code);                  \
    } while (0)

#define ZK_TRY(expr)                    \
    do {                                \
        int _r = (expr);                \
        if (_r != ZK_OK) return _r;     \
    } while (0)

hipError_t zk_stream_create(hipStream_t* st, bool high_priority);   // core.hip

// Returns a device buffer of at least `bytes` bound to `name`; contents are unspecified.
int zk_scratch(zk_ctx* ctx, const char* name, size_t bytes, void** out);

static inline unsigned zk_grid(size_t work, unsigned block, unsigned cap = 256 * 16) {
    size_t g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (unsigned)g;
}
