// ctx.hpp -- per-party context of libzkmpc_hip: device, stream, error string, device arena,
// NTT domain cache.  One context per MPC party / GPU; nothing is process-global, so several
// parties can live in one process (the reference's LocalTestNet does that,
// mpc-net/src/multi.rs:419-443).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <memory>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#define ZK_OK 0
#define ZK_ERR_HIP -1
#define ZK_ERR_ARG -2
#define ZK_ERR_NOMEM -3
#define ZK_ERR_STATE -4
#define ZK_ERR_MAC -5

struct zk_domain;  // ntt.hip

// The context's host helper threads (window Horner chains, the O(1) scalar multiplications of a proof, Marlin's blinding
// terms: ~20 short tasks per proof).  Workers are created on demand and then kept: a task is queued only when an idle worker
// will take it, otherwise a new worker is started first -- so a task that waits for other tasks can never starve them -- and
// if the system refuses another thread the task runs in the caller.  After the first proof no thread is created any more.
class ZkHostPool {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::vector<std::thread> workers;
    size_t idle = 0;
    bool stop = false;

    void run() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            idle++;
            cv.wait(lk, [&] { return stop || !q.empty(); });
            idle--;
            if (q.empty()) return;                  // stop, and nothing left to do
            std::function<void()> job = std::move(q.front());
            q.pop_front();
            lk.unlock();
            job();                                  // (a packaged_task: exceptions land in its future)
            lk.lock();
        }
    }

   public:
    ZkHostPool() = default;
    ZkHostPool(const ZkHostPool&) = delete;
    ZkHostPool& operator=(const ZkHostPool&) = delete;
    ~ZkHostPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        for (auto& t : workers) t.join();
    }
    size_t threads() {
        std::lock_guard<std::mutex> lk(m);
        return workers.size();
    }
    template <class Fn>
    auto submit(Fn&& fn) -> std::future<decltype(fn())> {
        using R = decltype(fn());
        auto task = std::make_shared<std::packaged_task<R()>>(std::forward<Fn>(fn));
        std::future<R> fut = task->get_future();
        bool inline_run = false;
        {
            std::lock_guard<std::mutex> lk(m);
            if (q.size() + 1 > idle) {              // every queued task has an idle worker of its own, or gets a new one
                try {
                    workers.emplace_back([this] { run(); });
                } catch (const std::system_error&) {
                    inline_run = true;              // thread exhaustion must not fail a proof
                }
            }
            if (!inline_run) q.emplace_back([task] { (*task)(); });
        }
        if (inline_run) (*task)();
        else cv.notify_one();
        return fut;
    }
};

struct ZkActivity {
    std::atomic<int> calls{0};
    std::atomic<int64_t> last_leave_ns{0};
    // no entry point of the context executing, and none has returned within the last `ns` (two calls of a burst -- seven transforms,
    // five MSMs -- are microseconds apart: not a gap; the caller's own scalar loops between two bursts are milliseconds: a gap)
    bool quiet_for(int64_t ns) const {
        const int64_t now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        return calls.load(std::memory_order_relaxed) == 0 && now - last_leave_ns.load(std::memory_order_relaxed) >= ns;
    }
};

struct zk_ctx {
    std::shared_ptr<ZkActivity> activity = std::make_shared<ZkActivity>();
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
    // core.hip: zk_graph_run -- a launch-bound sequence (the ~13 launches of a bucket sort) replayed as one captured graph.  Key = a
    // digest of everything the launches depend on, `scratch_gen` included (it moves whenever a scratch slot is allocated anew: every
    // graph that baked the old addresses in stops matching).
    struct GraphEntry { hipGraphExec_t exec = nullptr; int seen = 0; uint64_t gen = 0; hipStream_t last = nullptr; };
    std::map<std::string, GraphEntry> graphs;
    uint64_t scratch_gen = 0;
    uint64_t graphs_gen = 0;                  // the generation the entries of `graphs` were swept for
    void* comm = nullptr;                     // RCCL communicator of this party (comm.hip), created by zk_comm_init
    void* xfer = nullptr;                     // hostxfer.hip: the page-locked ring host slices travel through (ZkXfer)
    void* xfer_small = nullptr;               // hostxfer.hip: rotating page-locked slots for transfers below 128 KiB
    void* bases_cache = nullptr;              // bases_cache.hip: resident copies of host base slices seen by zk_msm_g1 / _g2 (ZkBasesCache)
    void* msm_spec = nullptr;                 // msm.hip: MSMs over the tables a caller asks for next with the same scalars, started ahead (ZkMsmSpec)
    void* presort = nullptr;                  // groth16_pipeline.hip: a sort of z[1..] enqueued ahead of zk_groth16_msms_dev (ZkPresort)
    const void* next_z = nullptr;             // groth16_pipeline.hip: zk_groth16_hint_next_dev
    // groth16_prove.hip: zk_groth16_hint_next (host-slice form): the announced assignment is uploaded on its own stream into the
    // idle one of two device slots while the current proof runs; next_z_ready is recorded behind that copy
    hipStream_t copy_stream = nullptr;
    hipEvent_t next_z_ready = nullptr;
    bool chain_fronts = true;                 // zk_groth16_chain_fronts: a small proof's front carries the next proof's whole device chain
    int front_parity = 0;                     // groth16_pipeline.hip: which pair of pinned result buffers the next chained front writes
    const void* next_z_host = nullptr;        // the host buffer that was announced (matched by address by zk_groth16_prove)
    const void* next_z_pk = nullptr;          // ... together with the key, the constraint system and the length it was announced for
    const void* next_z_r = nullptr;
    size_t next_z_m = 0;
    void* next_z_dev = nullptr;               // where its copy lives
    int z_slot = 0;                           // which of the two "prove_z" slots the CURRENT proof reads
    std::mutex mu;
    // timing of the most recent instrumented call (ms), filled when ZK_PROFILE env or explicit request
    struct Timer { float ms = 0; int count = 0; };
    std::map<std::string, Timer> timers;
    bool profiling = false;
    ZkHostPool pool;                // host helper threads (last member: joined first when the context goes)
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

// ---- the C-ABI boundary --------------------------------------------------------------------------------------------------
// Every `extern "C" int` entry point runs its body inside zk_api_guarded:
//   * the calling thread's current HIP device becomes ctx->device for the duration of the call and is restored afterwards --
//     a context owns its device, whatever the thread had selected (several parties on several GPUs inside one process, the
//     reference's LocalTestNet shape: mpc-net/src/multi.rs:419-443; SURVEY 8b "no process-global device state");
//   * no C++ exception crosses the boundary (the bodies use std::vector / std::map / std::string / std::async): bad_alloc
//     becomes ZK_ERR_NOMEM, anything else ZK_ERR_STATE, with the text in zk_last_error -- the header's "never abort" contract.
struct ZkDeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;           // a failing hipGetDevice / hipSetDevice: the body must NOT run on whatever device the thread holds
    explicit ZkDeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~ZkDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    ZkDeviceGuard(const ZkDeviceGuard&) = delete;
    ZkDeviceGuard& operator=(const ZkDeviceGuard&) = delete;
};

// Entry points executing on a context right now, and when the last one returned: the table cache's builder thread
// (bases_cache.hip) hands out its background work in the gaps -- the caller's own time between two bursts of calls.  Held through
// a shared pointer so that the mark of zk_ctx_destroy outlives the context it destroys.
struct ZkCallMark {
    std::shared_ptr<ZkActivity> a;
    explicit ZkCallMark(zk_ctx* c) : a(c->activity) { a->calls.fetch_add(1, std::memory_order_relaxed); }
    ~ZkCallMark() {
        a->last_leave_ns.store(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(),
                               std::memory_order_relaxed);
        a->calls.fetch_sub(1, std::memory_order_relaxed);
    }
    ZkCallMark(const ZkCallMark&) = delete;
    ZkCallMark& operator=(const ZkCallMark&) = delete;
};

template <class Fn>
static inline int zk_api_guarded(zk_ctx* ctx, Fn&& body) noexcept {
    // (ctx may be destroyed by the body -- zk_ctx_destroy -- so nothing below touches it after a normal return)
    try {
        if (!ctx) return body();
        ZkCallMark mark(ctx);
        ZkDeviceGuard guard(ctx->device);
        if (guard.err != hipSuccess) {
            // another party's GPU would take this context's pointers: silent corruption instead of an error code
            ctx->last_error = std::string("cannot make device ") + std::to_string(ctx->device) + " current: " + hipGetErrorString(guard.err);
            return ZK_ERR_HIP;
        }
        return body();
    } catch (const std::bad_alloc&) {
        try { if (ctx) ctx->last_error = "out of host memory (std::bad_alloc)"; } catch (...) {}
        return ZK_ERR_NOMEM;
    } catch (const std::exception& e) {
        try { if (ctx) ctx->last_error = std::string("C++ exception at the C ABI: ") + e.what(); } catch (...) {}
        return ZK_ERR_STATE;
    } catch (...) {
        try { if (ctx) ctx->last_error = "unknown C++ exception at the C ABI"; } catch (...) {}
        return ZK_ERR_STATE;
    }
}
// The handle of a host-side helper task: a future that JOINS in its destructor, as std::async's does and a packaged_task's does
// not.  The tasks capture their frame by reference (window sums, the host chains of a proof, Marlin's blinding terms), so a
// frame left early -- a ZK_TRY return, an exception on its way to the C-ABI barrier -- must not be unwound under a running
// task.  Declare a ZkTask AFTER everything its task references: locals are destroyed in reverse order.  (A deferred std::async
// future is not run by the destructor: nothing of it is in flight.)
template <class R>
class ZkTask {
    std::future<R> f;
    void join() noexcept {
        if (!f.valid()) return;
        try {
            if (f.wait_for(std::chrono::seconds(0)) != std::future_status::deferred) f.wait();
        } catch (...) {}
    }

   public:
    ZkTask() = default;
    ZkTask(std::future<R>&& x) : f(std::move(x)) {}          // (implicit: zk_async and std::async results both land here)
    ZkTask(ZkTask&& o) noexcept : f(std::move(o.f)) {}
    ZkTask& operator=(ZkTask&& o) noexcept {
        if (this != &o) { join(); f = std::move(o.f); }
        return *this;
    }
    ZkTask(const ZkTask&) = delete;
    ZkTask& operator=(const ZkTask&) = delete;
    ~ZkTask() { join(); }
    bool valid() const { return f.valid(); }
    void wait() { f.wait(); }
    R get() { return f.get(); }
};

// A host-side helper task on the context's worker pool.
template <class Fn>
static inline auto zk_async(zk_ctx* ctx, Fn&& fn) -> ZkTask<decltype(fn())> {
    return ZkTask<decltype(fn())>(ctx->pool.submit(std::forward<Fn>(fn)));
}

#define ZK_API_BEGIN(ctx) return zk_api_guarded((zk_ctx*)(ctx), [&]() -> int {
#define ZK_API_BEGIN_NOCTX return zk_api_guarded((zk_ctx*)nullptr, [&]() -> int {
#define ZK_API_END });

// Returns a device buffer of at least `bytes` bound to `name`; contents are unspecified.
int zk_scratch(zk_ctx* ctx, const char* name, size_t bytes, void** out);
// The same for a buffer whose users leave it all-zero behind them: it is zeroed when it is (re)allocated, never afterwards.
int zk_scratch_zeroed(zk_ctx* ctx, const char* name, size_t bytes, void** out);

static inline unsigned zk_grid(size_t work, unsigned block, unsigned cap = 256 * 16) {
    size_t g = (work + block - 1) / block;
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (unsigned)g;
}
