// hostxfer.hip -- caller-owned host slices <-> HBM for the trait-shaped entry points.
//
// The reference reaches the arithmetic through trait methods that take HOST slices: AffineCurve::multi_scalar_mul(&[G], &[F])
// (arkworks/algebra/ec/src/lib.rs:305-314), EvaluationDomain::*fft_in_place(&mut Vec<F>) (poly/src/domain/mod.rs:78-157),
// Field::batch_product_in_place (ff/src/fields/mod.rs:216-220).  A Groth16 proof at 2^20 moves ~800 MB through them
// (src/groth16.rs:106,110,193,278-303), from and to pageable memory the caller allocated a moment ago (`vec![zero; domain_size]`)
// and frees a moment later.
//
// Handing such memory to the runtime is fast on a good day (hipMemcpy pins the pages and lets the DMA engine work in place) and
// a 20 - 27 ms stall on a bad one: memory that was pinned -- by the runtime's lazy pin or by hipHostRegister / hipHostUnregister
// around the call, it makes no difference -- and is then FREED by its owner takes the mapping down through the kernel's MMU
// notifier, and the next submission of the process waits for the driver to restore its queues.  Measured on the composed prover
// (profiles/r5_trace_fft.txt, r5_trait_pin.jsonl): the first transform of every proof after the first spent 21 - 27 ms inside
// the H2D copy of a 2 MB vector with hipMemcpyAsync, and 20 ms with register / unregister at 8 MB, while the other six calls took
// 0.05 - 0.4 ms; which sizes are hit depends on the allocator's and the driver's state, not on anything a caller controls.
// So the library never shows caller memory to the driver.  A ring of page-locked memory (64 MiB per context, allocated once) is
// filled and drained by a small team of host threads in 256 KiB pieces while the DMA engine moves the pieces already done, in a
// few large copies (an API call costs ~15 us: one per piece would cost more than the transfer).  A transfer costs ~max(host
// copy, DMA); no driver state depends on what the caller does with its memory afterwards.
#include "internal.hpp"
#include <atomic>
#include <immintrin.h>
#include <string.h>

namespace {

// memcpy into the ring with non-temporal stores: the ring is read by the DMA engine only, and a line that sits dirty in a core's
// cache has to be snooped out by every DMA read -- the engine crawled while the team was still filling and did the whole copy
// afterwards (32 MB: fill 0.46 ms + 0.62 ms of DMA behind it; with streaming stores 0.35 + 0.38).  dst and len are multiples of
// 64 except for the tail of the last piece.
__attribute__((target("avx2"))) void copy_nt_avx2(char* dst, const char* src, size_t len) {
    size_t i = 0;
    if (((uintptr_t)dst & 31) == 0) {
        for (; i + 128 <= len; i += 128) {
            const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
            const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
            _mm256_stream_si256((__m256i*)(dst + i), a); _mm256_stream_si256((__m256i*)(dst + i + 32), b);
            _mm256_stream_si256((__m256i*)(dst + i + 64), c); _mm256_stream_si256((__m256i*)(dst + i + 96), d);
        }
        _mm_sfence();
    }
    if (i < len) memcpy(dst + i, src + i, len - i);
}
void copy_to_ring(char* dst, const char* src, size_t len) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) copy_nt_avx2(dst, src, len);
    else memcpy(dst, src, len);
}

constexpr size_t XF_RING = (size_t)64 << 20;      // page-locked bytes per context
constexpr size_t XF_PIECE = (size_t)256 << 10;    // what one thread copies at a time
constexpr size_t XF_NPIECE = XF_RING / XF_PIECE;  // 256
constexpr size_t XF_SMALL = (size_t)128 << 10;    // below this a transfer is one plain copy (the runtime's bounce buffers: no pinning)
constexpr size_t XF_MAXEV = 32;

// The team: persistent threads that spin for a moment after a job (the entry points come in bursts: seven transforms, five MSMs)
// before they go to sleep, so that handing over a piece costs a cache miss, not a futex.
class XferTeam {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<uint64_t> gen{0};
    std::atomic<size_t> next{0}, done{0};
    std::atomic<int> inside{0};
    std::atomic<bool> stop{false};
    size_t npieces = 0;
    const std::function<void(size_t)>* fn = nullptr;

    void pieces() {
        for (;;) {
            const size_t p = next.fetch_add(1);
            if (p >= npieces) return;
            (*fn)(p);
            done.fetch_add(1, std::memory_order_release);
        }
    }
    void worker(int device) {
        (void)hipSetDevice(device);
        uint64_t seen = 0;
        for (;;) {
            auto t0 = std::chrono::steady_clock::now();
            while (gen.load(std::memory_order_acquire) == seen && !stop.load()) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
                    std::unique_lock<std::mutex> lk(m);
                    cv.wait(lk, [&] { return gen.load() != seen || stop.load(); });
                    break;
                }
                std::this_thread::yield();
            }
            if (stop.load()) return;
            {
                // entering a job and setting one up exclude each other (m): a helper that wakes late, for a job that is already
                // done, must not read npieces / fn while run() writes the next job's
                std::lock_guard<std::mutex> lk(m);
                if (gen.load() == seen) continue;
                seen = gen.load();
                inside.fetch_add(1);
            }
            pieces();
            inside.fetch_sub(1);
        }
    }

   public:
    XferTeam(unsigned n, int device) {
        for (unsigned i = 0; i < n; i++) {
            try { th.emplace_back([this, device] { worker(device); }); } catch (const std::system_error&) { break; }     // fewer helpers, never a failed transfer
        }
    }
    ~XferTeam() {
        stop.store(true);
        { std::lock_guard<std::mutex> lk(m); }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
    size_t size() const { return th.size(); }
    // run f(0 .. n-1) on the team; `while_waiting` is called by the calling thread in a loop until every piece is done (the DMA
    // submitter of an upload), or NULL: the caller takes pieces itself
    void run(size_t n, const std::function<void(size_t)>& f, const std::function<void()>* while_waiting) {
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            if (inside.load() != 0) { lk.unlock(); std::this_thread::yield(); continue; }     // stragglers of the previous job are still leaving
            fn = &f; npieces = n;
            next.store(0); done.store(0);
            gen.fetch_add(1, std::memory_order_release);
            break;
        }
        cv.notify_all();
        if (!while_waiting || th.empty()) pieces();
        while (done.load(std::memory_order_acquire) < n) {
            if (while_waiting) (*while_waiting)(); else std::this_thread::yield();
        }
        while (inside.load() != 0) std::this_thread::yield();       // f and the caller's frame may go now
    }
};

struct ZkXfer {
    char* ring = nullptr;
    hipStream_t st = nullptr;                     // the DMA stream (never the context stream: copies overlap nothing on it)
    hipEvent_t ev[XF_MAXEV] = {};
    hipEvent_t fence = nullptr;
    std::unique_ptr<XferTeam> team;
    std::atomic<uint32_t> filled[XF_NPIECE];      // upload: piece p of the round is in the ring
    std::atomic<int> err{(int)hipSuccess};
};

int xfer_get(zk_ctx* ctx, ZkXfer** out) {
    if (!ctx->xfer) {
        ZkXfer* x = new ZkXfer();
        ctx->xfer = x;                            // (zk_xfer_free releases whatever was created)
        const unsigned hc = std::thread::hardware_concurrency();
        unsigned threads = hc >= 32 ? 8u : hc >= 8 ? 3u : 1u;
        if (hipHostMalloc((void**)&x->ring, XF_RING, hipHostMallocDefault) != hipSuccess) {
            x->ring = nullptr;
            ZK_FAIL(ctx, ZK_ERR_NOMEM, "host transfer ring: hipHostMalloc failed");
        }
        ZK_HIP(ctx, zk_stream_create(&x->st, false));
        for (size_t i = 0; i < XF_MAXEV; i++) ZK_HIP(ctx, hipEventCreateWithFlags(&x->ev[i], hipEventDisableTiming));
        ZK_HIP(ctx, hipEventCreateWithFlags(&x->fence, hipEventDisableTiming));
        x->team.reset(new XferTeam(threads, ctx->device));
    }
    *out = (ZkXfer*)ctx->xfer;
    if (!(*out)->ring || !(*out)->fence || !(*out)->team) ZK_FAIL(ctx, ZK_ERR_STATE, "host transfer ring: not initialised (an earlier allocation failed)");
    return ZK_OK;
}

// the DMA stream starts behind whatever the context stream holds now (the destination may still be read by an earlier kernel),
// and with an idle ring: the chunks of the PREVIOUS transfer may still be on their way (two uploads back to back: the operands of
// batch_product_in_place) -- their pieces must not be refilled under the DMA engine
int fence_in(zk_ctx* ctx, ZkXfer* x) {
    ZK_HIP(ctx, hipStreamSynchronize(x->st));
    ZK_HIP(ctx, hipEventRecord(x->fence, ctx->stream));
    ZK_HIP(ctx, hipStreamWaitEvent(x->st, x->fence, 0));
    return ZK_OK;
}
// ... and the context stream continues behind the DMA stream
int fence_out(zk_ctx* ctx, ZkXfer* x) {
    ZK_HIP(ctx, hipEventRecord(x->fence, x->st));
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, x->fence, 0));
    return ZK_OK;
}

// Transfers below XF_SMALL: through one of 16 page-locked slots.  (They used to be handed to the runtime as pageable copies: mostly
// 0.06 ms per 32 KB transform call, but on some boxes / allocator states 0.26 ms -- the runtime's own staging decisions; a copy into
// our own slot and a DMA from it is the same every time.)
constexpr unsigned XS_SLOTS = 16;
struct ZkXferSmall {
    char* buf = nullptr;
    hipEvent_t ev[XS_SLOTS] = {};
    bool used[XS_SLOTS] = {};
    unsigned next = 0;
};
int small_get(zk_ctx* ctx, ZkXferSmall** out) {
    if (!ctx->xfer_small) {
        ZkXferSmall* x = new ZkXferSmall();
        ctx->xfer_small = x;
        if (hipHostMalloc((void**)&x->buf, XS_SLOTS * XF_SMALL, hipHostMallocDefault) != hipSuccess) {
            x->buf = nullptr;
            ZK_FAIL(ctx, ZK_ERR_NOMEM, "host transfer slots: hipHostMalloc failed");
        }
        for (unsigned i = 0; i < XS_SLOTS; i++) ZK_HIP(ctx, hipEventCreateWithFlags(&x->ev[i], hipEventDisableTiming));
    }
    *out = (ZkXferSmall*)ctx->xfer_small;
    if (!(*out)->buf || !(*out)->ev[XS_SLOTS - 1]) ZK_FAIL(ctx, ZK_ERR_STATE, "host transfer slots: not initialised (an earlier allocation failed)");
    return ZK_OK;
}
// the next slot, free of its previous transfer
int small_slot(zk_ctx* ctx, ZkXferSmall* x, unsigned* slot) {
    const unsigned s = x->next++ % XS_SLOTS;
    if (x->used[s]) ZK_HIP(ctx, hipEventSynchronize(x->ev[s]));
    x->used[s] = false;
    *slot = s;
    return ZK_OK;
}

}  // namespace

bool zk_host_is_pinned(const void* host) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, host) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

void zk_xfer_free(zk_ctx* ctx) {
    if (ZkXferSmall* xs = (ZkXferSmall*)ctx->xfer_small) {
        for (auto& e : xs->ev) if (e) (void)hipEventDestroy(e);
        if (xs->buf) (void)hipHostFree(xs->buf);
        delete xs;
        ctx->xfer_small = nullptr;
    }
    ZkXfer* x = (ZkXfer*)ctx->xfer;
    if (!x) return;
    x->team.reset();
    if (x->st) { (void)hipStreamSynchronize(x->st); (void)hipStreamDestroy(x->st); }
    for (auto& e : x->ev) if (e) (void)hipEventDestroy(e);
    if (x->fence) (void)hipEventDestroy(x->fence);
    if (x->ring) (void)hipHostFree(x->ring);
    delete x;
    ctx->xfer = nullptr;
}

// dev[0 .. bytes) <- host.  On return the host buffer has been read completely; the context stream waits for the last DMA.
// pinned: the caller's own page-locked memory (zk_host_alloc): one DMA, in place.
int zk_xfer_h2d(zk_ctx* ctx, void* dev, const void* host, size_t bytes, bool pinned) {
    if (!bytes) return ZK_OK;
    if (pinned) {
        ZK_HIP(ctx, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
        return ZK_OK;
    }
    const char* src = (const char*)host;
    const ZkXferFill fill = [src](char* dst, size_t off, size_t len) { copy_to_ring(dst, src + off, len); };
    return zk_xfer_h2d_fn(ctx, dev, bytes, fill, true, nullptr);
}

// The same with the bytes PRODUCED piece by piece: fill(dst, off, len) writes bytes [off, off + len) of the logical vector into
// page-locked memory at dst (the team's threads call it side by side, one piece each: 256 KiB, the last one shorter) -- a gather
// out of the caller's own struct layout (MpcField / MpcGroup wrappers, GroupAffine with its flag) costs no extra pass.
// fence_ctx = false: the transfer is on the DMA stream ALONE (no wait for the context stream in front, none behind): for a
// destination nothing on the context stream touches -- the verification copy of a cached table, which runs UNDER the MSM that
// uses the cached one; the caller waits through zk_xfer_stream().  after_round(r0, rb, st): work to enqueue on the DMA stream
// behind a round's copies (bytes [r0, r0 + rb) have landed when it runs).
int zk_xfer_h2d_fn(zk_ctx* ctx, void* dev, size_t bytes, const ZkXferFill& fill_fn, bool fence_ctx,
                   const std::function<int(size_t, size_t, hipStream_t)>* after_round) {
    if (!bytes) return ZK_OK;
    if (bytes < XF_SMALL && fence_ctx && !after_round) {
        ZkXferSmall* xs;
        unsigned slot;
        ZK_TRY(small_get(ctx, &xs));
        ZK_TRY(small_slot(ctx, xs, &slot));
        char* p = xs->buf + (size_t)slot * XF_SMALL;
        fill_fn(p, 0, bytes);
        ZK_HIP(ctx, hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(ctx, hipEventRecord(xs->ev[slot], ctx->stream));
        xs->used[slot] = true;
        return ZK_OK;
    }
    ZkXfer* x;
    ZK_TRY(xfer_get(ctx, &x));
    for (size_t r0 = 0; r0 < bytes; r0 += XF_RING) {                  // rounds of one ring each (a second round only beyond 2^21 elements)
        const size_t rb = std::min(XF_RING, bytes - r0), np = (rb + XF_PIECE - 1) / XF_PIECE;
        if (fence_ctx) ZK_TRY(fence_in(ctx, x));
        else ZK_HIP(ctx, hipStreamSynchronize(x->st));               // the ring is idle again (the previous round's chunks have left it)
        for (size_t p = 0; p < np; p++) x->filled[p].store(0, std::memory_order_relaxed);
        x->err.store((int)hipSuccess);
        char* dst = (char*)dev + r0;
        const std::function<void(size_t)> fill = [&](size_t p) {
            const size_t off = p * XF_PIECE, len = std::min(XF_PIECE, rb - off);
            fill_fn(x->ring + off, r0 + off, len);
            x->filled[p].store(1, std::memory_order_release);
        };
        // the submitter: DMA for the filled prefix -- 1 MiB first, so that the engine starts early, then 2, then 4 MiB at a time: few
        // API calls in all, and what is left for the end is one 4 MiB copy (doubling all the way left HALF the vector for it)
        size_t sent = 0, want = 4;
        const std::function<void()> submit = [&] {
            size_t ready = sent;
            while (ready < np && x->filled[ready].load(std::memory_order_acquire)) ready++;
            if (ready - sent >= std::min(want, np - sent) && ready > sent && x->err.load() == (int)hipSuccess) {
                const size_t off = sent * XF_PIECE, len = std::min(ready * XF_PIECE, rb) - off;
                const hipError_t e = hipMemcpyAsync(dst + off, x->ring + off, len, hipMemcpyHostToDevice, x->st);
                if (e != hipSuccess) x->err.store((int)e);
                sent = ready;
                want = std::min<size_t>(want * 2, 16);
            } else {
                std::this_thread::yield();
            }
        };
        x->team->run(np, fill, &submit);
        while (sent < np && x->err.load() == (int)hipSuccess) submit();
        ZK_HIP(ctx, (hipError_t)x->err.load());
        if (after_round) ZK_TRY((*after_round)(r0, rb, x->st));
        if (fence_ctx) ZK_TRY(fence_out(ctx, x));
    }
    return ZK_OK;
}
// the DMA stream of the context's ring (created on first use): what a fence_ctx = false transfer is waited for on
int zk_xfer_stream(zk_ctx* ctx, hipStream_t* st) {
    ZkXfer* x;
    ZK_TRY(xfer_get(ctx, &x));
    *st = x->st;
    return ZK_OK;
}

// host <- dev[0 .. bytes), behind everything the context stream holds.  Returns when the host buffer is complete.
int zk_xfer_d2h(zk_ctx* ctx, void* host, const void* dev, size_t bytes, bool pinned) {
    if (!bytes) return ZK_OK;
    if (pinned) {
        ZK_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return ZK_OK;
    }
    char* dst = (char*)host;
    const ZkXferDrain drain = [dst](const char* src, size_t off, size_t len) { memcpy(dst + off, src, len); };
    return zk_xfer_d2h_fn(ctx, dev, bytes, drain);
}

// The same with the bytes CONSUMED piece by piece: drain(src, off, len) takes bytes [off, off + len) of the device vector out of
// page-locked memory at src (a scatter into the caller's own struct layout).
int zk_xfer_d2h_fn(zk_ctx* ctx, const void* dev, size_t bytes, const ZkXferDrain& drain_fn) {
    if (!bytes) return ZK_OK;
    if (bytes < XF_SMALL) {
        ZkXferSmall* xs;
        unsigned slot;
        ZK_TRY(small_get(ctx, &xs));
        ZK_TRY(small_slot(ctx, xs, &slot));
        char* p = xs->buf + (size_t)slot * XF_SMALL;
        ZK_HIP(ctx, hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        drain_fn(p, 0, bytes);
        return ZK_OK;
    }
    ZkXfer* x;
    ZK_TRY(xfer_get(ctx, &x));
    for (size_t r0 = 0; r0 < bytes; r0 += XF_RING) {
        const size_t rb = std::min(XF_RING, bytes - r0), np = (rb + XF_PIECE - 1) / XF_PIECE;
        // (a host-side wait for the kernels in front: an event between the two streams costs the engine ~0.13 ms before it starts)
        ZK_HIP(ctx, hipStreamSynchronize(x->st));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        x->err.store((int)hipSuccess);
        // DMA chunks: 2 MiB first (the drain starts early), then 8 MiB (every chunk costs the engine ~30 us of start-up), 4 MiB last
        // (what the team drains after the engine is through); an event behind each
        size_t cpieces[XF_MAXEV + 1], nchunks = 0, p0 = 0, sz = 8;
        while (p0 < np) {
            cpieces[nchunks] = p0;
            size_t take = std::min(sz, np - p0);
            if (np - p0 > 16 && np - p0 - take < 16) take = np - p0 - 16;             // leave 4 MiB for the last chunk
            if (nchunks + 1 == XF_MAXEV) take = np - p0;
            const size_t off = p0 * XF_PIECE, len = std::min((p0 + take) * XF_PIECE, rb) - off;
            ZK_HIP(ctx, hipMemcpyAsync(x->ring + off, (const char*)dev + r0 + off, len, hipMemcpyDeviceToHost, x->st));
            ZK_HIP(ctx, hipEventRecord(x->ev[nchunks], x->st));
            p0 += take;
            nchunks++;
            sz = 32;
        }
        cpieces[nchunks] = np;
        std::atomic<size_t> arrived{0};           // chunks known to have landed
        const std::function<void(size_t)> drain = [&](size_t p) {
            size_t c = 0;
            while (cpieces[c + 1] <= p) c++;
            while (arrived.load(std::memory_order_acquire) <= c) {           // one waiter per chunk calls into the runtime, the others watch the counter
                size_t a = arrived.load();
                if (x->err.load() != (int)hipSuccess) return;
                if (a <= c) {
                    const hipError_t e = hipEventSynchronize(x->ev[a]);
                    if (e != hipSuccess) { x->err.store((int)e); return; }
                    arrived.compare_exchange_strong(a, a + 1);
                }
            }
            const size_t off = p * XF_PIECE, len = std::min(XF_PIECE, rb - off);
            drain_fn(x->ring + off, r0 + off, len);
        };
        x->team->run(np, drain, nullptr);
        ZK_HIP(ctx, (hipError_t)x->err.load());
    }
    return ZK_OK;
}
