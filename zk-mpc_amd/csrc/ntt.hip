// ntt.hip -- radix-2 NTT over BLS12-377 Fr for gfx950.
//
// Replaces Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place
// (arkworks/algebra/poly/src/domain/radix2/mod.rs:98-114, radix2/fft.rs:22-70,185-307,
//  domain/mod.rs:92-157) and the domain constants of radix2/mod.rs:51-82.
//
// Same transform, different schedule.  The reference runs log2(N) butterfly sweeps over the whole
// vector (DIF "io" then a bit-reversal, or bit-reversal then DIT "oi").  Here the log2(N) DIF
// levels are grouped into ceil(log2(N)/10) passes; a pass gives each workgroup a tile of
// M x C elements (M = 2^logM points of C neighbouring sub-transforms, C*32 B contiguous runs),
// keeps the tile in LDS (limb-major unpacked 29-bit limbs, conflict-free b32 accesses) for its logM levels,
// applies the inter-pass twiddle w_B^(l*q) on the way out, and writes the tile back in place.
// A final kernel does the bit-reversal (as an in-place swap) fused with the iFFT's 1/N or the
// coset iFFT's g^-i/N scaling; the coset FFT's g^i scaling is fused into the first pass's load.
// The inverse transform uses the same kernels with the twiddle index negated (w^-e = w^(N-e)).
//
// Data stays in the reference's Montgomery form (R = 2^256); table entries are kept in the
// device's internal form (x * 2^261), so mmul(data, table) = data * x with no conversion.
// Work: (N/2) log2 N butterflies (the last level of every pass multiplies by 1 and is skipped),
// + N inter-pass twiddles per pass boundary.  Traffic: 2 * 32 B * N per pass + the swap pass.
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "frlazy.cuh"
#include "internal.hpp"

using namespace zk;

struct zk_domain {
    uint32_t log_n = 0;
    uint32_t* tw = nullptr;     // w^i, i < N          (internal form, nine 29-bit limbs each: tab_load)
    uint32_t* cos = nullptr;    // g^i                  (coset FFT pre-scale)
    uint32_t* icos = nullptr;   // g^-i / N             (coset iFFT post-scale)
    Fr size_inv;                // 1/N, internal form
    Fr zinv;                    // 1 / (g^N - 1), internal form (divide_by_vanishing_poly_on_coset)
};

namespace {

constexpr int LOGM_MAX = 10;
constexpr int NTT_THREADS = 1024;   // (512 / 256 threads per tile, two / four butterflies per lane and stage: 2^20 in 131 / 144 us against 126: round 5)
constexpr int TILE_ELEMS = 4096;  // x 36 B = 144 KiB of LDS

struct FrK { uint32_t l[9]; };
__device__ __forceinline__ Fr frk(const FrK& k) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = k.l[i];
    return r;
}
FrK to_frk(const Fr& a) {
    FrK k;
    for (int i = 0; i < 9; i++) k.l[i] = a.l[i];
    return k;
}

// Table entries (twiddles, coset scales) are kept as their nine 29-bit limbs, 36 B each: a product takes them as they are
// (re-slicing eight packed words into nine limbs was ~25 instructions in front of every twiddle product).
struct __attribute__((packed, aligned(4))) Limbs9 { uint32_t l[9]; };
__device__ __forceinline__ Fr tab_load(const uint32_t* tab, size_t i) {
    const Limbs9 t = reinterpret_cast<const Limbs9*>(tab)[i];
    Fr r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = t.l[k];
    return r;
}
__device__ __forceinline__ void tab_store(uint32_t* tab, size_t i, const Fr& v) {
    Limbs9 t;
#pragma unroll
    for (int k = 0; k < 9; k++) t.l[k] = v.l[k];
    reinterpret_cast<Limbs9*>(tab)[i] = t;
}

// out[i] = start * base^i  (internal form in, limb-form table out)
__global__ void __launch_bounds__(256) k_powers(uint32_t* out, FrK base_k, FrK start_k, size_t n) {
    constexpr int CH = 32;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t i0 = t * CH;
    if (i0 >= n) return;
    Fr base = frk(base_k);
    // base^i0 by square-and-multiply
    Fr p = fp_one<FrParams>();
    bool started = false;
    for (int b = 63; b >= 0; b--) {
        if (started) p = fp_sqr<FrParams>(p);
        if ((i0 >> b) & 1) { p = started ? fr_mul(p, base) : base; started = true; }
    }
    p = fr_mul(p, frk(start_k));
    for (int j = 0; j < CH && i0 + j < n; j++) {
        tab_store(out, i0 + j, p);
        p = fr_mul(p, base);
    }
}

// LDS tile layout: limb-major, 9 x 29-bit limbs per element (36 B): lds[k * E + element].  Limb-major makes every
// ds_read/ds_write_b32 of a wave hit 64 consecutive words (conflict-free); keeping the unpacked limbs avoids a
// pack + unpack (~100 instructions) per element per butterfly level.  4096 elements = 144 KiB.
__device__ __forceinline__ Fr lds_load(const uint32_t* lds, uint32_t E, uint32_t li) {
    Fr r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.l[k] = lds[k * E + li];
    return r;
}
__device__ __forceinline__ void lds_store(uint32_t* lds, uint32_t E, uint32_t li, const Fr& v) {
#pragma unroll
    for (int k = 0; k < 9; k++) lds[k * E + li] = v.l[k];
}

#define ZK_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// What a pass reads and writes in global memory.  Tile = C columns x M points; column id = block * C + c; element (m, c)
// of the tile is data[((col >> logS) << logB) + (m << logS) + (col & (S - 1))] (c fastest: C * 32 B contiguous runs).
// FINAL (last pass of a multi-pass transform, logS == 0): the element lands directly at its bit-reversed position in `out` (a
// different buffer: the scattered stores would otherwise overwrite tiles that have not been read yet), multiplied by the
// post-scale (POST 1: constant k = 1/N; POST 2: table post[dst] = g^-dst / N) -- this replaces a separate permutation pass.
template <int POST>
struct PassIO {
    const uint32_t* data;
    uint32_t* out;
    const uint32_t* __restrict__ tw;
    const uint32_t* __restrict__ pre;
    const uint32_t* __restrict__ post;
    uint32_t log_n, logS, logM, logB, N1, col0;
    int inverse, final_rev;

    __device__ __forceinline__ uint32_t index(uint32_t m, uint32_t c) const {
        const uint32_t col = col0 + c;
        return ((col >> logS) << logB) + (m << logS) + (col & ((1u << logS) - 1));
    }
    // canonical, or < 1.03 r from the pass before; the coset pre-scale g^idx rides on the first pass's load
    __device__ __forceinline__ Fr load(uint32_t m, uint32_t c) const {
        const uint32_t idx = index(m, c);
        Fr v = fr_load(data, idx);
        if (pre) v = frl_mul(v, tab_load(pre, idx));
        return v;
    }
    // v: any lazy value (wide limbs allowed).  A pass that is not the last multiplies by the inter-pass twiddle w_B^(l q),
    // q = bitrev(m), and hands on a value below 1.03 r (it fits the 256-bit words); the last one hands back canonical values.
    __device__ __forceinline__ void emit(uint32_t m, uint32_t c, Fr v, const FrK& post_k) const {
        const uint32_t idx = index(m, c);
        uint32_t dst = idx;
        if (logS) {
            const uint32_t l = (col0 + c) & ((1u << logS) - 1);
            const uint32_t q = __brev(m) >> (32 - logM);
            uint32_t ex = (l * q) << (log_n - logB);
            if (inverse) ex = (0u - ex) & N1;
            v = frl_mul(v, tab_load(tw, ex));
        } else if (final_rev && POST) {
            dst = __brev(idx) >> (32 - log_n);
            const Fr t = frl_mul(v, POST == 1 ? frk(post_k) : tab_load(post, dst));
            v = fp_reduce_once<FrParams>(t.l);
        } else {
            if (final_rev) dst = __brev(idx) >> (32 - log_n);
            v = frl_canon(v);
        }
        fr_store(out, dst, v);
    }
};

// One radix-4 butterfly of a stage (frlazy.cuh: frl_radix4, here with its loads and stores placed so that at most two tile
// elements, two sums / products and one twiddle are live at a time: four waves per SIMD without spills).  The tile elements are
// (m0 + k gB, c), k < 4.
// FIRST: the pass's first stage reads global memory itself (no load phase, no LDS round trip: the waves whose data arrives
// start their products while the others still wait).  LAST: the pass's last stage (its second level has twiddle 1) hands its
// four outputs, as wide sums, straight to PassIO::emit (whose product, or frl_canon, brings them back into range).
template <bool FIRST, bool LAST, int POST>
__device__ __forceinline__ void radix4_item(const PassIO<POST>& io, const FrK& post_k, uint32_t* lds, uint32_t E, uint32_t logC, uint32_t m0,
                                            uint32_t gB, uint32_t c, uint32_t ea, uint32_t eb, uint32_t ec) {
    const uint32_t i0 = (m0 << logC) | c, st = gB << logC;
    const Fr wa = tab_load(io.tw, ea), wb = tab_load(io.tw, eb);     // first: their latency runs under the reads below
    Fr s0, s1, d0, d1;
    {
        const Fr x0 = FIRST ? io.load(m0, c) : lds_load(lds, E, i0), x2 = FIRST ? io.load(m0 + 2 * gB, c) : lds_load(lds, E, i0 + 2 * st);
        s0 = frl_add(x0, x2);
        d0 = frl_mul(frl_sub<3>(x0, x2), wa);
    }
    ZK_SCHED_FENCE();
    Fr wc;
    if (!LAST) wc = tab_load(io.tw, ec);                             // lands under the second product
    {
        const Fr x1 = FIRST ? io.load(m0 + gB, c) : lds_load(lds, E, i0 + st), x3 = FIRST ? io.load(m0 + 3 * gB, c) : lds_load(lds, E, i0 + 3 * st);
        s1 = frl_add(x1, x3);
        d1 = frl_mul(frl_sub<3>(x1, x3), wb);
    }
    ZK_SCHED_FENCE();
    Fr y0 = frl_add(s0, s1), y2 = frl_add(d0, d1);
    Fr u = frl_sub<5>(s0, s1), v = frl_sub<2>(d0, d1);
    if (LAST) {
        io.emit(m0, c, y0, post_k);
        ZK_SCHED_FENCE();
        io.emit(m0 + gB, c, u, post_k);
        ZK_SCHED_FENCE();
        io.emit(m0 + 2 * gB, c, y2, post_k);
        ZK_SCHED_FENCE();
        io.emit(m0 + 3 * gB, c, v, post_k);
    } else {
        lds_store(lds, E, i0, frl_reduce(y0));
        lds_store(lds, E, i0 + 2 * st, frl_norm(y2));
        ZK_SCHED_FENCE();
        u = frl_mul(u, wc);
        ZK_SCHED_FENCE();
        v = frl_mul(v, wc);
        lds_store(lds, E, i0 + st, u);
        lds_store(lds, E, i0 + 3 * st, v);
    }
}

// One DIF pass: logM levels of the size-M transforms of a tile, as radix-4 register butterflies in the lazy domain
// (frlazy.cuh) with the tile in LDS between stages; one radix-2 level comes first when logM is odd.  Twiddle exponents in
// units of w = w_N: level s of the size-M transform uses w_M^(jj 2^s) = w^((jj << s) << (log_n - logM)).
// blockIdx.y picks one of up to NTT_BATCH same-size transforms: the witness map's three inverse and three coset transforms are
// one launch per pass (a pass is one 147-KB tile per CU: the transforms of a batch follow each other on a CU without a launch
// boundary, its drain and its ramp, between them).
constexpr int NTT_BATCH = 4;
struct NttBatch {
    const uint32_t* src[NTT_BATCH];
    uint32_t* dst[NTT_BATCH];
};
template <int POST>
__global__ void __launch_bounds__(NTT_THREADS)
k_ntt_pass(NttBatch nb, const uint32_t* __restrict__ tw,
           const uint32_t* __restrict__ pre, uint32_t log_n, uint32_t logS, uint32_t logM, uint32_t logC, int inverse,
           int final_rev, FrK post_k, const uint32_t* __restrict__ post) {
    extern __shared__ uint32_t lds[];
    const uint32_t* data = nb.src[blockIdx.y];
    uint32_t* out = nb.dst[blockIdx.y];
    const uint32_t C = 1u << logC, E = C << logM;
    const uint32_t N1 = (1u << log_n) - 1;
    const uint32_t tid = threadIdx.x, NT = blockDim.x;
    // neighbouring tiles share cache lines when C * 32 B < 128 B: blocks are dealt round-robin over the 8 XCDs, so block b takes
    // tile (b % 8) * (tiles / 8) + b / 8 -- the tiles of one line then sit on one XCD (one L2), next to each other in time
    uint32_t tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const PassIO<POST> io{data, out, tw, pre, post, log_n, logS, logM, logS + logM, N1, tile << logC, inverse, final_rev};
    const uint32_t tsh = log_n - logM;
    const uint32_t nst = (logM + 1) >> 1;       // stages: the last one stores to global memory, the first one loads from it
    uint32_t s = 0, stage = 0;
    if (logM & 1) {
        const uint32_t g = 1u << (logM - 1);
        for (uint32_t b = tid; b < (E >> 1); b += NT) {
            const uint32_t c = b & (C - 1), j = b >> logC;        // j < g: the only block of this level
            uint32_t ex = j << tsh;
            if (inverse) ex = (0u - ex) & N1;
            const Fr w = tab_load(tw, ex);
            Fr x = io.load(j, c), y = io.load(j + g, c);
            frl_radix2(x, y, w);
            if (nst == 1) {
                io.emit(j, c, x, post_k);
                io.emit(j + g, c, y, post_k);
            } else {
                lds_store(lds, E, (j << logC) | c, x);
                lds_store(lds, E, ((j + g) << logC) | c, y);
            }
        }
        if (nst > 1) __syncthreads();
        s = 1;
        stage = 1;
    }
    for (; s < logM; s += 2, stage++) {
        const uint32_t lgA = logM - 1 - s, lgB = lgA - 1, gB = 1u << lgB;
        const uint32_t quarter = 1u << (log_n - 2);
        const bool first = stage == 0, last = stage + 1 == nst;
        for (uint32_t b = tid; b < (E >> 2); b += NT) {
            const uint32_t c = b & (C - 1), j = b >> logC;
            const uint32_t jj = j & (gB - 1);
            const uint32_t m0 = ((j >> lgB) << (lgA + 1)) | jj;
            uint32_t ea = (jj << s) << tsh, eb = ea + quarter, ec = ea << 1;
            if (inverse) { ea = (0u - ea) & N1; eb = (0u - eb) & N1; ec = (0u - ec) & N1; }
            if (first && last) radix4_item<true, true>(io, post_k, lds, E, logC, m0, gB, c, ea, eb, ec);
            else if (first) radix4_item<true, false>(io, post_k, lds, E, logC, m0, gB, c, ea, eb, ec);
            else if (last) radix4_item<false, true>(io, post_k, lds, E, logC, m0, gB, c, ea, eb, ec);
            else radix4_item<false, false>(io, post_k, lds, E, logC, m0, gB, c, ea, eb, ec);
        }
        if (!last) __syncthreads();
    }
}

// In-place bit reversal fused with the post-scale: MODE 0 none, 1 constant k, 2 table post[i].
template <int MODE>
__global__ void __launch_bounds__(256) k_bitrev_scale(uint32_t* data, uint32_t log_n, FrK k, const uint32_t* post) {
    const size_t n = (size_t)1 << log_n;
    const Fr kk = frk(k);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = log_n ? (size_t)(__brev((uint32_t)i) >> (32 - log_n)) : 0;
        if (i > r) continue;
        Fr a = fr_load(data, i);
        if (i == r) {
            if (MODE == 1) a = fr_mul(a, kk);
            if (MODE == 2) a = fr_mul(a, tab_load(post, i));
            if (MODE) fr_store(data, i, a);
        } else {
            Fr b = fr_load(data, r);
            if (MODE == 1) { a = fr_mul(a, kk); b = fr_mul(b, kk); }
            if (MODE == 2) { a = fr_mul(a, tab_load(post, r)); b = fr_mul(b, tab_load(post, i)); }
            fr_store(data, i, b);
            fr_store(data, r, a);
        }
    }
}

Fr host_pow_u64(const Fr& a, uint64_t e) {
    Fr r = fp_one<FrParams>();
    for (int b = 63; b >= 0; b--) {
        r = fp_sqr<FrParams>(r);
        if ((e >> b) & 1) r = fp_mul<FrParams>(r, a);
    }
    return r;
}

int build_powers(zk_ctx* ctx, uint32_t** out, const Fr& base, const Fr& start, size_t n) {
    ZK_HIP(ctx, hipMalloc((void**)out, n * 36));
    size_t threads = (n + 31) / 32;
    hipLaunchKernelGGL(k_powers, (unsigned)((threads + 255) / 256), 256, 0, ctx->stream, *out, to_frk(base), to_frk(start), n);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

int get_domain(zk_ctx* ctx, uint32_t log_n, bool need_coset, zk_domain** out) {
    if (log_n > (uint32_t)FR_TWO_ADICITY || log_n > 28) ZK_FAIL(ctx, ZK_ERR_ARG, "NTT size unsupported (log_n > 28)");
    zk_domain* d = nullptr;
    auto it = ctx->domains.find(log_n);
    if (it != ctx->domains.end()) d = it->second;
    const size_t n = (size_t)1 << log_n;
    if (!d) {
        d = new zk_domain();
        d->log_n = log_n;
        // group_gen = two_adic_root^(2^(47 - log_n))   (ff get_root_of_unity; radix2/mod.rs:67-69)
        Fr w = fp_const<FrParams>(FrParams::TWO_ADIC_ROOT);
        for (uint32_t i = 0; i < (uint32_t)FR_TWO_ADICITY - log_n; i++) w = fp_sqr<FrParams>(w);
        Fr n_int = fp_canon_to_int<FrParams>([&] { Fr t = fp_zero<FrParams>(); t.l[0] = (uint32_t)(n & MASK29); t.l[1] = (uint32_t)(n >> 29); return t; }());
        d->size_inv = fp_inv<FrParams>(n_int);
        Fr g = fp_const<FrParams>(FrParams::GENERATOR);
        Fr gn = host_pow_u64(g, n);
        d->zinv = fp_inv<FrParams>(fp_sub<FrParams>(gn, fp_one<FrParams>()));
        ZK_TRY(build_powers(ctx, &d->tw, w, fp_one<FrParams>(), n));
        ctx->domains[log_n] = d;
    }
    if (need_coset && !d->cos) {
        Fr g = fp_const<FrParams>(FrParams::GENERATOR);
        Fr gi = fp_const<FrParams>(FrParams::GENERATOR_INV);
        ZK_TRY(build_powers(ctx, &d->cos, g, fp_one<FrParams>(), n));
        ZK_TRY(build_powers(ctx, &d->icos, gi, d->size_inv, n));
    }
    *out = d;
    return ZK_OK;
}

}  // namespace

void zk_domains_free(zk_ctx* ctx) {
    for (auto& kv : ctx->domains) {
        zk_domain* d = kv.second;
        if (d->tw) (void)hipFree(d->tw);
        if (d->cos) (void)hipFree(d->cos);
        if (d->icos) (void)hipFree(d->icos);
        delete d;
    }
    ctx->domains.clear();
}

// `count` (<= NTT_BATCH) transforms of the same size and kind, in place on bufs[k], as one launch per pass.
int zk_ntt_launch_batch(zk_ctx* ctx, void* const* bufs, int count, uint32_t log_n, int inverse, int coset) {
    if (count < 1 || count > NTT_BATCH) ZK_FAIL(ctx, ZK_ERR_ARG, "ntt: batch of 1..4 transforms");
    zk_domain* d;
    ZK_TRY(get_domain(ctx, log_n, coset != 0, &d));
    FrK zero{};
    const int post_mode = !inverse ? 0 : (!coset ? 1 : 2);
    const FrK post_k = post_mode == 1 ? to_frk(d->size_inv) : zero;
    const uint32_t* post_tab = post_mode == 2 ? d->icos : nullptr;
    auto launch = [&](const NttBatch& nb, uint32_t tiles, uint32_t nt, uint32_t E, const uint32_t* pre, uint32_t logS,
                      uint32_t logM, uint32_t logC, int final_rev) {
        const dim3 grid(tiles, (unsigned)count);
        if (post_mode == 0)
            hipLaunchKernelGGL(k_ntt_pass<0>, grid, nt, E * 36, ctx->stream, nb, d->tw, pre, log_n, logS, logM, logC, inverse, final_rev, post_k, post_tab);
        else if (post_mode == 1)
            hipLaunchKernelGGL(k_ntt_pass<1>, grid, nt, E * 36, ctx->stream, nb, d->tw, pre, log_n, logS, logM, logC, inverse, final_rev, post_k, post_tab);
        else
            hipLaunchKernelGGL(k_ntt_pass<2>, grid, nt, E * 36, ctx->stream, nb, d->tw, pre, log_n, logS, logM, logC, inverse, final_rev, post_k, post_tab);
    };
    bool permuted = false;      // the last pass already wrote the natural order (and the post-scale)
    if (log_n > 0) {
        uint32_t passes = (log_n + LOGM_MAX - 1) / LOGM_MAX;
        uint32_t base = log_n / passes, extra = log_n % passes;
        uint32_t remaining = log_n;
        if (!ctx->flags["ntt_lds"]) {
            ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass<0>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_ELEMS * 36));
            ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass<1>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_ELEMS * 36));
            ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass<2>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_ELEMS * 36));
            ctx->flags["ntt_lds"] = 1;
        }
        // with two or more passes the first one writes a scratch buffer and the last one scatters from it back into the
        // caller's buffer in natural order; a single-pass transform (N <= 2^LOGM_MAX) keeps the separate permutation
        uint32_t* tmp = nullptr;
        const size_t words = ((size_t)1 << log_n) * 8;
        if (passes >= 2) ZK_TRY(zk_scratch(ctx, "ntt_tmp", (size_t)count * words * 4, (void**)&tmp));
        for (uint32_t p = 0; p < passes; p++) {
            uint32_t logM = base + (p < extra ? 1 : 0);
            uint32_t logS = remaining - logM;
            // tile elements: 4096 for large transforms; small ones are cut into >= 256 tiles so that every CU gets work (a
            // thread runs one radix-4 butterfly per stage either way; one wave alone issues a multiply-add every ~10 cycles,
            // so a pass takes the same ~20 us whether a CU hosts one wave or four -- tools/ubench_chain.hip)
            // Three-pass transforms (N > 2^20) run 1024-element tiles (four independent blocks per CU: one block's loads and
            // stores run under the others' butterflies; 2^24: 2.8 -> 2.2 ms); 10-level passes need 4096 to keep 128-B runs.
            uint32_t logE = log_n < 20 ? (log_n > 8 ? log_n - 8 : 0) : (passes >= 3 ? 10 : 12);
            if (logE < base + (extra ? 1 : 0)) logE = base + (extra ? 1 : 0);
            if (passes == 1) logE = log_n;
            static const uint32_t env_loge = getenv("ZK_NTT_LOGE") ? (uint32_t)atoi(getenv("ZK_NTT_LOGE")) : 0u;      // experiment knob, read once
            if (env_loge && passes > 1 && env_loge >= base + (extra ? 1 : 0) && env_loge <= 12 && env_loge <= log_n) logE = env_loge;
            if (logE < logM || logE > 12) ZK_FAIL(ctx, ZK_ERR_STATE, "ntt: tile smaller than a pass's transform (logE < logM) or larger than the LDS tile");
            uint32_t logC = logE - logM;
            uint32_t E = 1u << logE;
            uint32_t nt = E / 4 > NTT_THREADS ? NTT_THREADS : (E / 4 < 64 ? 64 : E / 4);
            uint32_t tiles = 1u << (log_n - logE);
            const uint32_t* pre = (p == 0 && coset && !inverse) ? d->cos : nullptr;
            const bool last = p + 1 == passes;
            NttBatch nb{};
            for (int k = 0; k < count; k++) {
                uint32_t* data = (uint32_t*)bufs[k];
                uint32_t* t = tmp ? tmp + (size_t)k * words : nullptr;
                nb.src[k] = (t && p > 0) ? t : data;
                nb.dst[k] = t ? (last ? data : t) : data;
            }
            launch(nb, tiles, nt, E, pre, logS, logM, logC, (tmp && last) ? 1 : 0);
            ZK_HIP(ctx, hipGetLastError());
            remaining = logS;
        }
        permuted = tmp != nullptr;
    }
    if (!permuted) {
        unsigned g = zk_grid((size_t)1 << log_n, 256);
        for (int k = 0; k < count; k++) {
            uint32_t* data = (uint32_t*)bufs[k];
            if (!inverse) {
                if (log_n > 1) hipLaunchKernelGGL(k_bitrev_scale<0>, g, 256, 0, ctx->stream, data, log_n, zero, nullptr);
            } else if (!coset) {
                hipLaunchKernelGGL(k_bitrev_scale<1>, g, 256, 0, ctx->stream, data, log_n, to_frk(d->size_inv), nullptr);
            } else {
                hipLaunchKernelGGL(k_bitrev_scale<2>, g, 256, 0, ctx->stream, data, log_n, zero, d->icos);
            }
        }
        ZK_HIP(ctx, hipGetLastError());
    }
    return ZK_OK;
}

int zk_ntt_launch(zk_ctx* ctx, void* buf, uint32_t log_n, int inverse, int coset) { return zk_ntt_launch_batch(ctx, &buf, 1, log_n, inverse, coset); }

extern "C" int zk_fr_ntt_dev(zk_ctx* ctx, void* buf, uint32_t log_n, int inverse, int coset) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !buf) return ZK_ERR_ARG;
    return zk_ntt_launch(ctx, buf, log_n, inverse, coset);
    ZK_API_END
}

extern "C" int zk_fr_fft_in_place(zk_ctx* ctx, zk_fr* vec, size_t n, uint32_t log_n, int inverse, int coset) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !vec) return ZK_ERR_ARG;
    if (log_n > 28) ZK_FAIL(ctx, ZK_ERR_ARG, "NTT size unsupported (log_n > 28)");       // (before anything is sized by it)
    size_t N = (size_t)1 << log_n;
    if (n > N) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_fr_fft_in_place: n exceeds the domain size");
    void* d;
    ZK_TRY(zk_scratch(ctx, "fft_host", N * 32, &d));
    ZK_TRY(zk_xfer_h2d(ctx, d, vec, n * 32));
    if (N > n) ZK_HIP(ctx, hipMemsetAsync((char*)d + n * 32, 0, (N - n) * 32, ctx->stream));
    ZK_TRY(zk_ntt_launch(ctx, d, log_n, inverse, coset));
    zk_msm_spec_fft_begin(ctx, d, N, (inverse ? 2 : 0) + (coset ? 1 : 0));                        // (msm.hip: `h = witness_map(..)` is the H query's scalar vector next)
    ZK_TRY(zk_xfer_d2h(ctx, vec, d, N * 32));
    zk_msm_spec_fft_end(ctx, N, (inverse ? 2 : 0) + (coset ? 1 : 0), [vec](size_t n) { return zk_scalars_fingerprint(vec, n); });
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_fr_divide_by_vanishing_on_coset_dev(zk_ctx* ctx, void* evals, uint32_t log_n) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !evals) return ZK_ERR_ARG;
    zk_domain* d;
    ZK_TRY(get_domain(ctx, log_n, false, &d));
    return zk_vec_scale_launch(ctx, evals, d->zinv.l, evals, (size_t)1 << log_n);
    ZK_API_END
}

extern "C" int zk_fr_divide_by_vanishing_on_coset_in_place(zk_ctx* ctx, zk_fr* evals, uint32_t log_n) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !evals) return ZK_ERR_ARG;
    if (log_n > 28) ZK_FAIL(ctx, ZK_ERR_ARG, "NTT size unsupported (log_n > 28)");
    const size_t N = (size_t)1 << log_n;
    zk_domain* dom;
    ZK_TRY(get_domain(ctx, log_n, false, &dom));
    void* d;
    ZK_TRY(zk_scratch(ctx, "fft_host", N * 32, &d));
    ZK_TRY(zk_xfer_h2d(ctx, d, evals, N * 32));
    ZK_TRY(zk_vec_scale_launch(ctx, d, dom->zinv.l, d, N));
    ZK_TRY(zk_xfer_d2h(ctx, evals, d, N * 32));
    return ZK_OK;
    ZK_API_END
}

// used by r1cs.hip: (ab - c) / Z(g) fused
int zk_ntt_vanishing_inv(zk_ctx* ctx, uint32_t log_n, uint32_t out9[9]) {
    zk_domain* d;
    ZK_TRY(get_domain(ctx, log_n, false, &d));
    for (int i = 0; i < 9; i++) out9[i] = d->zinv.l[i];
    return ZK_OK;
}
