// fixed_base.hip -- batch fixed-base scalar multiplication out[i] = s_i * g and batch normalisation.
//
// Replaces FixedBaseMSM::{get_window_table, multi_scalar_mul} (arkworks/algebra/ec/src/msm/fixed_base.rs:11-95)
// and ProjectiveCurve::batch_normalization_into_affine (ec/src/models/short_weierstrass_jacobian.rs:536-550)
// as used by generate_parameters (arkworks/groth16/src/generator.rs:130-215).  Setup-side only: it
// produces the proving-key queries on the device (the bench's and the tests' input), it is not on
// the proving path.
//
// 8-bit windows: table[j][d] = d * 2^(8 j) * g (32 x 256 affine points), one thread per scalar does
// 32 mixed additions; normalisation is Montgomery's trick over chunks of 16 points per thread
// (one field inversion per chunk).
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "internal.hpp"

using namespace zk;

namespace {

constexpr int FB_WINDOWS = 32;  // 8-bit windows over 256 bits
constexpr int NORM_CHUNK = 16;

template <class F>
struct AffArg { uint32_t w[2 * F::WORDS]; };

template <class F>
__global__ void __launch_bounds__(64) k_fb_table(uint32_t* table, AffArg<F> gen) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FB_WINDOWS * 256) return;
    uint32_t j = t >> 8, d = t & 255;
    Affine<F> g = aff_load<F>(gen.w);
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    k[j >> 2] = d << (8 * (j & 3));
    XYZZ<F> r = xyzz_scalar_mul<F>(g, k, (int)(j >> 2) + 1);
    aff_store16<F>(table, t, xyzz_to_affine<F>(r));
}

template <class F>
__global__ void __launch_bounds__(256) k_fixed_base(const uint32_t* __restrict__ table, const void* scalars, size_t n, uint32_t* out_xyzz) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = fp_ext_to_canon<FrParams>(fr_load(scalars, i));
    uint32_t w[8];
    fp_pack<FrParams>(w, s);
    XYZZ<F> acc = xyzz_inf<F>();
#pragma unroll 1
    for (int j = 0; j < FB_WINDOWS; j++) {
        uint32_t word = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) word = (j >> 2) == q ? w[q] : word;
        uint32_t d = (word >> (8 * (j & 3))) & 255;
        if (d) acc = xyzz_madd<F>(acc, aff_load16<F>(table, (size_t)j * 256 + d));
    }
    xyzz_store16<F>(out_xyzz, i, acc);
}

// Montgomery's trick on ZZZ over a chunk; scratch holds the running products.
template <class F>
__global__ void __launch_bounds__(64) k_batch_affine(const uint32_t* in_xyzz, uint32_t* out_aff, uint32_t* scratch, size_t n) {
    using T = typename F::T;
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t lo = t * NORM_CHUNK;
    if (lo >= n) return;
    size_t hi = lo + NORM_CHUNK < n ? lo + NORM_CHUNK : n;
    T run = F::one();
    for (size_t i = lo; i < hi; i++) {
        felt_store16<F>(scratch + i * F::WORDS, run);  // product of the non-zero zzz before i
        T zzz = felt_load16<F>(in_xyzz + i * 4 * F::WORDS + 3 * F::WORDS);
        if (!F::is_zero(zzz)) run = F::mul(run, zzz);
    }
    T inv = F::inv(run);
    for (size_t i = hi; i-- > lo;) {
        XYZZ<F> p = xyzz_load16<F>(in_xyzz, i);
        if (xyzz_is_inf<F>(p)) {
            aff_store16<F>(out_aff, i, aff_inf<F>());
            continue;
        }
        T before = felt_load16<F>(scratch + i * F::WORDS);
        T zi3 = F::mul(inv, before);   // 1/zzz_i
        inv = F::mul(inv, p.zzz);
        T zi = F::mul(zi3, p.zz);      // 1/z
        T zi2 = F::sqr(zi);            // 1/zz
        aff_store16<F>(out_aff, i, Affine<F>{F::mul(p.x, zi2), F::mul(p.y, zi3)});
    }
}

Affine<G1Field> g1_generator() {
    return Affine<G1Field>{fp_const<FqParams>(FqParams::G1_GEN_X), fp_const<FqParams>(FqParams::G1_GEN_Y)};
}
Affine<G2Field> g2_generator() {
    return Affine<G2Field>{Fq2{fp_const<FqParams>(FqParams::G2_GEN_X0), fp_const<FqParams>(FqParams::G2_GEN_X1)},
                           Fq2{fp_const<FqParams>(FqParams::G2_GEN_Y0), fp_const<FqParams>(FqParams::G2_GEN_Y1)}};
}

template <class F>
int fixed_base_run(zk_ctx* ctx, const Affine<F>& gen_base, const zk_fr* gen_k, const void* scalars, size_t n, int group, zk_bases** out) {
    if (!ctx || !gen_k || !out || (n && !scalars)) return ZK_ERR_ARG;
    Fr k = fp_ext_to_canon<FrParams>(host_load_ext<FrParams>(gen_k->l));
    uint32_t kw[8];
    fp_pack<FrParams>(kw, k);
    Affine<F> g = xyzz_to_affine<F>(xyzz_scalar_mul<F>(gen_base, kw, 8));
    AffArg<F> ga;
    aff_store<F>(ga.w, g);

    zk_bases* b = new zk_bases();
    b->group = group;
    b->n = n;
    *out = b;
    if (!n) return ZK_OK;
    uint32_t *table, *xy, *scr;
    ZK_TRY(zk_scratch(ctx, "fb_table", (size_t)FB_WINDOWS * 256 * 2 * F::WORDS * 4, (void**)&table));
    ZK_TRY(zk_scratch(ctx, "fb_xyzz", n * 4 * F::WORDS * 4, (void**)&xy));
    ZK_TRY(zk_scratch(ctx, "fb_scr", n * F::WORDS * 4, (void**)&scr));
    if (hipMalloc((void**)&b->dev, n * 2 * F::WORDS * 4) != hipSuccess) { delete b; *out = nullptr; ZK_FAIL(ctx, ZK_ERR_NOMEM, "fixed base: hipMalloc failed"); }
    hipLaunchKernelGGL(k_fb_table<F>, FB_WINDOWS * 256 / 64, 64, 0, ctx->stream, table, ga);
    hipLaunchKernelGGL(k_fixed_base<F>, (unsigned)((n + 255) / 256), 256, 0, ctx->stream, table, scalars, n, xy);
    size_t chunks = (n + NORM_CHUNK - 1) / NORM_CHUNK;
    hipLaunchKernelGGL(k_batch_affine<F>, (unsigned)((chunks + 63) / 64), 64, 0, ctx->stream, xy, b->dev, scr, n);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// out[i] = 2^c * in[i]  (affine in, XYZZ out)
template <class F>
__global__ void __launch_bounds__(256) k_dbl_c(const uint32_t* in_aff, uint32_t* out_xyzz, size_t n, uint32_t c) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        XYZZ<F> p = xyzz_from_affine<F>(aff_load16<F>(in_aff, i));
        for (uint32_t k = 0; k < c; k++) p = xyzz_dbl<F>(p);
        xyzz_store16<F>(out_xyzz, i, p);
    }
}

// out[j n + i] = 2^(c (j + 1)) * in[i], j < g: the next g levels of a table's window multiples from one level, one lane per (level,
// point), c (j + 1) doublings.  For SMALL tables the level-by-level build is a chain of W latencies -- 20 levels x (20 dependent
// doublings + a field inversion on a lone wave) = 14 ms for a 1 024-point table, whatever its size -- while g levels together are one
// chain of c g doublings and ONE batched normalisation.  (g + 1) / 2 times the doublings, on a chip that is otherwise idle: tables of
// up to 2^13 points take all levels at once (~2 ms; the reference's circuits), up to 2^16 two at a time with the groups chained in
// XYZZ form, larger ones one level at a time (the work is the bound there).
// FROM_XYZZ: the level the group starts from is itself un-normalised (the previous group's last output): the groups of a table
// chain without a normalisation in between, and ONE k_batch_affine over all levels ends the build -- one inversion latency per table
// instead of one per group.
template <class F, bool FROM_XYZZ>
__global__ void __launch_bounds__(256) k_dbl_levels(const uint32_t* in, uint32_t* out_xyzz, size_t n, uint32_t c, uint32_t g) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n * g; t += (size_t)gridDim.x * blockDim.x) {
        const uint32_t j = (uint32_t)(t / n);
        const size_t i = t - (size_t)j * n;
        XYZZ<F> p = FROM_XYZZ ? xyzz_load16<F>(in, i) : xyzz_from_affine<F>(aff_load16<F>(in, i));
        for (uint32_t k = 0; k < c * (j + 1); k++) p = xyzz_dbl<F>(p);
        xyzz_store16<F>(out_xyzz, t, p);
    }
}

// packed points (2 * WORDS words each) -> one point per `stride` words; the padding words are never read
template <class F>
__global__ void __launch_bounds__(256) k_repack(const uint32_t* in, uint32_t* out, size_t n, uint32_t stride) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4* s = reinterpret_cast<const uint4*>(in + i * (2 * F::WORDS));
        uint4* d = reinterpret_cast<uint4*>(out + i * stride);
#pragma unroll
        for (int k = 0; k < 2 * F::WORDS / 4; k++) d[k] = s[k];
    }
}

// packed G1 points -> the accumulate kernel's own form, 64 words (two 128-byte lines) per point:
//   line 0: x as 13 limbs of 29 bits, y as 13 limbs  (what fp_unpack would produce: msm.hip::k_accum takes them as they are)
//   line 1: x, p - y                                   (the point of a NEGATIVE digit: the sign bit of a sorted entry picks the line)
// Infinity stays all-zero words in both lines.  An entry still costs one line of traffic; the kernel loses the unpacking (~55
// instructions per addition) and the conditional negation (~55): memory traded for instructions, 256 B per table point.
template <class F>
__global__ void __launch_bounds__(256) k_repack_limbs(const uint32_t* in, uint32_t* out, size_t n) {
    constexpr int L = sizeof(typename F::T) / sizeof(uint32_t);
    static_assert(2 * L <= 32, "x and y limbs fit one 128-byte line");
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Affine<F> p = aff_load16<F>(in, i);
        const bool inf = aff_is_inf<F>(p);
        const typename F::T ny = inf ? p.y : F::neg(p.y);
        uint32_t* d = out + i * 64;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            d[k] = k < L ? p.x.l[k] : (k < 2 * L ? p.y.l[k - L] : 0u);
            d[32 + k] = k < L ? p.x.l[k] : (k < 2 * L ? ny.l[k - L] : 0u);
        }
    }
}

// layout: 0 = the best form that fits the memory budget; 1 = packed (2 * WORDS words per point); 2 = one point per 128-byte line
// (G1); 3 = limbs with both signs, 256 bytes per point (G1).  A forced layout (tests, diagnostics) that does not fit fails.
// The table is a trade of memory for work, never a reason to run out of memory later: whatever is allocated here -- the packed
// table AND the re-laid copy that briefly lives beside it -- may take at most a third of what is free when the call starts (a
// 2^24-point G2 query would ask for 40 GB).  When the table is skipped the MSM falls back to per-window bucket sets; the reason
// is kept in zk_bases::pre_note (zk_bases_precompute_note) and in zk_last_error, although the call succeeds.
template <class F>
int precompute_t(zk_ctx* ctx, zk_bases* b, uint32_t c, uint32_t W, int layout) {
    const size_t n = b->n, PW = 2 * F::WORDS;
    // levels per launch pair (k_dbl_levels); a forced packed layout keeps the level-by-level build: tests compare the two
    const uint32_t G = layout == 1 || W < 2 ? 1u : (n <= 4096 ? W - 1 : (n <= 16384 ? 4u : (n <= 65536 ? 2u : 1u)));
    const size_t lv = G > 1 ? W - 1 : 1;                       // levels the scratch holds at once (G > 1: every level above the table itself)
    uint32_t *xy, *scr;
    ZK_TRY(zk_scratch(ctx, "fb_xyzz", lv * n * 4 * F::WORDS * 4, (void**)&xy));
    ZK_TRY(zk_scratch(ctx, "fb_scr", lv * n * F::WORDS * 4, (void**)&scr));
    size_t mem_free = 0, mem_total = 0;
    const size_t packed_bytes = (size_t)W * n * PW * 4;
    auto skip = [&](const char* why) {
        char msg[256];
        snprintf(msg, sizeof msg, "window multiples skipped for a %zu-point G%d table (c = %u, W = %u: %.2f GB packed): %s; %.2f GB free",
                 n, b->group, c, W, packed_bytes / 1e9, why, mem_free / 1e9);
        b->pre_note = msg;
        ctx->last_error = msg;
        return layout ? ZK_ERR_NOMEM : ZK_OK;
    };
    // (a small table just allocates: hipMemGetInfo is ~10 ms with a few GB resident -- most of what the window multiples of a
    // 2^10-point table used to cost)
    const bool ask = packed_bytes > ((size_t)64 << 20);
    if (ask && hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) return skip("hipMemGetInfo failed");
    const size_t budget = ask ? mem_free / 3 : (size_t)1 << 30;
    if (packed_bytes > budget) return skip("more than a third of the free device memory");
    if (hipMalloc((void**)&b->pre, packed_bytes) != hipSuccess) {
        b->pre = nullptr;
        (void)hipGetLastError();
        return skip("hipMalloc failed");
    }
    auto fail_free = [&](void* extra) {      // an error below must not leave a half-built table behind
        if (extra) (void)hipFree(extra);
        (void)hipFree(b->pre);
        b->pre = nullptr;
        b->c_pre = b->W_pre = b->pre_stride = 0;
    };
    hipError_t e = hipMemcpyAsync(b->pre, b->dev, n * PW * 4, hipMemcpyDeviceToDevice, ctx->stream);
    const size_t chunks = (n + NORM_CHUNK - 1) / NORM_CHUNK;
    for (uint32_t w0 = 0; G > 1 && w0 + 1 < W && e == hipSuccess; w0 += G) {     // levels w0 + 1 .. w0 + g from level w0, un-normalised: xy[(w - 1) n + i]
        const uint32_t g = std::min(G, W - 1 - w0);
        const size_t total = (size_t)g * n;
        uint32_t* out = xy + (size_t)w0 * n * 4 * F::WORDS;
        if (w0 == 0) hipLaunchKernelGGL((k_dbl_levels<F, false>), zk_grid(total, 256), 256, 0, ctx->stream, (const uint32_t*)b->dev, out, n, c, g);
        else hipLaunchKernelGGL((k_dbl_levels<F, true>), zk_grid(total, 256), 256, 0, ctx->stream, (const uint32_t*)(xy + (size_t)(w0 - 1) * n * 4 * F::WORDS), out, n, c, g);
        e = hipGetLastError();
    }
    if (G > 1 && W > 1 && e == hipSuccess) {                  // ... and one normalisation for all of them
        const size_t total = (size_t)(W - 1) * n, tchunks = (total + NORM_CHUNK - 1) / NORM_CHUNK;
        hipLaunchKernelGGL(k_batch_affine<F>, (unsigned)((tchunks + 63) / 64), 64, 0, ctx->stream, xy, b->pre + n * PW, scr, total);
        e = hipGetLastError();
    }
    for (uint32_t w = 1; w < W && e == hipSuccess && G == 1; w++) {
        hipLaunchKernelGGL(k_dbl_c<F>, zk_grid(n, 256), 256, 0, ctx->stream, b->pre + (size_t)(w - 1) * n * PW, xy, n, c);
        hipLaunchKernelGGL(k_batch_affine<F>, (unsigned)((chunks + 63) / 64), 64, 0, ctx->stream, xy, b->pre + (size_t)w * n * PW, scr, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { fail_free(nullptr); ZK_HIP(ctx, e); }
    b->c_pre = c;
    b->W_pre = W;
    b->pre_stride = 0;
    b->pre_note = "packed";
    // G1: a 96-byte point in a packed table straddles two 128-byte lines half of the time and the accumulate kernel's gather
    // then fetches 2.6x the bytes it uses (profiles/r2_pmc_traffic.json: 3.45 GB per launch for 1.31 GB of points).  The window
    // multiples -- read once per digit, at random -- are re-laid one point per line (a third more memory, half the traffic), or,
    // when that fits too, as limbs with the negative of every point in the second line of a 256-byte slot (k_repack_limbs:
    // nothing to unpack or negate in the accumulate kernel).  G2's 192-byte points take two lines either way.
    if constexpr (F::WORDS == 12) {
        for (int form : {3, 2}) {
            if (layout && layout != form) continue;
            if (!layout && b->pre_stride) break;
            const size_t words = form == 3 ? 64 : 32, bytes = (size_t)W * n * words * 4;
            if (packed_bytes + bytes > budget) {         // (the packed table is still alive while the copy is made)
                if (layout) { fail_free(nullptr); return skip("the forced layout does not fit a third of the free device memory"); }
                continue;
            }
            uint32_t* wide = nullptr;
            if (hipMalloc((void**)&wide, bytes) != hipSuccess) {
                (void)hipGetLastError();
                if (layout) { fail_free(nullptr); return skip("hipMalloc of the forced layout failed"); }
                continue;
            }
            if (form == 3)
                hipLaunchKernelGGL(k_repack_limbs<F>, zk_grid((size_t)W * n, 256), 256, 0, ctx->stream, (const uint32_t*)b->pre, wide, (size_t)W * n);
            else
                hipLaunchKernelGGL(k_repack<F>, zk_grid((size_t)W * n, 256), 256, 0, ctx->stream, (const uint32_t*)b->pre, wide, (size_t)W * n, 32u);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) { fail_free(wide); ZK_HIP(ctx, e); }
            (void)hipFree(b->pre);
            b->pre = wide;
            b->pre_stride = (uint32_t)words;
            b->pre_note = form == 3 ? "limbs, both signs (256 B per point)" : "one point per 128-byte line";
        }
    } else if (layout > 1) {
        fail_free(nullptr);
        ZK_FAIL(ctx, ZK_ERR_ARG, "zk_bases_precompute_as: layouts 2 and 3 are G1 forms");
    }
    return ZK_OK;
}

}  // namespace

extern "C" int zk_fixed_base_g1_dev(zk_ctx* ctx, const zk_fr* gen_k, const void* scalars, size_t n, zk_bases** out) {
    ZK_API_BEGIN(ctx)
    return fixed_base_run<G1Field>(ctx, g1_generator(), gen_k, scalars, n, 1, out);
    ZK_API_END
}
extern "C" int zk_fixed_base_g2_dev(zk_ctx* ctx, const zk_fr* gen_k, const void* scalars, size_t n, zk_bases** out) {
    ZK_API_BEGIN(ctx)
    return fixed_base_run<G2Field>(ctx, g2_generator(), gen_k, scalars, n, 2, out);
    ZK_API_END
}

// Window multiples 2^(c w) * base_i, w < W = ceil(255 / c), for tables that stay resident (proving-key queries, an SRS).
// With them all W digits of a scalar drop into ONE bucket set, so the window can be as wide as c ~ log2(n) -- W falls
// from 16 to 13 at n = 2^20 (19 % fewer mixed additions) while the bucket count stays ~ n/2 -- and only one set of
// buckets is reduced.  288 GB of HBM is what pays for it: a 2^20-point G1 query grows from 96 MiB to 1.2 GiB.
// Measured at n = 2^20 (whole Groth16 proof): 35.9 ms without, 32.1 ms with c = 20.  c = 16 tables (the first attempt)
// did not pay: same number of additions, 16x the gather footprint.
// The top window must keep >= 10 significant bits: 255 - c (W - 1) of only 3 bits (c = 18, 21) sends every point into
// <= 8 buckets and the sort's atomics serialise (65 ms instead of 32).
static uint32_t precompute_window_bits(size_t n) {
    uint32_t lg = 0;                                   // round(log2 n): a 2^20 - 1 point query is a 2^20 one
    while (((size_t)3 << lg) <= 2 * n) lg++;
    // >= 2^16 points, measured on whole proofs: 2^16 .. 2^18 -> 17, 2^19 .. 2^22 -> 20 (22 loses: 4x the buckets).
    // Smaller tables (round 5: the reference's own circuits are 2^2 .. 2^15 constraints) are latency, not throughput: a window
    // of about lg + 2 bits leaves ~W / 4 = 4-5 terms per bucket (the accumulate kernel lasts as long as its fullest bucket) and ONE
    // bucket set means the host's Horner chain over ~30 windows -- 0.7 ms of a 2 ms proof at 2^10 -- shrinks to ~2 log2(NB) additions.
    // (second half of round 5, with the reduce of small bucket sets on lane groups: lg + 1 from 2^13 up -- it only moves 2^14, 16 -> 15 bits:
    // proof 1.11 -> 0.94 ms; the other sizes land on the same width through the top-window rule below)
    // (end of round 5, whole proofs on one box, admissible widths 15 / 16 / 17 / 20: 2^16 1.83 (15) -> 1.67 (17); 2^17 3.32 (16) ->
    // 2.88 (17); 2^18 5.30 (17) = 5.23 (20); 2^19 9.46 (17) -> 8.4 (20))
    // (... and 2^15: 1.46 (16) -> 1.21 (15))
    int c0 = lg >= 19 ? 20 : lg >= 16 ? 17 : (lg >= 13 ? std::min((int)lg + 1, 15) : (int)lg + 2);
    // (the 2^16 class starts at 49 152 points: the tables of a Marlin proof at |H| = 2^14 -- 49 15x points -- take 15 bits, 6.1-6.2 ms
    // against 6.2-6.4 with 17 and 6.4-6.6 with 16; Groth16 at 2^16 -- 65 536 points -- takes 17)
    if (lg == 16 && n < 65000) c0 = 15;
    const int lo = lg >= 16 ? 13 : 9, hi = lg >= 16 ? 20 : 16;
    if (c0 < lo) c0 = lo;
    if (c0 > hi) c0 = hi;
    // the top window must keep enough significant bits: 255 - c (W - 1) of only 3 bits (c = 18, 21) sends every point into <= 8
    // buckets (2^20 points: the sort's atomics serialise, 65 ms instead of 32); for a small table 2^(top - 1) buckets >= n / 16 do
    const int need = std::min(10, std::max(3, (int)lg - 3));
    const int tries[9] = {0, -1, 1, -2, 2, -3, 3, 4, 5};
    for (int t : tries) {
        int c = c0 + t;
        if (c < lo - 1 || c > 20) continue;
        int W = (255 + c - 1) / c;
        if (255 - c * (W - 1) >= need) return (uint32_t)c;
    }
    return 16;
}

static int precompute_enabled_by_default() {
    const char* e = getenv("ZK_PRECOMP");
    return !e || atoi(e) != 0;
}
// Resident proving-key queries get their window multiples at upload / setup time (ZK_PRECOMP=0 turns it off).
int zk_bases_precompute_auto(zk_ctx* ctx, zk_bases* b) {
    return (precompute_enabled_by_default() && b && b->n >= ZK_PRECOMP_MIN_POINTS) ? zk_bases_precompute(ctx, b) : ZK_OK;
}

extern "C" uint32_t zk_bases_window_bits(const zk_bases* b) { return b ? b->c_pre : 0; }
// how many copies (windows) the multiples of an n-point table have: what bases_cache.hip budgets for
uint32_t zk_precompute_windows(size_t n) { const uint32_t c = precompute_window_bits(n); return (255 + c - 1) / c; }

static int precompute_run(zk_ctx* ctx, zk_bases* b, int layout) {
    if (!ctx || layout < 0 || layout > 3) return ZK_ERR_ARG;
    if (!b || b->pre || b->n < ZK_PRECOMP_MIN_POINTS) return ZK_OK;
    const uint32_t c = precompute_window_bits(b->n);
    const uint32_t W = (255 + c - 1) / c;
    if (b->group == 1) return precompute_t<G1Field>(ctx, b, c, W, layout);
    return precompute_t<G2Field>(ctx, b, c, W, layout);
}

extern "C" int zk_bases_precompute(zk_ctx* ctx, zk_bases* b) {
    ZK_API_BEGIN(ctx)
    return precompute_run(ctx, b, 0);
    ZK_API_END
}

extern "C" int zk_bases_precompute_as(zk_ctx* ctx, zk_bases* b, int layout) {
    ZK_API_BEGIN(ctx)
    return precompute_run(ctx, b, layout);
    ZK_API_END
}

// ---- the same table built BESIDE the caller's work (bases_cache.hip): every allocation up front, the launches in SLICES of a
// few hundred microseconds on a side stream, handed out one at a time by the cache's builder thread while no library call is in
// flight on the device, published by a later call once the last slice is through.  The table serves MSMs in its plain form until
// then.  (Building the multiples of a proving key's five queries inside the caller's second proof was a 356 - 410 ms call on a
// 40 ms path -- VERDICT r5 weak 4 iii; enqueued all at once on a side stream they still made the next four proofs 2 - 3x slower,
// the accumulate kernels sharing the chip with the build: measured in round 6, profiles/r6_trait_first.jsonl.)
namespace {
// points per slice of a level (a multiple of NORM_CHUNK): 2^18 lanes of twenty dependent doublings fill the chip (2^16 left it at one
// wave per SIMD: the build crawled -- one table in seven proofs' gaps); ~1 ms of kernels, the most a call can find in its way
constexpr size_t PRE_SLICE = (size_t)1 << 18;
constexpr size_t PRE_REPACK_SLICE = (size_t)1 << 21;

template <class F>
int precompute_begin_t(zk_ctx* ctx, zk_bases* b, uint32_t c, uint32_t W, size_t budget, ZkPrecompJob** out) {
    const size_t n = b->n, PW = 2 * F::WORDS;
    const size_t packed_bytes = (size_t)W * n * PW * 4, xy_bytes = n * 4 * F::WORDS * 4, scr_bytes = n * F::WORDS * 4;
    if (packed_bytes + xy_bytes + scr_bytes > budget) {
        char msg[256];
        snprintf(msg, sizeof msg, "window multiples skipped for a %zu-point G%d table (c = %u, W = %u: %.2f GB packed): more than the cache's budget", n, b->group, c, W, packed_bytes / 1e9);
        b->pre_note = msg;
        return ZK_OK;
    }
    // (no HIP call here: this runs inside the MSM call that earned the table its multiples -- hipMemGetInfo alone is ~10 ms with
    // a few GB of tables resident; the memory check and the allocations are the builder's first step)
    std::unique_ptr<ZkPrecompJob> j(new ZkPrecompJob());
    j->b = b; j->c = c; j->W = W;
    j->budget = budget;
    (void)ctx;
    *out = j.release();
    return ZK_OK;
}

// the next slice of the build on `st`; *more = false once the last one has been enqueued (the caller waits for `st` between slices)
template <class F>
hipError_t precompute_step_t(ZkPrecompJob* j, hipStream_t st, bool* more) {
    const zk_bases* b = j->b;
    const size_t n = b->n, PW = 2 * F::WORDS;
    *more = true;
    if (j->phase == -1) {
        // the allocations (several GB: ~10 ms of page-table work) -- here, in the builder's time, not in the call that earned the
        // table its multiples; a failure leaves the table plain
        const size_t packed_bytes = (size_t)j->W * n * PW * 4, xy_bytes = n * 4 * F::WORDS * 4, scr_bytes = n * F::WORDS * 4;
        if (packed_bytes + xy_bytes + scr_bytes > ((size_t)256 << 20)) {     // (a small table just allocates: hipMemGetInfo is ~10 ms)
            size_t mem_free = 0, mem_total = 0;
            if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) { (void)hipGetLastError(); return hipErrorOutOfMemory; }
            j->budget = std::min(j->budget, mem_free / 3);     // never more than a third of what is free (see precompute_t)
        }
        if (packed_bytes + xy_bytes + scr_bytes > j->budget) return hipErrorOutOfMemory;
        if (hipMalloc((void**)&j->packed, packed_bytes) != hipSuccess || hipMalloc((void**)&j->xy, xy_bytes) != hipSuccess ||
            hipMalloc((void**)&j->scr, scr_bytes) != hipSuccess) {
            (void)hipGetLastError();
            return hipErrorOutOfMemory;
        }
        if constexpr (F::WORDS == 12) {                    // the re-laid copy (see precompute_t): limbs with both signs, else a line per point
            for (uint32_t words : {64u, 32u}) {
                const size_t bytes = (size_t)j->W * n * words * 4;
                if (packed_bytes + xy_bytes + scr_bytes + bytes > j->budget) continue;
                if (hipMalloc((void**)&j->wide, bytes) != hipSuccess) { j->wide = nullptr; (void)hipGetLastError(); continue; }
                j->wide_words = words;
                break;
            }
        }
        j->phase = 0;
        return hipSuccess;
    }
    if (j->phase == 0) {                                   // level 0 = the table itself
        j->phase = j->W > 1 ? 1 : 2;
        j->w = 1; j->pos = 0;
        return hipMemcpyAsync(j->packed, b->dev, n * PW * 4, hipMemcpyDeviceToDevice, st);
    }
    if (j->phase == 1) {                                   // level w from level w - 1: 2^c times every point, normalised; points [pos, pos + len)
        const size_t lo = j->pos, len = std::min(PRE_SLICE, n - lo), chunks = (len + NORM_CHUNK - 1) / NORM_CHUNK;
        hipLaunchKernelGGL(k_dbl_c<F>, zk_grid(len, 256), 256, 0, st, j->packed + ((size_t)(j->w - 1) * n + lo) * PW, j->xy + lo * 4 * F::WORDS, len, j->c);
        hipLaunchKernelGGL(k_batch_affine<F>, (unsigned)((chunks + 63) / 64), 64, 0, st, j->xy + lo * 4 * F::WORDS, j->packed + ((size_t)j->w * n + lo) * PW,
                           j->scr + lo * F::WORDS, len);
        j->pos += len;
        if (j->pos >= n) { j->pos = 0; if (++j->w >= j->W) j->phase = 2; }
        return hipGetLastError();
    }
    if (j->phase == 2) {
        if constexpr (F::WORDS == 12) {
            if (j->wide) {
                const size_t total = (size_t)j->W * n, lo = j->pos, len = std::min(PRE_REPACK_SLICE, total - lo);
                if (j->wide_words == 64) hipLaunchKernelGGL(k_repack_limbs<F>, zk_grid(len, 256), 256, 0, st, (const uint32_t*)j->packed + lo * PW, j->wide + lo * 64, len);
                else hipLaunchKernelGGL(k_repack<F>, zk_grid(len, 256), 256, 0, st, (const uint32_t*)j->packed + lo * PW, j->wide + lo * 32, len, 32u);
                j->pos += len;
                if (j->pos >= total) j->phase = 3;
                return hipGetLastError();
            }
        }
        j->phase = 3;
        return hipSuccess;
    }
    if (j->phase == 3) {                                   // (the stream has been waited for) the scratch goes, in the builder's time as well
        (void)hipFree(j->xy); j->xy = nullptr;
        (void)hipFree(j->scr); j->scr = nullptr;
        if (j->wide) { (void)hipFree(j->packed); j->packed = nullptr; }
        j->phase = 4;
    }
    *more = false;
    return hipSuccess;
}
}  // namespace

int zk_bases_precompute_begin(zk_ctx* ctx, zk_bases* b, size_t budget, ZkPrecompJob** out) {
    *out = nullptr;
    if (!b || b->pre || b->n < ZK_PRECOMP_MIN_POINTS) return ZK_OK;
    const uint32_t c = precompute_window_bits(b->n);
    const uint32_t W = (255 + c - 1) / c;
    if (b->group == 1) return precompute_begin_t<G1Field>(ctx, b, c, W, budget, out);
    return precompute_begin_t<G2Field>(ctx, b, c, W, budget, out);
}
hipError_t zk_bases_precompute_step(ZkPrecompJob* j, hipStream_t st, bool* more) {
    if (j->b->group == 1) return precompute_step_t<G1Field>(j, st, more);
    return precompute_step_t<G2Field>(j, st, more);
}
// the stream the slices ran on has been waited for.  keep = false (or a slice failed): throw the table away
int zk_bases_precompute_finish(zk_ctx* ctx, ZkPrecompJob* j, bool keep) {
    zk_bases* b = j->b;
    if (j->xy) (void)hipFree(j->xy);
    if (j->scr) (void)hipFree(j->scr);
    const hipError_t e = j->err;
    if (e == hipSuccess && keep && j->phase == 4) {
        if (j->wide) {
            b->pre = j->wide;
            b->pre_stride = j->wide_words;
            b->pre_note = j->wide_words == 64 ? "limbs, both signs (256 B per point)" : "one point per 128-byte line";
        } else {
            b->pre = j->packed;
            b->pre_stride = 0;
            b->pre_note = "packed";
        }
        b->c_pre = j->c;
        b->W_pre = j->W;
    } else {
        if (j->packed) (void)hipFree(j->packed);
        if (j->wide) (void)hipFree(j->wide);
        if (e == hipErrorOutOfMemory) {                     // not an error of the call that happens to collect the job: the table stays plain
            b->pre_note = "window multiples skipped: more than a third of the free device memory, or hipMalloc failed";
            delete j;
            return ZK_OK;
        }
    }
    delete j;
    ZK_HIP(ctx, e);
    return ZK_OK;
}

// "" when nothing was attempted; the layout that was built; or why the table was skipped
extern "C" const char* zk_bases_precompute_note(const zk_bases* b) { return b ? b->pre_note.c_str() : ""; }
