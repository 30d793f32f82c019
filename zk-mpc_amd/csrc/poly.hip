// poly.hip -- dense-polynomial kernels over Fr and KZG10 commit / open on top of the MSM: the data-parallel
// pieces of the Marlin / poly-commit path.
//
// Replaces (reference):
//   ark_ff::batch_inversion                                arkworks/algebra/ff/src/fields/mod.rs:597-659
//   DensePolynomial::evaluate (Horner)                     arkworks/algebra/poly/src/polynomial/univariate/dense.rs:53-75
//   DensePolynomial mul via FFT                            dense.rs:568-584
//   DensePolynomial::divide_by_vanishing_poly              dense.rs:166-173 (-> divide_with_q_and_r, univariate/mod.rs:133-176)
//   p / (X - z)  (KZG10::compute_witness_polynomial)       arkworks/poly-commit/src/kzg10/mod.rs:212-235
//   KZG10::commit / open_with_witness_polynomial           kzg10/mod.rs:142-205, 237-293
//
// Polynomials are coefficient vectors of Fr in the reference's form (device memory).  Every product here has one
// operand that the host supplies (the point z, its powers), converted once to the device's internal Montgomery
// form, so data * constant needs a single Montgomery product (fp29.cuh header).
//
// Horner and synthetic division are first-order linear recurrences Q_j = c_j + z * Q_{j+1}.  They are solved as a
// three-level scan: every thread folds a chunk of 16 coefficients (the chunk's value for carry-in 0), a block combines its
// 256 chunks in LDS with the powers z^(16 2^l), one block combines the spans with the powers of z^4096; the division then
// replays every chunk from its now-known carry-in.  2n (evaluation: n) products, coalesced-by-chunk traffic, two or three
// launches of fixed depth (~60 dependent multiply-adds) whatever n.
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "hostgroup.hpp"
#include "internal.hpp"
#include <vector>

using namespace zk;

namespace {

constexpr int INV_CH = 32;   // batch-inversion chunk per thread

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
Fr host_int(const zk_fr* x) { return fp_ext_to_int<FrParams>(host_load_ext<FrParams>(x->l)); }
Fr host_pow(const Fr& a, uint64_t e) {
    Fr r = fp_one<FrParams>();
    bool started = false;
    for (int b = 63; b >= 0; b--) {
        if (started) r = fp_sqr<FrParams>(r);
        if ((e >> b) & 1) { r = started ? fp_mul<FrParams>(r, a) : a; started = true; }
    }
    return r;
}

// Division by X^N - 1: with the coefficients viewed as an m x N row-major matrix (row k = c[kN .. kN+N-1]) the quotient
// is the column-wise exclusive suffix sum, q[k][j] = sum_{k' > k} c[k'][j], and the remainder is r[j] = c[0][j] + q[0][j].
// Rows are cut into G groups of R; one thread owns one (group, column), so the work is O(n) for every N (a tiny N, like
// the |X| = 2 of Marlin's w(X) / v_X, included).
__global__ void __launch_bounds__(256) k_divv_sums(const void* c, size_t n, size_t N, size_t R, size_t G, void* sums) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < G * N; t += (size_t)gridDim.x * blockDim.x) {
        size_t g = t / N, j = t - g * N;
        Fr acc = fp_zero<FrParams>();
        for (size_t r = 0; r < R; r++) {
            size_t i = (g * R + r) * N + j;
            if (i < n) acc = fr_add(acc, fr_load(c, i));
        }
        fr_store(sums, t, acc);
    }
}

__global__ void __launch_bounds__(256) k_divv_fill(const void* c, size_t n, size_t N, size_t R, size_t G, const void* sums, void* q,
                                                   size_t nq, void* rem) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < G * N; t += (size_t)gridDim.x * blockDim.x) {
        size_t g = t / N, j = t - g * N;
        Fr acc = fp_zero<FrParams>();
        for (size_t g2 = g + 1; g2 < G; g2++) acc = fr_add(acc, fr_load(sums, g2 * N + j));
        for (size_t r = R; r-- > 0;) {
            size_t row = g * R + r, i = row * N + j;
            if (i < nq) fr_store(q, i, acc);
            Fr x = i < n ? fr_load(c, i) : fp_zero<FrParams>();
            if (row == 0) fr_store(rem, j, fr_add(acc, x));
            acc = fr_add(acc, x);
        }
    }
}

// Montgomery's trick on chunks of INV_CH; zeros stay zero (ff/src/fields/mod.rs:632-658).
__global__ void __launch_bounds__(64) k_batch_inverse(void* v, size_t n, uint32_t* scratch) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t * INV_CH < n; t += (size_t)gridDim.x * blockDim.x) {
        size_t lo = t * INV_CH, hi = lo + INV_CH < n ? lo + INV_CH : n;
        Fr run = fp_one<FrParams>();
        for (size_t i = lo; i < hi; i++) {
            fr_store(scratch, i, run);                                       // product of the non-zero elements before i (internal form)
            Fr x = fp_ext_to_int<FrParams>(fr_load(v, i));
            if (!fp_is_zero<FrParams>(x)) run = fr_mul(run, x);
        }
        Fr inv = fp_inv<FrParams>(run);
        for (size_t i = hi; i-- > lo;) {
            Fr x = fp_ext_to_int<FrParams>(fr_load(v, i));
            if (fp_is_zero<FrParams>(x)) continue;
            Fr xi = fr_mul(inv, fr_load(scratch, i));                        // 1/x, internal form
            inv = fr_mul(inv, x);
            fr_store(v, i, fp_int_to_ext<FrParams>(xi));
        }
    }
}

// The same in two halves around a HOST inversion, for short vectors: a field inversion is ~380 dependent products, 0.25 ms on one
// lane whatever the vector's length (the whole of k_batch_inverse at |K| = 2^10) and ~20 us on a host core.  k_inv_fwd leaves the
// prefix products and every chunk's total; the host inverts the totals (Montgomery's trick over them, one inversion); k_inv_bwd
// walks the chunks back.  totals[t] (internal form, 9 limbs) = product of the chunk's non-zero elements.
// (ch: elements per lane -- the lanes' chains are what these two kernels last: 8 for vectors of up to 2^14, 32 at 2^16)
__global__ void __launch_bounds__(64) k_inv_fwd(const void* v, size_t n, size_t ch, uint32_t* scratch, uint32_t* totals) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t * ch < n; t += (size_t)gridDim.x * blockDim.x) {
        size_t lo = t * ch, hi = lo + ch < n ? lo + ch : n;
        Fr run = fp_one<FrParams>();
        for (size_t i = lo; i < hi; i++) {
            fr_store(scratch, i, run);
            Fr x = fp_ext_to_int<FrParams>(fr_load(v, i));
            if (!fp_is_zero<FrParams>(x)) run = fr_mul(run, x);
        }
#pragma unroll
        for (int k = 0; k < 9; k++) totals[t * 9 + k] = run.l[k];
    }
}
__global__ void __launch_bounds__(64) k_inv_bwd(void* v, size_t n, size_t ch, const uint32_t* scratch, const uint32_t* inv_totals) {
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t * ch < n; t += (size_t)gridDim.x * blockDim.x) {
        size_t lo = t * ch, hi = lo + ch < n ? lo + ch : n;
        Fr inv;
#pragma unroll
        for (int k = 0; k < 9; k++) inv.l[k] = inv_totals[t * 9 + k];
        for (size_t i = hi; i-- > lo;) {
            Fr x = fp_ext_to_int<FrParams>(fr_load(v, i));
            if (fp_is_zero<FrParams>(x)) continue;
            Fr xi = fr_mul(inv, fr_load(scratch, i));
            inv = fr_mul(inv, x);
            fr_store(v, i, fp_int_to_ext<FrParams>(xi));
        }
    }
}

// out[i] = start * base^i
// (PC powers per lane: 32, or 8 for vectors of up to 2^16 -- the lane's chain is what a short launch lasts)
__global__ void __launch_bounds__(256) k_powers(void* out, FrK base_k, FrK start_k, size_t n, int PC) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t i0 = t * (size_t)PC;
    if (i0 >= n) return;
    const Fr base = frk(base_k);
    Fr p = fp_one<FrParams>();
    bool started = false;
    for (int b = 63; b >= 0; b--) {
        if (started) p = fp_sqr<FrParams>(p);
        if ((i0 >> b) & 1) { p = started ? fr_mul(p, base) : base; started = true; }
    }
    p = fr_mul(fp_int_to_ext<FrParams>(p), frk(start_k));                    // ext form of start * base^i0
    for (int j = 0; j < PC && i0 + j < n; j++) {
        fr_store(out, i0 + j, p);
        p = fr_mul(p, base);
    }
}

// ---- batched evaluation: any number of (polynomial, point) pairs in two launches and one copy back --------------------------
// Stage 1: a block owns a span of 256 x EV_CH coefficients of one polynomial; a thread folds EV_CH of them by Horner, the
// block combines its 256 chunk values by a tree (level l multiplies the right-hand neighbour by z^(EV_CH 2^l)) into the value
// of the span's own polynomial at z.  Stage 2: one block per polynomial folds R consecutive span values per thread with the
// multiplier z^SPAN and combines the 256 results the same way.  Depth: EV_CH + 8 + R + 8 dependent multiply-adds (~60 us)
// whatever the number of pairs; the recursive scan above needs three to six launches and a host round trip per polynomial.
constexpr int EV_CH = 16;
constexpr int EV_SPAN = 256 * EV_CH;
struct EvalJob {
    const void* c;
    size_t n;
    uint32_t first_block, nblocks, R, pad;
    uint32_t z[9], lvl1[8][9], zspan[9], lvl2[8][9];       // internal form
};
__device__ __forceinline__ Fr limbs9(const uint32_t* l) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = l[i];
    return r;
}
// h[0] <- sum_t h[t] * base^t over the 256 values of the block, base^(2^l) given as lvl[l]
__device__ __forceinline__ Fr block_power_tree(uint32_t (*lds)[256], const uint32_t (*lvl)[9], Fr v) {
    const uint32_t tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 9; k++) lds[k][tid] = v.l[k];
    __syncthreads();
    for (int l = 0; l < 8; l++) {
        const uint32_t d = 1u << l;
        if ((tid & (2 * d - 1)) == 0) {
            Fr r;
#pragma unroll
            for (int k = 0; k < 9; k++) r.l[k] = lds[k][tid + d];
            v = fr_add(v, fr_mul(r, limbs9(lvl[l])));
#pragma unroll
            for (int k = 0; k < 9; k++) lds[k][tid] = v.l[k];
        }
        __syncthreads();
    }
    return v;
}
__device__ __forceinline__ void eval_span(const EvalJob& J, void* partial) {
    __shared__ uint32_t lds[9][256];
    const Fr z = limbs9(J.z);
    const size_t lo = ((size_t)(blockIdx.x - J.first_block) * 256 + threadIdx.x) * EV_CH;
    const size_t hi = lo + EV_CH < J.n ? lo + EV_CH : J.n;
    Fr acc = fp_zero<FrParams>();
    for (size_t i = hi; i > lo; i--) acc = fr_add(fr_mul(acc, z), fr_load(J.c, i - 1));
    acc = block_power_tree(lds, J.lvl1, acc);
    if (threadIdx.x == 0) fr_store(partial, blockIdx.x, acc);
}
__global__ void __launch_bounds__(256) k_eval_spans(const EvalJob* jobs, uint32_t njobs, void* partial) {
    uint32_t j = 0;
    while (j + 1 < njobs && jobs[j + 1].first_block <= blockIdx.x) j++;
    eval_span(jobs[j], partial);
}
__global__ void __launch_bounds__(256) k_eval_spans1(const EvalJob J, void* partial) { eval_span(J, partial); }    // one job, by value
__global__ void __launch_bounds__(256) k_eval_join(const EvalJob* jobs, const void* partial, void* out) {
    __shared__ uint32_t lds[9][256];
    const EvalJob& J = jobs[blockIdx.x];
    const Fr zs = limbs9(J.zspan);
    const size_t lo = (size_t)threadIdx.x * J.R, hi = lo + J.R < J.nblocks ? lo + J.R : J.nblocks;
    Fr acc = fp_zero<FrParams>();
    for (size_t i = hi; i > lo; i--) acc = fr_add(fr_mul(acc, zs), fr_load(partial, J.first_block + i - 1));
    acc = block_power_tree(lds, J.lvl2, acc);
    if (threadIdx.x == 0) fr_store(out, blockIdx.x, acc);
}

// ---- division by X - z: the same recurrence with every Q_j written out, three launches, no host round trip ----------------
// Q_j = c_j + z Q_{j+1} (Q_n = 0); the quotient is Q_1 .. Q_{n-1}, the remainder Q_0 = p(z).  k_eval_spans gives every
// span's own value; k_span_carries turns them into every span's carry-in Q_{(b+1) SPAN} (a suffix scan over the spans with the
// multiplier z^SPAN: R spans per thread, Hillis-Steele over the 256 threads, replay) and the remainder; k_fill_spans redoes the
// chunk values of a span, scans them with its carry-in and replays every chunk from its now-known carry-in, writing Q_j to
// out[j + shift].  Depth ~ 2 EV_CH + 2 R + 3 * 8 multiply-adds (the recursive form this replaces: 3 x 64 per level, 3 to 4
// levels, and a copy to the host at the bottom).
// inclusive suffix scan over the block: returns I_t = sum_{u >= t} v_u M^(u - t), M^(2^l) = lvl[l]; buf: 2 x 9 x 256 words
__device__ __forceinline__ Fr block_suffix_scan(uint32_t (*buf)[9][256], const uint32_t (*lvl)[9], Fr v) {
    const uint32_t tid = threadIdx.x;
    int cur = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) buf[0][k][tid] = v.l[k];
    __syncthreads();
    for (int l = 0; l < 8; l++) {
        const uint32_t d = 1u << l;
        if (tid + d < 256) {
            Fr r;
#pragma unroll
            for (int k = 0; k < 9; k++) r.l[k] = buf[cur][k][tid + d];
            v = fr_add(v, fr_mul(r, limbs9(lvl[l])));
        }
#pragma unroll
        for (int k = 0; k < 9; k++) buf[cur ^ 1][k][tid] = v.l[k];
        cur ^= 1;
        __syncthreads();
    }
    return v;       // buf[cur] holds every thread's I_t
}
__global__ void __launch_bounds__(256) k_span_carries(const EvalJob J, const void* partial, void* carry, void* rem) {
    __shared__ uint32_t buf[2][9][256];
    const Fr zs = limbs9(J.zspan);
    const uint32_t tid = threadIdx.x;
    const size_t lo = (size_t)tid * J.R, hi = lo + J.R < J.nblocks ? lo + J.R : J.nblocks;
    Fr acc = fp_zero<FrParams>();
    for (size_t i = hi; i > lo; i--) acc = fr_add(fr_mul(acc, zs), fr_load(partial, i - 1));
    block_suffix_scan(buf, J.lvl2, acc);            // 8 levels: the final values sit in buf[0]
    Fr c = fp_zero<FrParams>();
    if (tid + 1 < 256) {
#pragma unroll
        for (int k = 0; k < 9; k++) c.l[k] = buf[0][k][tid + 1];
    }
    for (size_t i = hi; i > lo; i--) {
        fr_store(carry, i - 1, c);
        c = fr_add(fr_mul(c, zs), fr_load(partial, i - 1));
    }
    if (tid == 0 && rem) fr_store(rem, 0, c);
}
// carry == nullptr: a polynomial of ONE span (<= 4 096 coefficients) -- its carry-in is zero and this launch is the whole division
// (the remainder Q_0 goes to rem when asked for): one launch instead of three for the opening witnesses of a small Marlin proof.
__global__ void __launch_bounds__(256) k_fill_spans(const EvalJob J, const void* carry, void* out, int shift, void* rem) {
    __shared__ uint32_t buf[2][9][256];
    const Fr z = limbs9(J.z);
    const uint32_t tid = threadIdx.x;
    const size_t lo = ((size_t)blockIdx.x * 256 + tid) * EV_CH;
    const size_t hi = lo + EV_CH < J.n ? lo + EV_CH : (lo < J.n ? J.n : lo);
    const Fr cin = carry ? fr_load(carry, blockIdx.x) : fp_zero<FrParams>();
    Fr h = fp_zero<FrParams>();
    for (size_t i = hi; i > lo; i--) h = fr_add(fr_mul(h, z), fr_load(J.c, i - 1));
    if (tid == 255) h = fr_add(h, fr_mul(cin, limbs9(J.lvl1[0])));      // the span's carry-in enters behind its last chunk
    block_suffix_scan(buf, J.lvl1, h);
    Fr acc = cin;
    if (tid + 1 < 256) {
#pragma unroll
        for (int k = 0; k < 9; k++) acc.l[k] = buf[0][k][tid + 1];
    }
    for (size_t i = hi; i > lo; i--) {
        acc = fr_add(fr_mul(acc, z), fr_load(J.c, i - 1));
        const long long o = (long long)(i - 1) + shift;
        if (o >= 0) fr_store(out, (size_t)o, acc);
    }
    if (rem && blockIdx.x == 0 && tid == 0) fr_store(rem, 0, acc);         // Q_0 (an empty chunk 0 cannot occur: n >= 1)
}

}  // namespace

extern "C" int zk_fr_powers_dev(zk_ctx* ctx, const zk_fr* base, const zk_fr* start, size_t n, void* out_dev) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !base || !start || (n && !out_dev)) return ZK_ERR_ARG;
    if (!n) return ZK_OK;
    Fr b = host_int(base);
    Fr s_ext = host_load_ext<FrParams>(start->l);   // kept in ext form: multiplied in after the int->ext conversion of base^i0
    const int pc = n <= ((size_t)1 << 16) ? 8 : 32;
    size_t threads = (n + pc - 1) / pc;
    hipLaunchKernelGGL(k_powers, (unsigned)((threads + 255) / 256), 256, 0, ctx->stream, out_dev, to_frk(b),
                       to_frk(fp_ext_to_int<FrParams>(s_ext)), n, pc);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_fr_batch_inverse_dev(zk_ctx* ctx, void* v_dev, size_t n) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (n && !v_dev)) return ZK_ERR_ARG;
    if (!n) return ZK_OK;
    uint32_t* scr;
    ZK_TRY(zk_scratch(ctx, "poly_inv", n * 32 + 32, (void**)&scr));
    size_t ch = 8;                                      // the shortest chunk that leaves <= 2048 totals for the host
    while (ch < (size_t)INV_CH && (n + ch - 1) / ch > 2048) ch *= 2;
    size_t chunks = (n + ch - 1) / ch;
    if (chunks <= 2048) {
        // short vector: the one inversion on the host (see k_inv_fwd)
        uint32_t* tot;
        ZK_TRY(zk_scratch(ctx, "poly_inv_tot", chunks * 36 * 2, (void**)&tot));
        auto& pin = ctx->pinned[-4];
        if (pin.bytes < 2048 * 36 * 2) {
            if (pin.p) (void)hipHostFree(pin.p);
            pin.p = nullptr; pin.bytes = 0;
            ZK_HIP(ctx, hipHostMalloc(&pin.p, 2048 * 36 * 2, hipHostMallocDefault));
            pin.bytes = 2048 * 36 * 2;
        }
        uint32_t* h = (uint32_t*)pin.p;
        hipLaunchKernelGGL(k_inv_fwd, zk_grid(chunks, 64, 8192), 64, 0, ctx->stream, (const void*)v_dev, n, ch, scr, tot);
        ZK_HIP(ctx, hipGetLastError());
        ZK_HIP(ctx, hipMemcpyAsync(h, tot, chunks * 36, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // Montgomery's trick over the chunk totals (never zero: a chunk of zeros has total one)
        std::vector<Fr> pre(chunks);
        Fr run = fp_one<FrParams>();
        auto tot_at = [&](size_t t) { Fr x; for (int k = 0; k < 9; k++) x.l[k] = h[t * 9 + k]; return x; };
        for (size_t t = 0; t < chunks; t++) { pre[t] = run; run = fp_mul<FrParams>(run, tot_at(t)); }
        Fr inv = fp_inv<FrParams>(run);
        uint32_t* hi = h + chunks * 9;
        for (size_t t = chunks; t-- > 0;) {
            const Fr it = fp_mul<FrParams>(inv, pre[t]);
            inv = fp_mul<FrParams>(inv, tot_at(t));
            for (int k = 0; k < 9; k++) hi[t * 9 + k] = it.l[k];
        }
        ZK_HIP(ctx, hipMemcpyAsync(tot + chunks * 9, hi, chunks * 36, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_inv_bwd, zk_grid(chunks, 64, 8192), 64, 0, ctx->stream, v_dev, n, ch, (const uint32_t*)scr, (const uint32_t*)(tot + chunks * 9));
        ZK_HIP(ctx, hipGetLastError());
        return ZK_OK;
    }
    chunks = (n + INV_CH - 1) / INV_CH;
    hipLaunchKernelGGL(k_batch_inverse, zk_grid(chunks, 64, 8192), 64, 0, ctx->stream, v_dev, n, scr);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

namespace {
void eval_job_fill(EvalJob& J, const void* c, size_t n, uint32_t first_block, const Fr& z) {
    J.c = c;
    J.n = n;
    J.first_block = first_block;
    J.nblocks = (uint32_t)((n + EV_SPAN - 1) / EV_SPAN);
    if (!J.nblocks) J.nblocks = 1;
    J.R = (J.nblocks + 255) / 256;
    J.pad = 0;
    auto put = [](uint32_t* d, const Fr& v) { for (int k = 0; k < 9; k++) d[k] = v.l[k]; };
    put(J.z, z);
    Fr p = host_pow(z, EV_CH);
    for (int l = 0; l < 8; l++) { put(J.lvl1[l], p); p = fp_sqr<FrParams>(p); }     // p ends as z^(EV_CH * 256) = z^SPAN
    put(J.zspan, p);
    p = host_pow(p, J.R);
    for (int l = 0; l < 8; l++) { put(J.lvl2[l], p); p = fp_sqr<FrParams>(p); }
}
}  // namespace

// DensePolynomial::evaluate for `count` (polynomial, point) pairs at once: out[i] = polys[i](points[i]).
extern "C" int zk_poly_evaluate_batch_dev(zk_ctx* ctx, const zk_poly_ref* polys, const zk_fr* points, size_t count, zk_fr* out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (count && (!polys || !points || !out))) return ZK_ERR_ARG;
    if (!count) return ZK_OK;
    std::vector<EvalJob> jobs(count);
    uint32_t blocks = 0;
    for (size_t i = 0; i < count; i++) {
        if (polys[i].n && !polys[i].ptr) return ZK_ERR_ARG;
        eval_job_fill(jobs[i], polys[i].ptr, polys[i].n, blocks, host_int(&points[i]));
        blocks += jobs[i].nblocks;
    }
    char* scr;
    const size_t jobs_bytes = (count * sizeof(EvalJob) + 255) & ~(size_t)255;
    ZK_TRY(zk_scratch(ctx, "poly_eval_batch", jobs_bytes + (size_t)blocks * 32 + count * 32, (void**)&scr));
    void* partial = scr + jobs_bytes;
    void* res = scr + jobs_bytes + (size_t)blocks * 32;
    ZK_HIP(ctx, hipMemcpyAsync(scr, jobs.data(), count * sizeof(EvalJob), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_eval_spans, blocks, 256, 0, ctx->stream, (const EvalJob*)scr, (uint32_t)count, partial);
    hipLaunchKernelGGL(k_eval_join, (unsigned)count, 256, 0, ctx->stream, (const EvalJob*)scr, (const void*)partial, res);
    ZK_HIP(ctx, hipGetLastError());
    std::vector<uint32_t> w(count * 8);
    ZK_HIP(ctx, hipMemcpyAsync(w.data(), res, count * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));      // also keeps `jobs` alive until the upload has been consumed
    for (size_t i = 0; i < count; i++)
        for (int k = 0; k < 4; k++) out[i].l[k] = (uint64_t)w[8 * i + 2 * k] | ((uint64_t)w[8 * i + 2 * k + 1] << 32);
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_poly_evaluate_dev(zk_ctx* ctx, const void* coeffs_dev, size_t n, const zk_fr* point, zk_fr* out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !point || !out || (n && !coeffs_dev)) return ZK_ERR_ARG;
    const zk_poly_ref p{coeffs_dev, n};
    return zk_poly_evaluate_batch_dev(ctx, &p, point, 1, out);
    ZK_API_END
}

extern "C" int zk_poly_divide_by_linear_dev(zk_ctx* ctx, const void* coeffs_dev, size_t n, const zk_fr* z, void* q_dev, zk_fr* rem) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !z || (n && !coeffs_dev) || (n > 1 && !q_dev)) return ZK_ERR_ARG;
    EvalJob job;                                    // 680 B: travels as a kernel argument, no copy, nothing to keep alive
    eval_job_fill(job, coeffs_dev, n, 0, host_int(z));
    const uint32_t blocks = job.nblocks;
    char* scr;
    ZK_TRY(zk_scratch(ctx, "poly_divlin", (size_t)blocks * 64 + 32, (void**)&scr));
    void* partial = scr;
    void* carry = scr + (size_t)blocks * 32;
    void* rem_dev = scr + (size_t)blocks * 64;
    if (blocks == 1 && n > 1) {
        hipLaunchKernelGGL(k_fill_spans, 1, 256, 0, ctx->stream, job, (const void*)nullptr, q_dev, -1, rem ? rem_dev : nullptr);
    } else {
        hipLaunchKernelGGL(k_eval_spans1, blocks, 256, 0, ctx->stream, job, partial);
        hipLaunchKernelGGL(k_span_carries, 1, 256, 0, ctx->stream, job, (const void*)partial, carry, rem ? rem_dev : nullptr);
        if (n > 1) hipLaunchKernelGGL(k_fill_spans, blocks, 256, 0, ctx->stream, job, (const void*)carry, q_dev, -1, (void*)nullptr);
    }
    ZK_HIP(ctx, hipGetLastError());
    if (rem) {
        uint32_t w[8];
        ZK_HIP(ctx, hipMemcpyAsync(w, rem_dev, 32, hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < 4; k++) rem->l[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
    }
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_poly_divide_by_vanishing_dev(zk_ctx* ctx, const void* coeffs_dev, size_t n, uint32_t log_domain, void* q_dev, void* r_dev) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !r_dev || (n && !coeffs_dev) || log_domain > 28) return ZK_ERR_ARG;
    const size_t N = (size_t)1 << log_domain;
    const size_t nq = n > N ? n - N : 0;
    if (nq && !q_dev) return ZK_ERR_ARG;
    if (coeffs_dev == q_dev || coeffs_dev == r_dev) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_poly_divide_by_vanishing_dev: outputs must not alias the input");
    size_t m = (n + N - 1) / N;
    if (m == 0) m = 1;
    size_t G = 65536 / N;
    if (G < 1) G = 1;
    if (G > 1024) G = 1024;
    if (G > m) G = m;
    // (a lane walks the G - g group sums above its own and then its R rows: G R = m is shortest at G = sqrt(m) -- w(X) / v_X of a
    // small Marlin proof, N = 2 and m = 513, was one chain of 513 loads and additions, 0.15 ms)
    {
        size_t sq = 1;
        while (sq * sq < m) sq++;
        if (G > sq) G = sq;
    }
    const size_t R = (m + G - 1) / G;
    G = (m + R - 1) / R;
    void* sums = nullptr;
    if (G > 1) {
        ZK_TRY(zk_scratch(ctx, "poly_divv_sums", G * N * 32, &sums));
        hipLaunchKernelGGL(k_divv_sums, zk_grid(G * N, 256), 256, 0, ctx->stream, coeffs_dev, n, N, R, G, sums);
    }
    hipLaunchKernelGGL(k_divv_fill, zk_grid(G * N, 256), 256, 0, ctx->stream, coeffs_dev, n, N, R, G, (const void*)sums, q_dev, nq, r_dev);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_poly_mul_dev(zk_ctx* ctx, const void* a_dev, size_t na, const void* b_dev, size_t nb, void* out_dev) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !out_dev || !na || !nb || !a_dev || !b_dev) return ZK_ERR_ARG;
    const size_t nout = na + nb - 1;
    uint32_t lg = 0;
    while (((size_t)1 << lg) < nout) lg++;
    const size_t N = (size_t)1 << lg;
    void *ta, *tb;
    ZK_TRY(zk_scratch(ctx, "poly_mul_a", N * 32, &ta));
    ZK_TRY(zk_scratch(ctx, "poly_mul_b", N * 32, &tb));
    ZK_HIP(ctx, hipMemcpyAsync(ta, a_dev, na * 32, hipMemcpyDeviceToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync((char*)ta + na * 32, 0, (N - na) * 32, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(tb, b_dev, nb * 32, hipMemcpyDeviceToDevice, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync((char*)tb + nb * 32, 0, (N - nb) * 32, ctx->stream));
    ZK_TRY(zk_ntt_launch(ctx, ta, lg, 0, 0));
    ZK_TRY(zk_ntt_launch(ctx, tb, lg, 0, 0));
    ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_MUL, ta, tb, ta, N));
    ZK_TRY(zk_ntt_launch(ctx, ta, lg, 1, 0));
    ZK_HIP(ctx, hipMemcpyAsync(out_dev, ta, nout * 32, hipMemcpyDeviceToDevice, ctx->stream));
    return ZK_OK;
    ZK_API_END
}

// KZG10::commit: MSM(powers_of_g, coeffs) + MSM(powers_of_gamma_g, blinding coeffs)
extern "C" int zk_kzg_commit_dev(zk_ctx* ctx, const zk_bases* powers_g, const void* coeffs_dev, size_t n,
                                 const zk_bases* powers_gamma_g, const void* blind_dev, size_t n_blind, zk_g1_projective* out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !powers_g || !out || (n && !coeffs_dev) || powers_g->group != 1) return ZK_ERR_ARG;
    if (n > powers_g->n) ZK_FAIL(ctx, ZK_ERR_ARG, "kzg commit: polynomial degree exceeds the supported degree");   // check_degree_is_too_large
    zk_g1_projective c;
    ZK_TRY(zk_msm_run(ctx, powers_g, 0, coeffs_dev, n, &c));
    if (n_blind) {
        if (!powers_gamma_g || !blind_dev || n_blind > powers_gamma_g->n) ZK_FAIL(ctx, ZK_ERR_ARG, "kzg commit: hiding bound too large");
        zk_g1_projective r;
        ZK_TRY(zk_msm_run(ctx, powers_gamma_g, 0, blind_dev, n_blind, &r));
        zk_g1_add(&c, &r, &c);
    }
    *out = c;
    return ZK_OK;
    ZK_API_END
}

// KZG10::open: w = commit(p / (X - z)) [+ commit_gamma(blind / (X - z))], random_v = blind(z)
extern "C" int zk_kzg_open_dev(zk_ctx* ctx, const zk_bases* powers_g, const void* coeffs_dev, size_t n, const zk_fr* point,
                               const zk_bases* powers_gamma_g, const void* blind_dev, size_t n_blind,
                               zk_g1_projective* w_out, zk_fr* random_v_out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !powers_g || !point || !w_out || (n && !coeffs_dev) || powers_g->group != 1) return ZK_ERR_ARG;
    void* q;
    ZK_TRY(zk_scratch(ctx, "kzg_witness", (n ? n : 1) * 32, &q));
    ZK_TRY(zk_poly_divide_by_linear_dev(ctx, coeffs_dev, n, point, q, nullptr));
    zk_g1_projective w;
    ZK_TRY(zk_msm_run(ctx, powers_g, 0, q, n > 1 ? n - 1 : 0, &w));
    if (n_blind) {
        if (!powers_gamma_g || !blind_dev || !random_v_out) return ZK_ERR_ARG;
        void* qb;
        ZK_TRY(zk_scratch(ctx, "kzg_witness_b", n_blind * 32, &qb));
        ZK_TRY(zk_poly_divide_by_linear_dev(ctx, blind_dev, n_blind, point, qb, random_v_out));   // remainder = blind(z)
        zk_g1_projective wb;
        ZK_TRY(zk_msm_run(ctx, powers_gamma_g, 0, qb, n_blind > 1 ? n_blind - 1 : 0, &wb));
        zk_g1_add(&w, &wb, &w);
    }
    *w_out = w;
    return ZK_OK;
    ZK_API_END
}
