// diag.hip -- diagnostics a measurement needs from the device it runs on (not on any proving path).
//
// zk_diag_int_mad_peak: the issue rate of v_mad_u64_u32 on THIS device, measured when asked -- the roof the MSM kernels are
// priced against ("achieved int-mul-add/s vs measured peak of a pure v_mad_u64_u32 microbenchmark").  Eight
// independent 64-bit accumulator chains per lane, 32 768 rounds, 2 048 blocks of 256 lanes (two blocks per SIMD's worth of
// wave slots on 256 CUs): 1.4e11 lane multiply-adds per launch, ~4.4 ms -- long enough for the clocks to settle.  Same kernel as tools/ubench_int.hip::k_mad64.
//
// zk_diag_fq_pow_dev / zk_diag_fr_pow_dev: base^e as a square-and-multiply chain of DEVICE products -- hundreds of dependent
// Montgomery products through the same fp29.cuh templates the kernels run, in the exact or in the lazy domain.  They exist so that
// the two long-chain known answers the reference's own tests hold (GENERATOR^T == TWO_ADIC_ROOT_OF_UNITY:
// arkworks/curves/bls12_377/src/fields/tests.rs:352-370 for Fq, the same relation from fr.rs's constants for Fr) run on the GPU.
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "frlazy.cuh"
#include "internal.hpp"
#include <algorithm>
#include <vector>

using namespace zk;

namespace {

struct DiagExp { uint32_t w[12]; };

// lane 0 of one wave walks the exponent from its top bit down (Field::pow, ff/src/fields/mod.rs: square, then multiply on a set
// bit); in / out in the reference's Montgomery form.  lazy: products without the final subtraction (fp_mul_lazy: what the
// accumulate kernels run), one canonicalisation at the end.
__global__ void __launch_bounds__(64) k_diag_fq_pow(const uint32_t* base12, DiagExp e, int nbits, int lazy, uint32_t* out12) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    using F = FqField;
    const Fq b = F::ext_to_int(F::load(base12));
    Fq r = F::one();
    bool started = false;
    for (int i = nbits - 1; i >= 0; i--) {
        if (started) r = lazy ? F::sqr_l(r) : F::sqr(r);
        if ((e.w[i >> 5] >> (i & 31)) & 1) { r = started ? (lazy ? F::mul_l(r, b) : F::mul(r, b)) : b; started = true; }
    }
    if (lazy) r = F::canon(r);
    F::store(out12, F::int_to_ext(r));
}
__global__ void __launch_bounds__(64) k_diag_fr_pow(const uint32_t* base8, DiagExp e, int nbits, int lazy, uint32_t* out8) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Fr b = fp_ext_to_int<FrParams>(fr_load(base8, 0));
    Fr r = fp_one<FrParams>();
    bool started = false;
    for (int i = nbits - 1; i >= 0; i--) {
        if (started) r = lazy ? frl_mul(r, frl_canon(r)) : fp_sqr<FrParams>(r);     // frl_mul takes a table entry (< r) on the right
        if ((e.w[i >> 5] >> (i & 31)) & 1) { r = started ? (lazy ? frl_mul(r, b) : fr_mul(r, b)) : b; started = true; }
    }
    if (lazy) r = frl_canon(r);
    fr_store(out8, 0, fp_int_to_ext<FrParams>(r));
}

constexpr int DIAG_ITERS = 32768;
constexpr int DIAG_CH = 8;

__global__ void __launch_bounds__(256) k_diag_mad64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[DIAG_CH];
    const uint32_t x = a + threadIdx.x, y = b + blockIdx.x;
#pragma unroll
    for (int c = 0; c < DIAG_CH; c++) acc[c] = threadIdx.x + c;
    for (int i = 0; i < DIAG_ITERS; i++) {
#pragma unroll
        for (int c = 0; c < DIAG_CH; c++)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(x), "v"(y) : "vcc");
    }
    uint64_t s = 0;
#pragma unroll
    for (int c = 0; c < DIAG_CH; c++) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

extern "C" int zk_diag_int_mad_peak(zk_ctx* ctx, int launches, double* best_mads_per_s, double* median_mads_per_s) {
    ZK_API_BEGIN(ctx)
    if (!ctx || launches < 1 || launches > 1000 || (!best_mads_per_s && !median_mads_per_s)) return ZK_ERR_ARG;
    const unsigned blocks = (unsigned)ctx->n_cu * 8, threads = 256;
    uint64_t* out;
    ZK_TRY(zk_scratch(ctx, "diag_out", (size_t)blocks * threads * 8, (void**)&out));
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    ZK_HIP(ctx, hipEventCreate(&e0));
    ZK_HIP(ctx, hipEventCreate(&e1));
    const double mads = (double)blocks * threads * DIAG_ITERS * DIAG_CH;
    std::vector<double> rate;
    int rc = ZK_OK;
    for (int i = -2; i < launches && rc == ZK_OK; i++) {           // two untimed launches first (clocks, code upload)
        if (hipEventRecord(e0, st) != hipSuccess) rc = ZK_ERR_HIP;
        hipLaunchKernelGGL(k_diag_mad64, blocks, threads, 0, st, out, 3u, 5u);
        if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) rc = ZK_ERR_HIP;
        float ms = 0;
        if (rc == ZK_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = ZK_ERR_HIP;
        if (rc == ZK_OK && i >= 0 && ms > 0) rate.push_back(mads / (ms * 1e-3));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != ZK_OK || rate.empty()) ZK_FAIL(ctx, ZK_ERR_HIP, "zk_diag_int_mad_peak: timing failed");
    std::sort(rate.begin(), rate.end());
    if (best_mads_per_s) *best_mads_per_s = rate.back();
    if (median_mads_per_s) *median_mads_per_s = rate[rate.size() / 2];
    return ZK_OK;
    ZK_API_END
}

namespace {
template <int WORDS64>
int diag_pow(zk_ctx* ctx, const uint64_t* base, const uint64_t* exp, int lazy, uint64_t* out) {
    if (!ctx || !base || !exp || !out) return ZK_ERR_ARG;
    uint32_t* buf;
    ZK_TRY(zk_scratch(ctx, "diag_pow", 2 * WORDS64 * 8, (void**)&buf));
    DiagExp e;
    for (int i = 0; i < 12; i++) e.w[i] = i < 2 * WORDS64 ? (uint32_t)(exp[i / 2] >> (32 * (i & 1))) : 0u;
    ZK_HIP(ctx, hipMemcpyAsync(buf, base, WORDS64 * 8, hipMemcpyHostToDevice, ctx->stream));
    if (WORDS64 == 6) hipLaunchKernelGGL(k_diag_fq_pow, 1, 64, 0, ctx->stream, (const uint32_t*)buf, e, 64 * WORDS64, lazy, buf + 2 * WORDS64);
    else hipLaunchKernelGGL(k_diag_fr_pow, 1, 64, 0, ctx->stream, (const uint32_t*)buf, e, 64 * WORDS64, lazy, buf + 2 * WORDS64);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(out, buf + 2 * WORDS64, WORDS64 * 8, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}
}  // namespace

extern "C" int zk_diag_fq_pow_dev(zk_ctx* ctx, const zk_fq* base, const uint64_t exp[6], int lazy, zk_fq* out) {
    ZK_API_BEGIN(ctx)
    return diag_pow<6>(ctx, base ? base->l : nullptr, exp, lazy, out ? out->l : nullptr);
    ZK_API_END
}
extern "C" int zk_diag_fr_pow_dev(zk_ctx* ctx, const zk_fr* base, const uint64_t exp[4], int lazy, zk_fr* out) {
    ZK_API_BEGIN(ctx)
    return diag_pow<4>(ctx, base ? base->l : nullptr, exp, lazy, out ? out->l : nullptr);
    ZK_API_END
}
