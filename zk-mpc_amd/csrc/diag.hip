// diag.hip -- diagnostics a measurement needs from the device it runs on (not on any proving path).
//
// zk_diag_int_mad_peak: the issue rate of v_mad_u64_u32 on THIS device, measured when asked -- the roof the MSM kernels are
// priced against ("achieved int-mul-add/s vs measured peak of a pure v_mad_u64_u32 microbenchmark").  Eight
// independent 64-bit accumulator chains per lane, 32 768 rounds, 2 048 blocks of 256 lanes (two blocks per SIMD's worth of
// wave slots on 256 CUs): 1.4e11 lane multiply-adds per launch, ~4.4 ms -- long enough for the clocks to settle.  Same kernel as tools/ubench_int.hip::k_mad64.
#include "../../include/zkmpc_hip.h"
#include "ctx.hpp"
#include "internal.hpp"
#include <algorithm>
#include <vector>

namespace {

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
