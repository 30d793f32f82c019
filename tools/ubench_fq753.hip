// Micro-benchmark: dependent chains of MNT4-753 Fq (26 x 29-bit limbs) Montgomery products / squares / adds at
// 1/2/3 waves per SIMD.  Prints op/s and the v_mad_u64_u32 rate the products account for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../zk-mpc_amd/csrc/fp29.cuh"
using namespace zk;
using F7 = Fp<Fq753Params>;

constexpr int ITERS = 400;
template <int V>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* in, uint32_t* out) {
    extern __shared__ uint32_t pad[];
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    F7 x = fp_unpack<Fq753Params>(in + 24 * (i & 1023));
    F7 y = fp_unpack<Fq753Params>(in + 24 * ((i + 7) & 1023));
    for (int k = 0; k < ITERS; k++) {
        if (V == 0) x = fp_mul<Fq753Params>(x, y);
        else if (V == 1) x = fp_sqr<Fq753Params>(x);
        else if (V == 2) x = fp_add<Fq753Params>(x, y);
        else { F7 t = fp_mul<Fq753Params>(x, y); F7 u = fp_add<Fq753Params>(x, t); y = fp_sub<Fq753Params>(x, t); x = u; }  // butterfly
    }
    if (threadIdx.x == 9999) pad[0] = 1;
    uint32_t w[24];
    fp_pack<Fq753Params>(w, fp_add<Fq753Params>(x, y));
    for (int k = 0; k < 24; k++) out[24 * i + k] = w[k];
}

template <int V>
double run(const uint32_t* in, uint32_t* out, int waves_per_simd) {
    int blocks_per_cu = waves_per_simd;
    size_t lds = 160 * 1024 / blocks_per_cu - 1024;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k_chain<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_chain<V><<<blocks, 256, lds>>>(in, out); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_chain<V><<<blocks, 256, lds>>>(in, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256 * ITERS * 3 / (ms * 1e-3);
}

int main() {
    uint32_t *in, *out;
    hipMalloc(&in, 1024 * 96); hipMalloc(&out, (size_t)256 * 8 * 256 * 96);
    hipMemset(in, 0x11, 1024 * 96);
    const char* names[4] = {"mul", "sqr", "add", "butterfly"};
    for (int w : {1, 2, 3}) {
        double r[4] = {run<0>(in, out, w), run<1>(in, out, w), run<2>(in, out, w), run<3>(in, out, w)};
        for (int v = 0; v < 4; v++)
            printf("waves/SIMD=%d  %-10s %8.2f Gop/s  (%.1f T mad/s)\n", w, names[v], r[v] / 1e9,
                   v == 2 ? 0.0 : r[v] * (v == 1 ? 26 * 27 / 2 + 26 * 26 : 2 * 26 * 26) / 1e12);
    }
    return 0;
}
