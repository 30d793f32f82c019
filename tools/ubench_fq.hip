// Micro-benchmark: dependent chains of Fq Montgomery products (as in the MSM accumulate loop) at
// 1/2/4/8 waves per SIMD, single- vs dual-accumulator product.  Prints Fq modmul/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../zk-mpc_amd/csrc/fp29.cuh"
using namespace zk;

template <class P>
__device__ __forceinline__ Fp<P> fp_mul2(const Fp<P>& a, const Fp<P>& b) {   // two interleaved accumulators
    constexpr int L = P::L;
    uint32_t m[L], r[L];
    uint64_t acc = 0, acc2 = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { if (i & 1) acc2 += (uint64_t)a.l[i] * b.l[k - i]; else acc += (uint64_t)a.l[i] * b.l[k - i]; }
#pragma unroll
        for (int i = 0; i < k; i++) { if (i & 1) acc += (uint64_t)m[i] * P::P[k - i]; else acc2 += (uint64_t)m[i] * P::P[k - i]; }
        acc += acc2; acc2 = 0;
        m[k] = ((uint32_t)acc * P::INV) & MASK29;
        acc += (uint64_t)m[k] * P::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc2 += (uint64_t)m[i] * P::P[k - i];
        }
        acc += acc2; acc2 = 0;
        r[k - L] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r[L - 1] = (uint32_t)acc;
    return fp_reduce_once<P>(r);
}

constexpr int ITERS = 2000;
template <int V>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* in, uint32_t* out) {
    extern __shared__ uint32_t pad[];
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fq x = fp_unpack<FqParams>(in + 12 * (i & 1023));
    Fq y = fp_unpack<FqParams>(in + 12 * ((i + 7) & 1023));
    for (int k = 0; k < ITERS; k++) {
        if (V == 0) x = fp_mul<FqParams>(x, y);
        else if (V == 1) x = fp_mul2<FqParams>(x, y);
        else if (V == 2) x = fp_sqr<FqParams>(x);
        else if (V == 3) x = fp_add<FqParams>(x, y);
        else { Fq t = fp_mul<FqParams>(x, y); y = fp_mul<FqParams>(y, x); x = t; }   // two independent muls
    }
    if (threadIdx.x == 9999) pad[0] = 1;
    uint32_t w[12];
    fp_pack<FqParams>(w, fp_add<FqParams>(x, y));
    for (int k = 0; k < 12; k++) out[12 * i + k] = w[k];
}

template <int V>
double run(const uint32_t* in, uint32_t* out, int waves_per_simd) {
    // 256 threads = 4 waves = 1 per SIMD per block; blocks/CU limited through dynamic LDS
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
    double ops = (double)blocks * 256 * ITERS * 3 * (V == 4 ? 2 : 1);
    return ops / (ms * 1e-3);
}

int main() {
    uint32_t *in, *out;
    hipMalloc(&in, 1024 * 48); hipMalloc(&out, (size_t)256 * 8 * 256 * 48);
    hipMemset(in, 0x11, 1024 * 48);
    const char* names[5] = {"mul (1 acc)", "mul (2 acc)", "sqr", "add", "2 indep mul"};
    for (int w : {1, 2, 4, 8}) {
        double r[5] = {run<0>(in, out, w), run<1>(in, out, w), run<2>(in, out, w), run<3>(in, out, w), run<4>(in, out, w)};
        for (int v = 0; v < 5; v++) printf("waves/SIMD=%d  %-12s %8.2f Gop/s\n", w, names[v], r[v] / 1e9);
    }
    return 0;
}
