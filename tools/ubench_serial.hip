// ubench_serial.hip -- should a Montgomery product's column sums take the carry of the column before as the addend of their first
// multiply-add (one serial chain of 337 v_mad_u64_u32, no 64-bit join per column), or start from a constant and be joined with
// the carry by a v_lshl_add_u64 (what the compiler makes of fp29.cuh::fp_mul_lazy: 27 more half-rate instructions, but the
// columns' chains are independent of the carry)?  Dependent chains of Fq products at 1 / 2 / 4 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/ubench_serial.hip -o tools/_bin/ubench_serial
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../zk-mpc_amd/csrc/fp29.cuh"
using namespace zk;

__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint64_t mad64s(uint32_t a, uint32_t b_const, uint64_t c) {      // b in a scalar register
    uint64_t d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(b_const), "v"(c));
    return d;
}

// the same integers as fp_mul_lazy<FqParams>, every multiply-add in ONE chain
__device__ __forceinline__ Fq mul_serial(const Fq& a, const Fq& b) {
    using P = FqParams;
    constexpr int L = P::L, LR = P::LR;
    uint32_t m[LR], r[L];
    uint64_t acc = MASK29;
#pragma unroll
    for (int k = 0; k < LR; k++) {
#pragma unroll
        for (int i = (k >= L ? k - L + 1 : 0); i <= (k < L ? k : L - 1); i++) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
        for (int i = (k >= L ? k - L + 1 : 0); i < k; i++) acc = mad64s(m[i], P::P[k - i], acc);
        m[k] = ~(uint32_t)acc & MASK29;
        acc >>= 29;
        if (k + 1 < LR) acc += MASK29;
    }
#pragma unroll
    for (int k = LR; k < LR + L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
        for (int i = k - L + 1; i < LR; i++) acc = mad64s(m[i], P::P[k - i], acc);
        r[k - LR] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r[L - 1] = (uint32_t)acc;
    Fq o;
#pragma unroll
    for (int i = 0; i < L; i++) o.l[i] = r[i];
    return o;
}

constexpr int ITERS = 2000;
template <int V>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* in, uint32_t* out) {
    extern __shared__ uint32_t pad[];
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fq x = fp_unpack<FqParams>(in + 12 * (i & 1023));
    Fq y = fp_unpack<FqParams>(in + 12 * ((i + 7) & 1023));
    for (int k = 0; k < ITERS; k++) {
        if (V == 0) x = fp_mul_lazy<FqParams>(x, y);
        else x = mul_serial(x, y);
    }
    if (threadIdx.x == 9999) pad[0] = 1;
    uint32_t w[12];
    fp_pack<FqParams>(w, fp_reduce_once<FqParams>(x.l));
    for (int k = 0; k < 12; k++) out[12 * i + k] = w[k];
}

template <int V>
double run(const uint32_t* in, uint32_t* out, int waves_per_simd, uint32_t* first) {
    int blocks_per_cu = waves_per_simd;
    size_t lds = 160 * 1024 / blocks_per_cu - 1024;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_chain<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_chain<V><<<blocks, 256, lds>>>(in, out); (void)hipDeviceSynchronize();
    (void)hipMemcpy(first, out, 48, hipMemcpyDeviceToHost);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_chain<V><<<blocks, 256, lds>>>(in, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256 * ITERS * 3 / (ms * 1e-3);
}

int main() {
    uint32_t *in, *out;
    (void)hipMalloc(&in, 1024 * 48); (void)hipMalloc(&out, (size_t)256 * 8 * 256 * 48);
    (void)hipMemset(in, 0x11, 1024 * 48);
    uint32_t f0[12], f1[12];
    for (int w : {1, 2, 4}) {
        double a = run<0>(in, out, w, f0), b = run<1>(in, out, w, f1);
        bool same = true;
        for (int i = 0; i < 12; i++) same = same && f0[i] == f1[i];
        printf("%d waves/SIMD: compiler's columns %.2f G Fq-mul/s | one serial chain %.2f G Fq-mul/s (%.3fx) | same result: %s\n", w, a / 1e9, b / 1e9,
               b / a, same ? "yes" : "NO");
    }
    return 0;
}
