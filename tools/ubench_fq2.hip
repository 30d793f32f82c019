// Micro-benchmark: Fq2 multiplication / squaring chains, Karatsuba (3 Montgomery products, 8 add/sub) against the
// "two fused double products" form (a0 b0 + (-5 a1) b1 and a0 b1 + a1 b0, each ONE Montgomery reduction of a sum of two
// limb products; same 1 014 v_mad_u64_u32, fewer field add/sub).  Prints Fq2 op/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../zk-mpc_amd/csrc/fp29.cuh"
using namespace zk;

// Karatsuba reference form (what Fq2Field::mul was before the fused product went into fp29.cuh)
__device__ __forceinline__ Fq2 fq2_mul_karatsuba(const Fq2& a, const Fq2& b) {
    Fq v0 = fp_mul<FqParams>(a.c0, b.c0), v1 = fp_mul<FqParams>(a.c1, b.c1);
    Fq s = fp_mul<FqParams>(fp_add<FqParams>(a.c0, a.c1), fp_add<FqParams>(b.c0, b.c1));
    return Fq2{fp_sub<FqParams>(v0, Fq2Field::mul5(v1)), fp_sub<FqParams>(fp_sub<FqParams>(s, v0), v1)};
}
__device__ __forceinline__ Fq2 fq2_sqr_old(const Fq2& a) {
    Fq v = fp_mul<FqParams>(a.c0, a.c1);
    Fq t = fp_mul<FqParams>(fp_add<FqParams>(a.c0, a.c1), fp_sub<FqParams>(a.c0, Fq2Field::mul5(a.c1)));
    Fq v2 = fp_dbl<FqParams>(v);
    return Fq2{fp_add<FqParams>(t, fp_dbl<FqParams>(v2)), v2};
}
__device__ __forceinline__ Fq2 fq2_mul_fused(const Fq2& a, const Fq2& b) {
    Fq m5a1 = fp_neg<FqParams>(Fq2Field::mul5(a.c1));
    return Fq2{fp_mul2<FqParams>(a.c0, b.c0, m5a1, b.c1), fp_mul2<FqParams>(a.c0, b.c1, a.c1, b.c0)};
}
__device__ __forceinline__ Fq2 fq2_sqr_fused(const Fq2& a) {
    Fq m5a1 = fp_neg<FqParams>(Fq2Field::mul5(a.c1));
    Fq v = fp_mul<FqParams>(a.c0, a.c1);
    return Fq2{fp_mul2<FqParams>(a.c0, a.c0, m5a1, a.c1), fp_dbl<FqParams>(v)};
}

constexpr int ITERS = 600;
template <int V>
__global__ void __launch_bounds__(256) k_chain(const uint32_t* in, uint32_t* out) {
    extern __shared__ uint32_t pad[];
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fq2 x = Fq2Field::load(in + 24 * (i & 511)), y = Fq2Field::load(in + 24 * ((i + 7) & 511));
    for (int k = 0; k < ITERS; k++) {
        if (V == 0) x = fq2_mul_karatsuba(x, y);
        else if (V == 1) x = fq2_mul_fused(x, y);
        else if (V == 2) x = fq2_sqr_old(x);
        else x = fq2_sqr_fused(x);
    }
    if (threadIdx.x == 9999) pad[0] = 1;
    Fq2Field::store(out + 24 * i, Fq2Field::add(x, y));
}

template <int V>
double run(const uint32_t* in, uint32_t* out, int waves_per_simd) {
    size_t lds = 160 * 1024 / waves_per_simd - 1024;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_chain<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_chain<V><<<blocks, 256, lds>>>(in, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_chain<V><<<blocks, 256, lds>>>(in, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256 * ITERS * 3 / (ms * 1e-3);
}

int main() {
    uint32_t *in, *out;
    (void)hipMalloc(&in, 512 * 96);
    (void)hipMalloc(&out, (size_t)256 * 4 * 256 * 96);
    (void)hipMemset(in, 0x11, 512 * 96);
    const char* names[4] = {"mul karatsuba", "mul fused", "sqr current", "sqr fused"};
    for (int w : {1, 2, 4}) {
        double r[4] = {run<0>(in, out, w), run<1>(in, out, w), run<2>(in, out, w), run<3>(in, out, w)};
        for (int v = 0; v < 4; v++) printf("waves/SIMD=%d  %-14s %8.2f G Fq2-op/s\n", w, names[v], r[v] / 1e9);
    }
    return 0;
}
