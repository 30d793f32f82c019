// ubench_affine_batch.hip -- an UPPER BOUND for batched-affine bucket accumulation on gfx950, against the XYZZ mixed addition the
// accumulate kernels use ("batched-affine accumulation stays rejected on paper, not by measurement").
//
//   xyzz     the loop of msm.hip::k_accum<G1>: one accumulator per lane, LEN gathered table points, ec.cuh::xyzz_madd_lazy
//            (8M + 2S in the lazy Fq domain)                                                       -> G additions / s
//   affine   one ROUND of a pairwise tree with Montgomery's trick inside a lane: B pairs (P_j, Q_j) of gathered table points,
//            d_j = x2 - x1, prefix products, [ONE inversion per lane batch -- NOT executed here: the bound assumes it is shared
//            so widely that it is free], back-substitution, lambda = (y2 - y1) / d_j, x3 = lambda^2 - x1 - x2,
//            y3 = lambda (x1 - x3) - y1, the sum stored as a 96-byte affine point: 5M + 1S per addition     -> G additions / s
//
// What the bound leaves out (all of it costs the affine side more): the inversion and the tree / barrier that shares it, the
// further rounds' reads of the sums this round writes (a bucket of 26 points needs 5 rounds), equal-x / infinity handling, the
// bucket bookkeeping.  Verdict rule: the affine side must win by well over 1.3x HERE to be worth building.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/ubench_affine_batch.hip -o tools/_bin/ubench_affine_batch
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../zk-mpc_amd/csrc/devutil.cuh"
using namespace zk;
using F = G1Field;

constexpr uint32_t LOG_T = 22;                 // table of 2^22 points x 96 B = 403 MB: gathers miss every cache
constexpr uint32_t T = 1u << LOG_T;

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return (s >> 7) & (T - 1); }

__global__ void __launch_bounds__(256) k_fill(uint32_t* tab) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)T * 24; i += (size_t)gridDim.x * blockDim.x)
        tab[i] = ((uint32_t)i * 2654435761u) >> ((i % 12 == 11) ? 8 : 0);          // arbitrary residues below 2^376 (not curve points)
}

template <int LEN>
__global__ void __launch_bounds__(256, 1) k_xyzz(const uint32_t* __restrict__ tab, uint32_t* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = t * 747796405u + 1;
    XYZZ<F> acc = xyzz_inf<F>();
    Affine<F> p = aff_load16<F>(tab, rnd(s));
    for (int k = 0; k < LEN; k++) {
        Affine<F> cur = p;
        p = aff_load16<F>(tab, rnd(s));                                             // next point prefetched, as k_accum does
        acc = xyzz_madd_lazy<F>(acc, cur);
    }
    xyzz_store16<F>(out, t, xyzz_canon_lazy<F>(acc));
}

template <int B, int ROUNDS>
__global__ void __launch_bounds__(256, 1) k_affine(const uint32_t* __restrict__ tab, uint32_t* __restrict__ out) {
    using T_ = typename F::T;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s0 = t * 747796405u + 1;
    T_ carry = F::one();
    for (int r = 0; r < ROUNDS; r++) {
        uint32_t s = s0;
        T_ pref[B];
        T_ run = carry;
#pragma clang loop unroll(full)
        for (int j = 0; j < B; j++) {                                               // forward: d_j and the prefix products
            const uint32_t i1 = rnd(s), i2 = rnd(s);
            const T_ x1 = felt_load16<F>(tab + (size_t)i1 * 24), x2 = felt_load16<F>(tab + (size_t)i2 * 24);
            pref[j] = run;
            run = F::mul_l(run, F::template sub_kp<2>(x2, x1));
        }
        T_ inv = run;                                                               // (the shared inversion would return 1 / run here)
        s = s0;
        uint32_t idx[2 * B];
#pragma clang loop unroll(full)
        for (int j = 0; j < 2 * B; j++) idx[j] = rnd(s);
#pragma clang loop unroll(full)
        for (int j = B - 1; j >= 0; j--) {                                          // backward: 1 / d_j, then the addition itself
            const Affine<F> P = aff_load16<F>(tab, idx[2 * j]), Q = aff_load16<F>(tab, idx[2 * j + 1]);
            const T_ d = F::template sub_kp<2>(Q.x, P.x);
            const T_ inv_d = F::mul_l(inv, pref[j]);
            inv = F::mul_l(inv, d);
            const T_ lam = F::mul_l(F::template sub_kp<2>(Q.y, P.y), inv_d);
            const T_ x3 = F::template sub_kp<2>(F::template sub_kp<2>(F::sqr_l(lam), P.x), Q.x);
            const T_ y3 = F::template sub_kp<2>(F::mul_l(lam, F::template sub_kp<6>(P.x, x3)), P.y);
            aff_store16<F>(out, (size_t)t * B + j, Affine<F>{F::canon(x3), F::canon(y3)});
        }
        carry = inv;
        s0 = s;
    }
}

template <class L>
static float timed(L launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; r++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

int main() {
    uint32_t *tab, *out;
    if (hipMalloc(&tab, (size_t)T * 96) != hipSuccess) return 1;
    const int blocks = 256 * 8, threads = 256;                     // two waves per SIMD, as the accumulate kernel runs
    const size_t lanes = (size_t)blocks * threads;
    if (hipMalloc(&out, lanes * 8 * 192) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_fill, 4096, 256, 0, 0, tab);
    constexpr int LEN = 26;                                        // the mean bucket of a 2^20-scalar MSM with c = 20
    float t = timed([&] { hipLaunchKernelGGL(k_xyzz<LEN>, blocks, threads, 0, 0, tab, out); });
    const double xyzz = lanes * (double)LEN / (t * 1e-3) / 1e9;
    printf("xyzz    mixed additions (8M + 2S, lazy domain), %d per lane          : %7.3f G add/s  (%.3f ms)\n", LEN, xyzz, t);
    constexpr int R = 4;
    t = timed([&] { hipLaunchKernelGGL((k_affine<4, R>), blocks, threads, 0, 0, tab, out); });
    const double a4 = lanes * 4.0 * R / (t * 1e-3) / 1e9;
    printf("affine  batch of 4 per lane (5M + 1S, inversion NOT counted)          : %7.3f G add/s  (%.3f ms)  %.2fx\n", a4, t, a4 / xyzz);
    t = timed([&] { hipLaunchKernelGGL((k_affine<8, R>), blocks, threads, 0, 0, tab, out); });
    const double a8 = lanes * 8.0 * R / (t * 1e-3) / 1e9;
    printf("affine  batch of 8 per lane                                           : %7.3f G add/s  (%.3f ms)  %.2fx\n", a8, t, a8 / xyzz);
    printf("bound: the affine side needs well over 1.3x here (inversion, its sharing, 4 more rounds of traffic all come on top)\n");
    return 0;
}
