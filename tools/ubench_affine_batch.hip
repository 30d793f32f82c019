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
//   shared   (round 6: VERDICT r5 item 7) the same round with the inversion EXECUTED and shared as widely as one workgroup can:
//            256 lanes x B pairs = 1 024 (B = 4) additions per field inversion.  In-lane prefix products, a product tree over the
//            lanes' totals in LDS (up: the total; down: for every lane the product of all OTHER lanes' totals), lane 0 inverts the
//            total (fp_inv: a chain of ~570 dependent products), every lane takes 1 / (its own total) = inv(total) x others and
//            substitutes back.  6M + 1S per addition + the inversion's latency, which 1 023 lanes spend at a barrier.
//   split    the same with the inversion moved OUT of the kernel (one inversion per launch, Montgomery's trick over the workgroups'
//            totals in a second, one-block kernel; a third kernel substitutes back): kernels A and C timed, kernel B's latency
//            (a single lane's chain) reported beside them -- per ROUND of a bucket tree, and a 2^20-scalar MSM has ~5 rounds.
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

// ---- round 6: the inversion executed, shared by a workgroup (256 lanes x B pairs per inversion) ----
constexpr int SH_THREADS = 256;
using TF = typename F::T;
constexpr int LW = sizeof(TF) / 4;                                   // limbs per element (13)
__device__ __forceinline__ void lds_put(uint32_t* lds, int slot, const TF& v) {
#pragma unroll
    for (int k = 0; k < LW; k++) lds[k * (2 * SH_THREADS) + slot] = v.l[k];
}
__device__ __forceinline__ TF lds_get(const uint32_t* lds, int slot) {
    TF v;
#pragma unroll
    for (int k = 0; k < LW; k++) v.l[k] = lds[k * (2 * SH_THREADS) + slot];
    return v;
}
// MODE 0: everything in one kernel (lane 0 of the block inverts).  MODE 1 (kernel A of the split form): stops after the up-sweep and
// writes the block's total.  MODE 2 (kernel C): takes 1 / total from `inv_in` and finishes.
template <int B, int MODE>
__global__ void __launch_bounds__(SH_THREADS, 2) k_affine_shared(const uint32_t* __restrict__ tab, uint32_t* __restrict__ out, uint32_t* totals,
                                                                   const uint32_t* __restrict__ inv_in) {
    __shared__ uint32_t tree[LW * 2 * SH_THREADS];                    // heap layout: node 1 = root, leaves SH_THREADS .. 2 SH_THREADS - 1
    __shared__ uint32_t outs[LW * 2 * SH_THREADS];                    // the product of everything OUTSIDE a node's subtree
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x;
    uint32_t s = t * 747796405u + 1;
    uint32_t idx[2 * B];
    TF pref[B];
    TF run = F::one();
#pragma clang loop unroll(full)
    for (int j = 0; j < B; j++) {
        idx[2 * j] = rnd(s); idx[2 * j + 1] = rnd(s);
        const TF x1 = felt_load16<F>(tab + (size_t)idx[2 * j] * 24), x2 = felt_load16<F>(tab + (size_t)idx[2 * j + 1] * 24);
        pref[j] = run;
        run = F::mul_l(run, F::template sub_kp<2>(x2, x1));
    }
    lds_put(tree, SH_THREADS + lane, run);
    __syncthreads();
    for (int w = SH_THREADS / 2; w >= 1; w >>= 1) {                   // up: w nodes on this level
        if (lane < w) lds_put(tree, w + lane, F::mul_l(lds_get(tree, 2 * (w + lane)), lds_get(tree, 2 * (w + lane) + 1)));
        __syncthreads();
    }
    if (MODE == 1) {
        if (lane == 0) { const TF tot = F::canon(lds_get(tree, 1)); for (int k = 0; k < LW; k++) totals[(size_t)blockIdx.x * LW + k] = tot.l[k]; }
        return;
    }
    if (lane == 0) {
        TF inv;
        if (MODE == 0) inv = F::inv(F::canon(lds_get(tree, 1)));      // ~570 dependent products on ONE lane
        else for (int k = 0; k < LW; k++) inv.l[k] = inv_in[(size_t)blockIdx.x * LW + k];
        lds_put(outs, 1, inv);                                        // (the root's "outside" carries the inverse down: every leaf ends with inv x others)
    }
    __syncthreads();
    for (int w = 1; w < SH_THREADS; w <<= 1) {                        // down: children of the w nodes of this level
        if (lane < 2 * w) {
            const int node = 2 * w + lane;
            lds_put(outs, node, F::mul_l(lds_get(outs, node >> 1), lds_get(tree, node ^ 1)));
        }
        __syncthreads();
    }
    TF inv = lds_get(outs, SH_THREADS + lane);                        // 1 / (this lane's own total)
#pragma clang loop unroll(full)
    for (int j = B - 1; j >= 0; j--) {
        const Affine<F> P = aff_load16<F>(tab, idx[2 * j]), Q = aff_load16<F>(tab, idx[2 * j + 1]);
        const TF d = F::template sub_kp<2>(Q.x, P.x);
        const TF inv_d = F::mul_l(inv, pref[j]);
        inv = F::mul_l(inv, d);
        const TF lam = F::mul_l(F::template sub_kp<2>(Q.y, P.y), inv_d);
        const TF x3 = F::template sub_kp<2>(F::template sub_kp<2>(F::sqr_l(lam), P.x), Q.x);
        const TF y3 = F::template sub_kp<2>(F::mul_l(lam, F::template sub_kp<6>(P.x, x3)), P.y);
        aff_store16<F>(out, (size_t)t * B + j, Affine<F>{F::canon(x3), F::canon(y3)});
    }
}
// kernel B of the split form: Montgomery's trick over the n workgroup totals, ONE block, one inversion
__global__ void __launch_bounds__(256) k_invert_totals(const uint32_t* totals, uint32_t* inv, uint32_t* scratch, int n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;                  // (a single lane's chain: the latency the split form pays per round)
    TF run = F::one();
    for (int i = 0; i < n; i++) {
        TF v; for (int k = 0; k < LW; k++) v.l[k] = totals[(size_t)i * LW + k];
        for (int k = 0; k < LW; k++) scratch[(size_t)i * LW + k] = run.l[k];
        run = F::mul(run, v);
    }
    TF u = F::inv(run);
    for (int i = n - 1; i >= 0; i--) {
        TF v, pre; for (int k = 0; k < LW; k++) { v.l[k] = totals[(size_t)i * LW + k]; pre.l[k] = scratch[(size_t)i * LW + k]; }
        const TF r = F::mul(u, pre);
        for (int k = 0; k < LW; k++) inv[(size_t)i * LW + k] = r.l[k];
        u = F::mul(u, v);
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
    // ---- round 6: the inversion executed ----
    {
        const int sb = 256 * 8;                                    // 2 048 workgroups of 256 lanes: 2^19 lanes, 2^21 additions per launch at B = 4
        uint32_t *totals, *inv, *scr;
        if (hipMalloc(&totals, (size_t)sb * 13 * 4) != hipSuccess || hipMalloc(&inv, (size_t)sb * 13 * 4) != hipSuccess || hipMalloc(&scr, (size_t)sb * 13 * 4) != hipSuccess) return 1;
        const double adds = (double)sb * SH_THREADS * 4;
        t = timed([&] { hipLaunchKernelGGL((k_affine_shared<4, 0>), sb, SH_THREADS, 0, 0, tab, out, totals, inv); });
        const double s0 = adds / (t * 1e-3) / 1e9;
        printf("shared  1 024 additions per EXECUTED inversion (256 lanes x 4, one kernel)  : %7.3f G add/s  (%.3f ms)  %.2fx\n", s0, t, s0 / xyzz);
        float ta = timed([&] { hipLaunchKernelGGL((k_affine_shared<4, 1>), sb, SH_THREADS, 0, 0, tab, out, totals, inv); });
        float tb = timed([&] { hipLaunchKernelGGL(k_invert_totals, 1, 256, 0, 0, totals, inv, scr, sb); });
        float tc = timed([&] { hipLaunchKernelGGL((k_affine_shared<4, 2>), sb, SH_THREADS, 0, 0, tab, out, totals, inv); });
        const double s1 = adds / ((ta + tc) * 1e-3) / 1e9;
        printf("split   kernels A + C (2^21 additions, the inversion outside)              : %7.3f G add/s  (%.3f + %.3f ms)  %.2fx\n", s1, ta, tc, s1 / xyzz);
        printf("split   kernel B: ONE inversion + Montgomery's trick over %d totals, one lane : %.3f ms of latency per ROUND of a bucket tree\n", sb, tb);
        printf("        (a 2^20-scalar MSM over c = 20 window multiples: 13.6 M additions in ~5 dependent rounds; k_accum<G1> alone: 8.0 G add/s)\n");
    }
    return 0;
}
