// msm_reduce.cuh -- the bucket reduction  sum_b (b + 1) * B_b  of a window as a GRID of row and column sums (gfx950).
//
// Replaces the running-sum loop of VariableBaseMSM::multi_scalar_mul (arkworks/algebra/ec/src/msm/variable_base.rs:82-88:
// `running_sum += b; res += running_sum`, two dependent additions per bucket, strictly serial) with two INDEPENDENT
// additions per bucket that fill the chip:
//     bucket b = hi * C + lo  (R = 2^rl rows of C = 2^cl buckets)
//     sum_b (b + 1) B_b  =  C * sum_hi hi * Row_hi  +  sum_lo lo * Col_lo  +  sum_lo Col_lo
//     Row_hi = sum_lo B[hi][lo]      Col_lo = sum_hi B[hi][lo]
// k_grid_l1   every lane sums K = 8 buckets of one row (or one column), TW = 32 lanes then combine through an LDS tree whose
//             active lanes are kept contiguous (whole waves retire, none idles half-masked): 2^19 buckets -> 2 048 row partials
//             and 2 048 column partials in one launch, depth 7 + 5 additions, all of it at full occupancy
// k_grid_bits sum_hi hi * Row_hi = sum_j 2^j * (sum of the rows whose index has bit j): one block per bit (and one for the plain
//             sum), a strided pass over the selected partials and an LDS tree; rl + cl + 1 points per window go to the host,
//             which finishes with one Horner chain (msm.hip: msm_finish_t)
// The previous form (one chunk level of running sums + bit sums over 2^15-bucket slices) did ~3.6 additions per bucket in
// chains of 16 + 24 dependent additions at one wave per SIMD; this one does ~2.1 in chains of 12 + 12.
//
// The kernels are written once over a "point policy" P (msm.hip: G1, one lane per point, lazy Fq domain; msm_g2pair.hip: G2,
// a lane PAIR per point):
//   P::NT lanes per block, P::PTS points per block, P::pt() this lane's point slot, P::X the point type,
//   inf / load / store (packed XYZZ in memory) / add / pack (fit for the packed form) / canon (fully reduced) / lds_put / lds_get.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace zk {

struct GridGeom {
    uint32_t rl, cl;           // R = 2^rl rows, C = 2^cl columns per window
    uint32_t n_win;            // windows (bucket sets) back to back in `sums`
    uint32_t Kr, TWr, Pr, lgPr;   // row sums (over lo): K serial x TW tree lanes; Pr = C / (Kr TWr) partials per row remain
    uint32_t Kc, TWc, Pc;         // column sums (over hi): Pc = R / (Kc TWc) partials per column remain
    uint32_t row_blocks, col_blocks;
};

// kmax: serial additions per lane.  8 when there are buckets enough to fill the chip (2^19 buckets / 8 = 2 waves per SIMD); a
// small bucket set leaves most lanes idle and is bound by the chain's LATENCY: shorter serial runs, deeper trees (a 2^14-scalar
// MSM: 26 windows of 512 buckets, 7 + 2 dependent additions with K = 8, 1 + 4 with K = 2)
static inline void grid_split(uint32_t D, uint32_t kmax, uint32_t& K, uint32_t& TW, uint32_t& P) {
    K = D < kmax ? D : kmax;
    TW = D / K < 32 ? D / K : 32;
    P = D / (K * TW);
}
// log_nb = log2(buckets per window); pts = points per block of the kernels (P::PTS)
static inline GridGeom make_grid_geom(uint32_t log_nb, uint32_t n_win, uint32_t pts) {
    GridGeom g;
    g.cl = log_nb / 2;
    g.rl = log_nb - g.cl;
    g.n_win = n_win;
    const uint64_t buckets = (uint64_t)n_win << log_nb;
    const uint32_t kmax = buckets <= (1u << 15) ? 2u : buckets <= (1u << 17) ? 4u : 8u;
    grid_split(1u << g.cl, kmax, g.Kr, g.TWr, g.Pr);
    grid_split(1u << g.rl, kmax, g.Kc, g.TWc, g.Pc);
    g.lgPr = 0;
    while ((1u << g.lgPr) < g.Pr) g.lgPr++;
    const size_t row_groups = ((size_t)n_win << g.rl) * g.Pr, col_groups = ((size_t)n_win << g.cl) * g.Pc;
    const size_t gr = pts / g.TWr, gc = pts / g.TWc;           // groups per block
    g.row_blocks = (uint32_t)((row_groups + gr - 1) / gr);
    g.col_blocks = (uint32_t)((col_groups + gc - 1) / gc);
    return g;
}
static inline size_t grid_row_points(const GridGeom& g) { return ((size_t)g.n_win << g.rl) * g.Pr; }
static inline size_t grid_col_points(const GridGeom& g) { return ((size_t)g.n_win << g.cl) * g.Pc; }
static inline uint32_t grid_nout(const GridGeom& g) { return g.rl + g.cl + 1; }

// Where k_grid_l1 reads its bucket sums: one buffer (per_log = 0), or -- a group of small MSMs reduced by ONE launch per level, job w
// as "window" w -- the buffers of up to four jobs of 2^per_log buckets each (msm.hip: zk_msm_enqueue_reduce_group).
constexpr int GRID_SRC_MAX = 4;
struct GridSrc {
    const uint32_t* p[GRID_SRC_MAX];
    uint32_t per_log;
};

#ifdef __HIPCC__
__device__ __forceinline__ const uint32_t* grid_src(const GridSrc& s, size_t& idx) {
    if (!s.per_log) return s.p[0];
    const uint32_t w = (uint32_t)(idx >> s.per_log);
    idx &= ((size_t)1 << s.per_log) - 1;
    return s.p[w];
}
// Both kernels run ONE loop whose body holds the only call of P::add: the serial steps (operand from memory) and the tree
// levels (both operands from LDS) are iterations of the same loop.  A point addition is ~7 k instructions (~55 KB): two or
// three inlined copies do not fit the instruction cache a CU pair shares, beside an accumulate kernel's own 38 KB loop.
//
// rowP[(w R + hi) Pr + part], colP[(w Pc + part) C + lo]: partial sums, packed form
template <class P>
__global__ void __launch_bounds__(P::NT, P::MINW)
k_grid_l1(GridSrc sums, uint32_t* __restrict__ rowP, uint32_t* __restrict__ colP, GridGeom g) {
    extern __shared__ uint32_t grid_lds[];
    using X = typename P::X;
    const uint32_t pt = P::pt();
    const bool is_row = blockIdx.x < g.row_blocks;
    const uint32_t R = 1u << g.rl, C = 1u << g.cl;
    const uint32_t TW = is_row ? g.TWr : g.TWc, K = is_row ? g.Kr : g.Kc;
    const uint32_t G = P::PTS / TW;                                   // groups per block
    const uint32_t q = is_row ? pt / TW : pt % G, i = is_row ? pt % TW : pt / G;      // group in the block, lane in the group
    const size_t n_groups = is_row ? ((size_t)g.n_win << g.rl) * g.Pr : ((size_t)g.n_win << g.cl) * g.Pc;
    const size_t g0 = (size_t)(is_row ? blockIdx.x : blockIdx.x - g.row_blocks) * G;
    const size_t gg = g0 + q;
    uint32_t* out = is_row ? rowP : colP;
    size_t base = 0, step = 0;
    if (gg < n_groups) {
        if (is_row) {
            const uint32_t part = (uint32_t)(gg % g.Pr);
            const size_t row = gg / g.Pr;                                          // w R + hi
            base = row * C + (size_t)part * K * TW + i;                            // lanes of a group read neighbouring buckets
            step = TW;
        } else {
            const uint32_t lo = (uint32_t)(gg & (C - 1));
            const size_t rest = gg >> g.cl;
            const uint32_t part = (uint32_t)(rest % g.Pc);
            const size_t w = rest / g.Pc;
            base = (w * R + (size_t)part * K * TW + i) * C + lo;                   // neighbouring lanes read neighbouring columns
            step = (size_t)TW * C;
        }
    }
    uint32_t nlev = 0;
    while ((1u << nlev) < TW) nlev++;
    // Tree over the TW lanes of each group through LDS.  Slot of (group q, lane i) = the lane's own slot: q TW + i (rows),
    // i G + q (columns).  At a level with d active lanes per group the G d additions go to the FIRST G d lanes of the block, so
    // whole waves retire and none runs half-masked.  One barrier per level: a level writes slots (q, i < d), each read at that
    // level by its own writer only; the slots (q, d <= i < 2d) it also reads are not written.
    X acc = P::inf();
    uint32_t a = 0;
    for (uint32_t it = 0; it < K + nlev; it++) {
        X v = P::inf();
        bool act;
        uint32_t qq = 0;
        if (it < K) {
            act = gg < n_groups;
            if (act) {
                size_t idx = base + it * step;
                const uint32_t* src = grid_src(sums, idx);
                v = P::load(src, idx);
            }
        } else {
            if (it == K) P::lds_put(grid_lds, pt, P::pack(acc));
            __syncthreads();
            const uint32_t d = TW >> (it - K + 1);
            act = pt < G * d;
            qq = pt / d;
            const uint32_t ii = pt % d;
            a = is_row ? qq * TW + ii : ii * G + qq;
            if (act) {
                acc = P::lds_get(grid_lds, a);
                v = P::lds_get(grid_lds, is_row ? a + d : a + d * G);
            }
        }
        if (act) acc = P::add(acc, v);
        if (it >= K && it + 1 < K + nlev && act) P::lds_put(grid_lds, a, P::pack(acc));
        if (it + 1 == K + nlev) {
            const size_t og = nlev ? g0 + qq : gg;
            if (act && og < n_groups) P::store(out, og, P::pack(acc));
        }
    }
}

// out[w * nout + j], nout = rl + cl + 1:  j < rl: sum of the rows with bit j;  rl <= j < rl + cl: sum of the columns with bit
// j - rl;  j = rl + cl: sum of all columns.  Fully reduced (the host reads them).
template <class P>
__global__ void __launch_bounds__(P::NT, P::MINW)
k_grid_bits(const uint32_t* __restrict__ rowP, const uint32_t* __restrict__ colP, uint32_t* __restrict__ out, GridGeom g) {
    extern __shared__ uint32_t grid_lds[];
    using X = typename P::X;
    const uint32_t nout = g.rl + g.cl + 1;
    const uint32_t w = blockIdx.x / nout, j = blockIdx.x % nout;
    const uint32_t pt = P::pt();
    const uint32_t* src;
    uint32_t T, pos;                          // T points per window; the selected ones have bit `pos` of their index set
    bool all = false;
    if (j < g.rl) { src = rowP; T = (1u << g.rl) * g.Pr; pos = j + g.lgPr; }
    else { src = colP; T = g.Pc << g.cl; pos = j - g.rl; all = pos == g.cl; }
    const uint32_t nsel = all ? T : T >> 1, low = (1u << pos) - 1;
    const uint32_t nser = (nsel + P::PTS - 1) / P::PTS;
    // tree over the lanes that hold something: lanes >= nsel carry infinity (a small window selects a handful of partials)
    const uint32_t span = nsel < (uint32_t)P::PTS ? (nsel > 1 ? nsel : 2) : (uint32_t)P::PTS;
    uint32_t nlev = 0;
    while ((1u << nlev) < span) nlev++;
    X acc = P::inf();
    for (uint32_t it = 0; it < nser + nlev; it++) {
        X v = P::inf();
        bool act;
        if (it < nser) {
            const uint32_t s = it * P::PTS + pt;                     // the s-th selected index: every lane does the same number
            act = s < nsel;
            const uint32_t t = all ? s : (((s & ~low) << 1) | (1u << pos) | (s & low));
            if (act) v = P::load(src, (size_t)w * T + t);
        } else {
            if (it == nser) P::lds_put(grid_lds, pt, P::pack(acc));
            __syncthreads();
            const uint32_t d = (1u << nlev) >> (it - nser + 1);
            act = pt < d;
            if (act) {
                acc = P::lds_get(grid_lds, pt);
                v = P::lds_get(grid_lds, pt + d);
            }
        }
        if (act) acc = P::add(acc, v);
        if (it >= nser && it + 1 < nser + nlev && act) P::lds_put(grid_lds, pt, P::pack(acc));
        if (it + 1 == nser + nlev && pt == 0) P::store(out, blockIdx.x, P::canon(acc));
    }
}
#endif  // __HIPCC__

}  // namespace zk
