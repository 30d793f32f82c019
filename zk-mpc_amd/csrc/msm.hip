// msm.hip -- variable-base multi-scalar multiplication on G1 / G2 for gfx950.
//
// Replaces VariableBaseMSM::multi_scalar_mul (arkworks/algebra/ec/src/msm/variable_base.rs:11-106)
// behind AffineCurve::multi_scalar_mul (ec/src/lib.rs:305-314) and its MPC override
// (mpc-algebra/src/wire/pairing.rs:714-777 -> share/additive.rs:517-520 -> share/msm.rs:31-37).
//
// The reference: c = ln(n)+2 bit unsigned windows, one bucket array per window filled by a
// sequential sweep, running-sum reduction, Horner over windows.  This implementation computes the
// same group element sum_i s_i P_i with a GPU schedule:
//   1. k_digits   : scalar -> canonical integer (one Montgomery product), + bias so that every
//                   c-bit window becomes an independent SIGNED digit in [-2^(c-1), 2^(c-1)-1]
//                   (halves the bucket count); histogram of |digit| per window (L2 atomics).
//   2. k_scan     : per-window exclusive scan of the histogram -> bucket offsets.
//   3. k_scatter  : counting sort of (point index, sign) by bucket.  Order inside a bucket is
//                   whatever the atomics give: the group is commutative and the result is reduced
//                   to its canonical affine form, so the output bits do not depend on it.
//   1-3 for MSMs of >= 2^16 digits: the bucket sort of msm_sort.hip (bins, then per-bin LDS counting; zero digits never become
//                   pairs; no global atomics per digit) over ONE set of all buckets; the segment length is fitted to the input.
//   4. k_accum    : one thread per bucket: gather its points (96 B / 192 B random reads), mixed
//                   XYZZ additions (8M+2S in Fq / Fq2).  This is the dominant kernel; it is bound by
//                   the integer ALU (v_mad_u64_u32), not by HBM.
//   5. k_reduce   : sum_b b * B_b per window by chunked running sums, K=8 buckets per thread per
//                   level, log_8(2^(c-1)) levels (all windows in one launch per level).
//   6. host       : Horner over the <= 64 window sums, one inversion, output as Jacobian Z=1.
// Arithmetic is exact, so zero scalars, repeated bases, P + P, P - P and infinity need no special
// treatment beyond the complete formulas in ec.cuh.
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "hostgroup.hpp"
#include "hostfield64.hpp"
#include "internal.hpp"
#include "msm_reduce.cuh"
#include "msm_digits.cuh"
#include "ec_dual.cuh"
#include <algorithm>
#include <future>
#include <map>
#include <memory>
#include <string.h>
#include <vector>

using namespace zk;

namespace {

struct MsmPlan {
    uint32_t c;        // widest window (bits): buckets per window NB = 2^(c-1)
    uint32_t W;        // windows
    uint32_t NB;       // buckets per window
    uint32_t bias[9];  // sum_w 2^(off[w+1]-1): makes every window an independent signed digit
    uint16_t off[65];  // window w covers bits [off[w], off[w+1])
};

// 253-bit scalars + 2 bits of headroom for the signed-digit bias = 255 bits.  Uniform c-bit windows leave a nearly
// empty TOP window for most c (253 mod c is 1 for c = 12 and 14, 0 for c = 11, ...): two or three buckets then
// receive half of all points each, and the histogram / scatter atomics serialise on them (measured: the sort
// phase of a 2^18 MSM took 3.7 ms instead of ~0.5 ms).  So the 255 bits are spread EVENLY: W = ceil(255/c)
// windows of floor(255/W) or ceil(255/W) bits.  With precomputed window multiples (forced_c) the windows must
// be uniform, because the table holds 2^(c w) * base.
MsmPlan make_plan(size_t n, uint32_t forced_c = 0) {
    MsmPlan p;
    uint32_t lg = 0;                                   // round(log2 n)
    while (((size_t)3 << lg) <= 2 * n) lg++;           // 1.5 * 2^lg <= n  ->  round up
    // plain tables (no window multiples): up to 2^15 terms a job is a chain of latencies -- its accumulate kernel lasts as long as
    // its fullest bucket -- and wider windows (fewer terms per bucket, more buckets for the tree-shaped reduce) shorten it:
    // measured on whole proofs, c = lg - 2 against lg - 4: 2^10 2.24 -> 2.15 ms, 2^12 2.45 -> 2.22, 2^14 2.64 -> 2.43
    int c = (int)lg - (lg <= 15 ? 2 : 4);
    if (c < 4) c = 4;
    if (c > 16) c = 16;
    if (forced_c) c = (int)forced_c;
    p.W = (255 + (uint32_t)c - 1) / (uint32_t)c;
    for (int i = 0; i < 9; i++) p.bias[i] = 0;
    if (forced_c) {
        for (uint32_t w = 0; w <= p.W; w++) p.off[w] = (uint16_t)(w * c);
        p.c = (uint32_t)c;
    } else {
        const uint32_t base = 255 / p.W, extra = 255 % p.W;   // the `extra` lowest windows get one more bit
        uint32_t o = 0;
        for (uint32_t w = 0; w < p.W; w++) { p.off[w] = (uint16_t)o; o += base + (w < extra ? 1 : 0); }
        p.off[p.W] = (uint16_t)o;
        p.c = base + (extra ? 1 : 0);
    }
    p.NB = 1u << (p.c - 1);
    for (uint32_t w = 0; w < p.W; w++) {
        uint32_t bit = p.off[w + 1] - 1u;
        p.bias[bit >> 5] |= 1u << (bit & 31);
    }
    return p;
}

// dig[w*n + i] = |d| | (d<0 ? 1<<31 : 0), and counts[w*NB + |d| - 1]++ for |d| > 0.
__global__ void __launch_bounds__(256)
k_digits(const void* scalars, size_t n, WinOff wo, uint32_t W, uint32_t NB, Bias bias, uint32_t* dig, uint32_t* counts, int merged) {
    __shared__ uint32_t kw[9][256];
    const uint32_t tid = threadIdx.x;
    for (size_t i0 = blockIdx.x * (size_t)256; i0 < n; i0 += (size_t)gridDim.x * 256) {
        size_t i = i0 + tid;
        if (i < n) {
            uint32_t w9[9];
            scalar_biased_words(scalars, i, bias, w9);
#pragma unroll
            for (int k = 0; k < 9; k++) kw[k][tid] = w9[k];
            for (uint32_t w = 0; w < W; w++) {
                const uint32_t bit = wo.off[w], wi = bit >> 5;
                uint64_t two = kw[wi][tid];
                if (wi + 1 < 9) two |= (uint64_t)kw[wi + 1][tid] << 32;
                const int32_t d = signed_digit(two, bit, wo.off[w + 1] - bit);
                uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
                dig[(size_t)w * n + i] = mag | (d < 0 ? 0x80000000u : 0u);
                if (mag) atomicAdd(&counts[(merged ? 0 : (size_t)w * NB) + mag - 1], 1u);
            }
        }
    }
}

// sorted[w*n + pos] = i | sign ; counts are consumed (count down to zero).
// merged (precomputed window multiples): one bucket set; the entry is the TABLE index w*n_tab + tab_off + i.
__global__ void __launch_bounds__(256)
k_scatter(const uint32_t* dig, size_t n, uint32_t W, uint32_t NB, const uint32_t* offs, uint32_t* counts, uint32_t* sorted,
          int merged, uint32_t n_tab, uint32_t tab_off) {
    const size_t total = (size_t)W * n;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        uint32_t d = dig[t];
        uint32_t mag = d & 0x7fffffffu;
        if (!mag) continue;
        size_t w = t / n, i = t - w * n;
        if (merged) {
            uint32_t slot = atomicSub(&counts[mag - 1], 1u) - 1;
            sorted[offs[mag - 1] + slot] = (uint32_t)(w * n_tab + tab_off + i) | (d & 0x80000000u);
        } else {
            uint32_t slot = atomicSub(&counts[w * NB + mag - 1], 1u) - 1;
            sorted[w * n + offs[w * (NB + 1) + mag - 1] + slot] = (uint32_t)i | (d & 0x80000000u);
        }
    }
}


// ---- segments -------------------------------------------------------------------------------
// A bucket with cnt points is cut into max(1, ceil(cnt / SEG)) segments.  Single-segment buckets
// write their sum straight to sums[key]; the segments of a split ("heavy") bucket write partial
// sums to sums[n_keys + ...] and k_fold adds them up.  SEG is ~4x the mean bucket size, so
// for uniformly random scalars no bucket is split; splitting is what keeps 0/1-heavy witness
// vectors, repeated scalars and a nearly empty top window from serialising on one thread.
struct SegDesc { uint32_t start, len, dst; };
// parent: HEAVY_NONE, or -- for the group of a bucket that is folded in two levels -- the index of its second-level entry
struct HeavyDesc { uint32_t key, first, nseg, parent; };
constexpr uint32_t HEAVY_NONE = 0xffffffffu;

// One block per window: exclusive scans of counts (-> offs) and of the per-bucket segment counts.
__global__ void __launch_bounds__(1024)
k_scan(const uint32_t* counts, uint32_t* offs, uint32_t* seg_local, uint32_t* win_segs, uint32_t NB, const uint32_t* segp) {
    __shared__ uint32_t part[1024], part2[1024];
    const uint32_t seg = *segp;
    const uint32_t w = blockIdx.x, tid = threadIdx.x;
    const uint32_t per = (NB + 1023) / 1024;
    const uint32_t lo = tid * per, hi = min(lo + per, NB);
    uint32_t s = 0, s2 = 0;
    for (uint32_t b = lo; b < hi; b++) {
        uint32_t c = counts[(size_t)w * NB + b];
        s += c;
        s2 += c ? (c + seg - 1) / seg : 1;
    }
    part[tid] = s;
    part2[tid] = s2;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0, v2 = tid >= d ? part2[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        part2[tid] += v2;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0, run2 = tid ? part2[tid - 1] : 0;
    for (uint32_t b = lo; b < hi; b++) {
        uint32_t c = counts[(size_t)w * NB + b];
        offs[(size_t)w * (NB + 1) + b] = run;
        seg_local[(size_t)w * NB + b] = run2;
        run += c;
        run2 += c ? (c + seg - 1) / seg : 1;
    }
    if (tid == 1023) {
        offs[(size_t)w * (NB + 1) + NB] = part[1023];
        win_segs[w] = part2[1023];
    }
}

// ctr[0] = entries of the heavy list, ctr[1] = heavy segments, ctr[2] = total segments, ctr[3] = the segment length (written by the
// sort: msm_sort.hip::k_scan_bins, or copied from the host's plan), ctr[4] = non-zero digits, ctr[5] = entries of the second-level
// heavy list, ctr[6] = group slots handed out; hist[len] = #segments of that length.
// A split bucket's segment sums are folded back by k_fold.  A bucket cut into more than FOLD_GROUP segments (the "1" bucket of a
// boolean witness: hundreds of thousands of points) is folded in two levels: its segments in groups of FOLD_GROUP, one heavy-list
// entry per group writing a group sum (slots from grp_base on), and one second-level entry that adds the group sums up -- so that
// no block walks more than FOLD_GROUP partial sums.  The descriptors of a split bucket are written by the whole block (a lane
// that met a 15 000-segment bucket used to write them alone: 0.3 ms).
constexpr uint32_t FOLD_GROUP = 256;
constexpr uint32_t FOLD_LIGHT = 8;      // up to this many partial sums: one lane adds them; more: a 64-lane block (strided sums + an LDS tree)
struct HeavyFill { uint32_t base, rest, start, cnt, dst0; };      // segment 0 at desc[base], segments 1.. at desc[rest ...]

__global__ void __launch_bounds__(256)
k_build_segs(const uint32_t* offs, const uint32_t* seg_local, const uint32_t* win_segs, size_t n, uint32_t W, uint32_t NB,
             SegDesc* desc, HeavyDesc* heavy, HeavyDesc* heavy2, uint32_t* ctr, uint32_t* hist, uint32_t grp_base) {
    extern __shared__ uint32_t lh[];  // seg + 1 bins
    __shared__ HeavyFill q[64];
    __shared__ uint32_t qn;
    const uint32_t seg = ctr[3];
    // seg_local == NULL (one bucket set, after the bucket sort): no scan of segment counts at all -- bucket t's first segment is
    // descriptor t, the further segments of a split bucket are appended behind the last bucket (ctr[7] hands out the places), and
    // k_len_scan closes the count.  Four kernels fewer on the sort's chain, the front of every proof.
    const bool flat = seg_local == nullptr;
    for (uint32_t i = threadIdx.x; i <= seg; i += blockDim.x) lh[i] = 0;
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const size_t total = (size_t)W * NB;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        uint32_t w = (uint32_t)(t / NB), b = (uint32_t)(t - (size_t)w * NB);
        uint32_t base = (uint32_t)t;
        if (!flat) {
            base = seg_local[t];
            for (uint32_t k = 0; k < w; k++) base += win_segs[k];
        }
        uint32_t lo = offs[(size_t)w * (NB + 1) + b], hi = offs[(size_t)w * (NB + 1) + b + 1];
        uint32_t cnt = hi - lo, start = (uint32_t)(w * n) + lo;
        if (cnt <= seg) {
            desc[base] = SegDesc{start, cnt, (uint32_t)t};
            atomicAdd(&lh[cnt], 1u);
        } else {
            const uint32_t ns = (cnt + seg - 1) / seg;
            const uint32_t dst0 = (uint32_t)total + atomicAdd(&ctr[1], ns);
            if (ns <= FOLD_GROUP) {
                heavy[atomicAdd(&ctr[0], 1u)] = HeavyDesc{(uint32_t)t, dst0, ns, HEAVY_NONE};
            } else {
                const uint32_t ng = (ns + FOLD_GROUP - 1) / FOLD_GROUP;
                const uint32_t g0 = grp_base + atomicAdd(&ctr[6], ng), h0 = atomicAdd(&ctr[0], ng), up = atomicAdd(&ctr[5], 1u);
                for (uint32_t j = 0; j < ng; j++) heavy[h0 + j] = HeavyDesc{g0 + j, dst0 + j * FOLD_GROUP, min(FOLD_GROUP, ns - j * FOLD_GROUP), up};
                heavy2[up] = HeavyDesc{(uint32_t)t, g0, ng, HEAVY_NONE};
            }
            atomicAdd(&lh[seg], ns - 1);
            atomicAdd(&lh[cnt - (ns - 1) * seg], 1u);
            const uint32_t rest = flat ? (uint32_t)total + atomicAdd(&ctr[7], ns - 1) : base + 1;
            const uint32_t slot = atomicAdd(&qn, 1u);
            if (slot < 64) {
                q[slot] = HeavyFill{base, rest, start, cnt, dst0};
            } else {                     // more split buckets than the block's queue holds: this lane writes its own
                for (uint32_t j = 0; j < ns; j++) desc[j ? rest + j - 1 : base] = SegDesc{start + j * seg, min(seg, cnt - j * seg), dst0 + j};
            }
        }
        if (!flat && t == total - 1) ctr[2] = base + (cnt <= seg ? 1 : (cnt + seg - 1) / seg);
    }
    __syncthreads();
    const uint32_t nq = min(qn, 64u);
    for (uint32_t e = 0; e < nq; e++) {
        const HeavyFill f = q[e];
        const uint32_t ns = (f.cnt + seg - 1) / seg;
        for (uint32_t j = threadIdx.x; j < ns; j += blockDim.x)
            desc[j ? f.rest + j - 1 : f.base] = SegDesc{f.start + j * seg, min(seg, f.cnt - j * seg), f.dst0 + j};
    }
    for (uint32_t i = threadIdx.x; i <= seg; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// bin_start[len] for a DESCENDING order by length (longest segments first); one block.
// flat_buckets != 0 (k_build_segs' flat mode): the segment count is closed here, buckets + appended segments.
__global__ void __launch_bounds__(512) k_len_scan(const uint32_t* hist, uint32_t* bin_start, uint32_t* bin_cursor, uint32_t* ctr, uint32_t flat_buckets) {
    __shared__ uint32_t part[512];
    const uint32_t seg = ctr[3];
    if (flat_buckets && threadIdx.x == 0) ctr[2] = flat_buckets + ctr[7];
    const uint32_t tid = threadIdx.x, nb = seg + 1;
    const uint32_t per = (nb + 511) / 512;
    // position p = seg - len  (p = 0 is the longest)
    uint32_t lo = tid * per, hi = min(lo + per, nb), s = 0;
    for (uint32_t p = lo; p < hi; p++) s += hist[seg - p];
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 512; d <<= 1) {
        uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0;
    for (uint32_t p = lo; p < hi; p++) {
        bin_start[seg - p] = run;
        bin_cursor[seg - p] = run;
        run += hist[seg - p];
    }
}

// order[pos] = segment id, grouped by length (descending).  Per-block LDS counting, one global atomic per (block, bin).
__global__ void __launch_bounds__(256)
k_order(const SegDesc* desc, const uint32_t* ctr, uint32_t* bin_cursor, uint32_t* order) {
    extern __shared__ uint32_t lh[];  // [0..seg]: counts then base
    const uint32_t S = ctr[2], seg = ctr[3];
    const uint32_t per_block = (S + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per_block, hi = min(lo + per_block, S);
    for (uint32_t i = threadIdx.x; i <= seg; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    for (uint32_t s = lo + threadIdx.x; s < hi; s += blockDim.x) atomicAdd(&lh[desc[s].len], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= seg; i += blockDim.x) {
        uint32_t c = lh[i];
        lh[i] = c ? atomicAdd(&bin_cursor[i], c) : 0;
    }
    __syncthreads();
    for (uint32_t s = lo + threadIdx.x; s < hi; s += blockDim.x) order[atomicAdd(&lh[desc[s].len], 1u)] = s;
}

// The whole sort of a SMALL MSM over a table of window multiples (one bucket set, <= 2^16 digits, <= 2^15 buckets) as ONE block:
// digits -> bucket counts in LDS -> offsets, segment descriptors and the heavy list -> entries to their buckets -> segments ordered by
// length.  It replaces four memsets and six launches (k_digits, k_scan, k_scatter, k_build_segs, k_len_scan, k_order): a small MSM
// is a chain of launch latencies, ~10 us apiece on the sort's critical path, and the commitments of a small Marlin proof are
// fifteen of them.  Same products as those kernels (consecutive segment numbering, as their non-flat mode).
constexpr uint32_t SS_NT = 1024;
constexpr uint32_t SS_MAX_ENTRIES = 1u << 16, SS_MAX_NB = 1u << 15, SS_MAX_SEG = 64;   // (70 k entries in one block: 157 us against ~105 for the six launches)
struct SortSmallArgs {
    const void* scalars; uint32_t n; WinOff wo; uint32_t W, NB; Bias bias; uint32_t seg, n_tab, tab_off, grp_base;
    uint32_t* dig; uint32_t* sorted; SegDesc* desc; HeavyDesc* heavy; HeavyDesc* heavy2; uint32_t* order; uint32_t* ctr;
};
__device__ __forceinline__ void sort_small_body(const SortSmallArgs& a) {
    const void* scalars = a.scalars;
    const uint32_t n = a.n, W = a.W, NB = a.NB, seg = a.seg, n_tab = a.n_tab, tab_off = a.tab_off, grp_base = a.grp_base;
    const WinOff& wo = a.wo;
    const Bias& bias = a.bias;
    uint32_t* dig = a.dig; uint32_t* sorted = a.sorted; SegDesc* desc = a.desc; HeavyDesc* heavy = a.heavy; HeavyDesc* heavy2 = a.heavy2;
    uint32_t* order = a.order; uint32_t* ctr = a.ctr;
    extern __shared__ uint32_t ss_lds[];
    uint32_t* cnt = ss_lds;                       // NB: counts, later cursors
    uint32_t* part = cnt + NB;                    // 2 x SS_NT: the block scan
    uint32_t* lh = part + 2 * SS_NT;              // seg + 1: segments per length, later their cursors
    uint32_t* sc = lh + (SS_MAX_SEG + 1);         // [0] heavy entries [1] heavy segments [5] second-level entries [6] group slots
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = tid; b < NB; b += SS_NT) cnt[b] = 0;
    for (uint32_t i = tid; i <= seg; i += SS_NT) lh[i] = 0;
    if (tid < 8) sc[tid] = 0;
    __syncthreads();
    // digits
    for (uint32_t i = tid; i < n; i += SS_NT) {
        uint32_t w9[9];
        scalar_biased_words(scalars, i, bias, w9);
        for (uint32_t w = 0; w < W; w++) {
            const uint32_t bit = wo.off[w], wi = bit >> 5;
            uint64_t two = w9[0];
#pragma unroll
            for (int k = 1; k < 9; k++) if (wi == (uint32_t)k) two = w9[k];
            uint32_t hi = 0;
#pragma unroll
            for (int k = 1; k < 9; k++) if (wi + 1 == (uint32_t)k) hi = w9[k];
            two |= (uint64_t)hi << 32;
            const int32_t d = signed_digit(two, bit, wo.off[w + 1] - bit);
            const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
            dig[(size_t)w * n + i] = mag | (d < 0 ? 0x80000000u : 0u);
            if (mag) atomicAdd(&cnt[mag - 1], 1u);
        }
    }
    __syncthreads();
    // offsets and segment numbers: lane t owns buckets [t per, t per + per)
    const uint32_t per = (NB + SS_NT - 1) / SS_NT;
    const uint32_t lo = tid * per, hi = min(lo + per, NB);
    uint32_t s = 0, s2 = 0;
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t c = cnt[b];
        s += c;
        s2 += c ? (c + seg - 1) / seg : 1;
    }
    part[tid] = s;
    part[SS_NT + tid] = s2;
    __syncthreads();
    for (uint32_t d = 1; d < SS_NT; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0, v2 = tid >= d ? part[SS_NT + tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        part[SS_NT + tid] += v2;
        __syncthreads();
    }
    const uint32_t S = part[2 * SS_NT - 1], T = part[SS_NT - 1];
    uint32_t run = tid ? part[tid - 1] : 0, run2 = tid ? part[SS_NT + tid - 1] : 0;
    for (uint32_t b = lo; b < hi; b++) {
        const uint32_t c = cnt[b];
        cnt[b] = run;                             // from here on: the bucket's cursor
        if (c <= seg) {
            desc[run2] = SegDesc{run, c, b};
            atomicAdd(&lh[c], 1u);
            run2 += 1;
        } else {
            const uint32_t ns = (c + seg - 1) / seg;
            const uint32_t dst0 = NB + atomicAdd(&sc[1], ns);
            if (ns <= FOLD_GROUP) {
                heavy[atomicAdd(&sc[0], 1u)] = HeavyDesc{b, dst0, ns, HEAVY_NONE};
            } else {
                const uint32_t ng = (ns + FOLD_GROUP - 1) / FOLD_GROUP;
                const uint32_t g0 = grp_base + atomicAdd(&sc[6], ng), h0 = atomicAdd(&sc[0], ng), up = atomicAdd(&sc[5], 1u);
                for (uint32_t j = 0; j < ng; j++) heavy[h0 + j] = HeavyDesc{g0 + j, dst0 + j * FOLD_GROUP, min(FOLD_GROUP, ns - j * FOLD_GROUP), up};
                heavy2[up] = HeavyDesc{b, g0, ng, HEAVY_NONE};
            }
            for (uint32_t j = 0; j < ns; j++) desc[run2 + j] = SegDesc{run + j * seg, min(seg, c - j * seg), dst0 + j};
            atomicAdd(&lh[seg], ns - 1);
            atomicAdd(&lh[c - (ns - 1) * seg], 1u);
            run2 += ns;
        }
        run += c;
    }
    __syncthreads();
    // the entries to their buckets (merged form: entry = the table index of the window multiple)
    // (four digits in flight per lane: one block has sixteen waves to hide a global load behind, and 20 - 130 entries per lane)
    const uint32_t total = W * n;
    for (uint32_t t0 = tid; t0 < total; t0 += 4 * SS_NT) {
        uint32_t d[4];
#pragma unroll
        for (int u = 0; u < 4; u++) d[u] = t0 + u * SS_NT < total ? dig[t0 + u * SS_NT] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t mag = d[u] & 0x7fffffffu;
            if (!mag) continue;
            const uint32_t t = t0 + u * SS_NT, w = t / n, i = t - w * n;
            sorted[atomicAdd(&cnt[mag - 1], 1u)] = (w * n_tab + tab_off + i) | (d[u] & 0x80000000u);
        }
    }
    // segments by length, longest first: lh[len] becomes the first position of that length
    __syncthreads();
    if (tid == 0) {
        uint32_t pos = 0;
        for (uint32_t len = seg + 1; len-- > 0;) { const uint32_t c = lh[len]; lh[len] = pos; pos += c; }
        ctr[0] = sc[0]; ctr[1] = sc[1]; ctr[2] = S; ctr[3] = seg; ctr[4] = T; ctr[5] = sc[5]; ctr[6] = sc[6]; ctr[7] = 0;
    }
    __syncthreads();
    for (uint32_t q0 = tid; q0 < S; q0 += 4 * SS_NT) {
        uint32_t len[4];
#pragma unroll
        for (int u = 0; u < 4; u++) len[u] = q0 + u * SS_NT < S ? desc[q0 + u * SS_NT].len : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < 4; u++) if (len[u] != 0xffffffffu) order[atomicAdd(&lh[len[u]], 1u)] = q0 + u * SS_NT;
    }
}
__global__ void __launch_bounds__(SS_NT) k_sort_small(SortSmallArgs a) { sort_small_body(a); }
// ... of a GROUP of small jobs: one block per job (the commitments of a small Marlin round: four sorts of 50 - 100 us, two of them
// one behind the other on a stream, were a third of the round's device time)
struct SortSmallGroup { SortSmallArgs j[GRID_SRC_MAX]; };
__global__ void __launch_bounds__(SS_NT) k_sort_small_group(SortSmallGroup g) { sort_small_body(g.j[blockIdx.x]); }

// The operations of ec_dual.cuh for G1: a lane PAIR per point addition (small, latency-bound jobs: the accumulate kernel, the folds
// and the grid reduce below).  Both lanes carry the whole point; the even one stores.
struct G1DualOps {
    using F = G1Field;
    using T = typename F::T;
    static __device__ __forceinline__ bool hi() { return (threadIdx.x & 1u) != 0; }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return F::mul_l(a, b); }
    static __device__ __forceinline__ T mul_wide(const T& a, const T& b) { return F::mul_l(a, b); }    // (any operands below ~7p)
    static __device__ __forceinline__ T swap(const T& a) {
        T r;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(T) / sizeof(uint32_t)); i++)
            r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
        return r;
    }
    static __device__ __forceinline__ T zero() { return F::zero(); }
    static __device__ __forceinline__ T one() { return F::one(); }
    static __device__ __forceinline__ XYZZ<F> dbl_affine(const Affine<F>& q) { return xyzz_dbl_affine<F>(q); }
    static __device__ __forceinline__ bool is_zero(const T& a) { return F::is_zero(a); }
    static __device__ __forceinline__ bool maybe_multiple_of_p(const T& a) { return F::maybe_multiple_of_p(a); }
    static __device__ __forceinline__ T select(bool c, const T& a, const T& b) { return F::select(c, a, b); }
    template <int K> static __device__ __forceinline__ T sub_kp(const T& a, const T& b) { return F::template sub_kp<K>(a, b); }
    static __device__ __forceinline__ T x3_l(const T& rr, const T& ppp, const T& qq) { return F::x3_l(rr, ppp, qq); }
    static __device__ __forceinline__ T canon(const T& a) { return F::canon(a); }
    static __device__ __forceinline__ T canon1(const T& a) { return F::canon1(a); }
    static __device__ __forceinline__ XYZZ<F> dbl(const XYZZ<F>& a) { return xyzz_dbl<F>(a); }
};

// Persistent lanes over the length-sorted segment list: the grid is exactly 2 blocks per CU and thread g
// takes segments g, g + G, g + 2G, ... (G = total threads).  Neighbouring lanes always hold segments of
// (nearly) equal length, every thread gets the same long-to-short mix, and -- unlike a grid with one thread
// per segment -- the kernel never has thousands of blocks queued in front of the small sort / reduce
// kernels of the other in-flight MSMs (which were starved for milliseconds behind that queue).  The next
// point is fetched while the current one is added.
// G1 only (G2 runs on lane pairs: msm_g2pair.hip).  Left to the compiler the kernel takes 230 VGPRs = 2 waves / SIMD; forcing
// 3 waves makes it spill (176 B of scratch per lane) and the proof slower: the kernel is issue-bound, not latency-bound.
// Holding it to 224 registers (amdgpu_num_vgpr(112): counted in pairs on gfx90a+) so that two resident waves leave 64 free
// registers per SIMD lane for the sort kernels of the next job (k_digits needs 56, k_build_segs 64; scans in 512-thread
// blocks) was measured too: no change for Groth16 (23.7 ms) or Marlin (86.7 ms) -- the co-running sorts are not waiting
// for registers.
#ifndef ZK_ACCUM_WAVES_G1
#define ZK_ACCUM_WAVES_G1 1
#endif
#ifndef ZK_ACCUM_WAVES_G2
#define ZK_ACCUM_WAVES_G2 1
#endif
template <class F, bool LIMB_TABLE, bool DUAL = false>
__device__ __forceinline__ void accum_body(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted, const SegDesc* __restrict__ desc,
                                           const uint32_t* __restrict__ order, const uint32_t* __restrict__ ctr, uint32_t* __restrict__ sums, uint32_t stride) {
    // stride: words between consecutive points of `bases` (2 * WORDS packed; 32 for a G1 table of window multiples: one point per line)
    // The NEXT point is fetched as raw words while the current one is added and unpacked into limbs only when its turn comes: the
    // wait for the gather then sits behind a whole mixed addition.  (Fetching it in unpacked form put the wait, and the unpacking,
    // in front of the addition: the loads had nothing to hide behind but the other wave.)
    // stride 64 (G1 window multiples, fixed_base.hip::k_repack_limbs): the table holds limbs, and the negative of every point in
    // the second line of its slot -- the entry's sign bit picks the line; nothing is unpacked or negated here.
    constexpr int LIMBS = sizeof(typename F::T) / sizeof(uint32_t);
    constexpr int NW = LIMB_TABLE ? (2 * LIMBS + 3) / 4 : 2 * F::WORDS / 4;
    struct Raw { uint4 v[NW]; };
    auto fetch = [&](uint32_t e) {
        const uint4* w = reinterpret_cast<const uint4*>(bases + (size_t)(e & 0x7fffffffu) * stride + (LIMB_TABLE ? (e >> 31) * 32u : 0u));
        Raw r;
#pragma unroll
        for (int i = 0; i < NW; i++) r.v[i] = w[i];
        return r;
    };
    auto unpack = [&](const Raw& r) {
        uint32_t t[4 * NW];
#pragma unroll
        for (int i = 0; i < NW; i++) { t[4 * i] = r.v[i].x; t[4 * i + 1] = r.v[i].y; t[4 * i + 2] = r.v[i].z; t[4 * i + 3] = r.v[i].w; }
        if constexpr (LIMB_TABLE) {
            Affine<F> a;
#pragma unroll
            for (int i = 0; i < LIMBS; i++) { a.x.l[i] = t[i]; a.y.l[i] = t[LIMBS + i]; }
            return a;
        } else {
            return Affine<F>{F::load(t), F::load(t + F::WORDS)};
        }
    };
    const uint32_t S = ctr[2];
    const uint32_t G = (gridDim.x * blockDim.x) >> (DUAL ? 1 : 0);
    for (uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> (DUAL ? 1 : 0); t < S; t += G) {
        const SegDesc d = desc[order[t]];
        XYZZ<F> acc = xyzz_inf<F>();
        if (d.len) {
            const uint32_t* srt = sorted + d.start;
            uint32_t e = srt[0], e1 = d.len > 1 ? srt[1] : 0;         // the sorted entries run one step further ahead still
            Raw nxt = fetch(e);
            for (uint32_t k = 0; k < d.len; k++) {
                Affine<F> cur = unpack(nxt);
                const uint32_t ce = e;
                if (k + 1 < d.len) {
                    e = e1;
                    nxt = fetch(e);
                    if (k + 2 < d.len) e1 = srt[k + 2];
                }
                // lazy domain (fp29.cuh / ec.cuh::xyzz_madd_lazy): the accumulator is a representative in [0, ~5 p], a negative
                // digit takes p - y in one carry pass; nothing is compared or selected until the segment is through
                const bool inf = aff_is_inf<F>(cur);
                if (!LIMB_TABLE && (ce >> 31)) cur.y = F::template kp_minus<1>(cur.y);
                if (!inf) {
                    if constexpr (DUAL) acc = xyzz_madd_dual<G1DualOps, XYZZ<F>, Affine<F>>(acc, cur);
                    else acc = xyzz_madd_lazy<F>(acc, cur);
                }
            }
            if constexpr (DUAL) acc = XYZZ<F>{F::canon(acc.x), F::canon(acc.y), F::canon1(acc.zz), F::canon1(acc.zzz)};   // (y < 3p + eps there)
            else acc = xyzz_canon_lazy<F>(acc);
        }
        if (DUAL && G1DualOps::hi()) continue;
        xyzz_store16<F>(sums, d.dst, acc);
    }
}
template <class F, bool LIMB_TABLE = false>
__global__ void __launch_bounds__(256, (F::WORDS == 12 ? ZK_ACCUM_WAVES_G1 : ZK_ACCUM_WAVES_G2))
k_accum(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted, const SegDesc* __restrict__ desc,
        const uint32_t* __restrict__ order, const uint32_t* __restrict__ ctr, uint32_t* __restrict__ sums, uint32_t stride) {
    accum_body<F, LIMB_TABLE>(bases, sorted, desc, order, ctr, sums, stride);
}
// A GROUP of small jobs as one launch (blockIdx.y = job): a job of 2^10 .. 2^15 terms fills a tenth of the chip for the length of
// its longest segment; four of them one behind the other on a stream were most of a small proof's device time.
constexpr int MSM_GROUP_MAX = GRID_SRC_MAX;
struct AccumArgs { const uint32_t* bases; const uint32_t* sorted; const SegDesc* desc; const uint32_t* order; const uint32_t* ctr; uint32_t* sums; uint32_t stride; };
struct AccumGroup { AccumArgs j[MSM_GROUP_MAX]; };
template <class F, bool LIMB_TABLE, bool DUAL = false>
__global__ void __launch_bounds__(256, (F::WORDS == 12 ? ZK_ACCUM_WAVES_G1 : ZK_ACCUM_WAVES_G2)) k_accum_group(AccumGroup g) {
    const AccumArgs& a = g.j[blockIdx.y];
    accum_body<F, LIMB_TABLE, DUAL>(a.bases, a.sorted, a.desc, a.order, a.ctr, a.sums, a.stride);
}

// LDS staging of one XYZZ point per lane, word-major (word k of lane t at lds[k * NT + t]): consecutive lanes hit
// consecutive banks.  (A lane-major layout has a stride of 48/96 words between lanes = every lane on one bank.)
template <class F, int NT>
__device__ __forceinline__ void lds_put_xyzz(uint32_t* lds, uint32_t t, const XYZZ<F>& p) {
    uint32_t w[4 * F::WORDS];
    xyzz_store<F>(w, p);
#pragma unroll
    for (int k = 0; k < 4 * F::WORDS; k++) lds[k * NT + t] = w[k];
}
template <class F, int NT>
__device__ __forceinline__ XYZZ<F> lds_get_xyzz(const uint32_t* lds, uint32_t t) {
    uint32_t w[4 * F::WORDS];
#pragma unroll
    for (int k = 0; k < 4 * F::WORDS; k++) w[k] = lds[k * NT + t];
    return xyzz_load<F>(w);
}

// The running sums of the bucket reduction: G1 in the lazy domain (ec.cuh::xyzz_add_lazy: no conditional subtraction on the
// main path, ~16 % fewer instructions per addition); G2 has its own lane-pair chain (msm_g2pair.hip) and keeps the exact form
// here.  radd_pack: a sum made fit for the packed 12-word form (LDS tree, the level buffers); radd_canon: fully reduced
// (what leaves for the host).
template <class F>
__device__ __forceinline__ XYZZ<F> radd(const XYZZ<F>& a, const XYZZ<F>& b) {
    if constexpr (F::WORDS == 12) return xyzz_add_lazy<F>(a, b);
    else return xyzz_add<F>(a, b);
}
template <class F>
__device__ __forceinline__ XYZZ<F> rpack(const XYZZ<F>& a) {
    if constexpr (F::WORDS == 12) return xyzz_packable_lazy<F>(a);
    else return a;
}
template <class F>
__device__ __forceinline__ XYZZ<F> rcanon(const XYZZ<F>& a) {
    if constexpr (F::WORDS == 12) return xyzz_canon_lazy<F>(a);
    else return a;
}

// Split buckets are folded back into one sum per bucket by ONE launch (every launch of a reduce chain waits for a free
// slot beside the running accumulate kernel, so fewer launches is a shorter chain): the first `light_blocks` blocks take
// buckets with few segments, one thread per bucket adding them up serially; the remaining blocks take the heavy
// ones (repeated scalars, 0/1 witnesses), one 64-lane block per entry: strided partial sums, then an LDS tree.
// A bucket folded in two levels (k_build_segs) has one entry per group of FOLD_GROUP segments; the block that finishes the LAST
// group of a bucket (a counter per second-level entry in `done`, which it leaves at zero again) adds the group sums up.
// DUAL (small jobs, G1): every addition on a lane pair (ec_dual.cuh) -- 32 pairs per block instead of 64 lanes; sums are brought back
// to reduced coordinates where they are stored (the exact and the lazy readers both take them).
template <class F, bool DUAL>
__device__ __forceinline__ XYZZ<F> fold_add(const XYZZ<F>& a, const XYZZ<F>& b) {
    if constexpr (DUAL) return xyzz_add_dual<G1DualOps, XYZZ<F>>(a, b);
    else return radd<F>(a, b);
}
template <class F, bool DUAL>
__device__ __forceinline__ XYZZ<F> fold_pack(const XYZZ<F>& a) {
    if constexpr (DUAL) return XYZZ<F>{F::canon(a.x), F::canon(a.y), a.zz, a.zzz};
    else return rpack<F>(a);
}
template <class F, bool DUAL>
__device__ __forceinline__ XYZZ<F> fold_canon(const XYZZ<F>& a) {
    if constexpr (DUAL) return XYZZ<F>{F::canon(a.x), F::canon(a.y), F::canon1(a.zz), F::canon1(a.zzz)};
    else return rcanon<F>(a);
}
template <class F, bool DUAL = false>
__device__ __forceinline__ void fold_block(uint32_t* lds, uint32_t* sums, uint32_t first, uint32_t nseg, uint32_t key, uint32_t tid) {
    constexpr uint32_t SH = DUAL ? 1 : 0, NG = 64 >> SH;
    const uint32_t lt = tid >> SH;
    const bool writer = !DUAL || (tid & 1u) == 0;
    XYZZ<F> acc = xyzz_inf<F>();
    for (uint32_t j = lt; j < nseg; j += NG) acc = fold_add<F, DUAL>(acc, xyzz_load16<F>(sums, (size_t)first + j));
    if (writer) lds_put_xyzz<F, 64>(lds, lt, fold_pack<F, DUAL>(acc));
    __syncthreads();
    for (uint32_t d = NG / 2; d >= 1; d >>= 1) {
        XYZZ<F> r = acc;
        if (lt < d) r = fold_pack<F, DUAL>(fold_add<F, DUAL>(lds_get_xyzz<F, 64>(lds, lt), lds_get_xyzz<F, 64>(lds, lt + d)));
        __syncthreads();
        if (lt < d && writer) lds_put_xyzz<F, 64>(lds, lt, r);
        __syncthreads();
    }
    if (tid == 0) xyzz_store16<F>(sums, key, fold_canon<F, DUAL>(lds_get_xyzz<F, 64>(lds, 0)));
    __syncthreads();
}

template <class F, bool DUAL = false>
__device__ __forceinline__ void fold_body(const HeavyDesc* heavy, const HeavyDesc* heavy2, const uint32_t* ctr, uint32_t* done, uint32_t* sums, uint32_t light_blocks) {
    extern __shared__ uint32_t lds[];  // 64 * XW words (heavy blocks only)
    __shared__ uint32_t last_flag;
    constexpr uint32_t SH = DUAL ? 1 : 0, NG = 64 >> SH;
    const uint32_t nheavy = ctr[0];
    const uint32_t tid = threadIdx.x, lt = tid >> SH;
    const bool writer = !DUAL || (tid & 1u) == 0;
    if (blockIdx.x < light_blocks) {
        for (uint32_t hb = blockIdx.x * NG + lt; hb < nheavy; hb += light_blocks * NG) {
            const HeavyDesc h = heavy[hb];
            if (h.nseg > FOLD_LIGHT || h.parent != HEAVY_NONE) continue;
            XYZZ<F> acc = xyzz_load16<F>(sums, (size_t)h.first);
            for (uint32_t j = 1; j < h.nseg; j++) acc = fold_add<F, DUAL>(acc, xyzz_load16<F>(sums, (size_t)h.first + j));
            if (writer) xyzz_store16<F>(sums, h.key, fold_canon<F, DUAL>(acc));
        }
        return;
    }
    for (uint32_t hb = blockIdx.x - light_blocks; hb < nheavy; hb += gridDim.x - light_blocks) {
        const HeavyDesc h = heavy[hb];
        if (h.nseg <= FOLD_LIGHT && h.parent == HEAVY_NONE) continue;
        fold_block<F, DUAL>(lds, sums, h.first, h.nseg, h.key, tid);
        if (h.parent == HEAVY_NONE) continue;
        const HeavyDesc up = heavy2[h.parent];
        if (tid == 0) {
            __threadfence();                                       // the group sum above is visible before the count is
            const uint32_t before = atomicAdd(&done[h.parent], 1u);
            last_flag = before + 1 == up.nseg ? 1u : 0u;
            if (last_flag) done[h.parent] = 0;                     // (ready for the next MSM that uses this slot)
        }
        __syncthreads();
        if (last_flag) {
            __threadfence();
            fold_block<F, DUAL>(lds, sums, up.first, up.nseg, up.key, tid);
        }
        __syncthreads();
    }
}
template <class F>
__global__ void __launch_bounds__(64) k_fold(const HeavyDesc* heavy, const HeavyDesc* heavy2, const uint32_t* ctr, uint32_t* done, uint32_t* sums, uint32_t light_blocks) {
    fold_body<F>(heavy, heavy2, ctr, done, sums, light_blocks);
}
struct FoldArgs { const HeavyDesc* heavy; const HeavyDesc* heavy2; const uint32_t* ctr; uint32_t* done; uint32_t* sums; };
struct FoldGroup { FoldArgs j[MSM_GROUP_MAX]; };
template <class F, bool DUAL = false>
__global__ void __launch_bounds__(64) k_fold_group(FoldGroup g, uint32_t light_blocks) {
    const FoldArgs& a = g.j[blockIdx.y];
    fold_body<F, DUAL>(a.heavy, a.heavy2, a.ctr, a.done, a.sums, light_blocks);
}

// The bucket reduction proper is msm_reduce.cuh (row / column sums of the bucket grid, then bit sums); this is its point policy
// for G1: one lane per point, sums in the lazy domain, 256 points per block (48 KiB of LDS for the trees).
struct RedG1 {
    using F = G1Field;
    using X = XYZZ<F>;
    static constexpr int NT = 256, PTS = 256, MINW = 2;      // <= 256 registers: a wave shares a SIMD with an accumulate wave
    static __device__ __forceinline__ uint32_t pt() { return threadIdx.x; }
    static __device__ __forceinline__ X inf() { return xyzz_inf<F>(); }
    static __device__ __forceinline__ X load(const uint32_t* base, size_t i) { return xyzz_load16<F>(base, i); }
    static __device__ __forceinline__ void store(uint32_t* base, size_t i, const X& p) { xyzz_store16<F>(base, i, p); }
    static __device__ __forceinline__ X add(const X& a, const X& b) { return radd<F>(a, b); }
    static __device__ __forceinline__ X pack(const X& a) { return rpack<F>(a); }
    static __device__ __forceinline__ X canon(const X& a) { return rcanon<F>(a); }
    static __device__ __forceinline__ void lds_put(uint32_t* lds, uint32_t slot, const X& p) { lds_put_xyzz<F, PTS>(lds, slot, p); }
    static __device__ __forceinline__ X lds_get(const uint32_t* lds, uint32_t slot) { return lds_get_xyzz<F, PTS>(lds, slot); }
};

// ... for a SMALL bucket set (a chain of dependent additions on waves that sit alone on their SIMD): every addition on a lane PAIR,
// seven product times instead of fourteen (ec_dual.cuh).  Both lanes of a pair carry the whole point; the even one stores.
struct RedG1Dual {
    using F = G1Field;
    using X = XYZZ<F>;
    static constexpr int NT = 256, PTS = 128, MINW = 1;
    static __device__ __forceinline__ uint32_t pt() { return threadIdx.x >> 1; }
    static __device__ __forceinline__ X inf() { return xyzz_inf<F>(); }
    static __device__ __forceinline__ X load(const uint32_t* base, size_t i) { return xyzz_load16<F>(base, i); }
    static __device__ __forceinline__ void store(uint32_t* base, size_t i, const X& p) { if (!G1DualOps::hi()) xyzz_store16<F>(base, i, p); }
    static __device__ __forceinline__ X add(const X& a, const X& b) { return xyzz_add_dual<G1DualOps, X>(a, b); }
    // (y is a difference of two products here, < 3p + eps: brought below p with x wherever a sum is packed)
    static __device__ __forceinline__ X pack(const X& a) { return X{F::canon(a.x), F::canon(a.y), a.zz, a.zzz}; }
    static __device__ __forceinline__ X canon(const X& a) { return X{F::canon(a.x), F::canon(a.y), F::canon1(a.zz), F::canon1(a.zzz)}; }
    static __device__ __forceinline__ void lds_put(uint32_t* lds, uint32_t slot, const X& p) { if (!G1DualOps::hi()) lds_put_xyzz<F, PTS>(lds, slot, p); }
    static __device__ __forceinline__ X lds_get(const uint32_t* lds, uint32_t slot) { return lds_get_xyzz<F, PTS>(lds, slot); }
};
// bucket sets up to this size take the pair form (measured: 2^15 gains 20 - 25 % of the chain, 2^17 nothing -- the lanes are no longer alone)
constexpr size_t RED_DUAL_MAX_BUCKETS = (size_t)1 << 16;

// Arkworks-layout affine points (Montgomery R = 2^384) -> packed internal form.  all-zero = infinity stays zero.
template <class F>
__global__ void __launch_bounds__(256) k_bases_import(const uint32_t* in, uint32_t* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Affine<F> p = aff_load16<F>(in, i);
        p.x = F::ext_to_int(p.x);
        p.y = F::ext_to_int(p.y);
        aff_store16<F>(out, i, p);
    }
}
template <class F>
__global__ void __launch_bounds__(256) k_bases_export(const uint32_t* in, uint32_t* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Affine<F> p = aff_load16<F>(in, i);
        p.x = F::int_to_ext(p.x);
        p.y = F::int_to_ext(p.y);
        aff_store16<F>(out, i, p);
    }
}

// The device work of one MSM in three phases that can sit on different streams:
//   sort   (digits, histogram, counting sort, segment list)        latency / atomics bound
//   accum  (bucket accumulation)                                    integer-ALU bound, fills the chip
//   reduce (fold split buckets, chunk level, bit-decomposition sums, async copy of the partial sums)
// With several MSMs in flight the accumulate kernels go back to back on one stream while the sort
// and reduce phases of the neighbouring jobs run beside them on other streams.
template <class F>
struct MsmBufs {
    uint32_t *dig, *sorted, *counts, *offs, *seg_local, *small, *order, *sums, *rowP, *colP, *bits, *fold_done;
    SegDesc* desc;
    HeavyDesc *heavy, *heavy2;
};

template <class F>
int msm_prepare_t(zk_ctx* ctx, ZkMsmJob* job, const zk_bases* bases, size_t base_offset, const void* scalars, size_t n, int slot) {
    constexpr size_t XW = 4 * F::WORDS;
    job->group = bases->group;
    job->n = n;
    job->slot = slot;
    job->scalars = scalars;
    if (n == 0) return ZK_OK;
    if (base_offset + n > bases->n) ZK_FAIL(ctx, ZK_ERR_ARG, "msm: base range out of bounds");
    if (n >= ((size_t)1 << 27)) ZK_FAIL(ctx, ZK_ERR_ARG, "msm: n must be < 2^27");
    // window multiples: one bucket set of 2^(c_pre - 1) buckets whatever n is -- worth it when this MSM uses a fair share of the
    // table (a 100-term MSM over a 2^20-point table would reduce 2^19 buckets for nothing)
    const bool merged = bases->pre != nullptr && (n >= 4096 || n * 8 >= bases->n);
    const MsmPlan p = make_plan(n, merged ? bases->c_pre : 0);
    job->c = p.c; job->W = p.W; job->NB = p.NB;
    for (int i = 0; i < 65; i++) job->off[i] = p.off[i];
    if (merged) {
        if (p.W > bases->W_pre) ZK_FAIL(ctx, ZK_ERR_STATE, "msm: precomputed table has too few windows");
        job->Wb = 1;
        job->bases_dev = bases->pre;
        job->n_tab = (uint32_t)bases->n;
        job->tab_off = (uint32_t)base_offset;
        job->stride = bases->pre_stride ? bases->pre_stride : 2 * F::WORDS;
    } else {
        job->Wb = p.W;
        job->bases_dev = bases->dev + base_offset * (2 * F::WORDS);
        job->stride = 2 * F::WORDS;
    }
    // Segment length: long enough that a typical bucket (mean n/NB points) is one segment, short enough that
    // (a) there are at least as many segments as resident lanes (2 waves per SIMD) and (b) the longest serial
    // chain -- seg mixed additions by one lane, ~9 us each -- stays a small part of the kernel.
    const size_t lanes = (size_t)ctx->n_cu * 4 * 64 * 2;
    const size_t per_lane = ((size_t)n * p.W + lanes - 1) / lanes;
    // (a small MSM is a chain of latencies: its accumulate kernel lasts as long as its longest segment -- 16 - 20 terms in the
    // buckets of a 7-bit top window at 2^10, 0.25 ms -- so segments of 8; the halves of a split bucket meet in k_fold)
    uint32_t seg = (size_t)n * p.W <= ((size_t)1 << 19) ? 8 : 32;
    while (seg < per_lane && seg < 4096) seg <<= 1;
    job->seg = seg;
    // Reduce phase geometry (msm_reduce.cuh): every bucket set is a grid of 2^rl x 2^cl buckets, rl + cl = log2(NB)
    uint32_t lg = 0;
    while ((1u << lg) < p.NB) lg++;
    job->log_nb = lg;
    job->nout = lg + 1;
    return ZK_OK;
}

template <class F>
int msm_bufs_t(zk_ctx* ctx, ZkMsmJob* job, MsmBufs<F>& b, bool need_sort) {
    constexpr size_t XW = 4 * F::WORDS;
    const size_t n = job->n, W = job->W, NB = job->NB, seg = job->seg, Wb = job->Wb;
    const size_t nbuck = Wb * NB;
    // The segment length is settled on the device, per input, anywhere in [32, seg] (msm_sort.hip::k_scan_bins): T non-zero digits
    // are cut into segments of at least T / lanes, so an input never has more than max(W n / seg, lanes) full segments.
    const size_t lanes = (size_t)ctx->n_cu * 4 * 64 * 2;
    const size_t cap_segs = std::max(W * n / seg, lanes);
    job->max_segs = nbuck + cap_segs + W;                     // every bucket >= 1 segment
    job->max_heavy_segs = 2 * cap_segs + W;                   // segments of split buckets (a split bucket of cnt points has <= 2 cnt / seg)
    job->max_heavy2 = job->max_heavy_segs / FOLD_GROUP + 2;   // buckets folded in two levels
    job->max_groups = job->max_heavy_segs / FOLD_GROUP + job->max_heavy2 + 2;
    job->max_heavy = cap_segs + job->max_groups + 1;          // heavy-list entries: split buckets and groups
    char nm[64];
    auto slotname = [&](const char* base) { snprintf(nm, sizeof nm, "%s.%d", base, job->slot); return nm; };
    if (need_sort) {
        ZK_TRY(zk_scratch(ctx, slotname("msm_dig"), W * n * 4, (void**)&b.dig));
        ZK_TRY(zk_scratch(ctx, slotname("msm_sorted"), W * n * 4, (void**)&b.sorted));
        ZK_TRY(zk_scratch(ctx, slotname("msm_counts"), nbuck * 4, (void**)&b.counts));
        ZK_TRY(zk_scratch(ctx, slotname("msm_offs"), Wb * (NB + 1) * 4, (void**)&b.offs));
        ZK_TRY(zk_scratch(ctx, slotname("msm_segl"), nbuck * 4, (void**)&b.seg_local));
        // small: win_segs[64] | ctr[8] | hist[seg+1] | bin_start[seg+1] | bin_cursor[seg+1]
        ZK_TRY(zk_scratch(ctx, slotname("msm_small"), (64 + 8 + 3 * (seg + 1)) * 4, (void**)&b.small));
        ZK_TRY(zk_scratch(ctx, slotname("msm_desc"), job->max_segs * sizeof(SegDesc), (void**)&b.desc));
        ZK_TRY(zk_scratch(ctx, slotname("msm_heavy"), job->max_heavy * sizeof(HeavyDesc), (void**)&b.heavy));
        ZK_TRY(zk_scratch(ctx, slotname("msm_heavy2"), job->max_heavy2 * sizeof(HeavyDesc), (void**)&b.heavy2));
        ZK_TRY(zk_scratch(ctx, slotname("msm_order"), job->max_segs * 4, (void**)&b.order));
    }
    ZK_TRY(zk_scratch(ctx, slotname("msm_sums"), (nbuck + job->max_heavy_segs + job->max_groups) * XW * 4, (void**)&b.sums));
    ZK_TRY(zk_scratch_zeroed(ctx, slotname("msm_fold_done"), job->max_heavy2 * 4, (void**)&b.fold_done));
    const GridGeom gg = make_grid_geom(job->log_nb, (uint32_t)Wb, 256);     // the partial counts do not depend on the block size
    ZK_TRY(zk_scratch(ctx, slotname("msm_rowP"), grid_row_points(gg) * XW * 4, (void**)&b.rowP));
    ZK_TRY(zk_scratch(ctx, slotname("msm_colP"), grid_col_points(gg) * XW * 4, (void**)&b.colP));
    ZK_TRY(zk_scratch(ctx, slotname("msm_bits"), (size_t)Wb * job->nout * XW * 4, (void**)&b.bits));
    return ZK_OK;
}

template <class F>
int msm_enqueue_sort_t(zk_ctx* ctx, ZkMsmJob* job, hipStream_t st, const ZkMsmJob* share) {
    if (job->n == 0) return ZK_OK;
    const bool g1 = F::WORDS == 12;
    const size_t n = job->n;
    const bool share_merged = share && share->Wb == 1 && share->W > 1;       // entries are table indices w * n_tab + tab_off + i
    if (share && share->n == n && share->scalars == job->scalars && share->sorted && share->Wb == job->Wb &&
        share->n_tab == job->n_tab && (share->tab_off == job->tab_off || share_merged) && share->c == job->c) {
        // same scalar vector as an earlier job (A, B-in-G1 and B-in-G2 all use z[1..]): reuse its sort.  A different offset into a
        // table of window multiples (a polynomial's commitment over the shifted powers: marlin_pc/mod.rs:172-243) folds into
        // the base pointer -- the entries index the table linearly
        if (share->tab_off != job->tab_off)
            job->bases_dev = job->bases_dev + ((ptrdiff_t)job->tab_off - (ptrdiff_t)share->tab_off) * (ptrdiff_t)job->stride;
        job->sorted = share->sorted; job->desc = share->desc; job->order = share->order; job->ctr = share->ctr; job->heavy = share->heavy;
        job->heavy2 = share->heavy2;
        ZK_HIP(ctx, hipEventCreateWithFlags(&job->sort_done, hipEventDisableTiming));
        if (share->sort_stream != st) ZK_HIP(ctx, hipStreamWaitEvent(st, share->sort_done, 0));
        ZK_HIP(ctx, hipEventRecord(job->sort_done, st));
        job->sort_stream = st;
        return ZK_OK;
    }
    MsmBufs<F> b;
    ZK_TRY(msm_bufs_t<F>(ctx, job, b, true));
    const uint32_t W = job->W, NB = job->NB, seg = job->seg, Wb = job->Wb;
    const int merged = Wb == 1 && W > 1;
    const size_t nbuck = (size_t)Wb * NB;
    const MsmPlan p = make_plan(n, merged ? job->c : 0);
    Bias bias;
    WinOff wo;
    for (int i = 0; i < 65; i++) wo.off[i] = job->off[i];
    wo.off[65] = 0;
    for (int i = 0; i < 9; i++) bias.w[i] = p.bias[i];
    uint32_t* win_segs = b.small;
    uint32_t* ctr = b.small + 64;
    uint32_t* hist = b.small + 72;
    uint32_t* bin_start = hist + (seg + 1);
    uint32_t* bin_cursor = bin_start + (seg + 1);
    ZkPhaseTimer* tm = new ZkPhaseTimer(ctx, st);
    job->timers.push_back(tm);
    tm->begin(g1 ? "msm_g1.sort" : "msm_g2.sort");
    const bool one_block = merged && (size_t)W * n <= SS_MAX_ENTRIES && NB <= SS_MAX_NB && seg <= SS_MAX_SEG;      // k_sort_small: writes all of ctr itself
    const uint32_t grp_base = (uint32_t)(nbuck + job->max_heavy_segs);       // where the group sums of two-level buckets live in `sums`
    auto scans = [&](uint32_t Wx, uint32_t NBx) -> int {      // counting-sort path only: one block per window
        if (NBx > 65536) ZK_FAIL(ctx, ZK_ERR_ARG, "msm: a bucket set of more than 2^16 buckets needs the bucket sort (>= 2^16 digits)");
        hipLaunchKernelGGL(k_scan, Wx, 1024, 0, st, b.counts, b.offs, b.seg_local, win_segs, NBx, (const uint32_t*)(ctr + 3));
        return ZK_OK;
    };
    // The bucket sort (msm_sort.hip: no global atomics per digit, zero digits dropped, the segment length fitted to the input) for
    // every MSM with at least 2^16 digits; it sees ONE set of Wb*NB buckets (bucket ids w*NB + b, as the reduce phase numbers
    // them).  Small MSMs keep the counting sort with per-window offsets: a handful of short kernels.
    ZkGroupArgs ga;
    ga.scalars = job->scalars; ga.n = n; ga.wo = wo; ga.bias = bias; ga.W = W; ga.NB = NB; ga.merged = merged != 0;
    ga.n_tab = job->n_tab; ga.tab_off = job->tab_off; ga.NBt = (uint32_t)nbuck;
    ga.lanes = (uint32_t)ctx->n_cu * 4 * 64 * 2; ga.seg_max = seg;
    ga.sorted = b.sorted; ga.offs = b.offs; ga.ctr = ctr;
    // ... and for every bucket set the counting sort's one-block-per-window scan cannot take (a short MSM over a table of
    // window multiples with c = 20: 2^19 buckets whatever n is -- the bucket sort is correct for any total)
    if (one_block) {
        // a small MSM over window multiples: the whole sort in one block (k_sort_small)
        const size_t lds = ((size_t)NB + 2 * SS_NT + (SS_MAX_SEG + 1) + 8) * 4;
        if (!ctx->flags["sort_small_lds"]) {
            ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_sort_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)SS_MAX_NB + 2 * SS_NT + (SS_MAX_SEG + 1) + 8) * 4)));
            ctx->flags["sort_small_lds"] = 1;
        }
        hipLaunchKernelGGL(k_sort_small, 1, SS_NT, lds, st, SortSmallArgs{job->scalars, (uint32_t)n, wo, W, NB, bias, seg, job->n_tab, job->tab_off, grp_base,
                           b.dig, b.sorted, b.desc, b.heavy, b.heavy2, b.order, ctr});
        ZK_HIP(ctx, hipGetLastError());
        tm->end();
        job->sorted = b.sorted; job->desc = b.desc; job->order = b.order; job->ctr = ctr; job->heavy = b.heavy; job->heavy2 = b.heavy2;
        ZK_HIP(ctx, hipEventCreateWithFlags(&job->sort_done, hipEventDisableTiming));
        ZK_HIP(ctx, hipEventRecord(job->sort_done, st));
        job->sort_stream = st;
        return ZK_OK;
    }
    // The ~13 launches of the sort as ONE graph launch from the third sort with the same arguments on (core.hip: zk_graph_run): a
    // prover that works through proofs of one shape sorts the same buffers with the same geometry every time, and for jobs of
    // 2^12 .. 2^16 scalars the host's launch rate, not the device, was the length of a sort (4 sorts of a Marlin round at 2^14:
    // 0.5 ms one behind the other on three streams).
    std::string gkey("sort");
    {
        auto put = [&](const void* p, size_t nbytes) { gkey.append((const char*)p, nbytes); };
        const uint64_t words[] = {ctx->scratch_gen, (uint64_t)(uintptr_t)job->scalars, (uint64_t)n, W, NB, Wb, job->c, seg, (uint64_t)job->slot,
                                  (uint64_t)merged, job->n_tab, job->tab_off, (uint64_t)job->max_segs, (uint64_t)job->max_heavy_segs, (uint64_t)g1};
        put(words, sizeof words);
        put(job->off, sizeof job->off);
    }
    ZK_TRY(zk_graph_run(ctx, gkey, st, [&]() -> int {
        uint32_t flat_buckets = 0;
        if (!one_block) ZK_HIP(ctx, hipMemsetAsync(b.small, 0, (64 + 8 + 3 * (size_t)(seg + 1)) * 4, st));
        if (((size_t)W * n >= 65536 || NB > 65536) && zk_msm_group_supported(ga)) {
            const uint32_t NBt = (uint32_t)nbuck;
            ZK_TRY(zk_msm_group(ctx, st, job->slot, ga));
            hipLaunchKernelGGL(k_build_segs, zk_grid(nbuck, 256, 512), 256, (seg + 1) * 4, st, b.offs, (const uint32_t*)nullptr, win_segs,
                               (size_t)0, 1u, NBt, b.desc, b.heavy, b.heavy2, ctr, hist, grp_base);
            flat_buckets = NBt;
        } else {
            ZK_HIP(ctx, hipMemsetAsync(b.counts, 0, nbuck * 4, st));
            ZK_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)(ctr + 3), (int)seg, 1, st));             // the host's plan is the segment length
            hipLaunchKernelGGL(k_digits, zk_grid(n, 256), 256, 0, st, job->scalars, n, wo, W, NB, bias, b.dig, b.counts, merged);
            ZK_TRY(scans(Wb, NB));
            hipLaunchKernelGGL(k_scatter, zk_grid((size_t)W * n, 256), 256, 0, st, b.dig, n, W, NB, b.offs, b.counts, b.sorted, merged,
                               job->n_tab, job->tab_off);
            hipLaunchKernelGGL(k_build_segs, zk_grid(nbuck, 256, 512), 256, (seg + 1) * 4, st, b.offs, b.seg_local, win_segs,
                               merged ? (size_t)0 : n, Wb, NB, b.desc, b.heavy, b.heavy2, ctr, hist, grp_base);
        }
        hipLaunchKernelGGL(k_len_scan, 1, 512, 0, st, hist, bin_start, bin_cursor, ctr, flat_buckets);
        hipLaunchKernelGGL(k_order, 512, 256, (seg + 1) * 4, st, b.desc, ctr, bin_cursor, b.order);
        ZK_HIP(ctx, hipGetLastError());
        return ZK_OK;
    }));
    tm->end();
    job->sorted = b.sorted; job->desc = b.desc; job->order = b.order; job->ctr = ctr; job->heavy = b.heavy; job->heavy2 = b.heavy2;
    ZK_HIP(ctx, hipEventCreateWithFlags(&job->sort_done, hipEventDisableTiming));
    ZK_HIP(ctx, hipEventRecord(job->sort_done, st));
    job->sort_stream = st;
    return ZK_OK;
}

// A job whose accumulate kernel cannot give every SIMD a wave (<= 2^17 digits in segments of 8: <= 2^14 segments): its chains of
// dependent additions run on twice the lanes per point (ec_dual.cuh)
static bool msm_latency_bound(const ZkMsmJob* job) { return (size_t)job->n * job->W <= ((size_t)1 << 17); }

template <class F>
int msm_enqueue_accum_t(zk_ctx* ctx, ZkMsmJob* job, hipStream_t st) {
    if (job->n == 0) return ZK_OK;
    const bool g1 = F::WORDS == 12;
    MsmBufs<F> b;
    ZK_TRY(msm_bufs_t<F>(ctx, job, b, false));
    if (job->sort_stream != st) ZK_HIP(ctx, hipStreamWaitEvent(st, job->sort_done, 0));
    ZkPhaseTimer* tm = new ZkPhaseTimer(ctx, st);
    job->timers.push_back(tm);
    tm->begin(g1 ? "msm_g1.accum" : "msm_g2.accum");
    const unsigned accum_blocks = (unsigned)((job->max_segs + 255) / 256);      // one lane per segment (blocks beyond ctr[2] return at once)
    // G2 runs on lane pairs (msm_g2pair.hip): the one-lane-per-addition form needs the whole register file and is slower
    if constexpr (F::WORDS == 12) {
        if (job->stride == 64)      // limb-form table with both signs (fixed_base.hip::k_repack_limbs)
            hipLaunchKernelGGL((k_accum<F, true>), accum_blocks, 256, 0, st, job->bases_dev, job->sorted,
                               (const SegDesc*)job->desc, job->order, job->ctr, b.sums, job->stride);
        else
            hipLaunchKernelGGL((k_accum<F, false>), accum_blocks, 256, 0, st, job->bases_dev, job->sorted,
                               (const SegDesc*)job->desc, job->order, job->ctr, b.sums, job->stride);
    }
    else
        zk_launch_accum_g2pair(st, job->max_segs, job->bases_dev, job->sorted, job->desc, job->order, job->ctr, b.sums, msm_latency_bound(job));
    ZK_HIP(ctx, hipGetLastError());
    tm->end();
    ZK_HIP(ctx, hipEventCreateWithFlags(&job->accum_done, hipEventDisableTiming));
    ZK_HIP(ctx, hipEventRecord(job->accum_done, st));
    job->accum_stream = st;
    return ZK_OK;
}

template <class F>
int msm_enqueue_reduce_t(zk_ctx* ctx, ZkMsmJob* job, hipStream_t st) {
    constexpr size_t XW = 4 * F::WORDS;
    job->stream = st;
    if (job->n == 0) return ZK_OK;
    const bool g1 = F::WORDS == 12;
    MsmBufs<F> b;
    ZK_TRY(msm_bufs_t<F>(ctx, job, b, false));
    if (job->accum_stream != st) ZK_HIP(ctx, hipStreamWaitEvent(st, job->accum_done, 0));
    ZkPhaseTimer* tm = new ZkPhaseTimer(ctx, st);
    job->timers.push_back(tm);
    tm->begin(g1 ? "msm_g1.reduce" : "msm_g2.reduce");
    // (a small grid: every block of the chain waits for a wave slot beside the running accumulate kernel, and a uniform input has
    // nothing to fold at all; the blocks stride over the heavy list)
    const unsigned light_blocks = (unsigned)std::min<size_t>((job->max_heavy + 63) / 64, 128);
    const unsigned heavy_blocks = (unsigned)std::min<size_t>(job->max_heavy, 256);
    if constexpr (F::WORDS != 12) {
        // G2: the same chain on lane pairs (msm_g2pair.hip)
        ZkG2PairReduce a{job->heavy, job->heavy2, job->ctr, b.fold_done, b.sums, b.rowP, b.colP, b.bits, job->log_nb, job->Wb, 2 * light_blocks, heavy_blocks};
        a.quads = msm_latency_bound(job);
        ZK_TRY(zk_launch_reduce_g2pair(ctx, st, a));
    } else {
        hipLaunchKernelGGL(k_fold<F>, light_blocks + heavy_blocks, 64, 64 * XW * 4, st, (const HeavyDesc*)job->heavy, (const HeavyDesc*)job->heavy2,
                           (const uint32_t*)job->ctr, b.fold_done, b.sums, light_blocks);
        const GridSrc src{{(const uint32_t*)b.sums, nullptr, nullptr, nullptr}, 0u};
        if (((size_t)job->Wb << job->log_nb) <= RED_DUAL_MAX_BUCKETS) {
            const GridGeom gg = make_grid_geom(job->log_nb, job->Wb, RedG1Dual::PTS);
            hipLaunchKernelGGL(k_grid_l1<RedG1Dual>, gg.row_blocks + gg.col_blocks, RedG1Dual::NT, RedG1Dual::PTS * XW * 4, st, src, b.rowP, b.colP, gg);
            hipLaunchKernelGGL(k_grid_bits<RedG1Dual>, gg.n_win * job->nout, RedG1Dual::NT, RedG1Dual::PTS * XW * 4, st, (const uint32_t*)b.rowP,
                               (const uint32_t*)b.colP, b.bits, gg);
        } else {
            const GridGeom gg = make_grid_geom(job->log_nb, job->Wb, RedG1::PTS);
            hipLaunchKernelGGL(k_grid_l1<RedG1>, gg.row_blocks + gg.col_blocks, RedG1::NT, RedG1::PTS * XW * 4, st, src, b.rowP, b.colP, gg);
            hipLaunchKernelGGL(k_grid_bits<RedG1>, gg.n_win * job->nout, RedG1::NT, RedG1::PTS * XW * 4, st, (const uint32_t*)b.rowP,
                               (const uint32_t*)b.colP, b.bits, gg);                                                  // 48 KiB of LDS each
        }
    }
    ZK_HIP(ctx, hipGetLastError());
    tm->end();
    // pinned destination: a pageable one would make the "async" copy block the host until this job is done
    auto& pin = ctx->pinned[job->pin_key >= 0 ? job->pin_key : job->slot];
    const size_t bytes = std::max<size_t>((size_t)64 * 17, (size_t)job->Wb * job->nout) * XW * 4;
    if (pin.bytes < bytes) {
        if (pin.p) (void)hipHostFree(pin.p);
        ZK_HIP(ctx, hipHostMalloc(&pin.p, bytes, hipHostMallocDefault));
        pin.bytes = bytes;
    }
    job->hw = (uint32_t*)pin.p;
    ZK_HIP(ctx, hipMemcpyAsync(job->hw, b.bits, (size_t)job->Wb * job->nout * XW * 4, hipMemcpyDeviceToHost, st));
    // finish() waits for this event, not for the stream: later jobs' reduce phases may be queued behind on the same stream
    ZK_HIP(ctx, hipEventCreateWithFlags(&job->reduce_done, hipEventDisableTiming));
    ZK_HIP(ctx, hipEventRecord(job->reduce_done, st));
    return ZK_OK;
}

// ---- a group of small G1 jobs: one sort launch (a block per job), one accumulate launch, one launch per level of the reduce chain --
// The sorts of a group: every job must qualify for the one-block sort (zk_msm_sort_group_ok; else the per-job calls).
// A block's time grows with its digits (~1.3 us per thousand), the multi-launch path is ~105 us flat plus ten launches of host
// time, and since round 6 every job of a group that takes it has a stream of its own (msm_batch.hip).  Marlin, same box, limit
// 2^16 / 3 * 2^15 .. 2^17 / 2^18: |H| = 2^11 (mask: 123 k digits) 3.84 / 4.12-4.19 / -, 2^12 (w, z_a, z_b: 78 k each; mask 233 k)
// 4.80 / 4.32-4.38 / 5.4, 2^13 5.35 / 5.2-5.4 / 5.3-5.4: one block up to 96 k digits.
constexpr uint32_t SS_GROUP_MAX_ENTRIES = 3u << 15;
static bool sort_small_fits(const ZkMsmJob* j) {
    const bool merged = j->Wb == 1 && j->W > 1;
    return j->n > 0 && merged && (size_t)j->W * j->n <= SS_GROUP_MAX_ENTRIES && j->NB <= SS_MAX_NB && j->seg <= SS_MAX_SEG;
}
static bool msm_sort_group_ok(ZkMsmJob* const* jobs, int count) {
    if (count < 1 || count > MSM_GROUP_MAX) return false;
    for (int k = 0; k < count; k++)
        if (jobs[k]->group != 1 || !sort_small_fits(jobs[k])) return false;
    return true;
}
static int msm_enqueue_sort_group(zk_ctx* ctx, ZkMsmJob* const* jobs, int count, hipStream_t st) {
    using F = G1Field;
    SortSmallGroup g{};
    uint32_t nb_max = 0;
    for (int k = 0; k < count; k++) {
        ZkMsmJob* job = jobs[k];
        MsmBufs<F> b;
        ZK_TRY(msm_bufs_t<F>(ctx, job, b, true));
        const MsmPlan p = make_plan(job->n, job->c);
        SortSmallArgs& a = g.j[k];
        a.scalars = job->scalars; a.n = (uint32_t)job->n; a.W = job->W; a.NB = job->NB; a.seg = job->seg;
        for (int i = 0; i < 65; i++) a.wo.off[i] = job->off[i];
        a.wo.off[65] = 0;
        for (int i = 0; i < 9; i++) a.bias.w[i] = p.bias[i];
        a.n_tab = job->n_tab; a.tab_off = job->tab_off;
        a.grp_base = (uint32_t)((size_t)job->Wb * job->NB + job->max_heavy_segs);
        a.dig = b.dig; a.sorted = b.sorted; a.desc = b.desc; a.heavy = b.heavy; a.heavy2 = b.heavy2; a.order = b.order; a.ctr = b.small + 64;
        job->sorted = b.sorted; job->desc = b.desc; job->order = b.order; job->ctr = b.small + 64; job->heavy = b.heavy; job->heavy2 = b.heavy2;
        nb_max = std::max(nb_max, job->NB);
    }
    ZkPhaseTimer* tm = new ZkPhaseTimer(ctx, st);
    jobs[0]->timers.push_back(tm);
    tm->begin("msm_g1.sort");
    if (!ctx->flags["sort_small_group_lds"]) {
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_sort_small_group, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)SS_MAX_NB + 2 * SS_NT + (SS_MAX_SEG + 1) + 8) * 4)));
        ctx->flags["sort_small_group_lds"] = 1;
    }
    const size_t lds = ((size_t)nb_max + 2 * SS_NT + (SS_MAX_SEG + 1) + 8) * 4;
    hipLaunchKernelGGL(k_sort_small_group, count, SS_NT, lds, st, g);
    ZK_HIP(ctx, hipGetLastError());
    tm->end();
    for (int k = 0; k < count; k++) {
        ZK_HIP(ctx, hipEventCreateWithFlags(&jobs[k]->sort_done, hipEventDisableTiming));
        ZK_HIP(ctx, hipEventRecord(jobs[k]->sort_done, st));
        jobs[k]->sort_stream = st;
    }
    return ZK_OK;
}

// Conditions (zk_msm_group_ok): G1, 2 .. MSM_GROUP_MAX jobs, every one sorted already, over tables of window multiples in the same
// layout (one bucket set each, the same number of buckets).  The jobs keep their own sort products and bucket sums; row / column
// partials and bit sums of the group live in the FIRST job's slot (sized for the group), and one copy brings all of them back.
static bool msm_group_ok(ZkMsmJob* const* jobs, int count) {
    if (count < 2 || count > MSM_GROUP_MAX) return false;
    for (int k = 0; k < count; k++) {
        const ZkMsmJob* j = jobs[k];
        if (j->group != 1 || j->n == 0 || j->Wb != 1 || j->log_nb != jobs[0]->log_nb || (j->stride == 64) != (jobs[0]->stride == 64) || !j->sort_done) return false;
        // (a job that fills the chip by itself gains nothing from company: Marlin rounds, same box, limit 2^21 / 2^23 / 2^24 digits --
        // |H| = 2^16 11.4 / 10.6 / -, 2^17 16.2 / 14.8 / -, 2^18 22.0 / 21.7 / 21.0, 2^19 34.8 / 35.0 / 34.9, 2^20 62.9 / - / 64.1 ms)
        if ((size_t)j->n * j->W > ((size_t)1 << 23)) return false;
    }
    return true;
}

static int msm_enqueue_accum_group(zk_ctx* ctx, ZkMsmJob* const* jobs, int count, hipStream_t st) {
    using F = G1Field;
    AccumGroup g{};
    size_t max_segs = 0;
    for (int k = 0; k < count; k++) {
        ZkMsmJob* job = jobs[k];
        MsmBufs<F> b;
        ZK_TRY(msm_bufs_t<F>(ctx, job, b, false));
        if (job->sort_stream != st) ZK_HIP(ctx, hipStreamWaitEvent(st, job->sort_done, 0));
        g.j[k] = AccumArgs{job->bases_dev, job->sorted, (const SegDesc*)job->desc, job->order, job->ctr, b.sums, job->stride};
        max_segs = std::max(max_segs, job->max_segs);
    }
    for (int k = count; k < MSM_GROUP_MAX; k++) g.j[k] = g.j[0];
    ZkPhaseTimer* tm = new ZkPhaseTimer(ctx, st);
    jobs[0]->timers.push_back(tm);
    tm->begin("msm_g1.accum");
    // a group that cannot give every SIMD a wave (<= 2^18 digits = 2^15 segments of 8 in all): a lane pair per segment (ec_dual.cuh)
    size_t digits = 0;
    for (int k = 0; k < count; k++) digits += (size_t)jobs[k]->n * jobs[k]->W;
    const bool dual = digits <= ((size_t)1 << 18);
    const dim3 grid((unsigned)((max_segs + (dual ? 127 : 255)) / (dual ? 128 : 256)), (unsigned)count);
    if (jobs[0]->stride == 64) {
        if (dual) hipLaunchKernelGGL((k_accum_group<F, true, true>), grid, 256, 0, st, g);
        else hipLaunchKernelGGL((k_accum_group<F, true>), grid, 256, 0, st, g);
    } else {
        if (dual) hipLaunchKernelGGL((k_accum_group<F, false, true>), grid, 256, 0, st, g);
        else hipLaunchKernelGGL((k_accum_group<F, false>), grid, 256, 0, st, g);
    }
    ZK_HIP(ctx, hipGetLastError());
    tm->end();
    for (int k = 0; k < count; k++) {
        ZK_HIP(ctx, hipEventCreateWithFlags(&jobs[k]->accum_done, hipEventDisableTiming));
        ZK_HIP(ctx, hipEventRecord(jobs[k]->accum_done, st));
        jobs[k]->accum_stream = st;
    }
    return ZK_OK;
}

static int msm_enqueue_reduce_group(zk_ctx* ctx, ZkMsmJob* const* jobs, int count, hipStream_t st) {
    using F = G1Field;
    constexpr size_t XW = 4 * F::WORDS;
    FoldGroup fg{};
    GridSrc src{};
    src.per_log = jobs[0]->log_nb;
    size_t max_heavy = 0;
    for (int k = 0; k < count; k++) {
        ZkMsmJob* job = jobs[k];
        job->stream = st;
        MsmBufs<F> b;
        ZK_TRY(msm_bufs_t<F>(ctx, job, b, false));
        if (job->accum_stream != st) ZK_HIP(ctx, hipStreamWaitEvent(st, job->accum_done, 0));
        fg.j[k] = FoldArgs{(const HeavyDesc*)job->heavy, (const HeavyDesc*)job->heavy2, (const uint32_t*)job->ctr, b.fold_done, b.sums};
        src.p[k] = b.sums;
        max_heavy = std::max(max_heavy, job->max_heavy);
    }
    for (int k = count; k < MSM_GROUP_MAX; k++) { fg.j[k] = fg.j[0]; src.p[k] = src.p[0]; }
    const bool dual = ((size_t)count << jobs[0]->log_nb) <= RED_DUAL_MAX_BUCKETS;
    const GridGeom gg = make_grid_geom(jobs[0]->log_nb, (uint32_t)count, dual ? RedG1Dual::PTS : RedG1::PTS);
    const uint32_t nout = jobs[0]->nout;
    char nm[64];
    uint32_t *rowP, *colP, *bits;
    snprintf(nm, sizeof nm, "msm_grp_rowP.%d", jobs[0]->slot);
    ZK_TRY(zk_scratch(ctx, nm, grid_row_points(gg) * XW * 4, (void**)&rowP));
    snprintf(nm, sizeof nm, "msm_grp_colP.%d", jobs[0]->slot);
    ZK_TRY(zk_scratch(ctx, nm, grid_col_points(gg) * XW * 4, (void**)&colP));
    snprintf(nm, sizeof nm, "msm_grp_bits.%d", jobs[0]->slot);
    ZK_TRY(zk_scratch(ctx, nm, (size_t)count * nout * XW * 4, (void**)&bits));
    ZkPhaseTimer* tm = new ZkPhaseTimer(ctx, st);
    jobs[0]->timers.push_back(tm);
    tm->begin("msm_g1.reduce");
    const unsigned light_blocks = (unsigned)std::min<size_t>((max_heavy + 63) / 64, 128);
    const unsigned heavy_blocks = (unsigned)std::min<size_t>(max_heavy, 256);
    if (dual) hipLaunchKernelGGL((k_fold_group<F, true>), dim3(2 * light_blocks + heavy_blocks, (unsigned)count), 64, 64 * XW * 4, st, fg, 2 * light_blocks);
    else hipLaunchKernelGGL(k_fold_group<F>, dim3(light_blocks + heavy_blocks, (unsigned)count), 64, 64 * XW * 4, st, fg, light_blocks);
    if (dual) {
        hipLaunchKernelGGL(k_grid_l1<RedG1Dual>, gg.row_blocks + gg.col_blocks, RedG1Dual::NT, RedG1Dual::PTS * XW * 4, st, src, rowP, colP, gg);
        hipLaunchKernelGGL(k_grid_bits<RedG1Dual>, gg.n_win * nout, RedG1Dual::NT, RedG1Dual::PTS * XW * 4, st, (const uint32_t*)rowP, (const uint32_t*)colP, bits, gg);
    } else {
        hipLaunchKernelGGL(k_grid_l1<RedG1>, gg.row_blocks + gg.col_blocks, RedG1::NT, RedG1::PTS * XW * 4, st, src, rowP, colP, gg);
        hipLaunchKernelGGL(k_grid_bits<RedG1>, gg.n_win * nout, RedG1::NT, RedG1::PTS * XW * 4, st, (const uint32_t*)rowP, (const uint32_t*)colP, bits, gg);
    }
    ZK_HIP(ctx, hipGetLastError());
    tm->end();
    auto& pin = ctx->pinned[jobs[0]->pin_key >= 0 ? jobs[0]->pin_key : 32 + jobs[0]->slot];
    const size_t bytes = (size_t)MSM_GROUP_MAX * 32 * XW * 4;
    if (pin.bytes < bytes) {
        if (pin.p) (void)hipHostFree(pin.p);
        pin.p = nullptr; pin.bytes = 0;
        ZK_HIP(ctx, hipHostMalloc(&pin.p, bytes, hipHostMallocDefault));
        pin.bytes = bytes;
    }
    ZK_HIP(ctx, hipMemcpyAsync(pin.p, bits, (size_t)count * nout * XW * 4, hipMemcpyDeviceToHost, st));
    for (int k = 0; k < count; k++) {
        jobs[k]->hw = (uint32_t*)pin.p + (size_t)k * nout * XW;
        ZK_HIP(ctx, hipEventCreateWithFlags(&jobs[k]->reduce_done, hipEventDisableTiming));
        ZK_HIP(ctx, hipEventRecord(jobs[k]->reduce_done, st));
    }
    return ZK_OK;
}

template <class F>
int msm_finish_t(zk_ctx* ctx, ZkMsmJob* job, void* out_host) {
    constexpr size_t XW = 4 * F::WORDS;
    if (job->n == 0) {
        host_write_projective<F>(aff_inf<F>(), (uint64_t*)out_host);
        return ZK_OK;
    }
    if (job->reduce_done) ZK_HIP(ctx, hipEventSynchronize(job->reduce_done));
    else ZK_HIP(ctx, hipStreamSynchronize(job->stream));
    for (auto* t : job->timers) t->resolve();
    // window sum = 2^cl * sum_j 2^j rowbit_j + sum_j 2^j colbit_j + (plain sum): ONE Horner chain over the row bits followed by
    // the column bits (msm_reduce.cuh), then Horner over the windows, most significant first (variable_base.rs:94-105).  All
    // in the 64-bit host field.
    using H = typename Host64Of<F>::type;
    const uint32_t nout = job->nout;
    auto window_sum = [&](uint32_t w) {
        const uint32_t* base = job->hw + (size_t)w * nout * XW;
        XYZZ<H> ws = xyzz_inf<H>();
        const uint32_t cl = job->log_nb / 2, rl = job->log_nb - cl;
        for (int j = (int)rl - 1; j >= 0; j--) {
            ws = xyzz_dbl<H>(ws);
            ws = xyzz_add<H>(ws, xyzz_to_host64<F>(xyzz_load<F>(base + (size_t)j * XW)));
        }
        for (int j = (int)cl - 1; j >= 0; j--) {
            ws = xyzz_dbl<H>(ws);
            ws = xyzz_add<H>(ws, xyzz_to_host64<F>(xyzz_load<F>(base + (size_t)(rl + j) * XW)));
        }
        return xyzz_add<H>(ws, xyzz_to_host64<F>(xyzz_load<F>(base + (size_t)(nout - 1) * XW)));
    };
    XYZZ<H> total = xyzz_inf<H>();
    // the per-window Horner chains (~15 doublings + 16 additions each for 16 windows) are independent: four host threads take
    // them, because the last job's finish sits on the proof's critical path.  A merged bucket set is one window: one chain.
    const uint32_t nwin = job->Wb;
    std::vector<XYZZ<H>> wsum(nwin);
    {
        // (a window is log2(NB) + 1 points: ~16 doublings and additions of ~1 us each; 32 windows of a small MSM on eight threads)
        const uint32_t nthreads = nwin >= 16 ? 8u : nwin >= 8 ? 4u : 1u;
        std::vector<ZkTask<void>> tasks;                 // joining handles, declared after wsum: a throw below cannot free it under a task
        for (uint32_t t = 1; t < nthreads; t++)
            tasks.push_back(zk_async(ctx, [&, t] { for (uint32_t w = t; w < nwin; w += nthreads) wsum[w] = window_sum(w); }));
        for (uint32_t w = 0; w < nwin; w += nthreads) wsum[w] = window_sum(w);
        for (auto& f : tasks) f.get();
    }
    for (int w = (int)job->Wb - 1; w >= 0; w--) {
        const uint32_t cw = job->Wb == 1 ? 0u : (uint32_t)(job->off[w + 1] - job->off[w]);   // 2^cw * (sum of the higher windows) + this window
        for (uint32_t k = 0; k < cw; k++) total = xyzz_dbl<H>(total);
        total = xyzz_add<H>(total, wsum[(uint32_t)w]);
    }
    host64_write_projective<H>(xyzz_to_affine<H>(total), (uint64_t*)out_host);
    return ZK_OK;
}

template <class F>
int msm_run_t(zk_ctx* ctx, const zk_bases* bases, size_t base_offset, const void* scalars, size_t n, void* out_host) {
    ZkMsmJob job;
    int rc = msm_prepare_t<F>(ctx, &job, bases, base_offset, scalars, n, 0);
    if (rc == ZK_OK) rc = msm_enqueue_sort_t<F>(ctx, &job, ctx->stream, nullptr);
    if (rc == ZK_OK) rc = msm_enqueue_accum_t<F>(ctx, &job, ctx->stream);
    if (rc == ZK_OK) rc = msm_enqueue_reduce_t<F>(ctx, &job, ctx->stream);
    if (rc == ZK_OK) rc = msm_finish_t<F>(ctx, &job, out_host);
    return rc;
}

// A host table -> a resident one.  layout == NULL: the ABI's packed form (zk_g1_affine / zk_g2_affine, all-zero = infinity);
// otherwise the caller's own struct layout (GroupAffine<P> {x, y, infinity} as rustc lays it out): the points are gathered into the
// packed form in page-locked memory by the context's helper threads and copied from there.
template <class F>
int bases_upload_t(zk_ctx* ctx, const void* host, size_t n, int group, const ZkAffineLayout* layout, zk_bases** out) {
    if (!ctx || !out || (n && !host)) return ZK_ERR_ARG;
    constexpr size_t FE = F::WORDS * 4;                       // bytes per coordinate (48 / 96)
    const size_t bytes = n * 2 * FE;
    std::unique_ptr<zk_bases> b(new zk_bases());             // (freed on every failing exit: ADVICE r4)
    b->group = group;
    b->n = n;
    if (n) {
        void* stage;
        ZK_TRY(zk_scratch(ctx, "bases_stage", bytes, &stage));
        const void* src = host;
        if (layout) {
            if (layout->stride < 2 * FE || layout->off_x + FE > layout->stride || layout->off_y + FE > layout->stride ||
                (layout->off_inf != SIZE_MAX && layout->off_inf >= layout->stride))
                ZK_FAIL(ctx, ZK_ERR_ARG, "strided base table: the field offsets do not fit the stride");
            auto& pin = ctx->pinned[-3];
            if (pin.bytes < bytes) {
                if (pin.p) (void)hipHostFree(pin.p);
                pin.p = nullptr; pin.bytes = 0;
                ZK_HIP(ctx, hipHostMalloc(&pin.p, bytes, hipHostMallocDefault));
                pin.bytes = bytes;
            }
            char* dst = (char*)pin.p;
            const ZkAffineLayout lay = *layout;
            auto pack = [=](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; i++) {
                    const char* p = (const char*)host + i * lay.stride;
                    char* d = dst + i * 2 * FE;
                    if (lay.off_inf != SIZE_MAX && p[lay.off_inf]) { memset(d, 0, 2 * FE); continue; }
                    memcpy(d, p + lay.off_x, FE);
                    memcpy(d + FE, p + lay.off_y, FE);
                }
            };
            const size_t T = n >= 65536 ? 4 : 1, per = (n + T - 1) / T;
            {
                std::vector<ZkTask<void>> tasks;
                for (size_t t = 1; t < T; t++) tasks.push_back(zk_async(ctx, [=] { pack(std::min(n, t * per), std::min(n, (t + 1) * per)); }));
                pack(0, std::min(n, per));
            }
            src = dst;
        }
        uint32_t* dev = nullptr;
        if (hipMalloc((void**)&dev, bytes) != hipSuccess) { (void)hipGetLastError(); ZK_FAIL(ctx, ZK_ERR_NOMEM, "bases upload: hipMalloc failed"); }
        b->dev = dev;                                        // owned by *b from here: ~unique_ptr does not free device memory, so:
        auto fail = [&](hipError_t e) { (void)hipFree(dev); b->dev = nullptr; return e; };
        // (the packed table is the caller's memory -- possibly a temporary copy made for this call: through the ring; the gathered
        // one already sits in page-locked memory)
        hipError_t e = hipSuccess;
        if (layout) e = hipMemcpyAsync(stage, src, bytes, hipMemcpyHostToDevice, ctx->stream);
        else if (zk_xfer_h2d(ctx, stage, src, bytes) != ZK_OK) { (void)hipFree(dev); b->dev = nullptr; return ZK_ERR_HIP; }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_bases_import<F>, zk_grid(n, 256), 256, 0, ctx->stream, (const uint32_t*)stage, b->dev, n);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) ZK_HIP(ctx, fail(e));
    }
    *out = b.release();
    return ZK_OK;
}

template <class F>
int bases_download_t(zk_ctx* ctx, const zk_bases* b, size_t offset, size_t n, void* out) {
    if (!ctx || !b || (n && !out)) return ZK_ERR_ARG;
    if (offset + n > b->n) ZK_FAIL(ctx, ZK_ERR_ARG, "bases download: range out of bounds");
    if (!n) return ZK_OK;
    const size_t bytes = n * 2 * F::WORDS * 4;
    void* stage;
    ZK_TRY(zk_scratch(ctx, "bases_stage", bytes, &stage));
    hipLaunchKernelGGL(k_bases_export<F>, zk_grid(n, 256), 256, 0, ctx->stream, b->dev + offset * 2 * F::WORDS, (uint32_t*)stage, n);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(out, stage, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

// ---- MSMs started AHEAD for the tables a caller asks for next with the same scalars -----------------------------------------------
// The unchanged create_proof runs its MSMs one call behind the other, and three of them over ONE scalar vector: calculate_coeff with
// &pk.a_query, then &pk.b_g1_query, then &pk.b_g2_query over `assignment` (src/groth16.rs:137-160).  Every call is synchronous -- the
// trait returns the point -- so the upload, the sort, the reduce chain and the host's Horner half of one call never overlap the next
// call's kernels, as they do in the resident prover.  What CAN be known: which table followed which with the same scalars last time.
// So a context learns "after table T (scalars S) came table T' with the same S" (candidates by a fingerprint of 64 sampled scalars
// taken from host memory), and when T is asked for again it also starts the MSMs over T' and T'' -- on a private copy of the scalars,
// on the context's side stream, own scratch slots (6, 7) -- before it collects T's result.  The next call, if it names T' and
// its scalars are WORD FOR WORD the ones the speculative job ran on (compared on the device), takes that result; anything else
// drops the speculation (and after two wrong guesses for a table the pattern is forgotten).  Table hits are verified as always.
// Nothing speculative is ever returned unverified; ZK_MSM_SPEC=0 switches the whole thing off (A/B).
struct SpecJob { const zk_bases* table; ZkMsmJob job; };
struct ZkMsmSpec {
    const zk_bases* last_table = nullptr;
    uint64_t last_fp = 0;
    size_t last_n = 0;
    std::map<const zk_bases*, const zk_bases*> succ;
    std::map<const zk_bases*, int> bad;
    std::vector<std::unique_ptr<SpecJob>> jobs;          // in the order they will be asked for
    size_t n = 0;
    uint64_t fp = 0;
    uint32_t* flag_dev = nullptr;
    uint32_t* flag_host = nullptr;
    uint64_t started = 0, taken = 0, dropped = 0;
    bool off = false;                                     // zk_msm_speculate(ctx, 0)
    // the transform a caller ran last on a host vector (zk_msm_spec_fft_begin / _end): its size, the fingerprints its output has as
    // a scalar vector of N and of N - 1 elements (the min(len) rule against a query of D - 1 points), and -- learned -- which
    // table was asked for with exactly that output as scalars: `h = witness_map(..)` and then multi_scalar_mul(&pk.h_query, &h)
    // (src/groth16.rs:100-106)
    size_t fft_N = 0;
    int fft_kind = 0;                                     // inverse * 2 + coset
    uint64_t fft_fp[2] = {0, 0};
    struct AfterFft { const zk_bases* table; size_t n; };
    std::map<size_t, AfterFft> fft_succ;
    bool fp_pending = false;                              // a job started by fft_begin whose fingerprint fft_end still owes
    const void* late_dev = nullptr;                       // a LARGE job fft_begin left to fft_end
    const zk_bases* late_table = nullptr;
    size_t late_n = 0;
};
static bool spec_enabled() {
    static const bool on = !(getenv("ZK_MSM_SPEC") && atoi(getenv("ZK_MSM_SPEC")) == 0);
    return on;
}
static ZkMsmSpec* spec_of(zk_ctx* ctx) {
    if (!ctx->msm_spec) ctx->msm_spec = new ZkMsmSpec();
    return (ZkMsmSpec*)ctx->msm_spec;
}
__global__ void __launch_bounds__(256) k_scalars_differ(const uint4* a, const uint4* b, size_t n16, uint32_t* flag) {
    bool diff = false;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        diff = diff || x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w;
    }
    if (diff) *flag = 1;
}
// what is in flight is abandoned: its kernels read the scratch slots and the scalar copy, so they are waited for
static void spec_drop(zk_ctx* ctx, ZkMsmSpec* sp) {
    if (sp->jobs.empty()) return;
    if (!ctx->aux.empty()) (void)hipStreamSynchronize(ctx->aux[0]);
    if (ctx->acc_stream) (void)hipStreamSynchronize(ctx->acc_stream);
    sp->dropped += sp->jobs.size();
    sp->jobs.clear();
    sp->fp_pending = false;
}
// scalars (n elements on the device, final on the context stream) == the copy the speculative jobs run on?
static int spec_same_scalars(zk_ctx* ctx, ZkMsmSpec* sp, const void* scalars, size_t n, bool* same) {
    void* copy;
    ZK_TRY(zk_scratch(ctx, "spec_scalars", n * 32, &copy));
    if (!sp->flag_dev) {
        ZK_HIP(ctx, hipMalloc((void**)&sp->flag_dev, 16));
        ZK_HIP(ctx, hipHostMalloc((void**)&sp->flag_host, 16, hipHostMallocDefault));
    }
    ZK_HIP(ctx, hipMemsetAsync(sp->flag_dev, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_scalars_differ, zk_grid(n * 2, 256), 256, 0, ctx->stream, (const uint4*)scalars, (const uint4*)copy, n * 2, sp->flag_dev);
    ZK_HIP(ctx, hipGetLastError());
    ZK_HIP(ctx, hipMemcpyAsync(sp->flag_host, sp->flag_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *same = *sp->flag_host == 0;
    return ZK_OK;
}
// the tables that followed `b` last time, over a private copy of the scalars: every phase of every job on the context's ONE side
// stream, one job behind the other (groth16_pipeline.hip::zk_side_stream says why there is only one)
static int spec_start_tables(zk_ctx* ctx, ZkMsmSpec* sp, const zk_bases* const* next, int cnt, const void* scalars, size_t n, uint64_t fp);
static int spec_start(zk_ctx* ctx, ZkMsmSpec* sp, const zk_bases* b, const void* scalars, size_t n, uint64_t fp) {
    const zk_bases* next[2] = {nullptr, nullptr};
    int cnt = 0;
    auto it = sp->succ.find(b);
    if (it != sp->succ.end() && it->second != b && sp->bad[it->second] < 2 && n <= it->second->n) {
        next[cnt++] = it->second;
        auto it2 = sp->succ.find(it->second);
        if (it2 != sp->succ.end() && it2->second != b && it2->second != it->second && sp->bad[it2->second] < 2 && n <= it2->second->n) next[cnt++] = it2->second;
    }
    if (!cnt) return ZK_OK;
    return spec_start_tables(ctx, sp, next, cnt, scalars, n, fp);
}
static int spec_start_tables(zk_ctx* ctx, ZkMsmSpec* sp, const zk_bases* const* next, int cnt, const void* scalars, size_t n, uint64_t fp) {
    void* copy;
    ZK_TRY(zk_scratch(ctx, "spec_scalars", n * 32, &copy));
    hipStream_t s_sort;                                      // every phase of every job on the context's ONE side stream (zk_side_stream)
    ZK_TRY(zk_side_stream(ctx, &s_sort));
    ZK_HIP(ctx, hipMemcpyAsync(copy, scalars, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
    hipEvent_t e0;
    ZK_HIP(ctx, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    hipError_t e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(s_sort, e0, 0);
    (void)hipEventDestroy(e0);
    ZK_HIP(ctx, e);
    int rc = ZK_OK;
    for (int k = 0; k < cnt && rc == ZK_OK; k++) {
        std::unique_ptr<SpecJob> j(new SpecJob());
        j->table = next[k];
        j->job.pin_key = 40 + k;
        rc = zk_msm_prepare(ctx, &j->job, next[k], 0, copy, n, 6 + k);
        if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, &j->job, s_sort, nullptr);
        if (rc == ZK_OK) rc = zk_msm_enqueue_accum(ctx, &j->job, s_sort);
        if (rc == ZK_OK) rc = zk_msm_enqueue_reduce(ctx, &j->job, s_sort);
        sp->jobs.push_back(std::move(j));                    // (even a half-enqueued one: spec_drop waits for whatever it launched)
        if (rc == ZK_OK) sp->started++;
    }
    sp->n = n;
    sp->fp = fp;
    if (rc != ZK_OK) spec_drop(ctx, sp);
    return rc;
}

// AffineCurve::multi_scalar_mul on host slices.  The bases come through the context's table cache (bases_cache.hip): a table the
// context has seen before -- the queries of a proving key, the powers of an SRS -- is resident already (window multiples are built
// beside later calls), and only the scalars cross PCIe for the arithmetic; the caller's slice crosses once more, UNDER the MSM, to
// be compared in full with what the cached table was made from (a hit is a candidate until that comparison is through).
// n_lanes MSMs over the same table (the two lanes of a SPDZ multi_scale_pub_group: share/spdz.rs:482-488): scalars_dev[l] -> outs[l].
template <class F>
int msm_table_run_t(zk_ctx* ctx, const ZkHostTable& t, size_t nu, int n_lanes, const void* const* scalars_dev, size_t n, void* const* outs, uint64_t sfp) {
    ZkBasesLease lease;
    ZK_TRY(zk_bases_cache_get(ctx, t, nu, &lease));
    const bool spec_ok = spec_enabled() && n_lanes == 1 && !lease.temporary && sfp != 0 && !ctx->profiling &&
                         !(ctx->msm_spec && ((ZkMsmSpec*)ctx->msm_spec)->off);
    ZkMsmSpec* sp = spec_ok || ctx->msm_spec ? spec_of(ctx) : nullptr;
    int rc = ZK_OK;
    SpecJob* take = nullptr;
    if (sp && !sp->jobs.empty()) {                           // is this the call the speculation was made for?
        SpecJob* j = sp->jobs.front().get();
        bool same = false;
        if (spec_ok && j->table == lease.b && sp->n == n && sp->fp == sfp) rc = spec_same_scalars(ctx, sp, scalars_dev[0], n, &same);
        if (rc == ZK_OK && same) take = j;
        else {
            if ((spec_ok || j->table != lease.b) && sp->bad[j->table] < 3) sp->bad[j->table]++;
            spec_drop(ctx, sp);
        }
    }
    if (sp && sp->fft_N) {                                   // the transform before this call: were its outputs this call's scalars?
        const size_t N = sp->fft_N;
        if (spec_ok && rc == ZK_OK && (n == N || n + 1 == N) && sfp == sp->fft_fp[n == N ? 0 : 1]) {
            const size_t key = N * 4 + (size_t)sp->fft_kind;
            auto it = sp->fft_succ.find(key);
            if (it != sp->fft_succ.end() && it->second.table == lease.b && it->second.n == n) {
                auto bd = sp->bad.find(lease.b);
                if (bd != sp->bad.end() && bd->second > 0 && --bd->second == 0) sp->bad.erase(bd);
            }
            sp->fft_succ[key] = ZkMsmSpec::AfterFft{lease.b, n};
        }
        sp->fft_N = 0;
    }
    if (sp && spec_ok && rc == ZK_OK) {                      // learn: the same scalars as the call before, another table
        if (sp->last_table && sp->last_table != lease.b && sp->last_n == n && sp->last_fp == sfp) {
            auto it = sp->succ.find(sp->last_table);
            if (it != sp->succ.end() && it->second == lease.b) {     // the succession held again: a table given up after two wrong
                auto bd = sp->bad.find(lease.b);                     // guesses in a row earns its way back, one confirmation at a time
                if (bd != sp->bad.end() && bd->second > 0 && --bd->second == 0) sp->bad.erase(bd);
            }
            sp->succ[sp->last_table] = lease.b;
        }
        sp->last_table = lease.b; sp->last_n = n; sp->last_fp = sfp;
    } else if (sp) {
        sp->last_table = nullptr;
    }
    for (int attempt = 0; attempt < 2 && rc == ZK_OK; attempt++) {
        bool same = true;
        int vrc = ZK_OK;
        {
            ZkTask<void> verifier;                           // (joins when this block is left, whatever the MSM did)
            if (lease.verify)
                verifier = zk_async(ctx, [&] { (void)hipSetDevice(ctx->device); vrc = zk_bases_cache_verify(ctx, &lease, t, nu, &same); });
            if (take) {                                      // the result that was started during the previous call
                rc = zk_msm_finish(ctx, &take->job, outs[0]);
                sp->bad.erase(sp->jobs.front()->table);     // (two wrong guesses IN A ROW give a table up: spec_start)
                sp->jobs.erase(sp->jobs.begin());
                sp->taken++;
                take = nullptr;
            } else if (n_lanes == 1) {
                // The MSMs this caller asked for next the last time it came with this table, on the side stream.  A SMALL call is a chain
                // of latencies on a mostly idle chip: they start at once, behind nothing but the copy of the scalars, and run beside this
                // call's own kernels (2^10 .. 2^16: -12 .. -20 % per proof).  A LARGE call fills the chip: started at once they only delay
                // this call's result (2^20: the A call 4 -> 13.5 ms, the three calls 16.6 -> 21.6), so they queue behind this call's
                // whole device chain (a reduce chain beside a running accumulate kernel is starved) and run under this call's host
                // half and the caller's way back into the library.
                const bool want_spec = spec_ok && attempt == 0 && sp->jobs.empty();
                const bool small_call = lease.b->pre ? n * ((255 + lease.b->c_pre - 1) / lease.b->c_pre) <= ((size_t)1 << 22) : n <= ((size_t)1 << 17);
                if (want_spec && small_call) (void)spec_start(ctx, sp, lease.b, scalars_dev[0], n, sfp);
                ZkMsmJob job;
                rc = msm_prepare_t<F>(ctx, &job, lease.b, 0, scalars_dev[0], n, 0);
                if (rc == ZK_OK) rc = msm_enqueue_sort_t<F>(ctx, &job, ctx->stream, nullptr);
                if (rc == ZK_OK) rc = msm_enqueue_accum_t<F>(ctx, &job, ctx->stream);
                if (rc == ZK_OK) rc = msm_enqueue_reduce_t<F>(ctx, &job, ctx->stream);
                if (rc == ZK_OK && want_spec && !small_call) (void)spec_start(ctx, sp, lease.b, scalars_dev[0], n, sfp);
                if (rc == ZK_OK) rc = msm_finish_t<F>(ctx, &job, outs[0]);
                else (void)hipStreamSynchronize(ctx->stream);
            } else {
                const zk_bases* bs[2] = {lease.b, lease.b};
                const size_t lens[2] = {n, n};
                rc = zk_msm_batch_dev(ctx, (size_t)n_lanes, bs, nullptr, scalars_dev, lens, outs);
            }
        }
        if (rc == ZK_OK) rc = vrc;
        if (rc != ZK_OK || same) break;
        rc = zk_bases_cache_replace(ctx, &lease);            // the table at the caller's address is not the cached one any more: again, on the right one
    }
    if (rc != ZK_OK && sp) spec_drop(ctx, sp);
    zk_bases_lease_release(ctx, &lease);
    return rc;
}

// 64 sampled elements and the length: how a repeated scalar vector is recognised (a candidate: see spec_same_scalars)
uint64_t scalars_fingerprint(const zk_fr* s, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    const size_t S = n < 64 ? n : 64;
    for (size_t k = 0; k < S; k++) {
        const size_t i = S > 1 ? k * (n - 1) / (S - 1) : 0;
        for (int w = 0; w < 4; w++) { h ^= s[i].l[w]; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 32; }
    }
    return h ? h : 1;
}

ZkHostTable host_table(int group, const void* host, const ZkAffineLayout* layout) {
    ZkHostTable t;
    t.group = group;
    t.host = host;
    const size_t FE = group == 1 ? 48 : 96;
    if (layout) { t.stride = layout->stride; t.off_x = layout->off_x; t.off_y = layout->off_y; t.off_inf = layout->off_inf; }
    else { t.stride = 2 * FE; t.off_x = 0; t.off_y = FE; t.off_inf = SIZE_MAX; }
    return t;
}

template <class F>
int msm_host_t(zk_ctx* ctx, const void* bases_host, size_t nb, const ZkAffineLayout* layout, const zk_fr* scalars, size_t ns, int group, void* out) {
    if (!ctx || !out) return ZK_ERR_ARG;
    const size_t n = std::min(nb, ns);  // variable_base.rs:15-17
    if (n && (!bases_host || !scalars)) return ZK_ERR_ARG;
    if (n == 0) {
        host_write_projective<F>(aff_inf<F>(), (uint64_t*)out);
        return ZK_OK;
    }
    const size_t nu = nb <= 2 * n ? nb : n;                  // the whole slice (it is the cache's key) unless most of it is unused
    void* sdev = nullptr;
    ZK_TRY(zk_scratch(ctx, "msm_scalars_host", n * 32, &sdev));
    ZK_TRY(zk_xfer_h2d(ctx, sdev, scalars, n * 32));
    const void* sc[1] = {sdev};
    void* outs[1] = {out};
    return msm_table_run_t<F>(ctx, host_table(group, bases_host, layout), nu, 1, sc, n, outs, scalars_fingerprint(scalars, n));
}

}  // namespace

// ---- phase timer ----
ZkPhaseTimer::ZkPhaseTimer(zk_ctx* c, hipStream_t st) : ctx(c), stream(st ? st : c->stream) { enabled = c->profiling; }
ZkPhaseTimer::~ZkPhaseTimer() {
    for (auto& e : ev) { (void)hipEventDestroy(e.second.first); (void)hipEventDestroy(e.second.second); }
}
void ZkPhaseTimer::begin(const char* name) {
    if (!enabled) return;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, stream);
    ev.push_back({name, {a, b}});
}
void ZkPhaseTimer::end() {
    if (!enabled || ev.empty()) return;
    (void)hipEventRecord(ev.back().second.second, stream);
}
void ZkPhaseTimer::resolve() {
    if (!enabled || resolved) return;
    resolved = true;
    // wait for each phase's own end event, never for the stream: the accumulate stream holds the kernels of ALL in-flight
    // jobs, and synchronising it from the first job's finish() made the host wait for the whole proof (the host-side
    // chains of create_proof then started after the GPU had gone idle: +1.1 ms per proof with the timers on)
    for (auto& e : ev) {
        float ms = 0;
        if (hipEventSynchronize(e.second.second) == hipSuccess &&
            hipEventElapsedTime(&ms, e.second.first, e.second.second) == hipSuccess) {
            std::lock_guard<std::mutex> lk(ctx->mu);           // (the host halves of a group's jobs resolve side by side)
            ctx->timers[e.first].ms += ms;
            ctx->timers[e.first].count += 1;
        }
    }
}

ZkMsmJob::~ZkMsmJob() {
    for (auto* t : timers) delete t;
    if (accum_done) (void)hipEventDestroy(accum_done);
    if (sort_done) (void)hipEventDestroy(sort_done);
    if (reduce_done) (void)hipEventDestroy(reduce_done);
}

int zk_msm_prepare(zk_ctx* ctx, ZkMsmJob* job, const zk_bases* bases, size_t base_offset, const void* scalars_dev, size_t n, int slot) {
    if (bases->group == 1) return msm_prepare_t<G1Field>(ctx, job, bases, base_offset, scalars_dev, n, slot);
    return msm_prepare_t<G2Field>(ctx, job, bases, base_offset, scalars_dev, n, slot);
}
int zk_msm_enqueue_sort(zk_ctx* ctx, ZkMsmJob* job, hipStream_t st, const ZkMsmJob* share) {
    if (job->group == 1) return msm_enqueue_sort_t<G1Field>(ctx, job, st, share);
    return msm_enqueue_sort_t<G2Field>(ctx, job, st, share);
}
int zk_msm_enqueue_accum(zk_ctx* ctx, ZkMsmJob* job, hipStream_t st) {
    if (job->group == 1) return msm_enqueue_accum_t<G1Field>(ctx, job, st);
    return msm_enqueue_accum_t<G2Field>(ctx, job, st);
}
int zk_msm_enqueue_reduce(zk_ctx* ctx, ZkMsmJob* job, hipStream_t st) {
    if (job->group == 1) return msm_enqueue_reduce_t<G1Field>(ctx, job, st);
    return msm_enqueue_reduce_t<G2Field>(ctx, job, st);
}
// The host halves of several finished jobs (wait for the copy, Horner chain, one inversion: 30 - 60 us each) side by side on the
// helper threads: behind a group launch they all become ready at once, and four in a row were a tenth of a small proof.
int zk_msm_finish_many(zk_ctx* ctx, ZkMsmJob* const* jobs, void* const* outs, int count) {
    if (count < 2) {
        for (int k = 0; k < count; k++) ZK_TRY(zk_msm_finish(ctx, jobs[k], outs[k]));
        return ZK_OK;
    }
    std::vector<int> rc((size_t)count, ZK_OK);
    {
        std::vector<ZkTask<void>> tasks;
        for (int k = 1; k < count; k++) tasks.push_back(zk_async(ctx, [&, k] { (void)hipSetDevice(ctx->device); rc[(size_t)k] = zk_msm_finish(ctx, jobs[k], outs[k]); }));
        rc[0] = zk_msm_finish(ctx, jobs[0], outs[0]);
    }
    for (int k = 0; k < count; k++) ZK_TRY(rc[(size_t)k]);
    return ZK_OK;
}
bool zk_msm_group_ok(ZkMsmJob* const* jobs, int count) { return msm_group_ok(jobs, count); }
bool zk_msm_sort_group_ok(ZkMsmJob* const* jobs, int count) { return msm_sort_group_ok(jobs, count); }
int zk_msm_enqueue_sort_group(zk_ctx* ctx, ZkMsmJob* const* jobs, int count, hipStream_t st) { return msm_enqueue_sort_group(ctx, jobs, count, st); }
int zk_msm_enqueue_accum_group(zk_ctx* ctx, ZkMsmJob* const* jobs, int count, hipStream_t st) { return msm_enqueue_accum_group(ctx, jobs, count, st); }
int zk_msm_enqueue_reduce_group(zk_ctx* ctx, ZkMsmJob* const* jobs, int count, hipStream_t st) { return msm_enqueue_reduce_group(ctx, jobs, count, st); }
int zk_msm_finish(zk_ctx* ctx, ZkMsmJob* job, void* out) {
    if (job->group == 1) return msm_finish_t<G1Field>(ctx, job, out);
    return msm_finish_t<G2Field>(ctx, job, out);
}

int zk_msm_run(zk_ctx* ctx, const zk_bases* bases, size_t base_offset, const void* scalars_dev, size_t n, void* out) {
    if (bases->group == 1) return msm_run_t<G1Field>(ctx, bases, base_offset, scalars_dev, n, out);
    return msm_run_t<G2Field>(ctx, bases, base_offset, scalars_dev, n, out);
}

int zk_bases_upload_host(zk_ctx* ctx, int group, const void* host, size_t n, const ZkAffineLayout* layout, zk_bases** out) {
    if (group == 1) return bases_upload_t<G1Field>(ctx, host, n, 1, layout, out);
    return bases_upload_t<G2Field>(ctx, host, n, 2, layout, out);
}
int zk_bases_alloc_dev(zk_ctx* ctx, int group, size_t n, zk_bases** out) {
    std::unique_ptr<zk_bases> b(new zk_bases());
    b->group = group;
    b->n = n;
    if (n && hipMalloc((void**)&b->dev, n * (group == 1 ? 96 : 192)) != hipSuccess) {
        (void)hipGetLastError();
        ZK_FAIL(ctx, ZK_ERR_NOMEM, "base table: hipMalloc failed");
    }
    *out = b.release();
    return ZK_OK;
}
int zk_bases_import_launch(zk_ctx* ctx, zk_bases* b, const void* raw, hipStream_t st) {
    if (!b->n) return ZK_OK;
    if (b->group == 1) hipLaunchKernelGGL(k_bases_import<G1Field>, zk_grid(b->n, 256), 256, 0, st, (const uint32_t*)raw, b->dev, b->n);
    else hipLaunchKernelGGL(k_bases_import<G2Field>, zk_grid(b->n, 256), 256, 0, st, (const uint32_t*)raw, b->dev, b->n);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}
// the MSM of the MPC entry points (mpc_host.hip): n_lanes scalar vectors on the device against one host table
int zk_msm_table_run(zk_ctx* ctx, const ZkHostTable& t, size_t nu, int n_lanes, const void* const* scalars_dev, size_t n, void* const* outs, uint64_t sfp) {
    if (t.group == 1) return msm_table_run_t<G1Field>(ctx, t, nu, n_lanes, scalars_dev, n, outs, sfp);
    return msm_table_run_t<G2Field>(ctx, t, nu, n_lanes, scalars_dev, n, outs, sfp);
}
uint64_t zk_scalars_fingerprint(const void* fr, size_t n) { return scalars_fingerprint((const zk_fr*)fr, n); }
void zk_msm_spec_drop(zk_ctx* ctx) {
    if (ctx->msm_spec) spec_drop(ctx, (ZkMsmSpec*)ctx->msm_spec);
}
// A transform of 2^log_n elements on a caller's host vector has been enqueued (its result is in `dev`, final on the context stream,
// not yet on its way back): if the MSM that followed such a transform last time took the output as its scalars, that MSM starts
// now -- under the download of the result and the upload of the scalars the call will bring (which are compared with `dev`'s copy
// before the result is released, like every job started ahead).
void zk_msm_spec_fft_begin(zk_ctx* ctx, const void* dev, size_t N, int kind) {
    ZkMsmSpec* sp = (ZkMsmSpec*)ctx->msm_spec;
    if (sp) sp->late_dev = nullptr;                          // (a transform that failed between begin and end leaves nothing behind)
    if (!sp || sp->off || !spec_enabled() || ctx->profiling || !sp->jobs.empty()) return;
    auto it = sp->fft_succ.find(N * 4 + (size_t)kind);        // (of the seven transforms of a witness map only the last -- the one coset
    if (it == sp->fft_succ.end()) return;                     // inverse -- writes the H query's scalars: the kind is part of what is learned)
    const zk_bases* t = it->second.table;
    const size_t n = it->second.n;
    if (sp->bad[t] >= 2 || n > t->n || n > N) return;
    const bool small_call = t->pre ? n * ((255 + t->c_pre - 1) / t->c_pre) <= ((size_t)1 << 22) : n <= ((size_t)1 << 17);
    // A small job starts now, under the download of the transform's result.  A large one would share a hardware queue with that
    // download (2^20: the last transform 1.7 -> 3.2 ms, measured): it starts when the result is through (zk_msm_spec_fft_end) and
    // runs under the caller's way back into the library and the upload of the scalars.
    if (!small_call) { sp->late_dev = dev; sp->late_table = t; sp->late_n = n; return; }
    if (spec_start_tables(ctx, sp, &t, 1, dev, n, 0) == ZK_OK && !sp->jobs.empty()) sp->fp_pending = true;
}
// ... and the result has arrived in the caller's vector: fp_of(n) = the fingerprint an MSM entry point would compute over its first n
// elements (plain: scalars_fingerprint; MpcField: mpc_host.hip's)
void zk_msm_spec_fft_end(zk_ctx* ctx, size_t N, int kind, const std::function<uint64_t(size_t)>& fp_of) {
    if (!spec_enabled() || ctx->profiling || N < 2) return;
    ZkMsmSpec* sp = spec_of(ctx);
    if (sp->off) return;
    sp->fft_N = N;
    sp->fft_kind = kind;
    sp->fft_fp[0] = fp_of(N);
    sp->fft_fp[1] = fp_of(N - 1);
    if (sp->fp_pending && !sp->jobs.empty()) sp->fp = sp->n == N ? sp->fft_fp[0] : sp->fft_fp[1];
    sp->fp_pending = false;
    if (sp->late_dev && sp->jobs.empty() && (sp->late_n == N || sp->late_n + 1 == N))
        (void)spec_start_tables(ctx, sp, &sp->late_table, 1, sp->late_dev, sp->late_n, sp->fft_fp[sp->late_n == N ? 0 : 1]);
    sp->late_dev = nullptr;
}
extern "C" int zk_msm_speculate(zk_ctx* ctx, int on) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ZkMsmSpec* sp = spec_of(ctx);
    sp->off = on == 0;
    if (sp->off) { spec_drop(ctx, sp); sp->succ.clear(); sp->bad.clear(); sp->fft_succ.clear(); sp->fft_N = 0; sp->last_table = nullptr; }
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_msm_speculate_stats(zk_ctx* ctx, uint64_t out[3]) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !out) return ZK_ERR_ARG;
    ZkMsmSpec* sp = (ZkMsmSpec*)ctx->msm_spec;
    out[0] = sp ? sp->started : 0; out[1] = sp ? sp->taken : 0; out[2] = sp ? sp->dropped : 0;
    return ZK_OK;
    ZK_API_END
}
void zk_msm_spec_forget(zk_ctx* ctx, const zk_bases* b) {
    ZkMsmSpec* sp = (ZkMsmSpec*)ctx->msm_spec;
    if (!sp) return;
    for (auto& j : sp->jobs) if (j->table == b) { spec_drop(ctx, sp); break; }
    if (sp->last_table == b) sp->last_table = nullptr;
    sp->succ.erase(b);
    sp->bad.erase(b);
    for (auto it = sp->fft_succ.begin(); it != sp->fft_succ.end();) it = it->second.table == b ? sp->fft_succ.erase(it) : std::next(it);
    if (sp->late_table == b) sp->late_dev = nullptr;
    for (auto it = sp->succ.begin(); it != sp->succ.end();) it = it->second == b ? sp->succ.erase(it) : std::next(it);
}
void zk_msm_spec_free(zk_ctx* ctx) {
    ZkMsmSpec* sp = (ZkMsmSpec*)ctx->msm_spec;
    if (!sp) return;
    spec_drop(ctx, sp);
    if (sp->flag_dev) (void)hipFree(sp->flag_dev);
    if (sp->flag_host) (void)hipHostFree(sp->flag_host);
    delete sp;
    ctx->msm_spec = nullptr;
}
extern "C" int zk_bases_upload_g1(zk_ctx* ctx, const zk_g1_affine* h, size_t n, zk_bases** out) { ZK_API_BEGIN(ctx) return bases_upload_t<G1Field>(ctx, h, n, 1, nullptr, out); ZK_API_END }
extern "C" int zk_bases_upload_g2(zk_ctx* ctx, const zk_g2_affine* h, size_t n, zk_bases** out) { ZK_API_BEGIN(ctx) return bases_upload_t<G2Field>(ctx, h, n, 2, nullptr, out); ZK_API_END }
extern "C" int zk_bases_free(zk_ctx* ctx, zk_bases* b) {
    ZK_API_BEGIN(ctx)
    if (!b) return ZK_OK;
    zk_presort_free(ctx);          // a pending presort's job points into b->dev / b->pre (groth16_key.hip: zk_pk_free)
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    if (b->owned && b->dev) (void)hipFree(b->dev);
    if (b->pre) (void)hipFree(b->pre);
    delete b;
    return ZK_OK;
    ZK_API_END
}
extern "C" size_t zk_bases_len(const zk_bases* b) { return b ? b->n : 0; }
extern "C" int zk_bases_download_g1(zk_ctx* ctx, const zk_bases* b, size_t off, size_t n, zk_g1_affine* out) {
    ZK_API_BEGIN(ctx)
    if (b && b->group != 1) return ZK_ERR_ARG;
    return bases_download_t<G1Field>(ctx, b, off, n, out);
    ZK_API_END
}
extern "C" int zk_bases_download_g2(zk_ctx* ctx, const zk_bases* b, size_t off, size_t n, zk_g2_affine* out) {
    ZK_API_BEGIN(ctx)
    if (b && b->group != 2) return ZK_ERR_ARG;
    return bases_download_t<G2Field>(ctx, b, off, n, out);
    ZK_API_END
}

extern "C" int zk_msm_g1(zk_ctx* ctx, const zk_g1_affine* bases, size_t nb, const zk_fr* scalars, size_t ns, zk_g1_projective* out) {
    ZK_API_BEGIN(ctx)
    return msm_host_t<G1Field>(ctx, bases, nb, nullptr, scalars, ns, 1, out);
    ZK_API_END
}
extern "C" int zk_msm_g2(zk_ctx* ctx, const zk_g2_affine* bases, size_t nb, const zk_fr* scalars, size_t ns, zk_g2_projective* out) {
    ZK_API_BEGIN(ctx)
    return msm_host_t<G2Field>(ctx, bases, nb, nullptr, scalars, ns, 2, out);
    ZK_API_END
}
static bool layout_from_abi(const zk_affine_layout* l, ZkAffineLayout* o) {
    if (!l) return false;
    o->stride = l->stride; o->off_x = l->off_x; o->off_y = l->off_y; o->off_inf = l->off_infinity;
    return true;
}
extern "C" int zk_msm_g1_strided(zk_ctx* ctx, const void* bases, size_t nb, const zk_affine_layout* layout, const zk_fr* scalars, size_t ns,
                                 zk_g1_projective* out) {
    ZK_API_BEGIN(ctx)
    ZkAffineLayout lay;
    if (!layout_from_abi(layout, &lay)) return ZK_ERR_ARG;
    return msm_host_t<G1Field>(ctx, bases, nb, &lay, scalars, ns, 1, out);
    ZK_API_END
}
extern "C" int zk_msm_g2_strided(zk_ctx* ctx, const void* bases, size_t nb, const zk_affine_layout* layout, const zk_fr* scalars, size_t ns,
                                 zk_g2_projective* out) {
    ZK_API_BEGIN(ctx)
    ZkAffineLayout lay;
    if (!layout_from_abi(layout, &lay)) return ZK_ERR_ARG;
    return msm_host_t<G2Field>(ctx, bases, nb, &lay, scalars, ns, 2, out);
    ZK_API_END
}
extern "C" int zk_msm_g1_dev(zk_ctx* ctx, const zk_bases* bases, size_t off, const void* scalars, size_t n, zk_g1_projective* out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !bases || !out || bases->group != 1 || (n && !scalars)) return ZK_ERR_ARG;
    return msm_run_t<G1Field>(ctx, bases, off, scalars, n, out);
    ZK_API_END
}
extern "C" int zk_msm_g2_dev(zk_ctx* ctx, const zk_bases* bases, size_t off, const void* scalars, size_t n, zk_g2_projective* out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !bases || !out || bases->group != 2 || (n && !scalars)) return ZK_ERR_ARG;
    return msm_run_t<G2Field>(ctx, bases, off, scalars, n, out);
    ZK_API_END
}
