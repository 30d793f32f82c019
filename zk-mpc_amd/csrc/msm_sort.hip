// msm_sort.hip -- the bucket sort of the MSM: every non-zero digit of every scalar, grouped by bucket.
//
// Step of VariableBaseMSM::multi_scalar_mul (arkworks/algebra/ec/src/msm/variable_base.rs:47-76: "for each scalar, add the base
// to bucket[digit - 1]"; zero scalars are filtered out first, :40-44) restated so that one lane can own one bucket: the list of
// (table entry | sign) values ordered by bucket, and the bucket boundaries.  Only the GROUPING matters -- the order inside a bucket
// does not (the group is commutative, outputs are canonical) -- and the pairs need not exist before they are grouped, so this is
// not a general sort:
//   k_hist     digits are computed from the scalars and counted per BIN of 2^F consecutive buckets (LDS counters, one global add
//              per block and bin);
//   k_scan     one block: bin starts, the number T of non-zero digits, the segment length the accumulate kernel will use for THIS
//              input (ctr[3]; see below), and the work list of the bin kernels (a bin longer than CH entries is cut into chunks);
//   k_scatter  the digits are computed AGAIN (cheaper than keeping 13 pairs per scalar around), a block reserves one run per bin with
//              one global atomic and writes (low key bits, entry) pairs into its runs;
//   k_bins_pre chunks of long bins only: their low-key histogram, added into the bin's global one;
//   k_bins     one block per bin (or chunk): 2^F-counter histogram in LDS, scan -> the bucket offsets of the bin, written straight to
//              offs; the entries go to their bucket's range -- through an LDS stage and out as one contiguous copy for an ordinary
//              bin, straight to memory behind per-bucket global cursors for a chunk of a long one.
// A zero digit produces no pair at all: the all-zero scalars of a witness, and the twelve zero digits of every 0/1 or small-valued
// entry (the reference's circuits are boolean-heavy, docs/benchmark.md:45-58), cost one digit computation and nothing else.  A
// heavy bucket (every "1" of a boolean witness lands in bucket 1 of the lowest window) makes one bin long; its chunks are
// independent blocks, and equal keys inside a wave are counted with ONE LDS atomic (lds_inc_agg).
// Round 3 sorted materialised (bucket, entry) pairs with rocPRIM's radix_sort_pairs: content-independent, 0.38 ms for the 13.6 M
// pairs of a 2^20-scalar MSM, but its size is a host argument, so zero digits could not be dropped without a host round trip.
#include "msm_digits.cuh"
#include "internal.hpp"

using namespace zk;

namespace {

constexpr uint32_t G_TILE = 2048;        // scalars per block step of k_hist / k_scatter (1024 lanes x 2)
constexpr uint32_t G_NT = 1024;
constexpr uint32_t G_CH = 32768;         // entries a bin block stages in LDS (128 KiB); longer bins are cut into chunks of this size
constexpr uint32_t G_MAXNC = 4096;       // bins (k_scan: 4 per lane of one block)
constexpr uint32_t G_NONE = 0xffffffffu;

struct GroupGeom { uint32_t W, NB, merged, n_tab, tab_off, NBt, F, NC; };

// words of (canonical scalar + bias) of scalar i into column `col` of kw (9 x TW words)
template <uint32_t TW>
__device__ __forceinline__ void load_scalar_words(const void* scalars, size_t i, const Bias& bias, uint32_t (*kw)[TW], uint32_t col) {
    uint32_t w9[9];
    scalar_biased_words(scalars, i, bias, w9);
#pragma unroll
    for (int k = 0; k < 9; k++) kw[k][col] = w9[k];
}
// digit w of the scalar in column col: false for a zero digit; bucket = its id among all NBt buckets, neg = the sign bit of the entry
template <uint32_t TW>
__device__ __forceinline__ bool tile_digit(const uint32_t (*kw)[TW], uint32_t col, const WinOff& wo, uint32_t w, const GroupGeom& g,
                                           uint32_t& bucket, uint32_t& neg) {
    const uint32_t bit = wo.off[w], wi = bit >> 5;
    uint64_t two = kw[wi][col];
    if (wi + 1 < 9) two |= (uint64_t)kw[wi + 1][col] << 32;
    const int32_t d = signed_digit(two, bit, wo.off[w + 1] - bit);
    if (d == 0) return false;
    const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
    bucket = (g.merged ? 0u : w * g.NB) + mag - 1;
    neg = d < 0 ? 0x80000000u : 0u;
    return true;
}

__global__ void __launch_bounds__(G_NT)
k_hist(const void* scalars, size_t n, WinOff wo, Bias bias, GroupGeom g, uint32_t* bin_count) {
    extern __shared__ uint32_t g_lds[];
    uint32_t (*kw)[G_TILE] = reinterpret_cast<uint32_t (*)[G_TILE]>(g_lds);
    uint32_t* cnt = g_lds + 9 * G_TILE;
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = tid; b < g.NC; b += G_NT) cnt[b] = 0;
    __syncthreads();
    for (size_t t0 = (size_t)blockIdx.x * G_TILE; t0 < n; t0 += (size_t)gridDim.x * G_TILE) {
        for (uint32_t k = 0; k < 2; k++) {
            const uint32_t col = tid + k * G_NT;
            const size_t i = t0 + col;
            if (i >= n) continue;
            load_scalar_words<G_TILE>(scalars, i, bias, kw, col);
            for (uint32_t w = 0; w < g.W; w++) {
                uint32_t bucket, neg;
                if (tile_digit<G_TILE>(kw, col, wo, w, g, bucket, neg)) atomicAdd(&cnt[bucket >> g.F], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < g.NC; b += G_NT)
        if (cnt[b]) atomicAdd(&bin_count[b], cnt[b]);
}

// block-wide inclusive scan of three values per lane (1024 lanes)
__device__ __forceinline__ void scan3(uint32_t (*part)[G_NT], uint32_t tid) {
    __syncthreads();
    for (uint32_t d = 1; d < G_NT; d <<= 1) {
        uint32_t x[3];
#pragma unroll
        for (int q = 0; q < 3; q++) x[q] = tid >= d ? part[q][tid - d] : 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 3; q++) part[q][tid] += x[q];
        __syncthreads();
    }
}

// One block.  bin_start[b] (b <= NC), cursor = a copy k_scatter consumes; items[k] = (bin, chunk) for the bin kernels, hdr[0] = their
// number, hdr[1] = the number of long bins, bin_long[b] = the index of b among the long bins or G_NONE.
// ctr[4] = T, the number of non-zero digits; ctr[3] = the segment length of the accumulate kernel: the smallest power of two that
// still gives every resident lane a whole segment, T / lanes, inside [32, seg_max] -- seg_max is what a full-density input would
// get.  (A 0/1-heavy witness has a tenth of the digits: with the host's full-density guess its heavy bucket was cut into
// 128-addition segments, 128 x 8 us of one lane, while most of the chip had nothing to do.)
__global__ void __launch_bounds__(G_NT)
k_scan_bins(const uint32_t* bin_count, GroupGeom g, uint32_t lanes, uint32_t seg_max, uint32_t* bin_start, uint32_t* cursor, uint2* items,
            uint32_t* bin_long, uint32_t* hdr, uint32_t* ctr) {
    __shared__ uint32_t part[3][G_NT];
    const uint32_t tid = threadIdx.x;
    uint32_t v[4], ch[4], s = 0, s2 = 0, s3 = 0;
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t b = tid * 4 + k;
        v[k] = b < g.NC ? bin_count[b] : 0;
        ch[k] = b < g.NC ? (v[k] > G_CH ? (v[k] + G_CH - 1) / G_CH : 1u) : 0u;
        s += v[k]; s2 += ch[k]; s3 += ch[k] > 1 ? 1u : 0u;
    }
    part[0][tid] = s; part[1][tid] = s2; part[2][tid] = s3;
    scan3(part, tid);
    uint32_t run = part[0][tid] - s, item = part[1][tid] - s2, li = part[2][tid] - s3;
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t b = tid * 4 + k;
        if (b < g.NC) {
            bin_start[b] = run;
            cursor[b] = run;
            bin_long[b] = ch[k] > 1 ? li : G_NONE;
            for (uint32_t j = 0; j < ch[k]; j++) items[item + j] = make_uint2(b, j);
        }
        run += v[k]; item += ch[k]; li += ch[k] > 1 ? 1u : 0u;
    }
    if (tid == G_NT - 1) {
        const uint32_t T = part[0][G_NT - 1];
        bin_start[g.NC] = T;
        hdr[0] = part[1][G_NT - 1];
        hdr[1] = part[2][G_NT - 1];
        uint32_t seg = seg_max < 32 ? seg_max : 32;
        while (seg < seg_max && (uint64_t)seg * lanes < T) seg <<= 1;
        ctr[3] = seg;
        ctr[4] = T;
    }
}

// exclusive scan over the 1024 lanes of `mine` (lanes >= 2^F pass 0); wsum = 16 words of LDS
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t mine, uint32_t* wsum, uint32_t tid) {
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(inc, d, 64);
        if ((tid & 63) >= (uint32_t)d) inc += x;
    }
    __syncthreads();
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (tid >> 6); w++) before += wsum[w];
    return before + inc - mine;
}

// The scatter, staged: a block takes S_TILE scalars at a time and a group of windows whose pairs fit the LDS stage; it counts the
// group's pairs per bin, reserves one run per bin with one global atomic, ORDERS the pairs by bin in LDS and copies the stage out
// linearly -- consecutive lanes write consecutive words of a run, a wave's store touches a handful of lines instead of 64.
// (Round 4's first version stored every pair straight from its lane: 13.6 M pairs x two scattered stores were 114 of the
// kernel's 167 us at 2^20, and in the pipeline they competed with the accumulate kernel's gathers.)
constexpr uint32_t S_TILE = 1024;        // scalars per pass of k_scatter_bins (one per lane)
constexpr uint32_t S_WG = 8;             // windows per pass: S_TILE * S_WG pairs at most in the stage
constexpr uint32_t S_CAP = S_TILE * S_WG;
// LDS words: kw[9][S_TILE] | cursor[NC] | delta[NC] | wsum[16] | stage_val[S_CAP] | stage_key+bin[S_CAP] (two 16-bit halves)
static size_t scatter_lds_bytes(uint32_t NC) { return (size_t)(9 * S_TILE + 2 * NC + 16 + 2 * S_CAP) * 4; }

__global__ void __launch_bounds__(G_NT)
k_scatter_bins(const void* scalars, size_t n, WinOff wo, Bias bias, GroupGeom g, uint32_t* cursor, uint16_t* key_lo, uint32_t* val) {
    extern __shared__ uint32_t g_lds[];
    uint32_t (*kw)[S_TILE] = reinterpret_cast<uint32_t (*)[S_TILE]>(g_lds);
    uint32_t* cur = g_lds + 9 * S_TILE;            // per bin: count, then the bin's cursor inside the stage
    uint32_t* delta = cur + g.NC;                  // per bin: (start of this block's run in the bin) - (start of the bin in the stage)
    uint32_t* wsum = delta + g.NC;
    uint32_t* st_val = wsum + 16;
    uint32_t* st_kb = st_val + S_CAP;              // low half: the key inside the bin, high half: the bin
    const uint32_t tid = threadIdx.x, fmask = (1u << g.F) - 1;
    const uint32_t per = (g.NC + G_NT - 1) / G_NT; // bins per lane in the scan (<= 4)
    for (size_t t0 = (size_t)blockIdx.x * S_TILE; t0 < n; t0 += (size_t)gridDim.x * S_TILE) {
        const size_t i = t0 + tid;
        const bool have = i < n;
        __syncthreads();                           // the previous tile's last pass is through with kw
        if (have) load_scalar_words<S_TILE>(scalars, i, bias, kw, tid);
        for (uint32_t w0 = 0; w0 < g.W; w0 += S_WG) {
            const uint32_t w1 = min(w0 + S_WG, g.W);
            for (uint32_t b = tid; b < g.NC; b += G_NT) cur[b] = 0;
            __syncthreads();
            if (have)
                for (uint32_t w = w0; w < w1; w++) {
                    uint32_t bucket, neg;
                    if (tile_digit<S_TILE>(kw, tid, wo, w, g, bucket, neg)) atomicAdd(&cur[bucket >> g.F], 1u);
                }
            __syncthreads();
            // exclusive scan of the counts over the bins (lane t owns bins t*per .. t*per+per-1), one global reservation per bin
            uint32_t c[4], mine = 0;
            for (uint32_t k = 0; k < per; k++) {
                const uint32_t b = tid * per + k;
                c[k] = b < g.NC ? cur[b] : 0;
                mine += c[k];
            }
            uint32_t off = block_excl_scan(mine, wsum, tid);
            __syncthreads();
            for (uint32_t k = 0; k < per; k++) {
                const uint32_t b = tid * per + k;
                if (b < g.NC) {
                    cur[b] = off;
                    if (c[k]) delta[b] = atomicAdd(&cursor[b], c[k]) - off;
                    off += c[k];
                }
            }
            __syncthreads();
            if (have)
                for (uint32_t w = w0; w < w1; w++) {
                    uint32_t bucket, neg;
                    if (!tile_digit<S_TILE>(kw, tid, wo, w, g, bucket, neg)) continue;
                    const uint32_t bin = bucket >> g.F;
                    const uint32_t p = atomicAdd(&cur[bin], 1u);
                    st_val[p] = (g.merged ? (uint32_t)(w * g.n_tab + g.tab_off + i) : (uint32_t)i) | neg;
                    st_kb[p] = (bucket & fmask) | (bin << 16);
                }
            __syncthreads();
            uint32_t cnt_all = 0;                  // block_excl_scan left the sixteen wave totals in wsum
            for (uint32_t k = 0; k < 16; k++) cnt_all += wsum[k];
            for (uint32_t j = tid; j < cnt_all; j += G_NT) {
                const uint32_t kb = st_kb[j];
                const uint32_t gp = j + delta[kb >> 16];
                key_lo[gp] = (uint16_t)kb;
                val[gp] = st_val[j];
            }
        }
    }
}

// arr[key]++ for the active lanes, returning every lane's own position.  When ALL active lanes of the wave hold the same key -- the
// chunk of a heavy bucket -- one lane adds the count and the others take consecutive positions: one LDS atomic instead of 64
// serialised on one address.  The test is wave-uniform (two ballots and a broadcast).
__device__ __forceinline__ uint32_t lds_inc_agg(uint32_t* arr, uint32_t key) {
    const uint64_t act = __ballot(1);
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    const uint64_t same = __ballot(key == first);
    if (same == act) {
        const uint32_t lane = __lane_id();
        const uint32_t rank = (uint32_t)__popcll(same & (((uint64_t)1 << lane) - 1));
        uint32_t base = 0;
        if (rank == 0) base = atomicAdd(&arr[first], (uint32_t)__popcll(same));
        const int leader = __ffsll((unsigned long long)same) - 1;
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        return base + rank;
    }
    return atomicAdd(&arr[key], 1u);
}

// LDS of the bin kernels: hist / cursors [1024] | per-wave totals [64] | bucket offsets of a long bin [1024] | stage [G_CH]
constexpr uint32_t BINS_LDS_WORDS = 1024 + 64 + 1024 + G_CH;

__global__ void __launch_bounds__(G_NT)
k_bins_pre(const uint16_t* __restrict__ key_lo, const uint32_t* __restrict__ bin_start, const uint2* __restrict__ items,
           const uint32_t* __restrict__ bin_long, const uint32_t* __restrict__ hdr, uint32_t* __restrict__ ghist) {
    __shared__ uint32_t hist[1024];
    if (blockIdx.x >= hdr[0]) return;
    const uint2 it = items[blockIdx.x];
    const uint32_t li = bin_long[it.x];
    if (li == G_NONE) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t lo = bin_start[it.x] + it.y * G_CH, hi = min(lo + G_CH, bin_start[it.x + 1]);
    hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = lo + tid; i < hi; i += G_NT) (void)lds_inc_agg(hist, key_lo[i]);
    __syncthreads();
    if (hist[tid]) atomicAdd(&ghist[(size_t)li * 2048 + tid], hist[tid]);
}

__global__ void __launch_bounds__(G_NT)
k_bins(const uint16_t* __restrict__ key_lo, const uint32_t* __restrict__ val, const uint32_t* __restrict__ bin_start,
       const uint2* __restrict__ items, const uint32_t* __restrict__ bin_long, const uint32_t* __restrict__ hdr, uint32_t* __restrict__ ghist,
       GroupGeom g, uint32_t* __restrict__ sorted, uint32_t* __restrict__ offs) {
    extern __shared__ uint32_t g_lds[];
    uint32_t* hist = g_lds;                      // 2^F counters, later cursors
    uint32_t* wsum = g_lds + 1024;
    uint32_t* boff = g_lds + 1024 + 64;
    uint32_t* stage = g_lds + 1024 + 64 + 1024;
    if (blockIdx.x >= hdr[0]) return;
    const uint2 it = items[blockIdx.x];
    const uint32_t b = it.x, tid = threadIdx.x, nb = 1u << g.F;
    const uint32_t li = bin_long[b];
    const uint32_t blo = bin_start[b], bhi = bin_start[b + 1];
    const uint32_t lo = blo + it.y * G_CH, hi = li == G_NONE ? bhi : min(lo + G_CH, bhi), cnt = hi - lo;
    hist[tid] = 0;
    __syncthreads();
    for (uint32_t i0 = lo + tid; i0 < hi; i0 += 4 * G_NT) {           // four loads in flight per lane
        uint32_t k[4];
#pragma unroll
        for (int u = 0; u < 4; u++) k[u] = i0 + u * G_NT < hi ? key_lo[i0 + u * G_NT] : G_NONE;
#pragma unroll
        for (int u = 0; u < 4; u++) if (k[u] != G_NONE) (void)lds_inc_agg(hist, k[u]);
    }
    __syncthreads();
    const uint32_t id = (b << g.F) + tid;
    if (li == G_NONE) {
        // an ordinary bin: its own histogram is the bin's
        const uint32_t mine = tid < nb ? hist[tid] : 0;
        const uint32_t off = block_excl_scan(mine, wsum, tid);
        if (tid < nb && id <= g.NBt) offs[id] = blo + off;
        if (b == g.NC - 1 && tid == nb - 1 && ((g.NC << g.F) == g.NBt)) offs[g.NBt] = bhi;       // the end marker when the last bin is full
        __syncthreads();
        if (tid < nb) hist[tid] = off;                               // now the cursors
        __syncthreads();
        for (uint32_t i0 = lo + tid; i0 < hi; i0 += 4 * G_NT) {
            uint32_t k[4], v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool in = i0 + u * G_NT < hi;
                k[u] = in ? key_lo[i0 + u * G_NT] : G_NONE;
                v[u] = in ? val[i0 + u * G_NT] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (k[u] != G_NONE) stage[lds_inc_agg(hist, k[u])] = v[u];
        }
        __syncthreads();
        for (uint32_t i = tid; i < cnt; i += G_NT) sorted[blo + i] = stage[i];      // scattered 4-byte stores would cost a 64-byte transaction each
        return;
    }
    // a chunk of a long bin: the bin's histogram is the global one (k_bins_pre); this chunk reserves its part of every bucket's range
    uint32_t* gh = ghist + (size_t)li * 2048;
    const uint32_t total_k = tid < nb ? gh[tid] : 0;
    const uint32_t off = block_excl_scan(total_k, wsum, tid);
    if (it.y == 0) {
        if (tid < nb && id <= g.NBt) offs[id] = blo + off;
        if (b == g.NC - 1 && tid == nb - 1 && ((g.NC << g.F) == g.NBt)) offs[g.NBt] = bhi;
    }
    __syncthreads();
    if (tid < nb) {
        const uint32_t c = hist[tid];
        hist[tid] = off + (c ? atomicAdd(&gh[1024 + tid], c) : 0u);   // cursor: the bucket's start + what other chunks took before
    }
    __syncthreads();
    for (uint32_t i0 = lo + tid; i0 < hi; i0 += 4 * G_NT) {
        uint32_t k[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool in = i0 + u * G_NT < hi;
            k[u] = in ? key_lo[i0 + u * G_NT] : G_NONE;
            v[u] = in ? val[i0 + u * G_NT] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) if (k[u] != G_NONE) sorted[blo + lds_inc_agg(hist, k[u])] = v[u];
    }
    (void)boff;
}

}  // namespace

// Geometry: bins of 2^F buckets with at most 2^10 buckets, at most G_MAXNC bins, and -- where those two allow it -- a mean bin
// of ~0.85 G_CH entries at full density so that an ordinary bin fits the LDS stage.
static GroupGeom group_geom(const ZkGroupArgs& a) {
    GroupGeom g;
    g.W = a.W; g.NB = a.NB; g.merged = a.merged ? 1u : 0u; g.n_tab = a.n_tab; g.tab_off = a.tab_off; g.NBt = a.NBt;
    uint32_t lg_nbt = 0;
    while (((uint64_t)1 << lg_nbt) < (uint64_t)a.NBt) lg_nbt++;
    const uint64_t total = (uint64_t)a.W * a.n;
    const uint64_t want = (total + 27799) / 27800;
    uint32_t lg_nc = 0;
    while (((uint64_t)1 << lg_nc) < want) lg_nc++;
    const uint32_t upper = lg_nbt < 12 ? lg_nbt : 12, lower = lg_nbt > 10 ? lg_nbt - 10 : 0;
    if (lg_nc > upper) lg_nc = upper;
    if (lg_nc < lower) lg_nc = lower;                   // (lower > upper, a bucket set beyond 2^22: F comes out above 10 and is refused)
    g.F = lg_nbt - (lg_nc < lg_nbt ? lg_nc : lg_nbt);
    g.NC = (uint32_t)(((uint64_t)a.NBt + ((uint64_t)1 << g.F) - 1) >> g.F);   // bins cover the ids [0, NC 2^F); id NBt is the end marker
    if (g.NC == 0) g.NC = 1;
    return g;
}

bool zk_msm_group_supported(const ZkGroupArgs& a) {
    const GroupGeom g = group_geom(a);
    return g.NC <= G_MAXNC && g.F <= 10;
}

int zk_msm_group(zk_ctx* ctx, hipStream_t st, int slot, const ZkGroupArgs& a) {
    const GroupGeom g = group_geom(a);
    if (g.NC > G_MAXNC || g.F > 10) ZK_FAIL(ctx, ZK_ERR_ARG, "msm: bucket set too large for the bucket sort");
    const size_t total = (size_t)a.W * a.n;
    const size_t max_long = total / G_CH + 1, max_items = (size_t)g.NC + total / G_CH + 1;
    char nm[64];
    auto name = [&](const char* base) { snprintf(nm, sizeof nm, "%s.%d", base, slot); return nm; };
    uint16_t* key_lo;
    uint32_t *val, *bins, *ghist;
    uint2* items;
    ZK_TRY(zk_scratch(ctx, name("msm_keylo"), total * 2, (void**)&key_lo));
    ZK_TRY(zk_scratch(ctx, name("msm_vals"), total * 4, (void**)&val));
    // bins: bin_count[NC+1] | bin_start[NC+1] | cursor[NC+1] | bin_long[NC] | hdr[2]
    ZK_TRY(zk_scratch(ctx, name("msm_bins"), (size_t)(4 * (g.NC + 1) + 2) * 4, (void**)&bins));
    ZK_TRY(zk_scratch(ctx, name("msm_items"), max_items * sizeof(uint2), (void**)&items));
    ZK_TRY(zk_scratch(ctx, name("msm_ghist"), max_long * 2048 * 4, (void**)&ghist));       // per long bin: histogram[1024] | cursors[1024]
    uint32_t *bin_count = bins, *bin_start = bins + (g.NC + 1), *cursor = bins + 2 * (g.NC + 1), *bin_long = bins + 3 * (g.NC + 1),
             *hdr = bins + 4 * (g.NC + 1);
    if (!ctx->flags["group_lds"]) {
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (9 * G_TILE + G_MAXNC) * 4));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_scatter_bins, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds_bytes(G_MAXNC)));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_bins, hipFuncAttributeMaxDynamicSharedMemorySize, BINS_LDS_WORDS * 4));
        ctx->flags["group_lds"] = 1;
    }
    ZK_HIP(ctx, hipMemsetAsync(bin_count, 0, (size_t)(g.NC + 1) * 4, st));
    ZK_HIP(ctx, hipMemsetAsync(ghist, 0, max_long * 2048 * 4, st));
    const unsigned tiles = (unsigned)((a.n + G_TILE - 1) / G_TILE);
    const unsigned pg = tiles < 512 ? tiles : 512;
    hipLaunchKernelGGL(k_hist, pg, G_NT, (9 * G_TILE + g.NC) * 4, st, a.scalars, a.n, a.wo, a.bias, g, bin_count);
    hipLaunchKernelGGL(k_scan_bins, 1, G_NT, 0, st, (const uint32_t*)bin_count, g, a.lanes, a.seg_max, bin_start, cursor, items, bin_long, hdr,
                       a.ctr);
    hipLaunchKernelGGL(k_scatter_bins, pg, G_NT, scatter_lds_bytes(g.NC), st, a.scalars, a.n, a.wo, a.bias, g, cursor, key_lo, val);
    hipLaunchKernelGGL(k_bins_pre, (unsigned)max_items, G_NT, 0, st, (const uint16_t*)key_lo, (const uint32_t*)bin_start, (const uint2*)items,
                       (const uint32_t*)bin_long, (const uint32_t*)hdr, ghist);
    hipLaunchKernelGGL(k_bins, (unsigned)max_items, G_NT, BINS_LDS_WORDS * 4, st, (const uint16_t*)key_lo, (const uint32_t*)val,
                       (const uint32_t*)bin_start, (const uint2*)items, (const uint32_t*)bin_long, (const uint32_t*)hdr, ghist, g, a.sorted, a.offs);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}
