// msm_g2pair.hip -- G2 bucket accumulation with TWO lanes per point addition (gfx950).
//
// Same job as k_accum<Fq2Field> of msm.hip (the inner loop of VariableBaseMSM::multi_scalar_mul,
// arkworks/algebra/ec/src/msm/variable_base.rs:47-76, for G2Affine of bls12_377/src/curves/g2.rs), same
// inputs and the same group element out; only the mapping of the arithmetic onto lanes differs.
//
// Why: the one-lane-per-addition G2 kernel needs the whole register file (256 VGPR + 217 AGPR: one wave per
// SIMD, nothing to issue while a gather or a dependent mad is in flight) and its loop body is ~110 KB of code,
// more than the instruction cache.  Here the even lane of a pair holds the c0 component of every Fq2 value and
// the odd lane the c1 component (Fq2 = Fq[u]/(u^2 + 5), ff/src/fields/models/quadratic_extension.rs:632-643):
//   * add / sub / neg / dbl are component-wise: no communication;
//   * a product (a0 + a1 u)(b0 + b1 u) is ONE fused double product per lane (fp29.cuh::fp_mul2):
//       even: a0 b0 + (-5 a1) b1      odd: a1 b0 + a0 b1
//     the partner's limbs arrive by DPP quad_perm [1,0,3,2] (13 v_mov_dpp per operand, no LDS);
//   * zero tests OR the limbs and exchange one word.
// Per lane this is half the live state (256 registers: two waves per SIMD, no spills) and half the code; the number of
// v_mad_u64_u32 per addition is about the same (10 x 1 014; a squaring costs a product here).  By the hardware counters the
// kernel issues a VALU instruction in every cycle (profiles/r1_pmc_valu.json): 17 350 instructions per addition, 62 % of them
// 64-bit multiply-adds.
// A pair shares its control flow (all branch conditions are pair-wide), so the partner lane is always active.
#include "devutil.cuh"
#include "../../include/zkmpc_hip.h"
#include "internal.hpp"
#include "msm_reduce.cuh"
#include "ec_dual.cuh"
#include <stdlib.h>

using namespace zk;

namespace {

struct SegDesc { uint32_t start, len, dst; };   // msm.hip

using B = FqField;
constexpr int L = FqParams::L;
constexpr int FW = FqParams::W;   // packed words per Fq: 12

__device__ __forceinline__ uint32_t dpp_swap1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
__device__ __forceinline__ Fq xchg(const Fq& a) {
    Fq r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = dpp_swap1(a.l[i]);
    return r;
}
__device__ __forceinline__ Fq sel(bool c, const Fq& a, const Fq& b) { return B::select(c, a, b); }

// pair-wide "both components are zero" of the OR of some limbs
__device__ __forceinline__ bool pair_zero(uint32_t o) { return (o | dpp_swap1(o)) == 0; }
__device__ __forceinline__ uint32_t limbs_or(const Fq& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < L; i++) o |= a.l[i];
    return o;
}

// A left operand made ready for products, prepared once per distinct left operand: (p, q) such that the product with
// b = (b0, b1) is p * (own b) + q * (partner's b) on BOTH lanes --
//   even lane: p = a0, q = -5 a1  ->  a0 b0 + (-5 a1) b1        odd lane: p = a0, q = a1  ->  a0 b1 + a1 b0.
// The lane-dependent selects are paid here, not in every product.
struct Left { Fq p, q; };

__device__ __forceinline__ Left prep(const Fq& a, bool odd) {
    Fq m5 = fp_neg5_almost<FqParams>(a);   // -5 a, almost reduced: only ever the c operand of fp_mul2
    Fq r = xchg(sel(odd, m5, a));          // even receives -5 a1, odd receives a0
    return Left{sel(odd, r, a), sel(odd, a, r)};
}
__device__ __forceinline__ Fq mulp(const Left& A, const Fq& b, bool odd) {
    (void)odd;
    return fp_mul2<FqParams, true>(A.p, b, A.q, xchg(b));
}

template <bool TOPSPLIT = false>
__device__ __forceinline__ Fq mulp_l(const Left& A, const Fq& b) {     // lazy domain: no final subtraction, result < p + eps
    return fp_mul2_lazy<FqParams, TOPSPLIT, true>(A.p, b, A.q, xchg(b));
}

struct AffP { Fq x, y; };
struct XyzzP { Fq x, y, zz, zzz; };

__device__ __forceinline__ Fq one_p(bool odd) { return sel(odd, B::zero(), B::one()); }

// 2 * (x, y), affine and not infinity ("mdbl-2008-s-1", a = 0); rare path (a bucket receiving the same point twice)
__device__ __forceinline__ XyzzP dbl_affine_p(const AffP& q, bool odd) {
    Fq u = B::dbl(q.y);
    if (pair_zero(limbs_or(u))) return XyzzP{B::zero(), B::zero(), B::zero(), B::zero()};
    Left U = prep(u, odd);
    Fq v = mulp(U, u, odd);
    Fq w = mulp(U, v, odd);
    Left X = prep(q.x, odd);
    Fq s = mulp(X, v, odd);
    Fq xx = mulp(X, q.x, odd);
    Fq m = B::add(B::dbl(xx), xx);
    Left M = prep(m, odd);
    Fq x3 = B::sub(mulp(M, m, odd), B::dbl(s));
    Fq y3 = B::sub(mulp(M, B::sub(s, x3), odd), mulp(prep(w, odd), q.y, odd));
    return XyzzP{x3, y3, v, w};
}

// acc + q ("madd-2008-s"), complete like ec.cuh::xyzz_madd
__device__ __forceinline__ XyzzP madd_p(const XyzzP& acc, const AffP& q, bool odd) {
    if (pair_zero(limbs_or(q.x) | limbs_or(q.y))) return acc;
    if (pair_zero(limbs_or(acc.zz))) return XyzzP{q.x, q.y, one_p(odd), one_p(odd)};
    Fq u2 = mulp(prep(q.x, odd), acc.zz, odd);
    Fq s2 = mulp(prep(q.y, odd), acc.zzz, odd);
    Fq p = B::sub(u2, acc.x);
    Fq r = B::sub(s2, acc.y);
    if (pair_zero(limbs_or(p))) {
        if (pair_zero(limbs_or(r))) return dbl_affine_p(q, odd);
        return XyzzP{B::zero(), B::zero(), B::zero(), B::zero()};
    }
    Fq pp = mulp(prep(p, odd), p, odd);
    Left PP = prep(pp, odd);
    Fq ppp = mulp(PP, p, odd);
    Fq qq = mulp(PP, acc.x, odd);
    Left R = prep(r, odd);
    Fq x3 = B::sub(B::sub(mulp(R, r, odd), ppp), B::dbl(qq));
    Left PPP = prep(ppp, odd);
    Fq y3 = B::sub(mulp(R, B::sub(qq, x3), odd), mulp(PPP, acc.y, odd));
    return XyzzP{x3, y3, mulp(PP, acc.zz, odd), mulp(PPP, acc.zzz, odd)};
}

// acc + q in the lazy domain (fp29.cuh; the G1 form is ec.cuh::xyzz_madd_lazy): per lane the same ranges as there except
// that Y3 is a difference of two products, M1 + 2p - M2 < 3p + eps, hence R = s2 + 4p - Y1 < 5p + eps.  q is not infinity
// (the caller tests the table's own words); acc is all-zero words when it is infinity.  The equal-x case is found exactly:
// P = 0 in Fq2 iff both components are multiples of p, whose low limbs are 1 .. 7 -- only then is anything reduced.
__device__ __forceinline__ XyzzP madd_p_lazy(const XyzzP& acc, const AffP& q, bool odd) {
    if (pair_zero(limbs_or(acc.zz))) return XyzzP{q.x, q.y, one_p(odd), one_p(odd)};
    const Fq u2 = mulp_l(prep(q.x, odd), acc.zz);
    const Fq s2 = mulp_l(prep(q.y, odd), acc.zzz);
    const Fq p = B::sub_kp<6>(u2, acc.x);
    const Fq r = B::sub_kp<4>(s2, acc.y);
    const uint32_t maybe = B::maybe_multiple_of_p(p) ? 1u : 0u;
    if (maybe & dpp_swap1(maybe)) {
        if (pair_zero(limbs_or(B::canon(p)))) {
            if (pair_zero(limbs_or(B::canon(r)))) return dbl_affine_p(AffP{q.x, B::canon1(q.y)}, odd);
            return XyzzP{B::zero(), B::zero(), B::zero(), B::zero()};
        }
    }
    const Fq pp = mulp_l<true>(prep(p, odd), p);      // the odd lane's p0 p1 + p1 p0 has FOUR wide operands (fp_mul2_lazy: TOPSPLIT)
    const Left PP = prep(pp, odd);
    const Fq ppp = mulp_l(PP, p);
    const Fq qq = mulp_l(PP, acc.x);
    const Left R = prep(r, odd);
    const Fq x3 = B::x3_l(mulp_l(R, r), ppp, qq);
    const Left PPP = prep(ppp, odd);
    const Fq y3 = B::sub_kp<2>(mulp_l(R, B::sub_kp<6>(qq, x3)), mulp_l(PPP, acc.y));
    return XyzzP{x3, y3, mulp_l(PP, acc.zz), mulp_l(PPP, acc.zzz)};
}

// ---- lane QUADS for small jobs ---------------------------------------------------------------------------------------------------
// A small MSM is a chain of dependent additions on waves that sit alone on their SIMD: there every addition
// runs on two pairs, each taking one of the two independent Fq2 products of a step (ec_dual.cuh) -- seven product times instead of
// fourteen for a full addition, five instead of ten for a mixed one.  Lane bit 0 = the Fq2 component as above, lane bit 1 = the
// pair; both pairs carry the whole point, the first one stores.
struct G2QuadBase {
    using T = Fq;
    static __device__ __forceinline__ bool odd() { return (threadIdx.x & 1u) != 0; }
    static __device__ __forceinline__ bool hi() { return (threadIdx.x & 2u) != 0; }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return mulp_l(prep(a, odd()), b); }
    static __device__ __forceinline__ T mul_wide(const T& a, const T& b) { return mulp_l<true>(prep(a, odd()), b); }
    static __device__ __forceinline__ T swap(const T& a) {
        T r;
#pragma unroll
        for (int i = 0; i < L; i++) r.l[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.l[i], 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
        return r;
    }
    static __device__ __forceinline__ T zero() { return B::zero(); }
    static __device__ __forceinline__ T one() { return one_p(odd()); }
    static __device__ __forceinline__ bool is_zero(const T& a) { return pair_zero(limbs_or(a)); }
    static __device__ __forceinline__ bool maybe_multiple_of_p(const T& a) {
        const uint32_t m = B::maybe_multiple_of_p(a) ? 1u : 0u;
        return (m & dpp_swap1(m)) != 0;
    }
    static __device__ __forceinline__ T select(bool c, const T& a, const T& b) { return sel(c, a, b); }
    template <int K> static __device__ __forceinline__ T sub_kp(const T& a, const T& b) { return B::sub_kp<K>(a, b); }
    static __device__ __forceinline__ T x3_l(const T& rr, const T& ppp, const T& qq) { return B::x3_l(rr, ppp, qq); }
    static __device__ __forceinline__ T canon(const T& a) { return B::canon(a); }
    static __device__ __forceinline__ T canon1(const T& a) { return B::canon1(a); }
    static __device__ __forceinline__ XyzzP dbl_affine(const AffP& q) { return dbl_affine_p(q, odd()); }
};

__device__ __forceinline__ Fq fq_load16(const uint32_t* w) { return felt_load16<FqField>(w); }

// this lane's components of affine point i: x.c[odd] at words [odd*12, +12), y.c[odd] at [24 + odd*12, +12)
__device__ __forceinline__ AffP aff_load_p(const uint32_t* bases, size_t i, uint32_t odd) {
    const uint32_t* w = bases + i * (4 * FW) + odd * FW;
    return AffP{fq_load16(w), fq_load16(w + 2 * FW)};
}

struct RawP { uint4 x[FW / 4], y[FW / 4]; };
__device__ __forceinline__ RawP fetch_p(const uint32_t* bases, size_t i, uint32_t odd) {
    const uint4* w = reinterpret_cast<const uint4*>(bases + i * (4 * FW) + odd * FW);
    RawP r;
#pragma unroll
    for (int k = 0; k < FW / 4; k++) { r.x[k] = w[k]; r.y[k] = w[2 * FW / 4 + k]; }
    return r;
}
__device__ __forceinline__ AffP unpack_p(const RawP& r) {
    uint32_t tx[FW], ty[FW];
#pragma unroll
    for (int k = 0; k < FW / 4; k++) {
        tx[4 * k] = r.x[k].x; tx[4 * k + 1] = r.x[k].y; tx[4 * k + 2] = r.x[k].z; tx[4 * k + 3] = r.x[k].w;
        ty[4 * k] = r.y[k].x; ty[4 * k + 1] = r.y[k].y; ty[4 * k + 2] = r.y[k].z; ty[4 * k + 3] = r.y[k].w;
    }
    return AffP{B::load(tx), B::load(ty)};
}

// 256 registers, two waves per SIMD, no spills.  (One wave per SIMD with 192 registers per lane left free for the kernels of the
// other streams was measured slower: 7.7 vs 7.2 ms alone, 21.3 vs 20.8 ms per proof in round 2.)
// GL = 2: a lane pair per segment; GL = 4: a quad (small jobs, above).
template <int GL>
__global__ void __launch_bounds__(256, 2)
k_accum_g2pair(const uint32_t* __restrict__ bases, const uint32_t* __restrict__ sorted, const SegDesc* __restrict__ desc,
               const uint32_t* __restrict__ order, const uint32_t* __restrict__ ctr, uint32_t* __restrict__ sums) {
    constexpr int SH = GL == 4 ? 2 : 1;
    const uint32_t S = ctr[2];
    const uint32_t G = gridDim.x * (blockDim.x >> SH);
    const uint32_t oddw = threadIdx.x & 1u;
    const bool odd = oddw != 0;
    for (uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> SH; t < S; t += G) {
        const SegDesc d = desc[order[t]];
        XyzzP acc{B::zero(), B::zero(), B::zero(), B::zero()};
        if (d.len) {
            const uint32_t* srt = sorted + d.start;
            // the next point travels as raw words under the current addition and is unpacked when its turn comes (msm.hip::k_accum)
            uint32_t e = srt[0], e1 = d.len > 1 ? srt[1] : 0;
            RawP nxt = fetch_p(bases, e & 0x7fffffffu, oddw);
            for (uint32_t k = 0; k < d.len; k++) {
                AffP cur = unpack_p(nxt);
                const uint32_t ce = e;
                if (k + 1 < d.len) {
                    e = e1;
                    nxt = fetch_p(bases, e & 0x7fffffffu, oddw);
                    if (k + 2 < d.len) e1 = srt[k + 2];
                }
                // lazy domain: see msm.hip::k_accum
                const bool inf = pair_zero(limbs_or(cur.x) | limbs_or(cur.y));
                if (ce >> 31) cur.y = B::kp_minus<1>(cur.y);
                if (!inf) {
                    if constexpr (GL == 4) acc = xyzz_madd_dual<G2QuadBase, XyzzP, AffP>(acc, cur);
                    else acc = madd_p_lazy(acc, cur, odd);
                }
            }
            acc = XyzzP{B::canon(acc.x), B::canon(acc.y), B::canon1(acc.zz), B::canon1(acc.zzz)};
        }
        if (GL == 4 && G2QuadBase::hi()) continue;
        // XYZZ over Fq2 in memory: x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1, 12 words each
        uint32_t* w = sums + (size_t)d.dst * (8 * FW) + oddw * FW;
        felt_store16<FqField>(w, acc.x);
        felt_store16<FqField>(w + 2 * FW, acc.y);
        felt_store16<FqField>(w + 4 * FW, acc.zz);
        felt_store16<FqField>(w + 6 * FW, acc.zzz);
    }
}


// ---- the reduce phase (fold / chunked running sums / bit-decomposition sums of msm.hip) on lane pairs --------------------
// The one-lane G2 versions need 489-512 registers (k_reduce spills 692 B per lane) and, one wave per SIMD with 2^16
// threads per level, leave most of the chip idle; on pairs they fit 256 registers and have twice the waves.
// The curve formulas are ec.cuh's, instantiated with a field whose element is this lane's component.
struct Fq2PairField {
    using T = Fq;
    static constexpr int WORDS = FqParams::W;   // words per LANE per element
    static __device__ __forceinline__ bool odd() { return (threadIdx.x & 1u) != 0; }
    static __device__ __forceinline__ T zero() { return B::zero(); }
    static __device__ __forceinline__ T one() { return one_p(odd()); }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return B::add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return B::sub(a, b); }
    static __device__ __forceinline__ T dbl(const T& a) { return B::dbl(a); }
    static __device__ __forceinline__ T neg(const T& a) { return B::neg(a); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return mulp(prep(a, odd()), b, odd()); }
    static __device__ __forceinline__ T sqr(const T& a) { return mul(a, a); }
    static __device__ __forceinline__ T mulsub(const T& a, const T& b, const T& c, const T& d) { return B::sub(mul(a, b), mul(c, d)); }
    static __device__ __forceinline__ bool is_zero(const T& a) { return pair_zero(limbs_or(a)); }
    static __device__ __forceinline__ T load(const uint32_t* w) { return B::load(w); }
    static __device__ __forceinline__ void store(uint32_t* w, const T& a) { B::store(w, a); }
};
using FP = Fq2PairField;
using XP = XYZZ<FP>;

// XYZZ over Fq2 in memory: x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1 (12 words each); a lane touches its four
__device__ __forceinline__ XP xyzz_load_pair(const uint32_t* base, size_t i, uint32_t odd) {
    const uint32_t* w = base + i * (8 * FW) + odd * FW;
    return XP{fq_load16(w), fq_load16(w + 2 * FW), fq_load16(w + 4 * FW), fq_load16(w + 6 * FW)};
}
__device__ __forceinline__ void xyzz_store_pair(uint32_t* base, size_t i, uint32_t odd, const XP& p) {
    uint32_t* w = base + i * (8 * FW) + odd * FW;
    felt_store16<FqField>(w, p.x);
    felt_store16<FqField>(w + 2 * FW, p.y);
    felt_store16<FqField>(w + 4 * FW, p.zz);
    felt_store16<FqField>(w + 6 * FW, p.zzz);
}
// LDS staging of this lane's half of a point, word-major over the NT physical lanes (conflict-free)
template <int NT>
__device__ __forceinline__ void lds_put_pair(uint32_t* lds, uint32_t tid, const XP& p) {
    uint32_t w[4 * FW];
    xyzz_store<FP>(w, p);
#pragma unroll
    for (int k = 0; k < 4 * FW; k++) lds[k * NT + tid] = w[k];
}
template <int NT>
__device__ __forceinline__ XP lds_get_pair(const uint32_t* lds, uint32_t tid) {
    uint32_t w[4 * FW];
#pragma unroll
    for (int k = 0; k < 4 * FW; k++) w[k] = lds[k * NT + tid];
    return xyzz_load<FP>(w);
}

struct HeavyDesc { uint32_t key, first, nseg, parent; };   // msm.hip
constexpr uint32_t HEAVY_NONE = 0xffffffffu;
constexpr uint32_t FOLD_LIGHT = 8;

// the 32 lane pairs (GL = 4: 16 quads) of a block add nseg partial sums (strided, then an LDS tree) into sums[key]
template <int GL>
__device__ __forceinline__ XP fold_add(const XP& a, const XP& b);
template <int GL>
__device__ __forceinline__ void fold_block_pair(uint32_t* lds, uint32_t* sums, uint32_t first, uint32_t nseg, uint32_t key, uint32_t tid) {
    constexpr uint32_t SH = GL == 4 ? 2 : 1, NG = 64 >> SH;
    const uint32_t lt = tid >> SH, odd = tid & 1u;
    const bool writer = GL == 2 || (tid & 2u) == 0;
    const uint32_t slot = 2 * lt + odd;                              // a point's two halves in neighbouring LDS slots
    XP acc = xyzz_inf<FP>();
    for (uint32_t j = lt; j < nseg; j += NG) acc = fold_add<GL>(acc, xyzz_load_pair(sums, (size_t)first + j, odd));
    if (writer) lds_put_pair<64>(lds, slot, acc);
    __syncthreads();
    for (uint32_t d = NG / 2; d >= 1; d >>= 1) {
        XP r = acc;
        if (lt < d) r = fold_add<GL>(lds_get_pair<64>(lds, slot), lds_get_pair<64>(lds, slot + 2 * d));
        __syncthreads();
        if (lt < d && writer) lds_put_pair<64>(lds, slot, r);
        __syncthreads();
    }
    if (lt == 0 && writer) xyzz_store_pair(sums, key, odd, lds_get_pair<64>(lds, slot));
    __syncthreads();
}

// msm.hip::k_fold on pairs: 32 buckets (light part) or one entry (heavy part) per 64-lane block; two-level buckets as there.
// GL = 4: on quads (small jobs: 16 buckets per block).
template <int GL>
__global__ void __launch_bounds__(64)
k_fold_g2pair(const HeavyDesc* heavy, const HeavyDesc* heavy2, const uint32_t* ctr, uint32_t* done, uint32_t* sums, uint32_t light_blocks) {
    extern __shared__ uint32_t lds[];  // 64 * 48 words (heavy blocks only)
    __shared__ uint32_t last_flag;
    constexpr uint32_t SH = GL == 4 ? 2 : 1, NG = 64 >> SH;
    const uint32_t nheavy = ctr[0];
    const uint32_t tid = threadIdx.x, lt = tid >> SH, odd = tid & 1u;
    const bool writer = GL == 2 || (tid & 2u) == 0;
    if (blockIdx.x < light_blocks) {
        for (uint32_t hb = blockIdx.x * NG + lt; hb < nheavy; hb += light_blocks * NG) {
            const HeavyDesc h = heavy[hb];
            if (h.nseg > FOLD_LIGHT || h.parent != HEAVY_NONE) continue;
            XP acc = xyzz_load_pair(sums, (size_t)h.first, odd);
            for (uint32_t j = 1; j < h.nseg; j++) acc = fold_add<GL>(acc, xyzz_load_pair(sums, (size_t)h.first + j, odd));
            if (writer) xyzz_store_pair(sums, h.key, odd, acc);
        }
        return;
    }
    for (uint32_t hb = blockIdx.x - light_blocks; hb < nheavy; hb += gridDim.x - light_blocks) {
        const HeavyDesc h = heavy[hb];
        if (h.nseg <= FOLD_LIGHT && h.parent == HEAVY_NONE) continue;
        fold_block_pair<GL>(lds, sums, h.first, h.nseg, h.key, tid);
        if (h.parent == HEAVY_NONE) continue;
        const HeavyDesc up = heavy2[h.parent];
        __threadfence();                                           // both lanes of pair 0 stored a half of the group sum
        __syncthreads();
        if (tid == 0) {
            const uint32_t before = atomicAdd(&done[h.parent], 1u);
            last_flag = before + 1 == up.nseg ? 1u : 0u;
            if (last_flag) done[h.parent] = 0;
        }
        __syncthreads();
        if (last_flag) {
            __threadfence();
            fold_block_pair<GL>(lds, sums, up.first, up.nseg, up.key, tid);
        }
        __syncthreads();
    }
}

// a + b ("add-2008-s") on a lane pair in the lazy domain -- the pair form of ec.cuh::xyzz_add_lazy, with the ranges of
// madd_p_lazy above: x < 5p + eps, y < 3p + eps (a difference of two products), zz, zzz < p + eps; fully reduced coordinates
// included; infinity = all-zero words.  The result is in the same ranges.  U1, U2, S1, S2 < p + eps; P = U2 + 2p - U1 and
// R = S2 + 2p - S1 in (p - eps, 3p + eps): every double product stays inside the column bounds that R * T (5p x 7p on both
// lanes) already needs (tests/test_abi.py::test_lazy_domain_column_bounds).  P = 0 in Fq2 iff both components are p, 2p or 3p.
__device__ __forceinline__ XP add_p_lazy(const XP& a, const XP& b, bool odd) {
    if (pair_zero(limbs_or(a.zz))) return b;
    if (pair_zero(limbs_or(b.zz))) return a;
    const Fq u1 = mulp_l(prep(a.x, odd), b.zz);
    const Fq u2 = mulp_l(prep(b.x, odd), a.zz);
    const Fq s1 = mulp_l(prep(a.y, odd), b.zzz);
    const Fq s2 = mulp_l(prep(b.y, odd), a.zzz);
    const Fq p = B::sub_kp<2>(u2, u1);
    const Fq r = B::sub_kp<2>(s2, s1);
    const uint32_t maybe = B::maybe_multiple_of_p(p) ? 1u : 0u;
    if (maybe & dpp_swap1(maybe)) {
        if (pair_zero(limbs_or(B::canon(p)))) {
            if (pair_zero(limbs_or(B::canon(r))))
                return xyzz_dbl<FP>(XP{B::canon(a.x), B::canon(a.y), B::canon1(a.zz), B::canon1(a.zzz)});
            return xyzz_inf<FP>();
        }
    }
    const Fq pp = mulp_l(prep(p, odd), p);
    const Left PP = prep(pp, odd);
    const Fq ppp = mulp_l(PP, p);
    const Fq qq = mulp_l(PP, u1);
    const Left R = prep(r, odd);
    const Fq x3 = B::x3_l(mulp_l(R, r), ppp, qq);
    const Left PPP = prep(ppp, odd);
    const Fq y3 = B::sub_kp<2>(mulp_l(R, B::sub_kp<6>(qq, x3)), mulp_l(PPP, s1));
    return XP{x3, y3, mulp_l(PP, mulp_l(prep(a.zz, odd), b.zz)), mulp_l(PPP, mulp_l(prep(a.zzz, odd), b.zzz))};
}

// Point policy of msm_reduce.cuh for G2: a lane PAIR per point (even lane c0, odd lane c1 of every Fq2 coordinate), 128 points
// per 256-lane block; a point's two halves sit in neighbouring LDS slots.  Sums in the lazy domain; the packed 12-word form
// holds 377 bits, so x and y are brought below p where a sum is packed.
struct RedG2Pair {
    using X = XP;
    static constexpr int NT = 256, PTS = (int)ZK_G2PAIR_RED_PTS, MINW = 2;     // <= 256 registers, like the accumulate kernel
    static __device__ __forceinline__ uint32_t pt() { return threadIdx.x >> 1; }
    static __device__ __forceinline__ uint32_t odd() { return threadIdx.x & 1u; }
    static __device__ __forceinline__ X inf() { return xyzz_inf<FP>(); }
    static __device__ __forceinline__ X load(const uint32_t* base, size_t i) { return xyzz_load_pair(base, i, odd()); }
    static __device__ __forceinline__ void store(uint32_t* base, size_t i, const X& p) { xyzz_store_pair(base, i, odd(), p); }
    static __device__ __forceinline__ X add(const X& a, const X& b) { return add_p_lazy(a, b, odd() != 0); }
    static __device__ __forceinline__ X pack(const X& a) { return X{B::canon(a.x), B::canon(a.y), a.zz, a.zzz}; }
    static __device__ __forceinline__ X canon(const X& a) { return X{B::canon(a.x), B::canon(a.y), B::canon1(a.zz), B::canon1(a.zzz)}; }
    static __device__ __forceinline__ void lds_put(uint32_t* lds, uint32_t slot, const X& p) { lds_put_pair<NT>(lds, 2 * slot + odd(), p); }
    static __device__ __forceinline__ X lds_get(const uint32_t* lds, uint32_t slot) { return lds_get_pair<NT>(lds, 2 * slot + odd()); }
};

struct G2QuadOps : G2QuadBase {
    static __device__ __forceinline__ XP dbl(const XP& a) { return xyzz_dbl<FP>(a); }
};
// the folds' addition: exact on pairs (as before); on quads the lazy form, brought back to reduced coordinates (the fold's sums are
// read by both forms of the grid kernels and by the exact pair addition of a second-level fold)
template <> __device__ __forceinline__ XP fold_add<2>(const XP& a, const XP& b) { return xyzz_add<FP>(a, b); }
template <> __device__ __forceinline__ XP fold_add<4>(const XP& a, const XP& b) {
    const XP r = xyzz_add_dual<G2QuadOps, XP>(a, b);
    return XP{B::canon(r.x), B::canon(r.y), B::canon1(r.zz), B::canon1(r.zzz)};
}
struct RedG2Quad {
    using X = XP;
    static constexpr int NT = 256, PTS = 64, MINW = 1;
    static __device__ __forceinline__ uint32_t pt() { return threadIdx.x >> 2; }
    static __device__ __forceinline__ uint32_t odd() { return threadIdx.x & 1u; }
    static __device__ __forceinline__ X inf() { return xyzz_inf<FP>(); }
    static __device__ __forceinline__ X load(const uint32_t* base, size_t i) { return xyzz_load_pair(base, i, odd()); }
    static __device__ __forceinline__ void store(uint32_t* base, size_t i, const X& p) { if (!G2QuadOps::hi()) xyzz_store_pair(base, i, odd(), p); }
    static __device__ __forceinline__ X add(const X& a, const X& b) { return xyzz_add_dual<G2QuadOps, X>(a, b); }
    static __device__ __forceinline__ X pack(const X& a) { return X{B::canon(a.x), B::canon(a.y), a.zz, a.zzz}; }
    static __device__ __forceinline__ X canon(const X& a) { return X{B::canon(a.x), B::canon(a.y), B::canon1(a.zz), B::canon1(a.zzz)}; }
    static __device__ __forceinline__ void lds_put(uint32_t* lds, uint32_t slot, const X& p) { if (!G2QuadOps::hi()) lds_put_pair<2 * PTS>(lds, 2 * slot + odd(), p); }
    static __device__ __forceinline__ X lds_get(const uint32_t* lds, uint32_t slot) { return lds_get_pair<2 * PTS>(lds, 2 * slot + odd()); }
};
constexpr size_t RED_QUAD_MAX_BUCKETS = (size_t)1 << 14;      // (measured: 2^13 buckets 0.41 -> 0.31 ms of chain, 2^15 nothing)

}  // namespace

// One pair of lanes per segment: `segments` logical threads.
void zk_launch_accum_g2pair(hipStream_t st, size_t segments, const uint32_t* bases, const uint32_t* sorted, const void* desc,
                            const uint32_t* order, const uint32_t* ctr, uint32_t* sums, bool quads) {
    if (quads) {
        const unsigned blocks = (unsigned)((segments + 63) / 64);
        hipLaunchKernelGGL(k_accum_g2pair<4>, blocks, 256, 0, st, bases, sorted, (const SegDesc*)desc, order, ctr, sums);
        return;
    }
    const unsigned blocks = (unsigned)((segments + 127) / 128);
    hipLaunchKernelGGL(k_accum_g2pair<2>, blocks, 256, 0, st, bases, sorted, (const SegDesc*)desc, order, ctr, sums);
}

// The G2 reduce chain of msm.hip::msm_enqueue_reduce_t, same buffers and geometry, on lane pairs.
int zk_launch_reduce_g2pair(zk_ctx* ctx, hipStream_t st, const ZkG2PairReduce& a) {
    if (a.quads)
        hipLaunchKernelGGL(k_fold_g2pair<4>, 2 * a.light_blocks + a.heavy_blocks, 64, 64 * 4 * FW * 4, st, (const HeavyDesc*)a.heavy, (const HeavyDesc*)a.heavy2,
                           a.ctr, a.done, a.sums, 2 * a.light_blocks);
    else
        hipLaunchKernelGGL(k_fold_g2pair<2>, a.light_blocks + a.heavy_blocks, 64, 64 * 4 * FW * 4, st, (const HeavyDesc*)a.heavy, (const HeavyDesc*)a.heavy2,
                           a.ctr, a.done, a.sums, a.light_blocks);
    if (((size_t)a.n_win << a.log_nb) <= RED_QUAD_MAX_BUCKETS) {
        const GridGeom gg = make_grid_geom(a.log_nb, a.n_win, RedG2Quad::PTS);
        const size_t lds = (size_t)2 * RedG2Quad::PTS * 4 * FW * 4;
        hipLaunchKernelGGL(k_grid_l1<RedG2Quad>, gg.row_blocks + gg.col_blocks, RedG2Quad::NT, lds, st, GridSrc{{(const uint32_t*)a.sums, nullptr, nullptr, nullptr}, 0u}, a.rowP, a.colP, gg);
        hipLaunchKernelGGL(k_grid_bits<RedG2Quad>, gg.n_win * grid_nout(gg), RedG2Quad::NT, lds, st, (const uint32_t*)a.rowP, (const uint32_t*)a.colP,
                           a.bits, gg);
        ZK_HIP(ctx, hipGetLastError());
        return ZK_OK;
    }
    const GridGeom gg = make_grid_geom(a.log_nb, a.n_win, RedG2Pair::PTS);
    const size_t lds = (size_t)RedG2Pair::NT * 4 * FW * 4;            // 48 KiB
    hipLaunchKernelGGL(k_grid_l1<RedG2Pair>, gg.row_blocks + gg.col_blocks, RedG2Pair::NT, lds, st, GridSrc{{(const uint32_t*)a.sums, nullptr, nullptr, nullptr}, 0u}, a.rowP, a.colP, gg);
    hipLaunchKernelGGL(k_grid_bits<RedG2Pair>, gg.n_win * grid_nout(gg), RedG2Pair::NT, lds, st, (const uint32_t*)a.rowP, (const uint32_t*)a.colP,
                       a.bits, gg);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}
