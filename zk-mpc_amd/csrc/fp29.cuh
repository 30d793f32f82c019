// fp29.cuh -- prime-field arithmetic for gfx950 in radix 2^29.
//
// Replaces (reference, relative to /root/reference):
//   arkworks/algebra/ff/src/fields/arithmetic.rs:7-57   (Montgomery mul)
//   arkworks/algebra/ff/src/fields/arithmetic.rs:85-172 (squaring)
//   arkworks/algebra/ff/src/fields/macros.rs:698-717,638-651 (add/sub/neg with conditional reduce)
//   arkworks/algebra/ff/src/fields/arithmetic.rs:59-83, macros.rs:464-474 (into_repr / from_repr)
//
// Design (measured on MI355X, tools/ubench_int.hip): v_mad_u64_u32 issues at the same rate as
// v_add_co_u32 (~32 Tinst/s), so what matters is the instruction count, not the multiply count.
// With 29-bit limbs a column of a 13x13-limb product (plus the Montgomery m*p terms) sums to
// < 26 * 2^58 < 2^63, so every limb product is exactly ONE v_mad_u64_u32 into a 64-bit
// accumulator and there are no carry instructions at all inside the product.
//
// An element is L limbs of 29 bits, value = sum l[i] * 2^(29 i), always fully reduced (< p) outside the lazy domain (below).
// mmul(a, b) = a*b / RI mod p with RI = 2^(29 LR), LR = P::LR Montgomery digits (L for Fr, L + 1 for Fq).
// "internal" form of x is x * RI mod p.
// The reference's in-memory form ("external") is W 32-bit words holding x * 2^(32 W) mod p;
// conversion constants are in consts.cuh.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "consts.cuh"

#define ZK_HD __host__ __device__ __forceinline__

namespace zk {

static constexpr uint32_t MASK29 = (1u << 29) - 1;

template <class P>
struct Fp {
    uint32_t l[P::L];
};

template <class P>
ZK_HD Fp<P> fp_zero() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::L; i++) r.l[i] = 0;
    return r;
}

template <class P, int N>
ZK_HD Fp<P> fp_const(const uint32_t (&c)[N]) {
    static_assert(N == P::L, "limb count");
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < P::L; i++) r.l[i] = c[i];
    return r;
}

template <class P>
ZK_HD Fp<P> fp_one() { return fp_const<P>(P::ONE); }

template <class P>
ZK_HD bool fp_is_zero(const Fp<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::L; i++) o |= a.l[i];
    return o == 0;
}

template <class P>
ZK_HD bool fp_eq(const Fp<P>& a, const Fp<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::L; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// t: limbs < 2^29 except possibly the top one; value in [0, 2p).  Returns t mod p.
template <class P>
ZK_HD Fp<P> fp_reduce_once(const uint32_t (&t)[P::L]) {
    constexpr int L = P::L;
    uint32_t u[L];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        int32_t s = (int32_t)t[i] - (int32_t)P::P[i] + c;
        u[i] = (uint32_t)s & MASK29;
        c = s >> 29;
    }
    bool neg = c < 0;  // t < p
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = neg ? t[i] : u[i];
    return r;
}

template <class P>
ZK_HD Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
    constexpr int L = P::L;
    uint32_t t[L];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        uint32_t s = a.l[i] + b.l[i] + c;
        if (i < L - 1) { t[i] = s & MASK29; c = s >> 29; } else { t[i] = s; }
    }
    return fp_reduce_once<P>(t);
}

template <class P>
ZK_HD Fp<P> fp_dbl(const Fp<P>& a) { return fp_add<P>(a, a); }

template <class P>
ZK_HD Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
    constexpr int L = P::L;
    uint32_t d[L], w[L];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        int32_t s = (int32_t)a.l[i] - (int32_t)b.l[i] + c;
        d[i] = (uint32_t)s & MASK29;
        c = s >> 29;
    }
    bool neg = c < 0;
    uint32_t c2 = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        uint32_t s = d[i] + P::P[i] + c2;
        w[i] = s & MASK29;
        c2 = s >> 29;
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = neg ? w[i] : d[i];
    return r;
}

template <class P>
ZK_HD Fp<P> fp_neg(const Fp<P>& a) { return fp_sub<P>(fp_zero<P>(), a); }

// p - a in ONE carry pass: congruent to -a, in (0, p] (it is p, not 0, for a = 0), limbs < 2^29.  Only as an operand of
// a product (fp_mul2's bound 2.68 p holds for an operand <= p); half the instructions of fp_neg.
template <class P>
ZK_HD Fp<P> fp_neg_lazy(const Fp<P>& a) {
    constexpr int L = P::L;
    Fp<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        int32_t s = (int32_t)P::P[i] - (int32_t)a.l[i] + c;
        if (i < L - 1) { r.l[i] = (uint32_t)s & MASK29; c = s >> 29; } else { r.l[i] = (uint32_t)s; }
    }
    return r;
}

// ---- Montgomery products, product scanning with ONE 64-bit accumulator that is never left -------------------------------------
// LR = P::LR Montgomery digits (radix RI = 2^(29 LR)), L = P::L operand limbs.  The *_lazy forms return the value before any
// final subtraction:  (a b + m p) / RI  <  a b / RI + p.
//   LR = L  (Fr, the 753-bit field): operands < p give a result < 2p; fp_mul subtracts p once.
//   LR = L + 1 (Fq): RI = 2^29 * 2^(29 L) > 6e8 p, so ANY operands below 2^(29 L + 3) (up to ~7 p: the top limb may carry up to
//     ~31.6 bits, the limbs below it must be < 2^29) give a result in [0, p + 2^(29 (L - 1) + 6)): "almost reduced", all limbs
//     < 2^29.  That is what lets the bucket-accumulation kernels drop every conditional subtraction (the lazy domain, below).
// Column sums: a column holds at most L products a_i b_j and L products m_i p_j of 29 x 29 bits (2 L * 2^58) plus, for
// unnormalised operands, two products with one wide top limb and (top column only) one with two; for the operand bounds of
// the lazy domain the worst column is < 2^63.8 (tests/test_abi.py::test_lazy_domain_column_bounds recomputes it).
//
// Instruction shape.  Every limb product is one v_mad_u64_u32 whose addend is the running accumulator: a column STARTS from the
// carry of the column before it, so a column costs its products + 2 instructions (digit or mask, shift).  Left alone, LLVM's reassociation orders a column's sum by operand rank:
// the carry, computed last, is added last, i.e. every column becomes a chain from zero plus a 64-bit join: 22-33 more
// instructions per product.  ZK_PIN gives each partial sum a second use (an empty asm that
// only READS it), which is what makes the pass leave the chain in source order; it emits nothing.  (An asm that also WRITES the
// value would do, but gfx950's hazard recogniser puts an s_nop behind every register an asm defines.)
// PIN is a template argument of the lazy products: the bucket-accumulation loops set it (two waves per SIMD at ~200 registers:
// the instruction count is what they pay for).  Everything else keeps the compiler's order -- under a 128-register budget (the
// transforms) the pinned order costs spills (k_ntt_pass: 3-7 -> 61-65 spilled registers, measured slower).
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_PIN(x) do { if constexpr (PIN) asm volatile("" ::"v"(x)); } while (0)
#else
#define ZK_PIN(x) ((void)0)
#endif

// The end of reduction column k: returns the digit m_k (the multiple of p that clears the column) and leaves the start of
// column k + 1 in acc.
// p_0 = 1 (BLS12-377's Fr and Fq are 1 mod 2^46), so with S_k the column's true sum (carry in + products) the digit is any m
// with S_k + m = 0 mod 2^29 and the carry out is (S_k + m) >> 29.  No constant is ever added to a column:
//   k = 0: acc = S_0.  m_0 = 2^29 - (S_0 mod 2^29), in [1, 2^29] (2^29, not 0, for a column that is already clear), so that
//          S_0 + m_0 = ((S_0 >> 29) + 1) 2^29: the carry out is (acc >> 29) + 1, at least 1.
//   k > 0: acc = S_k - 1 (the "+ 1" of the carry is what is NOT in acc; S_k >= 1 because every carry is).  m_k = -S_k mod
//          2^29 = ~acc mod 2^29 (one v_bitop3; -(x + 1) = ~x), in [0, 2^29), and S_k + m_k = acc + 1 + (2^29 - 1 - acc mod
//          2^29) = ((acc >> 29) + 1) 2^29: the carry out is again (acc >> 29) + 1.
// Either way the next column starts from the plain shift, acc >> 29 = S_(k+1) - 1 once its products are in: the first
// multiply-add of a column takes the shifted accumulator as its addend and the "+ 1" is owed until the LAST reduction column,
// where it is paid once (`last`).  (Rounds 3-4 added 2^29 - 1 to every column instead: 13 64-bit additions per product.)
// Digits: m_0 <= 2^29, the others < 2^29; sum m_k 2^(29 k) <= RI, so the lazy result is <= a b / RI + p (p, not 0, for
// a b = 0: congruent, inside every range below, and fp_reduce_once maps it to 0).  Column sums are those of the textbook
// tail minus one.
template <class P, bool PIN>
ZK_HD uint32_t fp_redc_column(uint64_t& acc, bool first, bool last) {
    uint32_t m;
    if constexpr (P::P[0] == 1) {
        m = first ? (1u << 29) - ((uint32_t)acc & MASK29) : ~(uint32_t)acc & MASK29;
        acc >>= 29;
        if (last) acc += 1;
    } else {
        m = ((uint32_t)acc * P::INV) & MASK29;
        acc += (uint64_t)m * P::P[0];
        acc >>= 29;
    }
    ZK_PIN(acc);
    return m;
}
#define ZK_MAD(acc, x, y) do { (acc) += (uint64_t)(x) * (y); ZK_PIN(acc); } while (0)

template <class P, bool PIN = false>
ZK_HD Fp<P> fp_mul_lazy(const Fp<P>& a, const Fp<P>& b) {
    constexpr int L = P::L, LR = P::LR;
    uint32_t m[LR], r[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < LR; k++) {
#pragma unroll
        for (int i = (k >= L ? k - L + 1 : 0); i <= (k < L ? k : L - 1); i++) ZK_MAD(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = (k >= L ? k - L + 1 : 0); i < k; i++) ZK_MAD(acc, m[i], P::P[k - i]);
        m[k] = fp_redc_column<P, PIN>(acc, k == 0, k + 1 == LR);
    }
#pragma unroll
    for (int k = LR; k < LR + L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) ZK_MAD(acc, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - L + 1; i < LR; i++) ZK_MAD(acc, m[i], P::P[k - i]);
        r[k - LR] = (uint32_t)acc & MASK29;
        acc >>= 29;
        ZK_PIN(acc);
    }
    r[L - 1] = (uint32_t)acc;
    Fp<P> o;
#pragma unroll
    for (int i = 0; i < L; i++) o.l[i] = r[i];
    return o;
}

template <class P>
ZK_HD Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
    const Fp<P> t = fp_mul_lazy<P>(a, b);
    return fp_reduce_once<P>(t.l);
}

// (a b + c d) / RI with ONE Montgomery reduction: the two limb products share the column accumulator and the reduction
// terms (3 L^2 mads instead of 4 L^2 for two separate products, no field addition afterwards).  Lazy result < (a b + c d) / RI
// + p.  Used by the Fq2 product: measured (tools/ubench_fq2.hip) 21.9 G Fq2-mul/s against 18.5 for Karatsuba with three
// separate products.
// TOPSPLIT: all four operands may have wide top limbs (up to 7 p each: the odd lane of an Fq2 squaring, a0 a1 + a1 a0).  Then
// the top column alone -- a_top b_top + c_top d_top, 2 * 5.9^2 * 2^58 -- would pass 2^64, so the second of those products is
// entered in two parts, its low 29 bits in that column and the rest one column up (every other column stays below 2^63.8).
template <class P, bool TOPSPLIT = false, bool PIN = false>
ZK_HD Fp<P> fp_mul2_lazy(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
    constexpr int L = P::L, LR = P::LR;
    static_assert(!TOPSPLIT || LR > L, "the split top column is a column of the result part");
    uint32_t m[LR], r[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < LR; k++) {
#pragma unroll
        for (int i = (k >= L ? k - L + 1 : 0); i <= (k < L ? k : L - 1); i++) {
            ZK_MAD(acc, a.l[i], b.l[k - i]);
            ZK_MAD(acc, c.l[i], d.l[k - i]);
        }
#pragma unroll
        for (int i = (k >= L ? k - L + 1 : 0); i < k; i++) ZK_MAD(acc, m[i], P::P[k - i]);
        m[k] = fp_redc_column<P, PIN>(acc, k == 0, k + 1 == LR);
    }
#pragma unroll
    for (int k = LR; k < LR + L - 1; k++) {
        uint64_t up = 0;
#pragma unroll
        for (int i = k - L + 1; i < L; i++) {
            ZK_MAD(acc, a.l[i], b.l[k - i]);
            if (TOPSPLIT && k == 2 * L - 2) {
                const uint64_t t = (uint64_t)c.l[i] * d.l[k - i];
                acc += t & MASK29;
                up = t >> 29;
            } else {
                ZK_MAD(acc, c.l[i], d.l[k - i]);
            }
        }
#pragma unroll
        for (int i = k - L + 1; i < LR; i++) ZK_MAD(acc, m[i], P::P[k - i]);
        r[k - LR] = (uint32_t)acc & MASK29;
        acc = (acc >> 29) + up;
        ZK_PIN(acc);
    }
    r[L - 1] = (uint32_t)acc;
    Fp<P> o;
#pragma unroll
    for (int i = 0; i < L; i++) o.l[i] = r[i];
    return o;
}

// exact form: fully reduced.  LR = L: the value before the subtractions is < (2 p^2 + RI p) / RI < 2.68 p for a 0.84 * 2^(29 L)
// modulus (two conditional subtractions); LR > L: < p + 2^(29 (L - 1) + 7) (one).
template <class P, bool PIN = false>
ZK_HD Fp<P> fp_mul2(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
    constexpr int L = P::L;
    const Fp<P> t = fp_mul2_lazy<P, false, PIN>(a, b, c, d);
    if constexpr (P::LR > P::L) {
        return fp_reduce_once<P>(t.l);
    } else {
        // first subtraction: the difference can still be >= p, so its top limb is kept unmasked (it may exceed 29 bits)
        uint32_t u[L];
        int32_t cy = 0;
#pragma unroll
        for (int i = 0; i < L - 1; i++) {
            int32_t s = (int32_t)t.l[i] - (int32_t)P::P[i] + cy;
            u[i] = (uint32_t)s & MASK29;
            cy = s >> 29;
        }
        const int32_t top = (int32_t)t.l[L - 1] - (int32_t)P::P[L - 1] + cy;
        u[L - 1] = (uint32_t)top;
        const bool below = top < 0;        // t < p
        uint32_t t2[L];
#pragma unroll
        for (int i = 0; i < L; i++) t2[i] = below ? t.l[i] : u[i];
        return fp_reduce_once<P>(t2);      // now < 2p
    }
}

// -5 a mod p in ONE carry pass, almost reduced: returns V = k p - 5 a with k = floor(upper estimate of 5 a / p) + 1, so
// V = -5 a (mod p) and 0 < V <= p (1 + 2e-7) for any a < 8 p (a may be a lazy-domain value with a wide top limb); all limbs
// < 2^29 (p (1 + 2e-7) < 2^(29 L) for BLS12-377's q = 0.84 * 2^377).
// The estimate uses the top limb only: y = (a_top + 1) * ceil(5 * 2^58 / p_top) / 2^58 >= 5 a / p, too large by < 3e-8.
// V is meant as the c operand of fp_mul2 (whose pre-subtraction bound 2.68 p has room for it) in the Fq2 product with
// non-residue -5: the exact neg(mul5(a)) was three additions and a subtraction, each with its conditional reduction
// (~420 instructions, a sixth of a G2 mixed addition); this is ~55.
template <class P>
ZK_HD Fp<P> fp_neg5_almost(const Fp<P>& a) {
    constexpr int L = P::L;
    constexpr uint64_t M = ((5ull << 58) + P::P[L - 1] - 1) / P::P[L - 1];
    static_assert(M < (1ull << 32), "reciprocal must fit 32 bits");
    const uint32_t k = (uint32_t)(((uint64_t)(a.l[L - 1] + 1u) * (uint32_t)M) >> 58) + 1u;
    Fp<P> r;
    int64_t cy = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        int64_t s = (int64_t)((uint64_t)k * P::P[i]) + cy;
        s -= (int64_t)(5ull * a.l[i]);                     // a may come from the lazy domain: its top limb can exceed 31 bits
        if (i < L - 1) { r.l[i] = (uint32_t)s & MASK29; cy = s >> 29; } else { r.l[i] = (uint32_t)s; }
    }
    return r;
}

// Montgomery square: cross products taken once, against the doubled LOWER-index limb (that one is always < 2^29, so the
// doubling cannot overflow even when the operand's top limb is wide).
template <class P, bool PIN = false>
ZK_HD Fp<P> fp_sqr_lazy(const Fp<P>& a) {
    constexpr int L = P::L, LR = P::LR;
    uint32_t m[LR], r[L], a2[L];
#pragma unroll
    for (int i = 0; i < L; i++) a2[i] = a.l[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < LR + L - 1; k++) {
        if (k <= 2 * L - 2) {
#pragma unroll
            for (int i = (k >= L ? k - L + 1 : 0); 2 * i < k; i++) ZK_MAD(acc, a2[i], a.l[k - i]);    // i < j = k - i
            if ((k & 1) == 0) ZK_MAD(acc, a.l[k / 2], a.l[k / 2]);
        }
        if (k < LR) {
#pragma unroll
            for (int i = (k >= L ? k - L + 1 : 0); i < k; i++) ZK_MAD(acc, m[i], P::P[k - i]);
            m[k] = fp_redc_column<P, PIN>(acc, k == 0, k + 1 == LR);
        } else {
#pragma unroll
            for (int i = k - L + 1; i < LR; i++) ZK_MAD(acc, m[i], P::P[k - i]);
            r[k - LR] = (uint32_t)acc & MASK29;
            acc >>= 29;
            ZK_PIN(acc);
        }
    }
    r[L - 1] = (uint32_t)acc;
    Fp<P> o;
#pragma unroll
    for (int i = 0; i < L; i++) o.l[i] = r[i];
    return o;
}

template <class P>
ZK_HD Fp<P> fp_sqr(const Fp<P>& a) {
    const Fp<P> t = fp_sqr_lazy<P>(a);
    return fp_reduce_once<P>(t.l);
}

// ---- the lazy domain (fields with LR > L: Fq) ----------------------------------------------------------------------------------
// Values are kept as a representative in [0, ~7 p]: limbs 0..L-2 below 2^29, the top limb as wide as it needs to be (< 2^31.6).
// Products (fp_*_lazy above) bring anything back to [0, p + eps), eps = 2^(29 (L - 1) + 6); sums and differences are single
// carry passes with a multiple of p added so that they stay non-negative -- no comparison, no select.
//   fp_sub_kp<K>(a, b)   = a + K p - b                 needs b <= K p
//   fp_x3_lazy(rr,ppp,qq)= rr + 4 p - ppp - 2 qq       needs ppp + 2 qq <= 4 p   (all three are products: < 3 p + 3 eps)
//   fp_canon(a)          = a mod p, fully reduced      needs a < 8 p
template <class P, int K>
ZK_HD Fp<P> fp_sub_kp(const Fp<P>& a, const Fp<P>& b) {
    constexpr int L = P::L;
    Fp<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
        const int32_t s = (int32_t)(a.l[i] + P::KP[K][i]) - (int32_t)b.l[i] + c;     // in (-2^29, 2^30 + 2^29)
        r.l[i] = (uint32_t)s & MASK29;
        c = s >> 29;
    }
    r.l[L - 1] = a.l[L - 1] + P::KP[K][L - 1] - b.l[L - 1] + (uint32_t)c;                // the true value is in [0, 2^32): wraps are harmless
    return r;
}

template <class P>
ZK_HD Fp<P> fp_x3_lazy(const Fp<P>& rr, const Fp<P>& ppp, const Fp<P>& qq) {
    constexpr int L = P::L;
    Fp<P> r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
        const int32_t s = (int32_t)(rr.l[i] + P::KP[4][i]) - (int32_t)ppp.l[i] - (int32_t)(qq.l[i] << 1) + c;   // in (-2^31, 2^31)
        r.l[i] = (uint32_t)s & MASK29;
        c = s >> 29;
    }
    r.l[L - 1] = rr.l[L - 1] + P::KP[4][L - 1] - ppp.l[L - 1] - (qq.l[L - 1] << 1) + (uint32_t)c;
    return r;
}

// t = a - k p if that is non-negative, else a (a has a possibly wide top limb; so has the result)
template <class P, int K>
ZK_HD Fp<P> fp_cond_sub_kp(const Fp<P>& a) {
    constexpr int L = P::L;
    uint32_t u[L];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
        const int32_t s = (int32_t)a.l[i] - (int32_t)P::KP[K][i] + c;
        u[i] = (uint32_t)s & MASK29;
        c = s >> 29;
    }
    const int64_t top = (int64_t)a.l[L - 1] - (int64_t)P::KP[K][L - 1] + c;
    u[L - 1] = (uint32_t)top;
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = top < 0 ? a.l[i] : u[i];
    return r;
}
template <class P>
ZK_HD Fp<P> fp_canon(const Fp<P>& a) {
    return fp_cond_sub_kp<P, 1>(fp_cond_sub_kp<P, 2>(fp_cond_sub_kp<P, 4>(a)));
}

// a^e for a plain-integer exponent given as 29-bit limbs (most significant limb last).
template <class P>
ZK_HD Fp<P> fp_pow_limbs(const Fp<P>& a, const uint32_t* e, int nlimbs) {
    Fp<P> r = fp_one<P>();
    bool started = false;
    for (int i = nlimbs - 1; i >= 0; i--) {
        for (int bit = 28; bit >= 0; bit--) {
            if (started) r = fp_sqr<P>(r);
            if ((e[i] >> bit) & 1) {
                r = started ? fp_mul<P>(r, a) : a;
                started = true;
            }
        }
    }
    return r;
}

// Internal-form inverse by Fermat (a != 0): a^(p-2).  Input/ouput internal form.
template <class P>
ZK_HD Fp<P> fp_inv(const Fp<P>& a) {
    uint32_t e[P::L];
#pragma unroll
    for (int i = 0; i < P::L; i++) e[i] = P::P_MINUS_2[i];
    return fp_pow_limbs<P>(a, e, P::L);
}

// ---- packing: W 32-bit words <-> L 29-bit limbs (pure bit re-slicing, no arithmetic) ----
template <class P>
ZK_HD Fp<P> fp_unpack(const uint32_t* w) {
    constexpr int L = P::L, W = P::W;
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < L; i++) {
        int bit = 29 * i;
        int wi = bit >> 5, sh = bit & 31;
        uint32_t v = (wi < W) ? (w[wi] >> sh) : 0u;
        if (sh > 3 && wi + 1 < W) v |= w[wi + 1] << (32 - sh);
        r.l[i] = v & MASK29;
    }
    return r;
}

template <class P>
ZK_HD void fp_pack(uint32_t* w, const Fp<P>& a) {
    constexpr int L = P::L, W = P::W;
#pragma unroll
    for (int j = 0; j < W; j++) {
        int bit = 32 * j;
        int i0 = bit / 29, o = bit - 29 * i0;
        uint32_t v = a.l[i0] >> o;
        if (i0 + 1 < L) v |= a.l[i0 + 1] << (29 - o);
        if (i0 + 2 < L && 58 - o < 32) v |= a.l[i0 + 2] << (58 - o);
        w[j] = v;
    }
}

// external (arkworks Montgomery, R = 2^(32W)) <-> internal / canonical
template <class P>
ZK_HD Fp<P> fp_ext_to_int(const Fp<P>& x) { return fp_mul<P>(x, fp_const<P>(P::EXT_TO_INT)); }
template <class P>
ZK_HD Fp<P> fp_int_to_ext(const Fp<P>& x) { return fp_mul<P>(x, fp_const<P>(P::INT_TO_EXT)); }
template <class P>
ZK_HD Fp<P> fp_ext_to_canon(const Fp<P>& x) { return fp_mul<P>(x, fp_const<P>(P::EXT_TO_CANON)); }
template <class P>
ZK_HD Fp<P> fp_canon_to_ext(const Fp<P>& x) { return fp_mul<P>(x, fp_const<P>(P::CANON_TO_EXT)); }
template <class P>
ZK_HD Fp<P> fp_canon_to_int(const Fp<P>& x) { return fp_mul<P>(x, fp_const<P>(P::RI2)); }
template <class P>
ZK_HD Fp<P> fp_int_to_canon(const Fp<P>& x) { return fp_mul<P>(x, fp_const<P>(P::RAW_ONE)); }

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// ---- Fq2 = Fq[u]/(u^2 + 5)  (ff/src/fields/models/quadratic_extension.rs:632-643, fields/fq2.rs:13) ----
struct Fq2 {
    Fq c0, c1;
};

// Uniform static interface so the curve code can be written once for G1 (Fq) and G2 (Fq2).
struct FqField {
    using T = Fq;
    static ZK_HD T zero() { return fp_zero<FqParams>(); }
    static ZK_HD T one() { return fp_one<FqParams>(); }
    static ZK_HD T add(const T& a, const T& b) { return fp_add<FqParams>(a, b); }
    static ZK_HD T sub(const T& a, const T& b) { return fp_sub<FqParams>(a, b); }
    static ZK_HD T dbl(const T& a) { return fp_dbl<FqParams>(a); }
    static ZK_HD T neg(const T& a) { return fp_neg<FqParams>(a); }
    static ZK_HD T mul(const T& a, const T& b) { return fp_mul<FqParams>(a, b); }
    static ZK_HD T sqr(const T& a) { return fp_sqr<FqParams>(a); }
    static ZK_HD T mulsub(const T& a, const T& b, const T& c, const T& d) {   // a b - c d, one Montgomery reduction
        return fp_mul2<FqParams>(a, b, fp_neg_lazy<FqParams>(c), d);
    }
    static ZK_HD T inv(const T& a) { return fp_inv<FqParams>(a); }
    static ZK_HD bool is_zero(const T& a) { return fp_is_zero<FqParams>(a); }
    static ZK_HD bool eq(const T& a, const T& b) { return fp_eq<FqParams>(a, b); }
    // ---- lazy domain (see fp29.cuh "the lazy domain"): representatives in [0, ~7 p], no conditional subtractions ----
    static ZK_HD T mul_l(const T& a, const T& b) { return fp_mul_lazy<FqParams, true>(a, b); }
    static ZK_HD T sqr_l(const T& a) { return fp_sqr_lazy<FqParams, true>(a); }
    template <int K> static ZK_HD T sub_kp(const T& a, const T& b) { return fp_sub_kp<FqParams, K>(a, b); }       // a + K p - b
    template <int K> static ZK_HD T kp_minus(const T& b) { return fp_sub_kp<FqParams, K>(fp_zero<FqParams>(), b); }   // K p - b
    static ZK_HD T x3_l(const T& rr, const T& ppp, const T& qq) { return fp_x3_lazy<FqParams>(rr, ppp, qq); }
    // r t - ppp y with one Montgomery reduction; ppp enters as 2 p - ppp in (p - eps, 2 p]
    static ZK_HD T mulsub_l(const T& r, const T& t, const T& ppp, const T& y) {
        return fp_mul2_lazy<FqParams, false, true>(r, t, fp_sub_kp<FqParams, 2>(fp_zero<FqParams>(), ppp), y);
    }
    static ZK_HD T canon(const T& a) { return fp_canon<FqParams>(a); }                    // a < 8 p -> a mod p
    static ZK_HD T canon1(const T& a) { return fp_cond_sub_kp<FqParams, 1>(a); }          // a < 2 p -> a mod p
    // can a be a multiple of p?  k p has the low limb k (p = 1 mod 2^29), so for a in (0, 8 p) only low limbs 1..7 qualify
    static ZK_HD bool maybe_multiple_of_p(const T& a) { return a.l[0] - 1u < 7u; }
    static ZK_HD T select(bool c, const T& a, const T& b) {  // c ? a : b
        T r;
#pragma unroll
        for (int i = 0; i < FqParams::L; i++) r.l[i] = c ? a.l[i] : b.l[i];
        return r;
    }
    static constexpr int WORDS = FqParams::W;  // packed 32-bit words per element
    static ZK_HD T load(const uint32_t* w) { return fp_unpack<FqParams>(w); }
    static ZK_HD void store(uint32_t* w, const T& a) { fp_pack<FqParams>(w, a); }
    static ZK_HD T ext_to_int(const T& a) { return fp_ext_to_int<FqParams>(a); }
    static ZK_HD T int_to_ext(const T& a) { return fp_int_to_ext<FqParams>(a); }
};

struct Fq2Field {
    using T = Fq2;
    using B = FqField;
    static ZK_HD T zero() { return T{B::zero(), B::zero()}; }
    static ZK_HD T one() { return T{B::one(), B::zero()}; }
    static ZK_HD T add(const T& a, const T& b) { return T{B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
    static ZK_HD T sub(const T& a, const T& b) { return T{B::sub(a.c0, b.c0), B::sub(a.c1, b.c1)}; }
    static ZK_HD T dbl(const T& a) { return T{B::dbl(a.c0), B::dbl(a.c1)}; }
    static ZK_HD T neg(const T& a) { return T{B::neg(a.c0), B::neg(a.c1)}; }
    static ZK_HD Fq mul5(const Fq& a) { Fq t = B::dbl(B::dbl(a)); return B::add(t, a); }
    static ZK_HD T mul(const T& a, const T& b) {
        // nonresidue -5: c0 = a0 b0 + (-5 a1) b1 ; c1 = a0 b1 + a1 b0, each a fused double product with one reduction
        // (same 1 014 mads as Karatsuba's three products, 4 instead of 8 field add/sub)
        Fq m5a1 = fp_neg5_almost<FqParams>(a.c1);
        return T{fp_mul2<FqParams>(a.c0, b.c0, m5a1, b.c1), fp_mul2<FqParams>(a.c0, b.c1, a.c1, b.c0)};
    }
    static ZK_HD T sqr(const T& a) {
        // c0 = a0^2 - 5 a1^2 (fused) ; c1 = 2 a0 a1
        Fq m5a1 = fp_neg5_almost<FqParams>(a.c1);
        return T{fp_mul2<FqParams>(a.c0, a.c0, m5a1, a.c1), B::dbl(B::mul(a.c0, a.c1))};
    }
    static ZK_HD T mulsub(const T& a, const T& b, const T& c, const T& d) { return sub(mul(a, b), mul(c, d)); }
    static ZK_HD T inv(const T& a) {
        // 1/(c0 + c1 u) = (c0 - c1 u) / (c0^2 + 5 c1^2)   (quadratic_extension.rs:309-325)
        Fq n = B::add(B::sqr(a.c0), mul5(B::sqr(a.c1)));
        Fq ni = B::inv(n);
        return T{B::mul(a.c0, ni), B::neg(B::mul(a.c1, ni))};
    }
    static ZK_HD bool is_zero(const T& a) { return B::is_zero(a.c0) && B::is_zero(a.c1); }
    static ZK_HD bool eq(const T& a, const T& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }
    static ZK_HD T select(bool c, const T& a, const T& b) {
        return T{B::select(c, a.c0, b.c0), B::select(c, a.c1, b.c1)};
    }
    static constexpr int WORDS = 2 * FqParams::W;
    static ZK_HD T load(const uint32_t* w) { return T{B::load(w), B::load(w + FqParams::W)}; }
    static ZK_HD void store(uint32_t* w, const T& a) { B::store(w, a.c0); B::store(w + FqParams::W, a.c1); }
    static ZK_HD T ext_to_int(const T& a) { return T{B::ext_to_int(a.c0), B::ext_to_int(a.c1)}; }
    static ZK_HD T int_to_ext(const T& a) { return T{B::int_to_ext(a.c0), B::int_to_ext(a.c1)}; }
};

}  // namespace zk

#undef ZK_MAD
#undef ZK_PIN
