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
// An element is L limbs of 29 bits, value = sum l[i] * 2^(29 i), always fully reduced (< p).
// mmul(a, b) = a*b / 2^(29 L) mod p.   "internal" form of x is x * 2^(29 L) mod p.
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

// Montgomery product a*b/2^(29L) mod p, finely integrated product scanning, one 64-bit accumulator.
template <class P>
ZK_HD Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
    constexpr int L = P::L;
    uint32_t m[L], r[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
        m[k] = ((uint32_t)acc * P::INV) & MASK29;
        acc += (uint64_t)m[k] * P::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)m[i] * P::P[k - i];
        }
        r[k - L] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r[L - 1] = (uint32_t)acc;
    return fp_reduce_once<P>(r);
}

// (a b + c d) / 2^(29L) mod p with ONE Montgomery reduction: the two limb products share the column accumulator
// (39 products < 2^58 per column, < 2^64) and the reduction terms.  3 L^2 mads instead of 4 L^2 for two separate
// products, and no field addition afterwards.  The value before the final subtraction is < (2 p^2 + 2^(29L) p) / 2^(29L)
// < 2.68 p for BLS12-377's q (0.84 * 2^377), hence two conditional subtractions.  Used by the Fq2 product: measured
// (tools/ubench_fq2.hip) 21.9 G Fq2-mul/s against 18.5 for Karatsuba with three separate products.
template <class P>
ZK_HD Fp<P> fp_mul2(const Fp<P>& a, const Fp<P>& b, const Fp<P>& c, const Fp<P>& d) {
    constexpr int L = P::L;
    uint32_t m[L], r[L];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
        m[k] = ((uint32_t)acc * P::INV) & MASK29;
        acc += (uint64_t)m[k] * P::P[0];
        acc >>= 29;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = k - L + 1; i < L; i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
            acc += (uint64_t)m[i] * P::P[k - i];
        }
        r[k - L] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r[L - 1] = (uint32_t)acc;
    // first subtraction: the difference can still be >= p, so its top limb is kept unmasked (it may exceed 29 bits)
    uint32_t u[L];
    int32_t cy = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
        int32_t s = (int32_t)r[i] - (int32_t)P::P[i] + cy;
        u[i] = (uint32_t)s & MASK29;
        cy = s >> 29;
    }
    const int32_t top = (int32_t)r[L - 1] - (int32_t)P::P[L - 1] + cy;
    u[L - 1] = (uint32_t)top;
    const bool below = top < 0;        // r < p
    uint32_t t2[L];
#pragma unroll
    for (int i = 0; i < L; i++) t2[i] = below ? r[i] : u[i];
    return fp_reduce_once<P>(t2);      // now < 2p
}

// -5 a mod p in ONE carry pass, almost reduced: returns V = k p - 5 a with k = floor(upper estimate of 5 a / p) + 1, so
// V = -5 a (mod p) and 0 < V <= p (1 + 3e-8); all limbs < 2^29 (p (1 + 3e-8) < 2^(29 L) for BLS12-377's q = 0.84 * 2^377).
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
        s += (int64_t)(int32_t)a.l[i] * (int64_t)(-5);
        if (i < L - 1) { r.l[i] = (uint32_t)s & MASK29; cy = s >> 29; } else { r.l[i] = (uint32_t)s; }
    }
    return r;
}

// Montgomery square: cross products taken once against the doubled operand.
template <class P>
ZK_HD Fp<P> fp_sqr(const Fp<P>& a) {
    constexpr int L = P::L;
    uint32_t m[L], r[L], a2[L];
#pragma unroll
    for (int i = 0; i < L; i++) a2[i] = a.l[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; k++) {
#pragma unroll
        for (int i = 0; i < L; i++) {
            int j = k - i;
            if (j < 0 || j >= L) continue;
            if (i < j) acc += (uint64_t)a.l[i] * a2[j];
            else if (i == j) acc += (uint64_t)a.l[i] * a.l[i];
        }
        if (k < L) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P::P[k - i];
            m[k] = ((uint32_t)acc * P::INV) & MASK29;
            acc += (uint64_t)m[k] * P::P[0];
        } else {
#pragma unroll
            for (int i = k - L + 1; i < L; i++) acc += (uint64_t)m[i] * P::P[k - i];
            r[k - L] = (uint32_t)acc & MASK29;
        }
        acc >>= 29;
    }
    r[L - 1] = (uint32_t)acc;
    return fp_reduce_once<P>(r);
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
