// frlazy.cuh -- the lazy Fr domain of the NTT butterflies.
//
// The butterflies of Radix2EvaluationDomain::{fft,ifft}_in_place (arkworks/algebra/poly/src/domain/radix2/fft.rs:185-307)
// are x' = x + y, y' = (x - y) w with every sum, difference and product brought back below r.  r = 0.002 * 2^261, so nine
// 29-bit limbs have 8.8 bits of head room (2^261 / r = 438.8) and almost all of those reductions can go:
//   * sums are nine limb-wise additions, no carry pass (limbs may grow to 2^31.4: the products take any u32 limb as long as
//     a column of the product stays below 2^64 -- tests/test_abi.py::test_fr_lazy_domain recomputes the worst column);
//   * differences add a multiple of r written so that no limb can go negative (FrLazy::OFF*): nine add-subs, no borrow;
//   * a product by a table entry (< r) of anything below 2^261 lands below r (1 + a / 2^261) by itself: the Montgomery
//     reduction IS the range reduction, no conditional subtraction;
//   * the one output in four of a radix-4 butterfly that is a sum of sums passes no product: frl_reduce subtracts
//     q r with q estimated from the top limb, inside its carry pass (no comparison, no select).
// Canonical values (frl_canon) are produced once, where a transform hands its result back.
//
// Ranges (value / limbs), radix-4 DIF butterfly on inputs < 2.1 r with limbs < 2^29:
//   s0 = x0 + x2, s1 = x1 + x3                 < 4.2 r   / < 2^30
//   d0 = (x0 - x2 + 3r) wa, d1 = (...) wb      < 1.02 r  / < 2^29        (operand < 5.1 r, limbs < 2^30.6)
//   y0 = reduce(s0 + s1)                        < 1.13 r  / < 2^29        (operand < 8.4 r, limbs < 2^31)
//   y1 = (s0 - s1 + 5r) wc                      < 1.03 r  / < 2^29        (operand < 9.2 r, limbs < 2^31.33)
//   y2 = norm(d0 + d1)                          < 2.04 r  / < 2^29
//   y3 = (d0 - d1 + 2r) wc                      < 1.01 r  / < 2^29        (operand < 3.1 r, limbs < 2^30.6)
// so every output is again < 2.1 r with limbs < 2^29.  In the last stage of a pass wc = 1: all four outputs stay as they are
// (wide limbs, < 9.2 r) and go straight into the product that follows (the inter-pass twiddle, the post-scale) or into
// frl_canon.
#pragma once
#include "fp29.cuh"

namespace zk {

ZK_HD Fr frl_add(const Fr& a, const Fr& b) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a + K r - b, limb by limb: OFF holds K r with limbs 0..7 at least as large as any limb b can have there
template <int K>
ZK_HD Fr frl_sub(const Fr& a, const Fr& b) {
    static_assert(K == 2 || K == 3 || K == 5, "offsets generated: 2 r, 3 r (b normalised), 5 r (b's limbs < 2^30)");
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t o = K == 2 ? FrLazy::OFF2[i] : (K == 3 ? FrLazy::OFF3[i] : FrLazy::OFF5[i]);
        r.l[i] = a.l[i] + o - b.l[i];
    }
    return r;
}

// carry pass only: limbs < 2^29 again, same value (which must be < 2^261)
ZK_HD Fr frl_norm(const Fr& a) {
    Fr r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t s = a.l[i] + c;       // a.l[i] <= 2^32 - 2^4 (four normalised limbs summed)
        if (i < 8) { r.l[i] = s & MASK29; c = s >> 29; } else { r.l[i] = s; }
    }
    return r;
}

// Any a < 2^261 with u32 limbs (each <= 2^32 - 2^10) -> the representative a - q r in [0, 1.13 r), limbs < 2^29.
// q = floor(t MQ / 2^32) with t = a_8 + (a_7 >> 29) (the value's bits from 232 up, short by at most 2) and MQ = floor(2^264 / r):
//   q <= a / r, and q > a / r - 1 - t / 2^32 - 2 * 2^232 / r > a / r - 1.1251.
// a - q r is taken as a + q (2^261 - r): every term is non-negative, and the q 2^261 drops off the top of the carry pass.
ZK_HD Fr frl_reduce(const Fr& a) {
    const uint32_t t = a.l[8] + (a.l[7] >> 29);
    const uint32_t q = (uint32_t)(((uint64_t)t * FrLazy::MQ) >> 32);
    Fr r;
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc += (uint64_t)q * FrLazy::RC[i] + a.l[i];
        r.l[i] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    return r;
}

ZK_HD Fr frl_mul(const Fr& a, const Fr& w) { return fp_mul_lazy<FrParams>(a, w); }      // w < r (a table entry), a < 2^261

// a < 2^261, any limbs -> a mod r
ZK_HD Fr frl_canon(const Fr& a) {
    const Fr t = frl_reduce(a);
    return fp_reduce_once<FrParams>(t.l);
}

// Two DIF levels on (x0, x1, x2, x3) = the elements at m, m + g/2, m + g, m + 3g/2 (g = the gap of the first level):
// wa, wb = the first level's twiddles of the pairs (x0, x2) and (x1, x3) (wb = wa * w^(N/4)), wc = the second level's twiddle of
// both of its pairs (wa^2).  MUL_B false: the second level's twiddle is 1 and all four outputs are left as wide sums (a
// product or frl_canon follows).  ntt.hip::radix4_item is this sequence with its loads and stores placed in between (register
// pressure); this form is what the host-side test drives (hostapi.hip::zk_fr_lazy_raw).
template <bool MUL_B>
ZK_HD void frl_radix4(Fr& x0, Fr& x1, Fr& x2, Fr& x3, const Fr& wa, const Fr& wb, const Fr& wc) {
    const Fr s0 = frl_add(x0, x2), s1 = frl_add(x1, x3);
    const Fr d0 = frl_mul(frl_sub<3>(x0, x2), wa);
    const Fr d1 = frl_mul(frl_sub<3>(x1, x3), wb);
    x0 = frl_reduce(frl_add(s0, s1));
    x2 = frl_norm(frl_add(d0, d1));
    if (MUL_B) {
        x1 = frl_mul(frl_sub<5>(s0, s1), wc);
            x3 = frl_mul(frl_sub<2>(d0, d1), wc);
    } else {
        x1 = frl_sub<5>(s0, s1);
        x3 = frl_sub<2>(d0, d1);
    }
}

// One DIF level on a pair (the odd level of a pass with an odd number of levels): both outputs < 1.13 r, limbs < 2^29.
ZK_HD void frl_radix2(Fr& x0, Fr& x1, const Fr& w) {
    const Fr s = frl_add(x0, x1);
    x1 = frl_mul(frl_sub<3>(x0, x1), w);
    x0 = frl_reduce(s);
}

// 32 v mod r for ANY v below 2^256 with limbs below 2^29: the fix-up of a product of two elements in the reference's form.  The
// device's Montgomery product divides by RI = 2^261, so mmul(a 2^256, b 2^256) = a b 2^251; the missing 2^5 used to be a second
// Montgomery product (153 multiply-adds) and is ~90 plain instructions here: t = v << 5, q = floor((t >> 240) / ((r >> 240) + 1))
// by a reciprocal (exact for every 21-bit numerator: tests/test_abi.py), so q <= t / r < q + 1.01 and t - q r < 1.01 r, then
// one conditional subtraction.  Fully reduced result.
ZK_HD Fr fr_mul32(const Fr& v) {
    constexpr int L = FrParams::L;
    static_assert(L == 9, "nine 29-bit limbs");
    uint32_t t[L], s[L];
    t[0] = (v.l[0] << 5) & MASK29;
#pragma unroll
    for (int i = 1; i < L - 1; i++) t[i] = ((v.l[i] << 5) | (v.l[i - 1] >> 24)) & MASK29;
    t[L - 1] = (v.l[L - 1] << 5) | (v.l[L - 2] >> 24);                   // bits 232 .. 260 of t
    constexpr uint32_t R_HI = (FrParams::P[L - 1] >> 8) + 1;             // (r >> 240) + 1
    constexpr uint32_t M = (uint32_t)((1ull << 32) / R_HI) + 1;
    const uint32_t q = (uint32_t)(((uint64_t)(t[L - 1] >> 8) * M) >> 32);
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int64_t x = (int64_t)t[i] - (int64_t)((uint64_t)q * FrParams::P[i]) + c;
        if (i < L - 1) { s[i] = (uint32_t)x & MASK29; c = x >> 29; } else { s[i] = (uint32_t)x; }
    }
    return fp_reduce_once<FrParams>(s);
}

}  // namespace zk
