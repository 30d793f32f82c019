// ec.cuh -- short-Weierstrass (a = 0) group arithmetic for BLS12-377 G1 (over Fq) and G2 (over Fq2).
//
// Replaces (reference): arkworks/algebra/ec/src/models/short_weierstrass_jacobian.rs
//   add_assign_mixed :628-693, double_in_place :557-623, add_assign :721-784,
//   From<Projective> for Affine :823-845.
// The reference accumulates in Jacobian coordinates; here the accumulators are extended Jacobian
// "XYZZ" (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed add is 8M+2S instead of 7M+4S and, unlike the
// Jacobian madd, needs no field doubling chains.  The group law is the same, so every result is the
// same group element; bit-exactness is established on the affine (canonical) form.
//
// Coordinates are in the device "internal" Montgomery form (fp29.cuh).
// Point at infinity: affine (0,0) (not on either curve: b != 0); XYZZ with ZZ = 0.
#pragma once
#include "fp29.cuh"

namespace zk {

template <class F>
struct Affine {
    typename F::T x, y;
};

template <class F>
struct XYZZ {
    typename F::T x, y, zz, zzz;
};

template <class F>
ZK_HD bool aff_is_inf(const Affine<F>& p) { return F::is_zero(p.x) && F::is_zero(p.y); }

template <class F>
ZK_HD Affine<F> aff_inf() { return Affine<F>{F::zero(), F::zero()}; }

template <class F>
ZK_HD Affine<F> aff_neg(const Affine<F>& p) { return Affine<F>{p.x, F::neg(p.y)}; }  // -(0,0) = (0,0)

template <class F>
ZK_HD XYZZ<F> xyzz_inf() { return XYZZ<F>{F::zero(), F::zero(), F::zero(), F::zero()}; }

template <class F>
ZK_HD bool xyzz_is_inf(const XYZZ<F>& p) { return F::is_zero(p.zz); }

template <class F>
ZK_HD XYZZ<F> xyzz_from_affine(const Affine<F>& p) {
    if (aff_is_inf<F>(p)) return xyzz_inf<F>();
    return XYZZ<F>{p.x, p.y, F::one(), F::one()};
}

template <class F>
ZK_HD XYZZ<F> xyzz_neg(const XYZZ<F>& p) { return XYZZ<F>{p.x, F::neg(p.y), p.zz, p.zzz}; }

// 2*(x,y) for an affine point that is not infinity ("mdbl-2008-s-1", a = 0).
template <class F>
ZK_HD XYZZ<F> xyzz_dbl_affine(const Affine<F>& p) {
    using T = typename F::T;
    T u = F::dbl(p.y);
    if (F::is_zero(u)) return xyzz_inf<F>();  // order-2 point (none in the prime-order subgroups)
    T v = F::sqr(u);
    T w = F::mul(u, v);
    T s = F::mul(p.x, v);
    T xx = F::sqr(p.x);
    T m = F::add(F::dbl(xx), xx);
    T x3 = F::sub(F::sqr(m), F::dbl(s));
    T y3 = F::sub(F::mul(m, F::sub(s, x3)), F::mul(w, p.y));
    return XYZZ<F>{x3, y3, v, w};
}

// 2*P ("dbl-2008-s-1", a = 0).
template <class F>
ZK_HD XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
    using T = typename F::T;
    if (xyzz_is_inf<F>(p)) return p;
    T u = F::dbl(p.y);
    if (F::is_zero(u)) return xyzz_inf<F>();
    T v = F::sqr(u);
    T w = F::mul(u, v);
    T s = F::mul(p.x, v);
    T xx = F::sqr(p.x);
    T m = F::add(F::dbl(xx), xx);
    T x3 = F::sub(F::sqr(m), F::dbl(s));
    T y3 = F::sub(F::mul(m, F::sub(s, x3)), F::mul(w, p.y));
    return XYZZ<F>{x3, y3, F::mul(v, p.zz), F::mul(w, p.zzz)};
}

// acc + q, q affine ("madd-2008-s"), complete: handles acc = inf, q = inf, q = +-acc.
template <class F>
ZK_HD XYZZ<F> xyzz_madd(const XYZZ<F>& acc, const Affine<F>& q) {
    using T = typename F::T;
    if (aff_is_inf<F>(q)) return acc;
    if (xyzz_is_inf<F>(acc)) return XYZZ<F>{q.x, q.y, F::one(), F::one()};
    T u2 = F::mul(q.x, acc.zz);
    T s2 = F::mul(q.y, acc.zzz);
    T p = F::sub(u2, acc.x);
    T r = F::sub(s2, acc.y);
    if (F::is_zero(p)) {
        if (F::is_zero(r)) return xyzz_dbl_affine<F>(q);
        return xyzz_inf<F>();
    }
    // pp and ppp are the LEFT operands of all their products: the Fq2 product prepares -5 * (left).c1 once per distinct
    // left operand (common subexpression after inlining), which is a fifth of its cost
    T pp = F::sqr(p);
    T ppp = F::mul(pp, p);
    T qq = F::mul(pp, acc.x);
    T x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qq));
    T y3 = F::mulsub(r, F::sub(qq, x3), ppp, acc.y);
    return XYZZ<F>{x3, y3, F::mul(pp, acc.zz), F::mul(ppp, acc.zzz)};
}

// acc + q in the LAZY domain (fp29.cuh): the same formulas, no conditional subtraction anywhere on the main path.
//   acc: x < 5 p + eps, y < p + eps, zz, zzz < p + eps (eps = 2^354), infinity = all-zero words;   q: affine, x fully
//   reduced, y fully reduced or p - y (<= p), not infinity (the caller tests that on the table's own words).
// Ranges (every product lands in [0, p + eps)):  P = u2 + 6p - X1 < 7p + eps;  R = s2 + 2p - Y1 < 3p + eps;
//   X3 = R^2 + 4p - PPP - 2Q in (p - 3 eps, 5p + eps);  T = Q + 6p - X3 < 7p + eps;  Y3 = (R T + (2p - PPP) Y1) / RI + < p.
// The equal-x case (doubling / cancellation) is detected exactly: P = 0 mod p iff P is one of p .. 7p, whose low limbs are
// 1 .. 7; only then (2^-26 of random additions) are P and R reduced and compared.
template <class F>
ZK_HD XYZZ<F> xyzz_madd_lazy(const XYZZ<F>& acc, const Affine<F>& q) {
    using T = typename F::T;
    if (F::is_zero(acc.zz)) return XYZZ<F>{q.x, q.y, F::one(), F::one()};
    const T u2 = F::mul_l(q.x, acc.zz);
    const T s2 = F::mul_l(q.y, acc.zzz);
    const T p = F::template sub_kp<6>(u2, acc.x);
    const T r = F::template sub_kp<2>(s2, acc.y);
    if (F::maybe_multiple_of_p(p)) {
        if (F::is_zero(F::canon(p))) {
            if (F::is_zero(F::canon(r))) return xyzz_dbl_affine<F>(Affine<F>{q.x, F::canon1(q.y)});
            return xyzz_inf<F>();
        }
    }
    const T pp = F::sqr_l(p);
    const T ppp = F::mul_l(pp, p);
    const T qq = F::mul_l(pp, acc.x);
    const T x3 = F::x3_l(F::sqr_l(r), ppp, qq);
    const T y3 = F::mulsub_l(r, F::template sub_kp<6>(qq, x3), ppp, acc.y);
    return XYZZ<F>{x3, y3, F::mul_l(pp, acc.zz), F::mul_l(ppp, acc.zzz)};
}
// back to fully reduced coordinates (end of a segment)
template <class F>
ZK_HD XYZZ<F> xyzz_canon_lazy(const XYZZ<F>& a) { return XYZZ<F>{F::canon(a.x), F::canon1(a.y), F::canon1(a.zz), F::canon1(a.zzz)}; }

// a + b ("add-2008-s") in the LAZY domain: what the bucket reduction's running sums use (msm.hip: k_reduce, k_bitsum, k_fold).
//   a, b: coordinates in the ranges xyzz_madd_lazy leaves (x < 5p + eps, y, zz, zzz < p + eps; fully reduced ones included),
//   infinity = all-zero words.  The result is in the same ranges.
// Ranges: U1, U2, S1, S2 < p + eps;  P = U2 + 2p - U1 in (p - eps, 3p + eps), R likewise;  X3, T, Y3 as in xyzz_madd_lazy.
// The equal-x case is found exactly (P = 0 mod p iff P is p, 2p or 3p: low limbs 1 .. 3).
template <class F>
ZK_HD XYZZ<F> xyzz_add_lazy(const XYZZ<F>& a, const XYZZ<F>& b) {
    using T = typename F::T;
    if (F::is_zero(a.zz)) return b;
    if (F::is_zero(b.zz)) return a;
    const T u1 = F::mul_l(a.x, b.zz);
    const T u2 = F::mul_l(b.x, a.zz);
    const T s1 = F::mul_l(a.y, b.zzz);
    const T s2 = F::mul_l(b.y, a.zzz);
    const T p = F::template sub_kp<2>(u2, u1);
    const T r = F::template sub_kp<2>(s2, s1);
    if (F::maybe_multiple_of_p(p)) {
        if (F::is_zero(F::canon(p))) {
            if (F::is_zero(F::canon(r))) return xyzz_dbl<F>(xyzz_canon_lazy<F>(a));
            return xyzz_inf<F>();
        }
    }
    const T pp = F::sqr_l(p);
    const T ppp = F::mul_l(p, pp);
    const T qq = F::mul_l(u1, pp);
    const T x3 = F::x3_l(F::sqr_l(r), ppp, qq);
    const T y3 = F::mulsub_l(r, F::template sub_kp<6>(qq, x3), ppp, s1);
    return XYZZ<F>{x3, y3, F::mul_l(F::mul_l(a.zz, b.zz), pp), F::mul_l(F::mul_l(a.zzz, b.zzz), ppp)};
}
// x brought below p (the packed 12-word form holds 377 bits: y, zz, zzz < p + eps fit as they are, x < 5p + eps does not)
template <class F>
ZK_HD XYZZ<F> xyzz_packable_lazy(const XYZZ<F>& a) { return XYZZ<F>{F::canon(a.x), a.y, a.zz, a.zzz}; }

// a + b ("add-2008-s"), complete.
template <class F>
ZK_HD XYZZ<F> xyzz_add(const XYZZ<F>& a, const XYZZ<F>& b) {
    using T = typename F::T;
    if (xyzz_is_inf<F>(a)) return b;
    if (xyzz_is_inf<F>(b)) return a;
    T u1 = F::mul(a.x, b.zz);
    T u2 = F::mul(b.x, a.zz);
    T s1 = F::mul(a.y, b.zzz);
    T s2 = F::mul(b.y, a.zzz);
    T p = F::sub(u2, u1);
    T r = F::sub(s2, s1);
    if (F::is_zero(p)) {
        if (F::is_zero(r)) return xyzz_dbl<F>(a);
        return xyzz_inf<F>();
    }
    T pp = F::sqr(p);
    T ppp = F::mul(p, pp);
    T qq = F::mul(u1, pp);
    T x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qq));
    T y3 = F::mulsub(r, F::sub(qq, x3), s1, ppp);
    return XYZZ<F>{x3, y3, F::mul(F::mul(a.zz, b.zz), pp), F::mul(F::mul(a.zzz, b.zzz), ppp)};
}

// Canonical affine form (one field inversion).
template <class F>
ZK_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
    using T = typename F::T;
    if (xyzz_is_inf<F>(p)) return aff_inf<F>();
    // 1/ZZZ, then 1/ZZ = ZZZ^-1 * ZZZ * ZZ^-1 ... cheaper: zi3 = 1/ZZZ ; zi2 = (zi3 * ZZ)^2 since ZZ^3 = ZZZ^2
    T zi3 = F::inv(p.zzz);
    T zi = F::mul(zi3, p.zz);  // = ZZ/ZZZ = 1/Z
    T zi2 = F::sqr(zi);        // = 1/ZZ
    return Affine<F>{F::mul(p.x, zi2), F::mul(p.y, zi3)};
}

// k * P by double-and-add, MSB first, k given as `nwords` 32-bit words (plain integer).
template <class F>
ZK_HD XYZZ<F> xyzz_scalar_mul(const Affine<F>& p, const uint32_t* k, int nwords) {
    XYZZ<F> r = xyzz_inf<F>();
    for (int i = nwords - 1; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            r = xyzz_dbl<F>(r);
            if ((k[i] >> b) & 1) r = xyzz_madd<F>(r, p);
        }
    return r;
}

// ---- packed memory formats (internal Montgomery form, 32-bit words) ----
//  affine: x | y                        (2 * F::WORDS words)  G1: 96 B, G2: 192 B
//  xyzz  : x | y | zz | zzz             (4 * F::WORDS words)  G1: 192 B, G2: 384 B
template <class F>
ZK_HD Affine<F> aff_load(const uint32_t* w) { return Affine<F>{F::load(w), F::load(w + F::WORDS)}; }
template <class F>
ZK_HD void aff_store(uint32_t* w, const Affine<F>& p) { F::store(w, p.x); F::store(w + F::WORDS, p.y); }
template <class F>
ZK_HD XYZZ<F> xyzz_load(const uint32_t* w) {
    return XYZZ<F>{F::load(w), F::load(w + F::WORDS), F::load(w + 2 * F::WORDS), F::load(w + 3 * F::WORDS)};
}
template <class F>
ZK_HD void xyzz_store(uint32_t* w, const XYZZ<F>& p) {
    F::store(w, p.x); F::store(w + F::WORDS, p.y); F::store(w + 2 * F::WORDS, p.zz); F::store(w + 3 * F::WORDS, p.zzz);
}

using G1Field = FqField;
using G2Field = Fq2Field;

}  // namespace zk
