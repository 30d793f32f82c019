// ec_dual.cuh -- one XYZZ point addition spread over TWO lane groups (gfx950): the latency form of ec.cuh::xyzz_add_lazy.
//
// Why.  A small MSM's bucket reduction (msm_reduce.cuh) is a chain of ~10 DEPENDENT additions on waves that sit alone on their SIMD,
// and a lone wave issues one multiply-add every ~10 clocks however many independent chains it carries (profiles/r3_ubench_chain.txt):
// an addition of 14 field products is ~20 us for G1 and ~30 us on a G2 lane pair whatever the chip has idle, and these chains are
// most of a small proof.  The 14 products of "add-2008-s" are seven independent PAIRS:
//     (U1 | U2)  (S1 | S2)  (PP | ZZ1 ZZ2)  (PPP | Q)  (R^2 | ZZZ1 ZZZ2)  (R (Q - X3) | PPP S1)  (PP ZZ12 | PPP ZZZ12)
// so two lane groups holding the same operands (every lane carries the whole point, or its component of it) each take one product
// of a step and swap the results through DPP: seven product times instead of fourteen, plus ~65 selects / moves per step.
//
// O (the operations of one group, msm.hip: a single lane over Fq; msm_g2pair.hip: a lane pair over Fq2):
//   T                     this lane's share of a field element
//   hi()                  is this lane in the second group?
//   mul(a, b)             lazy product (operands as in the lazy domain of fp29.cuh, result < p + eps)
//   swap(a)               the other group's value of a (DPP)
//   is_zero / maybe_multiple_of_p / select / sub_kp<K> / x3_l / canon / canon1   group-wide where a test is involved
//   dbl(X)                the complete doubling of a point (rare path: both groups compute it)
// Ranges: those of msm_g2pair.hip::add_p_lazy (x < 5p + eps, y < 3p + eps -- a difference of two products --, zz, zzz < p + eps;
// fully reduced coordinates included; infinity = all-zero words); the result is in the same ranges.
#pragma once
#include "ec.cuh"

namespace zk {

template <class O>
__device__ __forceinline__ void dual_mul2(bool hi, const typename O::T& a0, const typename O::T& b0, const typename O::T& a1,
                                          const typename O::T& b1, typename O::T& r0, typename O::T& r1) {
    const typename O::T m = O::mul(O::select(hi, a1, a0), O::select(hi, b1, b0));
    const typename O::T o = O::swap(m);
    r0 = O::select(hi, o, m);
    r1 = O::select(hi, m, o);
}

template <class O, class X>
__device__ __forceinline__ X xyzz_add_dual(const X& a, const X& b) {
    using T = typename O::T;
    if (O::is_zero(a.zz)) return b;
    if (O::is_zero(b.zz)) return a;
    const bool hi = O::hi();
    T u1, u2, s1, s2;
    dual_mul2<O>(hi, a.x, b.zz, b.x, a.zz, u1, u2);
    dual_mul2<O>(hi, a.y, b.zzz, b.y, a.zzz, s1, s2);
    const T p = O::template sub_kp<2>(u2, u1);
    const T r = O::template sub_kp<2>(s2, s1);
    if (O::maybe_multiple_of_p(p)) {
        if (O::is_zero(O::canon(p))) {
            if (O::is_zero(O::canon(r))) return O::dbl(X{O::canon(a.x), O::canon(a.y), O::canon1(a.zz), O::canon1(a.zzz)});
            return X{O::zero(), O::zero(), O::zero(), O::zero()};
        }
    }
    T pp, zz12, ppp, qq, rr, zzz12, t1, t2, zz3, zzz3;
    dual_mul2<O>(hi, p, p, a.zz, b.zz, pp, zz12);
    dual_mul2<O>(hi, pp, p, pp, u1, ppp, qq);
    dual_mul2<O>(hi, r, r, a.zzz, b.zzz, rr, zzz12);
    const T x3 = O::x3_l(rr, ppp, qq);
    dual_mul2<O>(hi, r, O::template sub_kp<6>(qq, x3), ppp, s1, t1, t2);
    const T y3 = O::template sub_kp<2>(t1, t2);
    dual_mul2<O>(hi, pp, zz12, ppp, zzz12, zz3, zzz3);
    return X{x3, y3, zz3, zzz3};
}

// acc + q, q affine and not infinity ("madd-2008-s", the form of the bucket-accumulation loops: ec.cuh::xyzz_madd_lazy,
// msm_g2pair.hip::madd_p_lazy): ten products as five pairs
//     (U2 | S2)  (PP | R^2)  (PPP | Q)  (R (Q - X3) | PPP Y1)  (PP ZZ1 | PPP ZZZ1)
// Ranges as madd_p_lazy: acc.x < 5p + eps, acc.y < 3p + eps (Y3 is a difference of two products), zz, zzz < p + eps; P and R may
// be up to 7p and 5p: their squares go through O::mul_wide.  Further members of O: one(), dbl_affine(A) (q's double, q.y reduced).
template <class O, class X, class A>
__device__ __forceinline__ X xyzz_madd_dual(const X& acc, const A& q) {
    using T = typename O::T;
    if (O::is_zero(acc.zz)) return X{q.x, q.y, O::one(), O::one()};
    const bool hi = O::hi();
    T u2, s2;
    dual_mul2<O>(hi, q.x, acc.zz, q.y, acc.zzz, u2, s2);
    const T p = O::template sub_kp<6>(u2, acc.x);
    const T r = O::template sub_kp<4>(s2, acc.y);
    if (O::maybe_multiple_of_p(p)) {
        if (O::is_zero(O::canon(p))) {
            if (O::is_zero(O::canon(r))) return O::dbl_affine(A{q.x, O::canon1(q.y)});
            return X{O::zero(), O::zero(), O::zero(), O::zero()};
        }
    }
    T pp, rr, ppp, qq, t1, t2, zz3, zzz3;
    {
        const T m = O::mul_wide(O::select(hi, r, p), O::select(hi, r, p));
        const T o = O::swap(m);
        pp = O::select(hi, o, m);
        rr = O::select(hi, m, o);
    }
    dual_mul2<O>(hi, pp, p, pp, acc.x, ppp, qq);
    const T x3 = O::x3_l(rr, ppp, qq);
    dual_mul2<O>(hi, r, O::template sub_kp<6>(qq, x3), ppp, acc.y, t1, t2);
    const T y3 = O::template sub_kp<2>(t1, t2);
    dual_mul2<O>(hi, pp, acc.zz, ppp, acc.zzz, zz3, zzz3);
    return X{x3, y3, zz3, zzz3};
}

}  // namespace zk
