// hostapi.hip -- host-side group / field helpers of the C ABI (O(1) work per proof).
// Replaces, for the glue of create_proof and the MPC group Beaver steps:
//   GroupProjective::{add_assign, neg, mul} (ec/src/models/short_weierstrass_jacobian.rs:700-806),
//   CanonicalSerialize for points (:847-883), Fp256 add/sub/mul, from_repr / into_repr.
#include "../../include/zkmpc_hip.h"
#include "ctx.hpp"
#include "hostgroup.hpp"
#include "hostfield64.hpp"
#include "frlazy.cuh"

using namespace zk;

// Group helpers run in the 64-bit-limb host field (hostfield64.hpp): the C-ABI structs already hold the
// reference's Montgomery form, which is that field's native form, so there is no conversion at all.
namespace {
template <class H>
int add_t(const uint64_t* a, const uint64_t* b, uint64_t* out) {
    if (!a || !b || !out) return ZK_ERR_ARG;
    host64_write_projective<H>(xyzz_to_affine<H>(xyzz_add<H>(host64_proj_from_abi<H>(a), host64_proj_from_abi<H>(b))), out);
    return ZK_OK;
}
template <class H>
int mul_t(const uint64_t* a, const zk_fr* k, uint64_t* out) {
    if (!a || !k || !out) return ZK_ERR_ARG;
    uint32_t kw[8];
    fr_abi_to_canon_words(k->l, kw);
    host64_write_projective<H>(xyzz_to_affine<H>(host64_scalar_mul<H>(host64_proj_from_abi<H>(a), kw)), out);
    return ZK_OK;
}
template <class H>
Affine<H> aff_from_abi64(const uint64_t* p) {
    constexpr int FE = H::WORDS / 2;
    uint32_t w[H::WORDS];
    bool zero = true;
    for (int i = 0; i < 2 * FE; i++) zero = zero && p[i] == 0;
    if (zero) return aff_inf<H>();
    auto get = [&](const uint64_t* src) {
        for (int i = 0; i < FE; i++) { w[2 * i] = (uint32_t)src[i]; w[2 * i + 1] = (uint32_t)(src[i] >> 32); }
        return H::load(w);
    };
    return Affine<H>{get(p), get(p + FE)};
}
}  // namespace

extern "C" int zk_g1_add(const zk_g1_projective* a, const zk_g1_projective* b, zk_g1_projective* out) {
    ZK_API_BEGIN_NOCTX
    return add_t<Fq64Field>((const uint64_t*)a, (const uint64_t*)b, (uint64_t*)out);
    ZK_API_END
}
extern "C" int zk_g2_add(const zk_g2_projective* a, const zk_g2_projective* b, zk_g2_projective* out) {
    ZK_API_BEGIN_NOCTX
    return add_t<Fq264Field>((const uint64_t*)a, (const uint64_t*)b, (uint64_t*)out);
    ZK_API_END
}
extern "C" int zk_g1_neg(const zk_g1_projective* a, zk_g1_projective* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !out) return ZK_ERR_ARG;
    host64_write_projective<Fq64Field>(xyzz_to_affine<Fq64Field>(xyzz_neg<Fq64Field>(host64_proj_from_abi<Fq64Field>((const uint64_t*)a))), (uint64_t*)out);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_g2_neg(const zk_g2_projective* a, zk_g2_projective* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !out) return ZK_ERR_ARG;
    host64_write_projective<Fq264Field>(xyzz_to_affine<Fq264Field>(xyzz_neg<Fq264Field>(host64_proj_from_abi<Fq264Field>((const uint64_t*)a))), (uint64_t*)out);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_g1_mul(const zk_g1_projective* a, const zk_fr* k, zk_g1_projective* out) {
    ZK_API_BEGIN_NOCTX
    return mul_t<Fq64Field>((const uint64_t*)a, k, (uint64_t*)out);
    ZK_API_END
}
extern "C" int zk_diag_g1_mul_glv(const zk_g1_projective* a, const zk_fr* k, zk_g1_projective* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !k || !out) return ZK_ERR_ARG;
    using H = Fq64Field;
    uint32_t kw[8];
    fr_abi_to_canon_words(k->l, kw);
    host64_write_projective<H>(xyzz_to_affine<H>(host64_scalar_mul_glv(host64_proj_from_abi<H>((const uint64_t*)a), kw)), (uint64_t*)out);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_g2_mul(const zk_g2_projective* a, const zk_fr* k, zk_g2_projective* out) {
    ZK_API_BEGIN_NOCTX
    return mul_t<Fq264Field>((const uint64_t*)a, k, (uint64_t*)out);
    ZK_API_END
}
extern "C" int zk_g1_from_affine(const zk_g1_affine* a, zk_g1_projective* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !out) return ZK_ERR_ARG;
    host64_write_projective<Fq64Field>(aff_from_abi64<Fq64Field>((const uint64_t*)a), (uint64_t*)out);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_g2_from_affine(const zk_g2_affine* a, zk_g2_projective* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !out) return ZK_ERR_ARG;
    host64_write_projective<Fq264Field>(aff_from_abi64<Fq264Field>((const uint64_t*)a), (uint64_t*)out);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_g1_serialize(const zk_g1_projective* a, uint8_t out[48]) {
    ZK_API_BEGIN_NOCTX
    if (!a || !out) return ZK_ERR_ARG;
    g1_serialize(aff_from_host64<G1Field>(xyzz_to_affine<Fq64Field>(host64_proj_from_abi<Fq64Field>((const uint64_t*)a))), out);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_g2_serialize(const zk_g2_projective* a, uint8_t out[96]) {
    ZK_API_BEGIN_NOCTX
    if (!a || !out) return ZK_ERR_ARG;
    g2_serialize(aff_from_host64<G2Field>(xyzz_to_affine<Fq264Field>(host64_proj_from_abi<Fq264Field>((const uint64_t*)a))), out);
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_fr_add(const zk_fr* a, const zk_fr* b, zk_fr* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !b || !out) return ZK_ERR_ARG;
    host_store_ext<FrParams>(out->l, fp_add<FrParams>(host_load_ext<FrParams>(a->l), host_load_ext<FrParams>(b->l)));
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_fr_sub(const zk_fr* a, const zk_fr* b, zk_fr* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !b || !out) return ZK_ERR_ARG;
    host_store_ext<FrParams>(out->l, fp_sub<FrParams>(host_load_ext<FrParams>(a->l), host_load_ext<FrParams>(b->l)));
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_fr_mul(const zk_fr* a, const zk_fr* b, zk_fr* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !b || !out) return ZK_ERR_ARG;
    Fr t = fp_mul<FrParams>(host_load_ext<FrParams>(a->l), host_load_ext<FrParams>(b->l));
    host_store_ext<FrParams>(out->l, fp_mul<FrParams>(t, fp_const<FrParams>(FrParams::EXT_TO_INT)));
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_fr_inverse(const zk_fr* a, zk_fr* out) {
    ZK_API_BEGIN_NOCTX   // Field::inverse; zero has no inverse (macros.rs:389-443)
    if (!a || !out) return ZK_ERR_ARG;
    Fr x = fp_ext_to_int<FrParams>(host_load_ext<FrParams>(a->l));
    if (fp_is_zero<FrParams>(x)) return ZK_ERR_ARG;
    host_store_ext<FrParams>(out->l, fp_int_to_ext<FrParams>(fp_inv<FrParams>(x)));
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_fr_pow(const zk_fr* a, uint64_t e, zk_fr* out) {
    ZK_API_BEGIN_NOCTX   // Field::pow for a 64-bit exponent
    if (!a || !out) return ZK_ERR_ARG;
    Fr x = fp_ext_to_int<FrParams>(host_load_ext<FrParams>(a->l)), r = fp_one<FrParams>();
    for (int b = 63; b >= 0; b--) {
        r = fp_sqr<FrParams>(r);
        if ((e >> b) & 1) r = fp_mul<FrParams>(r, x);
    }
    host_store_ext<FrParams>(out->l, fp_int_to_ext<FrParams>(r));
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_fr_from_canonical(const uint64_t canon[4], zk_fr* out) {
    ZK_API_BEGIN_NOCTX
    if (!canon || !out) return ZK_ERR_ARG;
    host_store_ext<FrParams>(out->l, fp_canon_to_ext<FrParams>(host_load_ext<FrParams>(canon)));
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_fr_to_canonical(const zk_fr* a, uint64_t canon[4]) {
    ZK_API_BEGIN_NOCTX
    if (!a || !canon) return ZK_ERR_ARG;
    host_store_ext<FrParams>(canon, fp_ext_to_canon<FrParams>(host_load_ext<FrParams>(a->l)));
    return ZK_OK;
    ZK_API_END
}

// Fp384 (Fq) host helpers in the DEVICE representation (29-bit limbs): the same add / sub / mul templates the
// kernels run, exposed so that their top-limb decision logic can be tested on crafted boundary values.
extern "C" int zk_fq_add(const zk_fq* a, const zk_fq* b, zk_fq* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !b || !out) return ZK_ERR_ARG;
    host_store_ext<FqParams>(out->l, fp_add<FqParams>(host_load_ext<FqParams>(a->l), host_load_ext<FqParams>(b->l)));
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_fq_sub(const zk_fq* a, const zk_fq* b, zk_fq* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !b || !out) return ZK_ERR_ARG;
    host_store_ext<FqParams>(out->l, fp_sub<FqParams>(host_load_ext<FqParams>(a->l), host_load_ext<FqParams>(b->l)));
    return ZK_OK;
    ZK_API_END
}
// a b + c d through the fused double product the Fq2 multiplication uses (fp29.cuh::fp_mul2), for boundary tests
extern "C" int zk_fq_mul2(const zk_fq* a, const zk_fq* b, const zk_fq* c, const zk_fq* d, zk_fq* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !b || !c || !d || !out) return ZK_ERR_ARG;
    Fq t = fp_mul2<FqParams>(host_load_ext<FqParams>(a->l), host_load_ext<FqParams>(b->l), host_load_ext<FqParams>(c->l),
                             host_load_ext<FqParams>(d->l));
    host_store_ext<FqParams>(out->l, fp_mul<FqParams>(t, fp_const<FqParams>(FqParams::EXT_TO_INT)));
    return ZK_OK;
    ZK_API_END
}
// raw limbs (29-bit, 13 words) of fp_neg5_almost on the raw limbs of a: V = k p - 5 a, for range tests
extern "C" int zk_fq_neg5_almost_raw(const uint32_t a13[13], uint32_t out13[13]) {
    ZK_API_BEGIN_NOCTX
    if (!a13 || !out13) return ZK_ERR_ARG;
    Fq a;
    for (int i = 0; i < 13; i++) a.l[i] = a13[i];
    Fq v = fp_neg5_almost<FqParams>(a);
    for (int i = 0; i < 13; i++) out13[i] = v.l[i];
    return ZK_OK;
    ZK_API_END
}
// Test hook for the lazy domain of Fq (fp29.cuh): raw 29-bit limbs in (13 words per element, the top one may be wide), raw
// limbs out.  op 0: mul_lazy(a, b)  1: sqr_lazy(a)  2: mul2_lazy(a, b, c, d)  3..5: sub_kp<2|4|6>(a, b)  6: x3_lazy(rr, ppp, qq)
// 7: canon(a)  8: kp_minus<1>(a)  9: neg5_almost(a)  10: xyzz_madd_lazy(acc[4], q[2]) -> 4 elements, then 4 more: its canon form
// 11: mul2_lazy with the split top column (all four operands wide)
// 12: xyzz_add_lazy(a[4], b[4]) -> 4 elements, then 4 more: its canon form
extern "C" int zk_fq_lazy_raw(int op, const uint32_t* in, uint32_t* out) {
    ZK_API_BEGIN_NOCTX
    if (!in || !out) return ZK_ERR_ARG;
    auto ld = [&](int k) { Fq a; for (int i = 0; i < 13; i++) a.l[i] = in[13 * k + i]; return a; };
    auto st = [&](int k, const Fq& a) { for (int i = 0; i < 13; i++) out[13 * k + i] = a.l[i]; };
    using F = FqField;
    switch (op) {
        case 0: st(0, F::mul_l(ld(0), ld(1))); break;
        case 1: st(0, F::sqr_l(ld(0))); break;
        case 2: st(0, fp_mul2_lazy<FqParams>(ld(0), ld(1), ld(2), ld(3))); break;
        case 11: st(0, fp_mul2_lazy<FqParams, true>(ld(0), ld(1), ld(2), ld(3))); break;
        case 3: st(0, F::sub_kp<2>(ld(0), ld(1))); break;
        case 4: st(0, F::sub_kp<4>(ld(0), ld(1))); break;
        case 5: st(0, F::sub_kp<6>(ld(0), ld(1))); break;
        case 6: st(0, F::x3_l(ld(0), ld(1), ld(2))); break;
        case 7: st(0, F::canon(ld(0))); break;
        case 8: st(0, F::kp_minus<1>(ld(0))); break;
        case 9: st(0, fp_neg5_almost<FqParams>(ld(0))); break;
        case 10: {
            XYZZ<F> acc{ld(0), ld(1), ld(2), ld(3)};
            Affine<F> q{ld(4), ld(5)};
            XYZZ<F> r = xyzz_madd_lazy<F>(acc, q);
            st(0, r.x); st(1, r.y); st(2, r.zz); st(3, r.zzz);
            XYZZ<F> c = xyzz_canon_lazy<F>(r);
            st(4, c.x); st(5, c.y); st(6, c.zz); st(7, c.zzz);
            break;
        }
        case 12: {
            XYZZ<F> a{ld(0), ld(1), ld(2), ld(3)}, b{ld(4), ld(5), ld(6), ld(7)};
            XYZZ<F> r = xyzz_add_lazy<F>(a, b);
            st(0, r.x); st(1, r.y); st(2, r.zz); st(3, r.zzz);
            XYZZ<F> c = xyzz_canon_lazy<F>(r);
            st(4, c.x); st(5, c.y); st(6, c.zz); st(7, c.zzz);
            break;
        }
        default: return ZK_ERR_ARG;
    }
    return ZK_OK;
    ZK_API_END
}
// Test hook for the lazy Fr domain of the NTT butterflies (frlazy.cuh): raw limbs in (9 words per element, any u32), raw
// limbs out.  op 0: reduce(a)  1: norm(a)  2..4: sub<2|3|5>(a, b)  5: mul(a, w)  6: canon(a)
// 7 / 8: radix4<true|false>(x0..x3, wa, wb, wc) -> 4 elements  9: radix2(x0, x1, w) -> 2 elements
extern "C" int zk_fr_lazy_raw(int op, const uint32_t* in, uint32_t* out) {
    ZK_API_BEGIN_NOCTX
    if (!in || !out) return ZK_ERR_ARG;
    auto ld = [&](int k) { Fr a; for (int i = 0; i < 9; i++) a.l[i] = in[9 * k + i]; return a; };
    auto st = [&](int k, const Fr& a) { for (int i = 0; i < 9; i++) out[9 * k + i] = a.l[i]; };
    switch (op) {
        case 0: st(0, frl_reduce(ld(0))); break;
        case 1: st(0, frl_norm(ld(0))); break;
        case 2: st(0, frl_sub<2>(ld(0), ld(1))); break;
        case 3: st(0, frl_sub<3>(ld(0), ld(1))); break;
        case 4: st(0, frl_sub<5>(ld(0), ld(1))); break;
        case 5: st(0, frl_mul(ld(0), ld(1))); break;
        case 6: st(0, frl_canon(ld(0))); break;
        case 7: case 8: {
            Fr x0 = ld(0), x1 = ld(1), x2 = ld(2), x3 = ld(3);
            if (op == 7) frl_radix4<true>(x0, x1, x2, x3, ld(4), ld(5), ld(6));
            else frl_radix4<false>(x0, x1, x2, x3, ld(4), ld(5), ld(6));
            st(0, x0); st(1, x1); st(2, x2); st(3, x3);
            break;
        }
        case 9: {
            Fr x0 = ld(0), x1 = ld(1);
            frl_radix2(x0, x1, ld(2));
            st(0, x0); st(1, x1);
            break;
        }
        case 10: st(0, fr_mul32(ld(0))); break;
        default: return ZK_ERR_ARG;
    }
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_fq_mul(const zk_fq* a, const zk_fq* b, zk_fq* out) {
    ZK_API_BEGIN_NOCTX
    if (!a || !b || !out) return ZK_ERR_ARG;
    Fq t = fp_mul<FqParams>(host_load_ext<FqParams>(a->l), host_load_ext<FqParams>(b->l));
    host_store_ext<FqParams>(out->l, fp_mul<FqParams>(t, fp_const<FqParams>(FqParams::EXT_TO_INT)));
    return ZK_OK;
    ZK_API_END
}
