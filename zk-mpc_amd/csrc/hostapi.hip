// hostapi.hip -- host-side group / field helpers of the C ABI (O(1) work per proof).
// Replaces, for the glue of create_proof and the MPC group Beaver steps:
//   GroupProjective::{add_assign, neg, mul} (ec/src/models/short_weierstrass_jacobian.rs:700-806),
//   CanonicalSerialize for points (:847-883), Fp256 add/sub/mul, from_repr / into_repr.
#include "../../include/zkmpc_hip.h"
#include "hostgroup.hpp"

using namespace zk;

namespace {
template <class F>
int add_t(const uint64_t* a, const uint64_t* b, uint64_t* out) {
    if (!a || !b || !out) return ZK_ERR_ARG;
    host_write_projective<F>(xyzz_add<F>(host_proj_from_abi<F>(a), host_proj_from_abi<F>(b)), out);
    return ZK_OK;
}
template <class F>
int mul_t(const uint64_t* a, const zk_fr* k, uint64_t* out) {
    if (!a || !k || !out) return ZK_ERR_ARG;
    uint32_t kw[8];
    fr_abi_to_canon_words(k->l, kw);
    host_write_projective<F>(host_scalar_mul<F>(host_proj_from_abi<F>(a), kw), out);
    return ZK_OK;
}
}  // namespace

extern "C" int zk_g1_add(const zk_g1_projective* a, const zk_g1_projective* b, zk_g1_projective* out) {
    return add_t<G1Field>((const uint64_t*)a, (const uint64_t*)b, (uint64_t*)out);
}
extern "C" int zk_g2_add(const zk_g2_projective* a, const zk_g2_projective* b, zk_g2_projective* out) {
    return add_t<G2Field>((const uint64_t*)a, (const uint64_t*)b, (uint64_t*)out);
}
extern "C" int zk_g1_neg(const zk_g1_projective* a, zk_g1_projective* out) {
    if (!a || !out) return ZK_ERR_ARG;
    host_write_projective<G1Field>(xyzz_neg<G1Field>(host_proj_from_abi<G1Field>((const uint64_t*)a)), (uint64_t*)out);
    return ZK_OK;
}
extern "C" int zk_g1_mul(const zk_g1_projective* a, const zk_fr* k, zk_g1_projective* out) {
    return mul_t<G1Field>((const uint64_t*)a, k, (uint64_t*)out);
}
extern "C" int zk_g2_mul(const zk_g2_projective* a, const zk_fr* k, zk_g2_projective* out) {
    return mul_t<G2Field>((const uint64_t*)a, k, (uint64_t*)out);
}
extern "C" int zk_g1_from_affine(const zk_g1_affine* a, zk_g1_projective* out) {
    if (!a || !out) return ZK_ERR_ARG;
    host_write_projective<G1Field>(host_aff_from_abi<G1Field>((const uint64_t*)a), (uint64_t*)out);
    return ZK_OK;
}
extern "C" int zk_g2_from_affine(const zk_g2_affine* a, zk_g2_projective* out) {
    if (!a || !out) return ZK_ERR_ARG;
    host_write_projective<G2Field>(host_aff_from_abi<G2Field>((const uint64_t*)a), (uint64_t*)out);
    return ZK_OK;
}
extern "C" int zk_g1_serialize(const zk_g1_projective* a, uint8_t out[48]) {
    if (!a || !out) return ZK_ERR_ARG;
    g1_serialize(xyzz_to_affine<G1Field>(host_proj_from_abi<G1Field>((const uint64_t*)a)), out);
    return ZK_OK;
}
extern "C" int zk_g2_serialize(const zk_g2_projective* a, uint8_t out[96]) {
    if (!a || !out) return ZK_ERR_ARG;
    g2_serialize(xyzz_to_affine<G2Field>(host_proj_from_abi<G2Field>((const uint64_t*)a)), out);
    return ZK_OK;
}

extern "C" int zk_fr_add(const zk_fr* a, const zk_fr* b, zk_fr* out) {
    if (!a || !b || !out) return ZK_ERR_ARG;
    host_store_ext<FrParams>(out->l, fp_add<FrParams>(host_load_ext<FrParams>(a->l), host_load_ext<FrParams>(b->l)));
    return ZK_OK;
}
extern "C" int zk_fr_sub(const zk_fr* a, const zk_fr* b, zk_fr* out) {
    if (!a || !b || !out) return ZK_ERR_ARG;
    host_store_ext<FrParams>(out->l, fp_sub<FrParams>(host_load_ext<FrParams>(a->l), host_load_ext<FrParams>(b->l)));
    return ZK_OK;
}
extern "C" int zk_fr_mul(const zk_fr* a, const zk_fr* b, zk_fr* out) {
    if (!a || !b || !out) return ZK_ERR_ARG;
    Fr t = fp_mul<FrParams>(host_load_ext<FrParams>(a->l), host_load_ext<FrParams>(b->l));
    host_store_ext<FrParams>(out->l, fp_mul<FrParams>(t, fp_const<FrParams>(FrParams::EXT_TO_INT)));
    return ZK_OK;
}
extern "C" int zk_fr_from_canonical(const uint64_t canon[4], zk_fr* out) {
    if (!canon || !out) return ZK_ERR_ARG;
    host_store_ext<FrParams>(out->l, fp_canon_to_ext<FrParams>(host_load_ext<FrParams>(canon)));
    return ZK_OK;
}
extern "C" int zk_fr_to_canonical(const zk_fr* a, uint64_t canon[4]) {
    if (!a || !canon) return ZK_ERR_ARG;
    host_store_ext<FrParams>(canon, fp_ext_to_canon<FrParams>(host_load_ext<FrParams>(a->l)));
    return ZK_OK;
}
