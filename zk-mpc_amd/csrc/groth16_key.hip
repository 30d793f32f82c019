// groth16_key.hip -- the proving key on the device: upload, accessors, and generate_parameters with explicit toxic waste.
//
// Replaces (reference):
//   generate_parameters                               arkworks/groth16/src/generator.rs:44-231
//   R1CStoQAP::instance_map_with_evaluation           arkworks/groth16/src/r1cs_to_qap.rs:47-92
//   ProvingKey / VerifyingKey                         arkworks/groth16/src/data_structures.rs:10-151
// The proving key's five queries are zk_bases tables (resident, with window multiples from 2^16 points on).
#include "../../include/zkmpc_hip.h"
#include "groth16_int.hpp"

using namespace zk;

namespace {

Fr host_fr_from_u64(uint64_t v) {
    Fr t = fp_zero<FrParams>();
    t.l[0] = (uint32_t)(v & MASK29);
    t.l[1] = (uint32_t)((v >> 29) & MASK29);
    t.l[2] = (uint32_t)(v >> 58);
    return fp_canon_to_int<FrParams>(t);
}

Fr host_fr_pow(const Fr& a, uint64_t e) {
    Fr r = fp_one<FrParams>();
    bool started = false;
    for (int b = 63; b >= 0; b--) {
        if (started) r = fp_sqr<FrParams>(r);
        if ((e >> b) & 1) { r = started ? fp_mul<FrParams>(r, a) : a; started = true; }
    }
    return r;
}

void host_batch_inverse(std::vector<Fr>& v) {  // Montgomery's trick (ff/src/fields/mod.rs:597-659); zeros stay zero
    std::vector<Fr> pre(v.size());
    Fr run = fp_one<FrParams>();
    for (size_t i = 0; i < v.size(); i++) {
        pre[i] = run;
        if (!fp_is_zero<FrParams>(v[i])) run = fp_mul<FrParams>(run, v[i]);
    }
    Fr inv = fp_inv<FrParams>(run);
    for (size_t i = v.size(); i-- > 0;) {
        if (fp_is_zero<FrParams>(v[i])) continue;
        Fr t = fp_mul<FrParams>(inv, pre[i]);
        inv = fp_mul<FrParams>(inv, v[i]);
        v[i] = t;
    }
}

// upload a host vector of internal-form Fr as reference-form device vector
int upload_fr(zk_ctx* ctx, const std::vector<Fr>& v, const char* slot, void** dev) {
    std::vector<uint32_t> packed(v.size() * 8 + 8);
    for (size_t i = 0; i < v.size(); i++) fp_pack<FrParams>(&packed[8 * i], fp_int_to_ext<FrParams>(v[i]));
    ZK_TRY(zk_scratch(ctx, slot, v.size() * 32 + 32, dev));
    ZK_HIP(ctx, hipMemcpyAsync(*dev, packed.data(), v.size() * 32, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
}

template <class F>
Affine<F> host_gen_mul(const Affine<F>& g, const Fr& k_int) {
    uint32_t kw[8];
    fp_pack<FrParams>(kw, fp_int_to_canon<FrParams>(k_int));
    return xyzz_to_affine<F>(xyzz_scalar_mul<F>(g, kw, 8));
}

Affine<G1Field> g1_gen() { return Affine<G1Field>{fp_const<FqParams>(FqParams::G1_GEN_X), fp_const<FqParams>(FqParams::G1_GEN_Y)}; }
Affine<G2Field> g2_gen() {
    return Affine<G2Field>{Fq2{fp_const<FqParams>(FqParams::G2_GEN_X0), fp_const<FqParams>(FqParams::G2_GEN_X1)},
                           Fq2{fp_const<FqParams>(FqParams::G2_GEN_Y0), fp_const<FqParams>(FqParams::G2_GEN_Y1)}};
}

}  // namespace

// see zk_pk::l_pad
int zk_pk_make_l_pad(zk_ctx* ctx, zk_pk* pk) {
    if (!pk->a || !pk->l || pk->l->n == 0 || pk->a->n <= pk->l->n) return ZK_OK;
    const size_t n = pk->a->n, front = n - pk->l->n, PW = 2 * G1Field::WORDS * 4;
    zk_bases* b = new zk_bases();
    b->group = 1;
    b->n = n;
    if (hipMalloc((void**)&b->dev, n * PW) != hipSuccess) { delete b; (void)hipGetLastError(); return ZK_OK; }   // no memory: L keeps its own sort
    pk->l_pad = b;
    ZK_HIP(ctx, hipMemsetAsync(b->dev, 0, front * PW, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync((char*)b->dev + front * PW, pk->l->dev, pk->l->n * PW, hipMemcpyDeviceToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return zk_bases_precompute_auto(ctx, b);
}

extern "C" int zk_pk_free(zk_ctx* ctx, zk_pk* pk) {
    ZK_API_BEGIN(ctx)
    if (!pk) return ZK_OK;
    // a pending presort / front (ZkPresort) is matched by address: it must not outlive the objects it points to, or a new
    // key allocated at the same address would adopt a sort of the old key's tables
    zk_presort_free(ctx);
    zk_bases* all[7] = {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l, pk->gamma_abc, pk->l_pad};
    for (auto* b : all) zk_bases_free(ctx, b);
    delete pk;
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_pk_upload(zk_ctx* ctx, const zk_pk_host* h, zk_pk** out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !h || !out) return ZK_ERR_ARG;
    zk_pk* pk = new zk_pk();
    int rc = zk_bases_upload_g1(ctx, h->a_query, h->a_len, &pk->a);
    if (rc == ZK_OK) rc = zk_bases_upload_g1(ctx, h->b_g1_query, h->b_g1_len, &pk->b_g1);
    if (rc == ZK_OK) rc = zk_bases_upload_g2(ctx, h->b_g2_query, h->b_g2_len, &pk->b_g2);
    if (rc == ZK_OK) rc = zk_bases_upload_g1(ctx, h->h_query, h->h_len, &pk->h);
    if (rc == ZK_OK) rc = zk_bases_upload_g1(ctx, h->l_query, h->l_len, &pk->l);
    for (zk_bases* q : {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l})
        if (rc == ZK_OK) rc = zk_bases_precompute_auto(ctx, q);
    if (rc == ZK_OK) rc = zk_pk_make_l_pad(ctx, pk);
    if (rc != ZK_OK) { zk_pk_free(ctx, pk); return rc; }
    pk->alpha_g1 = host_aff_from_abi<G1Field>((const uint64_t*)&h->alpha_g1);
    pk->beta_g1 = host_aff_from_abi<G1Field>((const uint64_t*)&h->beta_g1);
    pk->delta_g1 = host_aff_from_abi<G1Field>((const uint64_t*)&h->delta_g1);
    pk->beta_g2 = host_aff_from_abi<G2Field>((const uint64_t*)&h->beta_g2);
    pk->delta_g2 = host_aff_from_abi<G2Field>((const uint64_t*)&h->delta_g2);
    pk->gamma_g2 = aff_inf<G2Field>();
    pk->a0 = h->a_len ? host_aff_from_abi<G1Field>((const uint64_t*)&h->a_query[0]) : aff_inf<G1Field>();
    pk->b0_g1 = h->b_g1_len ? host_aff_from_abi<G1Field>((const uint64_t*)&h->b_g1_query[0]) : aff_inf<G1Field>();
    pk->b0_g2 = h->b_g2_len ? host_aff_from_abi<G2Field>((const uint64_t*)&h->b_g2_query[0]) : aff_inf<G2Field>();
    *out = pk;
    return ZK_OK;
    ZK_API_END
}

extern "C" size_t zk_pk_query_len(const zk_pk* pk, int which) {
    if (!pk) return 0;
    const zk_bases* all[6] = {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l, pk->gamma_abc};
    return (which >= 0 && which < 6 && all[which]) ? all[which]->n : 0;
}
// Borrowed handle to one query table of a resident key (valid until zk_pk_free; do not free it).
extern "C" const zk_bases* zk_pk_query_bases(const zk_pk* pk, int which) {
    if (!pk || which < 0 || which > 5) return nullptr;
    const zk_bases* all[6] = {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l, pk->gamma_abc};
    return all[which];
}
extern "C" int zk_pk_download_g1(zk_ctx* ctx, const zk_pk* pk, int which, size_t off, size_t n, zk_g1_affine* out) {
    ZK_API_BEGIN(ctx)
    if (!pk || which == 2 || which < 0 || which > 5) return ZK_ERR_ARG;
    const zk_bases* all[6] = {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l, pk->gamma_abc};
    return zk_bases_download_g1(ctx, all[which], off, n, out);
    ZK_API_END
}
extern "C" int zk_pk_download_g2(zk_ctx* ctx, const zk_pk* pk, int which, size_t off, size_t n, zk_g2_affine* out) {
    ZK_API_BEGIN(ctx)
    if (!pk || which != 2) return ZK_ERR_ARG;
    return zk_bases_download_g2(ctx, pk->b_g2, off, n, out);
    ZK_API_END
}
extern "C" int zk_pk_vk_g1(const zk_pk* pk, int which, zk_g1_affine* out) {
    ZK_API_BEGIN_NOCTX
    if (!pk || !out || which < 0 || which > 2) return ZK_ERR_ARG;
    const Affine<G1Field>* v[3] = {&pk->alpha_g1, &pk->beta_g1, &pk->delta_g1};
    host_aff_to_abi<G1Field>((uint64_t*)out, *v[which]);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_pk_vk_g2(const zk_pk* pk, int which, zk_g2_affine* out) {
    ZK_API_BEGIN_NOCTX
    if (!pk || !out || which < 0 || which > 2) return ZK_ERR_ARG;
    const Affine<G2Field>* v[3] = {&pk->beta_g2, &pk->delta_g2, &pk->gamma_g2};
    host_aff_to_abi<G2Field>((uint64_t*)out, *v[which]);
    return ZK_OK;
    ZK_API_END
}

// generate_parameters with explicit toxic waste (generator.rs:44-231)
extern "C" int zk_groth16_setup(zk_ctx* ctx, const zk_r1cs* r, const zk_fr* alpha_, const zk_fr* beta_, const zk_fr* gamma_,
                                const zk_fr* delta_, const zk_fr* tau_, const zk_fr* g1_k, const zk_fr* g2_k, zk_pk** out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !r || !alpha_ || !beta_ || !gamma_ || !delta_ || !tau_ || !g1_k || !g2_k || !out) return ZK_ERR_ARG;
    auto ld = [](const zk_fr* x) { return fp_ext_to_int<FrParams>(host_load_ext<FrParams>(x->l)); };
    const Fr alpha = ld(alpha_), beta = ld(beta_), gamma = ld(gamma_), delta = ld(delta_), t = ld(tau_);
    const Fr one = fp_one<FrParams>();
    const size_t D = (size_t)1 << r->log_d, nc = r->nc, ni = r->ni;
    const size_t nvars = (ni - 1) + r->nw;
    if (fp_is_zero<FrParams>(gamma) || fp_is_zero<FrParams>(delta)) ZK_FAIL(ctx, ZK_ERR_ARG, "setup: gamma/delta must be non-zero");

    // domain constants
    Fr w = fp_const<FrParams>(FrParams::TWO_ADIC_ROOT);
    for (uint32_t i = 0; i < (uint32_t)FR_TWO_ADICITY - r->log_d; i++) w = fp_sqr<FrParams>(w);
    const Fr size_inv = fp_inv<FrParams>(host_fr_from_u64(D));
    const Fr zt = fp_sub<FrParams>(host_fr_pow(t, D), one);  // evaluate_vanishing_polynomial(t)
    if (fp_is_zero<FrParams>(zt)) ZK_FAIL(ctx, ZK_ERR_ARG, "setup: tau lies in the evaluation domain");

    // evaluate_all_lagrange_coefficients(t): u_i = (zt/D) w^i / (t - w^i)   (radix2/mod.rs:116-165)
    std::vector<Fr> u(D), den(D);
    {
        Fr l = fp_mul<FrParams>(zt, size_inv), rr = one;
        for (size_t i = 0; i < D; i++) {
            den[i] = fp_sub<FrParams>(t, rr);
            u[i] = l;
            l = fp_mul<FrParams>(l, w);
            rr = fp_mul<FrParams>(rr, w);
        }
        host_batch_inverse(den);
        for (size_t i = 0; i < D; i++) u[i] = fp_mul<FrParams>(u[i], den[i]);
        den.clear(); den.shrink_to_fit();
    }
    // instance_map_with_evaluation (r1cs_to_qap.rs:47-92)
    std::vector<Fr> a(nvars + 1, fp_zero<FrParams>()), b(nvars + 1, fp_zero<FrParams>()), c(nvars + 1, fp_zero<FrParams>());
    for (size_t i = 0; i < ni; i++) a[i] = u[nc + i];
    std::vector<Fr>* abc[3] = {&a, &b, &c};
    for (int k = 0; k < 3; k++) {
        const auto& m = r->m[k];
        auto& dst = *abc[k];
        for (size_t i = 0; i < nc; i++)
            for (uint32_t e = m.h_row_ptr[i]; e < m.h_row_ptr[i + 1]; e++) {
                Fr term = m.all_one ? u[i] : fp_mul<FrParams>(u[i], m.h_coeff[e]);
                dst[m.h_col[e]] = fp_add<FrParams>(dst[m.h_col[e]], term);
            }
    }
    u.clear(); u.shrink_to_fit();
    const Fr gamma_inv = fp_inv<FrParams>(gamma), delta_inv = fp_inv<FrParams>(delta);
    std::vector<Fr> gamma_abc(ni), l(nvars + 1 - ni);
    for (size_t i = 0; i <= nvars; i++) {
        Fr s = fp_add<FrParams>(fp_add<FrParams>(fp_mul<FrParams>(beta, a[i]), fp_mul<FrParams>(alpha, b[i])), c[i]);
        if (i < ni) gamma_abc[i] = fp_mul<FrParams>(s, gamma_inv);
        else l[i - ni] = fp_mul<FrParams>(s, delta_inv);
    }
    c.clear(); c.shrink_to_fit();
    std::vector<Fr> hq(D - 1);
    {
        Fr p = fp_mul<FrParams>(zt, delta_inv);
        for (size_t i = 0; i + 1 < D; i++) { hq[i] = p; p = fp_mul<FrParams>(p, t); }
    }

    zk_pk* pk = new zk_pk();
    void* dev;
    int rc = upload_fr(ctx, a, "setup_scalars", &dev);
    if (rc == ZK_OK) rc = zk_fixed_base_g1_dev(ctx, g1_k, dev, a.size(), &pk->a);
    if (rc == ZK_OK) rc = upload_fr(ctx, b, "setup_scalars", &dev);
    if (rc == ZK_OK) rc = zk_fixed_base_g1_dev(ctx, g1_k, dev, b.size(), &pk->b_g1);
    if (rc == ZK_OK) rc = zk_fixed_base_g2_dev(ctx, g2_k, dev, b.size(), &pk->b_g2);
    if (rc == ZK_OK) rc = upload_fr(ctx, hq, "setup_scalars", &dev);
    if (rc == ZK_OK) rc = zk_fixed_base_g1_dev(ctx, g1_k, dev, hq.size(), &pk->h);
    if (rc == ZK_OK) rc = upload_fr(ctx, l, "setup_scalars", &dev);
    if (rc == ZK_OK) rc = zk_fixed_base_g1_dev(ctx, g1_k, dev, l.size(), &pk->l);
    if (rc == ZK_OK) rc = upload_fr(ctx, gamma_abc, "setup_scalars", &dev);
    if (rc == ZK_OK) rc = zk_fixed_base_g1_dev(ctx, g1_k, dev, gamma_abc.size(), &pk->gamma_abc);
    for (zk_bases* q : {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l})
        if (rc == ZK_OK) rc = zk_bases_precompute_auto(ctx, q);
    if (rc == ZK_OK) rc = zk_pk_make_l_pad(ctx, pk);
    if (rc != ZK_OK) { zk_pk_free(ctx, pk); return rc; }
    pk->points_in_subgroup = true;            // every point of this key is a scalar multiple of a generator

    const Fr k1 = ld(g1_k), k2 = ld(g2_k);
    const Affine<G1Field> g1 = host_gen_mul<G1Field>(g1_gen(), k1);
    const Affine<G2Field> g2 = host_gen_mul<G2Field>(g2_gen(), k2);
    pk->alpha_g1 = host_gen_mul<G1Field>(g1, alpha);
    pk->beta_g1 = host_gen_mul<G1Field>(g1, beta);
    pk->delta_g1 = host_gen_mul<G1Field>(g1, delta);
    pk->beta_g2 = host_gen_mul<G2Field>(g2, beta);
    pk->delta_g2 = host_gen_mul<G2Field>(g2, delta);
    pk->gamma_g2 = host_gen_mul<G2Field>(g2, gamma);
    rc = first_point<G1Field>(ctx, pk->a, &pk->a0);
    if (rc == ZK_OK) rc = first_point<G1Field>(ctx, pk->b_g1, &pk->b0_g1);
    if (rc == ZK_OK) rc = first_point<G2Field>(ctx, pk->b_g2, &pk->b0_g2);
    if (rc != ZK_OK) { zk_pk_free(ctx, pk); return rc; }
    *out = pk;
    return ZK_OK;
    ZK_API_END
}

