// groth16.hip -- R1CS -> QAP witness map, Groth16 prover and known-trapdoor setup on the device.
//
// Replaces (reference):
//   R1CStoQAP::witness_map + evaluate_constraint      src/groth16.rs:205-306
//                                                     (stock: arkworks/groth16/src/r1cs_to_qap.rs:94-160)
//   create_proof / calculate_coeff                    src/groth16.rs:68-201 (stock prover.rs:44-153,189-203)
//   generate_parameters                               arkworks/groth16/src/generator.rs:44-231
//   R1CStoQAP::instance_map_with_evaluation           arkworks/groth16/src/r1cs_to_qap.rs:47-92
//   ConstraintMatrices                                arkworks/snark/relations/src/r1cs/constraint_system.rs:650-676
//
// Device layout: the three matrices are CSR (row_ptr u32, col u32, coeff = Fr in the device's
// internal Montgomery form so coeff * z needs no conversion); assignment, QAP vectors and h are
// Fr vectors in the reference's form.  The proving key's five queries are zk_bases tables.
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "hostgroup.hpp"
#include "hostfield64.hpp"
#include "internal.hpp"
#include "sharednet.hpp"
#include <chrono>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <vector>

using namespace zk;

struct zk_r1cs {
    size_t nc = 0, ni = 0, nw = 0;
    uint32_t log_d = 0;
    struct Mat {
        uint32_t* row_ptr = nullptr;
        uint32_t* col = nullptr;
        uint32_t* coeff = nullptr;  // nnz * 8 words, internal form
        size_t nnz = 0;
        bool all_one = false;       // every coefficient is 1: the product is skipped (src/groth16.rs:220-224)
        std::vector<uint32_t> h_row_ptr, h_col;
        std::vector<Fr> h_coeff;    // internal form (empty when all_one)
    } m[3];
};

struct zk_pk {
    zk_bases *a = nullptr, *b_g1 = nullptr, *b_g2 = nullptr, *h = nullptr, *l = nullptr, *gamma_abc = nullptr;
    // l_query behind as many points at infinity as a_query has entries for the instance (l_pad[ni + j] = l[j]), so that the
    // L job indexes its table by the position in z like A and B do and reuses their sort of z[1..]; the instance part adds
    // infinity, which the complete addition skips.  Only the prover's pipeline reads it.
    zk_bases* l_pad = nullptr;
    Affine<G1Field> alpha_g1, beta_g1, delta_g1, a0, b0_g1;
    Affine<G2Field> beta_g2, delta_g2, gamma_g2, b0_g2;
};

namespace {

uint32_t domain_log(size_t num_coeffs) {
    uint32_t lg = 0;
    while (((size_t)1 << lg) < num_coeffs) lg++;
    return lg;
}

// out[row] = sum_k coeff[k] * z[col[k]]  for row < nc;  out[nc + i] = z[i] for i < n_copy; rest 0.
__global__ void __launch_bounds__(256)
k_spmv(const uint32_t* row_ptr, const uint32_t* col, const uint32_t* coeff, int all_one, const void* z, size_t nc,
       size_t n_copy, size_t D, void* out) {
    for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < D; r += (size_t)gridDim.x * blockDim.x) {
        Fr acc = fp_zero<FrParams>();
        if (r < nc) {
            uint32_t lo = row_ptr[r], hi = row_ptr[r + 1];
            for (uint32_t k = lo; k < hi; k++) {
                Fr v = fr_load(z, col[k]);
                if (!all_one) v = fr_mul(v, fr_load(coeff, k));
                acc = fr_add(acc, v);
            }
        } else if (r < nc + n_copy) {
            acc = fr_load(z, r - nc);
        }
        fr_store(out, r, acc);
    }
}

int upload_mat(zk_ctx* ctx, zk_r1cs::Mat& m, size_t nc, const uint32_t* rp, const uint32_t* col, const zk_fr* coeff) {
    m.nnz = rp[nc];
    m.h_row_ptr.assign(rp, rp + nc + 1);
    m.h_col.assign(col, col + m.nnz);
    m.h_coeff.resize(m.nnz);
    const Fr one = fp_one<FrParams>();
    m.all_one = true;
    for (size_t k = 0; k < m.nnz; k++) {
        m.h_coeff[k] = fp_ext_to_int<FrParams>(host_load_ext<FrParams>(coeff[k].l));
        if (!fp_eq<FrParams>(m.h_coeff[k], one)) m.all_one = false;
    }
    ZK_HIP(ctx, hipMalloc((void**)&m.row_ptr, (nc + 1) * 4));
    ZK_HIP(ctx, hipMalloc((void**)&m.col, (m.nnz ? m.nnz : 1) * 4));
    ZK_HIP(ctx, hipMemcpy(m.row_ptr, rp, (nc + 1) * 4, hipMemcpyHostToDevice));
    if (m.nnz) ZK_HIP(ctx, hipMemcpy(m.col, col, m.nnz * 4, hipMemcpyHostToDevice));
    if (!m.all_one) {
        std::vector<uint32_t> packed(m.nnz * 8);
        for (size_t k = 0; k < m.nnz; k++) fp_pack<FrParams>(&packed[8 * k], m.h_coeff[k]);
        ZK_HIP(ctx, hipMalloc((void**)&m.coeff, m.nnz * 32));
        ZK_HIP(ctx, hipMemcpy(m.coeff, packed.data(), m.nnz * 32, hipMemcpyHostToDevice));
    } else {
        m.h_coeff.clear();
        m.h_coeff.shrink_to_fit();
    }
    return ZK_OK;
}

int spmv(zk_ctx* ctx, const zk_r1cs* r, int which, const void* z, size_t n_copy, void* out) {
    const auto& m = r->m[which];
    size_t D = (size_t)1 << r->log_d;
    hipLaunchKernelGGL(k_spmv, zk_grid(D, 256), 256, 0, ctx->stream, m.row_ptr, m.col, m.coeff, m.all_one ? 1 : 0, z, r->nc,
                       n_copy, D, out);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
}

}  // namespace

// out[r] = <row r of matrix `which`, z> for r < num_constraints, zero up to out_len (inner_prod_fn of the Marlin prover,
// marlin/src/ahp/prover.rs:258-278; the same product as evaluate_constraint, src/groth16.rs:205-234).
extern "C" int zk_r1cs_matvec_dev(zk_ctx* ctx, const zk_r1cs* r, int which, const void* z_dev, void* out_dev, size_t out_len) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !r || !z_dev || !out_dev || which < 0 || which > 2) return ZK_ERR_ARG;
    if (out_len < r->nc) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_r1cs_matvec_dev: out_len is smaller than the number of constraints");
    const auto& m = r->m[which];
    hipLaunchKernelGGL(k_spmv, zk_grid(out_len, 256), 256, 0, ctx->stream, m.row_ptr, m.col, m.coeff, m.all_one ? 1 : 0, z_dev, r->nc,
                       (size_t)0, out_len, out_dev);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

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

template <class F>
int first_point(zk_ctx* ctx, const zk_bases* b, Affine<F>* out) {
    if (!b || b->n == 0) { *out = aff_inf<F>(); return ZK_OK; }
    uint32_t w[2 * F::WORDS];
    ZK_HIP(ctx, hipMemcpy(w, b->dev, sizeof w, hipMemcpyDeviceToHost));
    *out = aff_load<F>(w);
    return ZK_OK;
}

}  // namespace

extern "C" uint32_t zk_r1cs_domain_log(const zk_r1cs* r) { return r ? r->log_d : 0; }

extern "C" int zk_r1cs_upload(zk_ctx* ctx, const zk_r1cs_host* h, zk_r1cs** out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !h || !out || h->num_instance == 0) return ZK_ERR_ARG;
    zk_r1cs* r = new zk_r1cs();
    r->nc = h->num_constraints; r->ni = h->num_instance; r->nw = h->num_witness;
    r->log_d = domain_log(r->nc + r->ni);  // src/groth16.rs:256-257
    int rc = upload_mat(ctx, r->m[0], r->nc, h->a_row_ptr, h->a_col, h->a_coeff);
    if (rc == ZK_OK) rc = upload_mat(ctx, r->m[1], r->nc, h->b_row_ptr, h->b_col, h->b_coeff);
    if (rc == ZK_OK) rc = upload_mat(ctx, r->m[2], r->nc, h->c_row_ptr, h->c_col, h->c_coeff);
    if (rc != ZK_OK) { zk_r1cs_free(ctx, r); return rc; }
    *out = r;
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_r1cs_free(zk_ctx* ctx, zk_r1cs* r) {
    ZK_API_BEGIN(ctx)
    if (!r) return ZK_OK;
    zk_presort_free(ctx);          // see zk_pk_free
    if (ctx) (void)hipStreamSynchronize(ctx->stream);
    for (auto& m : r->m) {
        if (m.row_ptr) (void)hipFree(m.row_ptr);
        if (m.col) (void)hipFree(m.col);
        if (m.coeff) (void)hipFree(m.coeff);
    }
    delete r;
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_r1cs_mul_chain(zk_ctx* ctx, size_t n, zk_r1cs** out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !out || n == 0 || n > ((size_t)1 << 27)) return ZK_ERR_ARG;
    // variables: [1, pub] ++ witness[w_0..w_n]; w_{n+1} is the public input (index 1)
    auto idx = [n](size_t j) -> uint32_t { return j <= n ? (uint32_t)(2 + j) : 1u; };
    std::vector<uint32_t> rp(n + 1), ca(n), cb(n), cc(n);
    for (size_t i = 0; i <= n; i++) rp[i] = (uint32_t)i;
    for (size_t i = 0; i < n; i++) { ca[i] = idx(i); cb[i] = idx(i + 1); cc[i] = idx(i + 2); }
    zk_fr one_ext;
    host_store_ext<FrParams>(one_ext.l, fp_int_to_ext<FrParams>(fp_one<FrParams>()));
    std::vector<zk_fr> ones(n, one_ext);
    zk_r1cs_host h;
    h.num_constraints = n; h.num_instance = 2; h.num_witness = n + 1;
    h.a_row_ptr = h.b_row_ptr = h.c_row_ptr = rp.data();
    h.a_col = ca.data(); h.b_col = cb.data(); h.c_col = cc.data();
    h.a_coeff = h.b_coeff = h.c_coeff = ones.data();
    return zk_r1cs_upload(ctx, &h, out);
    ZK_API_END
}

extern "C" int zk_mul_chain_assignment_dev(zk_ctx* ctx, size_t n, const zk_fr* w0, const zk_fr* w1, void* z_dev) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !w0 || !w1 || !z_dev || n == 0) return ZK_ERR_ARG;
    // the chain is inherently sequential: computed on the host (input generation, not on the proving path)
    std::vector<Fr> w(n + 2);
    w[0] = fp_ext_to_int<FrParams>(host_load_ext<FrParams>(w0->l));
    w[1] = fp_ext_to_int<FrParams>(host_load_ext<FrParams>(w1->l));
    for (size_t i = 0; i < n; i++) w[i + 2] = fp_mul<FrParams>(w[i], w[i + 1]);
    std::vector<uint32_t> packed((n + 3) * 8);
    fp_pack<FrParams>(&packed[0], fp_int_to_ext<FrParams>(fp_one<FrParams>()));
    fp_pack<FrParams>(&packed[8], fp_int_to_ext<FrParams>(w[n + 1]));
    for (size_t j = 0; j <= n; j++) fp_pack<FrParams>(&packed[8 * (2 + j)], fp_int_to_ext<FrParams>(w[j]));
    ZK_HIP(ctx, hipMemcpyAsync(z_dev, packed.data(), packed.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZK_OK;
    ZK_API_END
}

// ---- witness map -----------------------------------------------------------------------------

extern "C" int zk_groth16_witness_map_pre_dev(zk_ctx* ctx, const zk_r1cs* r, const void* z, int include_instance, void* a,
                                              void* b, void* c) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !r || !z || !a || !b || !c) return ZK_ERR_ARG;
    // a[nc..nc+ni] = instance assignment (src/groth16.rs:272-276).  For shares, z already holds each
    // party's share of the instance (the leader holds the public value, the others zero), so the
    // copy is the same linear operation; include_instance=0 lets a caller suppress it.
    ZK_TRY(spmv(ctx, r, 0, z, include_instance ? r->ni : 0, a));
    ZK_TRY(spmv(ctx, r, 1, z, 0, b));
    ZK_TRY(spmv(ctx, r, 2, z, 0, c));
    void* v[3] = {a, b, c};
    ZK_TRY(zk_ntt_launch_batch(ctx, v, 3, r->log_d, 1, 0));    // ifft of a, b, c       (:278-279,295), one launch per pass
    // coset_fft of a and b (:281-282).  c stays in coefficient form: the reference's coset_ifft((ab - c) / Z(g)) on coset
    // evaluations (:296-303) is, interpolation being linear, coset_ifft(ab) - c on coefficients -- the same field elements with
    // one transform less (witness_map_post subtracts there)
    return zk_ntt_launch_batch(ctx, v, 2, r->log_d, 0, 1);
    ZK_API_END
}

extern "C" int zk_groth16_witness_map_post_dev(zk_ctx* ctx, const zk_r1cs* r, void* ab, const void* c) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !r || !ab || !c) return ZK_ERR_ARG;
    uint32_t zinv[9];
    ZK_TRY(zk_ntt_vanishing_inv(ctx, r->log_d, zinv));
    ZK_TRY(zk_ntt_launch(ctx, ab, r->log_d, 1, 1));                                  // coset_ifft of ab  (:303)
    return zk_vec_sub_scale_launch(ctx, ab, c, zinv, ab, (size_t)1 << r->log_d);    // (. - c) / Z(g), c in coefficient form (:298-302)
    ZK_API_END
}

extern "C" int zk_groth16_witness_map_dev(zk_ctx* ctx, const zk_r1cs* r, const void* z, void* h) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !r || !z || !h) return ZK_ERR_ARG;
    size_t D = (size_t)1 << r->log_d;
    void *b, *c;
    ZK_TRY(zk_scratch(ctx, "wm_b", D * 32, &b));
    ZK_TRY(zk_scratch(ctx, "wm_c", D * 32, &c));
    ZK_TRY(zk_groth16_witness_map_pre_dev(ctx, r, z, 1, h, b, c));
    ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_MUL, h, b, h, D));  // batch_product_in_place (:285)
    return zk_groth16_witness_map_post_dev(ctx, r, h, c);
    ZK_API_END
}

// ---- proving key -----------------------------------------------------------------------------

// see zk_pk::l_pad.  ZK_L_SHARE=0 keeps the L job on its own table and its own sort.
static int pk_make_l_pad(zk_ctx* ctx, zk_pk* pk) {
    static const bool on = !(getenv("ZK_L_SHARE") && atoi(getenv("ZK_L_SHARE")) == 0);
    if (!on || !pk->a || !pk->l || pk->l->n == 0 || pk->a->n <= pk->l->n) return ZK_OK;
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
    if (rc == ZK_OK) rc = pk_make_l_pad(ctx, pk);
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
    if (rc == ZK_OK) rc = pk_make_l_pad(ctx, pk);
    if (rc != ZK_OK) { zk_pk_free(ctx, pk); return rc; }

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

// ---- prover ----------------------------------------------------------------------------------

// A sort of z[1..] (shared by the B-in-G2 / A / B-in-G1 / L jobs) enqueued ahead of the MSMs: the collaborative prover
// calls zk_groth16_msms_presort_dev right after the local half of the witness map, so the sort runs under the Beaver open
// (network time) instead of in front of the first accumulate kernel.  Owned by the context until run_msms takes it over.
struct ZkPresort {
    ZkMsmJob job;
    const zk_pk* pk = nullptr;
    const void* z = nullptr;
    // the whole FRONT of the next local proof (zk_groth16_hint_next_dev): besides the sort of z also its witness map and
    // the H job's sort, enqueued behind the current proof's last kernels so that they run under its reduce tail and the
    // host time between two proofs
    bool front = false;
    const zk_r1cs* r = nullptr;
    void* h = nullptr;                 // where the witness map put h
    ZkMsmJob jobh;
    hipEvent_t wm_done = nullptr;
    // zk_groth16_msms_begin_dev: not only the sort of z but the four MSMs over z -- A, B in G1, B in G2, L: sorted, their
    // accumulate kernels and reduce chains enqueued -- are under way; zk_groth16_msms_dev then adds the H job and collects all
    // five.  The collaborative prover calls it before its Beaver open: the exchange and the second half of the witness map run
    // under 12 ms of accumulate kernels that do not need h.
    bool begun = false;
    ZkMsmJob j1, j2, j3;
    ~ZkPresort() { if (wm_done) (void)hipEventDestroy(wm_done); }
};
void zk_presort_free(zk_ctx* ctx) {
    if (!ctx || !ctx->presort) return;
    ZkPresort* p = (ZkPresort*)ctx->presort;
    ctx->presort = nullptr;
    if (ctx->aux.size()) (void)hipStreamSynchronize(ctx->aux[0]);   // its kernels write the job's scratch slot
    if (p->front || p->begun) {                                     // these run on the accumulate and context streams as well
        if (ctx->acc_stream) (void)hipStreamSynchronize(ctx->acc_stream);
        (void)hipStreamSynchronize(ctx->stream);
    }
    delete p;
}

namespace {

int ensure_aux(zk_ctx* ctx, size_t k) {
    while (ctx->aux.size() < k) {
        hipStream_t st;
        ZK_HIP(ctx, zk_stream_create(&st, true));
        ctx->aux.push_back(st);
    }
    if (!ctx->acc_stream) {
        // ZK_ACC_CU_RESERVE=N: the accumulate stream is created with a CU mask that leaves N compute units to the other streams
        // (sorts, reduce chains, witness map).  An accumulate kernel fills every wave slot it is offered (232 registers per lane:
        // no other wave fits beside two of its own on a SIMD), and a co-running kernel otherwise only advances at the rate its
        // blocks retire -- a 0.45 ms sort takes 3 ms beside it.
        const char* e = getenv("ZK_ACC_CU_RESERVE");
        const int reserve = e ? atoi(e) : 0;
        if (reserve > 0 && reserve < ctx->n_cu) {
            const int words = (ctx->n_cu + 31) / 32;
            std::vector<uint32_t> mask(words, 0);
            const char* hi = getenv("ZK_ACC_CU_RESERVE_HIGH");     // which end of the mask is withheld (the bit -> CU mapping is the driver's)
            for (int i = 0; i < ctx->n_cu; i++) {
                const bool keep = hi ? i < ctx->n_cu - reserve : i >= reserve;
                if (keep) mask[i / 32] |= 1u << (i % 32);
            }
            ZK_HIP(ctx, hipExtStreamCreateWithCUMask(&ctx->acc_stream, (uint32_t)words, mask.data()));
        } else {
            ZK_HIP(ctx, zk_stream_create(&ctx->acc_stream, false));
        }
    }
    return ZK_OK;
}

// The five MSMs of create_proof as a pipeline over THREE streams (the runtime maps streams onto ~3
// usable hardware queues; a fourth stream lands on an occupied queue and serialises behind it):
//   main       : witness map, then the sort of the H job (its scalars come out of the witness map)
//   sort/reduce: sort of the z-dependent jobs first (B-in-G2 / A / B-in-G1 share one sort: same scalars
//                z[1..]), then, as the accumulate kernels complete, each job's reduce phase in job order
//   accum      : the five accumulate kernels back to back, B-in-G2 first (longest reduce), H last;
//                the first one is gated on the witness map, which would otherwise be starved 15x beside it
int run_msms(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z, const void* h_in, void* h_scratch,
             zk_g1_projective out_g1[4], zk_g2_projective* out_g2, const std::function<void()>& after_abc = nullptr) {
    const size_t D = (size_t)1 << r->log_d;
    const size_t nvars = (r->ni - 1) + r->nw;
    const char* zb = (const char*)z;
    if (pk->a->n != nvars + 1 || pk->b_g1->n != nvars + 1 || pk->b_g2->n != nvars + 1 || pk->l->n != r->nw)
        ZK_FAIL(ctx, ZK_ERR_ARG, "groth16: proving key does not match the constraint system");
    ZK_TRY(ensure_aux(ctx, 1));
    hipStream_t s_sort = ctx->aux[0], s_red = ctx->aux[0], s_acc = ctx->acc_stream;
    struct Events {                               // destroyed on every exit path (the error returns below used to leak them)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } evs;
    ZK_HIP(ctx, hipEventCreateWithFlags(&evs.e0, hipEventDisableTiming));
    ZK_HIP(ctx, hipEventCreateWithFlags(&evs.e1, hipEventDisableTiming));
    const hipEvent_t e0 = evs.e0, e1 = evs.e1;
    ZK_HIP(ctx, hipEventRecord(e0, ctx->stream));
    ZkMsmJob own[5];  // 0: B in G2, 1: A, 2: B in G1, 3: L, 4: H
    ZkMsmJob* J[5] = {&own[0], &own[1], &own[2], &own[3], &own[4]};
    // the sort of z[1..] may already be running (zk_groth16_msms_presort_dev, enqueued by the collaborative prover before
    // its Beaver open): take it over as job 0's
    ZkPresort* pre = (ZkPresort*)ctx->presort;
    ctx->presort = nullptr;
    std::unique_ptr<ZkPresort> pre_owner(pre);
    // (a begun set can only be taken over whole, by the call that brings h: anything else drains and drops it below -- its jobs
    // own the scratch slots this call is about to use)
    const bool presorted = pre && pre->pk == pk && pre->z == z && pre->job.n == nvars && (!pre->begun || (h_in && pre->r == r));
    const bool fronted = presorted && pre->front && !h_in && pre->r == r && pre->h == h_scratch;
    if (pre && !presorted) {                       // a front / sort for other inputs: let it drain before its scratch is reused
        pre_owner.release();
        ctx->presort = pre;
        zk_presort_free(ctx);
        pre = nullptr;
    }
    const bool begun = presorted && pre->begun && h_in && pre->r == r;     // the four z jobs are already enqueued to the end
    if (presorted) J[0] = &pre->job;
    if (fronted) J[4] = &pre->jobh;
    if (begun) { J[1] = &pre->j1; J[2] = &pre->j2; J[3] = &pre->j3; }
    // z was produced on the context stream.  (With a front that stream already carries this proof's witness map and H-sort:
    // waiting for it here would hold the sort stream -- and the G2 reduce chain on it -- until the H-sort is through.)
    if (!fronted) ZK_HIP(ctx, hipStreamWaitEvent(s_sort, e0, 0));
    int rc = presorted ? ZK_OK : zk_msm_prepare(ctx, J[0], pk->b_g2, 1, zb + 32, nvars, 1);                 // src/groth16.rs:160 (query[1..])
    if (rc == ZK_OK && !begun) rc = zk_msm_prepare(ctx, J[1], pk->a, 1, zb + 32, nvars, 2);              // :137
    if (rc == ZK_OK && !begun) rc = zk_msm_prepare(ctx, J[2], pk->b_g1, 1, zb + 32, nvars, 3);           // :148
    // :110: aux_assignment against l_query; over the padded table the same sum reads z[1..] (the instance meets infinity)
    const bool l_shared = pk->l_pad && pk->l_pad->n == nvars + 1 && (pk->l_pad->pre != nullptr) == (pk->a->pre != nullptr) &&
                          pk->l_pad->c_pre == pk->a->c_pre;
    if (rc == ZK_OK && !begun) rc = l_shared ? zk_msm_prepare(ctx, J[3], pk->l_pad, 1, zb + 32, nvars, 4)
                                             : zk_msm_prepare(ctx, J[3], pk->l, 0, zb + r->ni * 32, r->nw, 4);
    if (rc == ZK_OK && !presorted) rc = zk_msm_enqueue_sort(ctx, J[0], s_sort, nullptr);
    if (rc == ZK_OK && !begun) rc = zk_msm_enqueue_sort(ctx, J[1], s_sort, J[0]);
    if (rc == ZK_OK && !begun) rc = zk_msm_enqueue_sort(ctx, J[2], s_sort, J[0]);
    if (rc == ZK_OK && !begun && l_shared) rc = zk_msm_enqueue_sort(ctx, J[3], s_sort, J[0]);
    const void* h = h_in;
    ZkPhaseTimer tm(ctx);
    static const bool gate = !(getenv("ZK_WM_GATE") && atoi(getenv("ZK_WM_GATE")) == 0);
    if (fronted) {
        // witness map and H's sort were enqueued with the previous proof (enqueue_front below)
        h = h_scratch;
        if (gate) ZK_HIP(ctx, hipStreamWaitEvent(s_acc, pre->wm_done, 0));
    } else {
        if (rc == ZK_OK && !h_in) {
            tm.begin("witness_map");
            rc = zk_groth16_witness_map_dev(ctx, r, z, h_scratch);
            tm.end();
            h = h_scratch;
        }
        // h_acc: min(len) rule (variable_base.rs:15-17): h_query has D-1 entries, h has D
        if (rc == ZK_OK) rc = zk_msm_prepare(ctx, J[4], pk->h, 0, h, std::min(pk->h->n, D), 5);   // :106
        if (rc == ZK_OK) {
            ZK_HIP(ctx, hipEventRecord(e1, ctx->stream));
            if (gate) ZK_HIP(ctx, hipStreamWaitEvent(s_acc, e1, 0));
            rc = zk_msm_enqueue_sort(ctx, J[4], ctx->stream, nullptr);
        }
    }
    // L's sort after the witness map (it is not needed before the fourth accumulate kernel)
    if (rc == ZK_OK && !l_shared && !begun) {
        ZK_HIP(ctx, hipStreamWaitEvent(s_sort, e1, 0));
        rc = zk_msm_enqueue_sort(ctx, J[3], s_sort, nullptr);
    }
    // accumulate order (job numbers: 0 = B in G2, 1 = A, 2 = B in G1, 3 = L, 4 = H; H's scalars arrive last)
    int ord[5] = {0, 1, 2, 3, 4};
    if (const char* e = getenv("ZK_MSM_ORDER")) {
        bool seen[5] = {false, false, false, false, false};
        int tmp[5], cnt = 0;
        for (; cnt < 5 && e[cnt] >= '0' && e[cnt] <= '4' && !seen[e[cnt] - '0']; cnt++) { tmp[cnt] = e[cnt] - '0'; seen[tmp[cnt]] = true; }
        if (cnt == 5) for (int k = 0; k < 5; k++) ord[k] = tmp[k];
    }
    if (begun) { ord[0] = 0; ord[1] = 1; ord[2] = 2; ord[3] = 3; ord[4] = 4; }        // the order zk_groth16_msms_begin_dev used
    for (int k = 0; k < 5 && rc == ZK_OK; k++)
        if (!begun || ord[k] == 4) rc = zk_msm_enqueue_accum(ctx, J[ord[k]], s_acc);
    // B-in-G2's reduce chain (the long one) stays on the sort stream; the four G1 reduces go to the main stream, idle by
    // then, so that each runs right behind its own accumulate kernel instead of queueing behind the G2 chain (that
    // queueing left 4 x 0.7 ms of reduces after the last accumulate).
    // The last two reduce chains alternate between the two streams (the sort stream is idle again once the G2 chain is
    // through): on one stream the last job's chain queued behind its predecessor's, which was still waiting for slots
    // beside the last accumulate kernel, and ~0.6 ms of it ran after the GPU had otherwise gone idle.
    static const int alt = getenv("ZK_REDUCE_ALT") ? atoi(getenv("ZK_REDUCE_ALT")) : 2;   // 0: all on main, 1: last on sort, 2: every other one
    // experiment (ZK_REDUCE_TAIL=1): the G1 reduce chains held back until the last accumulate kernel is through, so that they run
    // beside each other at the tail instead of beside (and stretching) the accumulate kernels
    static const bool reduce_tail = getenv("ZK_REDUCE_TAIL") && atoi(getenv("ZK_REDUCE_TAIL")) != 0;
    for (int k = 0; k < 5 && rc == ZK_OK; k++) {
        hipStream_t rs = (ord[k] == 0 || (alt == 1 && k == 4) || (alt == 2 && (k & 1) == 0)) ? s_red : ctx->stream;
        if (begun && ord[k] == 0) continue;                 // B in G2's chain went out with zk_groth16_msms_begin_dev
        if (reduce_tail && ord[k] != 0 && J[ord[4]]->accum_done) ZK_HIP(ctx, hipStreamWaitEvent(rs, J[ord[4]]->accum_done, 0));
        rc = zk_msm_enqueue_reduce(ctx, J[ord[k]], rs);
    }
    // The caller announced the next assignment (zk_groth16_hint_next_dev): enqueue that proof's front now, behind this
    // proof's kernels.  Its z-sort goes on the accumulate stream (in order behind the five accumulate kernels, the readers of
    // this proof's sort products; it also waits for the reduce chains, whose fold kernels read the segment tables), its
    // witness map and H-sort on the context stream (behind this proof's H-sort and reduce chains; the H-sort also waits for
    // the last accumulate kernel, the reader of the H job's sort products).  No buffer is doubled: stream order and these
    // events keep every reader in front of the next writer.
    bool front_enqueued = false;
    if (rc == ZK_OK && ctx->next_z && !h_in && l_shared) {
        const void* zn = ctx->next_z;
        ctx->next_z = nullptr;
        std::unique_ptr<ZkPresort> nf(new ZkPresort());
        nf->pk = pk; nf->z = zn; nf->r = r; nf->h = h_scratch; nf->front = true;
        if (zn == ctx->next_z_dev && ctx->next_z_ready) {      // announced from the host: its upload runs on the copy stream
            ZK_HIP(ctx, hipStreamWaitEvent(s_acc, ctx->next_z_ready, 0));
            ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->next_z_ready, 0));
        }
        // a reduce chain reads its job's segment tables (k_fold: ctr / heavy live in the sort scratch of the job's slot).  The
        // next proof's z-sort rewrites slot 1, whose products the four z jobs share: it waits for THEIR chains (not for the H
        // job's, the last one: the z-sort is meant to run under that tail); the next H-sort rewrites slot 5 and waits for the
        // H job's chain below.
        for (int k = 0; k < 4; k++)
            if (J[k]->reduce_done) ZK_HIP(ctx, hipStreamWaitEvent(s_acc, J[k]->reduce_done, 0));
        rc = zk_msm_prepare(ctx, &nf->job, pk->b_g2, 1, (const char*)zn + 32, nvars, 1);
        if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, &nf->job, s_acc, nullptr);
        if (rc == ZK_OK) rc = zk_groth16_witness_map_dev(ctx, r, zn, h_scratch);
        if (rc == ZK_OK) {
            ZK_HIP(ctx, hipEventCreateWithFlags(&nf->wm_done, hipEventDisableTiming));
            ZK_HIP(ctx, hipEventRecord(nf->wm_done, ctx->stream));
            if (J[4]->accum_done) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, J[4]->accum_done, 0));
            // H's own chain (slot 5: k_fold reads ctr / heavy there) may be on the other stream
            if (J[4]->reduce_done) ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, J[4]->reduce_done, 0));
            rc = zk_msm_prepare(ctx, &nf->jobh, pk->h, 0, h_scratch, std::min(pk->h->n, D), 5);
            if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, &nf->jobh, ctx->stream, nullptr);
        }
        if (rc == ZK_OK) { ctx->presort = nf.release(); front_enqueued = true; }
    }
    ctx->next_z = nullptr;
    // finish in completion order: the host-side Horner of an early job overlaps the GPU work of the later ones
    void* outs[5] = {out_g2, &out_g1[2], &out_g1[3], &out_g1[1], &out_g1[0]};
    int abc_left = 3;
    for (int k = 0; k < 5 && rc == ZK_OK; k++) {
        rc = zk_msm_finish(ctx, J[ord[k]], outs[ord[k]]);
        if (ord[k] <= 2 && --abc_left == 0 && rc == ZK_OK && after_abc) after_abc();   // A, B1, B2 are in: the caller's host work overlaps the rest
    }
    if (!front_enqueued) {                         // (with a front in flight the streams carry the next proof's kernels)
        (void)hipStreamSynchronize(s_sort);
        (void)hipStreamSynchronize(s_acc);
        (void)hipStreamSynchronize(s_red);
        (void)hipStreamSynchronize(ctx->stream);
    }
    tm.resolve();
    return rc;
}

}  // namespace

// Several independent MSMs as a software pipeline over the sort / accumulate / reduce streams, enqueued in one go: three
// scratch slots rotate, a job's sort waits (on the device) for the reduce of the slot's previous user, so the sort stream runs up
// to two jobs ahead of the accumulate stream and the host only waits at the end.  Jobs run longest first: the head of the
// pipeline (one sort nothing hides) is paid once either way, and behind a long accumulate kernel the shorter jobs' sorts are
// ready in time -- in submission order the round-1 batch of Marlin (n, n, n, 3n) left the accumulate stream waiting ~1.5 ms
// for the 3n job's sort, which takes 3.3 ms beside an accumulate kernel (0.45 ms alone).
// Used for the commitments of one Marlin round (lib.rs:171-247: PC::commit over the round's oracles) and for the two MSMs of
// SpdzGroupShare::multi_scale_pub_group (share/spdz.rs:482-488).  Outputs are Jacobian points (G1 or G2 according to each job's table).
extern "C" int zk_msm_batch_dev(zk_ctx* ctx, size_t n_jobs, const zk_bases* const* bases, const size_t* base_offsets,
                                const void* const* scalars_dev, const size_t* lens, void* const* outs) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (n_jobs && (!bases || !scalars_dev || !lens || !outs))) return ZK_ERR_ARG;
    for (size_t k = 0; k < n_jobs; k++) {
        if (!bases[k] || !outs[k] || (lens[k] && !scalars_dev[k])) return ZK_ERR_ARG;
        const size_t off = base_offsets ? base_offsets[k] : 0;
        if (off + lens[k] > bases[k]->n) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_msm_batch_dev: a job reads past its base table");
    }
    if (n_jobs == 0) return ZK_OK;
    zk_presort_free(ctx);            // the batch rotates over the same scratch slots
    ZK_TRY(ensure_aux(ctx, 1));
    constexpr size_t SLOTS = 3;
    static const bool serial_sort = getenv("ZK_BATCH_SERIAL_SORT") != nullptr;      // experiment: sorts in line with the accumulates
    hipStream_t s_acc = ctx->acc_stream, s_sort = serial_sort ? s_acc : ctx->aux[0];
    hipEvent_t e0;
    ZK_HIP(ctx, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    ZK_HIP(ctx, hipEventRecord(e0, ctx->stream));        // the scalars were produced on the context stream
    ZK_HIP(ctx, hipStreamWaitEvent(s_sort, e0, 0));
    ZK_HIP(ctx, hipStreamWaitEvent(s_acc, e0, 0));
    std::vector<size_t> perm(n_jobs);
    for (size_t k = 0; k < n_jobs; k++) perm[k] = k;
    static const bool in_order = getenv("ZK_BATCH_IN_ORDER") != nullptr;      // experiment: submission order
    if (!in_order) std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return lens[a] > lens[b]; });
    std::vector<ZkMsmJob> jobs(n_jobs);
    std::vector<int> owner(n_jobs), sharer(n_jobs, -1);          // whose sort a job uses (itself unless it borrows), and who borrows a job's sort
    for (size_t k = 0; k < n_jobs; k++) owner[k] = (int)k;
    static const bool share_sorts = !(getenv("ZK_BATCH_SHARE_SORT") && atoi(getenv("ZK_BATCH_SHARE_SORT")) == 0);
    int rc = ZK_OK;
    static const bool red_on_main = !(getenv("ZK_BATCH_REDUCE_STREAM") && !strcmp(getenv("ZK_BATCH_REDUCE_STREAM"), "sort"));
    for (size_t k = 0; k < n_jobs && rc == ZK_OK; k++) {
        const size_t j = perm[k];
        jobs[k].pin_key = 16 + (int)k;                   // its own pinned result buffer: the host reads them all at the end
        rc = zk_msm_prepare(ctx, &jobs[k], bases[j], base_offsets ? base_offsets[j] : 0, scalars_dev[j], lens[j], 1 + (int)(k % SLOTS));
        // the slot's previous user must be through its reduce chain (k_fold reads the sort scratch, the chain the sums) before
        // this job's sort rewrites the slot; that job's accumulate kernel is then done as well
        if (rc == ZK_OK && k >= SLOTS && jobs[k - SLOTS].reduce_done) ZK_HIP(ctx, hipStreamWaitEvent(s_sort, jobs[k - SLOTS].reduce_done, 0));
        // ... and so must a job that borrowed that user's sort (its accumulate kernel reads the sorted entries, its k_fold the
        // segment tables of the slot)
        if (rc == ZK_OK && k >= SLOTS && sharer[k - SLOTS] >= 0 && jobs[sharer[k - SLOTS]].reduce_done)
            ZK_HIP(ctx, hipStreamWaitEvent(s_sort, jobs[sharer[k - SLOTS]].reduce_done, 0));
        // the same scalar vector as the previous job of the batch (a degree-bounded oracle's commitment and its shifted copy):
        // one sort for both
        const ZkMsmJob* share = nullptr;
        if (rc == ZK_OK && k > 0 && share_sorts && scalars_dev[j] == scalars_dev[perm[k - 1]] && lens[j] == lens[perm[k - 1]] &&
            owner[k - 1] == (int)(k - 1)) {
            share = &jobs[k - 1];
            owner[k] = (int)(k - 1);
            sharer[k - 1] = (int)k;
        }
        if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, &jobs[k], s_sort, share);
        if (rc == ZK_OK) rc = zk_msm_enqueue_accum(ctx, &jobs[k], s_acc);
        // reduces on the context stream (idle here), not behind the sorts: on the sort stream the sort of job k+2 queued
        // behind the reduce of job k, i.e. behind the accumulate of job k, and the accumulate stream then waited for it
        // (period = reduce + sort beside a running accumulate ~ 4 ms per 2^20-scalar job instead of the accumulate's 2.1 ms).
        // ZK_BATCH_REDUCE_STREAM=sort restores the old placement.
        if (rc == ZK_OK) rc = zk_msm_enqueue_reduce(ctx, &jobs[k], red_on_main ? ctx->stream : s_sort);
    }
    for (size_t k = 0; k < n_jobs && rc == ZK_OK; k++) rc = zk_msm_finish(ctx, &jobs[k], outs[perm[k]]);
    (void)hipStreamSynchronize(s_sort);
    (void)hipStreamSynchronize(s_acc);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(e0);
    return rc;
    ZK_API_END
}

// The next zk_groth16_prove_dev on this context will be for `z_next_dev` (same key, same constraint system): the proof
// in between enqueues that proof's front (z-sort, witness map, H-sort) behind its own kernels.  Pass NULL to withdraw.
extern "C" int zk_groth16_hint_next_dev(zk_ctx* ctx, const void* z_next_dev) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ctx->next_z = z_next_dev;
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_groth16_msms_presort_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z) return ZK_ERR_ARG;
    zk_presort_free(ctx);
    const size_t nvars = (r->ni - 1) + r->nw;
    if (pk->b_g2->n != nvars + 1) ZK_FAIL(ctx, ZK_ERR_ARG, "groth16: proving key does not match the constraint system");
    ZK_TRY(ensure_aux(ctx, 1));
    hipEvent_t e0;
    ZK_HIP(ctx, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    ZK_HIP(ctx, hipEventRecord(e0, ctx->stream));          // z was produced on the context stream
    ZK_HIP(ctx, hipStreamWaitEvent(ctx->aux[0], e0, 0));
    (void)hipEventDestroy(e0);
    std::unique_ptr<ZkPresort> p(new ZkPresort());
    p->pk = pk;
    p->z = z;
    ZK_TRY(zk_msm_prepare(ctx, &p->job, pk->b_g2, 1, (const char*)z + 32, nvars, 1));
    ZK_TRY(zk_msm_enqueue_sort(ctx, &p->job, ctx->aux[0], nullptr));
    ctx->presort = p.release();
    return ZK_OK;
    ZK_API_END
}

// The four MSMs over z -- B in G2, A, B in G1, L -- enqueued to the end (sort, accumulate, reduce); returns at once.
// zk_groth16_msms_dev(ctx, pk, r, z, h, ...) with the same pk / r / z then only adds the H job and collects the five results.
// For the collaborative prover (mpc.py::create_proof_shared): the Beaver open of the witness map's product and the second
// half of the witness map run under accumulate kernels that do not need h.  The context stream stays free for the caller's
// own kernels.
extern "C" int zk_groth16_msms_begin_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z) return ZK_ERR_ARG;
    zk_presort_free(ctx);
    const size_t nvars = (r->ni - 1) + r->nw;
    if (pk->a->n != nvars + 1 || pk->b_g1->n != nvars + 1 || pk->b_g2->n != nvars + 1 || pk->l->n != r->nw)
        ZK_FAIL(ctx, ZK_ERR_ARG, "groth16: proving key does not match the constraint system");
    ZK_TRY(ensure_aux(ctx, 1));
    hipStream_t s_sort = ctx->aux[0], s_acc = ctx->acc_stream;
    hipEvent_t e0;
    ZK_HIP(ctx, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    ZK_HIP(ctx, hipEventRecord(e0, ctx->stream));          // z was produced on the context stream
    ZK_HIP(ctx, hipStreamWaitEvent(s_sort, e0, 0));
    ZK_HIP(ctx, hipStreamWaitEvent(s_acc, e0, 0));
    (void)hipEventDestroy(e0);
    std::unique_ptr<ZkPresort> p(new ZkPresort());
    p->pk = pk; p->z = z; p->r = r; p->begun = true;
    const char* zb = (const char*)z;
    const bool l_shared = pk->l_pad && pk->l_pad->n == nvars + 1 && (pk->l_pad->pre != nullptr) == (pk->a->pre != nullptr) &&
                          pk->l_pad->c_pre == pk->a->c_pre;
    ZkMsmJob* J[4] = {&p->job, &p->j1, &p->j2, &p->j3};
    int rc = zk_msm_prepare(ctx, J[0], pk->b_g2, 1, zb + 32, nvars, 1);
    if (rc == ZK_OK) rc = zk_msm_prepare(ctx, J[1], pk->a, 1, zb + 32, nvars, 2);
    if (rc == ZK_OK) rc = zk_msm_prepare(ctx, J[2], pk->b_g1, 1, zb + 32, nvars, 3);
    if (rc == ZK_OK) rc = l_shared ? zk_msm_prepare(ctx, J[3], pk->l_pad, 1, zb + 32, nvars, 4)
                                   : zk_msm_prepare(ctx, J[3], pk->l, 0, zb + r->ni * 32, r->nw, 4);
    if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, J[0], s_sort, nullptr);
    if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, J[1], s_sort, J[0]);
    if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, J[2], s_sort, J[0]);
    if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, J[3], s_sort, l_shared ? J[0] : nullptr);
    for (int k = 0; k < 4 && rc == ZK_OK; k++) rc = zk_msm_enqueue_accum(ctx, J[k], s_acc);
    // only the G2 job's reduce chain here (sort stream); the G1 chains are enqueued by zk_groth16_msms_dev, which spreads them over
    // the context stream (the caller's own kernels are through by then) and the sort stream as the one-call form does
    if (rc == ZK_OK) rc = zk_msm_enqueue_reduce(ctx, J[0], s_sort);
    if (rc != ZK_OK) {                                   // whatever was enqueued drains before the jobs (and their events) go
        (void)hipStreamSynchronize(s_sort);
        (void)hipStreamSynchronize(s_acc);
        return rc;
    }
    ctx->presort = p.release();
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_groth16_msms_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z, const void* h,
                                   zk_g1_projective out_g1[4], zk_g2_projective* out_g2) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z || !h || !out_g1 || !out_g2) return ZK_ERR_ARG;
    return run_msms(ctx, pk, r, z, h, nullptr, out_g1, out_g2);
    ZK_API_END
}

extern "C" int zk_groth16_prove_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z, const zk_fr* r_, const zk_fr* s_,
                                    uint8_t proof[192]) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z || !r_ || !s_ || !proof) return ZK_ERR_ARG;
    const size_t D = (size_t)1 << r->log_d;
    void* h;
    ZK_TRY(zk_scratch(ctx, "prove_h", D * 32, &h));
    zk_g1_projective m1[4];
    zk_g2_projective m2;

    // ---- O(1) tail on the host, 64-bit limbs.  Everything that depends only on the A, B-in-G1 and B-in-G2 sums (and on
    // r, s, the key) starts as soon as those three jobs have delivered, while the GPU still works on L and H ----
    using H1 = Fq64Field;
    using H2 = Fq264Field;
    using X1 = XYZZ<H1>;
    using X2 = XYZZ<H2>;
    uint32_t rw[8], sw[8];
    fr_abi_to_canon_words(r_->l, rw);
    fr_abi_to_canon_words(s_->l, sw);
    const X1 delta1 = xyzz_from_affine<H1>(aff_to_host64<G1Field>(pk->delta_g1));
    const X2 delta2 = xyzz_from_affine<H2>(aff_to_host64<G2Field>(pk->delta_g2));
    const Affine<H1> a0 = aff_to_host64<G1Field>(pk->a0), alpha = aff_to_host64<G1Field>(pk->alpha_g1);
    const Affine<H1> b0 = aff_to_host64<G1Field>(pk->b0_g1), beta1 = aff_to_host64<G1Field>(pk->beta_g1);
    const Affine<H2> b02 = aff_to_host64<G2Field>(pk->b0_g2), beta2 = aff_to_host64<G2Field>(pk->beta_g2);
    X1 g_a, s_g_a, r_s_delta, r_g1_b;
    Affine<H2> b_aff;
    ZkTask<void> chain_a, chain_b, chain_g2;         // (declared after everything the chains write: joined before those go)
    std::chrono::steady_clock::time_point t_tail;
    auto after_abc = [&]() {
        const X1 a_acc = host64_proj_from_abi<H1>((const uint64_t*)&m1[2]);
        const X1 b1_acc = host64_proj_from_abi<H1>((const uint64_t*)&m1[3]);
        const X2 b2_acc = host64_proj_from_abi<H2>((const uint64_t*)&m2);
        // calculate_coeff (src/groth16.rs:185-201): initial + query[0] + acc + vk_param
        chain_a = zk_async(ctx, [&, a_acc] {
            const X1 r_g1 = host64_scalar_mul<H1>(delta1, rw);
            r_s_delta = host64_scalar_mul<H1>(r_g1, sw);                                                     // :115
            g_a = xyzz_madd<H1>(xyzz_add<H1>(xyzz_madd<H1>(r_g1, a0), a_acc), alpha);
            s_g_a = host64_scalar_mul<H1>(g_a, sw);                                                           // :140
        });
        chain_b = zk_async(ctx, [&, b1_acc] {
            const X1 s_g1 = host64_scalar_mul<H1>(delta1, sw);
            const X1 g1_b = xyzz_madd<H1>(xyzz_add<H1>(xyzz_madd<H1>(s_g1, b0), b1_acc), beta1);
            r_g1_b = host64_scalar_mul<H1>(g1_b, rw);                                                         // :161
        });
        chain_g2 = zk_async(ctx, [&, b2_acc] {
            const X2 s_g2 = host64_scalar_mul<H2>(delta2, sw);
            const X2 g2_b = xyzz_madd<H2>(xyzz_add<H2>(xyzz_madd<H2>(s_g2, b02), b2_acc), beta2);
            b_aff = xyzz_to_affine<H2>(g2_b);
        });
    };
    int rc_msm = run_msms(ctx, pk, r, z, nullptr, h, m1, &m2, after_abc);
    t_tail = std::chrono::steady_clock::now();      // what is left of the host work once the GPU is done
    if (chain_a.valid()) chain_a.get();
    if (chain_b.valid()) chain_b.get();
    if (chain_g2.valid()) chain_g2.get();
    ZK_TRY(rc_msm);
    const X1 h_acc = host64_proj_from_abi<H1>((const uint64_t*)&m1[0]);
    const X1 l_acc = host64_proj_from_abi<H1>((const uint64_t*)&m1[1]);
    X1 g_c = xyzz_add<H1>(s_g_a, r_g1_b);                                                                 // :169-174
    g_c = xyzz_add<H1>(g_c, xyzz_neg<H1>(r_s_delta));
    g_c = xyzz_add<H1>(g_c, l_acc);
    g_c = xyzz_add<H1>(g_c, h_acc);

    g1_serialize(aff_from_host64<G1Field>(xyzz_to_affine<H1>(g_a)), proof);
    g2_serialize(aff_from_host64<G2Field>(b_aff), proof + 48);
    g1_serialize(aff_from_host64<G1Field>(xyzz_to_affine<H1>(g_c)), proof + 144);
    if (ctx->profiling) {
        auto& t = ctx->timers["host.tail"];
        t.ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_tail).count();
        t.count += 1;
    }
    return ZK_OK;
    ZK_API_END
}

// Host-slice prover for a queue: `z_host` is proved now, `z_next_host` (or NULL) is the assignment of the next call.  The
// next assignment is uploaded at once on a copy stream into the one of two device slots this proof does not read, and this
// proof enqueues its front exactly as zk_groth16_hint_next_dev does.  When the next call names the announced buffer (matched
// by address: it must stay unchanged in between) nothing is copied again.  Page-locked memory (zk_host_alloc) makes the
// upload asynchronous.
extern "C" int zk_groth16_prove_queued(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const zk_fr* z_host, const zk_fr* r_,
                                       const zk_fr* s_, const zk_fr* z_next_host, uint8_t proof[192]) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z_host) return ZK_ERR_ARG;
    const size_t m = r->ni + r->nw;
    void* z;
    if (ctx->next_z_host == z_host && ctx->next_z_dev && ctx->next_z_pk == pk && ctx->next_z_r == r && ctx->next_z_m == m) {
        // announced by the previous call for this key, this system and this length: already uploaded (the front that reads it
        // waited for the copy)
        z = ctx->next_z_dev;
        ctx->z_slot ^= 1;
        ZK_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->next_z_ready, 0));
    } else {
        // Not the announced assignment (or announced for another key / system / length: the upload may be shorter than m).  A
        // front enqueued for the announced one still reads the other slot, and its z-sort / witness map may run for as long as
        // this proof takes: let it drain before either slot is written (run_msms would only drop it later).
        zk_presort_free(ctx);
        if (ctx->copy_stream) ZK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        ZK_TRY(zk_scratch(ctx, ctx->z_slot ? "prove_z1" : "prove_z0", m * 32, &z));
        ZK_HIP(ctx, hipMemcpyAsync(z, z_host, m * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->next_z_host = nullptr;
    ctx->next_z_dev = nullptr;
    ctx->next_z = nullptr;
    ctx->next_z_pk = ctx->next_z_r = nullptr;
    ctx->next_z_m = 0;
    if (z_next_host) {
        if (!ctx->copy_stream) ZK_HIP(ctx, zk_stream_create(&ctx->copy_stream, false));
        if (!ctx->next_z_ready) ZK_HIP(ctx, hipEventCreateWithFlags(&ctx->next_z_ready, hipEventDisableTiming));
        void* zn;
        // the other slot: its last reader was the previous proof, which has delivered its bytes
        ZK_TRY(zk_scratch(ctx, ctx->z_slot ? "prove_z0" : "prove_z1", m * 32, &zn));
        ZK_HIP(ctx, hipMemcpyAsync(zn, z_next_host, m * 32, hipMemcpyHostToDevice, ctx->copy_stream));
        ZK_HIP(ctx, hipEventRecord(ctx->next_z_ready, ctx->copy_stream));
        ctx->next_z_host = z_next_host;
        ctx->next_z_dev = zn;
        ctx->next_z = zn;
        ctx->next_z_pk = pk; ctx->next_z_r = r; ctx->next_z_m = m;
    }
    return zk_groth16_prove_dev(ctx, pk, r, z, r_, s_, proof);
    ZK_API_END
}

extern "C" int zk_groth16_prove(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const zk_fr* z_host, const zk_fr* r_, const zk_fr* s_,
                                uint8_t proof[192]) {
    ZK_API_BEGIN(ctx)
    return zk_groth16_prove_queued(ctx, pk, r, z_host, r_, s_, nullptr, proof);
    ZK_API_END
}

// ---- the collaborative prover as ONE entry point ---------------------------------------------------------------------------------
// create_proof::<MpcPairingEngine, C> over additive shares (src/groth16.rs:68-183): what mpc.py::Party.create_proof_shared
// sequences from ~40 calls, for a host that cannot call Python.  The transport is the caller's (zk_net_vtable: the reference's
// MpcNet::broadcast_bytes on its TCP mesh, mpc-net/src/lib.rs:60-64) for the small opens; the two vector opens of the Beaver
// product go through the vtable's open_sum_fr_dev or, when that is NULL, through the context's own RCCL communicator
// (zk_open_sum_fr_dev).  Opens happen in the fused order of create_proof_shared: the nine small opens of the three
// GroupShare::scale calls (share/group.rs:72-111, DummyGroupTripleSource) and of Proof::reveal travel in two exchanges; every
// opened value is the reference's, and so are the 192 bytes.
namespace {

struct FrK { uint32_t l[9]; };
__device__ __forceinline__ Fr frk(const FrK& k) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = k.l[i];
    return r;
}
FrK to_frk(const Fr& a) {
    FrK k;
    for (int i = 0; i < 9; i++) k.l[i] = a.l[i];
    return k;
}
// the 256-bit word of a field element is below the modulus (is_valid, ff/src/fields/macros.rs:255-260): what arrives from a peer
bool fr_abi_valid(const uint64_t l[4]) {
    Fr m;
    for (int i = 0; i < 9; i++) m.l[i] = FrParams::P[i];
    uint64_t pm[4];
    host_store_ext<FrParams>(pm, m);
    for (int i = 3; i >= 0; i--) {
        if (l[i] < pm[i]) return true;
        if (l[i] > pm[i]) return false;
    }
    return false;
}

__global__ void __launch_bounds__(256) k_vec_add_const(const void* a, FrK k, void* out, size_t n) {
    const Fr kk = frk(k);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        fr_store(out, i, fr_add(fr_load(a, i), kk));
}

}  // namespace

namespace {

__global__ void __launch_bounds__(256) k_vec_neg(const void* a, void* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        fr_store(out, i, fr_sub(fp_zero<FrParams>(), fr_load(a, i)));
}

}  // namespace

int zk_shared_spdz_open_vec(ZkSharedNet& nt, const void* sh, const void* mac, size_t n, void* out, void* dx) {
    zk_ctx* ctx = nt.ctx;
    ZK_TRY(nt.open_vec(sh, n, out));
    if (ctx->party_id == 0) {
        ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_SUB, out, mac, dx, n));
    } else {
        hipLaunchKernelGGL(k_vec_neg, zk_grid(n, 256), 256, 0, ctx->stream, mac, dx, n);
        ZK_HIP(ctx, hipGetLastError());
    }
    ZK_TRY(nt.open_vec(dx, n, dx));
    int zero = 0;
    ZK_TRY(zk_fr_vec_is_zero_dev(ctx, dx, n, &zero));
    if (!zero) ZK_FAIL(ctx, ZK_ERR_MAC, "SPDZ MAC check failed on a vector open");
    return ZK_OK;
}

int zk_shared_beaver_mul(ZkSharedNet& nt, int lanes, const void* const x[2], const void* const y[2], void* const out[2], size_t n,
                         const void* const tx[2], const void* const ty[2], const void* const tz[2], const char* tag) {
    zk_ctx* ctx = nt.ctx;
    const bool dummy = !tx[0] && !ty[0] && !tz[0];
    for (int l = 0; l < lanes; l++)
        if (dummy ? (tx[l] || ty[l] || tz[l]) : (!tx[l] || !ty[l] || !tz[l])) ZK_FAIL(ctx, ZK_ERR_ARG, "batch_mul: give a whole Beaver triple (every lane) or none");
    void *sxl[2], *oyl[2], *sx, *oy, *dx;
    char nm[64];
    auto buf = [&](const char* what, int l, void** p) { snprintf(nm, sizeof nm, "%s.%s%d", tag, what, l); return zk_scratch(ctx, nm, n * 32, p); };
    for (int l = 0; l < lanes; l++) { ZK_TRY(buf("sxl", l, &sxl[l])); ZK_TRY(buf("oyl", l, &oyl[l])); }
    ZK_TRY(buf("sx", 0, &sx)); ZK_TRY(buf("oy", 0, &oy)); ZK_TRY(buf("dx", 0, &dx));
    const Fr one_ext = fp_mul<FrParams>(fp_one<FrParams>(), fp_const<FrParams>(FrParams::INT_TO_EXT));
    for (int l = 0; l < lanes; l++) {
        if (dummy) {                                      // DummyFieldTripleSource: the leader holds 1 (in both lanes), the rest 0 (wire/field.rs:49-63)
            if (nt.leader()) {
                hipLaunchKernelGGL(k_vec_add_const, zk_grid(n, 256), 256, 0, ctx->stream, x[l], to_frk(one_ext), sxl[l], n);
                hipLaunchKernelGGL(k_vec_add_const, zk_grid(n, 256), 256, 0, ctx->stream, y[l], to_frk(one_ext), oyl[l], n);
                ZK_HIP(ctx, hipGetLastError());
            } else {
                ZK_HIP(ctx, hipMemcpyAsync(sxl[l], x[l], n * 32, hipMemcpyDeviceToDevice, ctx->stream));
                ZK_HIP(ctx, hipMemcpyAsync(oyl[l], y[l], n * 32, hipMemcpyDeviceToDevice, ctx->stream));
            }
        } else {
            ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_ADD, x[l], tx[l], sxl[l], n));
            ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_ADD, y[l], ty[l], oyl[l], n));
        }
    }
    if (lanes == 1) {
        ZK_TRY(nt.open_vec(sxl[0], n, sx));               // open(s + x), open(o + y)
        ZK_TRY(nt.open_vec(oyl[0], n, oy));
    } else {
        ZK_TRY(zk_shared_spdz_open_vec(nt, sxl[0], sxl[1], n, sx, dx));
        ZK_TRY(zk_shared_spdz_open_vec(nt, oyl[0], oyl[1], n, oy, dx));
    }
    // the local tail; the shift of sx * oy lands on the leader in BOTH lanes (mac_share = 1 there)
    for (int l = 0; l < lanes; l++) ZK_TRY(zk_beaver_combine_dev(ctx, sx, oy, tx[l], ty[l], tz[l], out[l], n));
    return ZK_OK;
}

namespace {

using SH1 = Fq64Field;
using SH2 = Fq264Field;
using SX1 = XYZZ<SH1>;
using SX2 = XYZZ<SH2>;

// LANES = 1: additive shares (AdditiveFieldShare / AdditiveGroupShare); LANES = 2: SPDZ (share lane, MAC lane), every open
// MAC-checked.  z / rs / ss / tx..tz are indexed by lane.
template <int LANES>
int prove_shared_impl(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* const z[2], const zk_fr* const rs[2],
                      const zk_fr* const ss[2], const void* const tx[2], const void* const ty[2], const void* const tz[2],
                      const zk_net_vtable* net, uint8_t proof[192], uint64_t* bytes_sent) {
    const bool dummy = !tx[0] && !ty[0] && !tz[0];
    for (int l = 0; l < LANES; l++)
        if (dummy ? (tx[l] || ty[l] || tz[l]) : (!tx[l] || !ty[l] || !tz[l])) ZK_FAIL(ctx, ZK_ERR_ARG, "prove_shared: give a whole Beaver triple (every lane) or none");
    const bool leader = ctx->party_id == 0;
    const size_t D = (size_t)1 << r->log_d;
    ZkSharedNet nt{ctx, net};
    uint32_t rw[2][8], sw[2][8];
    for (int l = 0; l < LANES; l++) { fr_abi_to_canon_words(rs[l]->l, rw[l]); fr_abi_to_canon_words(ss[l]->l, sw[l]); }
    const SX1 delta1 = xyzz_from_affine<SH1>(aff_to_host64<G1Field>(pk->delta_g1));
    const SX2 delta2 = xyzz_from_affine<SH2>(aff_to_host64<G2Field>(pk->delta_g2));
    // public point x shared scalar: local host arithmetic (scale_pub_group, share/additive.rs:502-508), under the device work
    ZkTask<SX1> f_r_g1[2], f_s_g1[2];                  // (joining handles: the tasks reference this frame and are waited for when it is left)
    ZkTask<SX2> f_s_g2[2];
    for (int l = 0; l < LANES; l++) {
        f_r_g1[l] = zk_async(ctx, [&, l] { return host64_scalar_mul<SH1>(delta1, rw[l]); });
        f_s_g1[l] = zk_async(ctx, [&, l] { return host64_scalar_mul<SH1>(delta1, sw[l]); });
        f_s_g2[l] = zk_async(ctx, [&, l] { return host64_scalar_mul<SH2>(delta2, sw[l]); });
    }
    void *a[2], *b[2], *c[2];
    char nm[32];
    for (int l = 0; l < LANES; l++) {
        const char* names[3] = {"shared_a%d", "shared_b%d", "shared_c%d"};
        void** dst[3] = {&a[l], &b[l], &c[l]};
        for (int k = 0; k < 3; k++) { snprintf(nm, sizeof nm, names[k], l); ZK_TRY(zk_scratch(ctx, nm, D * 32, dst[k])); }
    }
    for (int l = 0; l < LANES; l++) ZK_TRY(zk_groth16_witness_map_pre_dev(ctx, r, z[l], 1, a[l], b[l], c[l]));     // local: linear in the shares
    ZK_TRY(zk_groth16_msms_begin_dev(ctx, pk, r, z[0]));                          // the share lane's four MSMs over z run under the opens
    // FieldShare::batch_mul of the D-element product (share/field.rs:97-129): open(s + x), open(o + y), the local tail -- per lane
    {
        const void* xa[2] = {a[0], LANES == 2 ? a[1] : nullptr};
        const void* yb[2] = {b[0], LANES == 2 ? b[1] : nullptr};
        void* oa[2] = {a[0], LANES == 2 ? a[1] : nullptr};
        ZK_TRY(zk_shared_beaver_mul(nt, LANES, xa, yb, oa, D, tx, ty, tz, "shared_bv"));
    }
    zk_g1_projective m1[2][4];
    zk_g2_projective m2[2];
    for (int l = 0; l < LANES; l++) {
        ZK_TRY(zk_groth16_witness_map_post_dev(ctx, r, a[l], c[l]));                    // h shares in a[l]
        ZK_TRY(zk_groth16_msms_dev(ctx, pk, r, z[l], a[l], m1[l], &m2[l]));            // party-local MSMs (multi_scale_pub_group; spdz.rs:482-488: twice)
    }
    auto pub1 = [&](const Affine<G1Field>& p) { return leader ? xyzz_from_affine<SH1>(aff_to_host64<G1Field>(p)) : xyzz_inf<SH1>(); };   // shift(): leader only
    auto pub2 = [&](const Affine<G2Field>& p) { return leader ? xyzz_from_affine<SH2>(aff_to_host64<G2Field>(p)) : xyzz_inf<SH2>(); };
    SX1 h_acc[2], l_acc[2], r_g1[2], g_a[2], g1_b[2];
    SX2 g2_b[2];
    for (int l = 0; l < LANES; l++) {
        h_acc[l] = host64_proj_from_abi<SH1>((const uint64_t*)&m1[l][0]);
        l_acc[l] = host64_proj_from_abi<SH1>((const uint64_t*)&m1[l][1]);
        const SX1 a_acc = host64_proj_from_abi<SH1>((const uint64_t*)&m1[l][2]), b1_acc = host64_proj_from_abi<SH1>((const uint64_t*)&m1[l][3]);
        const SX2 b2_acc = host64_proj_from_abi<SH2>((const uint64_t*)&m2[l]);
        r_g1[l] = f_r_g1[l].get();
        g_a[l] = xyzz_add<SH1>(xyzz_add<SH1>(xyzz_add<SH1>(r_g1[l], pub1(pk->a0)), a_acc), pub1(pk->alpha_g1));                 // calculate_coeff (:185-201)
        g1_b[l] = xyzz_add<SH1>(xyzz_add<SH1>(xyzz_add<SH1>(f_s_g1[l].get(), pub1(pk->b0_g1)), b1_acc), pub1(pk->beta_g1));
        g2_b[l] = xyzz_add<SH2>(xyzz_add<SH2>(xyzz_add<SH2>(f_s_g2[l].get(), pub2(pk->b0_g2)), b2_acc), pub2(pk->beta_g2));
    }
    // first exchange: open(o + y) for o = s, r (y = the leader's 1: the dummy group triple; from_add_shared: mac = share), open(s + x)
    // for the three scaled points (x = 0) and the reveal of B.  SPDZ: a second exchange of [leader ? opened : 0] - mac, all zero.
    constexpr size_t MW = 2 * 4 + 3 * 18 + 36;
    const Fr y = leader ? fp_mul<FrParams>(fp_one<FrParams>(), fp_const<FrParams>(FrParams::INT_TO_EXT)) : fp_zero<FrParams>();
    auto pack = [&](int l, uint64_t* msg) {
        host_store_ext<FrParams>(msg, fp_add<FrParams>(host_load_ext<FrParams>(ss[l]->l), y));
        host_store_ext<FrParams>(msg + 4, fp_add<FrParams>(host_load_ext<FrParams>(rs[l]->l), y));
        host64_write_projective<SH1>(xyzz_to_affine<SH1>(r_g1[l]), msg + 8);
        host64_write_projective<SH1>(xyzz_to_affine<SH1>(g_a[l]), msg + 26);
        host64_write_projective<SH1>(xyzz_to_affine<SH1>(g1_b[l]), msg + 44);
        host64_write_projective<SH2>(xyzz_to_affine<SH2>(g2_b[l]), msg + 62);
    };
    struct Opened { Fr f[2]; SX1 g[3]; SX2 B; };
    auto sum = [&](const std::vector<uint8_t>& all, Opened& o) -> int {
        o.f[0] = o.f[1] = fp_zero<FrParams>();
        o.g[0] = o.g[1] = o.g[2] = xyzz_inf<SH1>();
        o.B = xyzz_inf<SH2>();
        for (int p = 0; p < nt.parties(); p++) {
            uint64_t w[MW];
            memcpy(w, all.data() + (size_t)p * sizeof w, sizeof w);
            for (int k = 0; k < 2; k++) {
                if (!fr_abi_valid(w + 4 * k)) ZK_FAIL(ctx, ZK_ERR_STATE, "prove_shared: a party sent a non-canonical field element");
                o.f[k] = fp_add<FrParams>(o.f[k], host_load_ext<FrParams>(w + 4 * k));
            }
            // points from a peer: canonical coordinates, on the curve; the malicious-security prover also checks the subgroup
            // (GroupAffine::deserialize behind MpcSerNet::broadcast, channel.rs:12-28).  A party's own message is its own output.
            bool pts_ok = true;
            const bool peer = p != ctx->party_id;
            for (int k = 0; k < 3; k++)
                o.g[k] = xyzz_add<SH1>(o.g[k], peer ? host64_peer_point<SH1>(w + 8 + 18 * k, LANES == 2, pts_ok) : host64_proj_from_abi<SH1>(w + 8 + 18 * k));
            o.B = xyzz_add<SH2>(o.B, peer ? host64_peer_point<SH2>(w + 62, LANES == 2, pts_ok) : host64_proj_from_abi<SH2>(w + 62));
            if (!pts_ok) ZK_FAIL(ctx, ZK_ERR_STATE, "prove_shared: a party sent a point that is not a valid group element");
        }
        return ZK_OK;
    };
    uint64_t msg[MW];
    std::vector<uint8_t> all;
    Opened op;
    pack(0, msg);
    ZK_TRY(nt.gather((const uint8_t*)msg, sizeof msg, all));
    ZK_TRY(sum(all, op));
    if (LANES == 2) {
        // [leader ? x : 0] - mac for every opened value, published and summed: SpdzFieldShare / SpdzGroupShare batch_open's check
        uint64_t mm[MW], dm[MW];
        pack(1, mm);
        for (int k = 0; k < 2; k++)
            host_store_ext<FrParams>(dm + 4 * k, fp_sub<FrParams>(leader ? op.f[k] : fp_zero<FrParams>(), host_load_ext<FrParams>(mm + 4 * k)));
        for (int k = 0; k < 3; k++)
            host64_write_projective<SH1>(xyzz_to_affine<SH1>(xyzz_add<SH1>(leader ? op.g[k] : xyzz_inf<SH1>(), xyzz_neg<SH1>(host64_proj_from_abi<SH1>(mm + 8 + 18 * k)))), dm + 8 + 18 * k);
        host64_write_projective<SH2>(xyzz_to_affine<SH2>(xyzz_add<SH2>(leader ? op.B : xyzz_inf<SH2>(), xyzz_neg<SH2>(host64_proj_from_abi<SH2>(mm + 62)))), dm + 62);
        Opened chk;
        ZK_TRY(nt.gather((const uint8_t*)dm, sizeof dm, all));
        ZK_TRY(sum(all, chk));
        bool ok = fp_is_zero<FrParams>(chk.f[0]) && fp_is_zero<FrParams>(chk.f[1]) && xyzz_is_inf<SH2>(chk.B);
        for (int k = 0; k < 3; k++) ok = ok && xyzz_is_inf<SH1>(chk.g[k]);
        if (!ok) ZK_FAIL(ctx, ZK_ERR_MAC, "SPDZ MAC check failed on a fused open");
    }
    const Fr oy_s = op.f[0], oy_r = op.f[1];
    const SX1 sx_rd = op.g[0], sx_a = op.g[1], sx_b = op.g[2];
    // the local part of GroupShare::scale behind its two opens: z - sx*y (+ sx*oy on the leader), z = 0, y = [leader]; both SPDZ
    // lanes hold the same value (key 1)
    auto scale_finish = [&](const SX1& sxp, const Fr& oyv) {
        if (!leader) return xyzz_inf<SH1>();
        uint64_t l4[4];
        uint32_t kw[8];
        host_store_ext<FrParams>(l4, oyv);
        fr_abi_to_canon_words(l4, kw);
        return xyzz_add<SH1>(host64_scalar_mul<SH1>(sxp, kw), xyzz_neg<SH1>(sxp));
    };
    auto p0 = zk_async(ctx, [&] { return scale_finish(sx_rd, oy_s); });       // r s delta            (:115)
    auto p1 = zk_async(ctx, [&] { return scale_finish(sx_a, oy_s); });        // s A                  (:140)
    const SX1 part2 = scale_finish(sx_b, oy_r);                          // r B1                 (:161)
    SX1 t = xyzz_add<SH1>(p1.get(), part2);
    t = xyzz_add<SH1>(t, xyzz_neg<SH1>(p0.get()));
    SX1 g_c[2];
    for (int l = 0; l < LANES; l++) g_c[l] = xyzz_add<SH1>(xyzz_add<SH1>(t, l_acc[l]), h_acc[l]);      // :169-174
    uint64_t cmsg[18];
    host64_write_projective<SH1>(xyzz_to_affine<SH1>(g_c[0]), cmsg);
    ZK_TRY(nt.gather((const uint8_t*)cmsg, sizeof cmsg, all));          // Proof::reveal of C (A and B were opened above)
    bool c_ok = true;
    auto sum_g1 = [&](const std::vector<uint8_t>& v) {
        SX1 acc = xyzz_inf<SH1>();
        for (int p = 0; p < nt.parties(); p++) {
            uint64_t w[18];
            memcpy(w, v.data() + (size_t)p * sizeof w, sizeof w);
            acc = xyzz_add<SH1>(acc, p != ctx->party_id ? host64_peer_point<SH1>(w, LANES == 2, c_ok) : host64_proj_from_abi<SH1>(w));
        }
        return acc;
    };
    const SX1 C = sum_g1(all);
    if (!c_ok) ZK_FAIL(ctx, ZK_ERR_STATE, "prove_shared: a party sent a point that is not a valid group element");
    if (LANES == 2) {
        host64_write_projective<SH1>(xyzz_to_affine<SH1>(xyzz_add<SH1>(leader ? C : xyzz_inf<SH1>(), xyzz_neg<SH1>(g_c[1]))), cmsg);
        ZK_TRY(nt.gather((const uint8_t*)cmsg, sizeof cmsg, all));
        const SX1 dsum = sum_g1(all);
        if (!c_ok) ZK_FAIL(ctx, ZK_ERR_STATE, "prove_shared: a party sent a point that is not a valid group element");
        if (!xyzz_is_inf<SH1>(dsum)) ZK_FAIL(ctx, ZK_ERR_MAC, "SPDZ MAC check failed on the reveal of C");
    }
    g1_serialize(aff_from_host64<G1Field>(xyzz_to_affine<SH1>(sx_a)), proof);
    g2_serialize(aff_from_host64<G2Field>(xyzz_to_affine<SH2>(op.B)), proof + 48);
    g1_serialize(aff_from_host64<G1Field>(xyzz_to_affine<SH1>(C)), proof + 144);
    if (bytes_sent) *bytes_sent = nt.bytes;
    return ZK_OK;
}

}  // namespace

extern "C" int zk_groth16_prove_shared(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z_share, const zk_fr* r_share,
                                       const zk_fr* s_share, const void* tx, const void* ty, const void* tz, const zk_net_vtable* net,
                                       uint8_t proof[192], uint64_t* bytes_sent) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z_share || !r_share || !s_share || !proof) return ZK_ERR_ARG;
    const void* z[2] = {z_share, nullptr};
    const zk_fr *rs[2] = {r_share, nullptr}, *ss[2] = {s_share, nullptr};
    const void *txs[2] = {tx, nullptr}, *tys[2] = {ty, nullptr}, *tzs[2] = {tz, nullptr};
    return prove_shared_impl<1>(ctx, pk, r, z, rs, ss, txs, tys, tzs, net, proof, bytes_sent);
    ZK_API_END
}

extern "C" int zk_groth16_prove_shared_spdz(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* const z_lanes[2],
                                            const zk_fr r_lanes[2], const zk_fr s_lanes[2], const void* const tx_lanes[2],
                                            const void* const ty_lanes[2], const void* const tz_lanes[2], const zk_net_vtable* net,
                                            uint8_t proof[192], uint64_t* bytes_sent) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z_lanes || !z_lanes[0] || !z_lanes[1] || !r_lanes || !s_lanes || !proof) return ZK_ERR_ARG;
    const zk_fr *rs[2] = {&r_lanes[0], &r_lanes[1]}, *ss[2] = {&s_lanes[0], &s_lanes[1]};
    const void* none[2] = {nullptr, nullptr};
    return prove_shared_impl<2>(ctx, pk, r, z_lanes, rs, ss, tx_lanes ? tx_lanes : none, ty_lanes ? ty_lanes : none,
                                tz_lanes ? tz_lanes : none, net, proof, bytes_sent);
    ZK_API_END
}

// ---- arkworks CanonicalSerialize framing of the Groth16 keys and of a KZG10 SRS --------------------------------------
// VerifyingKey  (arkworks/groth16/src/data_structures.rs:43-58):   alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | Vec gamma_abc_g1
// ProvingKey    (data_structures.rs:133-151):                      vk | beta_g1 | delta_g1 | Vec a_query | Vec b_g1_query |
//                                                                  Vec b_g2_query | Vec h_query | Vec l_query
// UniversalParams (poly-commit/src/kzg10/data_structures.rs:40-80; written by save_srs_to_file, src/marlin.rs:371-376):
//                 Vec powers_of_g | BTreeMap<usize, G1> powers_of_gamma_g | h | beta_h | BTreeMap<usize, G2> neg_powers_of_h
// Vec = u64 length + items (serialize/src/lib.rs:263-272), BTreeMap = u64 length + (u64 key, value) pairs (:740-754); points in
// the compressed or uncompressed form of zk_bases_serialize.  Point bytes are produced / parsed on the device.
namespace {

template <class F, class ABI>
int ser_small(zk_ctx* ctx, const Affine<F>* pts, size_t n, int group, int compressed, uint8_t* out) {
    std::vector<ABI> abi(n);
    for (size_t i = 0; i < n; i++) host_aff_to_abi<F>((uint64_t*)&abi[i], pts[i]);
    zk_bases* b = nullptr;
    int rc = group == 1 ? zk_bases_upload_g1(ctx, (const zk_g1_affine*)abi.data(), n, &b) : zk_bases_upload_g2(ctx, (const zk_g2_affine*)abi.data(), n, &b);
    if (rc == ZK_OK) rc = zk_bases_serialize(ctx, b, 0, n, compressed, out);
    zk_bases_free(ctx, b);
    return rc;
}
void put_u64(uint8_t*& p, uint64_t v) { for (int i = 0; i < 8; i++) *p++ = (uint8_t)(v >> (8 * i)); }
bool get_u64(const uint8_t*& p, const uint8_t* end, uint64_t* v) {
    if (end - p < 8) return false;
    *v = 0;
    for (int i = 0; i < 8; i++) *v |= (uint64_t)p[i] << (8 * i);
    p += 8;
    return true;
}
int put_vec(zk_ctx* ctx, uint8_t*& p, const zk_bases* b, int compressed) {
    const size_t n = b ? b->n : 0;
    put_u64(p, n);
    if (n) ZK_TRY(zk_bases_serialize(ctx, b, 0, n, compressed, p));
    p += n * zk_point_serialized_size(b ? b->group : 1, compressed);
    return ZK_OK;
}
int get_points(zk_ctx* ctx, const uint8_t*& p, const uint8_t* end, int group, size_t n, int compressed, zk_bases** out) {
    const size_t bytes = n * zk_point_serialized_size(group, compressed);
    if ((size_t)(end - p) < bytes) ZK_FAIL(ctx, ZK_ERR_ARG, "deserialize: truncated input");
    ZK_TRY(compressed ? zk_bases_deserialize_compressed(ctx, group, p, n, out) : zk_bases_deserialize_uncompressed(ctx, group, p, n, out));
    p += bytes;
    return ZK_OK;
}
int get_vec(zk_ctx* ctx, const uint8_t*& p, const uint8_t* end, int group, int compressed, zk_bases** out) {
    uint64_t n;
    if (!get_u64(p, end, &n)) ZK_FAIL(ctx, ZK_ERR_ARG, "deserialize: truncated input");
    if (n > ((uint64_t)1 << 32)) ZK_FAIL(ctx, ZK_ERR_ARG, "deserialize: implausible vector length");
    return get_points(ctx, p, end, group, (size_t)n, compressed, out);
}
template <class F>
int points_to_host(zk_ctx* ctx, const zk_bases* b, Affine<F>* out) {
    std::vector<uint32_t> w(b->n * 2 * F::WORDS);
    ZK_HIP(ctx, hipMemcpy(w.data(), b->dev, w.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < b->n; i++) out[i] = aff_load<F>(&w[i * 2 * F::WORDS]);
    return ZK_OK;
}

}  // namespace

extern "C" size_t zk_vk_serialized_size(const zk_pk* pk, int compressed) {
    if (!pk) return 0;
    return zk_point_serialized_size(1, compressed) * (1 + (pk->gamma_abc ? pk->gamma_abc->n : 0)) + zk_point_serialized_size(2, compressed) * 3 + 8;
}
extern "C" size_t zk_pk_serialized_size(const zk_pk* pk, int compressed) {
    if (!pk) return 0;
    const size_t g1 = zk_point_serialized_size(1, compressed), g2 = zk_point_serialized_size(2, compressed);
    return zk_vk_serialized_size(pk, compressed) + 2 * g1 + 5 * 8 + g1 * (pk->a->n + pk->b_g1->n + pk->h->n + pk->l->n) + g2 * pk->b_g2->n;
}
extern "C" int zk_vk_serialize(zk_ctx* ctx, const zk_pk* pk, int compressed, uint8_t* out, size_t cap) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !out) return ZK_ERR_ARG;
    if (cap < zk_vk_serialized_size(pk, compressed)) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_vk_serialize: buffer too small");
    uint8_t* p = out;
    ZK_TRY((ser_small<G1Field, zk_g1_affine>(ctx, &pk->alpha_g1, 1, 1, compressed, p)));
    p += zk_point_serialized_size(1, compressed);
    const Affine<G2Field> g2s[3] = {pk->beta_g2, pk->gamma_g2, pk->delta_g2};
    ZK_TRY((ser_small<G2Field, zk_g2_affine>(ctx, g2s, 3, 2, compressed, p)));
    p += 3 * zk_point_serialized_size(2, compressed);
    return put_vec(ctx, p, pk->gamma_abc, compressed);
    ZK_API_END
}
extern "C" int zk_pk_serialize(zk_ctx* ctx, const zk_pk* pk, int compressed, uint8_t* out, size_t cap) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !out) return ZK_ERR_ARG;
    if (cap < zk_pk_serialized_size(pk, compressed)) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_pk_serialize: buffer too small");
    ZK_TRY(zk_vk_serialize(ctx, pk, compressed, out, cap));
    uint8_t* p = out + zk_vk_serialized_size(pk, compressed);
    const Affine<G1Field> g1s[2] = {pk->beta_g1, pk->delta_g1};
    ZK_TRY((ser_small<G1Field, zk_g1_affine>(ctx, g1s, 2, 1, compressed, p)));
    p += 2 * zk_point_serialized_size(1, compressed);
    for (const zk_bases* q : {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l}) ZK_TRY(put_vec(ctx, p, q, compressed));
    return ZK_OK;
    ZK_API_END
}
// ProvingKey::deserialize / deserialize_uncompressed: the key becomes resident (window multiples, padded l_query) like one
// from zk_pk_upload; gamma_g2 and gamma_abc_g1 are kept for zk_vk_serialize / zk_pk_vk_g2 / zk_pk_download_g1(which = 5).
extern "C" int zk_pk_deserialize(zk_ctx* ctx, const uint8_t* bytes, size_t len, int compressed, zk_pk** out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !bytes || !out) return ZK_ERR_ARG;
    const uint8_t *p = bytes, *end = bytes + len;
    zk_pk* pk = new zk_pk();
    zk_bases *one1 = nullptr, *three2 = nullptr, *two1 = nullptr;
    int rc = get_points(ctx, p, end, 1, 1, compressed, &one1);
    if (rc == ZK_OK) rc = get_points(ctx, p, end, 2, 3, compressed, &three2);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->gamma_abc);
    if (rc == ZK_OK) rc = get_points(ctx, p, end, 1, 2, compressed, &two1);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->a);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->b_g1);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 2, compressed, &pk->b_g2);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->h);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->l);
    if (rc == ZK_OK && p != end) { ctx->last_error = "zk_pk_deserialize: trailing bytes after the proving key"; rc = ZK_ERR_ARG; }
    if (rc == ZK_OK) {
        Affine<G2Field> g2s[3];
        Affine<G1Field> g1s[2];
        rc = points_to_host<G1Field>(ctx, one1, &pk->alpha_g1);
        if (rc == ZK_OK) rc = points_to_host<G2Field>(ctx, three2, g2s);
        if (rc == ZK_OK) rc = points_to_host<G1Field>(ctx, two1, g1s);
        pk->beta_g2 = g2s[0]; pk->gamma_g2 = g2s[1]; pk->delta_g2 = g2s[2];
        pk->beta_g1 = g1s[0]; pk->delta_g1 = g1s[1];
    }
    for (zk_bases* q : {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l})
        if (rc == ZK_OK) rc = zk_bases_precompute_auto(ctx, q);
    if (rc == ZK_OK) rc = pk_make_l_pad(ctx, pk);
    if (rc == ZK_OK) rc = first_point<G1Field>(ctx, pk->a, &pk->a0);
    if (rc == ZK_OK) rc = first_point<G1Field>(ctx, pk->b_g1, &pk->b0_g1);
    if (rc == ZK_OK) rc = first_point<G2Field>(ctx, pk->b_g2, &pk->b0_g2);
    zk_bases_free(ctx, one1); zk_bases_free(ctx, three2); zk_bases_free(ctx, two1);
    if (rc != ZK_OK) { zk_pk_free(ctx, pk); return rc; }
    *out = pk;
    return ZK_OK;
    ZK_API_END
}

extern "C" size_t zk_kzg_srs_serialized_size(size_t n_powers_g, size_t n_powers_gamma_g, int compressed) {
    const size_t g1 = zk_point_serialized_size(1, compressed), g2 = zk_point_serialized_size(2, compressed);
    return 8 + n_powers_g * g1 + 8 + n_powers_gamma_g * (8 + g1) + 2 * g2 + 8;
}
// powers_of_gamma_g holds the keys 0 .. n - 1 (KZG10::setup fills 0 ..= max_degree + 1); neg_powers_of_h is written empty
// (setup(.., produce_g2_powers = false), which is what MarlinKZG10::setup asks for: marlin_pc/mod.rs:77).
extern "C" int zk_kzg_srs_serialize(zk_ctx* ctx, const zk_bases* powers_g, const zk_bases* powers_gamma_g, const zk_g2_affine* h,
                                    const zk_g2_affine* beta_h, int compressed, uint8_t* out, size_t cap) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !powers_g || !powers_gamma_g || !h || !beta_h || !out || powers_g->group != 1 || powers_gamma_g->group != 1) return ZK_ERR_ARG;
    if (cap < zk_kzg_srs_serialized_size(powers_g->n, powers_gamma_g->n, compressed)) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_kzg_srs_serialize: buffer too small");
    uint8_t* p = out;
    ZK_TRY(put_vec(ctx, p, powers_g, compressed));
    const size_t g1 = zk_point_serialized_size(1, compressed), n = powers_gamma_g->n;
    std::vector<uint8_t> tmp(n * g1);
    if (n) ZK_TRY(zk_bases_serialize(ctx, powers_gamma_g, 0, n, compressed, tmp.data()));
    put_u64(p, n);
    for (size_t i = 0; i < n; i++) { put_u64(p, i); memcpy(p, &tmp[i * g1], g1); p += g1; }
    const Affine<G2Field> hs[2] = {host_aff_from_abi<G2Field>((const uint64_t*)h), host_aff_from_abi<G2Field>((const uint64_t*)beta_h)};
    ZK_TRY((ser_small<G2Field, zk_g2_affine>(ctx, hs, 2, 2, compressed, p)));
    p += 2 * zk_point_serialized_size(2, compressed);
    put_u64(p, 0);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_kzg_srs_deserialize(zk_ctx* ctx, const uint8_t* bytes, size_t len, int compressed, zk_bases** powers_g,
                                      zk_bases** powers_gamma_g, zk_g2_affine* h, zk_g2_affine* beta_h) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !bytes || !powers_g || !powers_gamma_g || !h || !beta_h) return ZK_ERR_ARG;
    const uint8_t *p = bytes, *end = bytes + len;
    zk_bases *pg = nullptr, *pgg = nullptr, *hh = nullptr;
    int rc = get_vec(ctx, p, end, 1, compressed, &pg);
    uint64_t n = 0;
    if (rc == ZK_OK && !get_u64(p, end, &n)) { ctx->last_error = "zk_kzg_srs_deserialize: truncated input"; rc = ZK_ERR_ARG; }
    const size_t g1 = zk_point_serialized_size(1, compressed);
    if (rc == ZK_OK && (n > ((uint64_t)1 << 32) || (size_t)(end - p) < n * (8 + g1))) { ctx->last_error = "zk_kzg_srs_deserialize: truncated input"; rc = ZK_ERR_ARG; }
    if (rc == ZK_OK) {
        std::vector<uint8_t> tmp(n * g1);
        for (uint64_t i = 0; i < n && rc == ZK_OK; i++) {
            uint64_t key;
            get_u64(p, end, &key);
            if (key != i) { ctx->last_error = "zk_kzg_srs_deserialize: powers_of_gamma_g keys are not 0 .. n - 1"; rc = ZK_ERR_ARG; }
            memcpy(&tmp[i * g1], p, g1);
            p += g1;
        }
        const uint8_t* q = tmp.data();
        if (rc == ZK_OK) rc = get_points(ctx, q, tmp.data() + tmp.size(), 1, (size_t)n, compressed, &pgg);
    }
    if (rc == ZK_OK) rc = get_points(ctx, p, end, 2, 2, compressed, &hh);
    uint64_t nneg = 0;
    if (rc == ZK_OK && (!get_u64(p, end, &nneg) || nneg != 0 || p != end)) {
        ctx->last_error = "zk_kzg_srs_deserialize: neg_powers_of_h must be empty and nothing may follow";
        rc = ZK_ERR_ARG;
    }
    if (rc == ZK_OK) {
        Affine<G2Field> hs[2];
        rc = points_to_host<G2Field>(ctx, hh, hs);
        host_aff_to_abi<G2Field>((uint64_t*)h, hs[0]);
        host_aff_to_abi<G2Field>((uint64_t*)beta_h, hs[1]);
    }
    zk_bases_free(ctx, hh);
    if (rc != ZK_OK) { zk_bases_free(ctx, pg); zk_bases_free(ctx, pgg); return rc; }
    *powers_g = pg;
    *powers_gamma_g = pgg;
    return ZK_OK;
    ZK_API_END
}
