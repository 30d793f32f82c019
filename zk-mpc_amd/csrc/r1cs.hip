// r1cs.hip -- the constraint system on the device and the R1CS -> QAP witness map.
//
// Replaces (reference):
//   R1CStoQAP::witness_map + evaluate_constraint      src/groth16.rs:205-306
//                                                     (stock: arkworks/groth16/src/r1cs_to_qap.rs:94-160)
//   ConstraintMatrices                                arkworks/snark/relations/src/r1cs/constraint_system.rs:650-676
//
// Device layout: the three matrices are CSR (row_ptr u32, col u32, coeff = Fr in the device's
// internal Montgomery form so coeff * z needs no conversion); assignment, QAP vectors and h are
// Fr vectors in the reference's form.
#include "../../include/zkmpc_hip.h"
#include "groth16_int.hpp"

using namespace zk;

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

