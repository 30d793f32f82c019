// marlin.hip -- data-parallel pieces of the Marlin AHP prover that are not already an NTT, an MSM or a vector op.
//
// Replaces (reference, relative to /root/reference/arkworks/marlin/src/ahp):
//   prover.rs:258-278        inner_prod_fn (z_A = A z, z_B = B z)            -> zk_r1cs_matvec_dev (r1cs.hip's SpMV)
//   prover.rs:406-423        calculate_t                                      -> the same SpMV on the transposed matrices
//   prover.rs:335-353        w_poly_evals (index re-mapping of the witness)   -> zk_fr_gather_dev
//   constraint_systems.rs:183-216  row / col / val vectors of M^*             -> zk_fr_gather_dev
//   prover.rs:620-641        f evaluations on K (with ark_ff::batch_inversion)-> zk_marlin_round3_f_evals_dev
//   prover.rs:653-698        a and b evaluations on the domain B              -> zk_marlin_round3_ab_evals_dev
// All element-wise and HBM-bound: the ab kernel reads 12 vectors and writes 2 (448 B per element of B).
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "internal.hpp"

using namespace zk;

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
Fr host_int(const zk_fr* a) { return fp_ext_to_int<FrParams>(host_load_ext<FrParams>(a->l)); }

// out[i] = idx[i] == 0xFFFFFFFF ? 0 : src[idx[i]]
__global__ void __launch_bounds__(256) k_gather(const void* src, const uint32_t* idx, size_t n, void* out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t j = idx[i];
        fr_store(out, i, j == 0xFFFFFFFFu ? fp_zero<FrParams>() : fr_load(src, j));
    }
}

struct Mats { const void* row[3]; const void* col[3]; const void* val[3]; const void* row_col[3]; };

// den[m][i] = (beta - row_m[i]) * (alpha - col_m[i])
__global__ void __launch_bounds__(256) k_denoms_k(Mats M, FrK alpha_ext, FrK beta_ext, size_t n, void* den) {
    const Fr fix = fp_const<FrParams>(FrParams::EXT_TO_INT), a = frk(alpha_ext), b = frk(beta_ext);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int m = 0; m < 3; m++) {
            Fr x = fr_sub(b, fr_load(M.row[m], i)), y = fr_sub(a, fr_load(M.col[m], i));
            fr_store(den, (size_t)m * n + i, fr_mul(fr_mul(x, y), fix));
        }
    }
}

// f[i] = sum_m k_m * val_m[i] * inv[m][i]      (k_m = eta_m * v_H(alpha) * v_H(beta), passed as k_m * RI^2 / RE)
__global__ void __launch_bounds__(256) k_f_vals(Mats M, const void* inv, FrK k0, FrK k1, FrK k2, size_t n, void* f) {
    const Fr k[3] = {frk(k0), frk(k1), frk(k2)};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr acc = fp_zero<FrParams>();
#pragma unroll
        for (int m = 0; m < 3; m++)
            acc = fr_add(acc, fr_mul(fr_mul(fr_load(M.val[m], i), fr_load(inv, (size_t)m * n + i)), k[m]));
        fr_store(f, i, acc);
    }
}

// den_m = beta alpha - alpha row_m - beta col_m + row_col_m ;  b = den_a den_b den_c ;
// a = sum_m k_m val_m prod_{m' != m} den_m'
__global__ void __launch_bounds__(256) k_ab_on_b(Mats M, FrK alpha_int, FrK beta_int, FrK ba_ext, FrK k0, FrK k1, FrK k2, FrK fix2,
                                                 size_t n, void* a_out, void* b_out) {
    const Fr al = frk(alpha_int), be = frk(beta_int), ba = frk(ba_ext), f2 = frk(fix2);
    const Fr k[3] = {frk(k0), frk(k1), frk(k2)};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr d[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            Fr t = fr_sub(ba, fr_mul(fr_load(M.row[m], i), al));
            t = fr_sub(t, fr_mul(fr_load(M.col[m], i), be));
            d[m] = fr_add(t, fr_load(M.row_col[m], i));
        }
        Fr bc = fr_mul(d[1], d[2]), ac = fr_mul(d[0], d[2]), ab = fr_mul(d[0], d[1]);   // ext * RE / RI
        Fr a = fr_mul(fr_mul(fr_load(M.val[0], i), bc), k[0]);
        a = fr_add(a, fr_mul(fr_mul(fr_load(M.val[1], i), ac), k[1]));
        a = fr_add(a, fr_mul(fr_mul(fr_load(M.val[2], i), ab), k[2]));
        fr_store(a_out, i, a);
        fr_store(b_out, i, fr_mul(fr_mul(ab, d[2]), f2));
    }
}

int load_mats(zk_ctx* ctx, const zk_marlin_matrix_evals* e, bool need_row_col, Mats* M) {
    for (int m = 0; m < 3; m++) {
        if (!e[m].row || !e[m].col || !e[m].val || (need_row_col && !e[m].row_col)) ZK_FAIL(ctx, ZK_ERR_ARG, "marlin: missing matrix evaluation vector");
        M->row[m] = e[m].row; M->col[m] = e[m].col; M->val[m] = e[m].val; M->row_col[m] = e[m].row_col;
    }
    return ZK_OK;
}

}  // namespace

extern "C" int zk_fr_gather_dev(zk_ctx* ctx, const void* src, const uint32_t* idx_dev, size_t n, void* out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (n && (!src || !idx_dev || !out))) return ZK_ERR_ARG;
    if (n == 0) return ZK_OK;
    hipLaunchKernelGGL(k_gather, zk_grid(n, 256), 256, 0, ctx->stream, src, idx_dev, n, out);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_marlin_round3_f_evals_dev(zk_ctx* ctx, const zk_marlin_matrix_evals on_k[3], size_t k_size, const zk_fr* alpha,
                                            const zk_fr* beta, const zk_fr eta[3], const zk_fr* vh_alpha_vh_beta, void* f_out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !on_k || !alpha || !beta || !eta || !vh_alpha_vh_beta || !f_out || k_size == 0) return ZK_ERR_ARG;
    Mats M;
    ZK_TRY(load_mats(ctx, on_k, false, &M));
    void* den;
    ZK_TRY(zk_scratch(ctx, "marlin_den", 3 * k_size * 32, &den));
    hipLaunchKernelGGL(k_denoms_k, zk_grid(k_size, 256), 256, 0, ctx->stream, M, to_frk(host_load_ext<FrParams>(alpha->l)),
                       to_frk(host_load_ext<FrParams>(beta->l)), k_size, den);
    ZK_HIP(ctx, hipGetLastError());
    ZK_TRY(zk_fr_batch_inverse_dev(ctx, den, 3 * k_size));      // zero denominators stay zero, as in ark_ff::batch_inversion
    const Fr vv = host_int(vh_alpha_vh_beta), e2i = fp_const<FrParams>(FrParams::EXT_TO_INT);
    FrK k[3];
    for (int m = 0; m < 3; m++) k[m] = to_frk(fp_mul<FrParams>(fp_mul<FrParams>(host_int(&eta[m]), vv), e2i));
    hipLaunchKernelGGL(k_f_vals, zk_grid(k_size, 256), 256, 0, ctx->stream, M, (const void*)den, k[0], k[1], k[2], k_size, f_out);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_marlin_round3_ab_evals_dev(zk_ctx* ctx, const zk_marlin_matrix_evals on_b[3], size_t b_size, const zk_fr* alpha,
                                             const zk_fr* beta, const zk_fr eta[3], const zk_fr* vh_alpha_vh_beta, void* a_out,
                                             void* b_out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !on_b || !alpha || !beta || !eta || !vh_alpha_vh_beta || !a_out || !b_out || b_size == 0) return ZK_ERR_ARG;
    Mats M;
    ZK_TRY(load_mats(ctx, on_b, true, &M));
    const Fr al = host_int(alpha), be = host_int(beta), vv = host_int(vh_alpha_vh_beta);
    const Fr e2i = fp_const<FrParams>(FrParams::EXT_TO_INT), i2e = fp_const<FrParams>(FrParams::INT_TO_EXT);
    const Fr fix2 = fp_mul<FrParams>(e2i, e2i);                        // RI^3 / RE^2: repairs a triple product of ext values
    const Fr ba_ext = fp_mul<FrParams>(fp_mul<FrParams>(al, be), i2e);
    FrK k[3];
    for (int m = 0; m < 3; m++) k[m] = to_frk(fp_mul<FrParams>(fp_mul<FrParams>(host_int(&eta[m]), vv), fix2));
    hipLaunchKernelGGL(k_ab_on_b, zk_grid(b_size, 256), 256, 0, ctx->stream, M, to_frk(al), to_frk(be), to_frk(ba_ext), k[0], k[1], k[2],
                       to_frk(fix2), b_size, a_out, b_out);
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}
