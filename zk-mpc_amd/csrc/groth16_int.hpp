// groth16_int.hpp -- what the Groth16 translation units share: the device-side constraint system and proving key, the state of a
// proof's front that was enqueued ahead of its proof, and the pipeline's internal entry points.
//   r1cs.hip             constraint system upload, sparse mat-vec, R1CStoQAP::witness_map          (src/groth16.rs:205-306)
//   groth16_key.hip      proving key upload / accessors, generate_parameters with known toxic waste (generator.rs:44-231)
//   key_io.hip           CanonicalSerialize framing of keys and SRS                                (data_structures.rs:133-151)
//   groth16_pipeline.hip the five MSMs of create_proof as a stream pipeline, fronts, presorts       (src/groth16.rs:106-160)
//   msm_batch.hip        several independent MSMs enqueued in one go (Marlin's commitment rounds)
//   groth16_prove.hip    create_proof for a local prover                                           (src/groth16.rs:68-183)
//   groth16_multi.hip    one prover's MSMs spread over several devices
//   groth16_shared.hip   create_proof over additive / SPDZ shares as one call
#pragma once
#include "devutil.cuh"
#include "hostgroup.hpp"
#include "hostfield64.hpp"
#include "internal.hpp"
#include "../../include/zkmpc_hip.h"
#include <chrono>
#include <cstring>
#include <functional>
#include <future>
#include <memory>
#include <vector>

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
        std::vector<zk::Fr> h_coeff;    // internal form (empty when all_one)
    } m[3];
};

struct zk_pk {
    zk_bases *a = nullptr, *b_g1 = nullptr, *b_g2 = nullptr, *h = nullptr, *l = nullptr, *gamma_abc = nullptr;
    // l_query behind as many points at infinity as a_query has entries for the instance (l_pad[ni + j] = l[j]), so that the
    // L job indexes its table by the position in z like A and B do and reuses their sort of z[1..]; the instance part adds
    // infinity, which the complete addition skips.  Only the prover's pipeline reads it.
    zk_bases* l_pad = nullptr;
    // every point is a multiple of the generators (zk_groth16_setup): the proof tail may use the endomorphism (hostfield64.hpp:
    // host64_scalar_mul_glv); a deserialised key is not checked for subgroup membership and keeps the plain scalar multiplication
    bool points_in_subgroup = false;
    zk::Affine<zk::G1Field> alpha_g1, beta_g1, delta_g1, a0, b0_g1;
    zk::Affine<zk::G2Field> beta_g2, delta_g2, gamma_g2, b0_g2;
};

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
    // a SMALL local proof's front carries its whole device chain (groth16_pipeline.hip: "chained"): besides the sorts and the witness
    // map also the accumulate launches and reduce chains of all five jobs are enqueued behind the current proof's -- the device
    // goes from one proof's chain into the next while the host still finishes the first
    bool chained = false;
    ~ZkPresort() { if (wm_done) (void)hipEventDestroy(wm_done); }
};

// the context's helper streams: aux[0..k) and the accumulate stream (created on first use)
int zk_prover_streams(zk_ctx* ctx, size_t k);
// The five MSMs of create_proof as one pipeline (groth16_pipeline.hip).  h_in: the quotient's coefficients when the caller has them
// (the collaborative provers), else NULL and the witness map runs here into h_scratch.  out_g1 = H, L, A, B-in-G1 sums;
// after_abc runs on the calling thread as soon as A, B-in-G1 and B-in-G2 have delivered.
int zk_groth16_run_msms(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z, const void* h_in, void* h_scratch,
                        zk_g1_projective out_g1[4], zk_g2_projective* out_g2, const std::function<void()>& after_abc = nullptr);
int zk_pk_make_l_pad(zk_ctx* ctx, zk_pk* pk);     // groth16_key.hip: see zk_pk::l_pad

// The O(1) host tail of a proof (groth16_prove.hip).  The chains run on the context's helper threads and reference this object:
// declare it after the MSM sums it reads; the destructor joins.
class ZkProofTail {
    using H1 = zk::Fq64Field;
    using H2 = zk::Fq264Field;
    using X1 = zk::XYZZ<H1>;
    using X2 = zk::XYZZ<H2>;
    zk_ctx* ctx;
    bool glv;                                        // the key's points are in the prime-order subgroup: G1 scalar multiplications through the endomorphism
    uint32_t rw[8], sw[8];
    X1 mul1(const X1& p, const uint32_t* k) const { return glv ? zk::host64_scalar_mul_glv(p, k) : zk::host64_scalar_mul<H1>(p, k); }
    X1 delta1;
    X2 delta2;
    zk::Affine<H1> a0, alpha, b0, beta1;
    zk::Affine<H2> b02, beta2;
    X1 g_a, s_g_a, r_s_delta, r_g1_b;
    X1 r_g1, s_g1;                                   // delta r, delta s, delta_2 s: they depend on the key and on r, s only, and start
    X2 s_g2;                                         // with the proof (pre_*), under the device's work, not behind it
    zk::Affine<H2> b_aff;
    ZkTask<void> pre_a, pre_b, pre_2;                // (the chains wait for them; declared first = joined last)
    ZkTask<void> chain_a, chain_b, chain_g2;         // (last members: joined before the fields the chains write go)

   public:
    ZkProofTail(zk_ctx* c, const zk_pk* pk, const zk_fr* r_, const zk_fr* s_);
    void abc_ready(const zk_g1_projective& a_sum, const zk_g1_projective& b1_sum, const zk_g2_projective& b2_sum);
    void join();
    void finish(const zk_g1_projective& h_sum, const zk_g1_projective& l_sum, uint8_t proof[192]);
};

namespace zk {
template <class F>
inline int first_point(zk_ctx* ctx, const zk_bases* b, Affine<F>* out) {
    if (!b || b->n == 0) { *out = aff_inf<F>(); return ZK_OK; }
    uint32_t w[2 * F::WORDS];
    ZK_HIP(ctx, hipMemcpy(w, b->dev, sizeof w, hipMemcpyDeviceToHost));
    *out = aff_load<F>(w);
    return ZK_OK;
}

}  // namespace zk
