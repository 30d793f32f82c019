// groth16_prove.hip -- create_proof for a local prover (src/groth16.rs:68-183; stock arkworks/groth16/src/prover.rs:44-153):
// witness map, the five MSMs (groth16_pipeline.hip), calculate_coeff and the r / s terms on the host, 192 bytes out.
#include "../../include/zkmpc_hip.h"
#include <string.h>
#include "groth16_int.hpp"

using namespace zk;

// The O(1) tail of create_proof on the host, 64-bit limbs (hostfield64.hpp): calculate_coeff (src/groth16.rs:185-201: initial +
// query[0] + acc + vk_param) for A, B-in-G1 and B-in-G2, the r / s terms (:115, :140, :161), C (:169-174) and the 192 bytes.
// Everything that depends only on the A, B-in-G1 and B-in-G2 sums (and on r, s, the key) starts -- on the context's helper
// threads -- as soon as those three MSMs have delivered (abc_ready), while the devices still work on L and H.
ZkProofTail::ZkProofTail(zk_ctx* c, const zk_pk* pk, const zk_fr* r_, const zk_fr* s_) : ctx(c), glv(pk->points_in_subgroup) {
    fr_abi_to_canon_words(r_->l, rw);
    fr_abi_to_canon_words(s_->l, sw);
    delta1 = xyzz_from_affine<H1>(aff_to_host64<G1Field>(pk->delta_g1));
    delta2 = xyzz_from_affine<H2>(aff_to_host64<G2Field>(pk->delta_g2));
    a0 = aff_to_host64<G1Field>(pk->a0); alpha = aff_to_host64<G1Field>(pk->alpha_g1);
    b0 = aff_to_host64<G1Field>(pk->b0_g1); beta1 = aff_to_host64<G1Field>(pk->beta_g1);
    b02 = aff_to_host64<G2Field>(pk->b0_g2); beta2 = aff_to_host64<G2Field>(pk->beta_g2);
    // the scalar multiplications that need no MSM result: four of the seven of a proof (0.2 - 0.7 ms each on one host thread)
    pre_a = zk_async(ctx, [this] {
        r_g1 = mul1(delta1, rw);
        r_s_delta = mul1(r_g1, sw);                                                     // :115
    });
    pre_b = zk_async(ctx, [this] { s_g1 = mul1(delta1, sw); });
    pre_2 = zk_async(ctx, [this] { s_g2 = host64_scalar_mul<H2>(delta2, sw); });
}

void ZkProofTail::abc_ready(const zk_g1_projective& a_sum, const zk_g1_projective& b1_sum, const zk_g2_projective& b2_sum) {
    const X1 a_acc = host64_proj_from_abi<H1>((const uint64_t*)&a_sum);
    const X1 b1_acc = host64_proj_from_abi<H1>((const uint64_t*)&b1_sum);
    const X2 b2_acc = host64_proj_from_abi<H2>((const uint64_t*)&b2_sum);
    chain_a = zk_async(ctx, [this, a_acc] {
        pre_a.wait();
        g_a = xyzz_madd<H1>(xyzz_add<H1>(xyzz_madd<H1>(r_g1, a0), a_acc), alpha);
        s_g_a = mul1(g_a, sw);                                                           // :140
    });
    chain_b = zk_async(ctx, [this, b1_acc] {
        pre_b.wait();
        const X1 g1_b = xyzz_madd<H1>(xyzz_add<H1>(xyzz_madd<H1>(s_g1, b0), b1_acc), beta1);
        r_g1_b = mul1(g1_b, rw);                                                         // :161
    });
    chain_g2 = zk_async(ctx, [this, b2_acc] {
        pre_2.wait();
        const X2 g2_b = xyzz_madd<H2>(xyzz_add<H2>(xyzz_madd<H2>(s_g2, b02), b2_acc), beta2);
        b_aff = xyzz_to_affine<H2>(g2_b);
    });
}

void ZkProofTail::join() {
    // the chains first: each waits for its own pre_* task, and a future is waited for by one thread at a time
    if (chain_a.valid()) chain_a.get();
    if (chain_b.valid()) chain_b.get();
    if (chain_g2.valid()) chain_g2.get();
    if (pre_a.valid()) pre_a.wait();
    if (pre_b.valid()) pre_b.wait();
    if (pre_2.valid()) pre_2.wait();
}

void ZkProofTail::finish(const zk_g1_projective& h_sum, const zk_g1_projective& l_sum, uint8_t proof[192]) {
    join();
    const X1 h_acc = host64_proj_from_abi<H1>((const uint64_t*)&h_sum);
    const X1 l_acc = host64_proj_from_abi<H1>((const uint64_t*)&l_sum);
    X1 g_c = xyzz_add<H1>(s_g_a, r_g1_b);                                                                 // :169-174
    g_c = xyzz_add<H1>(g_c, xyzz_neg<H1>(r_s_delta));
    g_c = xyzz_add<H1>(g_c, l_acc);
    g_c = xyzz_add<H1>(g_c, h_acc);
    g1_serialize(aff_from_host64<G1Field>(xyzz_to_affine<H1>(g_a)), proof);
    g2_serialize(aff_from_host64<G2Field>(b_aff), proof + 48);
    g1_serialize(aff_from_host64<G1Field>(xyzz_to_affine<H1>(g_c)), proof + 144);
}

extern "C" int zk_groth16_prove_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z, const zk_fr* r_, const zk_fr* s_,
                                    uint8_t proof[192]) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z || !r_ || !s_ || !proof) return ZK_ERR_ARG;
    const size_t D = (size_t)1 << r->log_d;
    void* h;
    ZK_TRY(zk_scratch(ctx, "prove_h", D * 32, &h));
    zk_g1_projective m1[4];      // H, L, A, B-in-G1 (declared before the tail: its tasks are joined before these go)
    zk_g2_projective m2;
    ZkProofTail tail(ctx, pk, r_, s_);
    const int rc_msm = zk_groth16_run_msms(ctx, pk, r, z, nullptr, h, m1, &m2, [&] { tail.abc_ready(m1[2], m1[3], m2); });
    const auto t_tail = std::chrono::steady_clock::now();      // what is left of the host work once the GPU is done
    tail.join();
    ZK_TRY(rc_msm);
    tail.finish(m1[0], m1[1], proof);
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
        // page-locked memory (zk_host_alloc) goes to the DMA engine as it is; anything else through the context's ring, so that
        // the runtime never pins memory the caller is about to free (hostxfer.hip)
        ZK_TRY(zk_xfer_h2d(ctx, z, z_host, m * 32, zk_host_is_pinned(z_host)));
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
        if (zk_host_is_pinned(z_next_host)) {
            ZK_HIP(ctx, hipMemcpyAsync(zn, z_next_host, m * 32, hipMemcpyHostToDevice, ctx->copy_stream));
            ZK_HIP(ctx, hipEventRecord(ctx->next_z_ready, ctx->copy_stream));
        } else {
            // pageable: staged through the ring now (the call returns behind the host copy) on the ring's DMA stream ALONE -- no
            // fence with the context stream in either direction (the slot's last reader has delivered its bytes; the consumer waits
            // for next_z_ready), so the DMA runs beside this proof's witness map and sorts instead of inside their stream
            const char* src = (const char*)z_next_host;
            const ZkXferFill fill = [src](char* dst, size_t off, size_t len) { memcpy(dst, src + off, len); };
            ZK_TRY(zk_xfer_h2d_fn(ctx, zn, m * 32, fill, false, nullptr));
            hipStream_t xs;
            ZK_TRY(zk_xfer_stream(ctx, &xs));
            ZK_HIP(ctx, hipEventRecord(ctx->next_z_ready, xs));
        }
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

