// groth16_pipeline.hip -- the five MSMs of create_proof (src/groth16.rs:106,110,137,148,160) as a pipeline over three streams, with
// the next proof's front and the collaborative prover's presorts enqueued ahead of their proofs.
#include "../../include/zkmpc_hip.h"
#include "groth16_int.hpp"
#include <chrono>

using namespace zk;

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

// ONE side stream per context for everything that runs beside the caller's own calls on the trait-shaped path -- the slices of the
// table cache's builder (bases_cache.hip) and the MSMs started ahead (msm.hip) -- and no accumulate stream with it: with the default
// four hardware queues every stream beyond {null, context, transfer ring, this one} shares a queue with one of them, and the
// host-slice transforms and products then run 8 - 30 % longer even while the extra streams sit idle (measured, round 6: seven
// transforms at 2^20 12.9-13.4 -> 13.8-15.5 ms, the batch product 2.4 -> 3.1-3.3 ms with one to four idle streams more;
// GPU_MAX_HW_QUEUES=8 takes the effect away, but that is the host application's setting, and the one-call provers lose with it).
int zk_side_stream(zk_ctx* ctx, hipStream_t* out) {
    if (ctx->aux.empty()) {
        hipStream_t st;
        ZK_HIP(ctx, zk_stream_create(&st, true));
        ctx->aux.push_back(st);
    }
    *out = ctx->aux[0];
    return ZK_OK;
}

int zk_prover_streams(zk_ctx* ctx, size_t k) {
    while (ctx->aux.size() < k) {
        hipStream_t st;
        ZK_HIP(ctx, zk_stream_create(&st, true));
        ctx->aux.push_back(st);
    }
    // (the accumulate stream is an ordinary stream: a CU mask that withholds compute units for the other streams costs what it
    // frees -- round 3, ZK_ACC_CU_RESERVE)
    if (!ctx->acc_stream) ZK_HIP(ctx, zk_stream_create(&ctx->acc_stream, false));
    return ZK_OK;
}

// The five MSMs of create_proof as a pipeline over THREE streams (the runtime maps streams onto ~3
// usable hardware queues; a fourth stream lands on an occupied queue and serialises behind it):
//   main       : witness map, then the sort of the H job (its scalars come out of the witness map)
//   sort/reduce: sort of the z-dependent jobs first (B-in-G2 / A / B-in-G1 share one sort: same scalars
//                z[1..]), then, as the accumulate kernels complete, each job's reduce phase in job order
//   accum      : the five accumulate kernels back to back, B-in-G2 first (longest reduce), H last;
//                the first one is gated on the witness map, which would otherwise be starved 15x beside it
int zk_groth16_run_msms(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z, const void* h_in, void* h_scratch,
                        zk_g1_projective out_g1[4], zk_g2_projective* out_g2, const std::function<void()>& after_abc) {
    const auto t_enter = std::chrono::steady_clock::now();
    const size_t D = (size_t)1 << r->log_d;
    const size_t nvars = (r->ni - 1) + r->nw;
    const char* zb = (const char*)z;
    if (pk->a->n != nvars + 1 || pk->b_g1->n != nvars + 1 || pk->b_g2->n != nvars + 1 || pk->l->n != r->nw)
        ZK_FAIL(ctx, ZK_ERR_ARG, "groth16: proving key does not match the constraint system");
    ZK_TRY(zk_prover_streams(ctx, 1));
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
    const bool chained = fronted && pre->chained;                          // ... all five are (a small proof's front: see below)
    if (presorted) J[0] = &pre->job;
    if (fronted) J[4] = &pre->jobh;
    if (begun || chained) { J[1] = &pre->j1; J[2] = &pre->j2; J[3] = &pre->j3; }
    // z was produced on the context stream.  (With a front that stream already carries this proof's witness map and H-sort:
    // waiting for it here would hold the sort stream -- and the G2 reduce chain on it -- until the H-sort is through.)
    if (!fronted) ZK_HIP(ctx, hipStreamWaitEvent(s_sort, e0, 0));
    int rc = presorted ? ZK_OK : zk_msm_prepare(ctx, J[0], pk->b_g2, 1, zb + 32, nvars, 1);                 // src/groth16.rs:160 (query[1..])
    const bool have_z_jobs = begun || chained;
    if (rc == ZK_OK && !have_z_jobs) rc = zk_msm_prepare(ctx, J[1], pk->a, 1, zb + 32, nvars, 2);              // :137
    if (rc == ZK_OK && !have_z_jobs) rc = zk_msm_prepare(ctx, J[2], pk->b_g1, 1, zb + 32, nvars, 3);           // :148
    // :110: aux_assignment against l_query; over the padded table the same sum reads z[1..] (the instance meets infinity)
    const bool l_shared = pk->l_pad && pk->l_pad->n == nvars + 1 && (pk->l_pad->pre != nullptr) == (pk->a->pre != nullptr) &&
                          pk->l_pad->c_pre == pk->a->c_pre;
    if (rc == ZK_OK && !have_z_jobs) rc = l_shared ? zk_msm_prepare(ctx, J[3], pk->l_pad, 1, zb + 32, nvars, 4)
                                                   : zk_msm_prepare(ctx, J[3], pk->l, 0, zb + r->ni * 32, r->nw, 4);
    if (rc == ZK_OK && !presorted) rc = zk_msm_enqueue_sort(ctx, J[0], s_sort, nullptr);
    if (rc == ZK_OK && !have_z_jobs) rc = zk_msm_enqueue_sort(ctx, J[1], s_sort, J[0]);
    // (the G2 table may carry windows of another width than the G1 tables: then A sorts for itself and the other G1 jobs borrow A's)
    const ZkMsmJob* lender = J[0]->c == J[1]->c ? J[0] : J[1];
    if (rc == ZK_OK && !have_z_jobs) rc = zk_msm_enqueue_sort(ctx, J[2], s_sort, lender);
    if (rc == ZK_OK && !have_z_jobs && l_shared) rc = zk_msm_enqueue_sort(ctx, J[3], s_sort, lender);
    const void* h = h_in;
    ZkPhaseTimer tm(ctx);
    // the first accumulate kernel is gated on the witness map, which would otherwise be starved beside it (un-gating it: within
    // the noise of consecutive runs, round 3)
    if (fronted) {
        // witness map and H's sort were enqueued with the previous proof (enqueue_front below)
        h = h_scratch;
        ZK_HIP(ctx, hipStreamWaitEvent(s_acc, pre->wm_done, 0));
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
            ZK_HIP(ctx, hipStreamWaitEvent(s_acc, e1, 0));
            rc = zk_msm_enqueue_sort(ctx, J[4], ctx->stream, nullptr);
        }
    }
    // L's sort after the witness map (it is not needed before the fourth accumulate kernel)
    if (rc == ZK_OK && !l_shared && !have_z_jobs) {
        ZK_HIP(ctx, hipStreamWaitEvent(s_sort, e1, 0));
        rc = zk_msm_enqueue_sort(ctx, J[3], s_sort, nullptr);
    }
    // accumulate order (job numbers: 0 = B in G2, 1 = A, 2 = B in G1, 3 = L, 4 = H; H's scalars arrive last)
    const int ord[5] = {0, 1, 2, 3, 4};
    // A job of up to 2^16 terms occupies a tenth of the chip for the length of its longest bucket (~0.3 ms): five of them one
    // behind the other on the accumulate stream are most of a small proof.  There every job's accumulate kernel goes on the stream
    // of its own reduce chain instead (G2 alone on the accumulate stream, the G1 jobs alternating between the sort stream and the
    // context stream): the kernels overlap, the chain follows its kernel without an event.  (Ordering against the next proof's
    // front does not lean on the accumulate stream: that front waits for every job's reduce_done / accum_done event.)
    const bool small_jobs = !begun && D <= ((size_t)1 << 16);
    auto job_stream = [&](int k) -> hipStream_t { return k == 0 ? s_acc : ((k & 1) ? s_sort : ctx->stream); };
    // ... and where the four G1 jobs qualify (tables of window multiples with the same bucket count: every key of >= 256 points)
    // they are ONE accumulate launch and one launch per level of the reduce chain, on the sort stream (round 5: two jobs per stream,
    // one behind the other, were 2 x (0.26 + 0.2) ms of a 1.2 ms proof at 2^10)
    ZkMsmJob* g1jobs[4] = {J[1], J[2], J[3], J[4]};
    const bool grouped = chained || (small_jobs && rc == ZK_OK && zk_msm_group_ok(g1jobs, 4));
    for (int k = 0; k < 5 && rc == ZK_OK && !chained; k++) {
        if (grouped && ord[k] != 0) continue;
        if (!begun || ord[k] == 4) rc = zk_msm_enqueue_accum(ctx, J[ord[k]], small_jobs ? job_stream(ord[k]) : s_acc);
    }
    if (grouped && !chained && rc == ZK_OK) rc = zk_msm_enqueue_accum_group(ctx, g1jobs, 4, s_sort);
    // B-in-G2's reduce chain (the long one) stays on the sort stream; the four G1 reduces go to the main stream, idle by
    // then, so that each runs right behind its own accumulate kernel instead of queueing behind the G2 chain (that
    // queueing left 4 x 0.7 ms of reduces after the last accumulate).
    // The last two reduce chains alternate between the two streams (the sort stream is idle again once the G2 chain is
    // through): on one stream the last job's chain queued behind its predecessor's, which was still waiting for slots
    // beside the last accumulate kernel, and ~0.6 ms of it ran after the GPU had otherwise gone idle.
    for (int k = 0; k < 5 && rc == ZK_OK && !chained; k++) {
        hipStream_t rs = small_jobs ? job_stream(ord[k]) : ((ord[k] == 0 || (k & 1) == 0) ? s_red : ctx->stream);
        if (begun && ord[k] == 0) continue;                 // B in G2's chain went out with zk_groth16_msms_begin_dev
        if (grouped && ord[k] != 0) continue;
        rc = zk_msm_enqueue_reduce(ctx, J[ord[k]], rs);
    }
    if (grouped && !chained && rc == ZK_OK) rc = zk_msm_enqueue_reduce_group(ctx, g1jobs, 4, s_sort);
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
        // A SMALL proof (its four G1 jobs one group): the next proof's whole device chain goes out as well -- accumulate launches and
        // reduce chains of all five jobs, on the streams this proof's own chains are on, so stream order keeps every reader of a
        // scratch buffer in front of its next writer.  The device then runs from one proof's chain into the next while the host
        // finishes the first (Horner chains, the tail's scalar multiplications, the caller's next call: ~0.2 ms during which the
        // chip used to wait).  The results land in the other pair of pinned buffers: this proof's are read below.
        if (rc == ZK_OK && grouped && ctx->chain_fronts) {
            const int par = (ctx->front_parity ^= 1);
            nf->job.pin_key = 48 + par;
            rc = zk_msm_prepare(ctx, &nf->j1, pk->a, 1, (const char*)zn + 32, nvars, 2);
            if (rc == ZK_OK) rc = zk_msm_prepare(ctx, &nf->j2, pk->b_g1, 1, (const char*)zn + 32, nvars, 3);
            if (rc == ZK_OK) rc = zk_msm_prepare(ctx, &nf->j3, pk->l_pad, 1, (const char*)zn + 32, nvars, 4);
            nf->j1.pin_key = 50 + par;                               // (the group's results travel in its first job's buffer)
            if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, &nf->j1, s_acc, &nf->job);
            const ZkMsmJob* nl = nf->job.c == nf->j1.c ? &nf->job : &nf->j1;
            if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, &nf->j2, s_acc, nl);
            if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, &nf->j3, s_acc, nl);
            ZkMsmJob* ng[4] = {&nf->j1, &nf->j2, &nf->j3, &nf->jobh};
            if (rc == ZK_OK && zk_msm_group_ok(ng, 4)) {
                rc = zk_msm_enqueue_accum(ctx, &nf->job, s_acc);
                if (rc == ZK_OK) rc = zk_msm_enqueue_accum_group(ctx, ng, 4, s_sort);
                if (rc == ZK_OK) rc = zk_msm_enqueue_reduce(ctx, &nf->job, s_acc);
                if (rc == ZK_OK) rc = zk_msm_enqueue_reduce_group(ctx, ng, 4, s_sort);
                nf->chained = rc == ZK_OK;
            }
            if (rc != ZK_OK) {                                       // part of a chain is in flight: let it drain before its jobs go
                ctx->presort = nf.release();
                zk_presort_free(ctx);
            }
        }
        if (rc == ZK_OK) { ctx->presort = nf.release(); front_enqueued = true; }
    }
    ctx->next_z = nullptr;
    if (ctx->profiling) {                          // host time from the call to the last enqueue (a small proof's device chain starts late by it)
        auto& t = ctx->timers["host.enqueue"];
        t.ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
        t.count += 1;
    }
    // finish in completion order: the host-side Horner of an early job overlaps the GPU work of the later ones
    void* outs[5] = {out_g2, &out_g1[2], &out_g1[3], &out_g1[1], &out_g1[0]};
    int abc_left = 3;
    if (grouped && rc == ZK_OK) {
        // the four G1 jobs of a group become ready together: their host halves side by side, B in G2's on this thread
        void* gouts[4] = {outs[1], outs[2], outs[3], outs[4]};
        ZkTask<int> g1fin = zk_async(ctx, [&] { (void)hipSetDevice(ctx->device); return zk_msm_finish_many(ctx, g1jobs, gouts, 4); });
        rc = zk_msm_finish(ctx, J[0], outs[0]);
        const int rc1 = g1fin.get();
        if (rc == ZK_OK) rc = rc1;
        if (rc == ZK_OK && after_abc) after_abc();
    } else {
        for (int k = 0; k < 5 && rc == ZK_OK; k++) {
            rc = zk_msm_finish(ctx, J[ord[k]], outs[ord[k]]);
            if (ord[k] <= 2 && --abc_left == 0 && rc == ZK_OK && after_abc) after_abc();   // A, B1, B2 are in: the caller's host work overlaps the rest
        }
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


// The next zk_groth16_prove_dev on this context will be for `z_next_dev` (same key, same constraint system): the proof
// in between enqueues that proof's front (z-sort, witness map, H-sort) behind its own kernels.  Pass NULL to withdraw.
extern "C" int zk_groth16_hint_next_dev(zk_ctx* ctx, const void* z_next_dev) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ctx->next_z = z_next_dev;
    return ZK_OK;
    ZK_API_END
}

// Whether the front of an announced SMALL proof also carries that proof's accumulate launches and reduce chains (default: yes --
// one context proving a queue goes from 0.54 to 0.43 ms per proof at 2^10).  Several contexts sharing one GPU do better without:
// every context then keeps the hardware queues filled with its own next chain and they serialise (four contexts, eight queues:
// 0.27 ms per proof without, 0.47 with; profiles/r5_small_throughput.jsonl).
extern "C" int zk_groth16_chain_fronts(zk_ctx* ctx, int on) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ctx->chain_fronts = on != 0;
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_groth16_msms_presort_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z) return ZK_ERR_ARG;
    zk_presort_free(ctx);
    const size_t nvars = (r->ni - 1) + r->nw;
    if (pk->b_g2->n != nvars + 1) ZK_FAIL(ctx, ZK_ERR_ARG, "groth16: proving key does not match the constraint system");
    ZK_TRY(zk_prover_streams(ctx, 1));
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
    ZK_TRY(zk_prover_streams(ctx, 1));
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
    return zk_groth16_run_msms(ctx, pk, r, z, h, nullptr, out_g1, out_g2);
    ZK_API_END
}

