// msm_batch.hip -- several independent MSMs enqueued in one go.
#include "../../include/zkmpc_hip.h"
#include "groth16_int.hpp"
#include <algorithm>

using namespace zk;

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
    ZK_TRY(zk_prover_streams(ctx, 2));
    constexpr size_t SLOTS = 3;
    hipStream_t s_acc = ctx->acc_stream, s_sort = ctx->aux[0], s_sort2 = ctx->aux[1];
    hipEvent_t e0;
    ZK_HIP(ctx, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    ZK_HIP(ctx, hipEventRecord(e0, ctx->stream));        // the scalars were produced on the context stream
    ZK_HIP(ctx, hipStreamWaitEvent(s_sort, e0, 0));
    ZK_HIP(ctx, hipStreamWaitEvent(s_acc, e0, 0));
    ZK_HIP(ctx, hipStreamWaitEvent(s_sort2, e0, 0));
    std::vector<size_t> perm(n_jobs);
    for (size_t k = 0; k < n_jobs; k++) perm[k] = k;
    std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return lens[a] > lens[b]; });
    std::vector<ZkMsmJob> jobs(n_jobs);
    // G1 jobs of up to 2^23 digits over tables of window multiples (the commitments of a Marlin round up to |H| = 2^18: 4 - 7 jobs) go
    // in GROUPS of up to four: sorts one behind the other, then one accumulate launch and one launch per level
    // of the reduce chain for the group (msm.hip: zk_msm_enqueue_*_group).  Group position p uses scratch slot 1 + p; a slot's
    // next user waits for its previous user's chain.
    {
        bool small = n_jobs >= 2;
        for (size_t k = 0; k < n_jobs && small; k++) {
            const zk_bases* b = bases[k];
            small = b->group == 1 && b->pre && lens[k] > 0 && (lens[k] >= 4096 || lens[k] * 8 >= b->n) && b->c_pre == bases[0]->c_pre &&
                    (b->pre_stride == 64) == (bases[0]->pre_stride == 64) && lens[k] * ((255 + b->c_pre - 1) / b->c_pre) <= ((size_t)1 << 23);
        }
        if (small) {
            int rc = ZK_OK;
            constexpr size_t GM = 4;
            for (size_t g0 = 0; g0 < n_jobs && rc == ZK_OK; g0 += GM) {
                const size_t cnt = std::min(GM, n_jobs - g0);
                ZkMsmJob* grp[GM];
                for (size_t p = 0; p < cnt && rc == ZK_OK; p++) {
                    const size_t k = g0 + p, j = perm[k];
                    grp[p] = &jobs[k];
                    jobs[k].pin_key = 16 + (int)k;
                    rc = zk_msm_prepare(ctx, &jobs[k], bases[j], base_offsets ? base_offsets[j] : 0, scalars_dev[j], lens[j], 1 + (int)p);
                }
                if (rc != ZK_OK) break;
                {   // a group that leaves most of the chip idle (<= 2^21 digits in all) lasts as long as its longest segment: the one job
                    // of more than 2^19 digits in it (a 3|H|-coefficient polynomial of a 2^14 Marlin proof: 32-term segments, 0.29 ms of
                    // one lane's chain beside 8-term segments of the others) takes 16-term segments (Marlin 2^14: 6.1-6.2 -> 5.7-5.8 ms)
                    size_t digits = 0;
                    for (size_t p = 0; p < cnt; p++) digits += grp[p]->n * grp[p]->W;
                    if (digits <= ((size_t)1 << 21))
                        for (size_t p = 0; p < cnt; p++) if (grp[p]->seg == 32) grp[p]->seg = 16;
                }
                {
                    // the jobs that take the one-block sort: one launch, a block per job (sort stream); the others (a job of more than
                    // 2^16 digits: ~10 short launches each) side by side on the other streams -- one each (two behind each other on two
                    // streams: every job of a round at |H| = 2^14 is past the one-block sort; Marlin 2^13 .. 2^16, same box, two runs each:
                    // 5.40 / 5.43 -> 5.27 / 5.14, 6.25 / 6.26 -> 6.13 / 5.99, 7.58 / 7.58 -> 7.50 / 7.24, 9.70 / 9.70 -> 9.59 / 9.37 ms)
                    hipStream_t others[4] = {ctx->stream, s_acc, s_sort2, s_sort};
                    ZkMsmJob* fit[GM]; size_t nfit = 0, nother = 0;
                    ZkMsmJob* one[1];
                    for (size_t p = 0; p < cnt; p++) { one[0] = grp[p]; if (zk_msm_sort_group_ok(one, 1)) fit[nfit++] = grp[p]; }
                    if (nfit < 2) nfit = 0;
                    for (size_t p = 0; p < cnt && rc == ZK_OK; p++) {
                        const size_t k = g0 + p;
                        bool in_fit = false;
                        for (size_t f = 0; f < nfit; f++) in_fit = in_fit || fit[f] == grp[p];
                        hipStream_t ss = in_fit ? s_sort : others[nother++ % (nfit ? 3 : 4)];
                        if (g0 && jobs[k - GM].reduce_done) ZK_HIP(ctx, hipStreamWaitEvent(ss, jobs[k - GM].reduce_done, 0));
                        if (!in_fit) rc = zk_msm_enqueue_sort(ctx, &jobs[k], ss, nullptr);
                    }
                    if (rc == ZK_OK && nfit) rc = zk_msm_enqueue_sort_group(ctx, fit, (int)nfit, s_sort);
                }
                if (rc != ZK_OK) break;
                if (cnt >= 2 && zk_msm_group_ok(grp, (int)cnt)) {
                    rc = zk_msm_enqueue_accum_group(ctx, grp, (int)cnt, s_sort);
                    if (rc == ZK_OK) rc = zk_msm_enqueue_reduce_group(ctx, grp, (int)cnt, s_sort);
                } else {
                    for (size_t p = 0; p < cnt && rc == ZK_OK; p++) {
                        rc = zk_msm_enqueue_accum(ctx, grp[p], s_sort);
                        if (rc == ZK_OK) rc = zk_msm_enqueue_reduce(ctx, grp[p], s_sort);
                    }
                }
            }
            if (rc == ZK_OK) {
                std::vector<ZkMsmJob*> jp(n_jobs);
                std::vector<void*> op(n_jobs);
                for (size_t k = 0; k < n_jobs; k++) { jp[k] = &jobs[k]; op[k] = outs[perm[k]]; }
                rc = zk_msm_finish_many(ctx, jp.data(), op.data(), (int)n_jobs);
            }
            (void)hipStreamSynchronize(s_sort);
            (void)hipStreamSynchronize(s_sort2);
            (void)hipStreamSynchronize(s_acc);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipEventDestroy(e0);
            return rc;
        }
    }
    std::vector<int> owner(n_jobs), sharer(n_jobs, -1);          // whose sort a job uses (itself unless it borrows), and who borrows a job's sort
    for (size_t k = 0; k < n_jobs; k++) owner[k] = (int)k;
    int rc = ZK_OK;
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
        if (rc == ZK_OK && k > 0 && scalars_dev[j] == scalars_dev[perm[k - 1]] && lens[j] == lens[perm[k - 1]] &&
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
        if (rc == ZK_OK) rc = zk_msm_enqueue_reduce(ctx, &jobs[k], ctx->stream);
    }
    for (size_t k = 0; k < n_jobs && rc == ZK_OK; k++) rc = zk_msm_finish(ctx, &jobs[k], outs[perm[k]]);
    (void)hipStreamSynchronize(s_sort);
    (void)hipStreamSynchronize(s_acc);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(e0);
    return rc;
    ZK_API_END
}


// ---- an MSM (or two over one scalar vector: a degree-bounded oracle's commitment and its shifted copy) started EARLY ----
// Marlin commits a round's oracles together (lib.rs:171-247), but some of them exist long before the round's last polynomial does:
// the mask polynomial of round 1 (prover.rs:381-399), t of round 2 (:454-462), g_2 of round 3 (:642-648) -- and none of them is
// hiding, so no draw of the prover's rng depends on where their commitment is computed.  Their jobs are enqueued here, on the sort
// and accumulate streams, as soon as the coefficients are on the device; the context stream goes on with the round's polynomial
// arithmetic (launch-bound for small proofs: the device is otherwise idle under it); the round's batch (zk_msm_batch_dev) then
// carries fewer jobs -- without its longest one in round 1 -- and zk_msm_early_finish collects the result.  Own scratch slots
// (6, 7) and pinned result buffers: nothing the batch rotates over.
struct ZkEarlyMsm {
    ZkMsmJob jobs[2];
    int count = 0;
};
int zk_msm_early_begin(zk_ctx* ctx, int count, const zk_bases* bases, const size_t* base_offsets, const void* scalars_dev, size_t len, ZkEarlyMsm** out) {
    *out = nullptr;
    if (count < 1 || count > 2 || !bases || !scalars_dev || !len) return ZK_ERR_ARG;
    for (int k = 0; k < count; k++)
        if (base_offsets[k] + len > bases->n) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_msm_early_begin: a job reads past its base table");
    zk_msm_spec_drop(ctx);                                   // (msm.hip's speculative jobs use the same scratch slots)
    ZK_TRY(zk_prover_streams(ctx, 1));
    hipStream_t s_acc = ctx->acc_stream, s_sort = ctx->aux[0];
    hipEvent_t e0;
    ZK_HIP(ctx, hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    hipError_t e = hipEventRecord(e0, ctx->stream);          // the coefficients were produced on the context stream
    if (e == hipSuccess) e = hipStreamWaitEvent(s_sort, e0, 0);
    if (e == hipSuccess) e = hipStreamWaitEvent(s_acc, e0, 0);
    (void)hipEventDestroy(e0);
    ZK_HIP(ctx, e);
    std::unique_ptr<ZkEarlyMsm> em(new ZkEarlyMsm());
    em->count = count;
    int rc = ZK_OK;
    for (int k = 0; k < count && rc == ZK_OK; k++) {
        ZkMsmJob& j = em->jobs[k];
        j.pin_key = 40 + k;
        rc = zk_msm_prepare(ctx, &j, bases, base_offsets[k], scalars_dev, len, 6 + k);
        if (rc == ZK_OK) rc = zk_msm_enqueue_sort(ctx, &j, s_sort, k ? &em->jobs[0] : nullptr);
        if (rc == ZK_OK) rc = zk_msm_enqueue_accum(ctx, &j, s_acc);
        // the reduce chain: a LARGE job's behind its accumulate kernel on the sort stream (the batch's sorts queue behind it; the
        // accumulate stream stays free for the batch's kernels); a SMALL job's (latency chains: the batch will run as a group on the
        // sort stream) on the accumulate stream, so that the two chains run side by side
        const bool small = bases->pre && len * ((255 + bases->c_pre - 1) / bases->c_pre) <= ((size_t)1 << 23);
        if (rc == ZK_OK) rc = zk_msm_enqueue_reduce(ctx, &j, small ? s_acc : s_sort);
    }
    if (rc != ZK_OK) {                                       // whatever was enqueued reads the scratch: let it drain
        (void)hipStreamSynchronize(s_sort);
        (void)hipStreamSynchronize(s_acc);
        return rc;
    }
    *out = em.release();
    return ZK_OK;
}
// outs[k]: zk_g1_projective / zk_g2_projective of job k.  Always deletes the handle.
int zk_msm_early_finish(zk_ctx* ctx, ZkEarlyMsm* em, void* const* outs) {
    if (!em) return ZK_ERR_ARG;
    std::unique_ptr<ZkEarlyMsm> own(em);
    int rc = ZK_OK;
    for (int k = 0; k < em->count; k++) {
        const int r = outs ? zk_msm_finish(ctx, &em->jobs[k], outs[k]) : ZK_OK;
        if (rc == ZK_OK) rc = r;
    }
    if (!outs || rc != ZK_OK) {                              // abandoned: its kernels may still read the scratch slots and the coefficient vector
        (void)hipStreamSynchronize(ctx->aux[0]);
        (void)hipStreamSynchronize(ctx->acc_stream);
    }
    return rc;
}
