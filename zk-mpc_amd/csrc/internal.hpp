// internal.hpp -- declarations shared between the translation units of libzkmpc_hip.
#pragma once
#include "ctx.hpp"
#include <stddef.h>
#include <stdint.h>
#include <atomic>
#include <functional>
#include <string>

// ntt.hip
void zk_domains_free(zk_ctx* ctx);
void zk_presort_free(zk_ctx* ctx);   // groth16_pipeline.hip: drop a pending zk_groth16_msms_presort_dev
extern "C" int zk_comm_destroy(zk_ctx* ctx);
extern "C" int zk_fr_sum_parties_dev(zk_ctx* ctx, const void* gathered_dev, size_t n_parties, size_t n, void* out_dev);
int zk_ntt_launch(zk_ctx* ctx, void* buf_dev, uint32_t log_n, int inverse, int coset);
int zk_ntt_launch_batch(zk_ctx* ctx, void* const* bufs_dev, int count, uint32_t log_n, int inverse, int coset);   // count <= 4, same size and kind, one launch per pass
int zk_ntt_vanishing_inv(zk_ctx* ctx, uint32_t log_n, uint32_t out9[9]);  // 1/(g^N - 1), internal form

// vec_ops.hip
// core.hip: a launch-bound sequence on ONE stream as a captured graph from its third use with the same key (see there)
int zk_graph_run(zk_ctx* ctx, const std::string& key, hipStream_t st, const std::function<int()>& enqueue);
int zk_vec_op_launch(zk_ctx* ctx, int op, const void* a, const void* b, void* out, size_t n);
int zk_vec_scale_launch(zk_ctx* ctx, const void* a, const uint32_t* k_int_form9, void* out, size_t n);
// zk_fr_vec_is_zero_dev without the wait: *verdict points at a page-locked word that is 0 (all zero) or not once ctx->stream has
// passed this point (slot 0..7: tests in flight)
int zk_fr_vec_is_zero_launch(zk_ctx* ctx, const void* v, size_t n, int slot, const uint32_t** verdict);
// out = rp (kc s + ka a + kb b) - z tp, element-wise (Marlin round 2); out may alias s
int zk_fr_outer_q1_launch(zk_ctx* ctx, const void* s, const void* a, const void* b, const void* z, const void* rp, const void* tp,
                          const uint32_t* ka_int9, const uint32_t* kb_int9, const uint32_t* kc_int9, void* out, size_t n);
// out[i] = sum_t k_t p_t[i] (terms with i < ns[t]); constants in internal form; out may not alias a term
int zk_fr_lincomb_launch(zk_ctx* ctx, int terms, const void* const* ps, const size_t* ns, const uint32_t (*k_int_form9)[9], void* out, size_t n_out);
int zk_vec_sub_scale_launch(zk_ctx* ctx, const void* a, const void* b, const uint32_t* k_int_form9, void* out, size_t n);

// msm.hip
struct zk_bases {
    int group = 1;             // 1 = G1, 2 = G2
    size_t n = 0;              // number of points
    uint32_t* dev = nullptr;   // packed affine, internal Montgomery form: n * (24|48) words
    bool owned = true;
    // optional window multiples for resident bases: pre[w*n + i] = 2^(c_pre*w) * base_i, w < W_pre (pre[0..n) = dev copy).
    // With them every window of an MSM drops into ONE bucket set (fixed_base.hip: zk_bases_precompute).
    uint32_t* pre = nullptr;
    uint32_t c_pre = 0, W_pre = 0;
    uint32_t pre_stride = 0;   // 32-bit words per point in `pre` (0: packed, 2 * WORDS).  G1: 32 = one 128-byte line per 96-byte point;
                               // 64 = limb form, line 0 the point, line 1 its negative (fixed_base.hip::k_repack_limbs)
    std::string pre_note;      // which layout `pre` has, or why the window multiples were skipped (zk_bases_precompute_note)
};
// hostxfer.hip: caller-owned (pageable) host memory <-> device through the context's page-locked ring; h2d returns once the host
// buffer has been read (the context stream waits for the DMA), d2h once the host buffer is complete
// pinned = the caller's own page-locked memory (zk_host_alloc): one DMA on the context stream, in place
int zk_xfer_h2d(zk_ctx* ctx, void* dev, const void* host, size_t bytes, bool pinned = false);
int zk_xfer_d2h(zk_ctx* ctx, void* host, const void* dev, size_t bytes, bool pinned = false);
// ... with the bytes produced / consumed piece by piece by the ring's thread team (gathers out of and scatters into the caller's own
// struct layouts); fence_ctx = false: on the DMA stream alone (see hostxfer.hip)
using ZkXferFill = std::function<void(char* dst, size_t off, size_t len)>;
using ZkXferDrain = std::function<void(const char* src, size_t off, size_t len)>;
int zk_xfer_h2d_fn(zk_ctx* ctx, void* dev, size_t bytes, const ZkXferFill& fill, bool fence_ctx,
                   const std::function<int(size_t, size_t, hipStream_t)>* after_round);
int zk_xfer_d2h_fn(zk_ctx* ctx, const void* dev, size_t bytes, const ZkXferDrain& drain);
int zk_xfer_stream(zk_ctx* ctx, hipStream_t* st);
void zk_xfer_free(zk_ctx* ctx);
bool zk_host_is_pinned(const void* host);      // page-locked by HIP (hipHostMalloc / hipHostRegister): safe to hand to hipMemcpyAsync
// a host table in the caller's own struct layout (zk_affine_layout of the ABI); off_inf == SIZE_MAX: no flag, all-zero bytes = infinity
struct ZkAffineLayout { size_t stride, off_x, off_y, off_inf; };
int zk_bases_upload_host(zk_ctx* ctx, int group, const void* host, size_t n, const ZkAffineLayout* layout, zk_bases** out);   // msm.hip
// msm.hip: an empty device-form table of n points / its content from the packed ABI form (n * 96 | 192 bytes on the device), on `st`
int zk_bases_alloc_dev(zk_ctx* ctx, int group, size_t n, zk_bases** out);
int zk_bases_import_launch(zk_ctx* ctx, zk_bases* b, const void* raw_dev, hipStream_t st);
// bases_cache.hip: the resident table for a host slice, keyed by CONTENT.  A host table as the caller holds it: packed (stride = 96 |
// 192, off_y = stride / 2, off_inf = off_tag = SIZE_MAX), GroupAffine<P> {x, y, infinity}, or that inside an MpcGroup wrapper whose
// discriminant byte at off_tag must say Public (tag_public) for every element.
struct ZkHostTable {
    int group = 1;
    const void* host = nullptr;
    size_t stride = 0, off_x = 0, off_y = 0, off_inf = SIZE_MAX, off_tag = SIZE_MAX;
    uint8_t tag_public = 0;
};
struct ZkBasesLease {
    const zk_bases* b = nullptr;    // the table to run on
    bool temporary = false;         // made for this call only (cache off / tiny / over budget): zk_bases_lease_release frees it
    bool verify = false;            // a hit by fingerprint: zk_bases_cache_verify must confirm it before the result leaves the library
    uint32_t* raw_tmp = nullptr;
    void* stage = nullptr;          // device buffer the caller's slice is compared in
    const void* entry = nullptr;
};
int zk_bases_cache_get(zk_ctx* ctx, const ZkHostTable& t, size_t n, ZkBasesLease* out);
int zk_bases_cache_verify(zk_ctx* ctx, ZkBasesLease* l, const ZkHostTable& t, size_t n, bool* same);
int zk_bases_cache_replace(zk_ctx* ctx, ZkBasesLease* l);
void zk_bases_lease_release(zk_ctx* ctx, ZkBasesLease* l);
// msm.hip: n_lanes MSMs (device scalar vectors of n elements) over ONE host table, through the cache -- verified hit and all
// scalars_fp: a fingerprint of the scalar vector taken from the caller's HOST memory (64 sampled elements and the length; 0 = none):
// what the speculation below recognises a repeated vector by -- a candidate, confirmed word for word on the device before any
// speculative result is handed out
int zk_msm_table_run(zk_ctx* ctx, const ZkHostTable& t, size_t n_table, int n_lanes, const void* const* scalars_dev, size_t n, void* const* outs,
                     uint64_t scalars_fp = 0);
// msm.hip: MSMs started ahead for the tables a caller asks for next with the SAME scalar vector (create_proof: A, then B in G1, then B
// in G2 over one `assignment`: src/groth16.rs:137-160).  drop: abandon what is in flight (waits for its kernels); forget: a table is
// leaving the cache or changing content; free: with the context.
void zk_msm_spec_drop(zk_ctx* ctx);
int zk_side_stream(zk_ctx* ctx, hipStream_t* out);                       // groth16_pipeline.hip: the context's one side stream (= aux[0])
void zk_msm_spec_forget(zk_ctx* ctx, const zk_bases* b);
void zk_msm_spec_fft_begin(zk_ctx* ctx, const void* dev, size_t N, int kind);            // msm.hip: a host-slice transform's output as the next MSM's scalars
void zk_msm_spec_fft_end(zk_ctx* ctx, size_t N, int kind, const std::function<uint64_t(size_t)>& fp_of);
uint64_t zk_scalars_fingerprint(const void* fr, size_t n);                     // msm.hip: what zk_msm_g1/_g2 recognise a repeated scalar vector by
void zk_msm_spec_free(zk_ctx* ctx);
int zk_prover_streams(zk_ctx* ctx, size_t k);      // groth16_pipeline.hip: the context's helper streams
int zk_bases_cache_poll(zk_ctx* ctx);            // publish finished window multiples, start the next build (cheap: one event query)
void zk_bases_cache_free(zk_ctx* ctx);
extern "C" int zk_bases_free(zk_ctx* ctx, zk_bases* b);
constexpr size_t ZK_PRECOMP_MIN_POINTS = 256;                    // tables below this keep the plain form (zk_bases_precompute is a no-op)
extern "C" int zk_bases_precompute(zk_ctx* ctx, zk_bases* b);
uint32_t zk_precompute_windows(size_t n);                            // fixed_base.hip: copies a table of n points would get
int zk_bases_precompute_auto(zk_ctx* ctx, zk_bases* b);             // only when ZK_PRECOMP=1 (off by default: see fixed_base.hip)
// fixed_base.hip: the window multiples of a resident table built in slices on a side stream, published by a later call
// (bases_cache.hip drives it: begin = the allocations, step = the next few hundred microseconds of launches, finish = publish / drop)
struct ZkPrecompJob {
    zk_bases* b = nullptr;
    uint32_t c = 0, W = 0, wide_words = 0;
    uint32_t *packed = nullptr, *wide = nullptr, *xy = nullptr, *scr = nullptr;
    int phase = -1;                 // -1: allocations, 0: copy of level 0, 1: levels, 2: re-laid copy, 3: scratch to free, 4: complete
    uint32_t w = 1;
    size_t pos = 0, budget = 0;
    hipError_t err = hipSuccess;
};
int zk_bases_precompute_begin(zk_ctx* ctx, zk_bases* b, size_t budget_bytes, ZkPrecompJob** out);   // *out = NULL: skipped (b->pre_note says why)
hipError_t zk_bases_precompute_step(ZkPrecompJob* j, hipStream_t st, bool* more);
int zk_bases_precompute_finish(zk_ctx* ctx, ZkPrecompJob* j, bool keep);
int zk_msm_run(zk_ctx* ctx, const zk_bases* bases, size_t base_offset, const void* scalars_dev, size_t n,
               void* out_host_projective);

// event-based phase timer (two hipEventRecord per phase on the given stream; resolved lazily)
struct ZkPhaseTimer {
    zk_ctx* ctx;
    hipStream_t stream;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> ev;
    bool enabled;
    bool resolved = false;
    explicit ZkPhaseTimer(zk_ctx* c, hipStream_t st = nullptr);
    ~ZkPhaseTimer();
    void begin(const char* name);
    void end();
    void resolve();  // waits for the recorded end events (not for the stream) and accumulates into ctx->timers
};

// An MSM in flight (msm.hip): prepare -> enqueue_sort -> enqueue_accum -> enqueue_reduce -> finish.
struct ZkMsmJob {
    int group = 1, slot = 0;
    int pin_key = -1;                 // pinned result buffer (ctx->pinned key); -1: the slot's.  Jobs enqueued without a host wait in between need their own
    size_t n = 0, max_segs = 0, max_heavy = 0;
    uint32_t c = 0, W = 0, NB = 0, seg = 0;
    uint16_t off[65] = {0};           // window w covers scalar bits [off[w], off[w+1])
    uint32_t Wb = 0;                  // bucket sets: W, or 1 when the bases carry precomputed window multiples
    uint32_t log_nb = 0, nout = 0;    // reduce phase (msm_reduce.cuh): a bucket set of 2^log_nb buckets leaves nout = log_nb + 1 points for the host
    uint32_t n_tab = 0, tab_off = 0;  // merged mode: table stride and offset of this MSM's first base
    uint32_t stride = 0;              // 32-bit words between consecutive points of bases_dev (the packed 2 * WORDS, or zk_bases::pre_stride)
    const uint32_t* bases_dev = nullptr;
    const void* scalars = nullptr;
    hipStream_t stream = nullptr;     // the stream the reduce phase (and the copy to hw) is on
    hipStream_t sort_stream = nullptr, accum_stream = nullptr;   // where sort_done / accum_done were recorded: a consumer on the same stream needs no wait (3.9 us of host time each)
    hipEvent_t sort_done = nullptr;   // recorded once sorted/desc/order are final
    hipEvent_t accum_done = nullptr;  // recorded after the accumulate kernel
    hipEvent_t reduce_done = nullptr; // recorded after the copy of the partial sums to the host: what finish() waits for
    uint32_t* hw = nullptr;           // partial window sums (XYZZ, internal form) in pinned host memory
    // the sort products, so that a later job over the same scalar vector can reuse them
    uint32_t *sorted = nullptr, *order = nullptr, *ctr = nullptr;
    void *desc = nullptr, *heavy = nullptr, *heavy2 = nullptr;
    size_t max_heavy2 = 0, max_heavy_segs = 0, max_groups = 0;
    std::vector<ZkPhaseTimer*> timers;
    ~ZkMsmJob();
};
int zk_msm_prepare(zk_ctx* ctx, ZkMsmJob* job, const zk_bases* bases, size_t base_offset, const void* scalars_dev, size_t n, int slot);
int zk_msm_enqueue_sort(zk_ctx* ctx, ZkMsmJob* job, hipStream_t st, const ZkMsmJob* share_sort);
int zk_msm_enqueue_accum(zk_ctx* ctx, ZkMsmJob* job, hipStream_t st);
int zk_msm_enqueue_reduce(zk_ctx* ctx, ZkMsmJob* job, hipStream_t st);
int zk_msm_finish(zk_ctx* ctx, ZkMsmJob* job, void* out_host_projective);
// Several SMALL G1 jobs (sorted already, tables of window multiples with the same bucket count) as one accumulate launch and one
// launch per level of the reduce chain; zk_msm_group_ok says whether a set qualifies (else: the per-job calls above)
int zk_msm_finish_many(zk_ctx* ctx, ZkMsmJob* const* jobs, void* const* outs, int count);      // the host halves side by side
bool zk_msm_group_ok(ZkMsmJob* const* jobs, int count);
// (the sorts of such a group as ONE launch, a block per job: every job a one-block sort -- prepared, not yet sorted)
bool zk_msm_sort_group_ok(ZkMsmJob* const* jobs, int count);
int zk_msm_enqueue_sort_group(zk_ctx* ctx, ZkMsmJob* const* jobs, int count, hipStream_t st);
int zk_msm_enqueue_accum_group(zk_ctx* ctx, ZkMsmJob* const* jobs, int count, hipStream_t st);
int zk_msm_enqueue_reduce_group(zk_ctx* ctx, ZkMsmJob* const* jobs, int count, hipStream_t st);

// msm_batch.hip: one or two MSMs over one scalar vector started ahead of a batch (Marlin: the oracles of a round that exist early)
struct ZkEarlyMsm;
int zk_msm_early_begin(zk_ctx* ctx, int count, const zk_bases* bases, const size_t* base_offsets, const void* scalars_dev, size_t len, ZkEarlyMsm** out);
int zk_msm_early_finish(zk_ctx* ctx, ZkEarlyMsm* em, void* const* outs);      // outs = NULL: abandon (waits for its kernels); deletes the handle

// msm_sort.hip: the bucket sort (msm_digits.cuh declares its interface)

// msm_g2pair.hip: the G2 accumulate kernel with two lanes per point addition
void zk_launch_accum_g2pair(hipStream_t st, size_t segments, const uint32_t* bases, const uint32_t* sorted, const void* desc,
                            const uint32_t* order, const uint32_t* ctr, uint32_t* sums, bool quads);
struct ZkG2PairReduce {   // the arguments of msm.hip's reduce chain
    const void *heavy, *heavy2; const uint32_t* ctr; uint32_t* done;
    uint32_t *sums, *rowP, *colP, *bits;
    uint32_t log_nb, n_win, light_blocks, heavy_blocks;
    bool quads = false;          // a small job: the fold on lane quads (the grid kernels choose by the bucket count)
};
int zk_launch_reduce_g2pair(zk_ctx* ctx, hipStream_t st, const ZkG2PairReduce& a);
constexpr uint32_t ZK_G2PAIR_RED_PTS = 128;   // points (lane pairs) per block of the G2 reduce kernels
