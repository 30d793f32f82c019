/*
 * zkmpc_hip.h -- C ABI of libzkmpc_hip.so: the MI355X (gfx950) implementation of zk-mpc's
 * Groth16 proving hot path over BLS12-377 (plain or additively secret-shared witnesses).
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * Yoii-Inc/zk-mpc tree).  The Rust-side binding a maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 (ZK_OK) or a negative error code; zk_last_error() has the text.
 *    Nothing aborts or throws across the boundary: every `int` entry point runs inside an exception barrier
 *    (std::bad_alloc -> ZK_ERR_NOMEM, any other C++ exception -> ZK_ERR_STATE).
 *  - a context OWNS its device: every entry point that takes a zk_ctx makes ctx's device the calling thread's current HIP
 *    device for the duration of the call and restores the previous one on return, so several parties on several GPUs can
 *    live in one process (the reference's LocalTestNet, mpc-net/src/multi.rs:419-443) whatever device their threads hold.
 *  - "host" pointers are caller-owned and never retained.  "dev" pointers are HIP device
 *    pointers on the context's device.  Opaque handles are library-owned.
 *  - Field elements use the reference's in-memory form: little-endian u64 limbs in
 *    Montgomery form (Fr: 4 limbs, R = 2^256; Fq: 6 limbs, R = 2^384),
 *    arkworks/algebra/ff/src/fields/macros.rs:107-112.  Every output is fully reduced.
 *  - Points: zk_g1_affine = x|y (96 B), zk_g2_affine = x.c0|x.c1|y.c0|y.c1 (192 B); the point
 *    at infinity is all-zero bytes (the binding maps GroupAffine.infinity to that).
 *    Projective results are Jacobian (X,Y,Z) like GroupProjective, normalised to Z = 1
 *    (or (1,1,0) for zero, short_weierstrass_jacobian.rs zero()).
 *  - All calls on one context are serialised on that context's HIP stream; contexts are
 *    independent (one per MPC party / GPU, several per process allowed).
 */
#ifndef ZKMPC_HIP_H
#define ZKMPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_OK 0
#define ZK_ERR_HIP -1
#define ZK_ERR_ARG -2
#define ZK_ERR_NOMEM -3
#define ZK_ERR_STATE -4
#define ZK_ERR_MAC -5     /* a SPDZ MAC check failed (SpdzFieldShare / SpdzGroupShare batch_open: the reference asserts) */

typedef struct zk_ctx zk_ctx;
typedef struct zk_bases zk_bases;   /* device-resident MSM base table (G1 or G2) */
typedef struct zk_r1cs zk_r1cs;     /* device-resident ConstraintMatrices (CSR) */
typedef struct zk_pk zk_pk;         /* device-resident Groth16 ProvingKey */
typedef struct zk_rng zk_rng;       /* host generator: FiatShamirRng / ChaChaRng / StdRng (see "randomness" below) */

typedef struct { uint64_t l[4]; } zk_fr;                 /* Fp256<FrParameters> */
typedef struct { uint64_t l[6]; } zk_fq;                 /* Fp384<FqParameters> */
typedef struct { zk_fq x, y; } zk_g1_affine;             /* GroupAffine<g1::Parameters> sans flag */
typedef struct { zk_fq x, y, z; } zk_g1_projective;      /* GroupProjective<g1::Parameters> */
typedef struct { zk_fq x[2], y[2]; } zk_g2_affine;       /* x = c0 + c1 u */
typedef struct { zk_fq x[2], y[2], z[2]; } zk_g2_projective;

/* ---- context ------------------------------------------------------------------------- */
/* One per party; replaces the task-local MpcMultiNet + implicit CPU state
 * (mpc-net/src/multi.rs:598-663). */
int zk_ctx_create(int device, int party_id, int n_parties, zk_ctx** out);
int zk_ctx_destroy(zk_ctx* ctx);
const char* zk_last_error(zk_ctx* ctx);
int zk_ctx_sync(zk_ctx* ctx);
void* zk_ctx_stream(zk_ctx* ctx);          /* hipStream_t, for event timing by the caller */
int zk_version(void);
/* Diagnostic: raises a C++ exception INSIDE the boundary wrapper of this library and returns what a caller sees -- kind 0:
 * std::bad_alloc (-> ZK_ERR_NOMEM), 1: std::runtime_error, 2: a non-std object, 3: std::system_error (-> ZK_ERR_STATE);
 * kind 4 runs a helper task through the thread-exhaustion fallback (-> ZK_OK).  Needs no device. */
int zk_selftest_exception_barrier(int kind);

int zk_dev_alloc(zk_ctx* ctx, size_t bytes, void** dev_out);
int zk_dev_free(zk_ctx* ctx, void* dev);
/* Page-locked host memory for buffers that cross the boundary on the proving path (assignment vectors). */
int zk_host_alloc(zk_ctx* ctx, size_t bytes, void** host_out);
int zk_host_free(zk_ctx* ctx, void* host);
int zk_memcpy_h2d(zk_ctx* ctx, void* dev, const void* host, size_t bytes);
int zk_memcpy_d2h(zk_ctx* ctx, void* host, const void* dev, size_t bytes);
int zk_memcpy_d2d(zk_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);   /* asynchronous on the context stream */
int zk_dev_zero(zk_ctx* ctx, void* dev, size_t bytes);

/* ---- Fr vector arithmetic ---------------------------------------- */
#define ZK_OP_MUL 0
#define ZK_OP_ADD 1
#define ZK_OP_SUB 2
#define ZK_OP_NEG 3   /* SHE vectors only */
/* out[i] = a[i] (op) b[i], device buffers of n zk_fr (out may alias a or b).
 * Fp256::{mul,add,sub}_assign: ff/src/fields/arithmetic.rs:7-57, macros.rs:698-717. */
int zk_fr_vec_op_dev(zk_ctx* ctx, int op, const void* a_dev, const void* b_dev, void* out_dev, size_t n);
/* out[i] = a[i] * k  (k: one host zk_fr). */
int zk_fr_vec_scale_dev(zk_ctx* ctx, const void* a_dev, const zk_fr* k_host, void* out_dev, size_t n);
/* Field::batch_product_in_place(selfs, others): ff/src/fields/mod.rs:216-220 (host slices). */
int zk_fr_batch_product_in_place(zk_ctx* ctx, zk_fr* selfs_host, const zk_fr* others_host, size_t n);

/* ---- radix-2 NTT (row a7) -------------------------------------------------------------- */
/* In-place size-2^log_n transform of a device vector of zk_fr, in-order in and out.
 * inverse=0,coset=0: EvaluationDomain::fft_in_place        (poly/src/domain/radix2/mod.rs:98-101)
 * inverse=1,coset=0: ifft_in_place (includes * size_inv)   (:104-107, radix2/fft.rs:26-29)
 * inverse=0,coset=1: coset_fft_in_place  (x g^i first)     (domain/mod.rs:138-141)
 * inverse=1,coset=1: coset_ifft_in_place (x g^-i after)    (radix2/mod.rs:110-113, fft.rs:31-35) */
int zk_fr_ntt_dev(zk_ctx* ctx, void* buf_dev, uint32_t log_n, int inverse, int coset);
/* Host-slice form (the trait methods take &mut Vec<F>): `n` elements are read, zero-padded to
 * 2^log_n (Vec::resize in the reference), transformed; 2^log_n elements are written back. */
int zk_fr_fft_in_place(zk_ctx* ctx, zk_fr* vec_host, size_t n, uint32_t log_n, int inverse, int coset);
/* EvaluationDomain::divide_by_vanishing_poly_on_coset_in_place (domain/mod.rs:183-190). */
int zk_fr_divide_by_vanishing_on_coset_dev(zk_ctx* ctx, void* evals_dev, uint32_t log_n);
/* Host-slice form (the trait method takes &mut [F]): 2^log_n elements in place. */
int zk_fr_divide_by_vanishing_on_coset_in_place(zk_ctx* ctx, zk_fr* evals_host, uint32_t log_n);

/* ---- variable-base MSM (rows a5, a6) ---------------------------------------------------- */
/* AffineCurve::multi_scalar_mul(bases, scalars) (ec/src/lib.rs:305-314) for G1 / G2:
 * scalars are Montgomery-form zk_fr (the callee converts, like into_repr at :308-310);
 * uses min(n_bases, n_scalars) terms (msm/variable_base.rs:15-17). */
int zk_msm_g1(zk_ctx* ctx, const zk_g1_affine* bases_host, size_t n_bases,
              const zk_fr* scalars_host, size_t n_scalars, zk_g1_projective* out_host);
int zk_msm_g2(zk_ctx* ctx, const zk_g2_affine* bases_host, size_t n_bases,
              const zk_fr* scalars_host, size_t n_scalars, zk_g2_projective* out_host);
/* The same on a table in the CALLER'S OWN struct layout -- arkworks' GroupAffine<P> is {x, y, infinity: bool} laid out by rustc
 * (104 / 200 bytes per point), so a binding that wants the repr(C) form above has to copy the whole slice on every call.  Here it
 * passes the address of the slice and where the fields sit: stride = size_of::<GroupAffine<P>>(), off_x / off_y = the byte
 * offsets of the coordinates (Montgomery words as above; G2: c0 then c1 at off_x, off_y), off_infinity = the byte offset of the
 * flag (non-zero = infinity) or SIZE_MAX when there is none (then all-zero coordinates mean infinity). */
typedef struct { size_t stride, off_x, off_y, off_infinity; } zk_affine_layout;
int zk_msm_g1_strided(zk_ctx* ctx, const void* bases_host, size_t n_bases, const zk_affine_layout* layout,
                      const zk_fr* scalars_host, size_t n_scalars, zk_g1_projective* out_host);
int zk_msm_g2_strided(zk_ctx* ctx, const void* bases_host, size_t n_bases, const zk_affine_layout* layout,
                      const zk_fr* scalars_host, size_t n_scalars, zk_g2_projective* out_host);
/* The MSM entry points on host slices (the four above, zk_mpc_msm_g1 / _g2 below) keep the tables they are shown: a base slice of
 * >= 256 points stays resident in HBM, keyed by its CONTENT -- (group, length, a 64-bit fingerprint of 64 points spread over the
 * slice, taken in the packed form whatever layout the caller holds the points in) -- never by its address: the collaborative
 * caller presents the same table in a fresh Vec on every call (MpcGroup::all_public_or_shared, mpc-algebra/src/wire/group.rs:441-457).
 * A fingerprint match is only a candidate: by default the caller's WHOLE slice crosses PCIe once more, under the MSM that already
 * runs on the cached table, and is compared with it on the device word for word; if it differs (a table rewritten in place at
 * points the sample misses) the entry takes the new content and the MSM runs again -- no call returns a sum over a stale table.
 * From its `precompute_after`-th re-use on (default 1) a slice of >= 256 points gets window multiples (zk_bases_precompute),
 * built on a side stream beside later calls and published when finished; until then the plain table serves.
 *   zk_bases_cache_config  budget_bytes: HBM the cache may hold, least recently used out first (default: a quarter of the device
 *                          memory; 0 switches the cache off and frees it); precompute_after: 0 = never build window multiples
 *   zk_bases_cache_trust   fingerprint_only = 1: a fingerprint match IS a hit and nothing but the 64 sampled points is read from
 *                          the host -- for a caller that vouches for its tables (the queries of a ProvingKey that outlives the
 *                          prover: 2.4 ms of PCIe and host bandwidth less per 2^20-point G1 call); 0 (default): verified hits
 *   zk_bases_cache_drop    forget every table (frees the HBM)
 *   zk_bases_cache_sync    finish every window-multiple build that is running or due, and publish it (benchmarks, tests)
 *   zk_bases_cache_stats   out[0..9] = hits, misses, evictions, replaced (a hit whose full comparison failed: the entry took the
 *                          caller's content), uncached uploads, entries, entries with window multiples, resident bytes, bytes
 *                          uploaded in all, budget
 *   zk_bases_cache_stats2  out[0..3] = hits verified in full, bytes compared, window-multiple builds published, 1 while a build runs */
int zk_bases_cache_config(zk_ctx* ctx, size_t budget_bytes, int precompute_after);
int zk_bases_cache_trust(zk_ctx* ctx, int fingerprint_only);
int zk_bases_cache_drop(zk_ctx* ctx);
int zk_bases_cache_sync(zk_ctx* ctx);
int zk_bases_cache_stats(zk_ctx* ctx, uint64_t out[10]);
int zk_bases_cache_stats2(zk_ctx* ctx, uint64_t out[4]);
/* The MSM entry points on host slices also look one call ahead.  The unchanged create_proof asks for three MSMs over ONE scalar
 * vector, one call behind the other (calculate_coeff over &pk.a_query, &pk.b_g1_query, &pk.b_g2_query and `assignment`:
 * src/groth16.rs:137-160); a context remembers which table followed which with the same scalars and, the next time the first of them
 * is asked for, starts the MSMs over the others as well -- on a private copy of the scalars, on side streams.  A later call takes such
 * a result only if it names that table and its scalars are word for word the ones the job ran on (compared on the device); anything
 * else drops it.  Likewise for `h = witness_map(..)`: the output of the last host-slice transform is the scalar vector of the call
 * that follows (src/groth16.rs:100-106), so that MSM starts when such a transform ends -- same rule for its release.
 * Small circuits gain 12 - 20 % per proof, large ones 3 - 6 %.  Contexts that SHARE one GPU (several parties of one
 * process on one device) do better without: zk_msm_speculate(ctx, 0).  stats: out[0..2] = jobs started ahead, results taken, dropped. */
int zk_msm_speculate(zk_ctx* ctx, int on);
int zk_msm_speculate_stats(zk_ctx* ctx, uint64_t out[3]);
/* Resident bases: upload once (proving-key queries), then MSM against device scalars. */
int zk_bases_upload_g1(zk_ctx* ctx, const zk_g1_affine* bases_host, size_t n, zk_bases** out);
int zk_bases_upload_g2(zk_ctx* ctx, const zk_g2_affine* bases_host, size_t n, zk_bases** out);
int zk_bases_free(zk_ctx* ctx, zk_bases* b);
/* Trade HBM for work on a resident table: store 2^(c w) * base_i for w < ceil(255 / c), c ~ log2(n) capped at 20
 * (13x the memory at 2^20), so that all digits of later MSMs share ONE bucket set and a scalar has 13 digits instead
 * of 16; small tables get windows of ~log2(n) + 2 bits: latency, not throughput).  No-op for tables under 256 points.
 * zk_pk_upload / zk_groth16_setup apply it to proving-key queries of >= 256 points unless ZK_PRECOMP=0 (measured at 2^20: 35.9 -> 32.4 ms per proof). */
int zk_bases_precompute(zk_ctx* ctx, zk_bases* b);
/* The same with the table's memory layout chosen by the caller instead of by the memory budget (tests, diagnostics): 0 = as
 * zk_bases_precompute; 1 = packed; 2 = one 96-byte point per 128-byte line (G1); 3 = 29-bit limbs with both signs, 256 bytes per
 * point (G1: what the accumulate kernel reads without unpacking).  A forced layout that does not fit a third of the free device
 * memory fails with ZK_ERR_NOMEM; zk_bases_precompute never fails for lack of memory: it keeps the plain table. */
int zk_bases_precompute_as(zk_ctx* ctx, zk_bases* b, int layout);
/* Which layout the table's window multiples have, or -- when zk_bases_precompute skipped them -- why (also left in
 * zk_last_error): a table without them runs 16 digits per scalar over 16 bucket sets instead of 13 over one.  "" = never asked. */
const char* zk_bases_precompute_note(const zk_bases* b);
/* Window width of the table's precomputed multiples (0: none): MSMs over it use ceil(255 / c) digits per scalar. */
uint32_t zk_bases_window_bits(const zk_bases* b);
size_t zk_bases_len(const zk_bases* b);
/* out = sum_{i<n} scalars[i] * bases[base_offset + i]; scalars_dev: n Montgomery zk_fr on device. */
int zk_msm_g1_dev(zk_ctx* ctx, const zk_bases* bases, size_t base_offset, const void* scalars_dev,
                  size_t n, zk_g1_projective* out_host);
int zk_msm_g2_dev(zk_ctx* ctx, const zk_bases* bases, size_t base_offset, const void* scalars_dev,
                  size_t n, zk_g2_projective* out_host);
/* Fixed-base batch: out[i] = scalars[i] * G (G1/G2 generator scaled by gen_k), affine results on
 * device as a zk_bases table.  FixedBaseMSM::multi_scalar_mul + batch_normalization_into_affine
 * (ec/src/msm/fixed_base.rs:11-95, arkworks/groth16/src/generator.rs:130-215). */
int zk_fixed_base_g1_dev(zk_ctx* ctx, const zk_fr* gen_k_host, const void* scalars_dev, size_t n, zk_bases** out);
int zk_fixed_base_g2_dev(zk_ctx* ctx, const zk_fr* gen_k_host, const void* scalars_dev, size_t n, zk_bases** out);
int zk_bases_download_g1(zk_ctx* ctx, const zk_bases* b, size_t offset, size_t n, zk_g1_affine* out_host);
int zk_bases_download_g2(zk_ctx* ctx, const zk_bases* b, size_t offset, size_t n, zk_g2_affine* out_host);

/* ---- arkworks CanonicalSerialize of point tables ----
 * Bytes as GroupAffine::serialize (compressed: x with bit 7 of the last byte = y > -y, bit 6 = infinity) or
 * serialize_uncompressed (x | y, infinity = (0, 1) with bit 6 of the last byte) write them
 * (ec/src/models/short_weierstrass_jacobian.rs:847-883); n points back to back, no length prefix. */
size_t zk_point_serialized_size(int group, int compressed);      /* 48 / 96 (G1), 96 / 192 (G2) */
int zk_bases_serialize(zk_ctx* ctx, const zk_bases* b, size_t offset, size_t n, int compressed, uint8_t* out_host);
/* GroupAffine::deserialize_unchecked (:930-942) on n uncompressed points, plus an on-curve check (error instead of
 * a silently wrong table); no subgroup check. */
int zk_bases_deserialize_uncompressed(zk_ctx* ctx, int group, const uint8_t* bytes_host, size_t n, zk_bases** out);
/* GroupAffine::deserialize (:888-905) on n compressed points: y from x by a square root in Fq / Fq2 (get_point_from_x,
 * :110-125), sign from the flag; error if an x is not on the curve; no subgroup check. */
int zk_bases_deserialize_compressed(zk_ctx* ctx, int group, const uint8_t* bytes_host, size_t n, zk_bases** out);

/* CanonicalSerialize framing of the Groth16 keys (arkworks/groth16/src/data_structures.rs:43-58,133-151) and of a KZG10
 * UniversalParams (poly-commit/src/kzg10/data_structures.rs:40-80, the file save_srs_to_file writes: src/marlin.rs:371-376), in the
 * compressed or the uncompressed form, so that keys and SRS files interchange with the Rust side.  The *_size functions give
 * the exact byte count; serialisers fail if `cap` is smaller. */
size_t zk_vk_serialized_size(const zk_pk* pk, int compressed);
size_t zk_pk_serialized_size(const zk_pk* pk, int compressed);
int zk_vk_serialize(zk_ctx* ctx, const zk_pk* pk, int compressed, uint8_t* out_host, size_t cap);
int zk_pk_serialize(zk_ctx* ctx, const zk_pk* pk, int compressed, uint8_t* out_host, size_t cap);
int zk_pk_deserialize(zk_ctx* ctx, const uint8_t* bytes_host, size_t len, int compressed, zk_pk** out);   /* resident key, like zk_pk_upload */
size_t zk_kzg_srs_serialized_size(size_t n_powers_g, size_t n_powers_gamma_g, int compressed);
int zk_kzg_srs_serialize(zk_ctx* ctx, const zk_bases* powers_g, const zk_bases* powers_gamma_g, const zk_g2_affine* h,
                         const zk_g2_affine* beta_h, int compressed, uint8_t* out_host, size_t cap);
int zk_kzg_srs_deserialize(zk_ctx* ctx, const uint8_t* bytes_host, size_t len, int compressed, zk_bases** powers_g,
                           zk_bases** powers_gamma_g, zk_g2_affine* h, zk_g2_affine* beta_h);

/* n_jobs independent MSMs pipelined over the library's sort / accumulate streams (one job sorts while the previous one
 * accumulates).  outs[k] receives a zk_g1_projective or zk_g2_projective according to bases[k]'s group; base_offsets may
 * be NULL.  Same result as n_jobs calls of zk_msm_g1_dev / zk_msm_g2_dev. */
int zk_msm_batch_dev(zk_ctx* ctx, size_t n_jobs, const zk_bases* const* bases, const size_t* base_offsets,
                     const void* const* scalars_dev, const size_t* lens, void* const* outs);

/* ---- host-side group helpers (O(1) work per proof; rows a10, a12) ------------------------ */
int zk_g1_add(const zk_g1_projective* a, const zk_g1_projective* b, zk_g1_projective* out);
int zk_g2_add(const zk_g2_projective* a, const zk_g2_projective* b, zk_g2_projective* out);
int zk_g1_neg(const zk_g1_projective* a, zk_g1_projective* out);
int zk_g2_neg(const zk_g2_projective* a, zk_g2_projective* out);
int zk_g1_mul(const zk_g1_projective* a, const zk_fr* k, zk_g1_projective* out);   /* ProjectiveCurve::mul */
int zk_g2_mul(const zk_g2_projective* a, const zk_fr* k, zk_g2_projective* out);
int zk_g1_from_affine(const zk_g1_affine* a, zk_g1_projective* out);
int zk_g2_from_affine(const zk_g2_affine* a, zk_g2_projective* out);
/* CanonicalSerialize compressed: 48 / 96 bytes (short_weierstrass_jacobian.rs:847-859). */
int zk_g1_serialize(const zk_g1_projective* a, uint8_t out[48]);
int zk_g2_serialize(const zk_g2_projective* a, uint8_t out[96]);
/* Fr helpers for the share algebra done on the host (scalars r, s and Beaver opens). */
int zk_fr_add(const zk_fr* a, const zk_fr* b, zk_fr* out);
int zk_fr_sub(const zk_fr* a, const zk_fr* b, zk_fr* out);
int zk_fr_mul(const zk_fr* a, const zk_fr* b, zk_fr* out);
/* Fp384 add / sub / mul on Montgomery-form values (macros.rs:698-717, arithmetic.rs:7-57), computed with the
 * device's own field templates (radix 2^29) on the host. */
int zk_fq_add(const zk_fq* a, const zk_fq* b, zk_fq* out);
int zk_fq_sub(const zk_fq* a, const zk_fq* b, zk_fq* out);
int zk_fq_mul(const zk_fq* a, const zk_fq* b, zk_fq* out);
int zk_fq_neg5_almost_raw(const uint32_t a13[13], uint32_t out13[13]);   /* test hook: raw 29-bit limbs of k p - 5 a (fp29.cuh::fp_neg5_almost) */
int zk_fr_lazy_raw(int op, const uint32_t* in9s, uint32_t* out9s);     /* test hook: the lazy Fr domain of the NTT butterflies (frlazy.cuh) on raw 29-bit limbs */
int zk_fq_lazy_raw(int op, const uint32_t* in13s, uint32_t* out13s);   /* test hook: the lazy-domain primitives of fp29.cuh / ec.cuh on raw 29-bit limbs (hostapi.hip lists the ops) */
int zk_fq_mul2(const zk_fq* a, const zk_fq* b, const zk_fq* c, const zk_fq* d, zk_fq* out);   /* a b + c d, the fused double product of the Fq2 multiplication */
int zk_fr_inverse(const zk_fr* a, zk_fr* out);                    /* Field::inverse (macros.rs:389-443); error on zero */
int zk_fr_pow(const zk_fr* a, uint64_t e, zk_fr* out);           /* Field::pow */
int zk_fr_from_canonical(const uint64_t canon[4], zk_fr* out);   /* from_repr, macros.rs:464-474 */
int zk_fr_to_canonical(const zk_fr* a, uint64_t canon[4]);       /* into_repr, arithmetic.rs:59-83 */

/* ---- R1CS + Groth16 (rows a9, a10) ----------------------------------------------------- */
/* ConstraintMatrices{a,b,c} as CSR (relations/src/r1cs/constraint_system.rs:650-676):
 * row_ptr has num_constraints+1 entries; coeffs are Montgomery zk_fr; col indexes the full
 * assignment (instance first, instance[0] = 1, then witness). */
typedef struct {
    size_t num_constraints, num_instance, num_witness;
    const uint32_t* a_row_ptr; const uint32_t* a_col; const zk_fr* a_coeff;
    const uint32_t* b_row_ptr; const uint32_t* b_col; const zk_fr* b_coeff;
    const uint32_t* c_row_ptr; const uint32_t* c_col; const zk_fr* c_coeff;
} zk_r1cs_host;
int zk_r1cs_upload(zk_ctx* ctx, const zk_r1cs_host* m, zk_r1cs** out);
int zk_r1cs_free(zk_ctx* ctx, zk_r1cs* r);
/* The SURVEY 8(d) synthetic "mul-chain" R1CS built directly on the device: w_i*w_{i+1}=w_{i+2},
 * n constraints, 2 instance variables (1, last product), n+1 witness variables. */
int zk_r1cs_mul_chain(zk_ctx* ctx, size_t n, zk_r1cs** out);
/* ... and its satisfying full assignment from seeds w0, w1 (device vector of n+3 zk_fr). */
int zk_mul_chain_assignment_dev(zk_ctx* ctx, size_t n, const zk_fr* w0, const zk_fr* w1, void* z_dev);

/* ProvingKey (arkworks/groth16/src/data_structures.rs:133-151), host view for upload. */
typedef struct {
    zk_g1_affine alpha_g1, beta_g1, delta_g1;
    zk_g2_affine beta_g2, delta_g2;
    const zk_g1_affine* a_query;    size_t a_len;
    const zk_g1_affine* b_g1_query; size_t b_g1_len;
    const zk_g2_affine* b_g2_query; size_t b_g2_len;
    const zk_g1_affine* h_query;    size_t h_len;
    const zk_g1_affine* l_query;    size_t l_len;
} zk_pk_host;
int zk_pk_upload(zk_ctx* ctx, const zk_pk_host* pk, zk_pk** out);
int zk_pk_free(zk_ctx* ctx, zk_pk* pk);
/* generate_parameters with explicit toxic waste (arkworks/groth16/src/generator.rs:44-231);
 * `tau` is the evaluation point the reference samples with sample_element_outside_domain.
 * g1 = g1_k * G1 generator, g2 = g2_k * G2 generator.  The key stays resident on the device. */
int zk_groth16_setup(zk_ctx* ctx, const zk_r1cs* r1cs, const zk_fr* alpha, const zk_fr* beta,
                     const zk_fr* gamma, const zk_fr* delta, const zk_fr* tau,
                     const zk_fr* g1_k, const zk_fr* g2_k, zk_pk** out);
/* Sizes / elements of a resident key, for inspection by tests and serialisers.
 * which: 0=a_query 1=b_g1_query 2=b_g2_query 3=h_query 4=l_query 5=gamma_abc_g1 */
size_t zk_pk_query_len(const zk_pk* pk, int which);
/* Borrowed handle to one query table (same `which`); owned by the key, valid until zk_pk_free. */
const zk_bases* zk_pk_query_bases(const zk_pk* pk, int which);
int zk_pk_download_g1(zk_ctx* ctx, const zk_pk* pk, int which, size_t offset, size_t n, zk_g1_affine* out);
int zk_pk_download_g2(zk_ctx* ctx, const zk_pk* pk, int which, size_t offset, size_t n, zk_g2_affine* out);
/* vk elements: 0=alpha_g1 1=beta_g1 2=delta_g1 (G1);  0=beta_g2 1=delta_g2 2=gamma_g2 (G2) */
int zk_pk_vk_g1(const zk_pk* pk, int which, zk_g1_affine* out);
int zk_pk_vk_g2(const zk_pk* pk, int which, zk_g2_affine* out);

/* R1CStoQAP::witness_map (src/groth16.rs:240-306): z_dev = full assignment (num_instance +
 * num_witness zk_fr on device); h_dev receives domain_size zk_fr. */
int zk_groth16_witness_map_dev(zk_ctx* ctx, const zk_r1cs* r1cs, const void* z_dev, void* h_dev);
uint32_t zk_r1cs_domain_log(const zk_r1cs* r1cs);
/* Collaborative form, split at the one shared x shared product (src/groth16.rs:285):
 *  pre : a,b <- coset_fft(ifft(M z)), c <- ifft(C z)      (local: linear in the shares)
 *  post: h <- (coset_ifft(ab) - c) / Z(g), ab supplied by the caller (Beaver batch_mul).
 * c is an opaque value between the two calls: it stays in coefficient form, because the reference's
 * coset_ifft((ab - c) / Z(g)) over coset evaluations equals coset_ifft(ab) - c over coefficients (interpolation is linear)
 * -- the same h, one transform less. */
int zk_groth16_witness_map_pre_dev(zk_ctx* ctx, const zk_r1cs* r1cs, const void* z_dev,
                                   int include_instance, void* a_dev, void* b_dev, void* c_dev);
int zk_groth16_witness_map_post_dev(zk_ctx* ctx, const zk_r1cs* r1cs, void* ab_dev, const void* c_dev);
/* The five MSMs of create_proof (src/groth16.rs:106,110,137,148,160) on a resident key:
 * out = [h_acc, l_aux_acc, a_acc, b_g1_acc] and b_g2_acc; z_dev as above, h_dev from witness_map.
 * (For shares these are the party-local MSMs of multi_scale_pub_group, share/additive.rs:517-520.) */
/* Optional, for a prover that works through a queue of assignments (same key, same constraint system): announce the
 * assignment of the NEXT zk_groth16_prove_dev call.  The proof in between then enqueues that proof's front (sort of z,
 * witness map, sort of h) behind its own kernels, where it runs under the reduce tail and the host time between two
 * proofs.  z_next_dev must stay unchanged until that proof is done; NULL withdraws; a different next call simply drops it. */
int zk_groth16_hint_next_dev(zk_ctx* ctx, const void* z_next_dev);
/* For a context that shares its GPU with other contexts proving small circuits (several host threads, one context each): whether
 * the front enqueued for an announced small proof (domain <= 2^16) also carries that proof's accumulate launches and reduce chains.
 * Default 1: one context working through a queue gains ~20 % (the device runs from one proof into the next while the host
 * finishes the first); 0 leaves the hardware queues to the other contexts between two proofs. */
int zk_groth16_chain_fronts(zk_ctx* ctx, int on);
/* Optional: enqueue the sort of z[1..] that four of those MSMs share, ahead of zk_groth16_msms_dev on the same z_dev
 * (asynchronous; z_dev must stay unchanged).  The collaborative prover calls it before the Beaver open of
 * mpc-algebra/src/share/field.rs:97-129 so that the sort runs during the exchange.  Dropped if another MSM batch or a
 * different z_dev comes first. */
int zk_groth16_msms_presort_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r1cs, const void* z_dev);
/* The four MSMs over z (B in G2, A, B in G1, L) enqueued to the end; zk_groth16_msms_dev with the same pk / r / z then adds the H
 * job and collects all five: the collaborative prover's Beaver open and second witness-map half run under them. */
int zk_groth16_msms_begin_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z);
int zk_groth16_msms_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r1cs, const void* z_dev,
                        const void* h_dev, zk_g1_projective out_g1[4], zk_g2_projective* out_g2);
/* create_proof (src/groth16.rs:68-183 / arkworks/groth16/src/prover.rs:44-153) for a plain
 * (non-shared) assignment resident on the device; proof = a||b||c compressed, 192 bytes. */
int zk_groth16_prove_dev(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r1cs, const void* z_dev,
                         const zk_fr* r, const zk_fr* s, uint8_t proof_out[192]);
/* The same proof with the five MSMs of src/groth16.rs:106,110,137,148,160 spread over n_ctx contexts -- one per GPU of a node; for a
 * functional run several on one GPU -- each with ITS OWN copy of the proving key and the constraint system (pks[i], r1css[i]
 * belong to ctxs[i]).  The work is cut by cost into base ranges (a G2 term weighs 2.7 G1 terms), a context holding a piece of H runs
 * the witness map itself, the assignment (on ctxs[0]'s device) reaches the others by one peer copy, the partial sums are added
 * on the host: same 192 bytes as zk_groth16_prove_dev.  Errors of another context are reported through ctxs[0]. */
int zk_groth16_prove_multi(zk_ctx* const* ctxs, const zk_pk* const* pks, const zk_r1cs* const* r1css, int n_ctx, const void* z_dev0,
                           const zk_fr* r, const zk_fr* s, uint8_t proof_out[192]);
/* How that call deals a proof of this shape: one text line per piece, "ctx job lo n" with job 0 = B in G2, 1 = A, 2 = B in G1,
 * 3 = L, 4 = H and [lo, lo + n) the job's terms; *written = the bytes put into out (ZK_ERR_ARG when cap is too small). */
int zk_groth16_multi_plan(const zk_pk* pk, const zk_r1cs* r1cs, int n_ctx, char* out, size_t cap, size_t* written);
/* Same with the assignment in host memory (instance then witness): SURVEY 8(d)'s "witness vector on host to 192 proof
 * bytes on host". */
int zk_groth16_prove(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r1cs, const zk_fr* z_host,
                     const zk_fr* r, const zk_fr* s, uint8_t proof_out[192]);
/* The same for a prover that works through a queue of host assignments: z_next_host (or NULL) is the assignment of the
 * NEXT call.  It is uploaded at once on a copy stream (two device slots alternate) and this proof enqueues its front as
 * zk_groth16_hint_next_dev does; the next call, naming that buffer (matched by address: it must stay unchanged in between),
 * copies nothing.  Page-locked buffers (zk_host_alloc) make the upload asynchronous. */
int zk_groth16_prove_queued(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r1cs, const zk_fr* z_host,
                            const zk_fr* r, const zk_fr* s, const zk_fr* z_next_host, uint8_t proof_out[192]);

/* ---- dense polynomials over Fr and KZG10 (row a14: the data-parallel pieces of the Marlin / poly-commit path) ---- */
/* out[i] = start * base^i, i < n (device vector). */
int zk_fr_powers_dev(zk_ctx* ctx, const zk_fr* base, const zk_fr* start, size_t n, void* out_dev);
/* ark_ff::batch_inversion in place; zero entries stay zero (ff/src/fields/mod.rs:597-659). */
int zk_fr_batch_inverse_dev(zk_ctx* ctx, void* v_dev, size_t n);
/* DensePolynomial::evaluate (poly/src/polynomial/univariate/dense.rs:53-75); n coefficients, low degree first. */
int zk_poly_evaluate_dev(zk_ctx* ctx, const void* coeffs_dev, size_t n, const zk_fr* point, zk_fr* out);
/* The same for `count` (polynomial, point) pairs in two launches and one copy back: out[i] = polys[i](points[i]).  What
 * Marlin::prove's evaluation step (marlin/src/lib.rs:262-292: every polynomial of the query set at beta or gamma) calls. */
typedef struct { const void* ptr; size_t n; } zk_poly_ref;   /* device coefficients, low degree first */
int zk_poly_evaluate_batch_dev(zk_ctx* ctx, const zk_poly_ref* polys, const zk_fr* points, size_t count, zk_fr* out);
/* p / (X - z): quotient (n-1 coefficients) and remainder p(z) (KZG10::compute_witness_polynomial, kzg10/mod.rs:212-235). */
int zk_poly_divide_by_linear_dev(zk_ctx* ctx, const void* coeffs_dev, size_t n, const zk_fr* z, void* q_dev, zk_fr* rem);
/* DensePolynomial::divide_by_vanishing_poly for the radix-2 domain of size 2^log_domain (dense.rs:166-173):
 * q gets max(n - N, 0) coefficients, r gets N (both zero-padded, not truncated). */
int zk_poly_divide_by_vanishing_dev(zk_ctx* ctx, const void* coeffs_dev, size_t n, uint32_t log_domain, void* q_dev, void* r_dev);
/* DensePolynomial multiplication via FFT (dense.rs:568-584): out gets na + nb - 1 coefficients. */
int zk_poly_mul_dev(zk_ctx* ctx, const void* a_dev, size_t na, const void* b_dev, size_t nb, void* out_dev);
/* KZG10::commit (poly-commit/src/kzg10/mod.rs:142-205): MSM(powers_of_g, coeffs) [+ MSM(powers_of_gamma_g, blinding)]. */
int zk_kzg_commit_dev(zk_ctx* ctx, const zk_bases* powers_g, const void* coeffs_dev, size_t n,
                      const zk_bases* powers_gamma_g, const void* blind_dev, size_t n_blind, zk_g1_projective* out);
/* KZG10::open (kzg10/mod.rs:212-293): w = commit(p/(X-z)) [+ commit_gamma(blind/(X-z)), random_v = blind(z)]. */
int zk_kzg_open_dev(zk_ctx* ctx, const zk_bases* powers_g, const void* coeffs_dev, size_t n, const zk_fr* point,
                    const zk_bases* powers_gamma_g, const void* blind_dev, size_t n_blind,
                    zk_g1_projective* w_out, zk_fr* random_v_out);

/* ---- Marlin AHP prover pieces (row a14; marlin/src/ahp/prover.rs) ---- */
/* out[r] = <row r of matrix which (0=A,1=B,2=C), z> for r < num_constraints, zero up to out_len (prover.rs:258-278). */
int zk_r1cs_matvec_dev(zk_ctx* ctx, const zk_r1cs* r1cs, int which, const void* z_dev, void* out_dev, size_t out_len);
/* out[i] = src[idx[i]], or zero where idx[i] == 0xFFFFFFFF (index re-mappings of prover.rs:335-353 and
 * constraint_systems.rs:183-216); idx_dev is a device array of n u32. */
int zk_fr_gather_dev(zk_ctx* ctx, const void* src_dev, const uint32_t* idx_dev, size_t n, void* out_dev);
/* Evaluations of row / col / val (/ row_col) of A*, B*, C* on one domain: device vectors. */
typedef struct { const void* row; const void* col; const void* val; const void* row_col; } zk_marlin_matrix_evals;
/* f on K (prover.rs:620-641): f[i] = v_H(alpha) v_H(beta) sum_M eta_M val_M[i] / ((beta - row_M[i]) (alpha - col_M[i])). */
int zk_marlin_round3_f_evals_dev(zk_ctx* ctx, const zk_marlin_matrix_evals on_k[3], size_t k_size, const zk_fr* alpha,
                                 const zk_fr* beta, const zk_fr eta[3], const zk_fr* vh_alpha_vh_beta, void* f_out_dev);
/* a and b on the domain B (prover.rs:653-698). */
int zk_marlin_round3_ab_evals_dev(zk_ctx* ctx, const zk_marlin_matrix_evals on_b[3], size_t b_size, const zk_fr* alpha,
                                  const zk_fr* beta, const zk_fr eta[3], const zk_fr* vh_alpha_vh_beta, void* a_out_dev,
                                  void* b_out_dev);

/* Marlin::prove (arkworks/marlin/src/lib.rs:152-319) as one call: the three AHP rounds, MarlinKZG10 commitments (hiding bound 1 on
 * w, z_a, z_b, g_1; degree bounds |H| - 2 on g_1 and |K| - 2 on g_2 through the shifted powers), the Fiat-Shamir transcript
 * (FiatShamirRng<Blake2s> over the bytes to_bytes! writes), the evaluations and open_combinations, on an index the caller built
 * once (Marlin::index is set-up: zk-mpc_amd/marlin.py::Index / IndexKeys do it with the calls above) and hands over as
 * device-resident tables:
 *   index_polys     a_row a_col a_val a_row_col b_... c_... (ahp/mod.rs:33-40): coefficient vectors
 *   on_k / on_b     evaluations of row / col / val (/ row_col) of A*, B*, C* on K and on the domain B of size >= 3|K| - 3
 *   r1cs / r1cs_t   the balanced, padded matrices (z_A, z_B) and their transposes with rows re-indexed into H (calculate_t)
 *   w_idx / x_idx   the index maps of prover.rs:343-353 (|H| u32 each; 0xFFFFFFFF = zero)
 *   ivk_bytes       IndexVerifierKey::write: index_info | index_comms (marlin/src/data_structures.rs:36-43), the transcript's seed
 * powers_g: powers_of_g[0 ..= max_degree] of the SRS; powers_gamma_g: powers_of_gamma_g[0 .. 2].  z_dev: the padded assignment
 * (instance first).  zk_rng: the prover's randomness (zk_rng_from_seed / zk_fsrng_new), drawn in the reference's order;
 * mask_on_device != 0 samples the 3|H| mask coefficients on the device under a key taken from zk_rng (same distribution, a
 * different stream: 0.4 s of scalar ChaCha at 2^20 otherwise).  The proof is written in CanonicalSerialize form
 * (marlin/src/data_structures.rs:99-110); cap must be >= zk_marlin_proof_max_size().  Errors: an unsatisfied system fails
 * the outer sum-check (ZK_ERR_STATE), as the reference's assertions do. */
typedef struct {
    size_t num_constraints, num_variables, num_non_zero, num_instance;
    const zk_r1cs* r1cs;
    const zk_r1cs* r1cs_t;
    zk_poly_ref index_polys[12];
    zk_marlin_matrix_evals on_k[3], on_b[3];
    const uint32_t* w_idx;
    const uint32_t* x_idx;
    const uint8_t* ivk_bytes;
    size_t ivk_len;
} zk_marlin_index;
size_t zk_marlin_proof_max_size(void);
int zk_marlin_prove(zk_ctx* ctx, const zk_marlin_index* index, const zk_bases* powers_g, const zk_bases* powers_gamma_g,
                    const void* z_dev, zk_rng* zk_rng, int mask_on_device, uint8_t* proof_out, size_t cap, size_t* proof_len);

/* ---- SHE ring arithmetic of the preprocessing phase (row a15) ----
 * Elements are ark_mnt4_753::Fq = Fp768 (12 x u64 little-endian, Montgomery R = 2^768), the `Fq` of src/she.rs:17.
 * An Encodedtext of degree N is N consecutive elements; a Ciphertext is c0 | c1 | c2 (3N elements,
 * src/she/ciphertext.rs:10-14); batches are contiguous.  N may be any value in 1..2^14: powers of two >= 4 use the
 * negacyclic NTT, everything else an O(N^2) kernel; results are identical to the reference's
 * DensePolynomial::mul + poly_remainder2 by X^N + 1. */
typedef struct { uint64_t l[12]; } zk_fq753;
/* Texts<Fq> add / sub / neg (src/she/texts.rs:43-127); ZK_OP_MUL is the coefficient-wise product. */
int zk_she_vec_op_dev(zk_ctx* ctx, int op, const void* a_dev, const void* b_dev, void* out_dev, size_t n);
/* Encodedtext * Fq, Encodedtext * BigUint (src/she/encodedtext.rs:93-113). */
int zk_she_vec_scale_dev(zk_ctx* ctx, const void* a_dev, const zk_fq753* k, void* out_dev, size_t n);
/* Encodedtext * Encodedtext in F_q[X]/(X^N + 1) (src/she/encodedtext.rs:115-134, src/she/polynomial.rs:152-168). */
int zk_she_negacyclic_mul_dev(zk_ctx* ctx, const void* a_dev, const void* b_dev, void* out_dev, size_t n, size_t batch);
/* Ciphertext * Ciphertext (src/she/ciphertext.rs:113-122): c0 = x0 y0, c1 = x0 y1 + x1 y0, c2 = -x1 y1. */
int zk_she_ciphertext_mul_dev(zk_ctx* ctx, const void* x_dev, const void* y_dev, void* out_dev, size_t n, size_t batch);
/* Ciphertext::encrypt_from (src/she/ciphertext.rs:46-72): e_dev = batch x N, r_dev = batch x (u | v | w), one public key
 * (a, b), p = the plaintext modulus as an Fq element; out = batch ciphertexts. */
int zk_she_encrypt_dev(zk_ctx* ctx, const void* e_dev, const void* pk_a_dev, const void* pk_b_dev, const void* r_dev,
                       const zk_fq753* p, void* out_dev, size_t n, size_t batch);
/* Ciphertext::decrypt (src/she/ciphertext.rs:74-79): out = c0 - s c1 - s s c2, batch x N. */
int zk_she_decrypt_dev(zk_ctx* ctx, const void* ct_dev, const void* sk_dev, void* out_dev, size_t n, size_t batch);
/* Plaintexts::encode (src/she/plaintext.rs:45-59): batch x N Fr slots -> batch x N Fq coefficients (N a power of two). */
int zk_she_encode_dev(zk_ctx* ctx, const void* plain_fr_dev, void* out_dev, size_t n, size_t batch);
/* Encodedtext::decode (src/she/encodedtext.rs:24-52): batch x N Fq coefficients -> batch x N Fr slots. */
int zk_she_decode_dev(zk_ctx* ctx, const void* enc_dev, void* out_fr_dev, size_t n, size_t batch);

/* ---- transport of the vector opens inside the library (row a13): RCCL over xGMI, one communicator per context ----
 * The leader calls zk_comm_unique_id and ships the 128 bytes to the other parties over the channel it already has (the
 * reference's TCP mesh); then every party calls zk_comm_init(ctx, id, party_id, n_parties).  RCCL is dlopen'ed on first
 * use: ZK_RCCL_LIB if set, else the copy ALREADY mapped in the process (a host with PyTorch has one), else librccl.so.1 by
 * name -- never a second copy beside a mapped one. */
int zk_comm_unique_id(uint8_t out[128]);
int zk_comm_init(zk_ctx* ctx, const uint8_t id[128], int rank, int n_parties);
int zk_comm_destroy(zk_ctx* ctx);             /* also done by zk_ctx_destroy */
/* Diagnostics: rank count and rank as the communicator reports them (ncclCommCount / ncclCommUserRank; 0 / -1 without one), the
 * context's device, RCCL's version code, the path of the RCCL copy that was bound.  Out pointers may be NULL. */
int zk_comm_info(zk_ctx* ctx, int* n_ranks, int* rank, int* device, int* rccl_version, char* lib_path, size_t lib_path_cap);
/* Exchange pattern of zk_open_sum_fr_dev: 0 = by party count (default), 1 = all-gather + sum, 2 = all-to-all of slices +
 * sum + all-gather.  A property of the communicator: every party must make the same call (it is not read from the
 * environment, where parties could disagree and wait for each other in different collectives). */
int zk_comm_set_open_pattern(zk_ctx* ctx, int pattern);
/* AdditiveFieldShare::batch_open on a device vector (mpc-algebra/src/share/additive.rs:124-131 over
 * MpcSerNet::broadcast, channel.rs:12-28): out[i] = sum over parties of v[i] mod r, on every party; out may alias v.
 * Three or more parties: all-to-all of slices, local sum, all-gather of the summed slices (2 x 32 n bytes in per GPU);
 * two: one all-gather and a local sum.  Asynchronous on the context stream. */
int zk_open_sum_fr_dev(zk_ctx* ctx, const void* v_dev, size_t n, void* out_dev);

/* ---- the collaborative prover (rows a6, a10-a13) as one call --------------------------------- */
/* The transport a collaborative prover needs from its host: the reference's MpcNet (mpc-net/src/lib.rs:60-64) on whatever mesh the
 * host already has.  all_gather_bytes = MpcNet::broadcast_bytes: every party contributes `len` bytes, `out_all` receives
 * n_parties * len bytes ordered by party id; returns 0 on success.  open_sum_fr_dev (optional) = AdditiveFieldShare::batch_open of
 * a device vector (out may alias v); NULL: the context's RCCL communicator (zk_comm_init / zk_open_sum_fr_dev).  A single party
 * (n_parties = 1) needs neither. */
typedef struct zk_net_vtable {
    void* user;
    int (*all_gather_bytes)(void* user, const uint8_t* mine, size_t len, uint8_t* out_all);
    int (*open_sum_fr_dev)(void* user, const void* v_dev, size_t n, void* out_dev);
} zk_net_vtable;
/* create_proof::<MpcPairingEngine, C> over additive shares (src/groth16.rs:68-183): the local half of the witness map, the four
 * MSMs over z started, FieldShare::batch_mul of the D-element product (share/field.rs:97-129; tx/ty/tz = this party's Beaver triple
 * shares as device vectors, or all NULL for DummyFieldTripleSource), the second half, the H job, calculate_coeff on shares, the three
 * GroupShare::scale calls (share/group.rs:72-111, dummy group triples) and Proof::reveal (arkworks/groth16/src/reveal.rs:7-10).
 * z_share_dev: this party's share of the full assignment (the instance part shared like the rest); r_share / s_share: shares of
 * the proof's randomness.  Every party returns the same 192 bytes -- those of create_proof on the summed inputs.  *bytes_sent
 * (optional) = payload bytes this party contributed to opens. */
int zk_groth16_prove_shared(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r1cs, const void* z_share_dev, const zk_fr* r_share,
                            const zk_fr* s_share, const void* tx_dev, const void* ty_dev, const void* tz_dev,
                            const zk_net_vtable* net, uint8_t proof[192], uint64_t* bytes_sent);
/* The same with E = MpcPairingEngine<_, SpdzPairingShare> (the `malicious` feature: SpdzFieldShare / SpdzGroupShare,
 * mpc-algebra/src/share/spdz.rs:50-265,278-489, MAC key alpha = 1 held by the leader): every argument is a pair of lanes
 * [0] = share, [1] = MAC share; everything linear runs on both lanes (2 x 5 MSMs: spdz.rs:482-488), every open is followed by the
 * exchange of [leader ? opened : 0] - mac, whose sum must vanish (spdz.rs:177-196) -- otherwise ZK_ERR_MAC.  tx/ty/tz_lanes: NULL
 * (or NULL lanes) for the dummy triple source. */
int zk_groth16_prove_shared_spdz(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r1cs, const void* const z_lanes_dev[2],
                                 const zk_fr r_lanes[2], const zk_fr s_lanes[2], const void* const tx_lanes_dev[2],
                                 const void* const ty_lanes_dev[2], const void* const tz_lanes_dev[2], const zk_net_vtable* net,
                                 uint8_t proof[192], uint64_t* bytes_sent);

/* The collaborative Marlin prover as one call: MpcMarlin::prove (src/marlin.rs:56; arkworks/marlin/src/lib.rs:152-319 with
 * F = MpcField<Fr>) over this party's ADDITIVE shares.  z_share_dev = the party's share of the padded assignment (instance part
 * included: shared as from_public, opened for the transcript), zk_rng = the party's own generator (its draws are its shares of the
 * prover's randomness, in Marlin::prove's order); tx / ty / tz = device vectors of Beaver triple shares for the ONE product of two
 * witness vectors (z_A z_B over the multiplication domain: the power of two >= 3|H|), or all NULL for DummyFieldTripleSource
 * (mpc-algebra/src/share/additive.rs:352-375).  `net` as in zk_groth16_prove_shared.  Opens per proof: the public input, the
 * masked Beaver operands (2 x |MUL| elements), one element for the outer sum-check's zero test, the round commitments of the
 * witness-dependent oracles (w, z_a, z_b, mask_poly | g_1, h_1), the evaluations z_b(beta), g_1(beta), and the opening witness
 * and random_v at beta.  Every party writes the same proof bytes (Proof::serialize). */
int zk_marlin_prove_shared(zk_ctx* ctx, const zk_marlin_index* index, const zk_bases* powers_g, const zk_bases* powers_gamma_g,
                           const void* z_share_dev, zk_rng* zk_rng, int mask_on_device, const void* tx, const void* ty, const void* tz,
                           const zk_net_vtable* net, uint8_t* proof_out, size_t cap, size_t* proof_len, uint64_t* bytes_sent);
/* ... over SPDZ shares (the `malicious` feature, mpc-algebra/src/share/spdz.rs; BASELINE config 5's prover): every *_lanes[2] is
 * {share, MAC share}.  Everything linear runs on both lanes; every open is MAC-checked (a failure returns ZK_ERR_MAC and no
 * proof).  The MAC lane of this party's fresh randomness is the share itself (from_add_shared, MAC key 1 held by the leader). */
int zk_marlin_prove_shared_spdz(zk_ctx* ctx, const zk_marlin_index* index, const zk_bases* powers_g, const zk_bases* powers_gamma_g,
                                const void* const z_lanes_dev[2], zk_rng* zk_rng, int mask_on_device, const void* const tx_lanes[2],
                                const void* const ty_lanes[2], const void* const tz_lanes[2], const zk_net_vtable* net,
                                uint8_t* proof_out, size_t cap, size_t* proof_len, uint64_t* bytes_sent);

/* ---- the trait surface for the COLLABORATIVE element types, on the caller's own slices (rows a6, a7, a8, a11) ----------------
 * Under E = MpcPairingEngine the unchanged create_proof (src/groth16.rs:68-183,240-306) calls the same four dispatch points with
 * Vec<MpcField<Fr, S>> and &[MpcG1Affine]: an element is the enum { Public(Fr), Shared(S) } (mpc-algebra/src/wire/field.rs:37-40;
 * 40 bytes with S = AdditiveFieldShare, 72 with SpdzFieldShare { sh, mac }), a base the enum MpcGroup around a GroupAffine.  The
 * entry points below read and write those elements IN PLACE; the binding describes the layout once, off a value:
 *   stride      size_of::<MpcField<Fr, S>>()
 *   off_tag     byte offset of the discriminant's low byte; tag_public / tag_shared: its two values
 *   off_public  byte offset of the Fr inside Public(Fr)
 *   off_share   ... of .val (additive) or .sh.val (SPDZ) inside Shared(S)
 *   off_mac     ... of .mac.val inside Shared(SpdzFieldShare), or SIZE_MAX: additive shares (one lane)
 * Semantics are the reference's operator impls (wire/field.rs:339-362,414-437,463-492): next to shared values a Public(x) acts as
 * the share "x on the leader (party 0 of zk_ctx_create), 0 elsewhere", in the MAC lane too (mac_share = 1 on the leader). */
typedef struct {
    size_t stride, off_tag, off_public, off_share, off_mac;
    uint8_t tag_public, tag_shared;
} zk_mpc_field_layout;
/* MpcG1Affine / MpcG2Affine { val: MpcGroup<G, S> }: where the GroupAffine of the Public variant sits, and the byte that tells the
 * variants apart (off_tag = SIZE_MAX: plain GroupAffine elements, no wrapper).  Every base must be Public, as the reference asserts
 * (wire/pairing.rs:716): ZK_ERR_ARG otherwise. */
typedef struct {
    zk_affine_layout point;
    size_t off_tag;
    uint8_t tag_public;
} zk_mpc_group_layout;
/* EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place(&mut Vec<MpcField>) (src/groth16.rs:278-303 over
 * arkworks/algebra/poly/src/domain/mod.rs:78,89,138,154): n elements are read, 2^log_n written (the binding resizes the Vec with
 * Public(0) first, as the reference does).  At least one Shared element: the transform of the party's lane vector(s), every output
 * Shared; all Public: the transform of the values, outputs Public.  inverse / coset as zk_fr_ntt_dev. */
int zk_mpc_fft_in_place(zk_ctx* ctx, void* vec_host, size_t n, const zk_mpc_field_layout* layout, uint32_t log_n, int inverse, int coset);
/* EvaluationDomain::divide_by_vanishing_poly_on_coset_in_place(&mut [MpcField]) (domain/mod.rs:183-190): 2^log_n elements times
 * the public constant 1 / (g^N - 1); every element keeps its variant. */
int zk_mpc_divide_by_vanishing_on_coset_in_place(zk_ctx* ctx, void* evals_host, const zk_mpc_field_layout* layout, uint32_t log_n);
/* MpcField::batch_product_in_place(selfs, others) (mpc-algebra/src/wire/field.rs:917-958).  Both slices Shared: FieldShare::batch_mul
 * (share/field.rs:97-129) -- the two masked operands are opened through `net` (open_sum_fr_dev, or the context's RCCL communicator
 * when that is NULL; SPDZ: each open followed by its MAC check, ZK_ERR_MAC on failure), selfs receives the product shares.
 * triple_host: NULL = DummyFieldTripleSource (what the reference's wire passes: wire/field.rs:941-947), or 3 (additive) / 6 (SPDZ)
 * host vectors of n zk_fr each: x, y, z of the share lane, then of the MAC lane.  Otherwise the element-wise `*a *= b`.  A slice
 * that mixes variants is refused (the reference asserts).  *bytes_sent (optional): payload this party contributed to opens. */
int zk_mpc_batch_product_in_place(zk_ctx* ctx, void* selfs_host, const void* others_host, size_t n, const zk_mpc_field_layout* layout,
                                  const zk_fr* const* triple_host, const zk_net_vtable* net, uint64_t* bytes_sent);
/* MpcG1Affine::multi_scalar_mul(bases, scalars) / the G2 form (mpc-algebra/src/wire/pairing.rs:714-777): min(len) terms.
 * Some scalar Shared: GroupShare::multi_scale_pub_group (share/additive.rs:517-520; SPDZ share/spdz.rs:482-488) -- out_lanes[0] =
 * the MSM over this party's share values, out_lanes[1] = the one over its MAC values (SPDZ; additive: a copy of lane 0), Public
 * scalars read as from_public (wire/field.rs:75-100); *scalars_public = 0.  Every scalar Public: out_lanes[0] = the plain MSM (the
 * wire wraps it with from_public), *scalars_public = 1.  The bases go through the table cache above, keyed by content.
 * (share/spdz.rs:484-485 feeds the SHARE values to both MSMs of a SpdzGroupShare; every state the reference can reach has mac = share
 * on each party -- from_add_shared with the stand-in key 1 -- so the two readings give the same lanes there, and either one is a
 * valid MAC of the sum anywhere: the MAC lane here is the MSM over the MAC values, as in zk_groth16_prove_shared_spdz.) */
int zk_mpc_msm_g1(zk_ctx* ctx, const void* bases_host, size_t n_bases, const zk_mpc_group_layout* base_layout, const void* scalars_host,
                  size_t n_scalars, const zk_mpc_field_layout* scalar_layout, zk_g1_projective out_lanes[2], int* scalars_public);
int zk_mpc_msm_g2(zk_ctx* ctx, const void* bases_host, size_t n_bases, const zk_mpc_group_layout* base_layout, const void* scalars_host,
                  size_t n_scalars, const zk_mpc_field_layout* scalar_layout, zk_g2_projective out_lanes[2], int* scalars_public);

/* ---- share algebra on device vectors (rows a11, a13) ------------------------------------- */
/* out[i] = sum_p gathered[p*n + i] mod r: the receive side of AdditiveFieldShare::batch_open
 * (mpc-algebra/src/share/additive.rs:124-131) after an all-gather of the parties' vectors. */
int zk_fr_sum_parties_dev(zk_ctx* ctx, const void* gathered_dev, size_t n_parties, size_t n, void* out_dev);
/* Local tail of FieldShare::batch_mul (share/field.rs:118-128):
 * out = tz - sx*ty - oy*tx (+ sx*oy on the leader); tx,ty,tz = this party's triple shares or
 * NULL for DummyFieldTripleSource (wire/field.rs:49-63: leader holds 1, the rest 0). */
int zk_beaver_combine_dev(zk_ctx* ctx, const void* sx_open_dev, const void* oy_open_dev,
                          const void* tx_dev, const void* ty_dev, const void* tz_dev,
                          void* out_dev, size_t n);

/* *is_zero = 1 iff every element of the device vector is 0: the MAC check of SpdzFieldShare::batch_open
 * (assert!(sum.is_zero()) per element, mpc-algebra/src/share/spdz.rs:188-195) as one reduction. */
int zk_fr_vec_is_zero_dev(zk_ctx* ctx, const void* v_dev, size_t n, int* is_zero);

/* ---- randomness (rows a11 / a13: share sampling; f.1: the Fiat-Shamir generator of Marlin) ------------------------- */
/* out[i] = F::rand(rng), i < n, on the device: the N - 1 uniform share vectors of AdditiveFieldShare::king_share
 * (mpc-algebra/src/share/additive.rs:98-107; ff/src/fields/arithmetic.rs:200-219).  Element i is ChaCha20(key, block i,
 * stream_id) -- 512 bits -- reduced mod r.  key32 = NULL draws the key from the operating system's CSPRNG (getrandom);
 * a caller-supplied 32-byte key makes the vector reproducible (tests).  Asynchronous on the context stream. */
int zk_fr_random_dev(zk_ctx* ctx, const uint8_t* key32, uint64_t stream_id, void* out_dev, size_t n);
/* Byte-exact host generators.  zk_rng is rand_chacha's ChaChaRng behind rand_core's BlockRng word stream.
 *   zk_fsrng_new / zk_fsrng_absorb   FiatShamirRng::<Blake2s>::{from_seed, absorb} (arkworks/marlin/src/rng.rs:44-67) on the
 *                                    bytes the reference's to_bytes! produces (the caller serialises)
 *   zk_rng_from_seed(seed, rounds)   ChaChaRng::from_seed (20) or rand 0.8.5's StdRng::from_seed (12: ark_std::test_rng,
 *                                    arkworks/std/src/rand_helper.rs:31-39)
 *   zk_rng_next_fr                   Fr::rand: 4 x next_u64, top 3 bits cleared, rejected unless < r; the words are the
 *                                    element's Montgomery form (ff/src/fields/arithmetic.rs:200-219)
 *   zk_rng_next_u128                 u128::rand (marlin/src/lib.rs:300: the opening challenge), low half in out[0] */
int zk_fsrng_new(const uint8_t* seed_bytes, size_t len, zk_rng** out);
int zk_fsrng_absorb(zk_rng* rng, const uint8_t* bytes, size_t len);
int zk_rng_from_seed(const uint8_t seed[32], int rounds, zk_rng** out);
int zk_rng_free(zk_rng* rng);
int zk_rng_next_u64(zk_rng* rng, uint64_t* out);
int zk_rng_next_u128(zk_rng* rng, uint64_t out[2]);
int zk_rng_next_fr(zk_rng* rng, zk_fr* out);
int zk_rng_fill_fr(zk_rng* rng, zk_fr* out, size_t n);   /* n draws of zk_rng_next_fr (DensePolynomial::rand) */
int zk_rng_fill_bytes(zk_rng* rng, uint8_t* out, size_t n);
/* Blake2s-256 (RFC 7693; the digest of FiatShamirRng) and the ChaCha block function (RFC 8439 2.3; words 12..15 of the
 * state are passed in: (counter lo, counter hi, stream lo, stream hi) for rand_chacha, (counter, nonce[3]) for the RFC). */
int zk_blake2s(const uint8_t* data, size_t len, uint8_t out[32]);
int zk_chacha_block(const uint8_t key[32], const uint32_t words12_15[4], int rounds, uint8_t out[64]);

/* ---- instrumentation -------------------------------------------------------------------- */
/* zk_set_profiling(ctx, 1): bracket every phase (witness map, MSM sort / accumulate / reduce) with
 * HIP events on the context stream; zk_last_timers returns the accumulated device time (ms) and
 * launch count per phase since the last call and resets them.  Names are NUL-terminated,
 * `name_stride` bytes apart; returns the number of entries written. */
int zk_set_profiling(zk_ctx* ctx, int on);
int zk_last_timers(zk_ctx* ctx, char* names, size_t name_stride, float* ms, int* counts, int max_entries);
/* Diagnostic (no reference counterpart): the issue rate of v_mad_u64_u32 on this device in lane multiply-adds per second, measured
 * now with `launches` timed launches of a pure multiply-add kernel (~4.4 ms each) on the context stream: best and median launch.
 * This is the integer-ALU roof the MSM kernels are priced against. */
int zk_diag_int_mad_peak(zk_ctx* ctx, int launches, double* best_mads_per_s, double* median_mads_per_s);
/* Diagnostic: Field::pow (ff/src/fields/mod.rs: square-and-multiply from the top bit) as ONE chain of dependent DEVICE products through
 * the field templates the kernels run -- lazy = 0: fully reduced products; lazy = 1: the lazy domain of the accumulate kernels (Fq) /
 * of the NTT butterflies (Fr), canonicalised once at the end.  Values in the reference's Montgomery form, exponent little-endian
 * u64 limbs.  This is how the reference's own long-chain known answers run on the GPU: Fq::multiplicative_generator().pow(T) ==
 * two_adic_root_of_unity() (arkworks/curves/bls12_377/src/fields/tests.rs:352-370) and the same relation of fr.rs's constants. */
int zk_diag_fq_pow_dev(zk_ctx* ctx, const zk_fq* base, const uint64_t exp[6], int lazy, zk_fq* out);
int zk_diag_fr_pow_dev(zk_ctx* ctx, const zk_fr* base, const uint64_t exp[4], int lazy, zk_fr* out);
/* Diagnostic: k * a in G1 through the curve's endomorphism (hostfield64.hpp: host64_scalar_mul_glv: k split at lambda = z^2 - 1, one
 * joint chain) -- what the host tail of a Groth16 proof runs for keys made by zk_groth16_setup.  Equal to zk_g1_mul for every point of
 * the prime-order subgroup; kept apart from it because ProjectiveCurve::mul (zk_g1_mul) is defined on the whole curve. */
int zk_diag_g1_mul_glv(const zk_g1_projective* a, const zk_fr* k, zk_g1_projective* out);

#ifdef __cplusplus
}
#endif
#endif /* ZKMPC_HIP_H */
