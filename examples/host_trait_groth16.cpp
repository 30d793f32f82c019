// host_trait_groth16.cpp -- the reference's create_proof, line for line, over the TRAIT-SHAPED entry points only.
//
// What an unchanged zk-mpc caller executes once the four arkworks dispatch points are overridden:
// src/groth16.rs:68-183 (create_proof) with R1CStoQAP::witness_map (:240-306) -- host Vecs in, host Vecs out, nothing resident that
// the caller knows of.  Every library call below is one of
//     zk_fr_fft_in_place                            EvaluationDomain::{ifft, coset_fft, coset_ifft}_in_place   (x7, :278-303)
//     zk_fr_batch_product_in_place                  Field::batch_product_in_place                               (x1, :285)
//     zk_fr_divide_by_vanishing_on_coset_in_place   EvaluationDomain::divide_by_vanishing_poly_on_coset_in_place (x1, :302)
//     zk_msm_g1 / zk_msm_g2 (or the _strided forms) AffineCurve::multi_scalar_mul                               (x5, :106,110,193)
//     zk_g1_mul / zk_g1_add / ... / zk_g1_serialize GroupProjective arithmetic of the O(1) tail                 (:112-176)
// and everything else is the caller's own scalar code (evaluate_constraint, `ab_i -= c_i`), written here in plain C++ with 64-bit
// limbs as the Rust code runs it: single-threaded, timed separately.  The proving key lives in HOST vectors (downloaded once from
// a set-up with fixed toxic waste and then freed on the device); the constraint system is the SURVEY 8(d) mul-chain.
//
// Output: one JSON line per proof {"proof": hex, "ms": {total, lib, fft, batch_product, divide, msm_h, msm_l, msm_a, msm_b1,
// msm_b2, tail, caller_matvec, caller_sub}} and a last line with the cache counters.  tests/test_gpu_trait_path.py compares the
// bytes with the oracle's known-trapdoor prediction; bench.py's `trait_path` leg reports the times.
//
//   host_trait_groth16 <log2 of the QAP domain> <proofs> [cache|nocache] [packed|strided] [verify|trust]
// verify (default): every cache hit is confirmed by comparing the caller's whole table with the cached one on the device, under the
// MSM; trust: zk_bases_cache_trust(ctx, 1) -- a fingerprint of 64 points decides (the caller vouches for its key).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "zkmpc_hip.h"

static zk_ctx* CTX = nullptr;
#define CK(expr)                                                                                        \
    do {                                                                                                \
        int rc_ = (expr);                                                                               \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, CTX ? zk_last_error(CTX) : "(no context: no GPU?)");  \
                        exit(1); }                                                                      \
    } while (0)

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- the caller's own field code (Fp256 in Montgomery form, ff/src/fields/arithmetic.rs:7-57 shape): modulus and -1/r mod 2^64
// are read off the library (r - 1 = the canonical form of 0 - 1), not typed in ----
typedef unsigned __int128 u128;
static uint64_t MOD[4], INV;
static void field_init() {
    zk_fr zero{}, one, m1;
    uint64_t c1[4] = {1, 0, 0, 0}, c[4];
    zk_fr_from_canonical(c1, &one);
    zk_fr_sub(&zero, &one, &m1);
    zk_fr_to_canonical(&m1, c);                       // r - 1
    u128 carry = 1;
    for (int i = 0; i < 4; i++) { carry += c[i]; MOD[i] = (uint64_t)carry; carry >>= 64; }
    uint64_t x = 1;                                   // Newton: x = r^-1 mod 2^64
    for (int i = 0; i < 6; i++) x *= 2 - MOD[0] * x;
    INV = (uint64_t)0 - x;
}
static inline bool geq_mod(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] > MOD[i]) return true; if (a[i] < MOD[i]) return false; }
    return true;
}
static inline void sub_mod_raw(uint64_t a[4]) {
    u128 b = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - MOD[i] - (uint64_t)b; a[i] = (uint64_t)d; b = (d >> 64) & 1; }
}
static inline zk_fr fr_add(const zk_fr& x, const zk_fr& y) {
    zk_fr r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)x.l[i] + y.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (geq_mod(r.l)) sub_mod_raw(r.l);               // (r < 2^253: no carry out of the top limb)
    return r;
}
static inline zk_fr fr_sub(const zk_fr& x, const zk_fr& y) {
    zk_fr r; u128 b = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)x.l[i] - y.l[i] - (uint64_t)b; r.l[i] = (uint64_t)d; b = (d >> 64) & 1; }
    if (b) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static inline zk_fr fr_mul(const zk_fr& x, const zk_fr& y) {      // CIOS
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)x.l[j] * y.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * INV;
        c = ((u128)m * MOD[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    zk_fr r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_mod(r.l)) sub_mod_raw(r.l);
    return r;
}
static zk_fr fr(uint64_t v) { uint64_t c[4] = {v, 0, 0, 0}; zk_fr o; zk_fr_from_canonical(c, &o); return o; }

struct Csr { std::vector<uint32_t> row_ptr, col; std::vector<zk_fr> coeff; };
// evaluate_constraint (src/groth16.rs:205-234): sum of coeff * assignment[index]
static zk_fr evaluate_constraint(const Csr& m, size_t row, const std::vector<zk_fr>& z, const zk_fr& one) {
    zk_fr s{};
    for (uint32_t k = m.row_ptr[row]; k < m.row_ptr[row + 1]; k++) {
        const zk_fr& c = m.coeff[k];
        s = fr_add(s, memcmp(&c, &one, sizeof one) == 0 ? z[m.col[k]] : fr_mul(z[m.col[k]], c));
    }
    return s;
}

// GroupAffine<P> as rustc lays it out for BLS12-377: {x, y, infinity: bool} padded to the alignment of u64
struct RustG1 { zk_fq x, y; bool infinity; };
struct RustG2 { zk_fq x[2], y[2]; bool infinity; };
template <class R, class A> static std::vector<R> to_rust(const std::vector<A>& v) {
    std::vector<R> o(v.size());
    for (size_t i = 0; i < v.size(); i++) {
        bool zero = true;
        const uint64_t* w = (const uint64_t*)&v[i];
        for (size_t k = 0; k < sizeof(A) / 8; k++) zero = zero && w[k] == 0;
        memset(&o[i], 0, sizeof(R));
        if (zero) { o[i].infinity = true; continue; }          // (coordinates of an infinity are whatever: zero here)
        memcpy(&o[i], &v[i], sizeof(A));
    }
    return o;
}

int main(int argc, char** argv) {
    const unsigned log_d = argc > 1 ? (unsigned)atoi(argv[1]) : 10;
    const int proofs = argc > 2 ? atoi(argv[2]) : 3;
    const bool cache = !(argc > 3 && std::string(argv[3]) == "nocache");
    const bool strided = argc > 4 && std::string(argv[4]) == "strided";
    const bool trust = argc > 5 && std::string(argv[5]) == "trust";
    if (log_d < 2 || log_d > 24 || proofs < 1) { fprintf(stderr, "usage: %s <log2 domain 2..24> <proofs> [cache|nocache] [packed|strided]\n", argv[0]); return 2; }
    const size_t D = (size_t)1 << log_d, n = D - 2;                       // n constraints + 2 instance variables fill the domain
    CK(zk_ctx_create(0, 0, 1, &CTX));
    field_init();
    if (!cache) CK(zk_bases_cache_config(CTX, 0, 0));
    if (trust) CK(zk_bases_cache_trust(CTX, 1));
    const zk_fr one = fr(1);

    // ---- set-up (not the path under test): constraint system, key with fixed toxic waste, assignment; everything ends up in HOST memory
    const size_t ni = 2, nw = n + 1, m = ni + nw;
    Csr A, B, Cm;
    auto idx = [&](size_t j) { return (uint32_t)(j <= n ? 2 + j : 1); };
    for (Csr* M : {&A, &B, &Cm}) { M->row_ptr.resize(n + 1); M->col.resize(n); M->coeff.assign(n, one); }
    for (size_t i = 0; i <= n; i++) A.row_ptr[i] = B.row_ptr[i] = Cm.row_ptr[i] = (uint32_t)i;
    for (size_t i = 0; i < n; i++) { A.col[i] = idx(i); B.col[i] = idx(i + 1); Cm.col[i] = idx(i + 2); }
    zk_r1cs_host rh{n, ni, nw, A.row_ptr.data(), A.col.data(), A.coeff.data(), B.row_ptr.data(), B.col.data(), B.coeff.data(),
                    Cm.row_ptr.data(), Cm.col.data(), Cm.coeff.data()};
    zk_r1cs* r1cs = nullptr;
    CK(zk_r1cs_upload(CTX, &rh, &r1cs));
    const zk_fr alpha = fr(2), beta = fr(3), gamma = fr(5), delta = fr(7), tau = fr(11), g1k = fr(1), g2k = fr(1);
    zk_pk* pk = nullptr;
    CK(zk_groth16_setup(CTX, r1cs, &alpha, &beta, &gamma, &delta, &tau, &g1k, &g2k, &pk));
    std::vector<zk_g1_affine> a_query(zk_pk_query_len(pk, 0)), b_g1_query(zk_pk_query_len(pk, 1)), h_query(zk_pk_query_len(pk, 3)), l_query(zk_pk_query_len(pk, 4));
    std::vector<zk_g2_affine> b_g2_query(zk_pk_query_len(pk, 2));
    CK(zk_pk_download_g1(CTX, pk, 0, 0, a_query.size(), a_query.data()));
    CK(zk_pk_download_g1(CTX, pk, 1, 0, b_g1_query.size(), b_g1_query.data()));
    CK(zk_pk_download_g2(CTX, pk, 2, 0, b_g2_query.size(), b_g2_query.data()));
    CK(zk_pk_download_g1(CTX, pk, 3, 0, h_query.size(), h_query.data()));
    CK(zk_pk_download_g1(CTX, pk, 4, 0, l_query.size(), l_query.data()));
    zk_g1_affine alpha_g1, beta_g1, delta_g1;
    zk_g2_affine beta_g2, delta_g2;
    CK(zk_pk_vk_g1(pk, 0, &alpha_g1)); CK(zk_pk_vk_g1(pk, 1, &beta_g1)); CK(zk_pk_vk_g1(pk, 2, &delta_g1));
    CK(zk_pk_vk_g2(pk, 0, &beta_g2)); CK(zk_pk_vk_g2(pk, 1, &delta_g2));
    CK(zk_pk_free(CTX, pk));
    CK(zk_r1cs_free(CTX, r1cs));
    if (a_query.size() != m || b_g1_query.size() != m || b_g2_query.size() != m || l_query.size() != nw) { fprintf(stderr, "unexpected key shape\n"); return 1; }
    std::vector<zk_fr> z(m);                                            // instance [1, w_{n+1}] then witness w_0 .. w_n
    {
        std::vector<zk_fr> w(n + 2);
        w[0] = fr(3); w[1] = fr(5);
        for (size_t i = 0; i < n; i++) w[i + 2] = fr_mul(w[i], w[i + 1]);
        z[0] = one; z[1] = w[n + 1];
        for (size_t j = 0; j <= n; j++) z[2 + j] = w[j];
    }
    // the Rust-layout copies of the key (what `pk.h_query: Vec<G1Affine>` is in the caller's memory)
    std::vector<RustG1> ra, rb1, rh_, rl;
    std::vector<RustG2> rb2;
    if (strided) { ra = to_rust<RustG1>(a_query); rb1 = to_rust<RustG1>(b_g1_query); rh_ = to_rust<RustG1>(h_query); rl = to_rust<RustG1>(l_query); rb2 = to_rust<RustG2>(b_g2_query); }
    const zk_affine_layout lay1{sizeof(RustG1), offsetof(RustG1, x), offsetof(RustG1, y), offsetof(RustG1, infinity)};
    const zk_affine_layout lay2{sizeof(RustG2), offsetof(RustG2, x), offsetof(RustG2, y), offsetof(RustG2, infinity)};
    auto msm1 = [&](const std::vector<zk_g1_affine>& q, const std::vector<RustG1>& rq, size_t skip, const zk_fr* s, size_t ns, zk_g1_projective* out) {
        if (strided) CK(zk_msm_g1_strided(CTX, rq.data() + skip, rq.size() - skip, &lay1, s, ns, out));
        else CK(zk_msm_g1(CTX, q.data() + skip, q.size() - skip, s, ns, out));
    };
    const zk_fr r = fr(13), s = fr(17);

    for (int it = 0; it < proofs; it++) {
        double t_fft = 0, t_bp = 0, t_div = 0, t_mat = 0, t_sub = 0, t_tail = 0, t_msm[5] = {0, 0, 0, 0, 0};
        std::vector<double> fft_each;
        const double t0 = now_ms();
        double t = t0;
        auto lap = [&](double& acc) { const double u = now_ms(); acc += u - t; t = u; };
        auto fft = [&](std::vector<zk_fr>& v, int inverse, int coset) {
            const double u = now_ms();
            CK(zk_fr_fft_in_place(CTX, v.data(), D, log_d, inverse, coset));
            fft_each.push_back(now_ms() - u);
        };
        // ---- R1CStoQAP::witness_map (src/groth16.rs:240-306) ----
        std::vector<zk_fr> a(D), b(D);                                    // vec![zero; domain_size]
        for (size_t i = 0; i < n; i++) { a[i] = evaluate_constraint(A, i, z, one); b[i] = evaluate_constraint(B, i, z, one); }
        for (size_t i = 0; i < ni; i++) a[n + i] = z[i];                   // a[start..end].clone_from_slice(&full_assignment[..num_inputs])
        lap(t_mat);
        fft(a, 1, 0);                                                     // domain.ifft_in_place(&mut a)
        fft(b, 1, 0);
        fft(a, 0, 1);                                                     // domain.coset_fft_in_place(&mut a)
        fft(b, 0, 1);
        lap(t_fft);
        std::vector<zk_fr> ab(a);                                         // let mut ab = a.clone()
        lap(t_sub);
        CK(zk_fr_batch_product_in_place(CTX, ab.data(), b.data(), D));
        lap(t_bp);
        std::vector<zk_fr> c(D);
        for (size_t i = 0; i < n; i++) c[i] = evaluate_constraint(Cm, i, z, one);
        lap(t_mat);
        fft(c, 1, 0);
        fft(c, 0, 1);
        lap(t_fft);
        for (size_t i = 0; i < D; i++) ab[i] = fr_sub(ab[i], c[i]);        // ab_i -= c_i
        lap(t_sub);
        CK(zk_fr_divide_by_vanishing_on_coset_in_place(CTX, ab.data(), log_d));
        lap(t_div);
        fft(ab, 1, 1);                                                    // domain.coset_ifft_in_place(&mut ab)
        lap(t_fft);
        const std::vector<zk_fr>& h = ab;
        // ---- create_proof (src/groth16.rs:104-176) ----
        zk_g1_projective h_acc, l_aux_acc, acc_a, acc_b1;
        zk_g2_projective acc_b2;
        msm1(h_query, rh_, 0, h.data(), h.size(), &h_acc);                 // multi_scalar_mul(&pk.h_query, &h)
        lap(t_msm[0]);
        msm1(l_query, rl, 0, z.data() + ni, nw, &l_aux_acc);               // (&pk.l_query, &prover.witness_assignment)
        lap(t_msm[1]);
        zk_g1_projective d1, r_s_delta_g1, r_g1, s_g1, g_a, s_g_a, g1_b, r_g1_b, g_c, tmp, neg;
        CK(zk_g1_from_affine(&delta_g1, &d1));
        CK(zk_g1_mul(&d1, &r, &r_g1));                                     // delta_g1 * r
        CK(zk_g1_mul(&r_g1, &s, &r_s_delta_g1));                           // ... * s
        lap(t_tail);
        const zk_fr* assignment = z.data() + 1;                            // instance[1..] ++ witness
        const size_t na = m - 1;
        // calculate_coeff(initial, query, vk_param, assignment) = initial + query[0] + msm(query[1..], assignment) + vk_param
        msm1(a_query, ra, 1, assignment, na, &acc_a);
        lap(t_msm[2]);
        CK(zk_g1_from_affine(&a_query[0], &tmp)); CK(zk_g1_add(&r_g1, &tmp, &g_a)); CK(zk_g1_add(&g_a, &acc_a, &g_a));
        CK(zk_g1_from_affine(&alpha_g1, &tmp)); CK(zk_g1_add(&g_a, &tmp, &g_a));
        CK(zk_g1_mul(&g_a, &s, &s_g_a));
        CK(zk_g1_mul(&d1, &s, &s_g1));
        lap(t_tail);
        msm1(b_g1_query, rb1, 1, assignment, na, &acc_b1);
        lap(t_msm[3]);
        CK(zk_g1_from_affine(&b_g1_query[0], &tmp)); CK(zk_g1_add(&s_g1, &tmp, &g1_b)); CK(zk_g1_add(&g1_b, &acc_b1, &g1_b));
        CK(zk_g1_from_affine(&beta_g1, &tmp)); CK(zk_g1_add(&g1_b, &tmp, &g1_b));
        zk_g2_projective d2, s_g2, g2_b, tmp2;
        CK(zk_g2_from_affine(&delta_g2, &d2));
        CK(zk_g2_mul(&d2, &s, &s_g2));
        lap(t_tail);
        if (strided) CK(zk_msm_g2_strided(CTX, rb2.data() + 1, rb2.size() - 1, &lay2, assignment, na, &acc_b2));
        else CK(zk_msm_g2(CTX, b_g2_query.data() + 1, b_g2_query.size() - 1, assignment, na, &acc_b2));
        lap(t_msm[4]);
        CK(zk_g2_from_affine(&b_g2_query[0], &tmp2)); CK(zk_g2_add(&s_g2, &tmp2, &g2_b)); CK(zk_g2_add(&g2_b, &acc_b2, &g2_b));
        CK(zk_g2_from_affine(&beta_g2, &tmp2)); CK(zk_g2_add(&g2_b, &tmp2, &g2_b));
        CK(zk_g1_mul(&g1_b, &r, &r_g1_b));
        CK(zk_g1_add(&s_g_a, &r_g1_b, &g_c));                              // g_c = s_g_a + r_g1_b - r_s_delta_g1 + l_aux_acc + h_acc
        CK(zk_g1_neg(&r_s_delta_g1, &neg)); CK(zk_g1_add(&g_c, &neg, &g_c));
        CK(zk_g1_add(&g_c, &l_aux_acc, &g_c)); CK(zk_g1_add(&g_c, &h_acc, &g_c));
        uint8_t proof[192];
        CK(zk_g1_serialize(&g_a, proof)); CK(zk_g2_serialize(&g2_b, proof + 48)); CK(zk_g1_serialize(&g_c, proof + 144));
        lap(t_tail);
        const double total = now_ms() - t0;
        const double lib = t_fft + t_bp + t_div + t_msm[0] + t_msm[1] + t_msm[2] + t_msm[3] + t_msm[4] + t_tail;
        printf("{\"proof\": \"");
        for (int i = 0; i < 192; i++) printf("%02x", proof[i]);
        printf("\", \"ms\": {\"total\": %.3f, \"lib\": %.3f, \"fft\": %.3f, \"batch_product\": %.3f, \"divide\": %.3f, \"msm_h\": %.3f, \"msm_l\": %.3f, "
               "\"msm_a\": %.3f, \"msm_b1\": %.3f, \"msm_b2\": %.3f, \"tail\": %.3f, \"caller_matvec\": %.3f, \"caller_sub_clone\": %.3f}, \"fft_calls_ms\": [",
               total, lib, t_fft, t_bp, t_div, t_msm[0], t_msm[1], t_msm[2], t_msm[3], t_msm[4], t_tail, t_mat, t_sub);
        for (size_t i = 0; i < fft_each.size(); i++) printf("%s%.3f", i ? ", " : "", fft_each[i]);
        printf("]}\n");
        fflush(stdout);
    }
    uint64_t st[10], st2[4], before[10];
    CK(zk_bases_cache_stats(CTX, before));                 // how many tables had their window multiples when the last proof ended
    CK(zk_bases_cache_sync(CTX));                          // ... and with every build that was still running or due finished
    CK(zk_bases_cache_stats(CTX, st));
    CK(zk_bases_cache_stats2(CTX, st2));
    printf("{\"cache\": {\"hits\": %llu, \"misses\": %llu, \"evictions\": %llu, \"replaced\": %llu, \"uncached\": %llu, \"entries\": %llu, "
           "\"with_window_multiples\": %llu, \"resident_bytes\": %llu, \"uploaded_bytes\": %llu, \"budget\": %llu, \"verified\": %llu, "
           "\"verified_bytes\": %llu, \"builds\": %llu, \"with_window_multiples_at_last_proof\": %llu}, \"log_d\": %u, \"constraints\": %zu, "
           "\"layout\": \"%s\", \"hits\": \"%s\"}\n",
           (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3], (unsigned long long)st[4],
           (unsigned long long)st[5], (unsigned long long)st[6], (unsigned long long)st[7], (unsigned long long)st[8], (unsigned long long)st[9],
           (unsigned long long)st2[0], (unsigned long long)st2[1], (unsigned long long)st2[2], (unsigned long long)before[6],
           log_d, n, strided ? "strided" : "packed", trust ? "trusted" : "verified");
    CK(zk_ctx_destroy(CTX));
    return 0;
}
