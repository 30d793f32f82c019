// marlin_prove.hip -- Marlin::prove as ONE entry point: the three AHP rounds, MarlinKZG10 commitments, the Fiat-Shamir
// transcript, the evaluations and open_combinations, sequenced on the host in C++ over the library's own kernels.
//
// Replaces (reference):
//   Marlin::prove                                  arkworks/marlin/src/lib.rs:152-319
//   AHPForR1CS::prover_{init,first,second,third}_round   arkworks/marlin/src/ahp/prover.rs:216-716
//   AHPForR1CS::{verifier_*_round, verifier_query_set, construct_linear_combinations}   ahp/verifier.rs:42-170, ahp/mod.rs:112-290
//   MarlinKZG10::{commit, open}, Marlin::open_combinations   poly-commit/src/marlin/marlin_pc/mod.rs:172-340, marlin/mod.rs:213-306
//   FiatShamirRng / to_bytes! encodings            marlin/src/rng.rs, ff/src/bytes.rs, marlin_pc/data_structures.rs:252-263
// The same sequence exists as zk-mpc_amd/marlin.py::prove (kept: it is what the collaborative provers build on, and the two are
// tested against each other and against the oracle's independent prover byte for byte).  The index (Marlin::index: matrix
// arithmetisation, index commitments) is a one-off set-up and stays with the caller, who hands over device-resident tables.
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "hostgroup.hpp"
#include "internal.hpp"
#include "sharednet.hpp"
#include "hostfield64.hpp"
#include <algorithm>
#include <chrono>
#include <future>
#include <array>
#include <map>
#include <string>
#include <vector>

using namespace zk;

namespace {

// ---- Fr on the host (internal Montgomery form of the device arithmetic) ----
struct HF {
    Fr v;
    static HF zero() { return HF{fp_zero<FrParams>()}; }
    static HF one() { return HF{fp_one<FrParams>()}; }
    static HF from_u64(uint64_t x) {
        Fr t = fp_zero<FrParams>();
        t.l[0] = (uint32_t)(x & MASK29); t.l[1] = (uint32_t)((x >> 29) & MASK29); t.l[2] = (uint32_t)(x >> 58);
        return HF{fp_canon_to_int<FrParams>(t)};
    }
    static HF from_abi(const zk_fr& a) { return HF{fp_ext_to_int<FrParams>(host_load_ext<FrParams>(a.l))}; }
    zk_fr abi() const { zk_fr o; host_store_ext<FrParams>(o.l, fp_int_to_ext<FrParams>(v)); return o; }
    HF operator+(const HF& b) const { return HF{fp_add<FrParams>(v, b.v)}; }
    HF operator-(const HF& b) const { return HF{fp_sub<FrParams>(v, b.v)}; }
    HF operator*(const HF& b) const { return HF{fp_mul<FrParams>(v, b.v)}; }
    HF neg() const { return HF{fp_neg<FrParams>(v)}; }
    HF inv() const { return HF{fp_inv<FrParams>(v)}; }
    bool is_zero() const { return fp_is_zero<FrParams>(v); }
    bool operator==(const HF& b) const { return fp_eq<FrParams>(v, b.v); }
    HF pow(uint64_t e) const {
        HF r = one();
        bool started = false;
        for (int b = 63; b >= 0; b--) {
            if (started) r = r * r;
            if ((e >> b) & 1) { r = started ? r * *this : *this; started = true; }
        }
        return r;
    }
    void bytes(std::vector<uint8_t>& out) const {          // Fp::write: into_repr(), little endian
        uint32_t w[8];
        fp_pack<FrParams>(w, fp_int_to_canon<FrParams>(v));
        for (int i = 0; i < 8; i++) for (int b = 0; b < 4; b++) out.push_back((uint8_t)(w[i] >> (8 * b)));
    }
};

struct Dom {
    size_t size; uint32_t log; HF gen;
    explicit Dom(size_t num_coeffs) {
        log = 0;
        while (((size_t)1 << log) < num_coeffs) log++;
        size = (size_t)1 << log;
        Fr w = fp_const<FrParams>(FrParams::TWO_ADIC_ROOT);
        for (uint32_t i = 0; i < (uint32_t)FR_TWO_ADICITY - log; i++) w = fp_sqr<FrParams>(w);
        gen = HF{w};
    }
    HF vanishing(const HF& t) const { return t.pow(size) - HF::one(); }
};

struct Poly { char* p = nullptr; size_t n = 0; };        // n coefficients on the device
struct Comm { Affine<G1Field> c, s; bool has_shift = false; };

struct Prover {
    zk_ctx* ctx;
    const zk_marlin_index* ix;
    const zk_bases *pg, *pgg;
    size_t max_degree;
    zk_rng* rng;
    int lane = 0;                      // 0: the share lane (or the plain prover); 1: the MAC lane of a SPDZ prover -- own scratch names
    int rc = ZK_OK;
    std::map<std::string, Poly> polys;
    std::map<std::string, std::pair<std::vector<HF>, std::vector<HF>>> rands;   // label -> (blind, shifted blind)
    std::map<std::string, Comm> comms;
    std::map<std::string, size_t> bounds;

    char* dev(const std::string& name, size_t elems) {
        void* p = nullptr;
        if (rc == ZK_OK) rc = zk_scratch(ctx, ((lane ? "mp1." : "mp.") + name).c_str(), std::max<size_t>(elems, 1) * 32, &p);
        return (char*)p;
    }
    void ck(int r) { if (rc == ZK_OK) rc = r; }
    void d2d(void* dst, const void* src, size_t elems) { if (elems) ck(zk_memcpy_d2d(ctx, dst, src, elems * 32)); }
    void zero(void* dst, size_t elems) { if (elems) ck(zk_dev_zero(ctx, dst, elems * 32)); }
    void op(int o, const void* a, const void* b, void* out, size_t n) { if (n) ck(zk_fr_vec_op_dev(ctx, o, a, b, out, n)); }
    void scale(const void* a, const HF& k, void* out, size_t n) { zk_fr kk = k.abi(); if (n) ck(zk_fr_vec_scale_dev(ctx, a, &kk, out, n)); }
    void ntt(void* buf, const Dom& d, int inverse) { ck(zk_fr_ntt_dev(ctx, buf, d.log, inverse, 0)); }
    // up to four transforms of one size and kind as one launch per pass (ntt.hip::zk_ntt_launch_batch)
    void ntt_batch(std::initializer_list<void*> bufs, const Dom& d, int inverse) {
        std::vector<void*> v(bufs);
        if (rc == ZK_OK) ck(zk_ntt_launch_batch(ctx, v.data(), (int)v.size(), d.log, inverse, 0));
    }
    char* padded(const Dom& d, const Poly& p, const std::string& name) {  // the zero-padded copy evaluate_over_domain transforms
        char* out = dev(name, d.size);
        if (rc != ZK_OK) return out;
        if (p.n < d.size) zero(out + 32 * p.n, d.size - p.n);
        d2d(out, p.p, std::min(p.n, d.size));
        return out;
    }
    char* fft(const Dom& d, const Poly& p, const std::string& name) {     // evaluate_over_domain: zero-pad, forward transform
        char* out = padded(d, p, name);
        ntt(out, d, 0);
        return out;
    }
    bool is_zero(const void* v, size_t n) { int z = 0; ck(zk_fr_vec_is_zero_dev(ctx, v, n, &z)); return z != 0; }
    HF eval(const Poly& p, const HF& x) { zk_fr xx = x.abi(), o; ck(zk_poly_evaluate_dev(ctx, p.p, p.n, &xx, &o)); return HF::from_abi(o); }
    HF next_fr(zk_rng* r) { zk_fr o; ck(zk_rng_next_fr(r, &o)); return HF::from_abi(o); }
    // p + r (X^n - 1) for deg p < n: one more coefficient
    Poly blind(const char* poly, size_t n, const char* r_dev, const std::string& name) {
        char* out = dev(name, n + 1);
        d2d(out, poly, n);
        d2d(out + 32 * n, r_dev, 1);
        op(ZK_OP_SUB, out, r_dev, out, 1);
        return Poly{out, n + 1};
    }
};

void g1_tobytes(const Affine<G1Field>& a, std::vector<uint8_t>& out) {     // GroupAffine::write: x | y | infinity; zero() = (0, 1, true)
    uint8_t b[48];
    if (aff_is_inf<G1Field>(a)) {
        out.insert(out.end(), 48, 0);
        out.push_back(1); out.insert(out.end(), 47, 0);
        out.push_back(1);
        return;
    }
    fq_canonical_bytes(a.x, b); out.insert(out.end(), b, b + 48);
    fq_canonical_bytes(a.y, b); out.insert(out.end(), b, b + 48);
    out.push_back(0);
}
void comm_tobytes(const Comm& c, std::vector<uint8_t>& out) {              // marlin_pc::Commitment::write
    g1_tobytes(c.c, out);
    out.push_back(c.has_shift ? 1 : 0);
    g1_tobytes(c.has_shift ? c.s : aff_inf<G1Field>(), out);
}
Affine<G1Field> proj_to_aff(const zk_g1_projective& p) { return xyzz_to_affine<G1Field>(host_proj_from_abi<G1Field>((const uint64_t*)&p)); }
// ... of several points with ONE field inversion (the commitments of a round: 2 - 6 points, an inversion is ~400 products) in
// the 64-bit host field
std::vector<Affine<G1Field>> batch_to_aff(const std::vector<zk_g1_projective>& pts) {
    using H = Fq64Field;
    const size_t n = pts.size();
    std::vector<XYZZ<H>> x(n);
    std::vector<typename H::T> pre(n);
    typename H::T acc = H::one();
    for (size_t i = 0; i < n; i++) {
        x[i] = host64_proj_from_abi<H>((const uint64_t*)&pts[i]);
        if (xyzz_is_inf<H>(x[i])) continue;
        pre[i] = acc;
        acc = H::mul(acc, x[i].zzz);
    }
    typename H::T inv = H::inv(acc);
    std::vector<Affine<G1Field>> out(n);
    for (size_t i = n; i-- > 0;) {
        if (xyzz_is_inf<H>(x[i])) { out[i] = aff_inf<G1Field>(); continue; }
        const typename H::T zi3 = H::mul(inv, pre[i]);
        inv = H::mul(inv, x[i].zzz);
        const typename H::T zi = H::mul(zi3, x[i].zz), zi2 = H::sqr(zi);
        out[i] = aff_from_host64<G1Field>(Affine<H>{H::mul(x[i].x, zi2), H::mul(x[i].y, zi3)});
    }
    return out;
}

HF host_eval(const std::vector<HF>& c, const HF& x) {
    HF acc = HF::zero();
    for (size_t i = c.size(); i-- > 0;) acc = acc * x + c[i];
    return acc;
}
std::vector<HF> host_div_linear(const std::vector<HF>& c, const HF& z) {   // quotient of p / (X - z)
    std::vector<HF> q(c.size() > 1 ? c.size() - 1 : 0, HF::zero());
    HF acc = HF::zero();
    for (size_t i = c.size(); i-- > 1;) { acc = c[i] + acc * z; q[i - 1] = acc; }
    return q;
}
void acc_scaled(std::vector<HF>& dst, const std::vector<HF>& src, const HF& k) {
    if (dst.size() < src.size()) dst.resize(src.size(), HF::zero());
    for (size_t i = 0; i < src.size(); i++) dst[i] = dst[i] + src[i] * k;
}

// sum_i c_i G_i for the three powers_of_gamma_g a hiding bound of 1 uses: on the host (a three-term MSM through the device
// pipeline costs a full sort / accumulate / reduce round trip, ~0.5 ms; this is three scalar multiplications in 64-bit limbs)
zk_g1_projective small_msm(const zk_g1_projective* pts, const std::vector<HF>& c) {
    zk_g1_projective acc{};
    bool first = true;
    for (size_t i = 0; i < c.size(); i++) {
        zk_fr k = c[i].abi();
        zk_g1_projective t, u;
        zk_g1_mul(&pts[i], &k, &t);
        if (first) { acc = t; first = false; } else { zk_g1_add(&acc, &t, &u); acc = u; }
    }
    return acc;
}
// ... with the scalar multiplications side by side on the context's helper threads (0.2 ms each: three in a row were longer than
// the device batch of a small proof's round they are meant to hide under)
zk_g1_projective small_msm_par(zk_ctx* ctx, const zk_g1_projective* pts, const std::vector<HF>& c) {
    if (c.size() < 2) return small_msm(pts, c);
    std::vector<zk_g1_projective> t(c.size());
    {
        std::vector<ZkTask<void>> tasks;
        for (size_t i = 1; i < c.size(); i++)
            tasks.push_back(zk_async(ctx, [&t, pts, &c, i] { zk_fr k = c[i].abi(); zk_g1_mul(&pts[i], &k, &t[i]); }));
        zk_fr k0 = c[0].abi();
        zk_g1_mul(&pts[0], &k0, &t[0]);
    }
    zk_g1_projective acc = t[0], u;
    for (size_t i = 1; i < c.size(); i++) { zk_g1_add(&acc, &t[i], &u); acc = u; }
    return acc;
}

const char* const INDEX_LABELS[12] = {"a_row", "a_col", "a_val", "a_row_col", "b_row", "b_col", "b_val", "b_row_col",
                                      "c_row", "c_col", "c_val", "c_row_col"};
struct Term { HF c; const char* label; };      // label = nullptr: the constant term (LCTerm::One)

}  // namespace

extern "C" size_t zk_marlin_proof_max_size(void) { return 8 + 3 * 8 + 9 * 49 + 2 * 48 + 8 + 7 * 32 + 8 + 3 + 8 + 2 * (49 + 32) + 1; }

namespace {

bool fr_words_valid_abi(const uint64_t l[4]) {       // < r (is_valid, ff/src/fields/macros.rs:255-260): what arrives from a peer
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

// Opens of O(1) values on the caller's transport, batched into ONE exchange: scalars and G1 points of this party's share lane
// (AdditiveFieldShare / AdditiveGroupShare::open: the sum over parties).  LANES = 2 (SPDZ): frs[1] / g1s[1] are the MAC shares;
// a second exchange publishes [leader ? opened : 0] - mac for every item and every sum must vanish (SpdzFieldShare::batch_open,
// SpdzGroupShare::open: share/spdz.rs:177-196,312-336, key alpha = 1 held by the leader) -- otherwise ZK_ERR_MAC.
template <int LANES>
int open_small(ZkSharedNet& nt, const std::vector<HF> frs[2], const std::vector<zk_g1_projective> g1s[2], std::vector<HF>& out_fr,
               std::vector<zk_g1_projective>& out_g1) {
    using H1 = Fq64Field;
    using X1 = XYZZ<H1>;
    zk_ctx* ctx = nt.ctx;
    const size_t nf = frs[0].size(), ng = g1s[0].size(), words = 4 * nf + 18 * ng;
    out_fr.assign(nf, HF::zero());
    out_g1.resize(ng);
    if (!words) return ZK_OK;
    std::vector<uint64_t> msg(words);
    std::vector<uint8_t> all;
    std::vector<X1> og(ng);
    auto exchange = [&](std::vector<HF>& f, std::vector<X1>& g) -> int {
        ZK_TRY(nt.gather((const uint8_t*)msg.data(), words * 8, all));
        f.assign(nf, HF::zero());
        g.assign(ng, xyzz_inf<H1>());
        std::vector<uint64_t> w(words);
        for (int p = 0; p < nt.parties(); p++) {
            memcpy(w.data(), all.data() + (size_t)p * words * 8, words * 8);
            for (size_t i = 0; i < nf; i++) {
                if (!fr_words_valid_abi(&w[4 * i])) ZK_FAIL(ctx, ZK_ERR_STATE, "collaborative prover: a party sent a non-canonical field element");
                zk_fr a;
                memcpy(a.l, &w[4 * i], 32);
                f[i] = f[i] + HF::from_abi(a);
            }
            bool pts_ok = true;                      // a peer's points: canonical, on the curve, (malicious prover) in the subgroup
            const bool peer = p != ctx->party_id;
            for (size_t i = 0; i < ng; i++)
                g[i] = xyzz_add<H1>(g[i], peer ? host64_peer_point<H1>(&w[4 * nf + 18 * i], LANES == 2, pts_ok) : host64_proj_from_abi<H1>(&w[4 * nf + 18 * i]));
            if (!pts_ok) ZK_FAIL(ctx, ZK_ERR_STATE, "collaborative prover: a party sent a point that is not a valid group element");
        }
        return ZK_OK;
    };
    for (size_t i = 0; i < nf; i++) { zk_fr a = frs[0][i].abi(); memcpy(&msg[4 * i], a.l, 32); }
    for (size_t i = 0; i < ng; i++) memcpy(&msg[4 * nf + 18 * i], &g1s[0][i], 144);
    ZK_TRY(exchange(out_fr, og));
    if (LANES == 2) {
        const bool leader = nt.leader();
        for (size_t i = 0; i < nf; i++) { zk_fr a = ((leader ? out_fr[i] : HF::zero()) - frs[1][i]).abi(); memcpy(&msg[4 * i], a.l, 32); }
        for (size_t i = 0; i < ng; i++) {
            const X1 d = xyzz_add<H1>(leader ? og[i] : xyzz_inf<H1>(), xyzz_neg<H1>(host64_proj_from_abi<H1>((const uint64_t*)&g1s[1][i])));
            host64_write_projective<H1>(xyzz_to_affine<H1>(d), &msg[4 * nf + 18 * i]);
        }
        std::vector<HF> cf;
        std::vector<X1> cg;
        ZK_TRY(exchange(cf, cg));
        bool ok = true;
        for (auto& v : cf) ok = ok && v.is_zero();
        for (auto& v : cg) ok = ok && xyzz_is_inf<H1>(v);
        if (!ok) ZK_FAIL(ctx, ZK_ERR_MAC, "SPDZ MAC check failed on an opened commitment / evaluation / witness");
    }
    for (size_t i = 0; i < ng; i++) host64_write_projective<H1>(xyzz_to_affine<H1>(og[i]), (uint64_t*)&out_g1[i]);
    return ZK_OK;
}

bool label_shared(const char* l) {           // the witness-dependent oracles (mpc.py: Party.SHARED_POLYS); t, g_2, h_2 and the index are public
    return !strcmp(l, "w") || !strcmp(l, "z_a") || !strcmp(l, "z_b") || !strcmp(l, "mask_poly") || !strcmp(l, "g_1") || !strcmp(l, "h_1");
}

// Marlin::prove, plain (shared = false, LANES = 1: zk_marlin_prove) or over this party's shares (zk_marlin_prove_shared[_spdz]):
// MpcMarlin::prove, src/marlin.rs:56 / arkworks/marlin/src/lib.rs:152-319 with F = MpcField.  Every step of the rounds is linear
// in the witness except z_A * z_B in round 2 (FieldShare::batch_mul over the 4|H| multiplication domain) and the zero test of the
// outer sum-check (an open); commitments / evaluations / opening witnesses of witness-dependent oracles are computed on the
// shares and opened (`publicize()`, lib.rs:171-228,296); public oracles enter a shared combination on the leader only (shift()).
// LANES = 2: everything linear runs on the share lane and on the MAC lane, every open is MAC-checked; the MAC lane of this
// party's fresh randomness is the share itself (from_add_shared with key 1).
template <int LANES>
int marlin_impl(zk_ctx* ctx, const zk_marlin_index* ix, const zk_bases* powers_g, const zk_bases* powers_gamma_g, const void* const z_lanes[2],
                zk_rng* zk_rng_, int mask_on_device, bool shared, const void* const tx[2], const void* const ty[2], const void* const tz[2],
                const zk_net_vtable* net, uint8_t* proof_out, size_t cap, size_t* proof_len, uint64_t* bytes_sent) {
    if (powers_g->group != 1 || powers_gamma_g->group != 1 || powers_gamma_g->n < 3) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_marlin_prove: SRS tables");
    if (cap < zk_marlin_proof_max_size()) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_marlin_prove: output buffer smaller than zk_marlin_proof_max_size()");
    if (ix->num_constraints != ix->num_variables) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_marlin_prove: NonSquareMatrix");
    if (ix->num_instance == 0 || (ix->num_instance & (ix->num_instance - 1))) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_marlin_prove: InvalidPublicInputLength");
    // with zk_set_profiling(ctx, 1): host wall-clock laps of the phases land in the context's timers as "marlin.<phase>" (a lap
    // includes whatever device work the host waited for)
    struct Laps {
        zk_ctx* ctx;
        bool on;
        std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
        explicit Laps(zk_ctx* c) : ctx(c), on(c->profiling) {}
        void lap(const char* what) {
            if (!on) return;
            const auto now = std::chrono::steady_clock::now();
            auto& tm = ctx->timers[std::string("marlin.") + what];
            tm.ms += (float)std::chrono::duration<double, std::milli>(now - t).count();
            tm.count += 1;
            t = now;
        }
    } laps(ctx);
    ZkSharedNet nt{ctx, net};
    // divisibility / zero-sum checks whose verdict nothing waits for: the test is enqueued, the round's commitments go out behind
    // it, and the verdict is read once the batch has synchronised the streams (a wait here left the device idle for the whole of
    // the host's preparation of the batch: 0.1 - 0.15 ms per round of a small proof)
    struct Pending { const uint32_t* verdict; const char* msg; };
    std::vector<Pending> pending;
    auto check_later = [&](const void* v, size_t n, const char* msg) -> int {
        const uint32_t* h = nullptr;
        ZK_TRY(zk_fr_vec_is_zero_launch(ctx, v, n, (int)pending.size(), &h));
        pending.push_back({h, msg});
        return ZK_OK;
    };
    auto settle = [&]() -> int {
        const char* failed = nullptr;
        for (auto& c : pending)
            if (!failed && *c.verdict != 0) failed = c.msg;
        pending.clear();
        if (failed) ZK_FAIL(ctx, ZK_ERR_STATE, failed);
        return ZK_OK;
    };
    const bool leader = nt.leader();
    Prover PL[2] = {Prover{ctx, ix, powers_g, powers_gamma_g, powers_g->n - 1, zk_rng_}, Prover{ctx, ix, powers_g, powers_gamma_g, powers_g->n - 1, zk_rng_}};
    PL[1].lane = 1;
    Prover& P = PL[0];                            // lane 0 also carries everything public: blinds, commitments, bounds, public oracles
    auto lanes_rc = [&]() { for (int l = 0; l < LANES; l++) if (PL[l].rc != ZK_OK) return PL[l].rc; return (int)ZK_OK; };
    const Dom H(ix->num_constraints), K(ix->num_non_zero), X(ix->num_instance), B(3 * Dom(ix->num_non_zero).size - 3);
    const size_t n = H.size, ni = ix->num_instance;
    {   // AHPForR1CS::max_degree (ahp/mod.rs:75-97)
        const size_t need = std::max(std::max(2 * n - 1, 3 * n - 1), std::max(n, 3 * K.size - 3));
        if (P.max_degree < need) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_marlin_prove: IndexTooLarge for this SRS");
    }
    if (B.size < 4 * K.size - 3) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_marlin_prove: |K| < 4 is not supported by this entry point");
    P.bounds["g_1"] = n - 2;
    P.bounds["g_2"] = K.size - 2;
    const char* zb[2] = {(const char*)z_lanes[0], LANES == 2 ? (const char*)z_lanes[1] : nullptr};
    for (int l = 0; l < LANES; l++)
        for (int i = 0; i < 12; i++) PL[l].polys[INDEX_LABELS[i]] = Poly{(char*)ix->index_polys[i].ptr, ix->index_polys[i].n};
    for (int i = 0; i < 12; i++) P.rands[INDEX_LABELS[i]] = {};

    zk_g1_projective gamma_pts[3];
    {
        zk_g1_affine a[3];
        ZK_TRY(zk_bases_download_g1(ctx, powers_gamma_g, 0, 3, a));
        for (int i = 0; i < 3; i++) zk_g1_from_affine(&a[i], &gamma_pts[i]);
    }

    // ---- transcript seed: PROTOCOL_NAME | index_vk | public_input (lib.rs:161-164).  Over shares the instance part of the
    // assignment is shared like the rest (from_public: the leader holds it) and opened here ----
    std::vector<HF> pub(ni - 1);
    {
        std::vector<zk_fr> tmp(ni);
        if (shared && ni == 1) {
            // no public input beside the constant 1: nothing to open
        } else if (shared) {
            char* po = P.dev("pub_open", ni); char* pd = P.dev("pub_dx", ni);
            ZK_TRY(P.rc);
            if (LANES == 2) ZK_TRY(zk_shared_spdz_open_vec(nt, zb[0], zb[1], ni, po, pd));
            else ZK_TRY(nt.open_vec(zb[0], ni, po));
            ZK_TRY(zk_memcpy_d2h(ctx, tmp.data(), po, ni * 32));
        } else {
            ZK_TRY(zk_memcpy_d2h(ctx, tmp.data(), zb[0], ni * 32));
        }
        for (size_t i = 1; i < ni; i++) pub[i - 1] = HF::from_abi(tmp[i]);
    }
    std::vector<uint8_t> seed;
    const char* name = "MARLIN-2019";
    seed.insert(seed.end(), name, name + 11);
    seed.insert(seed.end(), ix->ivk_bytes, ix->ivk_bytes + ix->ivk_len);
    for (auto& v : pub) v.bytes(seed);
    zk_rng* fs = nullptr;
    ZK_TRY(zk_fsrng_new(seed.data(), seed.size(), &fs));
    struct FsGuard { zk_rng* r; ~FsGuard() { zk_rng_free(r); } } guard{fs};
    auto sample_outside = [&](const Dom& d) { HF t = P.next_fr(fs); while (d.vanishing(t).is_zero()) t = P.next_fr(fs); return t; };

    // Oracles of a round that exist before the round's last polynomial does -- the mask polynomial (round 1), t (round 2), g_2
    // (round 3); none of them hiding, so no draw of the prover's rng depends on where their commitment is computed -- have their MSMs
    // started as soon as their coefficients are on the device (msm_batch.hip: zk_msm_early_begin); the context stream goes on with
    // the round's polynomial arithmetic, commit_round collects them.  ZK_MARLIN_EARLY=0 keeps every job in the round's batch (A/B).
    struct Early {
        zk_ctx* ctx;
        std::map<std::string, ZkEarlyMsm*> jobs;
        ~Early() { for (auto& kv : jobs) if (kv.second) (void)zk_msm_early_finish(ctx, kv.second, nullptr); }     // an error path: let them drain
    } early{ctx, {}};
    static const bool early_on = !(getenv("ZK_MARLIN_EARLY") && atoi(getenv("ZK_MARLIN_EARLY")) == 0);
    auto start_early = [&](const char* l) -> int {
        // from |H| = 2^18 up only: measured on one box, alternating (profiles/r6_marlin_early_ab.jsonl) -- 2^20 65.0 / 64.7 -> 63.9 / 64.0 ms,
        // 2^18 21.8 -> 21.6; below that a round is a chain of latencies and the early job, which runs alone instead of in the round's
        // group launches (its own one-block sort, accumulate launch, reduce chain and host half), makes the proof LONGER:
        // 2^10 3.5 -> 4.2 ms, 2^12 4.3 -> 5.4, 2^14 6.1 -> 7.0, 2^16 9.5 -> 10.2
        if (!early_on || H.size < ((size_t)1 << 18)) return ZK_OK;
        const Poly& p = P.polys[l];
        const bool bounded = P.bounds.count(l) != 0;
        if (!p.n) return ZK_OK;
        if (bounded && p.n - 1 > P.bounds[l]) return ZK_OK;                      // (commit_round reports it)
        const size_t offs[2] = {0, bounded ? P.max_degree - P.bounds[l] : 0};
        ZkEarlyMsm* em = nullptr;
        ZK_TRY(zk_msm_early_begin(ctx, bounded ? 2 : 1, P.pg, offs, p.p, p.n, &em));
        early.jobs[l] = em;
        return ZK_OK;
    };

    // MarlinKZG10::commit for one round (marlin_pc/mod.rs:172-243): blinding polynomials in the reference's rng order, all MSMs
    // of the round (every lane) as one pipelined batch; shared oracles' commitments opened; then the round's bytes into the transcript
    auto commit_round = [&](std::initializer_list<const char*> labels) -> int {
        std::vector<const zk_bases*> jb; std::vector<size_t> joff, jlen; std::vector<const void*> jsc;
        struct Slot { std::string label; int which, lane; };
        std::vector<Slot> slot;                                                  // which: 0 = comm / 1 = shifted
        std::map<std::string, zk_g1_projective> acc[2][2];                       // [lane][which]
        // the blinding terms (three host scalar multiplications each, ~1.2 ms) run on host threads under the device batch
        std::vector<std::pair<std::pair<int, std::string>, ZkTask<zk_g1_projective>>> blinds;
        auto blind_async = [&](int which, const char* l, const std::vector<HF>& c) {
            blinds.push_back({{which, l}, zk_async(ctx, [ctx, &gamma_pts, c] { return small_msm_par(ctx, gamma_pts, c); })});
        };
        for (const char* l : labels) {
            const bool hiding = !strcmp(l, "w") || !strcmp(l, "z_a") || !strcmp(l, "z_b") || !strcmp(l, "g_1");
            const bool bounded = P.bounds.count(l) != 0;
            std::vector<HF> blind, sblind;
            if (hiding) for (int i = 0; i < 3; i++) blind.push_back(P.next_fr(P.rng));
            if (hiding && bounded) for (int i = 0; i < 3; i++) sblind.push_back(P.next_fr(P.rng));
            P.rands[l] = {blind, sblind};
            for (int lane = 0; lane < LANES; lane++) {
                if (lane == 1 && !label_shared(l)) continue;                     // public oracles are the same on every lane: committed once
                if (lane == 0 && early.jobs.count(l)) continue;                  // started early: collected below
                const Poly& p = PL[lane].polys[l];
                jb.push_back(P.pg); joff.push_back(0); jsc.push_back(p.p); jlen.push_back(p.n); slot.push_back({l, 0, lane});
                if (bounded) {
                    if (p.n - 1 > P.bounds[l]) { ctx->last_error = std::string("zk_marlin_prove: ") + l + " exceeds its degree bound"; return ZK_ERR_STATE; }
                    jb.push_back(P.pg); joff.push_back(P.max_degree - P.bounds[l]); jsc.push_back(p.p); jlen.push_back(p.n); slot.push_back({l, 1, lane});
                }
            }
            if (hiding) blind_async(0, l, blind);
            if (hiding && bounded) blind_async(1, l, sblind);
        }
        ZK_TRY(lanes_rc());
        std::vector<zk_g1_projective> outs(jb.size());
        std::vector<void*> outp(jb.size());
        for (size_t i = 0; i < jb.size(); i++) outp[i] = &outs[i];
        laps.lap("commit.prep");
        int brc = zk_msm_batch_dev(ctx, jb.size(), jb.data(), joff.data(), jsc.data(), jlen.data(), outp.data());
        for (const char* l : labels) {                                           // the jobs that were started early
            auto it = early.jobs.find(l);
            if (it == early.jobs.end()) continue;
            zk_g1_projective eo[2];
            void* eop[2] = {&eo[0], &eo[1]};
            ZkEarlyMsm* em = it->second;
            early.jobs.erase(it);
            const int erc = zk_msm_early_finish(ctx, em, eop);
            if (brc == ZK_OK) brc = erc;
            acc[0][0][l] = eo[0];
            if (P.bounds.count(l)) acc[0][1][l] = eo[1];
        }
        laps.lap("commit.msm");
        std::vector<std::pair<std::pair<int, std::string>, zk_g1_projective>> bl;
        for (auto& b : blinds) bl.push_back({b.first, b.second.get()});          // joined before any return
        laps.lap("commit.blinds");
        ZK_TRY(brc);
        for (size_t i = 0; i < jb.size(); i++) acc[slot[i].lane][slot[i].which][slot[i].label] = outs[i];
        for (auto& b : bl)                                                       // the MAC lane of this party's fresh blinds is the share itself
            for (int lane = 0; lane < LANES; lane++) {
                auto it = acc[lane][b.first.first].find(b.first.second);
                if (it == acc[lane][b.first.first].end()) continue;
                zk_g1_projective t;
                zk_g1_add(&it->second, &b.second, &t);
                it->second = t;
            }
        if (shared) {                                                            // first_comms.publicize() (lib.rs:180,205,228)
            std::vector<HF> nofr[2], ofr;
            std::vector<zk_g1_projective> pts[2], opened;
            std::vector<std::pair<std::string, int>> what;
            for (const char* l : labels) {
                if (!label_shared(l)) continue;
                for (int which = 0; which < 2; which++) {
                    if (!acc[0][which].count(l)) continue;
                    for (int lane = 0; lane < LANES; lane++) pts[lane].push_back(acc[lane][which][l]);
                    what.push_back({l, which});
                }
            }
            ZK_TRY(open_small<LANES>(nt, nofr, pts, ofr, opened));
            for (size_t i = 0; i < what.size(); i++) acc[0][what[i].second][what[i].first] = opened[i];
        }
        std::vector<uint8_t> bytes;
        std::vector<zk_g1_projective> all;
        for (const char* l : labels) {
            all.push_back(acc[0][0][l]);
            if (acc[0][1].count(l)) all.push_back(acc[0][1][l]);
        }
        const std::vector<Affine<G1Field>> aff = batch_to_aff(all);
        size_t ai = 0;
        for (const char* l : labels) {
            Comm c;
            c.c = aff[ai++];
            c.has_shift = acc[0][1].count(l) != 0;
            c.s = c.has_shift ? aff[ai++] : aff_inf<G1Field>();
            P.comms[l] = c;
            comm_tobytes(c, bytes);
        }
        return zk_fsrng_absorb(fs, bytes.data(), bytes.size());                  // to_bytes![comms, EmptyMessage]
    };

    laps.lap("setup");
    // =========================== round 1 (prover.rs:216-404), every lane ===========================
    const size_t md = 3 * n + 2 - 3;                                             // mask polynomial degree, zk_bound = 1
    char* rnd = P.dev("rnd", 3 + md + 1);                                        // this party's (share of the) prover randomness: both lanes read it
    ZK_TRY(P.rc);
    {
        const size_t host_n = mask_on_device ? 3 : 3 + md + 1;
        std::vector<zk_fr> h(host_n);
        ZK_TRY(zk_rng_fill_fr(P.rng, h.data(), host_n));
        ZK_TRY(zk_memcpy_h2d(ctx, rnd, h.data(), host_n * 32));
        if (mask_on_device) {
            uint8_t key[32];
            ZK_TRY(zk_rng_fill_bytes(P.rng, key, 32));
            ZK_TRY(zk_fr_random_dev(ctx, key, 0, rnd + 96, md + 1));
        }
    }
    const size_t nwq = n + 1 - X.size;
    char *z_a[2], *z_b[2], *xb[2], *wq[2], *mask[2], *tmp[2];
    for (int l = 0; l < LANES; l++) {                                            // the mask polynomial first: its commitment (the round's longest job) starts now
        Prover& Q = PL[l];
        mask[l] = Q.dev("mask", md + 1);
        char* mq = Q.dev("mask_q", md + 1); char* mr = Q.dev("mask_r", n);
        Q.d2d(mask[l], rnd + 96, md + 1);
        ZK_TRY(Q.rc);
        ZK_TRY(zk_poly_divide_by_vanishing_dev(ctx, mask[l], md + 1, H.log, mq, mr));
        Q.op(ZK_OP_SUB, mask[l], mr, mask[l], 1);                                // the sum over H becomes zero
        Q.polys["mask_poly"] = Poly{mask[l], md + 1};
        ZK_TRY(Q.rc);
    }
    if (!shared) ZK_TRY(start_early("mask_poly"));                               // (over shares it is a shared oracle: every lane's job stays in the batch)
    for (int l = 0; l < LANES; l++) {
        Prover& Q = PL[l];
        z_a[l] = Q.dev("z_a", n); z_b[l] = Q.dev("z_b", n);
        ZK_TRY(Q.rc);
        ZK_TRY(zk_r1cs_matvec_dev(ctx, ix->r1cs, 0, zb[l], z_a[l], n));
        ZK_TRY(zk_r1cs_matvec_dev(ctx, ix->r1cs, 1, zb[l], z_b[l], n));
        xb[l] = Q.dev("x_poly", X.size);
        Q.d2d(xb[l], zb[l], X.size);
        Q.ntt(xb[l], X, 1);
        const Poly x_poly{xb[l], X.size};
        char* x_evals = Q.fft(H, x_poly, "x_evals");
        char* w_evals = Q.dev("w_evals", n);
        tmp[l] = Q.dev("tmp_h", n);
        ZK_TRY(Q.rc);
        ZK_TRY(zk_fr_gather_dev(ctx, zb[l], ix->w_idx, n, w_evals));
        ZK_TRY(zk_fr_gather_dev(ctx, x_evals, ix->x_idx, n, tmp[l]));
        Q.op(ZK_OP_SUB, w_evals, tmp[l], w_evals, n);
        char* za = Q.dev("za_c", n); char* zbb = Q.dev("zb_c", n);
        Q.d2d(za, z_a[l], n); Q.d2d(zbb, z_b[l], n);
        Q.ntt_batch({w_evals, za, zbb}, H, 1);                                   // the three interpolations of the round: one launch per pass
        const Poly w_h = Q.blind(w_evals, n, rnd, "w_h");
        wq[l] = Q.dev("w_poly", nwq);
        char* wr = Q.dev("w_rem", X.size);
        ZK_TRY(Q.rc);
        ZK_TRY(zk_poly_divide_by_vanishing_dev(ctx, w_h.p, n + 1, X.log, wq[l], wr));
        // (over shares the remainder is a share of zero: the reference's assert!(remainder.is_zero()) cannot be evaluated locally)
        if (!shared) ZK_TRY(check_later(wr, X.size, "zk_marlin_prove: w polynomial is not divisible by v_X"));
        Q.polys["w"] = Poly{wq[l], nwq};
        Q.polys["z_a"] = Q.blind(za, n, rnd + 32, "z_a_poly");
        Q.polys["z_b"] = Q.blind(zbb, n, rnd + 64, "z_b_poly");
        ZK_TRY(Q.rc);
    }
    laps.lap("polys");
    ZK_TRY(commit_round({"w", "z_a", "z_b", "mask_poly"}));
    ZK_TRY(settle());
    laps.lap("commit");
    const HF alpha = sample_outside(H), eta_a = P.next_fr(fs), eta_b = P.next_fr(fs), eta_c = P.next_fr(fs);

    laps.lap("round1");
    // =========================== round 2 (prover.rs:438-565) ===========================
    // public: r(alpha, X) on H, t = sum_M eta_M M^T r, their evaluations over the multiplication domain (computed once)
    const HF v_h_alpha = H.vanishing(alpha), one = HF::one();
    char* elems = P.dev("h_elems", n);
    char* ra = P.dev("r_alpha", n);
    ZK_TRY(P.rc);
    { zk_fr g = H.gen.abi(), o = one.abi(), a = alpha.abi();
      ZK_TRY(zk_fr_powers_dev(ctx, &g, &o, n, elems));
      ZK_TRY(zk_fr_powers_dev(ctx, &o, &a, n, ra)); }                            // the constant vector alpha
    P.op(ZK_OP_SUB, ra, elems, ra, n);
    ZK_TRY(P.rc);
    ZK_TRY(zk_fr_batch_inverse_dev(ctx, ra, n));
    P.scale(ra, v_h_alpha, ra, n);                                               // r(alpha, X) on H (ahp/mod.rs:352-360)
    char* t_ev = P.dev("t_ev", n);
    const HF etas[3] = {eta_a, eta_b, eta_c};
    for (int which = 0; which < 3; which++) {                                    // calculate_t on the transposed matrices
        ZK_TRY(P.rc);
        ZK_TRY(zk_r1cs_matvec_dev(ctx, ix->r1cs_t, which, ra, which == 0 ? t_ev : tmp[0], n));
        if (which == 0) P.scale(t_ev, etas[0], t_ev, n);
        else { P.scale(tmp[0], etas[which], tmp[0], n); P.op(ZK_OP_ADD, t_ev, tmp[0], t_ev, n); }
    }
    P.ntt_batch({t_ev, ra}, H, 1);
    for (int l = 0; l < LANES; l++) PL[l].polys["t"] = Poly{t_ev, n};
    ZK_TRY(P.rc);
    ZK_TRY(start_early("t"));                                                    // public: the product of the two witness vectors and h_1 are still to come
    const Poly r_alpha_poly{ra, n};
    const Dom MUL(std::max(std::max(md + 1, n + 2 * n + 1), n + n + 1));
    char* e_rp = P.padded(MUL, r_alpha_poly, "e_r"); char* e_tp = P.padded(MUL, P.polys["t"], "e_t");
    P.ntt_batch({e_rp, e_tp}, MUL, 0);
    char *e_a[2], *e_b[2], *e_s[2], *e_z[2];
    for (int l = 0; l < LANES; l++) {
        Prover& Q = PL[l];
        char* zp = Q.dev("z_poly", n + 1);                                       // z = w v_X + x
        Q.zero(zp, n + 1);
        Q.d2d(zp + 32 * X.size, wq[l], nwq);
        Q.op(ZK_OP_SUB, zp, wq[l], zp, nwq);
        Q.op(ZK_OP_ADD, zp, xb[l], zp, X.size);
        const Poly z_poly{zp, n + 1};
        e_a[l] = Q.padded(MUL, Q.polys["z_a"], "e_a"); e_b[l] = Q.padded(MUL, Q.polys["z_b"], "e_b");
        e_s[l] = Q.dev("e_s", MUL.size);
        e_z[l] = Q.padded(MUL, z_poly, "e_z");
        Q.ntt_batch({e_a[l], e_b[l], e_z[l]}, MUL, 0);
        ZK_TRY(Q.rc);
    }
    // z_a z_b: the one product of two witness vectors (`DensePolynomial::mul` on MpcField = FieldShare::batch_mul)
    if (!shared) P.op(ZK_OP_MUL, e_a[0], e_b[0], e_s[0], MUL.size);
    else ZK_TRY(zk_shared_beaver_mul(nt, LANES, (const void* const*)e_a, (const void* const*)e_b, (void* const*)e_s, MUL.size, tx, ty, tz, "mp_bv"));
    char *hq[2], *hr[2];
    for (int l = 0; l < LANES; l++) {
        Prover& Q = PL[l];
        // r(alpha, X) (eta_c z_a z_b + eta_a z_a + eta_b z_b) - z t on the multiplication domain (public * own value: local): one pass
        ZK_TRY(Q.rc);
        ZK_TRY(zk_fr_outer_q1_launch(ctx, e_s[l], e_a[l], e_b[l], e_z[l], e_rp, e_tp, eta_a.v.l, eta_b.v.l, eta_c.v.l, e_s[l], MUL.size));
        Q.ntt(e_s[l], MUL, 1);                                                   // q_1 (prover.rs:517-541)
        Q.op(ZK_OP_ADD, e_s[l], mask[l], e_s[l], md + 1);
        hq[l] = Q.dev("h1_q", MUL.size - n); hr[l] = Q.dev("h1_r", n);
        ZK_TRY(Q.rc);
        ZK_TRY(zk_poly_divide_by_vanishing_dev(ctx, e_s[l], MUL.size, H.log, hq[l], hr[l]));
    }
    {   // the outer sum-check's zero test (prover.rs:547-550): over shares the constant term is opened
        bool zero_sum = true;
        if (!shared) ZK_TRY(check_later(hr[0], 1, "zk_marlin_prove: outer sum-check: the sum over H is not zero (unsatisfied constraint system)"));
        else {
            char* zo = P.dev("zero_open", 1); char* zd = P.dev("zero_dx", 1);
            ZK_TRY(P.rc);
            if (LANES == 2) ZK_TRY(zk_shared_spdz_open_vec(nt, hr[0], hr[1], 1, zo, zd));
            else ZK_TRY(nt.open_vec(hr[0], 1, zo));
            zero_sum = P.is_zero(zo, 1);
        }
        ZK_TRY(P.rc);
        if (!zero_sum) ZK_FAIL(ctx, ZK_ERR_STATE, "zk_marlin_prove: outer sum-check: the sum over H is not zero (unsatisfied constraint system)");
    }
    for (int l = 0; l < LANES; l++) {
        PL[l].polys["g_1"] = Poly{hr[l] + 32, n - 1};
        PL[l].polys["h_1"] = Poly{hq[l], std::min(MUL.size - n, 2 * n + 2 - 1)};
    }
    laps.lap("polys");
    ZK_TRY(commit_round({"t", "g_1", "h_1"}));
    ZK_TRY(settle());
    laps.lap("commit");
    const HF beta = sample_outside(H);

    laps.lap("round2");
    // =========================== round 3 (prover.rs:583-716): public values only ===========================
    const HF vv = v_h_alpha * H.vanishing(beta);
    char* f_ev = P.dev("f_ev", K.size);
    char* a_ev = P.dev("a_ev", B.size); char* b_ev = P.dev("b_ev", B.size);
    ZK_TRY(P.rc);
    { zk_fr al = alpha.abi(), be = beta.abi(), v = vv.abi(), et[3] = {eta_a.abi(), eta_b.abi(), eta_c.abi()};
      ZK_TRY(zk_marlin_round3_f_evals_dev(ctx, ix->on_k, K.size, &al, &be, et, &v, f_ev));
      ZK_TRY(zk_marlin_round3_ab_evals_dev(ctx, ix->on_b, B.size, &al, &be, et, &v, a_ev, b_ev)); }
    P.ntt(f_ev, K, 1);
    const Poly f{f_ev, K.size};
    for (int l = 0; l < LANES; l++) PL[l].polys["g_2"] = Poly{f_ev + 32, K.size - 1};
    ZK_TRY(P.rc);
    ZK_TRY(start_early("g_2"));                                                  // both of its commitments (degree bound |K| - 2), under the division that yields h_2
    char* f_on_b = P.fft(B, f, "f_on_b");                                       // a - b f on B itself (degree <= 4|K| - 4 < |B|)
    P.op(ZK_OP_MUL, b_ev, f_on_b, b_ev, B.size);
    P.op(ZK_OP_SUB, a_ev, b_ev, a_ev, B.size);
    P.ntt(a_ev, B, 1);
    char* h2q = P.dev("h2_q", B.size - K.size); char* h2r = P.dev("h2_r", K.size);
    ZK_TRY(P.rc);
    ZK_TRY(zk_poly_divide_by_vanishing_dev(ctx, a_ev, B.size, K.log, h2q, h2r));
    ZK_TRY(check_later(h2r, K.size, "zk_marlin_prove: inner sum-check: a - b f is not divisible by v_K"));
    for (int l = 0; l < LANES; l++) {
        PL[l].polys["g_2"] = Poly{f_ev + 32, K.size - 1};
        PL[l].polys["h_2"] = Poly{h2q, B.size - K.size};
    }
    laps.lap("polys");
    ZK_TRY(commit_round({"g_2", "h_2"}));
    ZK_TRY(settle());
    laps.lap("commit");
    const HF gamma = P.next_fr(fs);

    laps.lap("round3");
    // =========================== evaluations and linear combinations ===========================
    std::map<std::string, HF> single;
    {   // the evaluations of the query set in one batch (two launches, one copy back); z_b and g_1 are shared: every lane's
        // evaluation, opened (`evaluations.publicize()`, lib.rs:296)
        std::vector<std::pair<std::string, HF>> want = {{"z_b", beta}, {"g_1", beta}, {"t", beta}, {"g_2", gamma}};
        for (const char* m : {"a", "b", "c"})
            for (const char* part : {"_row", "_col", "_row_col"}) want.push_back({std::string(m) + part, gamma});
        std::vector<zk_poly_ref> refs;
        std::vector<zk_fr> pts;
        for (size_t i = 0; i < want.size(); i++) {
            const Poly& p = P.polys[want[i].first];
            refs.push_back(zk_poly_ref{p.p, p.n});
            pts.push_back(want[i].second.abi());
        }
        if (LANES == 2)
            for (const char* l : {"z_b", "g_1"}) { const Poly& p = PL[1].polys[l]; refs.push_back(zk_poly_ref{p.p, p.n}); pts.push_back(beta.abi()); }
        std::vector<zk_fr> vals(refs.size());
        ZK_TRY(zk_poly_evaluate_batch_dev(ctx, refs.data(), pts.data(), refs.size(), vals.data()));
        std::map<std::string, HF> at;
        for (size_t i = 0; i < want.size(); i++) at[want[i].first] = HF::from_abi(vals[i]);
        if (shared) {
            std::vector<HF> frs[2] = {{at["z_b"], at["g_1"]}, {}}, ofr;
            if (LANES == 2) frs[1] = {HF::from_abi(vals[want.size()]), HF::from_abi(vals[want.size() + 1])};
            std::vector<zk_g1_projective> nog[2], og;
            ZK_TRY(open_small<LANES>(nt, frs, nog, ofr, og));
            at["z_b"] = ofr[0];
            at["g_1"] = ofr[1];
        }
        for (const char* l : {"z_b", "g_1", "t", "g_2"}) single[l] = at[l];
        const HF ba0 = beta * alpha;
        for (const char* m : {"a", "b", "c"}) {
            const std::string s(m);
            single[s + "_denom"] = ba0 - alpha * at[s + "_row"] - beta * at[s + "_col"] + at[s + "_row_col"];
        }
    }
    const HF ba = beta * alpha;
    ZK_TRY(P.rc);
    // construct_linear_combinations (ahp/mod.rs:112-290)
    const HF v_h_beta = H.vanishing(beta), v_x_beta = beta.pow(ni) - one;
    const HF r_alpha_at_beta = (alpha == beta) ? HF::from_u64(n) * alpha.pow(n - 1) : (v_h_alpha - v_h_beta) * (alpha - beta).inv();
    HF x_beta = HF::zero();
    {
        std::vector<HF> x{one};
        x.insert(x.end(), pub.begin(), pub.end());
        if (v_x_beta.is_zero()) {
            HF g = one;
            for (size_t k = 0; k < ni; k++, g = g * X.gen) if (g == beta) x_beta = x[k];
        } else {
            const HF l0 = v_x_beta * HF::from_u64(ni).inv();
            HF g = one;
            for (size_t k = 0; k < ni; k++, g = g * X.gen) x_beta = x_beta + x[k] * (l0 * g * (beta - g).inv());
        }
    }
    const HF z_b_beta = single["z_b"], t_beta = single["t"], g_1_beta = single["g_1"], g_2_gamma = single["g_2"];
    const HF da = single["a_denom"], db = single["b_denom"], dc = single["c_denom"];
    std::map<std::string, std::vector<Term>> lcs;
    lcs["z_b"] = {{one, "z_b"}}; lcs["g_1"] = {{one, "g_1"}}; lcs["t"] = {{one, "t"}}; lcs["g_2"] = {{one, "g_2"}};
    lcs["outer_sumcheck"] = {{one, "mask_poly"}, {r_alpha_at_beta * (eta_a + eta_c * z_b_beta), "z_a"}, {r_alpha_at_beta * eta_b * z_b_beta, nullptr},
                             {(t_beta * v_x_beta).neg(), "w"}, {(t_beta * x_beta).neg(), nullptr}, {v_h_beta.neg(), "h_1"},
                             {(beta * g_1_beta).neg(), nullptr}};
    lcs["a_denom"] = {{ba, nullptr}, {alpha.neg(), "a_row"}, {beta.neg(), "a_col"}, {one, "a_row_col"}};
    lcs["b_denom"] = {{ba, nullptr}, {alpha.neg(), "b_row"}, {beta.neg(), "b_col"}, {one, "b_row_col"}};
    lcs["c_denom"] = {{ba, nullptr}, {alpha.neg(), "c_row"}, {beta.neg(), "c_col"}, {one, "c_row_col"}};
    const HF b_expr = da * db * dc * (gamma * g_2_gamma + t_beta * HF::from_u64(K.size).inv());
    lcs["inner_sumcheck"] = {{eta_a * db * dc * vv, "a_val"}, {eta_b * da * dc * vv, "b_val"}, {eta_c * db * da * vv, "c_val"},
                             {b_expr.neg(), nullptr}, {K.vanishing(gamma).neg(), "h_2"}};
    const char* const EVAL_LABELS[7] = {"a_denom", "b_denom", "c_denom", "g_1", "g_2", "t", "z_b"};
    std::vector<HF> evaluations;
    std::vector<uint8_t> ev_bytes;
    for (const char* l : EVAL_LABELS) { evaluations.push_back(single[l]); single[l].bytes(ev_bytes); }
    ZK_TRY(zk_fsrng_absorb(fs, ev_bytes.data(), ev_bytes.size()));
    HF xi;
    {   // u128::rand(&mut fs_rng).into()  (lib.rs:300)
        uint64_t w[2];
        ZK_TRY(zk_rng_next_u128(fs, w));
        xi = HF::from_u64(w[0]) + HF::from_u64(w[1]) * HF::from_u64((uint64_t)1 << 32) * HF::from_u64((uint64_t)1 << 32);
    }

    laps.lap("evals+lc");
    // =========================== open_combinations (marlin/mod.rs:213-306, marlin_pc/mod.rs:245-340) ===========================
    // Over shares: the witness of a share combination is a share of the witness.  A combination with a shared oracle in it runs on
    // every lane, public oracles entering it on the leader only (shift(): in both lanes, mac_share = 1 there), and its witness
    // (and random_v) is opened; a combination of public oracles only (the query point gamma) is computed alike by every party.
    const std::vector<std::string> QUERY[2] = {{"g_1", "outer_sumcheck", "t", "z_b"}, {"a_denom", "b_denom", "c_denom", "g_2", "inner_sumcheck"}};
    const HF points[2] = {beta, gamma};
    std::vector<const zk_bases*> jb; std::vector<size_t> joff, jlen; std::vector<const void*> jsc;
    size_t counts[2][2] = {{0, 0}, {0, 0}};                                      // [query point][lane]
    std::vector<ZkTask<zk_g1_projective>> extra[2];                         // the blinding witnesses: host threads, joined after the batch
    auto small_async = [&](const std::vector<HF>& c) { return zk_async(ctx, [ctx, &gamma_pts, c] { return small_msm_par(ctx, gamma_pts, c); }); };
    bool has_rv[2] = {false, false}, q_shared[2] = {false, false};
    HF rvs[2];
    for (int q = 0; q < 2; q++) {
        const HF z = points[q];
        std::vector<std::pair<std::string, HF>> terms;                            // polynomial label -> accumulated coefficient (first-use order)
        std::vector<HF> r_comb, sr, srw;
        std::vector<std::pair<std::string, HF>> shifted;
        HF cj = one;                                                              // xi^j
        for (const std::string& label : QUERY[q]) {
            const auto& lc = lcs[label];
            std::vector<Term> ps;
            for (const Term& t : lc) if (t.label) ps.push_back(t);
            const HF c0 = cj;
            cj = cj * xi;
            for (const Term& t : ps) {
                auto it = std::find_if(terms.begin(), terms.end(), [&](const std::pair<std::string, HF>& e) { return e.first == t.label; });
                if (it == terms.end()) terms.push_back({t.label, t.c * c0}); else it->second = it->second + t.c * c0;
                acc_scaled(r_comb, P.rands[t.label].first, t.c * c0);
            }
            if (lc.size() == 1 && P.bounds.count(ps[0].label)) {
                const HF c1 = cj;
                cj = cj * xi;
                shifted.push_back({ps[0].label, c1});
                acc_scaled(sr, P.rands[ps[0].label].second, c1);
            }
        }
        bool any_shared = false;
        for (auto& t : terms) any_shared = any_shared || (shared && label_shared(t.first.c_str()));
        q_shared[q] = any_shared;
        size_t cn = 0;
        for (auto& t : terms) cn = std::max(cn, P.polys[t.first].n);
        for (int l = 0; l < (any_shared ? LANES : 1); l++) {
            Prover& Q = PL[l];
            char* comb = Q.dev("comb" + std::to_string(q), cn);
            {   // one launch for the whole combination (vec_ops.hip::k_lincomb)
                std::vector<const void*> tp; std::vector<size_t> tn; std::vector<std::array<uint32_t, 9>> tk;
                for (auto& t : terms) {
                    if (any_shared && !label_shared(t.first.c_str()) && !leader) continue;   // a public oracle in a shared combination: the leader's
                    const Poly& p = Q.polys[t.first];
                    tp.push_back(p.p); tn.push_back(p.n);
                    std::array<uint32_t, 9> k;
                    for (int i = 0; i < 9; i++) k[i] = t.second.v.l[i];
                    tk.push_back(k);
                }
                ZK_TRY(Q.rc);
                ZK_TRY(zk_fr_lincomb_launch(ctx, (int)tp.size(), tp.data(), tn.data(), (const uint32_t (*)[9])tk.data(), comb, cn));
            }
            char* quo = Q.dev("quo" + std::to_string(q), cn);
            ZK_TRY(Q.rc);
            { zk_fr zz = z.abi(); ZK_TRY(zk_poly_divide_by_linear_dev(ctx, comb, cn, &zz, quo, nullptr)); }
            const size_t first_job = jb.size();
            jb.push_back(P.pg); joff.push_back(0); jsc.push_back(quo); jlen.push_back(cn - 1);
            int si = 0;
            for (auto& sh : shifted) {
                if (any_shared && !label_shared(sh.first.c_str()) && !leader) continue;
                const Poly& p = Q.polys[sh.first];
                char* wq2 = Q.dev("swit" + std::to_string(q) + "_" + std::to_string(si++), p.n);
                ZK_TRY(Q.rc);
                { zk_fr zz = z.abi(); ZK_TRY(zk_poly_divide_by_linear_dev(ctx, p.p, p.n, &zz, wq2, nullptr)); }
                Q.scale(wq2, sh.second, wq2, p.n - 1);
                jb.push_back(P.pg); joff.push_back(P.max_degree - P.bounds[sh.first]); jsc.push_back(wq2); jlen.push_back(p.n - 1);
            }
            counts[q][l] = jb.size() - first_job;
        }
        bool hiding = false;
        for (auto& v : r_comb) hiding = hiding || !v.is_zero();
        if (hiding) {
            extra[q].push_back(small_async(host_div_linear(r_comb, z)));
            has_rv[q] = true;
            rvs[q] = host_eval(r_comb, z);
        }
        for (auto& sh : shifted) {
            const std::vector<HF>& sb = P.rands[sh.first].second;
            if (!sb.empty()) acc_scaled(srw, host_div_linear(sb, z), sh.second);
        }
        if (!srw.empty()) extra[q].push_back(small_async(srw));
        if (!shifted.empty() && has_rv[q]) rvs[q] = rvs[q] + host_eval(sr, z);
    }
    ZK_TRY(lanes_rc());
    std::vector<zk_g1_projective> outs(jb.size());
    std::vector<void*> outp(jb.size());
    for (size_t i = 0; i < jb.size(); i++) outp[i] = &outs[i];
    const int orc = zk_msm_batch_dev(ctx, jb.size(), jb.data(), joff.data(), jsc.data(), jlen.data(), outp.data());
    std::vector<zk_g1_projective> extra_pts[2];
    for (int q = 0; q < 2; q++) for (auto& f : extra[q]) extra_pts[q].push_back(f.get());
    ZK_TRY(orc);
    Affine<G1Field> wit[2];
    {
        size_t k = 0;
        std::vector<zk_g1_projective> wits;
        for (int q = 0; q < 2; q++) {
            zk_g1_projective wl[2];
            const int nl = q_shared[q] ? LANES : 1;
            for (int l = 0; l < nl; l++) {
                zk_g1_projective w = outs[k];
                for (size_t i = 1; i < counts[q][l]; i++) { zk_g1_projective t; zk_g1_add(&w, &outs[k + i], &t); w = t; }
                for (auto& e : extra_pts[q]) { zk_g1_projective t; zk_g1_add(&w, &e, &t); w = t; }    // own randomness: the same on the MAC lane
                k += counts[q][l];
                wl[l] = w;
            }
            if (q_shared[q]) {                                                  // the witness (and random_v) of a shared combination: opened
                std::vector<HF> frs[2], ofr;
                std::vector<zk_g1_projective> pts[2], og;
                for (int l = 0; l < LANES; l++) { pts[l].push_back(wl[l]); if (has_rv[q]) frs[l].push_back(rvs[q]); }
                ZK_TRY(open_small<LANES>(nt, frs, pts, ofr, og));
                wl[0] = og[0];
                if (has_rv[q]) rvs[q] = ofr[0];
            }
            wits.push_back(wl[0]);
        }
        const std::vector<Affine<G1Field>> wa = batch_to_aff(wits);
        wit[0] = wa[0];
        wit[1] = wa[1];
    }

    laps.lap("open");
    // =========================== Proof::serialize (data_structures.rs:99-110, derive order) ===========================
    std::vector<uint8_t> out;
    auto u64 = [&](uint64_t v) { for (int i = 0; i < 8; i++) out.push_back((uint8_t)(v >> (8 * i))); };
    auto g1c = [&](const Affine<G1Field>& a) { uint8_t b[48]; g1_serialize(a, b); out.insert(out.end(), b, b + 48); };
    const std::vector<std::vector<const char*>> ROUNDS = {{"w", "z_a", "z_b", "mask_poly"}, {"t", "g_1", "h_1"}, {"g_2", "h_2"}};
    u64(3);
    for (auto& rnd_labels : ROUNDS) {
        u64(rnd_labels.size());
        for (const char* l : rnd_labels) {
            const Comm& c = P.comms[l];
            g1c(c.c);
            out.push_back(c.has_shift ? 1 : 0);
            if (c.has_shift) g1c(c.s);
        }
    }
    u64(evaluations.size());
    for (auto& e : evaluations) e.bytes(out);
    u64(3); out.push_back(0); out.push_back(0); out.push_back(0);               // three EmptyMessage
    u64(2);
    for (int q = 0; q < 2; q++) {
        g1c(wit[q]);
        out.push_back(has_rv[q] ? 1 : 0);
        if (has_rv[q]) rvs[q].bytes(out);
    }
    out.push_back(0);                                                            // BatchLCProof.evals = None
    if (out.size() > cap) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_marlin_prove: output buffer too small");
    memcpy(proof_out, out.data(), out.size());
    *proof_len = out.size();
    if (bytes_sent) *bytes_sent = nt.bytes;
    return ZK_OK;
}

}  // namespace

extern "C" int zk_marlin_prove(zk_ctx* ctx, const zk_marlin_index* ix, const zk_bases* powers_g, const zk_bases* powers_gamma_g,
                               const void* z_dev, zk_rng* zk_rng_, int mask_on_device, uint8_t* proof_out, size_t cap, size_t* proof_len) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !ix || !powers_g || !powers_gamma_g || !z_dev || !zk_rng_ || !proof_out || !proof_len) return ZK_ERR_ARG;
    const void* z[2] = {z_dev, nullptr};
    const void* none[2] = {nullptr, nullptr};
    return marlin_impl<1>(ctx, ix, powers_g, powers_gamma_g, z, zk_rng_, mask_on_device, false, none, none, none, nullptr, proof_out, cap,
                          proof_len, nullptr);
    ZK_API_END
}

// MpcMarlin::prove over additive shares (src/marlin.rs:56): z_share_dev = this party's share of the padded assignment, zk_rng =
// this party's own generator (its share of the prover's randomness); tx / ty / tz = Beaver triple shares for the one product of
// round 2 (4|H| elements... the multiplication domain) or NULL for DummyFieldTripleSource.  Every party returns the same bytes.
extern "C" int zk_marlin_prove_shared(zk_ctx* ctx, const zk_marlin_index* ix, const zk_bases* powers_g, const zk_bases* powers_gamma_g,
                                      const void* z_share_dev, zk_rng* zk_rng_, int mask_on_device, const void* tx, const void* ty,
                                      const void* tz, const zk_net_vtable* net, uint8_t* proof_out, size_t cap, size_t* proof_len,
                                      uint64_t* bytes_sent) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !ix || !powers_g || !powers_gamma_g || !z_share_dev || !zk_rng_ || !proof_out || !proof_len) return ZK_ERR_ARG;
    const void* z[2] = {z_share_dev, nullptr};
    const void *txs[2] = {tx, nullptr}, *tys[2] = {ty, nullptr}, *tzs[2] = {tz, nullptr};
    return marlin_impl<1>(ctx, ix, powers_g, powers_gamma_g, z, zk_rng_, mask_on_device, true, txs, tys, tzs, net, proof_out, cap, proof_len,
                          bytes_sent);
    ZK_API_END
}

// ... over SPDZ shares (the `malicious` feature; BASELINE config 5's prover): lanes [0] = share, [1] = MAC share.
extern "C" int zk_marlin_prove_shared_spdz(zk_ctx* ctx, const zk_marlin_index* ix, const zk_bases* powers_g, const zk_bases* powers_gamma_g,
                                           const void* const z_lanes_dev[2], zk_rng* zk_rng_, int mask_on_device,
                                           const void* const tx_lanes[2], const void* const ty_lanes[2], const void* const tz_lanes[2],
                                           const zk_net_vtable* net, uint8_t* proof_out, size_t cap, size_t* proof_len, uint64_t* bytes_sent) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !ix || !powers_g || !powers_gamma_g || !z_lanes_dev || !z_lanes_dev[0] || !z_lanes_dev[1] || !zk_rng_ || !proof_out || !proof_len)
        return ZK_ERR_ARG;
    const void* none[2] = {nullptr, nullptr};
    return marlin_impl<2>(ctx, ix, powers_g, powers_gamma_g, z_lanes_dev, zk_rng_, mask_on_device, true, tx_lanes ? tx_lanes : none,
                          ty_lanes ? ty_lanes : none, tz_lanes ? tz_lanes : none, net, proof_out, cap, proof_len, bytes_sent);
    ZK_API_END
}
