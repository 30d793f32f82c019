// host_trait_collab_groth16.cpp -- create_proof::<MpcPairingEngine, C>, line for line, over the TRAIT-SHAPED entry points only.
//
// What the reference's collaborative prover executes once its dispatch points are overridden:
// src/groth16.rs:68-183 (create_proof) with R1CStoQAP::witness_map (:240-306) over E = MpcPairingEngine<Bls12_377, S> -- P parties
// (threads of this process, the reference's LocalTestNet shape: mpc-net/src/multi.rs:419-443), each with its own context, its own
// copy of the proving key and its own shares, all in HOST memory in the caller's element types:
//     MpcField<Fr, S>  = enum { Public(Fr), Shared(S) }                     mpc-algebra/src/wire/field.rs:37-40
//                        S = AdditiveFieldShare { val } (40-byte elements) or SpdzFieldShare { sh, mac } (72 bytes)
//     MpcG1Affine      = { val: MpcGroup<G1Affine, ..> } = enum { Public(GroupAffine {x, y, infinity}), Shared(..) }
// laid out as rustc lays such enums out: a discriminant byte and the payload at 8-byte alignment -- discriminant first (what
// rustc does today) or last (`tag last`: nothing in the library may depend on it; the discriminant VALUES are swapped there too).
// Every library call below is one of
//     zk_mpc_fft_in_place                            EvaluationDomain::{ifft, coset_fft, coset_ifft}_in_place      (x7, :278-303)
//     zk_mpc_batch_product_in_place                  MpcField::batch_product_in_place -> FieldShare::batch_mul      (x1, :285)
//     zk_mpc_divide_by_vanishing_on_coset_in_place   divide_by_vanishing_poly_on_coset_in_place                     (x1, :302)
//     zk_mpc_msm_g1 / zk_mpc_msm_g2                  MpcG1Affine / MpcG2Affine::multi_scalar_mul                    (x5, :106,110,193)
//     zk_g1_mul / zk_g1_add / ... / zk_g1_serialize  the arithmetic of the O(1) tail (share/group.rs, wire/group.rs), (:112-176)
// plus the transport (zk_net_vtable: all_gather_bytes = MpcNet::broadcast_bytes, open_sum_fr_dev = batch_open of a device vector),
// implemented here over shared memory and a barrier.  Everything else is the caller's own code: evaluate_constraint and
// `ab_i -= c_i` over MpcField (wire/field.rs:339-362,414-437), the three GroupShare::scale calls (share/group.rs:72-111 with
// DummyGroupTripleSource, wire/group.rs:36-55) and Proof::reveal (arkworks/groth16/src/reveal.rs:7-10), written out below.
//
// Output: one JSON line per proof {"proof": hex (identical on every party, checked here), "ms": {...party 0's laps...},
// "ms_max_lib": the slowest party's library time}, then a line with r, s (the sums of the parties' shares, canonical hex) and the
// cache counters.  tests/test_gpu_trait_path.py compares the bytes with the oracle's known-trapdoor prediction on the summed
// inputs; bench.py's `trait_path_collab` leg reports the times.
//
//   host_trait_collab_groth16 <log2 of the QAP domain> <proofs> <parties> [additive|spdz] [tagfirst|taglast] [verify|trust] [sync2]
// sync2: after the second proof every party waits for its window-multiple builds (zk_bases_cache_sync, outside the timed laps), so that
// the proofs from the third on are the steady state -- for P parties on ONE GPU, whose builders otherwise find no quiet device.
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "zkmpc_hip.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- the caller's own Fr (Fp256 in Montgomery form, as in host_trait_groth16.cpp) ----
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
    uint64_t x = 1;
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
    if (geq_mod(r.l)) sub_mod_raw(r.l);
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
static const zk_fr FR_ZERO{};
static bool fr_is_zero(const zk_fr& a) { return !(a.l[0] | a.l[1] | a.l[2] | a.l[3]); }

// ---- the transport the parties share (LocalTestNet) ----
struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0;
    unsigned long gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long g = gen;
        if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};
struct LocalNet {
    int parties;
    Barrier bar;
    std::vector<uint8_t> host;
    void* gathered = nullptr;     // device: parties x n field elements
    size_t gathered_elems = 0;
    explicit LocalNet(int p) : parties(p), bar(p) {}
};
struct Party {
    LocalNet* net; int id; zk_ctx* ctx;
    bool leader() const { return id == 0; }
};
static int all_gather_bytes(void* user, const uint8_t* mine, size_t len, uint8_t* out_all) {
    Party* me = (Party*)user;
    LocalNet* net = me->net;
    if (me->id == 0) net->host.assign((size_t)net->parties * len, 0);
    net->bar.wait();
    memcpy(net->host.data() + (size_t)me->id * len, mine, len);
    net->bar.wait();
    memcpy(out_all, net->host.data(), (size_t)net->parties * len);
    net->bar.wait();
    return 0;
}
static int open_sum_fr_dev(void* user, const void* v_dev, size_t n, void* out_dev) {
    Party* me = (Party*)user;
    LocalNet* net = me->net;
    if (n > net->gathered_elems) return -1;
    if (zk_memcpy_d2d(me->ctx, (char*)net->gathered + (size_t)me->id * n * 32, v_dev, n * 32)) return -1;
    if (zk_ctx_sync(me->ctx)) return -1;
    net->bar.wait();
    if (zk_fr_sum_parties_dev(me->ctx, net->gathered, (size_t)net->parties, n, out_dev)) return -1;
    if (zk_ctx_sync(me->ctx)) return -1;
    net->bar.wait();
    return 0;
}

// ---- MpcField<Fr, S> as the caller holds it ----
struct MF { bool shared = false; zk_fr v[2] = {FR_ZERO, FR_ZERO}; };   // Public: v[0]; Shared: v[0] = sh, v[1] = mac (SPDZ)
struct Env {                                                            // what `Net` and the share type are to the Rust code
    zk_mpc_field_layout fl;
    int lanes;
    bool leader;
    Party* me;
    zk_net_vtable vt;
    double t_net = 0;
};
static MF mf_load(const Env& e, const uint8_t* p) {
    MF m;
    m.shared = p[e.fl.off_tag] != e.fl.tag_public;
    if (!m.shared) memcpy(&m.v[0], p + e.fl.off_public, 32);
    else { memcpy(&m.v[0], p + e.fl.off_share, 32); if (e.lanes == 2) memcpy(&m.v[1], p + e.fl.off_mac, 32); }
    return m;
}
static void mf_store(const Env& e, uint8_t* p, const MF& m) {
    p[e.fl.off_tag] = m.shared ? e.fl.tag_shared : e.fl.tag_public;
    if (!m.shared) memcpy(p + e.fl.off_public, &m.v[0], 32);
    else { memcpy(p + e.fl.off_share, &m.v[0], 32); if (e.lanes == 2) memcpy(p + e.fl.off_mac, &m.v[1], 32); }
}
static MF mf_public(const zk_fr& x) { MF m; m.v[0] = x; return m; }
// FieldShare::shift (share/additive.rs:145-152, share/spdz.rs:214-218): the leader adds x; the MAC lane adds mac_share * x
static void mf_shift(const Env& e, MF& s, const zk_fr& x) {
    if (!e.leader) return;
    s.v[0] = fr_add(s.v[0], x);
    if (e.lanes == 2) s.v[1] = fr_add(s.v[1], x);
}
static MF mf_add(const Env& e, const MF& a, const MF& b) {             // wire/field.rs:339-362
    if (!a.shared && !b.shared) return mf_public(fr_add(a.v[0], b.v[0]));
    if (!a.shared) { MF t = b; mf_shift(e, t, a.v[0]); return t; }
    if (!b.shared) { MF t = a; mf_shift(e, t, b.v[0]); return t; }
    MF t = a;
    for (int l = 0; l < e.lanes; l++) t.v[l] = fr_add(a.v[l], b.v[l]);
    return t;
}
static MF mf_sub(const Env& e, const MF& a, const MF& b) {             // wire/field.rs:414-437
    if (!a.shared && !b.shared) return mf_public(fr_sub(a.v[0], b.v[0]));
    if (!a.shared) { MF t = b; for (int l = 0; l < e.lanes; l++) t.v[l] = fr_sub(FR_ZERO, b.v[l]); mf_shift(e, t, a.v[0]); return t; }
    if (!b.shared) { MF t = a; mf_shift(e, t, fr_sub(FR_ZERO, b.v[0])); return t; }
    MF t = a;
    for (int l = 0; l < e.lanes; l++) t.v[l] = fr_sub(a.v[l], b.v[l]);
    return t;
}
static MF mf_mul_public(const Env& e, const MF& a, const zk_fr& c) {    // wire/field.rs:463-476 with a Public right-hand side
    MF t = a;
    for (int l = 0; l < (a.shared ? e.lanes : 1); l++) t.v[l] = fr_mul(a.v[l], c);
    return t;
}
struct MVec {                                                           // Vec<MpcField<Fr, S>>
    const Env* e;
    std::vector<uint8_t> bytes;
    size_t n;
    MVec(const Env& env, size_t n_, const MF& fill) : e(&env), bytes(n_ * env.fl.stride, 0), n(n_) { for (size_t i = 0; i < n; i++) set(i, fill); }
    uint8_t* at(size_t i) { return bytes.data() + i * e->fl.stride; }
    const uint8_t* at(size_t i) const { return bytes.data() + i * e->fl.stride; }
    MF get(size_t i) const { return mf_load(*e, at(i)); }
    void set(size_t i, const MF& m) { mf_store(*e, at(i), m); }
};

struct Csr { std::vector<uint32_t> row_ptr, col; std::vector<zk_fr> coeff; };
// evaluate_constraint (src/groth16.rs:205-234) over MpcField: sum of assignment[index] (* coeff unless it is one)
static MF evaluate_constraint(const Env& e, const Csr& m, size_t row, const MVec& z, const zk_fr& one) {
    MF s;                                                               // R::zero() = Public(0)
    for (uint32_t k = m.row_ptr[row]; k < m.row_ptr[row + 1]; k++) {
        const zk_fr& c = m.coeff[k];
        const MF val = z.get(m.col[k]);
        s = mf_add(e, s, memcmp(&c, &one, sizeof one) == 0 ? val : mf_mul_public(e, val, c));
    }
    return s;
}

// ---- MpcGroup<G, S> in projective form, for the O(1) tail ----
struct G1T {
    typedef zk_g1_projective P; typedef zk_g1_affine A;
    static int add(const P* a, const P* b, P* o) { return zk_g1_add(a, b, o); }
    static int neg(const P* a, P* o) { return zk_g1_neg(a, o); }
    static int mul(const P* a, const zk_fr* k, P* o) { return zk_g1_mul(a, k, o); }
    static int from_affine(const A* a, P* o) { return zk_g1_from_affine(a, o); }
};
struct G2T {
    typedef zk_g2_projective P; typedef zk_g2_affine A;
    static int add(const P* a, const P* b, P* o) { return zk_g2_add(a, b, o); }
    static int neg(const P* a, P* o) { return zk_g2_neg(a, o); }
    static int mul(const P* a, const zk_fr* k, P* o) { return zk_g2_mul(a, k, o); }
    static int from_affine(const A* a, P* o) { return zk_g2_from_affine(a, o); }
};
static void die(const char* what) { fprintf(stderr, "%s\n", what); exit(1); }
#define GK(expr) do { if ((expr) != 0) die(#expr " failed"); } while (0)

template <class G> struct MG { bool shared = false; typename G::P v[2]; };
template <class G> static typename G::P g_zero() { typename G::A inf; memset(&inf, 0, sizeof inf); typename G::P z; GK(G::from_affine(&inf, &z)); return z; }
template <class G> static bool g_is_zero(const typename G::P& p) { const typename G::P z = g_zero<G>(); return memcmp(&p, &z, sizeof p) == 0; }
template <class G> static MG<G> mg_public(const typename G::A& a) { MG<G> m; GK(G::from_affine(&a, &m.v[0])); m.v[1] = m.v[0]; return m; }
// GroupShare::shift (share/additive.rs:510-515, share/spdz.rs:470-479)
template <class G> static void mg_shift(const Env& e, MG<G>& s, const typename G::P& x) {
    if (!e.leader) return;
    for (int l = 0; l < e.lanes; l++) GK(G::add(&s.v[l], &x, &s.v[l]));
}
template <class G> static MG<G> mg_add(const Env& e, const MG<G>& a, const MG<G>& b) {          // wire/group.rs AddAssign
    MG<G> t;
    if (!a.shared && !b.shared) { GK(G::add(&a.v[0], &b.v[0], &t.v[0])); t.v[1] = t.v[0]; return t; }
    if (!a.shared) { t = b; mg_shift(e, t, a.v[0]); return t; }
    if (!b.shared) { t = a; mg_shift(e, t, b.v[0]); return t; }
    t.shared = true;
    for (int l = 0; l < e.lanes; l++) GK(G::add(&a.v[l], &b.v[l], &t.v[l]));
    return t;
}
template <class G> static MG<G> mg_neg(const Env& e, const MG<G>& a) {
    MG<G> t = a;
    for (int l = 0; l < (a.shared ? e.lanes : 1); l++) GK(G::neg(&a.v[l], &t.v[l]));
    return t;
}
// ---- opens on the wire (MpcSerNet::broadcast under Reveal::reveal / open) ----
static void net_gather(Env& e, const void* mine, size_t len, std::vector<uint8_t>& all) {
    const double t0 = now_ms();
    all.resize((size_t)e.me->net->parties * len);
    if (e.me->net->parties == 1) memcpy(all.data(), mine, len);
    else if (all_gather_bytes(e.me, (const uint8_t*)mine, len, all.data())) die("all_gather_bytes failed");
    e.t_net += now_ms() - t0;
}
// AdditiveFieldShare / SpdzFieldShare::open (share/additive.rs:81-83, share/spdz.rs:121-131)
static zk_fr mf_open(Env& e, const MF& s) {
    std::vector<uint8_t> all;
    net_gather(e, &s.v[0], 32, all);
    zk_fr x = FR_ZERO;
    for (int p = 0; p < e.me->net->parties; p++) { zk_fr t; memcpy(&t, all.data() + 32 * p, 32); x = fr_add(x, t); }
    if (e.lanes == 2) {                                                 // dx_t = mac_share * x - mac; the sum must vanish
        const zk_fr dx = fr_sub(e.leader ? x : FR_ZERO, s.v[1]);
        net_gather(e, &dx, 32, all);
        zk_fr sum = FR_ZERO;
        for (int p = 0; p < e.me->net->parties; p++) { zk_fr t; memcpy(&t, all.data() + 32 * p, 32); sum = fr_add(sum, t); }
        if (!fr_is_zero(sum)) die("MAC check failed on a field open");
    }
    return x;
}
template <class G> static typename G::P mg_open(Env& e, const MG<G>& s) {
    typedef typename G::P P;
    std::vector<uint8_t> all;
    net_gather(e, &s.v[0], sizeof(P), all);
    P x = g_zero<G>();
    for (int p = 0; p < e.me->net->parties; p++) { P t; memcpy(&t, all.data() + sizeof(P) * p, sizeof(P)); GK(G::add(&x, &t, &x)); }
    if (e.lanes == 2) {
        P dx, nm;
        GK(G::neg(&s.v[1], &nm));
        if (e.leader) GK(G::add(&x, &nm, &dx)); else dx = nm;
        net_gather(e, &dx, sizeof(P), all);
        P sum = g_zero<G>();
        for (int p = 0; p < e.me->net->parties; p++) { P t; memcpy(&t, all.data() + sizeof(P) * p, sizeof(P)); GK(G::add(&sum, &t, &sum)); }
        if (!g_is_zero<G>(sum)) die("MAC check failed on a group open");
    }
    return x;
}
// MpcGroup *= MpcField (wire/group.rs:366-396): public x shared -> scale_pub_group; shared x shared -> GroupShare::scale
// (share/group.rs:72-111) with DummyGroupTripleSource (x = 0, y = leader ? 1 : 0, z = 0: wire/group.rs:45-54)
template <class G> static MG<G> mg_scalar_mul(Env& e, const MG<G>& a, const MF& k) {
    MG<G> t;
    if (!a.shared && !k.shared) { GK(G::mul(&a.v[0], &k.v[0], &t.v[0])); t.v[1] = t.v[0]; return t; }
    t.shared = true;
    if (!a.shared) { for (int l = 0; l < e.lanes; l++) GK(G::mul(&a.v[0], &k.v[l], &t.v[l])); return t; }       // scale_pub_group
    if (!k.shared) { for (int l = 0; l < e.lanes; l++) GK(G::mul(&a.v[l], &k.v[0], &t.v[l])); return t; }       // scale_pub_scalar
    const zk_fr one = fr(1);
    MF y; y.shared = true; y.v[0] = y.v[1] = e.leader ? one : FR_ZERO;
    const typename G::P sx = mg_open<G>(e, a);                                                  // open(s + x), x = 0
    const zk_fr oy = mf_open(e, mf_add(e, k, y));                                               // open(o + y)
    // out = z - scale_pub_group(sx, y) - x.scale_pub_scalar(oy), then shift(sx * oy)
    for (int l = 0; l < e.lanes; l++) {
        typename G::P sy;
        GK(G::mul(&sx, &y.v[l], &sy));
        GK(G::neg(&sy, &t.v[l]));
    }
    typename G::P sxoy;
    GK(G::mul(&sx, &oy, &sxoy));
    mg_shift(e, t, sxoy);
    return t;
}

// ---- layouts: how rustc lays the wrappers out; `tag last` moves the discriminant behind the payload and swaps its values ----
struct Layouts {
    zk_mpc_field_layout f;
    zk_mpc_group_layout g1, g2;
};
static Layouts make_layouts(bool spdz, bool tag_last) {
    Layouts L;
    const size_t fpay = spdz ? 64 : 32;
    L.f.stride = fpay + 8;
    L.f.off_tag = tag_last ? fpay : 0;
    const size_t fb = tag_last ? 0 : 8;
    L.f.off_public = fb; L.f.off_share = fb; L.f.off_mac = spdz ? fb + 32 : SIZE_MAX;
    L.f.tag_public = tag_last ? 1 : 0; L.f.tag_shared = tag_last ? 0 : 1;
    auto group = [&](zk_mpc_group_layout& g, size_t fe) {           // GroupAffine {x, y, infinity: bool} padded to 8: 2 fe + 8 bytes; a SPDZ share holds two
        const size_t aff = 2 * fe + 8, pay = spdz ? 2 * aff : aff, gb = tag_last ? 0 : 8;
        g.point.stride = pay + 8;
        g.point.off_x = gb; g.point.off_y = gb + fe; g.point.off_infinity = gb + 2 * fe;
        g.off_tag = tag_last ? pay : 0;
        g.tag_public = tag_last ? 1 : 0;
    };
    group(L.g1, 48);
    group(L.g2, 96);
    return L;
}
template <class A> static std::vector<uint8_t> wrap_public(const std::vector<A>& v, const zk_mpc_group_layout& g) {
    std::vector<uint8_t> o(v.size() * g.point.stride, 0xA5);            // (padding bytes are whatever: never read)
    for (size_t i = 0; i < v.size(); i++) {
        uint8_t* p = o.data() + i * g.point.stride;
        bool zero = true;
        const uint64_t* w = (const uint64_t*)&v[i];
        for (size_t k = 0; k < sizeof(A) / 8; k++) zero = zero && w[k] == 0;
        p[g.off_tag] = g.tag_public;
        p[g.point.off_infinity] = zero ? 1 : 0;
        memcpy(p + g.point.off_x, &v[i], sizeof(A) / 2);
        memcpy(p + g.point.off_y, (const char*)&v[i] + sizeof(A) / 2, sizeof(A) / 2);
    }
    return o;
}

struct Key {                                                            // ProvingKey<MpcPairingEngine>: every element Public
    std::vector<zk_g1_affine> a, b1, h, l;
    std::vector<zk_g2_affine> b2;
    zk_g1_affine alpha_g1, beta_g1, delta_g1;
    zk_g2_affine beta_g2, delta_g2;
};

int main(int argc, char** argv) {
    const unsigned log_d = argc > 1 ? (unsigned)atoi(argv[1]) : 10;
    const int proofs = argc > 2 ? atoi(argv[2]) : 3;
    const int P = argc > 3 ? atoi(argv[3]) : 3;
    const bool spdz = argc > 4 && std::string(argv[4]) == "spdz";
    const bool tag_last = argc > 5 && std::string(argv[5]) == "taglast";
    const bool trust = argc > 6 && std::string(argv[6]) == "trust";
    const bool sync2 = argc > 7 && std::string(argv[7]) == "sync2";
    if (log_d < 2 || log_d > 22 || proofs < 1 || P < 1 || P > 8) {
        fprintf(stderr, "usage: %s <log2 domain 2..22> <proofs> <parties 1..8> [additive|spdz] [tagfirst|taglast] [verify|trust]\n", argv[0]);
        return 2;
    }
    const size_t D = (size_t)1 << log_d, n = D - 2, ni = 2, nw = n + 1, m = ni + nw;
    const Layouts L = make_layouts(spdz, tag_last);
    const int lanes = spdz ? 2 : 1;

    // ---- set-up on one context (not the path under test): constraint system, key with fixed toxic waste, assignment ----
    zk_ctx* c0 = nullptr;
    if (zk_ctx_create(0, 0, 1, &c0)) die("zk_ctx_create failed (no GPU?)");
    field_init();
    const zk_fr one = fr(1);
    Csr A, B, Cm;
    auto idx = [&](size_t j) { return (uint32_t)(j <= n ? 2 + j : 1); };
    for (Csr* M : {&A, &B, &Cm}) { M->row_ptr.resize(n + 1); M->col.resize(n); M->coeff.assign(n, one); }
    for (size_t i = 0; i <= n; i++) A.row_ptr[i] = B.row_ptr[i] = Cm.row_ptr[i] = (uint32_t)i;
    for (size_t i = 0; i < n; i++) { A.col[i] = idx(i); B.col[i] = idx(i + 1); Cm.col[i] = idx(i + 2); }
    Key key;
    {
        zk_r1cs_host rh{n, ni, nw, A.row_ptr.data(), A.col.data(), A.coeff.data(), B.row_ptr.data(), B.col.data(), B.coeff.data(),
                        Cm.row_ptr.data(), Cm.col.data(), Cm.coeff.data()};
        zk_r1cs* r1cs = nullptr;
        GK(zk_r1cs_upload(c0, &rh, &r1cs));
        const zk_fr alpha = fr(2), beta = fr(3), gamma = fr(5), delta = fr(7), tau = fr(11), g1k = fr(1), g2k = fr(1);
        zk_pk* pk = nullptr;
        GK(zk_groth16_setup(c0, r1cs, &alpha, &beta, &gamma, &delta, &tau, &g1k, &g2k, &pk));
        key.a.resize(zk_pk_query_len(pk, 0)); key.b1.resize(zk_pk_query_len(pk, 1)); key.b2.resize(zk_pk_query_len(pk, 2));
        key.h.resize(zk_pk_query_len(pk, 3)); key.l.resize(zk_pk_query_len(pk, 4));
        GK(zk_pk_download_g1(c0, pk, 0, 0, key.a.size(), key.a.data()));
        GK(zk_pk_download_g1(c0, pk, 1, 0, key.b1.size(), key.b1.data()));
        GK(zk_pk_download_g2(c0, pk, 2, 0, key.b2.size(), key.b2.data()));
        GK(zk_pk_download_g1(c0, pk, 3, 0, key.h.size(), key.h.data()));
        GK(zk_pk_download_g1(c0, pk, 4, 0, key.l.size(), key.l.data()));
        GK(zk_pk_vk_g1(pk, 0, &key.alpha_g1)); GK(zk_pk_vk_g1(pk, 1, &key.beta_g1)); GK(zk_pk_vk_g1(pk, 2, &key.delta_g1));
        GK(zk_pk_vk_g2(pk, 0, &key.beta_g2)); GK(zk_pk_vk_g2(pk, 1, &key.delta_g2));
        GK(zk_pk_free(c0, pk));
        GK(zk_r1cs_free(c0, r1cs));
    }
    if (key.a.size() != m || key.b1.size() != m || key.b2.size() != m || key.l.size() != nw) die("unexpected key shape");
    std::vector<zk_fr> z(m);                                            // instance [1, w_{n+1}] then witness w_0 .. w_n
    {
        std::vector<zk_fr> w(n + 2);
        w[0] = fr(3); w[1] = fr(5);
        for (size_t i = 0; i < n; i++) w[i + 2] = fr_mul(w[i], w[i + 1]);
        z[0] = one; z[1] = w[n + 1];
        for (size_t j = 0; j <= n; j++) z[2 + j] = w[j];
    }
    // additive shares of everything but the constant: parties 1.. draw theirs, party 0 holds the rest (king_share); r and s likewise
    std::vector<std::vector<zk_fr>> zs(P, std::vector<zk_fr>(m, FR_ZERO));
    std::vector<zk_fr> rs(P), ss(P);
    zk_fr r_tot = FR_ZERO, s_tot = FR_ZERO;
    zs[0] = z;
    for (int p = 0; p < P; p++) {
        uint8_t seed[32];
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)(37 * p + i + 1);
        zk_rng* g = nullptr;
        GK(zk_rng_from_seed(seed, 20, &g));
        std::vector<zk_fr> draw(m);
        zk_rng_fill_fr(g, draw.data(), m);
        zk_rng_fill_fr(g, &rs[p], 1);
        zk_rng_fill_fr(g, &ss[p], 1);
        zk_rng_free(g);
        r_tot = fr_add(r_tot, rs[p]);
        s_tot = fr_add(s_tot, ss[p]);
        if (p == 0) continue;
        for (size_t i = 1; i < m; i++) { zs[p][i] = draw[i]; zs[0][i] = fr_sub(zs[0][i], draw[i]); }
    }
    GK(zk_ctx_destroy(c0));

    LocalNet net(P);
    std::vector<std::vector<std::string>> lines(P);                     // per party, per proof
    std::vector<std::vector<std::vector<uint8_t>>> out(P, std::vector<std::vector<uint8_t>>(proofs, std::vector<uint8_t>(192)));
    std::vector<std::vector<double>> lib_ms(P, std::vector<double>(proofs, 0));
    std::vector<std::string> cache_line(P);
    std::vector<std::thread> th;
    for (int p = 0; p < P; p++)
        th.emplace_back([&, p] {
            zk_ctx* ctx = nullptr;
            if (zk_ctx_create(0, p, P, &ctx)) die("zk_ctx_create failed");
            auto CK = [&](int rc, const char* what) { if (rc) { fprintf(stderr, "party %d: %s -> %d: %s\n", p, what, rc, zk_last_error(ctx)); exit(1); } };
#define CKX(expr) CK((expr), #expr)
            if (trust) CKX(zk_bases_cache_trust(ctx, 1));
            if (P > 1) CKX(zk_msm_speculate(ctx, 0));        // P contexts on ONE device here: the MSMs a context starts ahead would only take the chip from the others
            Party me{&net, p, ctx};
            Env e;
            e.fl = L.f; e.lanes = lanes; e.leader = p == 0; e.me = &me;
            e.vt = zk_net_vtable{&me, all_gather_bytes, open_sum_fr_dev};
            if (p == 0) {
                net.gathered_elems = D;
                if (zk_dev_alloc(ctx, (size_t)P * D * 32, &net.gathered)) die("zk_dev_alloc failed");
            }
            // the party's ProvingKey<MpcPairingEngine> (host memory, wrapper layout) and its assignment: instance[0] = Public(1), the rest Shared
            const std::vector<uint8_t> a_query = wrap_public(key.a, L.g1), b_g1_query = wrap_public(key.b1, L.g1), h_query = wrap_public(key.h, L.g1),
                                       l_query = wrap_public(key.l, L.g1), b_g2_query = wrap_public(key.b2, L.g2);
            const size_t S1 = L.g1.point.stride, S2 = L.g2.point.stride;
            MVec full_assignment(e, m, MF());
            full_assignment.set(0, mf_public(one));
            for (size_t i = 1; i < m; i++) { MF s; s.shared = true; s.v[0] = s.v[1] = zs[p][i]; full_assignment.set(i, s); }   // from_add_shared: mac = share (key 1)
            MF r, s;
            r.shared = s.shared = true;
            r.v[0] = r.v[1] = rs[p]; s.v[0] = s.v[1] = ss[p];
            net.bar.wait();

            for (int it = 0; it < proofs; it++) {
                double t_fft = 0, t_bp = 0, t_div = 0, t_mat = 0, t_sub = 0, t_tail = 0, t_msm[5] = {0, 0, 0, 0, 0};
                e.t_net = 0;
                net.bar.wait();
                const double t0 = now_ms();
                double t = t0;
                auto lap = [&](double& acc) { const double u = now_ms(); acc += u - t; t = u; };
                auto fft = [&](MVec& v, int inverse, int coset) { CKX(zk_mpc_fft_in_place(ctx, v.bytes.data(), v.n, &L.f, log_d, inverse, coset)); };
                // ---- R1CStoQAP::witness_map (src/groth16.rs:240-306) ----
                MVec a(e, D, MF()), b(e, D, MF());                               // vec![zero; domain_size]: Public(0)
                for (size_t i = 0; i < n; i++) { a.set(i, evaluate_constraint(e, A, i, full_assignment, one)); b.set(i, evaluate_constraint(e, B, i, full_assignment, one)); }
                for (size_t i = 0; i < ni; i++) a.set(n + i, full_assignment.get(i));   // a[start..end].clone_from_slice(&full_assignment[..num_inputs])
                lap(t_mat);
                fft(a, 1, 0);                                                     // domain.ifft_in_place(&mut a)
                fft(b, 1, 0);
                fft(a, 0, 1);                                                     // domain.coset_fft_in_place(&mut a)
                fft(b, 0, 1);
                lap(t_fft);
                MVec ab(a);                                                       // let mut ab = a.clone()
                lap(t_sub);
                uint64_t sent = 0;
                CKX(zk_mpc_batch_product_in_place(ctx, ab.bytes.data(), b.bytes.data(), D, &L.f, nullptr, P > 1 ? &e.vt : nullptr, &sent));   // F::batch_product_in_place(&mut ab, &b)
                lap(t_bp);
                MVec c(e, D, MF());
                for (size_t i = 0; i < n; i++) c.set(i, evaluate_constraint(e, Cm, i, full_assignment, one));
                lap(t_mat);
                fft(c, 1, 0);
                fft(c, 0, 1);
                lap(t_fft);
                for (size_t i = 0; i < D; i++) ab.set(i, mf_sub(e, ab.get(i), c.get(i)));      // ab_i -= c_i
                lap(t_sub);
                CKX(zk_mpc_divide_by_vanishing_on_coset_in_place(ctx, ab.bytes.data(), &L.f, log_d));
                lap(t_div);
                fft(ab, 1, 1);                                                    // domain.coset_ifft_in_place(&mut ab)
                lap(t_fft);
                const MVec& h = ab;
                // ---- create_proof (src/groth16.rs:104-176) ----
                auto msm1 = [&](const std::vector<uint8_t>& q, size_t skip, const uint8_t* sc, size_t ns) {
                    MG<G1T> acc;
                    int pub = 0;
                    CKX(zk_mpc_msm_g1(ctx, q.data() + skip * S1, q.size() / S1 - skip, &L.g1, sc, ns, &L.f, acc.v, &pub));
                    acc.shared = true;                                            // Shared(from_public(r)) when every scalar was public (wire/pairing.rs:726-741)
                    if (pub && !e.leader) acc.v[0] = acc.v[1] = g_zero<G1T>();
                    return acc;
                };
                const MG<G1T> h_acc = msm1(h_query, 0, h.bytes.data(), h.n);      // multi_scalar_mul(&pk.h_query, &h)
                lap(t_msm[0]);
                const MG<G1T> l_aux_acc = msm1(l_query, 0, full_assignment.at(ni), nw);      // (&pk.l_query, &prover.witness_assignment)
                lap(t_msm[1]);
                const MG<G1T> delta_g1 = mg_public<G1T>(key.delta_g1);
                const MG<G2T> delta_g2 = mg_public<G2T>(key.delta_g2);
                const MG<G1T> r_s_delta_g1 = mg_scalar_mul<G1T>(e, mg_scalar_mul<G1T>(e, delta_g1, r), s);   // delta_g1 * r * s: the first GroupShare::scale
                lap(t_tail);
                const uint8_t* assignment = full_assignment.at(1);               // instance[1..] ++ witness (contiguous in full_assignment)
                const size_t na = m - 1;
                const MG<G1T> r_g1 = mg_scalar_mul<G1T>(e, delta_g1, r);
                lap(t_tail);
                // calculate_coeff(initial, query, vk_param, assignment) = initial + query[0] + msm(query[1..], assignment) + vk_param
                const MG<G1T> acc_a = msm1(a_query, 1, assignment, na);
                lap(t_msm[2]);
                MG<G1T> g_a = mg_add<G1T>(e, mg_add<G1T>(e, mg_add<G1T>(e, r_g1, mg_public<G1T>(key.a[0])), acc_a), mg_public<G1T>(key.alpha_g1));
                const MG<G1T> s_g_a = mg_scalar_mul<G1T>(e, g_a, s);              // the second scale
                const MG<G1T> s_g1 = mg_scalar_mul<G1T>(e, delta_g1, s);
                lap(t_tail);
                const MG<G1T> acc_b1 = msm1(b_g1_query, 1, assignment, na);
                lap(t_msm[3]);
                const MG<G1T> g1_b = mg_add<G1T>(e, mg_add<G1T>(e, mg_add<G1T>(e, s_g1, mg_public<G1T>(key.b1[0])), acc_b1), mg_public<G1T>(key.beta_g1));
                const MG<G2T> s_g2 = mg_scalar_mul<G2T>(e, delta_g2, s);
                lap(t_tail);
                MG<G2T> acc_b2;
                {
                    int pub = 0;
                    CKX(zk_mpc_msm_g2(ctx, b_g2_query.data() + S2, b_g2_query.size() / S2 - 1, &L.g2, assignment, na, &L.f, acc_b2.v, &pub));
                    acc_b2.shared = true;
                    if (pub && !e.leader) acc_b2.v[0] = acc_b2.v[1] = g_zero<G2T>();
                }
                lap(t_msm[4]);
                const MG<G2T> g2_b = mg_add<G2T>(e, mg_add<G2T>(e, mg_add<G2T>(e, s_g2, mg_public<G2T>(key.b2[0])), acc_b2), mg_public<G2T>(key.beta_g2));
                const MG<G1T> r_g1_b = mg_scalar_mul<G1T>(e, g1_b, r);            // the third scale
                MG<G1T> g_c = mg_add<G1T>(e, s_g_a, r_g1_b);                      // g_c = s_g_a + r_g1_b - r_s_delta_g1 + l_aux_acc + h_acc
                g_c = mg_add<G1T>(e, g_c, mg_neg<G1T>(e, r_s_delta_g1));
                g_c = mg_add<G1T>(e, g_c, l_aux_acc);
                g_c = mg_add<G1T>(e, g_c, h_acc);
                // Proof::reveal (arkworks/groth16/src/reveal.rs:7-10): a, b, c
                const zk_g1_projective pa = mg_open<G1T>(e, g_a);
                const zk_g2_projective pb = mg_open<G2T>(e, g2_b);
                const zk_g1_projective pc = mg_open<G1T>(e, g_c);
                uint8_t* proof = out[p][it].data();
                GK(zk_g1_serialize(&pa, proof)); GK(zk_g2_serialize(&pb, proof + 48)); GK(zk_g1_serialize(&pc, proof + 144));
                lap(t_tail);
                const double total = now_ms() - t0;
                const double lib = t_fft + t_bp + t_div + t_msm[0] + t_msm[1] + t_msm[2] + t_msm[3] + t_msm[4];
                lib_ms[p][it] = lib;
                char buf[1024];
                snprintf(buf, sizeof buf,
                         "\"ms\": {\"total\": %.3f, \"lib\": %.3f, \"fft\": %.3f, \"batch_product\": %.3f, \"divide\": %.3f, \"msm_h\": %.3f, \"msm_l\": %.3f, "
                         "\"msm_a\": %.3f, \"msm_b1\": %.3f, \"msm_b2\": %.3f, \"tail_and_small_opens\": %.3f, \"caller_matvec\": %.3f, \"caller_sub_clone\": %.3f}, "
                         "\"beaver_bytes_sent\": %llu",
                         total, lib, t_fft, t_bp, t_div, t_msm[0], t_msm[1], t_msm[2], t_msm[3], t_msm[4], t_tail, t_mat, t_sub, (unsigned long long)sent);
                lines[p].push_back(buf);
                if (sync2 && it == 1) CKX(zk_bases_cache_sync(ctx));
            }
            net.bar.wait();
            uint64_t st[10], st2[4];
            CKX(zk_bases_cache_sync(ctx));
            CKX(zk_bases_cache_stats(ctx, st));
            CKX(zk_bases_cache_stats2(ctx, st2));
            char buf[768];
            snprintf(buf, sizeof buf,
                     "{\"hits\": %llu, \"misses\": %llu, \"evictions\": %llu, \"replaced\": %llu, \"uncached\": %llu, \"entries\": %llu, "
                     "\"with_window_multiples\": %llu, \"resident_bytes\": %llu, \"uploaded_bytes\": %llu, \"budget\": %llu, \"verified\": %llu, "
                     "\"verified_bytes\": %llu, \"builds\": %llu}",
                     (unsigned long long)st[0], (unsigned long long)st[1], (unsigned long long)st[2], (unsigned long long)st[3], (unsigned long long)st[4],
                     (unsigned long long)st[5], (unsigned long long)st[6], (unsigned long long)st[7], (unsigned long long)st[8], (unsigned long long)st[9],
                     (unsigned long long)st2[0], (unsigned long long)st2[1], (unsigned long long)st2[2]);
            cache_line[p] = buf;
            net.bar.wait();
            if (p == 0 && net.gathered) zk_dev_free(ctx, net.gathered);
            zk_ctx_destroy(ctx);
        });
    for (auto& t : th) t.join();
    for (int it = 0; it < proofs; it++) {
        for (int p = 1; p < P; p++)
            if (out[p][it] != out[0][it]) die("the parties ended with different proof bytes");
        double mx = 0;
        for (int p = 0; p < P; p++) mx = lib_ms[p][it] > mx ? lib_ms[p][it] : mx;
        printf("{\"proof\": \"");
        for (int i = 0; i < 192; i++) printf("%02x", out[0][it][i]);
        printf("\", %s, \"ms_max_lib\": %.3f}\n", lines[0][it].c_str(), mx);
    }
    uint64_t rc[4], sc[4];
    zk_fr_to_canonical(&r_tot, rc);
    zk_fr_to_canonical(&s_tot, sc);
    printf("{\"r\": \"");
    for (int i = 0; i < 32; i++) printf("%02x", ((const uint8_t*)rc)[i]);
    printf("\", \"s\": \"");
    for (int i = 0; i < 32; i++) printf("%02x", ((const uint8_t*)sc)[i]);
    printf("\", \"cache\": %s, \"log_d\": %u, \"constraints\": %zu, \"parties\": %d, \"shares\": \"%s\", \"tag\": \"%s\", \"hits\": \"%s\", "
           "\"element_bytes\": %zu, \"g1_base_bytes\": %zu, \"g2_base_bytes\": %zu}\n",
           cache_line[0].c_str(), log_d, n, P, spdz ? "spdz" : "additive", tag_last ? "last" : "first", trust ? "trusted" : "verified",
           L.f.stride, L.g1.point.stride, L.g2.point.stride);
    return 0;
}
