// host_collab_groth16.cpp -- the collaborative Groth16 prover driven from a compiled host with the HOST'S OWN transport:
// P parties as threads of one process (the reference's LocalTestNet shape, mpc-net/src/multi.rs:357-453), every party with
// its own context, key and share of the assignment, all of them calling zk_groth16_prove_shared with a zk_net_vtable whose two
// callbacks are implemented here in C++ over shared memory and a barrier:
//   all_gather_bytes  = MpcNet::broadcast_bytes (mpc-net/src/lib.rs:60-64): every party's bytes, ordered by party id
//   open_sum_fr_dev   = AdditiveFieldShare::batch_open (mpc-algebra/src/share/additive.rs:124-131) on device vectors: the
//                       parties copy their vectors side by side into one device buffer and each sums them (zk_fr_sum_parties_dev)
// Every party must end with the same 192 bytes, and they must be the bytes of the plain prover on the summed inputs
// (zk_groth16_prove), which the program checks itself; tests/test_host_example.py checks them against the oracle's prediction.
//   host_collab_groth16 [parties = 3] [constraints = 1000]
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "zkmpc_hip.h"

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

struct LocalNet {                 // what the parties share
    int parties;
    Barrier bar;
    std::vector<uint8_t> host;    // all_gather staging: parties x len
    void* gathered = nullptr;     // device: parties x n field elements
    size_t gathered_elems = 0;
    explicit LocalNet(int p) : parties(p), bar(p) {}
};
struct Party { LocalNet* net; int id; zk_ctx* ctx; };

static int all_gather_bytes(void* user, const uint8_t* mine, size_t len, uint8_t* out_all) {
    Party* me = (Party*)user;
    LocalNet* net = me->net;
    if (me->id == 0) net->host.assign((size_t)net->parties * len, 0);
    net->bar.wait();
    memcpy(net->host.data() + (size_t)me->id * len, mine, len);
    net->bar.wait();
    memcpy(out_all, net->host.data(), (size_t)net->parties * len);
    net->bar.wait();              // nobody resizes the staging while another party still reads it
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

static zk_fr fr(uint64_t v) {
    uint64_t c[4] = {v, 0, 0, 0};
    zk_fr o;
    zk_fr_from_canonical(c, &o);
    return o;
}
static void hex(const char* tag, const uint8_t* b, size_t n) {
    printf("%s ", tag);
    for (size_t i = 0; i < n; i++) printf("%02x", b[i]);
    printf("\n");
}

int main(int argc, char** argv) {
    const int P = argc > 1 ? atoi(argv[1]) : 3;
    const size_t n = argc > 2 ? (size_t)atoll(argv[2]) : 1000;
    if (P < 1 || P > 16 || n < 1) { fprintf(stderr, "usage: host_collab_groth16 [parties 1..16] [constraints]\n"); return 2; }
    const size_t m = n + 3;                                    // full assignment: 1, pub, w_0 .. w_n
    // the witness of the mul-chain circuit (w_i w_{i+1} = w_{i+2}; the last product is the public input), on the host
    std::vector<zk_fr> z(m);
    {
        std::vector<zk_fr> w(n + 2);
        w[0] = fr(3); w[1] = fr(5);
        for (size_t i = 0; i < n; i++) zk_fr_mul(&w[i], &w[i + 1], &w[i + 2]);
        z[0] = fr(1); z[1] = w[n + 1];
        for (size_t j = 0; j <= n; j++) z[2 + j] = w[j];
    }
    // additive shares: parties 1.. draw theirs, party 0 holds the rest; the instance part sits on party 0 (from_public)
    std::vector<std::vector<zk_fr>> zs(P, std::vector<zk_fr>(m, fr(0)));
    std::vector<zk_fr> rs(P), ss(P);
    zk_fr r_tot = fr(0), s_tot = fr(0);
    zs[0] = z;
    for (int p = 0; p < P; p++) {
        uint8_t seed[32];
        for (int i = 0; i < 32; i++) seed[i] = (uint8_t)(31 * p + i);
        zk_rng* g = nullptr;
        if (zk_rng_from_seed(seed, 20, &g)) return 1;
        std::vector<zk_fr> draw(m);
        zk_rng_fill_fr(g, draw.data(), m);
        zk_rng_fill_fr(g, &rs[p], 1);
        zk_rng_fill_fr(g, &ss[p], 1);
        zk_rng_free(g);
        zk_fr_add(&r_tot, &rs[p], &r_tot);
        zk_fr_add(&s_tot, &ss[p], &s_tot);
        if (p == 0) continue;
        for (size_t i = 2; i < m; i++) { zs[p][i] = draw[i]; zk_fr_sub(&zs[0][i], &draw[i], &zs[0][i]); }
    }
    LocalNet net(P);
    std::vector<std::vector<uint8_t>> proofs(P, std::vector<uint8_t>(192));
    std::vector<int> rc(P, 0);
    std::vector<std::string> err(P);
    const zk_fr alpha = fr(2), beta = fr(3), gamma = fr(5), delta = fr(7), tau = fr(11), one = fr(1);
    std::vector<std::thread> th;
    for (int p = 0; p < P; p++)
        th.emplace_back([&, p] {
            zk_ctx* ctx = nullptr;
            zk_r1cs* r1cs = nullptr;
            zk_pk* pk = nullptr;
            void* zd = nullptr;
            auto fail = [&](int code) { rc[p] = code; if (ctx) err[p] = zk_last_error(ctx); };
            int e = zk_ctx_create(0, p, P, &ctx);
            if (!e) e = zk_r1cs_mul_chain(ctx, n, &r1cs);
            if (!e) e = zk_groth16_setup(ctx, r1cs, &alpha, &beta, &gamma, &delta, &tau, &one, &one, &pk);
            if (!e) e = zk_dev_alloc(ctx, m * 32, &zd);
            if (!e) e = zk_memcpy_h2d(ctx, zd, zs[p].data(), m * 32);
            if (!e && p == 0) {                                   // the opens are over the QAP domain: D elements per party
                const size_t D = (size_t)1 << zk_r1cs_domain_log(r1cs);
                net.gathered_elems = D;
                e = zk_dev_alloc(ctx, (size_t)P * D * 32, &net.gathered);
            }
            if (e) fail(e);
            net.bar.wait();                                       // everybody is set up (or has failed)
            bool all_ok = true;
            for (int q = 0; q < P; q++) all_ok = all_ok && rc[q] == 0;
            if (all_ok) {
                Party me{&net, p, ctx};
                zk_net_vtable vt{&me, all_gather_bytes, open_sum_fr_dev};
                uint64_t sent = 0;
                e = zk_groth16_prove_shared(ctx, pk, r1cs, zd, &rs[p], &ss[p], nullptr, nullptr, nullptr, P > 1 ? &vt : nullptr,
                                            proofs[p].data(), &sent);
                if (e) fail(e);
                else if (p == 0) printf("party 0 sent %llu bytes\n", (unsigned long long)sent);
                if (!e && p == 0) {                               // the plain prover on the summed inputs: the same bytes
                    uint8_t local[192];
                    e = zk_groth16_prove(ctx, pk, r1cs, z.data(), &r_tot, &s_tot, local);
                    if (e) fail(e);
                    else if (memcmp(local, proofs[0].data(), 192)) { rc[p] = -100; err[p] = "collaborative and local proofs differ"; }
                }
            }
            net.bar.wait();
            if (p == 0 && net.gathered) zk_dev_free(ctx, net.gathered);
            if (zd) zk_dev_free(ctx, zd);
            if (pk) zk_pk_free(ctx, pk);
            if (r1cs) zk_r1cs_free(ctx, r1cs);
            if (ctx) zk_ctx_destroy(ctx);
        });
    for (auto& t : th) t.join();
    for (int p = 0; p < P; p++)
        if (rc[p]) { fprintf(stderr, "party %d: error %d: %s\n", p, rc[p], err[p].c_str()); return 1; }
    for (int p = 1; p < P; p++)
        if (proofs[p] != proofs[0]) { fprintf(stderr, "party %d ended with different proof bytes\n", p); return 1; }
    uint64_t c[4];
    zk_fr_to_canonical(&r_tot, c); hex("r", (const uint8_t*)c, 32);          // little-endian canonical integers
    zk_fr_to_canonical(&s_tot, c); hex("s", (const uint8_t*)c, 32);
    hex("proof", proofs[0].data(), 192);
    return 0;
}
