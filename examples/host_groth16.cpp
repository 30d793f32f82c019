// host_groth16.cpp -- a compiled-language host above the C ABI: the flow of the reference's own Groth16 test
// (arkworks/groth16/src/test.rs:14-77: MySillyCircuit, a * b = c enforced num_constraints times; generate_random_parameters,
// create_random_proof, verify_proof) written against include/zkmpc_hip.h only -- what a Rust host does through the `extern "C"`
// block of INTEGRATION.md.  Toxic waste and prover randomness are fixed small integers so that the known-trapdoor prediction of
// the oracle (tests/test_host_example.py) names the exact 192 bytes this program must print.
//
// Build and run (tests/test_host_example.py does exactly this):
//   g++ -std=c++17 -I include examples/host_groth16.cpp -L zk-mpc_amd/lib -lzkmpc_hip -Wl,-rpath,$PWD/zk-mpc_amd/lib
//       -Wl,--allow-shlib-undefined -o /tmp/host_groth16 && /tmp/host_groth16 [num_constraints]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "zkmpc_hip.h"

static zk_ctx* CTX = nullptr;
#define CK(expr)                                                                                        \
    do {                                                                                                \
        int rc_ = (expr);                                                                               \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, CTX ? zk_last_error(CTX) : "(no context: no GPU?)");  \
                        return 1; }                                                                     \
    } while (0)

static zk_fr fr(uint64_t v) {          // Fr::from(v): canonical integer -> Montgomery words (from_repr)
    uint64_t c[4] = {v, 0, 0, 0};
    zk_fr o;
    zk_fr_from_canonical(c, &o);
    return o;
}

int main(int argc, char** argv) {
    const size_t nc = argc > 1 ? (size_t)atoll(argv[1]) : 100;
    CK(zk_ctx_create(0, 0, 1, &CTX));
    // MySillyCircuit::generate_constraints: instance = [1, c], witness = [a, b]; a * b = c, nc times
    std::vector<uint32_t> row_ptr(nc + 1), col_a(nc, 2), col_b(nc, 3), col_c(nc, 1);
    std::vector<zk_fr> ones(nc, fr(1));
    for (size_t i = 0; i <= nc; i++) row_ptr[i] = (uint32_t)i;
    zk_r1cs_host host{nc, 2, 2, row_ptr.data(), col_a.data(), ones.data(), row_ptr.data(), col_b.data(), ones.data(),
                      row_ptr.data(), col_c.data(), ones.data()};
    zk_r1cs* r1cs = nullptr;
    CK(zk_r1cs_upload(CTX, &host, &r1cs));
    // generate_parameters with explicit toxic waste (generator.rs:44-231)
    const zk_fr alpha = fr(2), beta = fr(3), gamma = fr(5), delta = fr(7), tau = fr(11), g1k = fr(1), g2k = fr(1);
    zk_pk* pk = nullptr;
    CK(zk_groth16_setup(CTX, r1cs, &alpha, &beta, &gamma, &delta, &tau, &g1k, &g2k, &pk));
    // create_proof: full assignment = instance (1, c) then witness (a, b)
    const uint64_t a = 3, b = 5;
    const zk_fr z[4] = {fr(1), fr(a * b), fr(a), fr(b)};
    const zk_fr r = fr(13), s = fr(17);
    uint8_t proof[192];
    CK(zk_groth16_prove(CTX, pk, r1cs, z, &r, &s, proof));
    // a second proof of the same statement through the queued entry point must be the same bytes
    uint8_t again[192];
    CK(zk_groth16_prove_queued(CTX, pk, r1cs, z, &r, &s, nullptr, again));
    for (int i = 0; i < 192; i++)
        if (proof[i] != again[i]) { fprintf(stderr, "zk_groth16_prove and zk_groth16_prove_queued differ\n"); return 1; }
    printf("proof ");
    for (int i = 0; i < 192; i++) printf("%02x", proof[i]);
    printf("\n");
    CK(zk_pk_free(CTX, pk));
    CK(zk_r1cs_free(CTX, r1cs));
    CK(zk_ctx_destroy(CTX));
    return 0;
}
