// sharednet.hpp -- what the collaborative provers (groth16_shared.hip: zk_groth16_prove_shared[_spdz]; marlin_prove.hip:
// zk_marlin_prove_shared[_spdz]) share: the party's transport behind zk_net_vtable, the MAC-checked vector open and the vector
// Beaver product.  Replaces MpcSerNet::broadcast (mpc-algebra/src/channel.rs:12-28), AdditiveFieldShare / SpdzFieldShare::batch_open
// (share/additive.rs:124-131, share/spdz.rs:177-196) and FieldShare::batch_mul (share/field.rs:97-129) on device vectors.
#pragma once
#include "../../include/zkmpc_hip.h"
#include "ctx.hpp"
#include <string.h>
#include <vector>

struct ZkSharedNet {
    zk_ctx* ctx;
    const zk_net_vtable* vt;
    size_t bytes = 0;                       // payload bytes this party contributed to opens
    int parties() const { return ctx->n_parties; }
    bool leader() const { return ctx->party_id == 0; }
    // all[p * len ..] = party p's bytes (MpcNet::broadcast_bytes); a single party needs no transport
    int gather(const uint8_t* mine, size_t len, std::vector<uint8_t>& all) {
        all.resize((size_t)parties() * len);
        bytes += len;
        if (parties() == 1) { memcpy(all.data(), mine, len); return ZK_OK; }
        if (!vt || !vt->all_gather_bytes) ZK_FAIL(ctx, ZK_ERR_ARG, "collaborative prover: several parties need zk_net_vtable::all_gather_bytes");
        if (vt->all_gather_bytes(vt->user, mine, len, all.data()) != 0) ZK_FAIL(ctx, ZK_ERR_STATE, "collaborative prover: all_gather_bytes failed");
        return ZK_OK;
    }
    // out = sum over parties of v (n field elements on the device); out may alias v
    int open_vec(const void* v, size_t n, void* out) {
        bytes += n * 32;
        if (vt && vt->open_sum_fr_dev) {
            ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));             // the callback may use its own stream
            if (vt->open_sum_fr_dev(vt->user, v, n, out) != 0) ZK_FAIL(ctx, ZK_ERR_STATE, "collaborative prover: open_sum_fr_dev callback failed");
            return ZK_OK;
        }
        if (parties() == 1) {
            if (out != v) ZK_HIP(ctx, hipMemcpyAsync(out, v, n * 32, hipMemcpyDeviceToDevice, ctx->stream));
            return ZK_OK;
        }
        return zk_open_sum_fr_dev(ctx, v, n, out);
    }
};

// groth16_shared.hip
// SpdzFieldShare::batch_open on a device vector (key alpha = 1 held by the leader): out = open(share lane); then every party
// publishes [leader ? out : 0] - mac (into dx, n elements of scratch) and the sum must vanish -- otherwise ZK_ERR_MAC.
int zk_shared_spdz_open_vec(ZkSharedNet& nt, const void* sh, const void* mac, size_t n, void* out, void* dx);
// FieldShare::batch_mul on device vectors, lanes = 1 (additive) or 2 (SPDZ: share lane, MAC lane): out[l] = shares of x * y.
// tx / ty / tz: this party's Beaver triple shares per lane, or all NULL for DummyFieldTripleSource (the leader holds 1 in every
// lane).  out[l] may alias x[l].  scratch: 2 * lanes + 3 vectors of n elements, named by `tag` in the context's arena.
int zk_shared_beaver_mul(ZkSharedNet& nt, int lanes, const void* const x[2], const void* const y[2], void* const out[2], size_t n,
                         const void* const tx[2], const void* const ty[2], const void* const tz[2], const char* tag);
