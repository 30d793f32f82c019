// groth16_shared.hip -- create_proof over additive / SPDZ shares as one C-ABI call.
#include "../../include/zkmpc_hip.h"
#include "groth16_int.hpp"
#include "sharednet.hpp"

using namespace zk;

// ---- the collaborative prover as ONE entry point ---------------------------------------------------------------------------------
// create_proof::<MpcPairingEngine, C> over additive shares (src/groth16.rs:68-183): what mpc.py::Party.create_proof_shared
// sequences from ~40 calls, for a host that cannot call Python.  The transport is the caller's (zk_net_vtable: the reference's
// MpcNet::broadcast_bytes on its TCP mesh, mpc-net/src/lib.rs:60-64) for the small opens; the two vector opens of the Beaver
// product go through the vtable's open_sum_fr_dev or, when that is NULL, through the context's own RCCL communicator
// (zk_open_sum_fr_dev).  Opens happen in the fused order of create_proof_shared: the nine small opens of the three
// GroupShare::scale calls (share/group.rs:72-111, DummyGroupTripleSource) and of Proof::reveal travel in two exchanges; every
// opened value is the reference's, and so are the 192 bytes.
namespace {

struct FrK { uint32_t l[9]; };
__device__ __forceinline__ Fr frk(const FrK& k) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = k.l[i];
    return r;
}
FrK to_frk(const Fr& a) {
    FrK k;
    for (int i = 0; i < 9; i++) k.l[i] = a.l[i];
    return k;
}
// the 256-bit word of a field element is below the modulus (is_valid, ff/src/fields/macros.rs:255-260): what arrives from a peer
bool fr_abi_valid(const uint64_t l[4]) {
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

__global__ void __launch_bounds__(256) k_vec_add_const(const void* a, FrK k, void* out, size_t n) {
    const Fr kk = frk(k);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        fr_store(out, i, fr_add(fr_load(a, i), kk));
}

}  // namespace

namespace {

__global__ void __launch_bounds__(256) k_vec_neg(const void* a, void* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        fr_store(out, i, fr_sub(fp_zero<FrParams>(), fr_load(a, i)));
}

}  // namespace

int zk_shared_spdz_open_vec(ZkSharedNet& nt, const void* sh, const void* mac, size_t n, void* out, void* dx) {
    zk_ctx* ctx = nt.ctx;
    ZK_TRY(nt.open_vec(sh, n, out));
    if (ctx->party_id == 0) {
        ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_SUB, out, mac, dx, n));
    } else {
        hipLaunchKernelGGL(k_vec_neg, zk_grid(n, 256), 256, 0, ctx->stream, mac, dx, n);
        ZK_HIP(ctx, hipGetLastError());
    }
    ZK_TRY(nt.open_vec(dx, n, dx));
    int zero = 0;
    ZK_TRY(zk_fr_vec_is_zero_dev(ctx, dx, n, &zero));
    if (!zero) ZK_FAIL(ctx, ZK_ERR_MAC, "SPDZ MAC check failed on a vector open");
    return ZK_OK;
}

int zk_shared_beaver_mul(ZkSharedNet& nt, int lanes, const void* const x[2], const void* const y[2], void* const out[2], size_t n,
                         const void* const tx[2], const void* const ty[2], const void* const tz[2], const char* tag) {
    zk_ctx* ctx = nt.ctx;
    const bool dummy = !tx[0] && !ty[0] && !tz[0];
    for (int l = 0; l < lanes; l++)
        if (dummy ? (tx[l] || ty[l] || tz[l]) : (!tx[l] || !ty[l] || !tz[l])) ZK_FAIL(ctx, ZK_ERR_ARG, "batch_mul: give a whole Beaver triple (every lane) or none");
    void *sxl[2], *oyl[2], *sx, *oy, *dx;
    char nm[64];
    auto buf = [&](const char* what, int l, void** p) { snprintf(nm, sizeof nm, "%s.%s%d", tag, what, l); return zk_scratch(ctx, nm, n * 32, p); };
    for (int l = 0; l < lanes; l++) { ZK_TRY(buf("sxl", l, &sxl[l])); ZK_TRY(buf("oyl", l, &oyl[l])); }
    ZK_TRY(buf("sx", 0, &sx)); ZK_TRY(buf("oy", 0, &oy)); ZK_TRY(buf("dx", 0, &dx));
    const Fr one_ext = fp_mul<FrParams>(fp_one<FrParams>(), fp_const<FrParams>(FrParams::INT_TO_EXT));
    for (int l = 0; l < lanes; l++) {
        if (dummy) {                                      // DummyFieldTripleSource: the leader holds 1 (in both lanes), the rest 0 (wire/field.rs:49-63)
            if (nt.leader()) {
                hipLaunchKernelGGL(k_vec_add_const, zk_grid(n, 256), 256, 0, ctx->stream, x[l], to_frk(one_ext), sxl[l], n);
                hipLaunchKernelGGL(k_vec_add_const, zk_grid(n, 256), 256, 0, ctx->stream, y[l], to_frk(one_ext), oyl[l], n);
                ZK_HIP(ctx, hipGetLastError());
            } else {
                ZK_HIP(ctx, hipMemcpyAsync(sxl[l], x[l], n * 32, hipMemcpyDeviceToDevice, ctx->stream));
                ZK_HIP(ctx, hipMemcpyAsync(oyl[l], y[l], n * 32, hipMemcpyDeviceToDevice, ctx->stream));
            }
        } else {
            ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_ADD, x[l], tx[l], sxl[l], n));
            ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_ADD, y[l], ty[l], oyl[l], n));
        }
    }
    if (lanes == 1) {
        ZK_TRY(nt.open_vec(sxl[0], n, sx));               // open(s + x), open(o + y)
        ZK_TRY(nt.open_vec(oyl[0], n, oy));
    } else {
        ZK_TRY(zk_shared_spdz_open_vec(nt, sxl[0], sxl[1], n, sx, dx));
        ZK_TRY(zk_shared_spdz_open_vec(nt, oyl[0], oyl[1], n, oy, dx));
    }
    // the local tail; the shift of sx * oy lands on the leader in BOTH lanes (mac_share = 1 there)
    for (int l = 0; l < lanes; l++) ZK_TRY(zk_beaver_combine_dev(ctx, sx, oy, tx[l], ty[l], tz[l], out[l], n));
    return ZK_OK;
}

namespace {

using SH1 = Fq64Field;
using SH2 = Fq264Field;
using SX1 = XYZZ<SH1>;
using SX2 = XYZZ<SH2>;

// LANES = 1: additive shares (AdditiveFieldShare / AdditiveGroupShare); LANES = 2: SPDZ (share lane, MAC lane), every open
// MAC-checked.  z / rs / ss / tx..tz are indexed by lane.
template <int LANES>
int prove_shared_impl(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* const z[2], const zk_fr* const rs[2],
                      const zk_fr* const ss[2], const void* const tx[2], const void* const ty[2], const void* const tz[2],
                      const zk_net_vtable* net, uint8_t proof[192], uint64_t* bytes_sent) {
    const bool dummy = !tx[0] && !ty[0] && !tz[0];
    for (int l = 0; l < LANES; l++)
        if (dummy ? (tx[l] || ty[l] || tz[l]) : (!tx[l] || !ty[l] || !tz[l])) ZK_FAIL(ctx, ZK_ERR_ARG, "prove_shared: give a whole Beaver triple (every lane) or none");
    const bool leader = ctx->party_id == 0;
    const size_t D = (size_t)1 << r->log_d;
    ZkSharedNet nt{ctx, net};
    uint32_t rw[2][8], sw[2][8];
    for (int l = 0; l < LANES; l++) { fr_abi_to_canon_words(rs[l]->l, rw[l]); fr_abi_to_canon_words(ss[l]->l, sw[l]); }
    const SX1 delta1 = xyzz_from_affine<SH1>(aff_to_host64<G1Field>(pk->delta_g1));
    const SX2 delta2 = xyzz_from_affine<SH2>(aff_to_host64<G2Field>(pk->delta_g2));
    // public point x shared scalar: local host arithmetic (scale_pub_group, share/additive.rs:502-508), under the device work
    ZkTask<SX1> f_r_g1[2], f_s_g1[2];                  // (joining handles: the tasks reference this frame and are waited for when it is left)
    ZkTask<SX2> f_s_g2[2];
    for (int l = 0; l < LANES; l++) {
        f_r_g1[l] = zk_async(ctx, [&, l] { return host64_scalar_mul<SH1>(delta1, rw[l]); });
        f_s_g1[l] = zk_async(ctx, [&, l] { return host64_scalar_mul<SH1>(delta1, sw[l]); });
        f_s_g2[l] = zk_async(ctx, [&, l] { return host64_scalar_mul<SH2>(delta2, sw[l]); });
    }
    void *a[2], *b[2], *c[2];
    char nm[32];
    for (int l = 0; l < LANES; l++) {
        const char* names[3] = {"shared_a%d", "shared_b%d", "shared_c%d"};
        void** dst[3] = {&a[l], &b[l], &c[l]};
        for (int k = 0; k < 3; k++) { snprintf(nm, sizeof nm, names[k], l); ZK_TRY(zk_scratch(ctx, nm, D * 32, dst[k])); }
    }
    for (int l = 0; l < LANES; l++) ZK_TRY(zk_groth16_witness_map_pre_dev(ctx, r, z[l], 1, a[l], b[l], c[l]));     // local: linear in the shares
    ZK_TRY(zk_groth16_msms_begin_dev(ctx, pk, r, z[0]));                          // the share lane's four MSMs over z run under the opens
    // FieldShare::batch_mul of the D-element product (share/field.rs:97-129): open(s + x), open(o + y), the local tail -- per lane
    {
        const void* xa[2] = {a[0], LANES == 2 ? a[1] : nullptr};
        const void* yb[2] = {b[0], LANES == 2 ? b[1] : nullptr};
        void* oa[2] = {a[0], LANES == 2 ? a[1] : nullptr};
        ZK_TRY(zk_shared_beaver_mul(nt, LANES, xa, yb, oa, D, tx, ty, tz, "shared_bv"));
    }
    zk_g1_projective m1[2][4];
    zk_g2_projective m2[2];
    for (int l = 0; l < LANES; l++) {
        ZK_TRY(zk_groth16_witness_map_post_dev(ctx, r, a[l], c[l]));                    // h shares in a[l]
        ZK_TRY(zk_groth16_msms_dev(ctx, pk, r, z[l], a[l], m1[l], &m2[l]));            // party-local MSMs (multi_scale_pub_group; spdz.rs:482-488: twice)
    }
    auto pub1 = [&](const Affine<G1Field>& p) { return leader ? xyzz_from_affine<SH1>(aff_to_host64<G1Field>(p)) : xyzz_inf<SH1>(); };   // shift(): leader only
    auto pub2 = [&](const Affine<G2Field>& p) { return leader ? xyzz_from_affine<SH2>(aff_to_host64<G2Field>(p)) : xyzz_inf<SH2>(); };
    SX1 h_acc[2], l_acc[2], r_g1[2], g_a[2], g1_b[2];
    SX2 g2_b[2];
    for (int l = 0; l < LANES; l++) {
        h_acc[l] = host64_proj_from_abi<SH1>((const uint64_t*)&m1[l][0]);
        l_acc[l] = host64_proj_from_abi<SH1>((const uint64_t*)&m1[l][1]);
        const SX1 a_acc = host64_proj_from_abi<SH1>((const uint64_t*)&m1[l][2]), b1_acc = host64_proj_from_abi<SH1>((const uint64_t*)&m1[l][3]);
        const SX2 b2_acc = host64_proj_from_abi<SH2>((const uint64_t*)&m2[l]);
        r_g1[l] = f_r_g1[l].get();
        g_a[l] = xyzz_add<SH1>(xyzz_add<SH1>(xyzz_add<SH1>(r_g1[l], pub1(pk->a0)), a_acc), pub1(pk->alpha_g1));                 // calculate_coeff (:185-201)
        g1_b[l] = xyzz_add<SH1>(xyzz_add<SH1>(xyzz_add<SH1>(f_s_g1[l].get(), pub1(pk->b0_g1)), b1_acc), pub1(pk->beta_g1));
        g2_b[l] = xyzz_add<SH2>(xyzz_add<SH2>(xyzz_add<SH2>(f_s_g2[l].get(), pub2(pk->b0_g2)), b2_acc), pub2(pk->beta_g2));
    }
    // first exchange: open(o + y) for o = s, r (y = the leader's 1: the dummy group triple; from_add_shared: mac = share), open(s + x)
    // for the three scaled points (x = 0) and the reveal of B.  SPDZ: a second exchange of [leader ? opened : 0] - mac, all zero.
    constexpr size_t MW = 2 * 4 + 3 * 18 + 36;
    const Fr y = leader ? fp_mul<FrParams>(fp_one<FrParams>(), fp_const<FrParams>(FrParams::INT_TO_EXT)) : fp_zero<FrParams>();
    auto pack = [&](int l, uint64_t* msg) {
        host_store_ext<FrParams>(msg, fp_add<FrParams>(host_load_ext<FrParams>(ss[l]->l), y));
        host_store_ext<FrParams>(msg + 4, fp_add<FrParams>(host_load_ext<FrParams>(rs[l]->l), y));
        host64_write_projective<SH1>(xyzz_to_affine<SH1>(r_g1[l]), msg + 8);
        host64_write_projective<SH1>(xyzz_to_affine<SH1>(g_a[l]), msg + 26);
        host64_write_projective<SH1>(xyzz_to_affine<SH1>(g1_b[l]), msg + 44);
        host64_write_projective<SH2>(xyzz_to_affine<SH2>(g2_b[l]), msg + 62);
    };
    struct Opened { Fr f[2]; SX1 g[3]; SX2 B; };
    auto sum = [&](const std::vector<uint8_t>& all, Opened& o) -> int {
        o.f[0] = o.f[1] = fp_zero<FrParams>();
        o.g[0] = o.g[1] = o.g[2] = xyzz_inf<SH1>();
        o.B = xyzz_inf<SH2>();
        for (int p = 0; p < nt.parties(); p++) {
            uint64_t w[MW];
            memcpy(w, all.data() + (size_t)p * sizeof w, sizeof w);
            for (int k = 0; k < 2; k++) {
                if (!fr_abi_valid(w + 4 * k)) ZK_FAIL(ctx, ZK_ERR_STATE, "prove_shared: a party sent a non-canonical field element");
                o.f[k] = fp_add<FrParams>(o.f[k], host_load_ext<FrParams>(w + 4 * k));
            }
            // points from a peer: canonical coordinates, on the curve; the malicious-security prover also checks the subgroup
            // (GroupAffine::deserialize behind MpcSerNet::broadcast, channel.rs:12-28).  A party's own message is its own output.
            bool pts_ok = true;
            const bool peer = p != ctx->party_id;
            for (int k = 0; k < 3; k++)
                o.g[k] = xyzz_add<SH1>(o.g[k], peer ? host64_peer_point<SH1>(w + 8 + 18 * k, LANES == 2, pts_ok) : host64_proj_from_abi<SH1>(w + 8 + 18 * k));
            o.B = xyzz_add<SH2>(o.B, peer ? host64_peer_point<SH2>(w + 62, LANES == 2, pts_ok) : host64_proj_from_abi<SH2>(w + 62));
            if (!pts_ok) ZK_FAIL(ctx, ZK_ERR_STATE, "prove_shared: a party sent a point that is not a valid group element");
        }
        return ZK_OK;
    };
    uint64_t msg[MW];
    std::vector<uint8_t> all;
    Opened op;
    pack(0, msg);
    ZK_TRY(nt.gather((const uint8_t*)msg, sizeof msg, all));
    ZK_TRY(sum(all, op));
    if (LANES == 2) {
        // [leader ? x : 0] - mac for every opened value, published and summed: SpdzFieldShare / SpdzGroupShare batch_open's check
        uint64_t mm[MW], dm[MW];
        pack(1, mm);
        for (int k = 0; k < 2; k++)
            host_store_ext<FrParams>(dm + 4 * k, fp_sub<FrParams>(leader ? op.f[k] : fp_zero<FrParams>(), host_load_ext<FrParams>(mm + 4 * k)));
        for (int k = 0; k < 3; k++)
            host64_write_projective<SH1>(xyzz_to_affine<SH1>(xyzz_add<SH1>(leader ? op.g[k] : xyzz_inf<SH1>(), xyzz_neg<SH1>(host64_proj_from_abi<SH1>(mm + 8 + 18 * k)))), dm + 8 + 18 * k);
        host64_write_projective<SH2>(xyzz_to_affine<SH2>(xyzz_add<SH2>(leader ? op.B : xyzz_inf<SH2>(), xyzz_neg<SH2>(host64_proj_from_abi<SH2>(mm + 62)))), dm + 62);
        Opened chk;
        ZK_TRY(nt.gather((const uint8_t*)dm, sizeof dm, all));
        ZK_TRY(sum(all, chk));
        bool ok = fp_is_zero<FrParams>(chk.f[0]) && fp_is_zero<FrParams>(chk.f[1]) && xyzz_is_inf<SH2>(chk.B);
        for (int k = 0; k < 3; k++) ok = ok && xyzz_is_inf<SH1>(chk.g[k]);
        if (!ok) ZK_FAIL(ctx, ZK_ERR_MAC, "SPDZ MAC check failed on a fused open");
    }
    const Fr oy_s = op.f[0], oy_r = op.f[1];
    const SX1 sx_rd = op.g[0], sx_a = op.g[1], sx_b = op.g[2];
    // the local part of GroupShare::scale behind its two opens: z - sx*y (+ sx*oy on the leader), z = 0, y = [leader]; both SPDZ
    // lanes hold the same value (key 1)
    auto scale_finish = [&](const SX1& sxp, const Fr& oyv) {
        if (!leader) return xyzz_inf<SH1>();
        uint64_t l4[4];
        uint32_t kw[8];
        host_store_ext<FrParams>(l4, oyv);
        fr_abi_to_canon_words(l4, kw);
        return xyzz_add<SH1>(host64_scalar_mul<SH1>(sxp, kw), xyzz_neg<SH1>(sxp));
    };
    auto p0 = zk_async(ctx, [&] { return scale_finish(sx_rd, oy_s); });       // r s delta            (:115)
    auto p1 = zk_async(ctx, [&] { return scale_finish(sx_a, oy_s); });        // s A                  (:140)
    const SX1 part2 = scale_finish(sx_b, oy_r);                          // r B1                 (:161)
    SX1 t = xyzz_add<SH1>(p1.get(), part2);
    t = xyzz_add<SH1>(t, xyzz_neg<SH1>(p0.get()));
    SX1 g_c[2];
    for (int l = 0; l < LANES; l++) g_c[l] = xyzz_add<SH1>(xyzz_add<SH1>(t, l_acc[l]), h_acc[l]);      // :169-174
    uint64_t cmsg[18];
    host64_write_projective<SH1>(xyzz_to_affine<SH1>(g_c[0]), cmsg);
    ZK_TRY(nt.gather((const uint8_t*)cmsg, sizeof cmsg, all));          // Proof::reveal of C (A and B were opened above)
    bool c_ok = true;
    auto sum_g1 = [&](const std::vector<uint8_t>& v) {
        SX1 acc = xyzz_inf<SH1>();
        for (int p = 0; p < nt.parties(); p++) {
            uint64_t w[18];
            memcpy(w, v.data() + (size_t)p * sizeof w, sizeof w);
            acc = xyzz_add<SH1>(acc, p != ctx->party_id ? host64_peer_point<SH1>(w, LANES == 2, c_ok) : host64_proj_from_abi<SH1>(w));
        }
        return acc;
    };
    const SX1 C = sum_g1(all);
    if (!c_ok) ZK_FAIL(ctx, ZK_ERR_STATE, "prove_shared: a party sent a point that is not a valid group element");
    if (LANES == 2) {
        host64_write_projective<SH1>(xyzz_to_affine<SH1>(xyzz_add<SH1>(leader ? C : xyzz_inf<SH1>(), xyzz_neg<SH1>(g_c[1]))), cmsg);
        ZK_TRY(nt.gather((const uint8_t*)cmsg, sizeof cmsg, all));
        const SX1 dsum = sum_g1(all);
        if (!c_ok) ZK_FAIL(ctx, ZK_ERR_STATE, "prove_shared: a party sent a point that is not a valid group element");
        if (!xyzz_is_inf<SH1>(dsum)) ZK_FAIL(ctx, ZK_ERR_MAC, "SPDZ MAC check failed on the reveal of C");
    }
    g1_serialize(aff_from_host64<G1Field>(xyzz_to_affine<SH1>(sx_a)), proof);
    g2_serialize(aff_from_host64<G2Field>(xyzz_to_affine<SH2>(op.B)), proof + 48);
    g1_serialize(aff_from_host64<G1Field>(xyzz_to_affine<SH1>(C)), proof + 144);
    if (bytes_sent) *bytes_sent = nt.bytes;
    return ZK_OK;
}

}  // namespace

extern "C" int zk_groth16_prove_shared(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* z_share, const zk_fr* r_share,
                                       const zk_fr* s_share, const void* tx, const void* ty, const void* tz, const zk_net_vtable* net,
                                       uint8_t proof[192], uint64_t* bytes_sent) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z_share || !r_share || !s_share || !proof) return ZK_ERR_ARG;
    const void* z[2] = {z_share, nullptr};
    const zk_fr *rs[2] = {r_share, nullptr}, *ss[2] = {s_share, nullptr};
    const void *txs[2] = {tx, nullptr}, *tys[2] = {ty, nullptr}, *tzs[2] = {tz, nullptr};
    return prove_shared_impl<1>(ctx, pk, r, z, rs, ss, txs, tys, tzs, net, proof, bytes_sent);
    ZK_API_END
}

extern "C" int zk_groth16_prove_shared_spdz(zk_ctx* ctx, const zk_pk* pk, const zk_r1cs* r, const void* const z_lanes[2],
                                            const zk_fr r_lanes[2], const zk_fr s_lanes[2], const void* const tx_lanes[2],
                                            const void* const ty_lanes[2], const void* const tz_lanes[2], const zk_net_vtable* net,
                                            uint8_t proof[192], uint64_t* bytes_sent) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !r || !z_lanes || !z_lanes[0] || !z_lanes[1] || !r_lanes || !s_lanes || !proof) return ZK_ERR_ARG;
    const zk_fr *rs[2] = {&r_lanes[0], &r_lanes[1]}, *ss[2] = {&s_lanes[0], &s_lanes[1]};
    const void* none[2] = {nullptr, nullptr};
    return prove_shared_impl<2>(ctx, pk, r, z_lanes, rs, ss, tx_lanes ? tx_lanes : none, ty_lanes ? ty_lanes : none,
                                tz_lanes ? tz_lanes : none, net, proof, bytes_sent);
    ZK_API_END
}

