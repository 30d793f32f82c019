// groth16_multi.hip -- ONE prover's create_proof spread over several devices (second level: "the 5 MSMs are independent
// of each other and can go to different GPUs; an MSM splits by base ranges into independent partial sums").
//
// Replaces the same reference code as groth16_prove.hip (src/groth16.rs:68-183); what changes is where the five sums of
// :106 (H), :110 (L), :137 (A), :148 (B in G1), :160 (B in G2) are computed.  The work is cut by COST, not by job: B-in-G2 terms
// weigh 2.7 G1 terms (k_accum_g2pair 5.6 ms against 2.1 ms per 2^20 terms), the jobs are laid end to end in the order B2, A, B1,
// L, H and every context takes an equal stretch of that line -- whole jobs where it can, base ranges where a boundary falls
// inside one.  A context that holds a piece of H runs the witness map itself (1 ms, all local: no exchange of D-element
// vectors); the assignment reaches the other devices by one peer copy.  Every context needs its own copy of the key and of the
// constraint system on its device.  The partial sums (one Jacobian point per piece) meet on the host, which adds them and runs the
// usual O(1) tail.  No collective: the only exchange is 32 n bytes of assignment out and a few hundred bytes of points back.
#include "../../include/zkmpc_hip.h"
#include "groth16_int.hpp"
#include <algorithm>
#include <thread>

using namespace zk;

namespace {

struct Piece { int job; size_t lo, n; };        // job: 0 = B in G2, 1 = A, 2 = B in G1, 3 = L, 4 = H; terms [lo, lo + n) of that job

constexpr double G2_WEIGHT = 2.7;               // cost of a G2 term in G1 terms
constexpr size_t MIN_PIECE = 4096;              // no piece smaller than this (unless it is all that is left of its job)

// the stretch [share * d, share * (d + 1)) of the cost line, as pieces
std::vector<std::vector<Piece>> deal(const size_t len[5], int n_ctx) {
    const double w[5] = {G2_WEIGHT, 1, 1, 1, 1};
    double total = 0;
    for (int j = 0; j < 5; j++) total += w[j] * (double)len[j];
    std::vector<std::vector<Piece>> out(n_ctx);
    int d = 0;
    double room = total / n_ctx;
    for (int j = 0; j < 5; j++) {
        size_t lo = 0;
        while (lo < len[j]) {
            size_t take = len[j] - lo;
            if (d < n_ctx - 1) {
                const size_t fit = (size_t)(room / w[j]);
                if (fit < take) take = fit;
                if (take < MIN_PIECE && take < len[j] - lo) {          // too small a crumb: the next context starts here
                    d++;
                    room = total / n_ctx;
                    continue;
                }
                if (len[j] - lo - take < MIN_PIECE) take = len[j] - lo;  // do not leave a crumb either
            }
            out[d].push_back(Piece{j, lo, take});
            lo += take;
            room -= w[j] * (double)take;
            if (d < n_ctx - 1 && room < w[j] * (double)MIN_PIECE) { d++; room = total / n_ctx; }
        }
    }
    return out;
}

}  // namespace

static int prove_multi(zk_ctx* ctx0, zk_ctx* const* ctxs, const zk_pk* const* pks, const zk_r1cs* const* rs, int n_ctx, const void* z_dev0,
                       const zk_fr* r_, const zk_fr* s_, uint8_t proof[192]) {
    for (int d = 0; d < n_ctx; d++) {
        if (!ctxs[d] || !pks[d] || !rs[d]) return ZK_ERR_ARG;
        for (int e = 0; e < d; e++)
            if (ctxs[e] == ctxs[d]) ZK_FAIL(ctx0, ZK_ERR_ARG, "zk_groth16_prove_multi: the same context twice (a context runs one MSM batch at a time)");
        if (rs[d]->nc != rs[0]->nc || rs[d]->ni != rs[0]->ni || rs[d]->nw != rs[0]->nw || pks[d]->a->n != pks[0]->a->n ||
            pks[d]->h->n != pks[0]->h->n || pks[d]->l->n != pks[0]->l->n)
            ZK_FAIL(ctx0, ZK_ERR_ARG, "zk_groth16_prove_multi: the contexts' keys / constraint systems differ in shape");
    }
    const zk_r1cs* r0 = rs[0];
    const size_t D = (size_t)1 << r0->log_d, ni = r0->ni, nw = r0->nw, nvars = (ni - 1) + nw, m = ni + nw;
    if (pks[0]->a->n != nvars + 1 || pks[0]->b_g1->n != nvars + 1 || pks[0]->b_g2->n != nvars + 1 || pks[0]->l->n != nw)
        ZK_FAIL(ctx0, ZK_ERR_ARG, "groth16: proving key does not match the constraint system");
    const size_t len[5] = {nvars, nvars, nvars, nw, std::min(pks[0]->h->n, D)};
    const std::vector<std::vector<Piece>> plan = deal(len, n_ctx);

    // the assignment was produced on context 0's stream
    ZK_HIP(ctx0, hipStreamSynchronize(ctx0->stream));
    // a context whose device has no peer access to context 0's gets the assignment through page-locked host memory instead
    void* z_stage = nullptr;
    for (int d = 1; d < n_ctx && !z_stage; d++) {
        int peer = 1;
        if (ctxs[d]->device != ctx0->device && (hipDeviceCanAccessPeer(&peer, ctxs[d]->device, ctx0->device) != hipSuccess || !peer)) {
            auto& pin = ctx0->pinned[-2];
            if (pin.bytes < m * 32) {
                if (pin.p) (void)hipHostFree(pin.p);
                pin.p = nullptr; pin.bytes = 0;
                ZK_HIP(ctx0, hipHostMalloc(&pin.p, m * 32, hipHostMallocPortable));
                pin.bytes = m * 32;
            }
            z_stage = pin.p;
            ZK_HIP(ctx0, hipMemcpyAsync(z_stage, z_dev0, m * 32, hipMemcpyDeviceToHost, ctx0->stream));
            ZK_HIP(ctx0, hipStreamSynchronize(ctx0->stream));
        }
    }
    std::vector<int> rc(n_ctx, ZK_OK);
    std::vector<std::vector<zk_g1_projective>> part1(n_ctx);
    std::vector<std::vector<zk_g2_projective>> part2(n_ctx);
    auto run = [&](int d) -> int {
        zk_ctx* c = ctxs[d];
        const std::vector<Piece>& pc = plan[d];
        if (pc.empty()) return ZK_OK;
        ZkDeviceGuard guard(c->device);
        if (guard.err != hipSuccess) ZK_FAIL(c, ZK_ERR_HIP, "zk_groth16_prove_multi: cannot make the context's device current");
        const void* z = z_dev0;
        if (d > 0) {
            void* zc;
            ZK_TRY(zk_scratch(c, "multi_z", m * 32, &zc));
            int peer = 1;
            if (c->device != ctx0->device && hipDeviceCanAccessPeer(&peer, c->device, ctx0->device) != hipSuccess) peer = 0;
            if (peer) {
                ZK_HIP(c, hipMemcpyPeerAsync(zc, c->device, z_dev0, ctx0->device, m * 32, c->stream));     // (a plain copy when both contexts share a device)
            } else {
                // no peer path between the two devices: through the host copy made once below (z_stage)
                ZK_HIP(c, hipMemcpyAsync(zc, z_stage, m * 32, hipMemcpyHostToDevice, c->stream));
            }
            z = zc;
        }
        const char* zb = (const char*)z;
        void* h = nullptr;
        for (const Piece& p : pc)
            if (p.job == 4 && !h) {
                ZK_TRY(zk_scratch(c, "prove_h", D * 32, &h));
                ZK_TRY(zk_groth16_witness_map_dev(c, rs[d], z, h));
            }
        const zk_pk* pk = pks[d];
        std::vector<const zk_bases*> bases;
        std::vector<size_t> offs, lens;
        std::vector<const void*> scal;
        std::vector<void*> outs;
        size_t n1 = 0, n2 = 0;
        for (const Piece& p : pc) (p.job == 0 ? n2 : n1)++;
        part1[d].resize(n1);
        part2[d].resize(n2);
        size_t k1 = 0, k2 = 0;
        for (const Piece& p : pc) {
            switch (p.job) {
                case 0: bases.push_back(pk->b_g2); offs.push_back(1 + p.lo); scal.push_back(zb + 32 + p.lo * 32); break;     // query[1..] x z[1..]
                case 1: bases.push_back(pk->a); offs.push_back(1 + p.lo); scal.push_back(zb + 32 + p.lo * 32); break;
                case 2: bases.push_back(pk->b_g1); offs.push_back(1 + p.lo); scal.push_back(zb + 32 + p.lo * 32); break;
                case 3: bases.push_back(pk->l); offs.push_back(p.lo); scal.push_back(zb + (ni + p.lo) * 32); break;          // l_query x the witness part
                default: bases.push_back(pk->h); offs.push_back(p.lo); scal.push_back((const char*)h + p.lo * 32); break;
            }
            lens.push_back(p.n);
            outs.push_back(p.job == 0 ? (void*)&part2[d][k2++] : (void*)&part1[d][k1++]);
        }
        return zk_msm_batch_dev(c, bases.size(), bases.data(), offs.data(), scal.data(), lens.data(), outs.data());
    };
    {
        std::vector<std::thread> th;
        th.reserve(n_ctx);
        for (int d = 1; d < n_ctx; d++) {
            try {
                th.emplace_back([&, d] {
                    try { rc[d] = run(d); } catch (...) { rc[d] = ZK_ERR_STATE; ctxs[d]->last_error = "zk_groth16_prove_multi: exception on a helper thread"; }
                });
            } catch (...) {                      // no thread to be had: this context's share runs here, after context 0's
                rc[d] = -1000;
            }
        }
        try { rc[0] = run(0); } catch (...) { rc[0] = ZK_ERR_STATE; }
        for (int d = 1; d < n_ctx; d++)
            if (rc[d] == -1000) { try { rc[d] = run(d); } catch (...) { rc[d] = ZK_ERR_STATE; } }
        for (auto& t : th) t.join();
    }
    for (int d = 0; d < n_ctx; d++)
        if (rc[d] != ZK_OK) {
            if (d > 0) ctx0->last_error = "context " + std::to_string(d) + ": " + ctxs[d]->last_error;
            return rc[d];
        }
    // the partial sums of every job, added on the host
    using H1 = Fq64Field;
    using H2 = Fq264Field;
    XYZZ<H1> sum1[5];
    for (auto& x : sum1) x = xyzz_inf<H1>();
    XYZZ<H2> sum2 = xyzz_inf<H2>();
    for (int d = 0; d < n_ctx; d++) {
        size_t k1 = 0, k2 = 0;
        for (const Piece& p : plan[d]) {
            if (p.job == 0) sum2 = xyzz_add<H2>(sum2, host64_proj_from_abi<H2>((const uint64_t*)&part2[d][k2++]));
            else sum1[p.job] = xyzz_add<H1>(sum1[p.job], host64_proj_from_abi<H1>((const uint64_t*)&part1[d][k1++]));
        }
    }
    zk_g1_projective a_sum, b1_sum, l_sum, h_sum;
    zk_g2_projective b2_sum;
    host64_write_projective<H1>(xyzz_to_affine<H1>(sum1[1]), (uint64_t*)&a_sum);
    host64_write_projective<H1>(xyzz_to_affine<H1>(sum1[2]), (uint64_t*)&b1_sum);
    host64_write_projective<H1>(xyzz_to_affine<H1>(sum1[3]), (uint64_t*)&l_sum);
    host64_write_projective<H1>(xyzz_to_affine<H1>(sum1[4]), (uint64_t*)&h_sum);
    host64_write_projective<H2>(xyzz_to_affine<H2>(sum2), (uint64_t*)&b2_sum);
    ZkProofTail tail(ctx0, pks[0], r_, s_);
    tail.abc_ready(a_sum, b1_sum, b2_sum);
    tail.finish(h_sum, l_sum, proof);
    return ZK_OK;
}

extern "C" int zk_groth16_prove_multi(zk_ctx* const* ctxs, const zk_pk* const* pks, const zk_r1cs* const* rs, int n_ctx, const void* z_dev0,
                                      const zk_fr* r_, const zk_fr* s_, uint8_t proof[192]) {
    ZK_API_BEGIN_NOCTX
    if (!ctxs || !pks || !rs || n_ctx < 1 || n_ctx > 64 || !ctxs[0] || !z_dev0 || !r_ || !s_ || !proof) return ZK_ERR_ARG;
    // the body runs under context 0's guard: its device current, its error string, the exception barrier
    return zk_api_guarded(ctxs[0], [&]() -> int { return prove_multi(ctxs[0], ctxs, pks, rs, n_ctx, z_dev0, r_, s_, proof); });
    ZK_API_END
}

// How zk_groth16_prove_multi would deal a proof of this shape over n_ctx contexts: one line per piece, "ctx job lo n" (job 0 = B in
// G2, 1 = A, 2 = B in G1, 3 = L, 4 = H).  Diagnostics / tests; *written = the bytes put into out (excluding the terminator).
extern "C" int zk_groth16_multi_plan(const zk_pk* pk, const zk_r1cs* r, int n_ctx, char* out, size_t cap, size_t* written) {
    ZK_API_BEGIN_NOCTX
    if (!pk || !r || n_ctx < 1 || n_ctx > 64 || !out || !cap || !written) return ZK_ERR_ARG;
    const size_t D = (size_t)1 << r->log_d, nvars = (r->ni - 1) + r->nw;
    const size_t len[5] = {nvars, nvars, nvars, r->nw, std::min(pk->h->n, D)};
    std::string s;
    const auto plan = deal(len, n_ctx);
    for (int d = 0; d < n_ctx; d++)
        for (const Piece& p : plan[d]) s += std::to_string(d) + " " + std::to_string(p.job) + " " + std::to_string(p.lo) + " " + std::to_string(p.n) + "\n";
    if (s.size() + 1 > cap) return ZK_ERR_ARG;
    memcpy(out, s.data(), s.size());
    out[s.size()] = 0;
    *written = s.size();
    return ZK_OK;
    ZK_API_END
}
