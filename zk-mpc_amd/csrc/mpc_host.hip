// mpc_host.hip -- the trait surface for the COLLABORATIVE element types, on the caller's own host slices.
//
// Under E = MpcPairingEngine the unchanged create_proof (src/groth16.rs:68-183, witness_map :240-306) reaches the arithmetic with
//     domain.{ifft, coset_fft, coset_ifft}_in_place(&mut Vec<MpcField<Fr, S>>)               src/groth16.rs:278-303
//     MpcField::batch_product_in_place(&mut [MpcField], &[MpcField])                          mpc-algebra/src/wire/field.rs:917-958
//         -> FieldShare::batch_mul (Beaver, two vector opens)                                   mpc-algebra/src/share/field.rs:97-129
//     domain.divide_by_vanishing_poly_on_coset_in_place(&mut [MpcField])                       arkworks/algebra/poly/src/domain/mod.rs:183-190
//     MpcG1Affine::multi_scalar_mul(&[MpcG1Affine], &[MpcField])   (and G2)                     mpc-algebra/src/wire/pairing.rs:714-777
//         -> GroupShare::multi_scale_pub_group                                                  share/additive.rs:517-520, share/spdz.rs:482-488
// where an element is the enum MpcField { Public(Fr), Shared(S) } (wire/field.rs:37-40) -- a discriminant and a payload, 40 bytes
// with S = AdditiveFieldShare { val }, 72 with S = SpdzFieldShare { sh, mac } -- and a base is MpcGroup { Public(G), Shared(..) }
// around a GroupAffine.  None of that is a contiguous array of Fr, so the entry points below read and write the caller's
// elements IN PLACE through a layout descriptor (stride, where the discriminant sits, where each payload's words sit) that the
// binding takes off a value: nothing is repacked per call on the Rust side, and nothing depends on how rustc orders an enum.
//
// Semantics, from the reference's operator impls (wire/field.rs:339-362, 414-437, 463-492; share/additive.rs:145-152,
// share/spdz.rs:207-219): a Public(x) met by a shared value acts as the share "x on the leader, 0 elsewhere" in BOTH lanes
// (shift: the leader adds x, the MAC lane adds mac_share * x with mac_share = 1 on the leader), a product with a public value
// scales every lane.  Hence
//   a linear transform of a vector holding at least one Shared element = the same transform of the party's lane vectors with
//     Public(x) read as (leader ? x : 0); every output is Shared.  All elements Public: the transform of the values, outputs Public.
//   an element-wise product with a public constant keeps every element's variant.
//   an MSM over public bases = the MSM over those lane vectors (MpcField::all_public_or_shared forces a mixed vector to shares:
//     wire/field.rs:75-100); all scalars Public: the plain MSM, which the wire wraps with from_public.
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "internal.hpp"
#include "sharednet.hpp"
#include <atomic>
#include <immintrin.h>
#include <string.h>

using namespace zk;

namespace {

struct FieldView {
    char* host;
    zk_mpc_field_layout lay;
    bool leader;
    int lanes() const { return lay.off_mac == SIZE_MAX ? 1 : 2; }
    bool is_public(size_t i) const { return (unsigned char)host[i * lay.stride + lay.off_tag] == lay.tag_public; }
};

int check_layout(zk_ctx* ctx, const zk_mpc_field_layout* l) {
    if (!l) return ZK_ERR_ARG;
    if (l->stride < 32 || l->off_tag >= l->stride || l->off_public + 32 > l->stride || l->off_share + 32 > l->stride ||
        (l->off_mac != SIZE_MAX && l->off_mac + 32 > l->stride) || l->tag_public == l->tag_shared)
        ZK_FAIL(ctx, ZK_ERR_ARG, "MpcField layout: the field offsets do not fit the stride (or the two discriminants are equal)");
    return ZK_OK;
}

__attribute__((target("avx2"))) inline void stream32_avx2(char* dst, const char* src) {
    _mm256_stream_si256((__m256i*)dst, _mm256_loadu_si256((const __m256i*)src));
}
__attribute__((target("avx2"))) inline void zero32_avx2(char* dst) { _mm256_stream_si256((__m256i*)dst, _mm256_setzero_si256()); }

// what a gather saw (OR-ed in by the ring's threads)
struct Seen { std::atomic<int> pub{0}, shared{0}; };

// lane `lane` of elements [off / 32, (off + len) / 32) as contiguous Fr words.  Public elements: their value when keep_public (or
// on the leader), zero otherwise.
ZkXferFill gatherer(const FieldView v, int lane, bool keep_public, Seen* seen) {
    return [=](char* dst, size_t off, size_t len) {
        static const bool avx2 = __builtin_cpu_supports("avx2");
        const bool nt = avx2 && ((uintptr_t)dst & 31) == 0;
        const size_t lo = off / 32, cnt = len / 32;
        const size_t off_sh = lane == 0 ? v.lay.off_share : v.lay.off_mac;
        const bool take_pub = keep_public || v.leader;
        bool sp = false, ss = false;
        for (size_t k = 0; k < cnt; k++) {
            const char* p = v.host + (lo + k) * v.lay.stride;
            const bool pub = (unsigned char)p[v.lay.off_tag] == v.lay.tag_public;
            sp = sp || pub; ss = ss || !pub;
            char* d = dst + k * 32;
            if (pub && !take_pub) { if (nt) zero32_avx2(d); else memset(d, 0, 32); }
            else {
                const char* s = p + (pub ? v.lay.off_public : off_sh);
                if (nt) stream32_avx2(d, s); else memcpy(d, s, 32);
            }
        }
        if (nt) _mm_sfence();
        if (sp) seen->pub.store(1);
        if (ss) seen->shared.store(1);
    };
}

enum class Out { Shared, Public, Keep };
// elements [off / 32, ..) of a device lane vector back into the caller's structs.  Shared: every element becomes Shared (the
// discriminant is written with lane 0); Public: values to the Public payload, discriminants untouched (they all say Public); Keep:
// the element's own variant decides where the words go (Public elements take lane 0's, and ignore the MAC lane).
ZkXferDrain scatterer(const FieldView v, int lane, Out mode) {
    return [=](const char* src, size_t off, size_t len) {
        const size_t lo = off / 32, cnt = len / 32;
        const size_t off_sh = lane == 0 ? v.lay.off_share : v.lay.off_mac;
        for (size_t k = 0; k < cnt; k++) {
            char* p = v.host + (lo + k) * v.lay.stride;
            const char* s = src + k * 32;
            if (mode == Out::Shared) {
                if (lane == 0) p[v.lay.off_tag] = (char)v.lay.tag_shared;
                memcpy(p + off_sh, s, 32);
            } else if (mode == Out::Public) {
                memcpy(p + v.lay.off_public, s, 32);
            } else {
                const bool pub = (unsigned char)p[v.lay.off_tag] == v.lay.tag_public;
                if (pub) { if (lane == 0) memcpy(p + v.lay.off_public, s, 32); }
                else memcpy(p + off_sh, s, 32);
            }
        }
    };
}

// The lane vectors of n caller elements on the device, for a LINEAR operation (transform, MSM): *all_public says which reading was
// taken -- true: one lane of public values (every party holds them), false: lanes() lanes with Public(x) read as (leader ? x : 0).
// A guess from three discriminants decides which gather runs first; the gather itself sees every discriminant and a wrong guess
// is repaired by one more pass (never on the leader, whose words are the same under both readings).
int gather_linear(zk_ctx* ctx, const FieldView& v, size_t n, void* const lane_dev[2], bool* all_public) {
    if (n == 0) { *all_public = true; return ZK_OK; }
    const bool guess_public = v.is_public(0) && v.is_public(n / 2) && v.is_public(n - 1);
    Seen seen;
    ZK_TRY(zk_xfer_h2d_fn(ctx, lane_dev[0], n * 32, gatherer(v, 0, guess_public, &seen), true, nullptr));
    const bool pub = !seen.shared.load();
    if (pub != guess_public && !v.leader && seen.pub.load()) {
        Seen again;
        ZK_TRY(zk_xfer_h2d_fn(ctx, lane_dev[0], n * 32, gatherer(v, 0, pub, &again), true, nullptr));
    }
    if (!pub && v.lanes() == 2) {
        Seen s1;
        ZK_TRY(zk_xfer_h2d_fn(ctx, lane_dev[1], n * 32, gatherer(v, 1, false, &s1), true, nullptr));
    }
    *all_public = pub;
    return ZK_OK;
}

// a fingerprint of a scalar vector from host memory (64 sampled elements: discriminant and first lane): what msm.hip's speculation
// recognises the `assignment` of calculate_coeff's three calls -- and the `h` a transform has just written -- by (confirmed on the
// device before anything is used)
uint64_t mpc_scalars_fingerprint(const FieldView& v, size_t n) {
    uint64_t sfp = 0x9E3779B97F4A7C15ull ^ (uint64_t)n;
    const size_t S = n < 64 ? n : 64;
    for (size_t k = 0; k < S; k++) {
        const size_t i = S > 1 ? k * (n - 1) / (S - 1) : 0;
        const char* p = v.host + i * v.lay.stride;
        const bool is_pub = (unsigned char)p[v.lay.off_tag] == v.lay.tag_public;
        uint64_t w[4];
        memcpy(w, p + (is_pub ? v.lay.off_public : v.lay.off_share), 32);
        sfp ^= is_pub ? 0x5bd1e995u : 0;
        for (int q = 0; q < 4; q++) { sfp ^= w[q]; sfp *= 0xFF51AFD7ED558CCDull; sfp ^= sfp >> 32; }
    }
    return sfp ? sfp : 1;
}

int lane_bufs(zk_ctx* ctx, const char* tag, size_t n, int lanes, void* out[2]) {
    char nm[48];
    out[0] = out[1] = nullptr;
    for (int l = 0; l < lanes; l++) {
        snprintf(nm, sizeof nm, "%s%d", tag, l);
        ZK_TRY(zk_scratch(ctx, nm, n * 32, &out[l]));
    }
    return ZK_OK;
}

template <int G>
int mpc_msm(zk_ctx* ctx, const void* bases, size_t nb, const zk_mpc_group_layout* bl, const void* scalars, size_t ns,
            const zk_mpc_field_layout* sl, void* out_lanes, int* scalars_public) {
    constexpr size_t PROJ = G == 1 ? sizeof(zk_g1_projective) : sizeof(zk_g2_projective);
    if (!ctx || !out_lanes || !bl || !sl) return ZK_ERR_ARG;
    ZK_TRY(check_layout(ctx, sl));
    const size_t n = std::min(nb, ns);                      // variable_base.rs:15-17
    if (n && (!bases || !scalars)) return ZK_ERR_ARG;
    const int lanes = sl->off_mac == SIZE_MAX ? 1 : 2;
    char* const out = (char*)out_lanes;
    if (n == 0) {                                           // the empty sum in every lane
        for (int l = 0; l < 2; l++) {
            if (G == 1) { zk_g1_affine inf{}; ZK_TRY(zk_g1_from_affine(&inf, (zk_g1_projective*)(out + l * PROJ))); }
            else { zk_g2_affine inf{}; ZK_TRY(zk_g2_from_affine(&inf, (zk_g2_projective*)(out + l * PROJ))); }
        }
        if (scalars_public) *scalars_public = 0;
        return ZK_OK;
    }
    ZK_TRY(zk_bases_cache_poll(ctx));
    FieldView v{(char*)const_cast<void*>(scalars), *sl, ctx->party_id == 0};
    void* lane[2];
    ZK_TRY(lane_bufs(ctx, "mpc_msm_s", n, lanes, lane));
    bool pub = false;
    ZK_TRY(gather_linear(ctx, v, n, lane, &pub));
    ZkHostTable t;
    t.group = G;
    t.host = bases;
    t.stride = bl->point.stride; t.off_x = bl->point.off_x; t.off_y = bl->point.off_y; t.off_inf = bl->point.off_infinity;
    t.off_tag = bl->off_tag; t.tag_public = bl->tag_public;
    const size_t nu = nb <= 2 * n ? nb : n;
    const int run = pub ? 1 : lanes;
    void* outs[2] = {out, out + PROJ};
    const void* sc[2] = {lane[0], lane[1]};
    ZK_TRY(zk_msm_table_run(ctx, t, nu, run, sc, n, outs, mpc_scalars_fingerprint(v, n)));
    if (run == 1) memcpy(out + PROJ, out, PROJ);            // (one lane: the second slot mirrors it, never garbage)
    if (scalars_public) *scalars_public = pub ? 1 : 0;
    return ZK_OK;
}

}  // namespace

extern "C" int zk_mpc_fft_in_place(zk_ctx* ctx, void* vec, size_t n, const zk_mpc_field_layout* lay, uint32_t log_n, int inverse, int coset) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !vec) return ZK_ERR_ARG;
    ZK_TRY(check_layout(ctx, lay));
    if (log_n > 28) ZK_FAIL(ctx, ZK_ERR_ARG, "NTT size unsupported (log_n > 28)");
    const size_t N = (size_t)1 << log_n;
    if (n > N) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_mpc_fft_in_place: n exceeds the domain size");
    ZK_TRY(zk_bases_cache_poll(ctx));
    FieldView v{(char*)vec, *lay, ctx->party_id == 0};
    void* lane[2];
    ZK_TRY(lane_bufs(ctx, "mpc_fft", N, v.lanes(), lane));
    bool pub = true;
    ZK_TRY(gather_linear(ctx, v, n, lane, &pub));
    const int run = pub ? 1 : v.lanes();
    for (int l = 0; l < run; l++)
        if (N > n) ZK_HIP(ctx, hipMemsetAsync((char*)lane[l] + n * 32, 0, (N - n) * 32, ctx->stream));   // Vec::resize(size, zero): Public(0)
    if (run == 1) ZK_TRY(zk_ntt_launch(ctx, lane[0], log_n, inverse, coset));
    else ZK_TRY(zk_ntt_launch_batch(ctx, lane, 2, log_n, inverse, coset));
    if (run == 1) zk_msm_spec_fft_begin(ctx, lane[0], N, (inverse ? 2 : 0) + (coset ? 1 : 0));    // (msm.hip: `h = witness_map(..)` is the H query's scalar vector next)
    for (int l = 0; l < run; l++) ZK_TRY(zk_xfer_d2h_fn(ctx, lane[l], N * 32, scatterer(v, l, pub ? Out::Public : Out::Shared)));
    if (run == 1) zk_msm_spec_fft_end(ctx, N, (inverse ? 2 : 0) + (coset ? 1 : 0), [&v](size_t m) { return mpc_scalars_fingerprint(v, m); });
    if (pub && N > n)                                       // the elements the caller appended are Public(0) already; say so for a caller that did not initialise them
        for (size_t i = n; i < N; i++) v.host[i * lay->stride + lay->off_tag] = (char)lay->tag_public;
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_mpc_divide_by_vanishing_on_coset_in_place(zk_ctx* ctx, void* evals, const zk_mpc_field_layout* lay, uint32_t log_n) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !evals) return ZK_ERR_ARG;
    ZK_TRY(check_layout(ctx, lay));
    if (log_n > 28) ZK_FAIL(ctx, ZK_ERR_ARG, "NTT size unsupported (log_n > 28)");
    const size_t N = (size_t)1 << log_n;
    uint32_t zinv[9];
    ZK_TRY(zk_ntt_vanishing_inv(ctx, log_n, zinv));
    FieldView v{(char*)evals, *lay, ctx->party_id == 0};
    void* lane[2];
    ZK_TRY(lane_bufs(ctx, "mpc_fft", N, v.lanes(), lane));
    // element-wise `*eval *= &i` with a public i (domain/mod.rs:186-189 over wire/field.rs:463-492): every element keeps its variant
    for (int l = 0; l < v.lanes(); l++) {
        Seen seen;
        ZK_TRY(zk_xfer_h2d_fn(ctx, lane[l], N * 32, gatherer(v, l, true, &seen), true, nullptr));
        ZK_TRY(zk_vec_scale_launch(ctx, lane[l], zinv, lane[l], N));
        ZK_TRY(zk_xfer_d2h_fn(ctx, lane[l], N * 32, scatterer(v, l, Out::Keep)));
        if (l == 0 && !seen.shared.load()) break;            // all Public: no MAC lane to scale
    }
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_mpc_batch_product_in_place(zk_ctx* ctx, void* selfs, const void* others, size_t n, const zk_mpc_field_layout* lay,
                                             const zk_fr* const* triple_host, const zk_net_vtable* net, uint64_t* bytes_sent) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (n && (!selfs || !others))) return ZK_ERR_ARG;
    ZK_TRY(check_layout(ctx, lay));
    if (bytes_sent) *bytes_sent = 0;
    if (n == 0) return ZK_OK;
    ZK_TRY(zk_bases_cache_poll(ctx));
    FieldView a{(char*)selfs, *lay, ctx->party_id == 0}, b{(char*)const_cast<void*>(others), *lay, ctx->party_id == 0};
    const int lanes = a.lanes();
    const bool a_shared = !a.is_public(0), b_shared = !b.is_public(0);      // selfs[0].is_shared(), others[0].is_shared() (wire/field.rs:918-919)
    void *x[2], *y[2];
    ZK_TRY(lane_bufs(ctx, "mpc_bp_a", n, lanes, x));
    ZK_TRY(lane_bufs(ctx, "mpc_bp_b", n, lanes, y));
    Seen sa, sb;
    // every element's own words: a Public one keeps its value on every party (the products below are element-wise)
    for (int l = 0; l < (a_shared ? lanes : 1); l++) ZK_TRY(zk_xfer_h2d_fn(ctx, x[l], n * 32, gatherer(a, l, true, &sa), true, nullptr));
    for (int l = 0; l < (b_shared ? lanes : 1); l++) ZK_TRY(zk_xfer_h2d_fn(ctx, y[l], n * 32, gatherer(b, l, true, &sb), true, nullptr));
    if ((a_shared ? sa.pub.load() : sa.shared.load())) ZK_FAIL(ctx, ZK_ERR_ARG, "batch_product_in_place: Selfs heterogenously shared! (wire/field.rs:920-923)");
    if ((b_shared ? sb.pub.load() : sb.shared.load())) ZK_FAIL(ctx, ZK_ERR_ARG, "batch_product_in_place: others heterogenously shared! (wire/field.rs:924-927)");
    if (a_shared && b_shared) {                              // S::batch_mul over a Beaver triple (share/field.rs:97-129)
        const void *tx[2] = {nullptr, nullptr}, *ty[2] = {nullptr, nullptr}, *tz[2] = {nullptr, nullptr};
        if (triple_host) {
            char nm[32];
            for (int l = 0; l < lanes; l++)
                for (int k = 0; k < 3; k++) {
                    const zk_fr* src = triple_host[3 * l + k];
                    if (!src) ZK_FAIL(ctx, ZK_ERR_ARG, "batch_product_in_place: give a whole Beaver triple (x, y, z per lane) or NULL");
                    void* d;
                    snprintf(nm, sizeof nm, "mpc_bp_t%d%d", l, k);
                    ZK_TRY(zk_scratch(ctx, nm, n * 32, &d));
                    ZK_TRY(zk_xfer_h2d(ctx, d, src, n * 32));
                    (k == 0 ? tx : k == 1 ? ty : tz)[l] = d;
                }
        }
        ZkSharedNet nt{ctx, net};
        ZK_TRY(zk_shared_beaver_mul(nt, lanes, x, y, x, n, tx, ty, tz, "mpc_bp"));
        if (bytes_sent) *bytes_sent = nt.bytes;
        for (int l = 0; l < lanes; l++) ZK_TRY(zk_xfer_d2h_fn(ctx, x[l], n * 32, scatterer(a, l, Out::Shared)));
        return ZK_OK;
    }
    // `*a *= b` element by element (wire/field.rs:463-476): public x public stays Public; a public factor scales every lane
    if (!a_shared && !b_shared) {
        ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_MUL, x[0], y[0], x[0], n));
        ZK_TRY(zk_xfer_d2h_fn(ctx, x[0], n * 32, scatterer(a, 0, Out::Public)));
        return ZK_OK;
    }
    for (int l = 0; l < lanes; l++) {
        const void* sh = a_shared ? x[l] : y[l];
        const void* pb = a_shared ? y[0] : x[0];
        void* o = a_shared ? x[l] : y[l];
        ZK_TRY(zk_vec_op_launch(ctx, ZK_OP_MUL, sh, pb, o, n));
        ZK_TRY(zk_xfer_d2h_fn(ctx, o, n * 32, scatterer(a, l, Out::Shared)));
    }
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_mpc_msm_g1(zk_ctx* ctx, const void* bases, size_t nb, const zk_mpc_group_layout* bl, const void* scalars, size_t ns,
                             const zk_mpc_field_layout* sl, zk_g1_projective out_lanes[2], int* scalars_public) {
    ZK_API_BEGIN(ctx)
    return mpc_msm<1>(ctx, bases, nb, bl, scalars, ns, sl, out_lanes, scalars_public);
    ZK_API_END
}
extern "C" int zk_mpc_msm_g2(zk_ctx* ctx, const void* bases, size_t nb, const zk_mpc_group_layout* bl, const void* scalars, size_t ns,
                             const zk_mpc_field_layout* sl, zk_g2_projective out_lanes[2], int* scalars_public) {
    ZK_API_BEGIN(ctx)
    return mpc_msm<2>(ctx, bases, nb, bl, scalars, ns, sl, out_lanes, scalars_public);
    ZK_API_END
}
