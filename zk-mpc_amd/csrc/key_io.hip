// key_io.hip -- wire formats of the Groth16 keys and of a KZG10 SRS (arkworks CanonicalSerialize).
#include "../../include/zkmpc_hip.h"
#include "groth16_int.hpp"

using namespace zk;

// ---- arkworks CanonicalSerialize framing of the Groth16 keys and of a KZG10 SRS --------------------------------------
// VerifyingKey  (arkworks/groth16/src/data_structures.rs:43-58):   alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | Vec gamma_abc_g1
// ProvingKey    (data_structures.rs:133-151):                      vk | beta_g1 | delta_g1 | Vec a_query | Vec b_g1_query |
//                                                                  Vec b_g2_query | Vec h_query | Vec l_query
// UniversalParams (poly-commit/src/kzg10/data_structures.rs:40-80; written by save_srs_to_file, src/marlin.rs:371-376):
//                 Vec powers_of_g | BTreeMap<usize, G1> powers_of_gamma_g | h | beta_h | BTreeMap<usize, G2> neg_powers_of_h
// Vec = u64 length + items (serialize/src/lib.rs:263-272), BTreeMap = u64 length + (u64 key, value) pairs (:740-754); points in
// the compressed or uncompressed form of zk_bases_serialize.  Point bytes are produced / parsed on the device.
namespace {

template <class F, class ABI>
int ser_small(zk_ctx* ctx, const Affine<F>* pts, size_t n, int group, int compressed, uint8_t* out) {
    std::vector<ABI> abi(n);
    for (size_t i = 0; i < n; i++) host_aff_to_abi<F>((uint64_t*)&abi[i], pts[i]);
    zk_bases* b = nullptr;
    int rc = group == 1 ? zk_bases_upload_g1(ctx, (const zk_g1_affine*)abi.data(), n, &b) : zk_bases_upload_g2(ctx, (const zk_g2_affine*)abi.data(), n, &b);
    if (rc == ZK_OK) rc = zk_bases_serialize(ctx, b, 0, n, compressed, out);
    zk_bases_free(ctx, b);
    return rc;
}
void put_u64(uint8_t*& p, uint64_t v) { for (int i = 0; i < 8; i++) *p++ = (uint8_t)(v >> (8 * i)); }
bool get_u64(const uint8_t*& p, const uint8_t* end, uint64_t* v) {
    if (end - p < 8) return false;
    *v = 0;
    for (int i = 0; i < 8; i++) *v |= (uint64_t)p[i] << (8 * i);
    p += 8;
    return true;
}
int put_vec(zk_ctx* ctx, uint8_t*& p, const zk_bases* b, int compressed) {
    const size_t n = b ? b->n : 0;
    put_u64(p, n);
    if (n) ZK_TRY(zk_bases_serialize(ctx, b, 0, n, compressed, p));
    p += n * zk_point_serialized_size(b ? b->group : 1, compressed);
    return ZK_OK;
}
int get_points(zk_ctx* ctx, const uint8_t*& p, const uint8_t* end, int group, size_t n, int compressed, zk_bases** out) {
    const size_t bytes = n * zk_point_serialized_size(group, compressed);
    if ((size_t)(end - p) < bytes) ZK_FAIL(ctx, ZK_ERR_ARG, "deserialize: truncated input");
    ZK_TRY(compressed ? zk_bases_deserialize_compressed(ctx, group, p, n, out) : zk_bases_deserialize_uncompressed(ctx, group, p, n, out));
    p += bytes;
    return ZK_OK;
}
int get_vec(zk_ctx* ctx, const uint8_t*& p, const uint8_t* end, int group, int compressed, zk_bases** out) {
    uint64_t n;
    if (!get_u64(p, end, &n)) ZK_FAIL(ctx, ZK_ERR_ARG, "deserialize: truncated input");
    if (n > ((uint64_t)1 << 32)) ZK_FAIL(ctx, ZK_ERR_ARG, "deserialize: implausible vector length");
    return get_points(ctx, p, end, group, (size_t)n, compressed, out);
}
template <class F>
int points_to_host(zk_ctx* ctx, const zk_bases* b, Affine<F>* out) {
    std::vector<uint32_t> w(b->n * 2 * F::WORDS);
    ZK_HIP(ctx, hipMemcpy(w.data(), b->dev, w.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < b->n; i++) out[i] = aff_load<F>(&w[i * 2 * F::WORDS]);
    return ZK_OK;
}

}  // namespace

extern "C" size_t zk_vk_serialized_size(const zk_pk* pk, int compressed) {
    if (!pk) return 0;
    return zk_point_serialized_size(1, compressed) * (1 + (pk->gamma_abc ? pk->gamma_abc->n : 0)) + zk_point_serialized_size(2, compressed) * 3 + 8;
}
extern "C" size_t zk_pk_serialized_size(const zk_pk* pk, int compressed) {
    if (!pk) return 0;
    const size_t g1 = zk_point_serialized_size(1, compressed), g2 = zk_point_serialized_size(2, compressed);
    return zk_vk_serialized_size(pk, compressed) + 2 * g1 + 5 * 8 + g1 * (pk->a->n + pk->b_g1->n + pk->h->n + pk->l->n) + g2 * pk->b_g2->n;
}
extern "C" int zk_vk_serialize(zk_ctx* ctx, const zk_pk* pk, int compressed, uint8_t* out, size_t cap) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !out) return ZK_ERR_ARG;
    if (cap < zk_vk_serialized_size(pk, compressed)) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_vk_serialize: buffer too small");
    uint8_t* p = out;
    ZK_TRY((ser_small<G1Field, zk_g1_affine>(ctx, &pk->alpha_g1, 1, 1, compressed, p)));
    p += zk_point_serialized_size(1, compressed);
    const Affine<G2Field> g2s[3] = {pk->beta_g2, pk->gamma_g2, pk->delta_g2};
    ZK_TRY((ser_small<G2Field, zk_g2_affine>(ctx, g2s, 3, 2, compressed, p)));
    p += 3 * zk_point_serialized_size(2, compressed);
    return put_vec(ctx, p, pk->gamma_abc, compressed);
    ZK_API_END
}
extern "C" int zk_pk_serialize(zk_ctx* ctx, const zk_pk* pk, int compressed, uint8_t* out, size_t cap) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !pk || !out) return ZK_ERR_ARG;
    if (cap < zk_pk_serialized_size(pk, compressed)) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_pk_serialize: buffer too small");
    ZK_TRY(zk_vk_serialize(ctx, pk, compressed, out, cap));
    uint8_t* p = out + zk_vk_serialized_size(pk, compressed);
    const Affine<G1Field> g1s[2] = {pk->beta_g1, pk->delta_g1};
    ZK_TRY((ser_small<G1Field, zk_g1_affine>(ctx, g1s, 2, 1, compressed, p)));
    p += 2 * zk_point_serialized_size(1, compressed);
    for (const zk_bases* q : {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l}) ZK_TRY(put_vec(ctx, p, q, compressed));
    return ZK_OK;
    ZK_API_END
}
// ProvingKey::deserialize / deserialize_uncompressed: the key becomes resident (window multiples, padded l_query) like one
// from zk_pk_upload; gamma_g2 and gamma_abc_g1 are kept for zk_vk_serialize / zk_pk_vk_g2 / zk_pk_download_g1(which = 5).
extern "C" int zk_pk_deserialize(zk_ctx* ctx, const uint8_t* bytes, size_t len, int compressed, zk_pk** out) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !bytes || !out) return ZK_ERR_ARG;
    const uint8_t *p = bytes, *end = bytes + len;
    zk_pk* pk = new zk_pk();
    zk_bases *one1 = nullptr, *three2 = nullptr, *two1 = nullptr;
    int rc = get_points(ctx, p, end, 1, 1, compressed, &one1);
    if (rc == ZK_OK) rc = get_points(ctx, p, end, 2, 3, compressed, &three2);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->gamma_abc);
    if (rc == ZK_OK) rc = get_points(ctx, p, end, 1, 2, compressed, &two1);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->a);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->b_g1);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 2, compressed, &pk->b_g2);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->h);
    if (rc == ZK_OK) rc = get_vec(ctx, p, end, 1, compressed, &pk->l);
    if (rc == ZK_OK && p != end) { ctx->last_error = "zk_pk_deserialize: trailing bytes after the proving key"; rc = ZK_ERR_ARG; }
    if (rc == ZK_OK) {
        Affine<G2Field> g2s[3];
        Affine<G1Field> g1s[2];
        rc = points_to_host<G1Field>(ctx, one1, &pk->alpha_g1);
        if (rc == ZK_OK) rc = points_to_host<G2Field>(ctx, three2, g2s);
        if (rc == ZK_OK) rc = points_to_host<G1Field>(ctx, two1, g1s);
        pk->beta_g2 = g2s[0]; pk->gamma_g2 = g2s[1]; pk->delta_g2 = g2s[2];
        pk->beta_g1 = g1s[0]; pk->delta_g1 = g1s[1];
    }
    for (zk_bases* q : {pk->a, pk->b_g1, pk->b_g2, pk->h, pk->l})
        if (rc == ZK_OK) rc = zk_bases_precompute_auto(ctx, q);
    if (rc == ZK_OK) rc = zk_pk_make_l_pad(ctx, pk);
    if (rc == ZK_OK) rc = first_point<G1Field>(ctx, pk->a, &pk->a0);
    if (rc == ZK_OK) rc = first_point<G1Field>(ctx, pk->b_g1, &pk->b0_g1);
    if (rc == ZK_OK) rc = first_point<G2Field>(ctx, pk->b_g2, &pk->b0_g2);
    zk_bases_free(ctx, one1); zk_bases_free(ctx, three2); zk_bases_free(ctx, two1);
    if (rc != ZK_OK) { zk_pk_free(ctx, pk); return rc; }
    *out = pk;
    return ZK_OK;
    ZK_API_END
}

extern "C" size_t zk_kzg_srs_serialized_size(size_t n_powers_g, size_t n_powers_gamma_g, int compressed) {
    const size_t g1 = zk_point_serialized_size(1, compressed), g2 = zk_point_serialized_size(2, compressed);
    return 8 + n_powers_g * g1 + 8 + n_powers_gamma_g * (8 + g1) + 2 * g2 + 8;
}
// powers_of_gamma_g holds the keys 0 .. n - 1 (KZG10::setup fills 0 ..= max_degree + 1); neg_powers_of_h is written empty
// (setup(.., produce_g2_powers = false), which is what MarlinKZG10::setup asks for: marlin_pc/mod.rs:77).
extern "C" int zk_kzg_srs_serialize(zk_ctx* ctx, const zk_bases* powers_g, const zk_bases* powers_gamma_g, const zk_g2_affine* h,
                                    const zk_g2_affine* beta_h, int compressed, uint8_t* out, size_t cap) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !powers_g || !powers_gamma_g || !h || !beta_h || !out || powers_g->group != 1 || powers_gamma_g->group != 1) return ZK_ERR_ARG;
    if (cap < zk_kzg_srs_serialized_size(powers_g->n, powers_gamma_g->n, compressed)) ZK_FAIL(ctx, ZK_ERR_ARG, "zk_kzg_srs_serialize: buffer too small");
    uint8_t* p = out;
    ZK_TRY(put_vec(ctx, p, powers_g, compressed));
    const size_t g1 = zk_point_serialized_size(1, compressed), n = powers_gamma_g->n;
    std::vector<uint8_t> tmp(n * g1);
    if (n) ZK_TRY(zk_bases_serialize(ctx, powers_gamma_g, 0, n, compressed, tmp.data()));
    put_u64(p, n);
    for (size_t i = 0; i < n; i++) { put_u64(p, i); memcpy(p, &tmp[i * g1], g1); p += g1; }
    const Affine<G2Field> hs[2] = {host_aff_from_abi<G2Field>((const uint64_t*)h), host_aff_from_abi<G2Field>((const uint64_t*)beta_h)};
    ZK_TRY((ser_small<G2Field, zk_g2_affine>(ctx, hs, 2, 2, compressed, p)));
    p += 2 * zk_point_serialized_size(2, compressed);
    put_u64(p, 0);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_kzg_srs_deserialize(zk_ctx* ctx, const uint8_t* bytes, size_t len, int compressed, zk_bases** powers_g,
                                      zk_bases** powers_gamma_g, zk_g2_affine* h, zk_g2_affine* beta_h) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !bytes || !powers_g || !powers_gamma_g || !h || !beta_h) return ZK_ERR_ARG;
    const uint8_t *p = bytes, *end = bytes + len;
    zk_bases *pg = nullptr, *pgg = nullptr, *hh = nullptr;
    int rc = get_vec(ctx, p, end, 1, compressed, &pg);
    uint64_t n = 0;
    if (rc == ZK_OK && !get_u64(p, end, &n)) { ctx->last_error = "zk_kzg_srs_deserialize: truncated input"; rc = ZK_ERR_ARG; }
    const size_t g1 = zk_point_serialized_size(1, compressed);
    if (rc == ZK_OK && (n > ((uint64_t)1 << 32) || (size_t)(end - p) < n * (8 + g1))) { ctx->last_error = "zk_kzg_srs_deserialize: truncated input"; rc = ZK_ERR_ARG; }
    if (rc == ZK_OK) {
        std::vector<uint8_t> tmp(n * g1);
        for (uint64_t i = 0; i < n && rc == ZK_OK; i++) {
            uint64_t key;
            get_u64(p, end, &key);
            if (key != i) { ctx->last_error = "zk_kzg_srs_deserialize: powers_of_gamma_g keys are not 0 .. n - 1"; rc = ZK_ERR_ARG; }
            memcpy(&tmp[i * g1], p, g1);
            p += g1;
        }
        const uint8_t* q = tmp.data();
        if (rc == ZK_OK) rc = get_points(ctx, q, tmp.data() + tmp.size(), 1, (size_t)n, compressed, &pgg);
    }
    if (rc == ZK_OK) rc = get_points(ctx, p, end, 2, 2, compressed, &hh);
    uint64_t nneg = 0;
    if (rc == ZK_OK && (!get_u64(p, end, &nneg) || nneg != 0 || p != end)) {
        ctx->last_error = "zk_kzg_srs_deserialize: neg_powers_of_h must be empty and nothing may follow";
        rc = ZK_ERR_ARG;
    }
    if (rc == ZK_OK) {
        Affine<G2Field> hs[2];
        rc = points_to_host<G2Field>(ctx, hh, hs);
        host_aff_to_abi<G2Field>((uint64_t*)h, hs[0]);
        host_aff_to_abi<G2Field>((uint64_t*)beta_h, hs[1]);
    }
    zk_bases_free(ctx, hh);
    if (rc != ZK_OK) { zk_bases_free(ctx, pg); zk_bases_free(ctx, pgg); return rc; }
    *powers_g = pg;
    *powers_gamma_g = pgg;
    return ZK_OK;
    ZK_API_END
}
