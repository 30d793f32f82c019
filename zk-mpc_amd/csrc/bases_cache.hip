// bases_cache.hip -- resident copies of the host base slices the trait-shaped MSM entry points are called with.
//
// AffineCurve::multi_scalar_mul(bases: &[Self], scalars: &[Self::ScalarField]) (arkworks/algebra/ec/src/lib.rs:305-314) takes the
// bases as a host slice on EVERY call, and the callers of the hot path pass the same few slices proof after proof: the five
// queries of a ProvingKey (src/groth16.rs:106,110 and calculate_coeff :193 with `&query[1..]`), the powers of a KZG SRS
// (arkworks/poly-commit/src/kzg10/mod.rs:142-205).  Uploading 96 / 192 bytes per point for every call costs more than the MSM
// (a 2^20-point G1 slice: 100 MB, ~2.4 ms of PCIe + an import kernel + hipMalloc / hipFree, against 2.5 ms of arithmetic), and a
// table that lives for one call can never carry window multiples.  So a context keeps what it has been shown:
//   key      (group, host address, length, struct layout) + a 64-bit fingerprint of 64 points spread over the slice (first and last
//            included).  Same key, same fingerprint: the resident table is used and NOTHING but the 64 sampled points is read
//            from the host.  Same key, other fingerprint: the entry is replaced.  Base tables are key material -- `&[G]`,
//            immutable for the callers above; a host that rewrites a table in place between calls, leaving all 64 sampled points
//            as they were, must call zk_bases_cache_drop (or switch the cache off: zk_bases_cache_config(ctx, 0, 0)).
//   budget   bytes of HBM the cache may hold (plain tables + window multiples), least recently used out first; a table larger
//            than the budget is uploaded for its call only, as before.  Default: a quarter of the device memory.
//   multiples  when a slice of >= 256 points is seen for the `precompute_after`-th time after its upload (default: the first hit)
//            its window multiples are built (fixed_base.hip: 13x the memory at 2^20, one bucket set, 13 digits instead of 16):
//            a key's queries are worth it from the second proof on, a one-off slice never pays for them.
// Counters (hits, misses, evictions, bytes uploaded, ...) are read with zk_bases_cache_stats.
#include "../../include/zkmpc_hip.h"
#include "internal.hpp"
#include <string.h>

namespace {

struct CacheEntry {
    int group;
    const void* host;
    size_t n;
    ZkAffineLayout lay;
    bool packed;                  // the ABI's packed form (no layout given)
    uint64_t fp;
    zk_bases* b;
    uint64_t last;
    uint32_t hits;
    bool pre_tried;
    size_t bytes;
};

struct ZkBasesCache {
    std::vector<CacheEntry> e;
    size_t budget = 0;            // 0 until the first use: then a quarter of the device memory
    bool configured = false;
    int precompute_after = 1;     // 0: never
    uint64_t tick = 0;
    uint64_t hits = 0, misses = 0, evictions = 0, uploaded = 0, replaced = 0, uncached = 0;
};

constexpr size_t SAMPLE = 64;
constexpr size_t MIN_CACHED = 256;        // smaller tables are cheaper to upload than to look up
constexpr size_t MAX_ENTRIES = 64;

inline uint64_t mix(uint64_t h, uint64_t v) {
    h ^= v;
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

uint64_t fingerprint(int group, const void* host, size_t n, const ZkAffineLayout* lay) {
    const size_t FE = group == 1 ? 48 : 96;
    const size_t stride = lay ? lay->stride : 2 * FE, ox = lay ? lay->off_x : 0, oy = lay ? lay->off_y : FE;
    uint64_t h = 0xCBF29CE484222325ull ^ (uint64_t)n;
    const size_t S = n < SAMPLE ? n : SAMPLE;
    for (size_t k = 0; k < S; k++) {
        const size_t i = S > 1 ? k * (n - 1) / (S - 1) : 0;
        const char* p = (const char*)host + i * stride;
        uint64_t w;
        for (size_t j = 0; j < FE; j += 8) { memcpy(&w, p + ox + j, 8); h = mix(h, w); }
        for (size_t j = 0; j < FE; j += 8) { memcpy(&w, p + oy + j, 8); h = mix(h, w); }
        if (lay && lay->off_inf != SIZE_MAX) h = mix(h, (uint64_t)(unsigned char)p[lay->off_inf]);
    }
    return h;
}

size_t table_bytes(const zk_bases* b) {
    const size_t pw = b->group == 1 ? 24 : 48;
    size_t bytes = b->n * pw * 4;
    if (b->pre) bytes += (size_t)b->W_pre * b->n * (b->pre_stride ? b->pre_stride : pw) * 4;
    return bytes;
}

ZkBasesCache* cache_of(zk_ctx* ctx) {
    if (!ctx->bases_cache) ctx->bases_cache = new ZkBasesCache();
    ZkBasesCache* c = (ZkBasesCache*)ctx->bases_cache;
    if (!c->configured && c->budget == 0) {
        size_t fr = 0, total = 0;
        c->budget = hipMemGetInfo(&fr, &total) == hipSuccess ? total / 4 : (size_t)16 << 30;
    }
    return c;
}

size_t resident(const ZkBasesCache* c) {
    size_t s = 0;
    for (auto& x : c->e) s += x.bytes;
    return s;
}

void drop_entry(zk_ctx* ctx, ZkBasesCache* c, size_t i) {
    (void)zk_bases_free(ctx, c->e[i].b);
    c->e.erase(c->e.begin() + (ptrdiff_t)i);
}

// least recently used out until `need` more bytes fit (never the entry `keep`)
void make_room(zk_ctx* ctx, ZkBasesCache* c, size_t need, const zk_bases* keep) {
    while (!c->e.empty() && (resident(c) + need > c->budget || c->e.size() >= MAX_ENTRIES)) {
        size_t lru = SIZE_MAX;
        for (size_t i = 0; i < c->e.size(); i++)
            if (c->e[i].b != keep && (lru == SIZE_MAX || c->e[i].last < c->e[lru].last)) lru = i;
        if (lru == SIZE_MAX) return;
        drop_entry(ctx, c, lru);
        c->evictions++;
    }
}

}  // namespace

void zk_bases_cache_free(zk_ctx* ctx) {
    ZkBasesCache* c = (ZkBasesCache*)ctx->bases_cache;
    if (!c) return;
    while (!c->e.empty()) drop_entry(ctx, c, c->e.size() - 1);
    delete c;
    ctx->bases_cache = nullptr;
}

int zk_bases_cache_get(zk_ctx* ctx, int group, const void* host, size_t n, const ZkAffineLayout* layout, const zk_bases** out, bool* temporary) {
    ZkBasesCache* c = cache_of(ctx);
    const size_t plain = n * (group == 1 ? 96 : 192);
    if (c->budget == 0 || n < MIN_CACHED || plain > c->budget) {          // cache off, or not worth / not able to keep
        zk_bases* b = nullptr;
        ZK_TRY(zk_bases_upload_host(ctx, group, host, n, layout, &b));
        c->uncached++;
        c->uploaded += plain;
        *out = b;
        *temporary = true;
        return ZK_OK;
    }
    *temporary = false;
    const uint64_t fp = fingerprint(group, host, n, layout);
    for (size_t i = 0; i < c->e.size(); i++) {
        CacheEntry& x = c->e[i];
        if (x.group != group || x.host != host || x.n != n || x.packed != (layout == nullptr)) continue;
        if (layout && (x.lay.stride != layout->stride || x.lay.off_x != layout->off_x || x.lay.off_y != layout->off_y || x.lay.off_inf != layout->off_inf))
            continue;
        if (x.fp != fp) {                                     // the slice at this address is a different table now
            drop_entry(ctx, c, i);
            c->replaced++;
            break;
        }
        x.hits++;
        x.last = ++c->tick;
        c->hits++;
        zk_bases* const xb = x.b;
        if (c->precompute_after > 0 && !x.pre_tried && x.hits >= (uint32_t)c->precompute_after && n >= ZK_PRECOMP_MIN_POINTS) {
            x.pre_tried = true;                               // (one attempt: a table skipped for lack of room stays plain)
            // W copies: G1 in 256-byte limb slots with the packed copy beside them while they are built, G2 packed
            const size_t need = (size_t)zk_precompute_windows(n) * n * (group == 1 ? 256 + 96 : 192);
            make_room(ctx, c, need, xb);                      // (may erase other entries: `x` is dead from here on)
            if (resident(c) + need <= c->budget) {            // the budget holds for the multiples as well: else the table stays plain
                ZK_TRY(zk_bases_precompute(ctx, xb));
                for (auto& y : c->e) if (y.b == xb) y.bytes = table_bytes(xb);
            }
        }
        *out = xb;
        return ZK_OK;
    }
    c->misses++;
    make_room(ctx, c, plain, nullptr);
    zk_bases* b = nullptr;
    ZK_TRY(zk_bases_upload_host(ctx, group, host, n, layout, &b));
    c->uploaded += plain;
    CacheEntry x;
    x.group = group; x.host = host; x.n = n; x.packed = layout == nullptr;
    x.lay = layout ? *layout : ZkAffineLayout{0, 0, 0, 0};
    x.fp = fp; x.b = b; x.last = ++c->tick; x.hits = 0; x.pre_tried = false; x.bytes = table_bytes(b);
    c->e.push_back(x);
    *out = b;
    return ZK_OK;
}

extern "C" int zk_bases_cache_config(zk_ctx* ctx, size_t budget_bytes, int precompute_after) {
    ZK_API_BEGIN(ctx)
    if (!ctx || precompute_after < 0) return ZK_ERR_ARG;
    ZkBasesCache* c = cache_of(ctx);
    c->configured = true;
    c->budget = budget_bytes;
    c->precompute_after = precompute_after;
    make_room(ctx, c, 0, nullptr);
    if (budget_bytes == 0) while (!c->e.empty()) drop_entry(ctx, c, c->e.size() - 1);
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_bases_cache_drop(zk_ctx* ctx) {
    ZK_API_BEGIN(ctx)
    if (!ctx) return ZK_ERR_ARG;
    ZkBasesCache* c = (ZkBasesCache*)ctx->bases_cache;
    if (c) while (!c->e.empty()) drop_entry(ctx, c, c->e.size() - 1);
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_bases_cache_stats(zk_ctx* ctx, uint64_t out[10]) {
    ZK_API_BEGIN(ctx)
    if (!ctx || !out) return ZK_ERR_ARG;
    ZkBasesCache* c = cache_of(ctx);
    uint64_t pre = 0;
    for (auto& x : c->e) pre += x.b->pre ? 1 : 0;
    out[0] = c->hits; out[1] = c->misses; out[2] = c->evictions; out[3] = c->replaced; out[4] = c->uncached;
    out[5] = c->e.size(); out[6] = pre; out[7] = resident(c); out[8] = c->uploaded; out[9] = c->budget;
    return ZK_OK;
    ZK_API_END
}
