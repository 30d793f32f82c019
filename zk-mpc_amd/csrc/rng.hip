// rng.hip -- random field elements: the share sampler on the device and the byte-exact generators on the host.
//
// Replaces (reference):
//   F::rand(rng) over a share vector                    mpc-algebra/src/share/additive.rs:98-107 (king_share draws N-1 uniform
//                                                       shares per value), ff/src/fields/arithmetic.rs:200-219 (UniformRand)
//   FiatShamirRng<Blake2s>, ChaChaRng, test_rng()       see fsrng.hpp
//
// Device sampler: element i of a vector is ChaCha20(key, block counter = i, stream id) -- 512 bits -- reduced mod r
// (bias < 2^-258; the reference rejection-samples 253 bits, equally uniform).  The key comes from the operating system
// (getrandom) unless the caller supplies one (tests).  HBM-bound: 32 B written per element, ~1.2 k integer instructions.
#include "../../include/zkmpc_hip.h"
#include "devutil.cuh"
#include "fsrng.hpp"
#include "internal.hpp"
#include <sys/random.h>

using namespace zk;

struct zk_rng {
    zkfs::FiatShamirRng fs;     // its ChaChaRng serves the plain generators too (absorb is then never called)
    bool fiat_shamir = false;
};

namespace {

struct ChaKey { uint32_t k[8]; };

__global__ void __launch_bounds__(256) k_fr_random(ChaKey key, uint64_t stream_id, uint64_t first, void* out, size_t n) {
    const Fr k_lo = fp_const<FrParams>(FrParams::RI2), k_hi = fp_const<FrParams>(FrParams::WIDE_HI);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t ctr = first + i;
        const uint32_t w[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
        uint32_t blk[16];
        zkfs::chacha_block(key.k, w, 20, blk);
        // (lo + 2^256 hi) mod r as lo * RI + hi * 2^256 * RI: a uniform residue (any fixed invertible factor keeps it uniform)
        const Fr lo = fp_unpack<FrParams>(blk), hi = fp_unpack<FrParams>(blk + 8);
        fr_store(out, i, fp_add<FrParams>(fp_mul<FrParams>(lo, k_lo), fp_mul<FrParams>(hi, k_hi)));
    }
}

bool fr_words_valid(const uint64_t l[4]) {   // < r  (is_valid, ff/src/fields/macros.rs:255-260)
    static const uint64_t R[4] = {0x0a11800000000001ull, 0x59aa76fed0000001ull, 0x60b44d1e5c37b001ull, 0x12ab655e9a2ca556ull};
    for (int i = 3; i >= 0; i--) {
        if (l[i] < R[i]) return true;
        if (l[i] > R[i]) return false;
    }
    return false;
}

}  // namespace

// out[i] = a uniformly random element of Fr, i < n (device vector in the reference's layout).  key32 = NULL: a fresh key
// from the operating system's CSPRNG; otherwise the 32-byte ChaCha20 key (deterministic: tests only).  Vectors drawn under
// one key must use different stream ids.
extern "C" int zk_fr_random_dev(zk_ctx* ctx, const uint8_t* key32, uint64_t stream_id, void* out_dev, size_t n) {
    ZK_API_BEGIN(ctx)
    if (!ctx || (n && !out_dev)) return ZK_ERR_ARG;
    uint8_t kb[32];
    if (key32) {
        memcpy(kb, key32, 32);
    } else {
        size_t got = 0;
        while (got < 32) {
            ssize_t r = getrandom(kb + got, 32 - got, 0);
            if (r <= 0) ZK_FAIL(ctx, ZK_ERR_STATE, "zk_fr_random_dev: getrandom failed");
            got += (size_t)r;
        }
    }
    ChaKey key;
    for (int i = 0; i < 8; i++) key.k[i] = (uint32_t)kb[4 * i] | ((uint32_t)kb[4 * i + 1] << 8) | ((uint32_t)kb[4 * i + 2] << 16) | ((uint32_t)kb[4 * i + 3] << 24);
    // the key masks secret shares (king_share): it must not outlive the call in this frame.  (The launch copies its
    // arguments; the runtime's copy of the kernel-argument block is beyond reach.)
    auto wipe = [&] {
        volatile uint8_t* p = kb;
        for (int i = 0; i < 32; i++) p[i] = 0;
        volatile uint32_t* q = key.k;
        for (int i = 0; i < 8; i++) q[i] = 0;
    };
    if (!n) { wipe(); return ZK_OK; }
    hipLaunchKernelGGL(k_fr_random, zk_grid(n, 256), 256, 0, ctx->stream, key, stream_id, (uint64_t)0, out_dev, n);
    wipe();
    ZK_HIP(ctx, hipGetLastError());
    return ZK_OK;
    ZK_API_END
}

// ---- byte-level primitives (host) ---------------------------------------------------------------------------------------
extern "C" int zk_blake2s(const uint8_t* data, size_t len, uint8_t out[32]) {
    ZK_API_BEGIN_NOCTX
    if ((len && !data) || !out) return ZK_ERR_ARG;
    zkfs::Blake2s::digest(data, len, out);
    return ZK_OK;
    ZK_API_END
}

extern "C" int zk_chacha_block(const uint8_t key[32], const uint32_t words12_15[4], int rounds, uint8_t out[64]) {
    ZK_API_BEGIN_NOCTX
    if (!key || !words12_15 || !out || rounds <= 0 || (rounds & 1)) return ZK_ERR_ARG;
    uint32_t k[8], o[16];
    for (int i = 0; i < 8; i++) k[i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
    zkfs::chacha_block(k, words12_15, rounds, o);
    for (int i = 0; i < 16; i++) for (int b = 0; b < 4; b++) out[4 * i + b] = (uint8_t)(o[i] >> (8 * b));
    return ZK_OK;
    ZK_API_END
}

// FiatShamirRng::<Blake2s>::from_seed(bytes) (marlin/src/rng.rs:44-57).
extern "C" int zk_fsrng_new(const uint8_t* seed_bytes, size_t len, zk_rng** out) {
    ZK_API_BEGIN_NOCTX
    if ((len && !seed_bytes) || !out) return ZK_ERR_ARG;
    zk_rng* r = new zk_rng();
    r->fs = zkfs::FiatShamirRng::from_seed(seed_bytes, len);
    r->fiat_shamir = true;
    *out = r;
    return ZK_OK;
    ZK_API_END
}
// ChaChaRng::from_seed (rounds = 20) / StdRng::from_seed (rounds = 12: rand 0.8.5) without a transcript.
extern "C" int zk_rng_from_seed(const uint8_t seed[32], int rounds, zk_rng** out) {
    ZK_API_BEGIN_NOCTX
    if (!seed || !out || (rounds != 8 && rounds != 12 && rounds != 20)) return ZK_ERR_ARG;
    zk_rng* r = new zk_rng();
    r->fs.r = zkfs::ChaChaRng::from_seed(seed, rounds);
    memcpy(r->fs.seed, seed, 32);
    *out = r;
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_rng_free(zk_rng* r) { ZK_API_BEGIN_NOCTX delete r; return ZK_OK; ZK_API_END }
// FiatShamirRng::absorb(bytes) (rng.rs:59-67): seed = H(bytes || seed), generator restarted from it.
extern "C" int zk_fsrng_absorb(zk_rng* r, const uint8_t* bytes, size_t len) {
    ZK_API_BEGIN_NOCTX
    if (!r || !r->fiat_shamir || (len && !bytes)) return ZK_ERR_ARG;
    r->fs.absorb(bytes, len);
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_rng_next_u64(zk_rng* r, uint64_t* out) {
    ZK_API_BEGIN_NOCTX
    if (!r || !out) return ZK_ERR_ARG;
    *out = r->fs.r.next_u64();
    return ZK_OK;
    ZK_API_END
}
extern "C" int zk_rng_fill_bytes(zk_rng* r, uint8_t* out, size_t n) {
    ZK_API_BEGIN_NOCTX
    if (!r || (n && !out)) return ZK_ERR_ARG;
    r->fs.r.fill_bytes(out, n);
    return ZK_OK;
    ZK_API_END
}
// u128::rand (rand 0.8.5 Standard: low half first), the opening challenge of Marlin (marlin/src/lib.rs:300).
extern "C" int zk_rng_next_u128(zk_rng* r, uint64_t out[2]) {
    ZK_API_BEGIN_NOCTX
    if (!r || !out) return ZK_ERR_ARG;
    out[0] = r->fs.r.next_u64();
    out[1] = r->fs.r.next_u64();
    return ZK_OK;
    ZK_API_END
}
// F::rand for Fr (ff/src/fields/arithmetic.rs:200-219): four u64 in limb order, the top 3 bits masked away, rejected unless
// below the modulus; the accepted words ARE the element's in-memory (Montgomery) form.
extern "C" int zk_rng_next_fr(zk_rng* r, zk_fr* out) {
    ZK_API_BEGIN_NOCTX
    if (!r || !out) return ZK_ERR_ARG;
    for (;;) {
        uint64_t l[4];
        for (int i = 0; i < 4; i++) l[i] = r->fs.r.next_u64();
        l[3] &= 0xffffffffffffffffull >> 3;
        if (fr_words_valid(l)) { memcpy(out->l, l, 32); return ZK_OK; }
    }
    ZK_API_END
}
// n x Fr::rand in a row (the 3 |H| coefficients of Marlin's mask polynomial are drawn this way, prover.rs:371-376).
extern "C" int zk_rng_fill_fr(zk_rng* r, zk_fr* out, size_t n) {
    ZK_API_BEGIN_NOCTX
    if (!r || (n && !out)) return ZK_ERR_ARG;
    for (size_t i = 0; i < n; i++) {
        const int rc = zk_rng_next_fr(r, out + i);
        if (rc != ZK_OK) return rc;
    }
    return ZK_OK;
    ZK_API_END
}
