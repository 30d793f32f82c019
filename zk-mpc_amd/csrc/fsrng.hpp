// fsrng.hpp -- host-side byte-level primitives of Marlin's Fiat-Shamir transcript and of the share sampler.
//
// Replaces (reference):
//   FiatShamirRng<Blake2s>                         arkworks/marlin/src/rng.rs:11-67  (seed = H(new || old seed), ChaChaRng::from_seed)
//   rand_chacha 0.3.1 ChaChaRng (= ChaCha20Rng)    Cargo.lock:1776 (un-vendored dependency: restated from RFC 8439 section 2.3 with
//                                                  rand_chacha's 64-bit block counter in words 12-13 and the 64-bit stream id 0 in
//                                                  words 14-15; the output stream is the concatenation of blocks 0, 1, 2, ...)
//   blake2 0.9.2 Blake2s (32-byte digest, no key)  Cargo.lock:691 (un-vendored: restated from RFC 7693 section 3)
//   rand 0.8.5 StdRng = ChaCha12Rng                arkworks/std/src/rand_helper.rs:31-39 (test_rng): same generator with 12 rounds
//   BlockRng::{next_u32, next_u64, fill_bytes}     rand_core 0.6: words are consumed in stream order; next_u64 = lo word then hi
//                                                  word; fill_bytes consumes whole words (a partial word's tail is dropped)
// Pinned by the RFC vectors (tests/test_fsrng.py: RFC 7693 appendix B, RFC 8439 2.3.2 through the 32-bit-counter view).
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace zkfs {

// ---- Blake2s-256 (RFC 7693) -------------------------------------------------------------------
struct Blake2s {
    uint32_t h[8];
    uint8_t buf[64];
    size_t buflen = 0;
    uint64_t t = 0;

    static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    static const uint32_t* iv() {
        static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
        return IV;
    }
    Blake2s() {
        for (int i = 0; i < 8; i++) h[i] = iv()[i];
        h[0] ^= 0x01010000u ^ 32u;   // digest length 32, no key, fanout = depth = 1
    }
    void compress(const uint8_t* block, bool last) {
        static const uint8_t S[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint32_t m[16], v[16];
        for (int i = 0; i < 16; i++) m[i] = (uint32_t)block[4 * i] | ((uint32_t)block[4 * i + 1] << 8) | ((uint32_t)block[4 * i + 2] << 16) | ((uint32_t)block[4 * i + 3] << 24);
        for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = iv()[i]; }
        v[12] ^= (uint32_t)t;
        v[13] ^= (uint32_t)(t >> 32);
        if (last) v[14] = ~v[14];
        auto G = [&](int a, int b, int c, int d, uint32_t x, uint32_t y) {
            v[a] = v[a] + v[b] + x; v[d] = rotr(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 12);
            v[a] = v[a] + v[b] + y; v[d] = rotr(v[d] ^ v[a], 8);
            v[c] = v[c] + v[d];     v[b] = rotr(v[b] ^ v[c], 7);
        };
        for (int r = 0; r < 10; r++) {
            const uint8_t* s = S[r];
            G(0, 4, 8, 12, m[s[0]], m[s[1]]);   G(1, 5, 9, 13, m[s[2]], m[s[3]]);
            G(2, 6, 10, 14, m[s[4]], m[s[5]]);  G(3, 7, 11, 15, m[s[6]], m[s[7]]);
            G(0, 5, 10, 15, m[s[8]], m[s[9]]);  G(1, 6, 11, 12, m[s[10]], m[s[11]]);
            G(2, 7, 8, 13, m[s[12]], m[s[13]]); G(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const uint8_t* p, size_t n) {
        while (n) {
            if (buflen == 64) {            // the buffered block is not the last one
                t += 64;
                compress(buf, false);
                buflen = 0;
            }
            size_t k = 64 - buflen;
            if (k > n) k = n;
            memcpy(buf + buflen, p, k);
            buflen += k; p += k; n -= k;
        }
    }
    void final(uint8_t out[32]) {
        t += buflen;
        memset(buf + buflen, 0, 64 - buflen);
        compress(buf, true);
        for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)h[i]; out[4 * i + 1] = (uint8_t)(h[i] >> 8); out[4 * i + 2] = (uint8_t)(h[i] >> 16); out[4 * i + 3] = (uint8_t)(h[i] >> 24); }
    }
    static void digest(const uint8_t* p, size_t n, uint8_t out[32]) {
        Blake2s b;
        b.update(p, n);
        b.final(out);
    }
};

// ---- ChaCha block function (RFC 8439 2.3), `rounds` = 20 (ChaChaRng) or 12 (StdRng) ----------
// words 12..15 of the state are given by the caller: rand_chacha uses (counter lo, counter hi, stream lo, stream hi);
// RFC 8439 uses (32-bit counter, 96-bit nonce).
#define ZKFS_HD
#if defined(__HIPCC__)
#undef ZKFS_HD
#define ZKFS_HD __host__ __device__
#endif
ZKFS_HD inline void chacha_block(const uint32_t key[8], const uint32_t w12_15[4], int rounds, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                      key[4], key[5], key[6], key[7], w12_15[0], w12_15[1], w12_15[2], w12_15[3]};
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define ZKFS_ROTL(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define ZKFS_QR(a, b, c, d)                                  \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = ZKFS_ROTL(x[d], 16);  \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = ZKFS_ROTL(x[b], 12);  \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = ZKFS_ROTL(x[d], 8);   \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = ZKFS_ROTL(x[b], 7);
    for (int r = 0; r < rounds; r += 2) {
        ZKFS_QR(0, 4, 8, 12) ZKFS_QR(1, 5, 9, 13) ZKFS_QR(2, 6, 10, 14) ZKFS_QR(3, 7, 11, 15)
        ZKFS_QR(0, 5, 10, 15) ZKFS_QR(1, 6, 11, 12) ZKFS_QR(2, 7, 8, 13) ZKFS_QR(3, 4, 9, 14)
    }
#undef ZKFS_QR
#undef ZKFS_ROTL
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

// rand_chacha's generator seen as a word stream (BlockRng semantics, see the header comment).
struct ChaChaRng {
    uint32_t key[8];
    uint64_t counter = 0;
    int rounds = 20;
    uint32_t buf[16];
    int idx = 16;
    ChaChaRng() { memset(key, 0, sizeof key); }
    static ChaChaRng from_seed(const uint8_t seed[32], int rounds = 20) {
        ChaChaRng r;
        r.rounds = rounds;
        for (int i = 0; i < 8; i++) r.key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
        return r;
    }
    uint32_t next_u32() {
        if (idx == 16) {
            const uint32_t w[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
            chacha_block(key, w, rounds, buf);
            counter++;
            idx = 0;
        }
        return buf[idx++];
    }
    uint64_t next_u64() {
        const uint64_t lo = next_u32();
        return lo | ((uint64_t)next_u32() << 32);
    }
    void fill_bytes(uint8_t* dst, size_t n) {
        while (n) {
            const uint32_t w = next_u32();
            for (int k = 0; k < 4 && n; k++, n--) *dst++ = (uint8_t)(w >> (8 * k));
        }
    }
};

// FiatShamirRng<Blake2s> (marlin/src/rng.rs:44-67).
struct FiatShamirRng {
    ChaChaRng r;
    uint8_t seed[32];
    static FiatShamirRng from_seed(const uint8_t* bytes, size_t n) {
        FiatShamirRng f;
        Blake2s::digest(bytes, n, f.seed);
        f.r = ChaChaRng::from_seed(f.seed, 20);
        return f;
    }
    void absorb(const uint8_t* bytes, size_t n) {
        Blake2s b;
        b.update(bytes, n);
        b.update(seed, 32);
        b.final(seed);
        r = ChaChaRng::from_seed(seed, 20);
    }
    void absorb(const std::vector<uint8_t>& v) { absorb(v.data(), v.size()); }
};

}  // namespace zkfs
