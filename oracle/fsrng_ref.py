"""oracle/fsrng_ref.py -- TEST INFRASTRUCTURE ONLY (imported by tests/ only).

Pure-Python restatement of the byte-level generators on the reference's Marlin / sampling path:
  FiatShamirRng<Blake2s>                arkworks/marlin/src/rng.rs:11-67
  Blake2s (blake2 0.9.2, un-vendored)   RFC 7693 section 3
  ChaChaRng / ChaCha12 StdRng           rand_chacha 0.3.1 / rand 0.8.5 (un-vendored): RFC 8439 2.3 block function, 64-bit
                                        counter in words 12-13, stream id 0 in words 14-15, BlockRng word-stream semantics
  ark_std::test_rng()                   arkworks/std/src/rand_helper.rs:31-39
  Fr::rand                              arkworks/algebra/ff/src/fields/arithmetic.rs:200-219
Pinned by RFC 7693 appendix B ("abc"), CPython's hashlib.blake2s, RFC 8439 2.3.2 and A.1 (tests/test_fsrng.py).
"""
from __future__ import annotations

import struct

R_MOD = 8444461749428370424248824938781546531375899335154063827935233455917409239041
M32 = 0xFFFFFFFF

_IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
_SIGMA = [
    [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
    [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
    [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
    [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
    [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]


def _rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M32


def _compress(h, block, t, last):
    m = list(struct.unpack("<16I", block))
    v = h[:] + _IV[:]
    v[12] ^= t & M32
    v[13] ^= (t >> 32) & M32
    if last:
        v[14] ^= M32

    def g(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & M32; v[d] = _rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & M32;     v[b] = _rotr(v[b] ^ v[c], 12)
        v[a] = (v[a] + v[b] + y) & M32; v[d] = _rotr(v[d] ^ v[a], 8)
        v[c] = (v[c] + v[d]) & M32;     v[b] = _rotr(v[b] ^ v[c], 7)
    for r in range(10):
        s = _SIGMA[r]
        g(0, 4, 8, 12, m[s[0]], m[s[1]]); g(1, 5, 9, 13, m[s[2]], m[s[3]])
        g(2, 6, 10, 14, m[s[4]], m[s[5]]); g(3, 7, 11, 15, m[s[6]], m[s[7]])
        g(0, 5, 10, 15, m[s[8]], m[s[9]]); g(1, 6, 11, 12, m[s[10]], m[s[11]])
        g(2, 7, 8, 13, m[s[12]], m[s[13]]); g(3, 4, 9, 14, m[s[14]], m[s[15]])
    return [h[i] ^ v[i] ^ v[i + 8] for i in range(8)]


def blake2s(data: bytes) -> bytes:
    h = _IV[:]
    h[0] ^= 0x01010000 ^ 32
    n = len(data)
    off = 0
    while n - off > 64:
        h = _compress(h, data[off:off + 64], off + 64, False)
        off += 64
    h = _compress(h, data[off:].ljust(64, b"\0"), n, True)
    return struct.pack("<8I", *h)


def chacha_block(key: bytes, words12_15, rounds: int) -> bytes:
    s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(struct.unpack("<8I", key)) + [w & M32 for w in words12_15]
    x = s[:]

    def rotl(v, n):
        return ((v << n) | (v >> (32 - n))) & M32

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = rotl(x[b] ^ x[c], 7)
    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return struct.pack("<16I", *[(x[i] + s[i]) & M32 for i in range(16)])


class ChaChaRng:
    def __init__(self, seed: bytes, rounds: int = 20):
        self.key, self.rounds, self.counter, self.buf, self.idx = bytes(seed), rounds, 0, [], 0

    def next_u32(self) -> int:
        if self.idx == len(self.buf):
            blk = chacha_block(self.key, (self.counter & M32, self.counter >> 32, 0, 0), self.rounds)
            self.buf, self.idx = list(struct.unpack("<16I", blk)), 0
            self.counter += 1
        w = self.buf[self.idx]
        self.idx += 1
        return w

    def next_u64(self) -> int:
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)

    def next_u128(self) -> int:
        lo = self.next_u64()
        return lo | (self.next_u64() << 64)

    def fill_bytes(self, n: int) -> bytes:
        out = b""
        while len(out) < n:
            out += struct.pack("<I", self.next_u32())
        return out[:n]

    def next_fr_words(self) -> list:
        """Fr::rand: the accepted 4 x u64 ARE the Montgomery-form words of the element."""
        while True:
            l = [self.next_u64() for _ in range(4)]
            l[3] &= (1 << 61) - 1
            if sum(x << (64 * i) for i, x in enumerate(l)) < R_MOD:
                return l

    def next_fr(self) -> int:
        """The field element (canonical integer) Fr::rand returns."""
        l = self.next_fr_words()
        return sum(x << (64 * i) for i, x in enumerate(l)) * pow(1 << 256, -1, R_MOD) % R_MOD


def test_rng() -> ChaChaRng:
    return ChaChaRng(bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16), 12)


class FiatShamirRng(ChaChaRng):
    def __init__(self, seed_bytes: bytes):
        self.seed = blake2s(seed_bytes)
        super().__init__(self.seed, 20)

    def absorb(self, data: bytes):
        self.seed = blake2s(data + self.seed)
        super().__init__(self.seed, 20)
