"""Byte-level primitives of the Fiat-Shamir generator (zk-mpc_amd/csrc/fsrng.hpp through the C ABI), pinned to the
published vectors of the algorithms the reference takes from un-vendored crates (blake2 0.9.2, rand_chacha 0.3.1,
rand 0.8.5; Cargo.lock:691,1776,1765), and the oracle's independent Python restatement (oracle/fsrng_ref.py) against the
same vectors and against the product.  Host code only: runs without a GPU."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import fsrng_ref as FR
import zk_mpc_amd as Z
from zk_mpc_amd.api import Rng


def lib():
    return Z.load()


def blake2s(data: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    assert lib().zk_blake2s(data, len(data), out) == 0
    return bytes(out)


def chacha_block(key: bytes, words, rounds: int) -> bytes:
    out = (C.c_uint8 * 64)()
    assert lib().zk_chacha_block(key, (C.c_uint32 * 4)(*words), rounds, out) == 0
    return bytes(out)


def test_blake2s_rfc7693_and_hashlib():
    # RFC 7693 appendix B: BLAKE2s-256("abc")
    want = bytes.fromhex("508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982")
    assert blake2s(b"abc") == want == FR.blake2s(b"abc")
    # every block-boundary case against CPython's implementation (an independent third party)
    for n in (0, 1, 31, 32, 55, 63, 64, 65, 127, 128, 129, 1000, 4096 + 17):
        data = bytes((i * 131 + 7) & 0xff for i in range(n))
        h = hashlib.blake2s(data).digest()
        assert blake2s(data) == h
        assert FR.blake2s(data) == h


def test_chacha20_rfc8439_block():
    # RFC 8439 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00
    key = bytes(range(32))
    words = (1, 0x09000000, 0x4a000000, 0)
    want = bytes.fromhex(
        "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
        "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    assert chacha_block(key, words, 20) == want
    assert FR.chacha_block(key, words, 20) == want


def test_chacha_rng_word_stream():
    """rand_chacha's generator: blocks 0, 1, 2, ... of ChaCha with a 64-bit counter and stream id 0; next_u64 = two
    consecutive words, low first; fill_bytes consumes whole words.  (First block of the all-zero key / zero nonce:
    RFC 8439 appendix A.1 test vector #1 for 20 rounds.)"""
    zero = bytes(32)
    a1 = bytes.fromhex("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                       "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
    assert chacha_block(zero, (0, 0, 0, 0), 20) == a1
    r = Rng.from_seed(zero, 20)
    ref = FR.ChaChaRng(zero, 20)
    words = [int.from_bytes(a1[4 * i:4 * i + 4], "little") for i in range(16)]
    assert r.next_u64() == words[0] | (words[1] << 32) == ref.next_u64()
    assert r.fill_bytes(5) == a1[8:13] == ref.fill_bytes(5)          # consumes words 2 and 3 (the tail of word 3 is dropped)
    assert r.next_u64() == words[4] | (words[5] << 32) == ref.next_u64()
    for _ in range(5):                                                # words 6..15
        assert r.next_u64() == ref.next_u64()
    blk1 = chacha_block(zero, (1, 0, 0, 0), 20)                       # the stream continues with block 1
    assert r.next_u64() == int.from_bytes(blk1[:8], "little") == ref.next_u64()
    assert r.next_u128() == ref.next_u128()


def test_test_rng_is_chacha12_from_the_fixed_seed():
    """ark_std::test_rng() = StdRng::from_seed([1,0,0,0,23,0,0,0,200,1,0,0,210,30,0,...]) and rand 0.8.5's StdRng is
    ChaCha12Rng: the first words equal block 0 of the 12-round function under that key."""
    seed = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)
    blk = chacha_block(seed, (0, 0, 0, 0), 12)
    r = Rng.test_rng()
    ref = FR.test_rng()
    assert r.next_u64() == int.from_bytes(blk[:8], "little") == ref.next_u64()
    # Fr::rand on it: 4 words, top 3 bits cleared, rejection; product and oracle agree on a run of samples
    for _ in range(50):
        got = r.next_fr()
        want = ref.next_fr_words()
        assert [int(x) for x in got] == want
        v = sum(int(x) << (64 * i) for i, x in enumerate(got))
        assert v < FR.R_MOD and v < (1 << 253)


def test_fiat_shamir_rng_reseeding():
    """FiatShamirRng<Blake2s> (marlin/src/rng.rs:44-67): seed = Blake2s(bytes); absorb: seed = Blake2s(bytes || seed);
    after each the generator is ChaCha20 from that seed, counter 0."""
    init = b"MARLIN-2019" + bytes(range(40))
    fs = Rng.fiat_shamir(init)
    ref = FR.FiatShamirRng(init)
    s0 = hashlib.blake2s(init).digest()
    assert fs.next_u64() == int.from_bytes(chacha_block(s0, (0, 0, 0, 0), 20)[:8], "little") == ref.next_u64()
    extra = bytes(range(200, 256)) * 3
    fs.absorb(extra)
    ref.absorb(extra)
    s1 = hashlib.blake2s(extra + s0).digest()
    blk = chacha_block(s1, (0, 0, 0, 0), 20)
    assert fs.next_u64() == int.from_bytes(blk[:8], "little") == ref.next_u64()
    assert [int(x) for x in fs.next_fr()] == ref.next_fr_words()
    assert fs.next_u128() == ref.next_u128()
    with pytest.raises(Exception):
        Rng.from_seed(bytes(32)).absorb(b"x")                         # a plain generator has no transcript
