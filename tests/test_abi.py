"""CPU: the C-ABI library loads and exports every symbol include/zkmpc_hip.h declares, the Python binding
covers exactly that set, and the product fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

import zkref as O
import zk_mpc_amd as Z
import zk_mpc_amd.convert as cv
from zk_mpc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zkmpc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = Z.load()
    syms = declared_symbols()
    assert len(syms) >= 60
    for s in syms:
        assert hasattr(lib, s), "libzkmpc_hip.so does not export " + s
    assert sorted(_lib.PROTOTYPES) == syms


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Z.ZkError):
        Z.Context(0)


def test_no_oracle_import_in_product():
    """The product package must never reach into oracle/ (it would void every parity claim)."""
    pkg = os.path.join(ROOT, "zk-mpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "zkref" not in src and "oracle/" not in src.replace("oracle/ (", ""), f


class _Host(Z.Context):
    def __init__(self):
        self.lib = Z.load()
        self.h = None


def test_host_group_helpers_against_oracle():
    """The O(1)-per-proof host helpers of the C ABI (same field / curve templates as the kernels)."""
    h = _Host()
    rng = O.Prng(7)
    P, Q = O.g1_mul(O.G1_GEN, rng.fr()), O.g1_mul(O.G1_GEN, rng.fr())
    pa, qa = h.g1_from_affine(cv.g1_affine_to_array([P])[0]), h.g1_from_affine(cv.g1_affine_to_array([Q])[0])
    assert cv.g1_projective_to_affine(h.g1_add(pa, qa)) == O.g1_add(P, Q)
    assert cv.g1_projective_to_affine(h.g1_add(pa, pa)) == O.g1_add(P, P)
    assert cv.g1_projective_to_affine(h.g1_add(pa, h.g1_neg(pa))) is None
    k = rng.fr()
    assert cv.g1_projective_to_affine(h.g1_mul(pa, cv.fr_to_mont([k])[0])) == O.g1_mul(P, k)
    assert h.g1_serialize(pa) == O.g1_serialize(P)
    assert h.g1_serialize(h.g1_from_affine(np.zeros(12, dtype=np.uint64))) == O.g1_serialize(None)
    P2, Q2 = O.g2_mul(O.G2_GEN, rng.fr()), O.g2_mul(O.G2_GEN, rng.fr())
    pa2, qa2 = h.g2_from_affine(cv.g2_affine_to_array([P2])[0]), h.g2_from_affine(cv.g2_affine_to_array([Q2])[0])
    assert cv.g2_projective_to_affine(h.g2_add(pa2, qa2)) == O.g2_add(P2, Q2)
    assert cv.g2_projective_to_affine(h.g2_mul(pa2, cv.fr_to_mont([k])[0])) == O.g2_mul(P2, k)
    assert h.g2_serialize(pa2) == O.g2_serialize(P2)
    for _ in range(20):
        a, b = rng.fr(), rng.fr()
        am, bm = cv.fr_to_mont([a])[0], cv.fr_to_mont([b])[0]
        assert cv.fr_from_mont(h.fr_op("mul", am, bm)) == [a * b % O.R_MOD]
        assert cv.fr_from_mont(h.fr_op("add", am, bm)) == [(a + b) % O.R_MOD]
        assert cv.fr_from_mont(h.fr_op("sub", am, bm)) == [(a - b) % O.R_MOD]


def test_field_add_sub_boundaries():
    """fp_add / fp_sub / the final reduction of fp_mul decide `>= p` from the top 29-bit limb and fall back to an exact
    slow path when that limb is within 1 of p's: hit every side of that decision, for Fr and Fq, on the host build of
    the very templates the kernels use."""
    import ctypes as C
    lib = Z.load()

    def run(fn, a6, b6, n):
        out = np.zeros(n, dtype=np.uint64)
        a6 = np.ascontiguousarray(a6, dtype=np.uint64); b6 = np.ascontiguousarray(b6, dtype=np.uint64)
        assert fn(a6.ctypes.data_as(C.c_void_p), b6.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
        return out

    rng = O.Prng(99)
    for mod, nl, topshift, add, sub, mul, to_m, from_m in (
            (O.R_MOD, 4, 29 * 8, lib.zk_fr_add, lib.zk_fr_sub, lib.zk_fr_mul,
             lambda v: cv.fr_to_mont([v])[0], lambda a: cv.fr_from_mont(a.reshape(1, 4))[0]),
            (O.Q_MOD, 6, 29 * 12, lib.zk_fq_add, lib.zk_fq_sub, lib.zk_fq_mul,
             lambda v: cv._ints_to_limbs([cv.fq_to_mont_int(v)], 6)[0], lambda a: cv.fq_from_mont_int(cv._limbs_to_ints(a.reshape(1, 6))[0]))):
        R = (1 << (64 * nl)) % mod
        Rinv = pow(R, -1, mod)
        ptop = mod >> topshift
        # Montgomery residues (what the limbs hold) with chosen top limbs: pick raw values, convert back to field values
        raws = []
        for t in (0, 1, 2, ptop - 3, ptop - 2, ptop - 1, ptop, (ptop // 2) - 1, ptop // 2, (ptop // 2) + 1):
            for low in (0, 1, (1 << topshift) - 1, rng.fq() % (1 << topshift)):
                v = (t << topshift) | low
                if v < mod:
                    raws.append(v)
        raws += [mod - 1, mod - 2, 0, 1]
        vals = [(x * Rinv) % mod for x in raws]          # field values whose Montgomery residue is the crafted raw
        for x in vals:
            for y in vals[::3]:
                xm, ym = to_m(x), to_m(y)
                assert from_m(run(add, xm, ym, nl)) == (x + y) % mod
                assert from_m(run(sub, xm, ym, nl)) == (x - y) % mod
        for x in vals[::2]:
            for y in vals[::5]:
                assert from_m(run(mul, to_m(x), to_m(y), nl)) == (x * y) % mod


def test_fused_double_product_boundaries():
    """fp_mul2 (a b + c d with one Montgomery reduction, the core of the Fq2 product) leaves up to 2.68 p before its two
    conditional subtractions: operands next to p drive it through every range ([0,p), [p,2p), [2p,2.68p))."""
    import ctypes as C
    lib = Z.load()
    q = O.Q_MOD
    to_m = lambda v: np.ascontiguousarray(cv._ints_to_limbs([cv.fq_to_mont_int(v)], 6)[0])
    from_m = lambda a: cv.fq_from_mont_int(cv._limbs_to_ints(a.reshape(1, 6))[0])
    RI = 1 << (29 * 13)
    R = (1 << 384) % q
    rng = O.Prng(123)
    seen = set()
    # residues (the limbs the kernel sees are x * 2^384 mod q) crafted next to q and across the range
    raws = [q - 1, q - 2, q - 1 - (rng.u64() & 0xFFFF), q // 2, 1, 0] + [rng.fq() for _ in range(6)] + [q - 1 - rng.u64() for _ in range(6)]
    vals = [(x * pow(R, -1, q)) % q for x in raws]
    for a in vals:
        for b in vals[::2]:
            for c in vals[::3]:
                d = vals[(vals.index(a) + 5) % len(vals)]
                out = np.zeros(6, dtype=np.uint64)
                args = [to_m(v) for v in (a, b, c, d)]
                assert lib.zk_fq_mul2(*[x.ctypes.data_as(C.c_void_p) for x in args], out.ctypes.data_as(C.c_void_p)) == 0
                assert from_m(out) == (a * b + c * d) % q
                ra, rb, rc, rd = [(v * R) % q for v in (a, b, c, d)]
                t = ra * rb + rc * rd
                pre = (t + ((-t * pow(q, -1, RI)) % RI) * q) // RI
                seen.add(min(pre // q, 2))
    assert seen == {0, 1, 2}


def test_neg5_almost_range_and_congruence():
    """fp_neg5_almost (the -5 a1 operand of the Fq2 product, one carry pass): V = k q - 5 a with 0 < V <= q (1 + 2^-24),
    every limb below 2^29, for values at the quotient boundaries j q / 5, at the top-limb boundaries and at random."""
    import ctypes as C, random
    lib = Z.load()
    q = O.Q_MOD
    limbs = lambda v: np.array([(v >> (29 * i)) & ((1 << 29) - 1) for i in range(13)], dtype=np.uint32)
    val = lambda a: sum(int(x) << (29 * i) for i, x in enumerate(a))
    rnd = random.Random(5)
    ptop = q >> 348
    cases = [0, 1, 2, q - 1, q - 2, ptop << 348, (ptop << 348) - 1, (ptop - 1) << 348, (1 << 348) - 1, 1 << 348]
    for j in range(1, 6):
        t = ((j * q) // 5) >> 348
        cases += [v for v in [(j * q) // 5 + d for d in range(-3, 4)] if 0 <= v < q]
        cases += [v for v in [((t + dt) << 348) + low for dt in (-1, 0, 1) for low in (0, (1 << 348) - 1)] if 0 <= v < q]
    cases += [rnd.randrange(q) for _ in range(3000)]
    for a in cases:
        out = np.zeros(13, dtype=np.uint32)
        assert lib.zk_fq_neg5_almost_raw(limbs(a).ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
        assert all(int(x) < (1 << 29) for x in out)
        v = val(out)
        assert (v + 5 * a) % q == 0 and 0 < v <= q + (q >> 24)
