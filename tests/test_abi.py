"""CPU: the C-ABI library loads and exports every symbol include/zkmpc_hip.h declares, the Python binding
covers exactly that set, and the product fails loudly without a GPU."""
import json
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


def test_header_is_plain_c():
    """include/zkmpc_hip.h is the boundary a cgo / Rust-bindgen / ctypes host reads: it must parse as C99 on its own (every type
    declared before its first use, no C++ in the signatures)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    r = subprocess.run(["gcc", "-x", "c", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", os.path.join(ROOT, "include", "zkmpc_hip.h")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


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


def test_every_int_entry_point_runs_inside_the_boundary_wrapper():
    """Every `extern "C" int` definition of csrc/*.hip starts with ZK_API_BEGIN (ctx.hpp: the calling thread's device becomes
    ctx->device for the call, and no C++ exception leaves the library); the getters that return something else make no HIP
    call and allocate nothing."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import wrap_abi
    import glob
    total, bare = 0, []
    for path in sorted(glob.glob(os.path.join(ROOT, "zk-mpc_amd", "csrc", "*.hip"))):
        src = open(path).read()
        for name, has_ctx, a, b in wrap_abi.entries(src):
            total += 1
            body = src[a + 1:b].lstrip()
            if not body.startswith("ZK_API_BEGIN(ctx)" if has_ctx else "ZK_API_BEGIN_NOCTX"):
                bare.append(name)
        for m in re.finditer(r'^extern "C" (?!int )[^\n]*?(zk_\w+)\(', src, re.M):       # non-int getters
            a = src.index("{", m.end())
            body = src[a:wrap_abi.match_brace(src, a)]
            assert not re.search(r"\bhip[A-Z]|std::|\bnew\b", body), m.group(1)
    assert total >= 125 and not bare, bare


def test_exception_barrier_and_thread_fallback():
    lib = Z.load()
    assert [lib.zk_selftest_exception_barrier(k) for k in range(6)] == [-3, -4, -4, -4, 0, -2]


def test_no_mutable_process_global_state_in_csrc():
    """re-entrant, per-party context, no process-global device state.  What may be static: constants, env knobs read
    once (`static const`), and the dlopen'ed RCCL table (`static const Rccl`, initialised once by the language)."""
    import glob
    for path in glob.glob(os.path.join(ROOT, "zk-mpc_amd", "csrc", "*")):
        if not path.endswith((".hip", ".hpp", ".cuh")):
            continue
        for i, line in enumerate(open(path).read().split("\n")):
            t = line.strip()
            if t.startswith("//") or "static" not in t:
                continue
            t = re.sub(r"//.*", "", t)
            if re.search(r"\bstatic\s+(?!const\b|constexpr\b|inline\b|__device__|ZK_HD\b|_assert)", t) and not re.search(r"\)\s*(const\s*)?\{|\)\s*;|\)\s*$|\(", t):
                raise AssertionError("%s:%d: %s" % (os.path.basename(path), i + 1, line))


class _Host(Z.Context):
    def __init__(self):
        self.lib = Z.load()
        self.h = None


def test_g1_scalar_mul_through_the_endomorphism():
    """hostfield64.hpp::host64_scalar_mul_glv (the Groth16 tail's G1 scalar multiplications for keys made by zk_groth16_setup): k P by
    k = k1 + k2 lambda and phi(x, y) = (beta x, y), against the oracle's double-and-add and the plain host chain -- random scalars,
    the split's edges (0, 1, lambda - 1, lambda, lambda + 1, multiples of lambda, r - 1), the point at infinity."""
    import ctypes as C
    h = _Host()
    rng = O.Prng(77)
    z = 0x8508c00000000001
    lam = z * z - 1
    assert (lam * lam + lam + 1) % O.R_MOD == 0
    P = O.g1_mul(O.G1_GEN, rng.fr())
    pa = h.g1_from_affine(cv.g1_affine_to_array([P])[0])
    ks = [0, 1, 2, 3, lam - 1, lam, lam + 1, 2 * lam, 3 * lam + 2, lam * lam % O.R_MOD, O.R_MOD - 1, O.R_MOD - lam, (1 << 252) + 5]
    ks += [rng.fr() for _ in range(12)]
    for k in ks:
        km = np.ascontiguousarray(cv.fr_to_mont([k])[0])
        out = np.zeros(18, dtype=np.uint64)
        assert h.lib.zk_diag_g1_mul_glv(pa.ctypes.data_as(C.c_void_p), km.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
        assert cv.g1_projective_to_affine(out) == O.g1_mul(P, k), hex(k)
        assert cv.g1_projective_to_affine(out) == cv.g1_projective_to_affine(h.g1_mul(pa, km))
    inf = h.g1_from_affine(np.zeros(12, dtype=np.uint64))
    out = np.ones(18, dtype=np.uint64)
    km = np.ascontiguousarray(cv.fr_to_mont([12345])[0])
    assert h.lib.zk_diag_g1_mul_glv(inf.ctypes.data_as(C.c_void_p), km.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    assert cv.g1_projective_to_affine(out) is None


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
    """fp_mul2 (a b + c d with one Montgomery reduction, the core of the Fq2 product), exact form: with Fq's 14 Montgomery
    digits (RI = 2^406) the value before the single conditional subtraction is below q + 2^355; operands next to q."""
    import ctypes as C
    lib = Z.load()
    q = O.Q_MOD
    to_m = lambda v: np.ascontiguousarray(cv._ints_to_limbs([cv.fq_to_mont_int(v)], 6)[0])
    from_m = lambda a: cv.fq_from_mont_int(cv._limbs_to_ints(a.reshape(1, 6))[0])
    RI = 1 << (29 * 14)
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
                t = (ra * RI * pow(1 << 384, -1, q) % q) * (rb * RI * pow(1 << 384, -1, q) % q) + \
                    (rc * RI * pow(1 << 384, -1, q) % q) * (rd * RI * pow(1 << 384, -1, q) % q)      # on the device's internal residues
                pre = (t + ((-t * pow(q, -1, RI)) % RI) * q) // RI
                assert pre < q + (1 << 355)
                seen.add(pre // q)
    assert seen <= {0, 1}


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


# ---- the lazy domain of Fq (fp29.cuh): representatives in [0, ~7 q], no conditional subtractions -----------------------------

Q = O.Q_MOD
RI14 = 1 << (29 * 14)
EPS = 1 << 354


def _limbs_wide(v):
    """13 limbs: 12 of 29 bits and a top limb holding the rest (< 2^32)."""
    assert 0 <= v < (1 << (29 * 12 + 32))
    return [(v >> (29 * i)) & ((1 << 29) - 1) for i in range(12)] + [v >> (29 * 12)]


def _val(a):
    return sum(int(x) << (29 * i) for i, x in enumerate(a))


def _lazy(op, *vals, n_out=1):
    import ctypes as C
    lib = Z.load()
    inp = np.array([l for v in vals for l in _limbs_wide(v)], dtype=np.uint32)
    out = np.zeros(13 * n_out, dtype=np.uint32)
    assert lib.zk_fq_lazy_raw(op, inp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    return [out[13 * k:13 * k + 13] for k in range(n_out)]


def _normalised(a, top_bits=29):
    return all(int(x) < (1 << 29) for x in a[:12]) and int(a[12]) < (1 << top_bits)


def _worst_columns(tops, split_top=False):
    """Worst-case column sums (in the order the kernels accumulate them) of the product scanning with every low limb at
    2^29 - 1, the given top limbs per operand ((a, b) or (a, b, c, d)) and every Montgomery digit at its maximum (2^29 for the
    first one, fp29.cuh::fp_redc_column).  The m_i p_0 terms stand for the "+ 1" carries the kernels never add: an upper bound."""
    L, LR = 13, 14
    pl = [(Q >> (29 * i)) & ((1 << 29) - 1) for i in range(13)]
    mmax = (1 << 29) - 1
    ops = [[(1 << 29) - 1] * 12 + [t] for t in tops]
    carry, cols = 0, []
    for k in range(LR + L - 1):
        col, up = carry, 0
        for i in range(L):
            j = k - i
            if 0 <= j < L:
                col += ops[0][i] * ops[1][j]
                if len(ops) == 4:
                    t = ops[2][i] * ops[3][j]
                    if split_top and k == 2 * L - 2:
                        col += t & ((1 << 29) - 1)
                        up = t >> 29
                    else:
                        col += t
        for i in range(LR):
            if 0 <= k - i < L:
                col += (mmax + 1 if i == 0 else mmax) * pl[k - i]
        cols.append(col)
        carry = (col >> 29) + up
    return cols


def test_lazy_domain_column_bounds():
    """The 64-bit column accumulator of the product-scanning multiplications cannot overflow for the operand ranges the lazy
    domain uses (ec.cuh::xyzz_madd_lazy, msm_g2pair.hip::madd_p_lazy): every low limb at 2^29 - 1, top limbs at the range
    ends.  Four operands of 7 q DO overflow the top column -- that is what fp_mul2_lazy's TOPSPLIT is for."""
    top = lambda k: ((k * Q + 2 * EPS) >> 348) + 1
    assert max(_worst_columns((top(7), top(7)))) < (1 << 64)                       # P^2, P * X1-wide, R * T in G1
    assert max(_worst_columns((top(3), top(7), top(2), top(2)))) < (1 << 64)       # G1: R T + (2p - PPP) Y1
    assert max(_worst_columns((top(7), top(7), top(1), top(7)))) < (1 << 64)       # G2 even lane of P^2: P0 P0 + (-5 P1) P1
    assert max(_worst_columns((top(5), top(7), top(5), top(7)))) < (1 << 64)       # G2 odd lane of R * T
    assert max(_worst_columns((top(5), top(5), top(5), top(5)))) < (1 << 64)       # G2 odd lane of R^2
    four = _worst_columns((top(7), top(7), top(7), top(7)))
    assert max(four) >= (1 << 64) and max(four[:24]) < (1 << 64)                    # only the top column (24) passes 2^64 ...
    assert max(_worst_columns((top(7), top(7), top(7), top(7)), split_top=True)) < (1 << 64)   # ... and not with the split


def test_lazy_domain_primitives_against_big_integers():
    """mul / sqr / mul2 without the final subtraction, the K p - b carry passes, X3's fused pass, the -5 a estimate and the full
    reduction, on raw limbs: congruent to the exact result modulo q, inside the stated range, limbs normalised -- for operands
    at the range ends (0, q, 2q .. 7q + eps) and random ones."""
    import random
    rnd = random.Random(99)
    ends = [0, 1, Q - 1, Q, Q + EPS - 1, 2 * Q, 3 * Q + EPS, 5 * Q + EPS - 1, 7 * Q + EPS - 1, 7 * Q + 2 * EPS - 1]
    wide = ends + [rnd.randrange(7 * Q + EPS) for _ in range(40)]
    inv = pow(RI14, -1, Q)
    for a in wide:
        for b in wide[::3]:
            (r,) = _lazy(0, a, b)
            assert _val(r) % Q == a * b * inv % Q and _val(r) < Q + EPS and _normalised(r)
        (r,) = _lazy(1, a)
        assert _val(r) % Q == a * a * inv % Q and _val(r) < Q + EPS and _normalised(r)
        (v,) = _lazy(9, a)
        assert (_val(v) + 5 * a) % Q == 0 and 0 < _val(v) <= Q + (Q >> 22) and _normalised(v)
        (c,) = _lazy(7, a)
        assert _val(c) == a % Q and _normalised(c)
    # operands whose low limbs are zero: the first column(s) of the product are already clear, the first Montgomery digit is
    # 2^29 (fp29.cuh::fp_redc_column: never 0, so that every carry is at least 1) and the following columns start from a bare 1
    sparse = [1 << (29 * k) for k in (1, 2, 5, 12)] + [3 << 87, (Q >> 58) << 58, ((7 * Q) >> 29) << 29, 1 << 376]
    for a in sparse + [0]:
        for b in sparse + [0, 1, Q, rnd.randrange(7 * Q)]:
            (r,) = _lazy(0, a, b)
            assert _val(r) % Q == a * b * inv % Q and _val(r) <= Q + EPS and _normalised(r)
            (r,) = _lazy(11 if min(a, b) > 3 * Q else 2, a, b, b, a)          # (four wide operands: the split top column)
            assert _val(r) % Q == 2 * a * b * inv % Q and _val(r) < Q + 2 * EPS and _normalised(r)
        (r,) = _lazy(1, a)
        assert _val(r) % Q == a * a * inv % Q and _val(r) <= Q + EPS and _normalised(r)
    for _ in range(300):
        a, b, c, d = [rnd.choice(wide) for _ in range(4)]
        (r,) = _lazy(2, a, b, c, d)
        assert _val(r) % Q == (a * b + c * d) * inv % Q and _val(r) < Q + 2 * EPS and _normalised(r)
    maxw = 7 * Q + 2 * EPS - 1
    for quad in [(maxw, maxw, maxw, maxw), (maxw, maxw - 5, maxw - 1, maxw)] + [tuple(rnd.randrange(maxw) for _ in range(4)) for _ in range(100)]:
        (r,) = _lazy(11, *quad)                     # split top column: four wide operands
        a, b, c, d = quad
        assert _val(r) % Q == (a * b + c * d) * inv % Q and _val(r) < Q + 2 * EPS and _normalised(r)
    for op, K in ((3, 2), (4, 4), (5, 6)):
        for _ in range(200):
            a = rnd.choice(wide[:12] + [rnd.randrange(Q + EPS)])
            a = min(a, Q + EPS - 1)
            b = rnd.randrange(K * Q + 1) if rnd.random() < 0.8 else rnd.choice([0, K * Q, K * Q - 1, Q])
            (r,) = _lazy(op, a, b)
            assert _val(r) == a + K * Q - b and _normalised(r, 32)
    for _ in range(300):
        rr, ppp, qq = [rnd.choice([0, Q + EPS - 1, Q - 1, 1] + [rnd.randrange(Q + EPS)] * 3) for _ in range(3)]
        (r,) = _lazy(6, rr, ppp, qq)
        assert _val(r) == rr + 4 * Q - ppp - 2 * qq and _normalised(r, 32) and _val(r) < 5 * Q + EPS
    for y in (1, Q - 1, rnd.randrange(Q)):
        (r,) = _lazy(8, y)
        assert _val(r) == Q - y and _normalised(r)


def test_lazy_madd_matches_the_group_law():
    """ec.cuh::xyzz_madd_lazy on the host against the oracle's affine group law: chains of mixed additions started from a
    point, with accumulator coordinates left in the lazy ranges between steps (x < 5q + eps, y, zz, zzz < q + eps), incl.
    the equal-x cases P + P (doubling) and P - P (infinity) and negated points given as q - y."""
    rng = O.Prng(2024)
    to_int = lambda v: v * RI14 % Q                      # internal Montgomery form of the device (RI = 2^406)
    from_int = lambda v: v * pow(RI14, -1, Q) % Q

    def affine_of(c):                                    # canonical XYZZ limbs -> affine point (or None)
        x, y, zz, zzz = [from_int(_val(a)) for a in c]
        if zz == 0:
            return None
        return (x * pow(zz, -1, Q) % Q, y * pow(zzz, -1, Q) % Q)

    pts = [O.g1_mul(O.G1_GEN, rng.fr()) for _ in range(12)]
    acc_pt = pts[0]
    acc = [to_int(pts[0][0]), to_int(pts[0][1]), to_int(1), to_int(1)]
    seq = pts[1:] + [None, "dbl", "neg"]
    for step in seq * 2:
        if step == "dbl":
            q = acc_pt
        elif step == "neg":
            q = (acc_pt[0], Q - acc_pt[1])
        elif step is None:
            q = (pts[3][0], Q - pts[3][1])               # a negated table point, passed as q - y like the kernel does
        else:
            q = step
        out = _lazy(10, *acc, to_int(q[0]), to_int(q[1]), n_out=8)
        lazy, canon = out[:4], out[4:]
        want = O.g1_add(acc_pt, q)
        assert affine_of(canon) == want, step
        vals = [_val(a) for a in lazy]
        assert vals[0] < 5 * Q + EPS and vals[1] < Q + 2 * EPS and vals[2] < Q + EPS and vals[3] < Q + EPS
        assert all(_val(c) == v % Q for c, v in zip(canon, vals))
        if want is None:                                 # restart from a fresh point (the kernel's accumulator would be all-zero)
            acc_pt = pts[5]
            acc = [to_int(pts[5][0]), to_int(pts[5][1]), to_int(1), to_int(1)]
        else:
            acc_pt, acc = want, vals                     # continue from the LAZY representative


# ---- the lazy Fr domain of the NTT butterflies (csrc/frlazy.cuh) -----------------------------------------------------------
RR = O.R_MOD
RI9 = 1 << 261
M29 = (1 << 29) - 1


def _fr_lazy(op, *elems, n_out=1):
    """elems: lists of nine u32 limbs (any limb width)."""
    import ctypes as C
    lib = Z.load()
    inp = np.array([l for e in elems for l in e], dtype=np.uint32)
    out = np.zeros(9 * n_out, dtype=np.uint32)
    assert lib.zk_fr_lazy_raw(op, inp.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    return [[int(x) for x in out[9 * k:9 * k + 9]] for k in range(n_out)]


def _l9(v):
    assert 0 <= v < RI9
    return [(v >> (29 * i)) & M29 for i in range(9)]


def _spread(v, rnd, limb_cap):
    """The same value with limbs pushed above 29 bits where the value allows: limb i borrows from limb i + 1."""
    l = _l9(v)
    for i in range(8):
        k = min(l[i + 1], (limb_cap - l[i]) >> 29)
        k = rnd.randrange(k + 1) if k > 0 else 0
        l[i] += k << 29
        l[i + 1] -= k
    assert sum(x << (29 * i) for i, x in enumerate(l)) == v and all(0 <= x < (1 << 32) for x in l)
    return l


def test_fr_lazy_domain_column_bounds():
    """fp_mul_lazy<FrParams> with a wide left operand: the worst column of the product scanning (every limb of the data
    operand at the largest value a butterfly can produce, 2^31.33 for s0 - s1 + 5r; twiddle limbs and Montgomery digits at
    2^29 - 1) stays below 2^64, and the offsets of frl_sub dominate the limbs they are meant to absorb."""
    pl = _l9(RR)
    worst_a = (1 << 30) - 2 + max(int(x) for x in _fr_consts()["OFF5"][:8])          # s0_i + OFF5_i - 0
    assert worst_a < 2 ** 31.34
    carry, top = 0, 0
    for k in range(17):
        col = carry
        for i in range(9):
            if 0 <= k - i < 9:
                col += worst_a * M29 + (M29 + 1 if i == 0 else M29) * pl[k - i]      # digit 0 may be 2^29 (fp_redc_column)
        top = max(top, col)
        carry = col >> 29
    assert top < (1 << 64)
    c = _fr_consts()
    for name, K, j in (("OFF2", 2, 1), ("OFF3", 3, 1), ("OFF5", 5, 2)):
        off = [int(x) for x in c[name]]
        assert sum(x << (29 * i) for i, x in enumerate(off)) == K * RR
        assert all(j << 29 <= x < (j + 1) << 29 for x in off[:8])
    # top limbs: the subtrahend's top limb never exceeds the offset's (b < 1.03 r / 2.1 r / 4.2 r respectively)
    assert c["OFF2"][8] >= (103 * RR // 100) >> 232 and c["OFF3"][8] >= (21 * RR // 10) >> 232 and c["OFF5"][8] >= (42 * RR // 10) >> 232
    assert sum(int(x) << (29 * i) for i, x in enumerate(c["RC"])) == RI9 - RR and c["MQ"] == (1 << 264) // RR


def test_she_lazy_domain_bounds():
    """The lazy domain of the SHE butterflies (she.hip: f7l_red / f7l_sub / f7l_mul over the MNT4-753 base field, 26 limbs of 29
    bits, Montgomery radix 2^754), modelled on Python integers: the quotient estimate of f7l_red never exceeds a / q and leaves
    less than 2.01 q for every input the butterflies can produce (worst-case limb spreads included); the ranges close -- a
    product of anything below 2.26 q by a reduced twiddle is below 1.89 q, the forward and inverse butterflies map [0, 2.01 q)
    to itself; the offsets of f7l_sub dominate the limbs they absorb; the worst product column stays below 2^64."""
    import random
    from fractions import Fraction
    q = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constants.json")))["mnt4_753_fq"]["MODULUS"]["value"]
    q = int(q)
    src = open(os.path.join(ROOT, "zk-mpc_amd", "csrc", "consts.cuh")).read()
    body = src[src.index("struct Fq753Lazy"):]
    body = body[:body.index("\n};")]
    c = {m.group(1): [int(x.strip().rstrip("u"), 16) for x in m.group(2).split(",")] for m in re.finditer(r"(\w+)\[26\] = \{([^}]*)\}", body)}
    mq = int(re.search(r"MQ = (\d+)u", body).group(1))
    val = lambda limbs: sum(int(x) << (29 * i) for i, x in enumerate(limbs))
    RI = 1 << 754
    assert val(c["RC"]) == RI - q and mq == (1 << 56) // ((q >> 725) + 1)
    for name, K in (("OFF2", 2), ("OFF3", 3)):
        assert val(c[name]) == K * q and all(1 << 29 <= x < 1 << 30 for x in c[name][:25])
    # f7l_sub: the subtrahend's limbs are below 2^29 and its top limb below the offset's (b < 1.89 q for K = 2, < 2.26 q for K = 3)
    assert c["OFF2"][25] >= (189 * q // 100) >> 725 and c["OFF3"][25] >= (226 * q // 100) >> 725

    def red(limbs):                                   # she.hip::f7l_red, instruction for instruction
        t = (limbs[25] + (limbs[24] >> 29)) & 0xffffffff
        k = (t * mq) >> 56
        out, carry = [], 0
        for i in range(26):
            assert limbs[i] + carry < 1 << 32
            acc = k * c["RC"][i] + limbs[i] + carry
            out.append(acc & M29)
            carry = acc >> 29
        return k, out

    def spread(v, rnd, wide):                         # v as 26 limbs, the lower ones pushed up to `wide` bits where v allows it
        limbs = [(v >> (29 * i)) & M29 for i in range(25)] + [v >> 725]
        for i in range(24, -1, -1):
            room = min(((1 << wide) - 1 - limbs[i]) >> 29, limbs[i + 1])
            if room > 0:
                mv = rnd.randint(0, room)
                limbs[i] += mv << 29
                limbs[i + 1] -= mv
        assert val(limbs) == v
        return limbs

    rnd = random.Random(753)
    worst = Fraction(0)
    cases = [0, 1, q - 1, q, 2 * q, 201 * q // 100, 3 * q - 1, 501 * q // 100, 56 * q // 10, 79 * q // 10 - 1]
    cases += [k * q + d for k in range(8) for d in (-1, 0, 1) if k * q + d >= 0]
    cases += [rnd.randrange(79 * q // 10) for _ in range(300)]
    for v in cases:
        for wide in (29, 30, 31):
            limbs = spread(v, rnd, wide)
            k, out = red(limbs)
            r = val(out)
            assert k * q <= v and r == v - k * q and all(x < 1 << 29 for x in out)
            worst = max(worst, Fraction(r, q))
    assert worst < Fraction(201, 100)
    # the ranges close: product of x < 2.26 q by a reduced y is (x y + m q) / RI < x q / RI + q
    B = Fraction(201, 100)
    prod = lambda x: x * Fraction(q, RI) + 1          # in units of q
    assert prod(Fraction(226, 100)) < Fraction(2) and prod(B) < Fraction(189, 100)
    assert B + prod(B) < Fraction(79, 10) and B + 2 < Fraction(79, 10) and B + 3 < Fraction(79, 10)          # red's inputs
    assert 2 * (B * B * Fraction(q, RI) + 1) < Fraction(79, 10)                                              # x0 y0 + x1 y1, lazy operands
    assert (Fraction(226, 100) * q).__floor__() >> 725 < 1 << 29                                             # every limb of a legal operand < 2^29
    # worst column of fp_mul_lazy<Fq753Params>: 26 operand products and 26 reduction products of 29 x 29 bits, plus the carry
    carry = top = 0
    pl = [(q >> (29 * i)) & M29 for i in range(26)]
    for k in range(51):
        col = carry
        for i in range(26):
            if 0 <= k - i < 26:
                col += M29 * M29 + M29 * pl[k - i]
        top = max(top, col)
        carry = col >> 29
    assert top < 1 << 64


def _fr_consts():
    src = open(os.path.join(os.path.dirname(__file__), "..", "zk-mpc_amd", "csrc", "consts.cuh")).read()
    body = src[src.index("struct FrLazy"):]
    body = body[:body.index("\n};")]
    out = {}
    for m in re.finditer(r"(\w+)\[9\] = \{([^}]*)\}", body):
        out[m.group(1)] = [int(x.strip().rstrip("u"), 16) for x in m.group(2).split(",")]
    out["MQ"] = int(re.search(r"MQ = (\d+)u", body).group(1))
    return out


def test_fr_lazy_primitives_against_big_integers():
    """frl_reduce / frl_norm / frl_sub / frl_mul / frl_canon on raw limbs against Python integers: exact values where the
    operation is exact, congruence and the stated range where it reduces; operands at the range ends and with limbs spread
    above 29 bits."""
    import random
    rnd = random.Random(7)
    inv = pow(RI9, -1, RR)
    ends = [0, 1, RR - 1, RR, 2 * RR, 21 * RR // 10, 42 * RR // 10, 84 * RR // 10, 92 * RR // 10 - 1, 16 * RR, 438 * RR, RI9 - 1]
    for v in ends + [rnd.randrange(RI9) for _ in range(200)] + [rnd.randrange(10 * RR) for _ in range(200)]:
        for limbs in (_l9(v), _spread(v, rnd, (1 << 32) - (1 << 10))):
            (r,) = _fr_lazy(0, limbs)
            assert _val(r) % RR == v % RR and _val(r) < 113 * RR // 100 and all(x <= M29 for x in r)
            (c,) = _fr_lazy(6, limbs)
            assert _val(c) == v % RR and all(x <= M29 for x in c)
        lim = _spread(v, rnd, (1 << 32) - 16)
        (n,) = _fr_lazy(1, lim)
        assert _val(n) == v and all(x <= M29 for x in n)
    for op, K, bmax, blimb in ((2, 2, 103 * RR // 100, 1 << 29), (3, 3, 21 * RR // 10, 1 << 29), (4, 5, 42 * RR // 10, 1 << 30)):
        for _ in range(300):
            a = rnd.choice([0, RR, bmax - 1, rnd.randrange(bmax)])
            b = rnd.choice([0, bmax - 1, rnd.randrange(bmax)])
            la = _spread(a, rnd, blimb - 1) if blimb > (1 << 29) else _l9(a)
            lb = _spread(b, rnd, blimb - 1) if blimb > (1 << 29) else _l9(b)
            (r,) = _fr_lazy(op, la, lb)
            assert _val(r) == a + K * RR - b and all(0 <= x < 2 ** 31.34 for x in r)
    for _ in range(300):
        a = rnd.choice([92 * RR // 10 - 1, RI9 - 1, rnd.randrange(RI9), rnd.randrange(10 * RR)])
        w = rnd.choice([RR - 1, 1, rnd.randrange(RR)])
        (r,) = _fr_lazy(5, _spread(a, rnd, int(2 ** 31.33)), _l9(w))
        assert _val(r) % RR == a * w * inv % RR and _val(r) * RI9 <= RR * (RI9 + a) and all(x <= M29 for x in r)


def test_fr_mul32_against_big_integers():
    """frlazy.cuh::fr_mul32 (the 2^5 fix-up of a product of two elements in the reference's form, replacing a second Montgomery
    product in k_vec_op / k_beaver): 32 v mod r, fully reduced, for every v below 2^256 -- the reciprocal it estimates the
    quotient with is exact for every numerator it can meet, and the remainder it leaves needs one subtraction."""
    import random
    rnd = random.Random(32)
    d = (RR >> 240) + 1
    M = (1 << 32) // d + 1
    assert all((n * M) >> 32 == n // d for n in range(1 << 21))                 # t >> 240 < 2^21 for any v < 2^256
    ends = [0, 1, RR - 1, RR, RR + 1, (RR - 1) // 32, RR // 32 + 1, 2 * RR, (1 << 256) - 1, (1 << 253), (1 << 240) - 1, 1 << 240]
    ends += [k * RR // 32 + e for k in range(1, 33) for e in (-1, 0, 1)]        # the quotient boundaries
    for v in ends + [rnd.randrange(RR) for _ in range(3000)] + [rnd.randrange(1 << 256) for _ in range(500)]:
        (r,) = _fr_lazy(10, _l9(v))
        assert _val(r) == 32 * v % RR and all(x <= M29 for x in r), hex(v)


def test_fr_lazy_butterflies():
    """frl_radix4 / frl_radix2 (the sequences ntt.hip runs) against the two DIF levels of radix2/fft.rs:185-307 computed with
    Python integers; inputs anywhere in the stage-input range (< 2.1 r), outputs back inside it."""
    import random
    rnd = random.Random(11)
    inv = pow(RI9, -1, RR)
    hi = 21 * RR // 10
    for it in range(300):
        xs = [rnd.choice([0, hi - 1, RR, rnd.randrange(hi)]) for _ in range(4)]
        wa, wb, wc = [rnd.choice([RR - 1, rnd.randrange(RR)]) for _ in range(3)]
        m = lambda a, w: a * w * inv % RR
        want = [(xs[0] + xs[1] + xs[2] + xs[3]) % RR,
                m(xs[0] + xs[2] - xs[1] - xs[3], wc),
                (m(xs[0] - xs[2], wa) + m(xs[1] - xs[3], wb)) % RR,
                m(m(xs[0] - xs[2], wa) - m(xs[1] - xs[3], wb), wc)]
        ys = _fr_lazy(7, *[_l9(x) for x in xs], _l9(wa), _l9(wb), _l9(wc), n_out=4)
        assert [_val(y) % RR for y in ys] == want
        assert all(_val(y) < hi and all(l <= M29 for l in y) for y in ys)
        # last stage: second-level twiddle 1, outputs stay wide (< 9.2 r, limbs < 2^31.34) for the product that follows
        ys = _fr_lazy(8, *[_l9(x) for x in xs], _l9(wa), _l9(wb), _l9(wc), n_out=4)
        want1 = [want[0], (xs[0] + xs[2] - xs[1] - xs[3]) % RR, want[2], (m(xs[0] - xs[2], wa) - m(xs[1] - xs[3], wb)) % RR]
        assert [_val(y) % RR for y in ys] == want1
        assert all(_val(y) < 92 * RR // 10 and all(l < 2 ** 31.34 for l in y) for y in ys)
        y2 = _fr_lazy(9, _l9(xs[0]), _l9(xs[1]), _l9(wa), n_out=2)
        assert [_val(y) % RR for y in y2] == [(xs[0] + xs[1]) % RR, m(xs[0] - xs[1], wa)]
        assert all(_val(y) < 113 * RR // 100 and all(l <= M29 for l in y) for y in y2)


def test_lazy_add_matches_the_group_law():
    """ec.cuh::xyzz_add_lazy (the bucket reduction's running sums) on the host against the oracle's group law: sums of
    multiples of the generator given in XYZZ form with non-trivial zz / zzz, operands left in the lazy ranges between steps,
    incl. P + P, P - P and the infinity operands."""
    rng = O.Prng(777)
    to_int = lambda v: v * RI14 % Q
    from_int = lambda v: v * pow(RI14, -1, Q) % Q

    def xyzz_of(pt, lam):                                  # affine -> XYZZ with Z = lam (internal form)
        if pt is None:
            return [0, 0, 0, 0]
        zz, zzz = lam * lam % Q, lam * lam * lam % Q
        return [to_int(pt[0] * zz % Q), to_int(pt[1] * zzz % Q), to_int(zz), to_int(zzz)]

    def affine_of(c):
        x, y, zz, zzz = [from_int(_val(a) % Q) for a in c]
        if zz == 0:
            return None
        return (x * pow(zz, -1, Q) % Q, y * pow(zzz, -1, Q) % Q)

    pts = [O.g1_mul(O.G1_GEN, rng.fr()) for _ in range(6)]
    neg = lambda p: (p[0], (Q - p[1]) % Q)
    cases = [(pts[0], pts[1]), (pts[2], pts[2]), (pts[3], neg(pts[3])), (None, pts[4]), (pts[4], None), (None, None)]
    for pa, pb in cases:
        a = xyzz_of(pa, rng.fr() % Q or 1)
        b = xyzz_of(pb, rng.fr() % Q or 1)
        out = _lazy(12, *a, *b, n_out=8)
        assert affine_of(out[4:]) == O.g1_add(pa, pb)
        assert all(_normalised(c) for c in out[4:]) and all(_val(c) < Q for c in out[4:])
    # a chain that keeps its accumulator in the lazy ranges (x < 5q + eps, the rest < q + eps)
    acc_pt, acc = pts[0], xyzz_of(pts[0], 5)
    for k in range(1, 6):
        out = _lazy(12, *[_val(c) if not isinstance(c, int) else c for c in acc], *xyzz_of(pts[k], 3 + k), n_out=8)
        acc_pt = O.g1_add(acc_pt, pts[k])
        acc = out[:4]
        assert _val(acc[0]) < 5 * Q + EPS and all(_val(c) < Q + EPS for c in acc[1:])
        assert affine_of(out[4:]) == acc_pt


# ---- the Rust side of the boundary (bindings/, generated by tools/gen_rust_ffi.py; no Rust toolchain here: kept from rotting) ----
def _rust_prototypes():
    text = open(os.path.join(ROOT, "bindings", "hip_ffi.rs")).read()
    protos = {}
    for m in re.finditer(r"pub fn (zk_\w+)\((.*)\)( -> ([^;]+))?;", text):
        args = re.sub(r"/\*.*?\*/", "", m.group(2)).strip()
        protos[m.group(1)] = ([a.split(":", 1)[1].strip() for a in args.split(", ")] if args else [], (m.group(4) or "").strip())
    return protos


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_generated_rust_bindings_match_header_library_and_ctypes():
    """bindings/hip_ffi.rs is what tools/gen_rust_ffi.py makes of include/zkmpc_hip.h TODAY (--check), and agrees symbol for symbol,
    arity for arity and return kind for return kind with the header's declarations and with the ctypes table the test-suite
    drives the library through (zk-mpc_amd/_lib.py::PROTOTYPES).  What it must override in the reference:
    ec/src/lib.rs:305-318, poly/src/domain/mod.rs:78-190, ff/src/fields/mod.rs:216-220, mpc-net/src/lib.rs:60-64."""
    import ctypes as C
    import subprocess
    import sys
    from zk_mpc_amd import _lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rust = _rust_prototypes()
    assert sorted(rust) == declared_symbols() == sorted(_lib.PROTOTYPES)
    kinds = {"i32": C.c_int, "usize": C.c_size_t, "u32": C.c_uint32, "*const c_char": C.c_char_p, "*mut c_void": C.c_void_p}
    for name, (args, ret) in rust.items():
        restype, argtypes = _lib.PROTOTYPES[name]
        assert len(args) == len(argtypes), name
        if ret.startswith("*") and ret not in kinds:           # a typed handle (zk_pk_query_bases): any pointer-sized restype
            assert restype in (C.c_void_p, C.c_char_p), (name, ret, restype)
        else:
            assert kinds[ret] is restype, (name, ret, restype)
        for a, ct in zip(args, argtypes):                       # pointers stay pointers, integers keep their width class
            is_ptr = a.startswith("*")
            ct_ptr = ct in (C.c_void_p, C.c_char_p) or hasattr(ct, "contents") or (isinstance(ct, type) and issubclass(ct, C._Pointer))
            assert is_ptr == ct_ptr, (name, a, ct)
            if not is_ptr:
                assert {"i32": C.c_int, "u32": C.c_uint32, "usize": C.c_size_t, "u64": C.c_uint64}[a] is ct, (name, a, ct)
    text = open(os.path.join(ROOT, "bindings", "hip_ffi.rs")).read()
    for const in ("ZK_OK", "ZK_ERR_HIP", "ZK_ERR_ARG", "ZK_ERR_NOMEM", "ZK_ERR_STATE", "ZK_ERR_MAC"):
        assert re.search(r"pub const %s: i32 = " % const, text)
    for struct, size in (("ZkFr", "[u64; 4]"), ("ZkFq", "[u64; 6]"), ("ZkFq753", "[u64; 12]")):
        assert re.search(r"pub struct %s \{[^}]*pub l: %s" % (struct, re.escape(size)), text)


def test_rust_overrides_call_the_abi_with_the_declared_arity():
    """bindings/overrides.rs (the trait overrides of INTEGRATION.md section 2 as source): every zk_* call names a declared entry
    point and passes as many arguments as the header declares; the four dispatch points of SURVEY 8(b) are all there."""
    rust = _rust_prototypes()
    src = open(os.path.join(ROOT, "bindings", "overrides.rs")).read()
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    calls = 0
    for m in re.finditer(r"\b(zk_\w+)\(", src):
        name = m.group(1)
        assert name in rust, name
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        assert len(_split_args(src[m.end():i - 1])) == len(rust[name][0]), name
        calls += 1
    assert calls >= 10
    for needed in ("zk_msm_g1", "zk_msm_g2", "zk_fr_fft_in_place", "zk_fr_divide_by_vanishing_on_coset_dev", "zk_fr_batch_product_in_place",
                   "zk_comm_init", "zk_open_sum_fr_dev", "zk_groth16_prove_shared", "all_gather_bytes"):
        assert needed in src, needed
