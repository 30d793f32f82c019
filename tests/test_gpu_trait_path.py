"""The reference-shaped boundary, composed: examples/host_trait_groth16.cpp is src/groth16.rs:68-183,240-306 written over the
trait-shaped entry points only (zk_fr_fft_in_place x7, zk_fr_batch_product_in_place, zk_fr_divide_by_vanishing_on_coset_in_place,
zk_msm_g1 x4, zk_msm_g2 x1, the host group helpers), with the proving key in HOST vectors.  Its 192 bytes must be the oracle's
known-trapdoor prediction at 2^10, 2^16 and 2^20 -- with the base-table cache (first call: uploads; second: window multiples are
built; from then on: hits), without it, and with the key in a Rust-shaped {x, y, infinity} layout through zk_msm_*_strided.
Then the cache's own contract through ctypes: hit, replacement when a sampled point changes, drop, budget, tiny tables."""
import ctypes as C
import json
import subprocess

import numpy as np
import pytest

import zkref as O
import zk_mpc_amd.convert as cv
from test_host_example import build

pytestmark = pytest.mark.gpu

TD = (2, 3, 5, 7, 11, 1, 1)          # alpha beta gamma delta tau g1_k g2_k: the composer's fixed toxic waste
R, S, W0, W1 = 13, 17, 3, 5


def predicted(log_d):
    n = (1 << log_d) - 2
    if log_d <= 10:
        r1cs, z = O.mul_chain_r1cs(n, W0, W1)
        return O.proof_serialize(*O.predict_proof(r1cs, O.ProvingKeyScalars(r1cs, O.Trapdoor(*TD)), z, R, S))
    import zkref_c as OC
    w = [W0, W1]
    for i in range(n):
        w.append(w[i] * w[i + 1] % O.R_MOD)
    zarr = cv.fr_to_mont([1, w[n + 1]] + w[:n + 1])
    cr = OC.R1cs(2, n + 1, *OC.mul_chain_csr(n))
    mont = lambda v: cv.fr_to_mont([v])[0]
    return OC.groth16_predict(cr, np.stack([mont(v) for v in TD]), zarr, OC.witness_map(cr, zarr, OC.num_threads()), mont(R), mont(S))


def run(exe, log_d, proofs, *mode):
    r = subprocess.run([exe, str(log_d), str(proofs), *mode], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
    return lines[:-1], lines[-1]


@pytest.mark.parametrize("log_d", [10, 16, 20])
def test_trait_path_proof_is_the_predicted_proof(tmp_path, log_d):
    exe = build(tmp_path, "host_trait_groth16")
    want = predicted(log_d).hex()
    proofs, last = run(exe, log_d, 4)
    assert [p["proof"] for p in proofs] == [want] * 4
    c = last["cache"]
    # five slices: first proof 5 misses, then 15 hits; slices of >= 2^16 points got their window multiples on the first hit
    assert (c["misses"], c["hits"], c["entries"], c["replaced"], c["uncached"]) == (5, 15, 5, 0, 0)
    assert c["with_window_multiples"] == 5            # (from 256 points on: small tables are where the host's Horner chain hurts most)
    D = 1 << log_d
    assert c["uploaded_bytes"] == 96 * (2 * (D - 1) + 2 * D) + 192 * D                  # every table crossed PCIe exactly once
    if log_d >= 16:
        assert proofs[-1]["ms"]["lib"] < proofs[0]["ms"]["lib"]


@pytest.mark.parametrize("mode", [("nocache",), ("cache", "strided"), ("nocache", "strided")])
def test_trait_path_without_the_cache_and_with_a_rust_shaped_key(tmp_path, mode):
    exe = build(tmp_path, "host_trait_groth16")
    log_d = 16
    want = predicted(log_d).hex()
    proofs, last = run(exe, log_d, 3, *mode)
    assert [p["proof"] for p in proofs] == [want] * 3
    c = last["cache"]
    if mode[0] == "nocache":
        assert c["entries"] == 0 and c["hits"] == 0 and c["uncached"] == 15 and c["budget"] == 0
    else:
        assert (c["misses"], c["hits"], c["entries"]) == (5, 10, 5) and last["layout"] == "strided"


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _stats(ctx):
    out = np.zeros(10, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_bases_cache_stats(ctx.h, _p(out)))
    return dict(zip(("hits", "misses", "evictions", "replaced", "uncached", "entries", "pre", "resident", "uploaded", "budget"), [int(v) for v in out]))


def test_cache_contract(ctx):
    """Hit on the same slice, replacement when a sampled point changes in place, a sub-slice is its own table, drop, a budget too
    small to keep anything, tables under 256 points never cached -- every result against the discrete-log identity."""
    n = 1000
    rng = O.Prng(515)
    ks = [rng.fr() for _ in range(n)]
    sc = [rng.fr() for _ in range(n)]
    dk = ctx.upload(cv.fr_to_mont(ks))
    tab = ctx.fixed_base(dk.ptr, n, 1, cv.fr_to_mont([1])[0])
    pts = np.ascontiguousarray(tab.download())                            # (n, 12) host table: k_i G
    scal = cv.fr_to_mont(sc)
    want = lambda lo, m, kk=ks: O.g1_mul(O.G1_GEN, sum(s * k for s, k in zip(sc[:m], kk[lo:lo + m])) % O.R_MOD)
    msm = lambda arr, m: cv.g1_projective_to_affine(ctx.multi_scalar_mul_g1(arr, scal[:m]))
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 1 << 30, 1))
    s0 = _stats(ctx)
    assert msm(pts, n) == want(0, n)
    assert msm(pts, n) == want(0, n)
    s1 = _stats(ctx)
    assert (s1["misses"] - s0["misses"], s1["hits"] - s0["hits"], s1["entries"]) == (1, 1, 1)
    sub = pts[100:]                                                       # `&query[1..]`-style sub-slice: another address, another table
    assert msm(sub, 800) == want(100, 800)
    assert _stats(ctx)["entries"] == 2
    # the table changes IN PLACE at a sampled position (index 0 is always sampled): same address, new content -> replaced, right answer
    old0 = pts[0].copy()
    pts[0] = pts[1]
    ks2 = [ks[1]] + ks[1:]
    assert msm(pts, n) == want(0, n, ks2)
    s2 = _stats(ctx)
    assert s2["replaced"] - s1["replaced"] == 1
    pts[0] = old0
    assert msm(pts, n) == want(0, n)
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    assert _stats(ctx)["entries"] == 0 and _stats(ctx)["resident"] == 0
    # a budget that holds one table: the second one pushes the first out
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 1000 * 96 + 10, 1))
    assert msm(pts, n) == want(0, n) and msm(sub, 800) == want(100, 800) and msm(pts, n) == want(0, n)
    s3 = _stats(ctx)
    assert s3["entries"] == 1 and s3["evictions"] >= 2
    # building the window multiples of one table pushes the other one out (the entry list shifts under the hit: ADVICE-style trap)
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    need = 20 * 900 * (256 + 96)                                                     # sub: 900 points, c = 13 -> 20 copies, limb slots + the packed copy
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, need + 900 * 96 + 50_000, 1))           # room for sub's multiples and sub -- not for pts beside them
    assert msm(sub, 800) == want(100, 800) and msm(pts, n) == want(0, n)             # two plain tables resident (sub is the older one)
    assert _stats(ctx)["entries"] == 2
    assert msm(sub, 800) == want(100, 800)                                           # first hit on sub: its multiples need 6.3 MB -> pts goes
    s4 = _stats(ctx)
    assert s4["entries"] == 1 and s4["pre"] == 1
    assert msm(sub, 800) == want(100, 800) and msm(pts, n) == want(0, n)             # sub from its multiples; pts uploaded again
    # smaller than any table: nothing is kept, everything still right
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 1024, 1))
    assert msm(pts, n) == want(0, n)
    assert _stats(ctx)["entries"] == 0
    # tiny tables are never cached
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 1 << 30, 1))
    assert msm(pts[:200], 200) == want(0, 200) and _stats(ctx)["entries"] == 0
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 64 << 30, 1))                        # (the session's context goes on with a roomy cache)
    tab.free(); dk.free()


def test_strided_tables_with_infinity_flags(ctx):
    """zk_msm_g1_strided / _g2_strided on a {x, y, infinity: bool} layout (104 / 200 bytes per point): flagged points contribute
    nothing whatever their coordinate bytes hold; the result equals the packed call's."""
    rng = O.Prng(616)
    for group, words, n in ((1, 12, 700), (2, 24, 300)):
        ks = [rng.fr() for _ in range(n)]
        sc = [rng.fr() for _ in range(n)]
        dk = ctx.upload(cv.fr_to_mont(ks))
        tab = ctx.fixed_base(dk.ptr, n, group, cv.fr_to_mont([1])[0])
        pts = np.ascontiguousarray(tab.download())
        stride = words * 8 + 8
        raw = np.zeros((n, stride), dtype=np.uint8)
        raw[:, :words * 8] = pts.view(np.uint8).reshape(n, words * 8)
        inf = [i for i in range(n) if i % 17 == 3]
        for i in inf:
            raw[i, words * 8] = 1                                          # flagged: the (valid-looking) coordinates must be ignored
        lay = (C.c_size_t * 4)(stride, 0, words * 4, words * 8)
        out = np.zeros(18 if group == 1 else 36, dtype=np.uint64)
        scal = cv.fr_to_mont(sc)
        fn = ctx.lib.zk_msm_g1_strided if group == 1 else ctx.lib.zk_msm_g2_strided
        for _ in range(2):                                                 # miss, then hit
            ctx._ck(fn(ctx.h, _p(raw), n, lay, _p(scal), n, _p(out)))
            e = sum(s * k for i, (s, k) in enumerate(zip(sc, ks)) if i not in inf) % O.R_MOD
            if group == 1:
                assert cv.g1_projective_to_affine(out) == O.g1_mul(O.G1_GEN, e)
            else:
                assert cv.g2_projective_to_affine(out) == O.g2_mul(O.G2_GEN, e)
        # a layout whose fields do not fit the stride is refused
        bad = (C.c_size_t * 4)(words * 8 - 8, 0, words * 4, words * 8)
        assert fn(ctx.h, _p(raw), n, bad, _p(scal), n, _p(out)) != 0
        tab.free(); dk.free()
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
