"""The reference-shaped boundary, composed: examples/host_trait_groth16.cpp is src/groth16.rs:68-183,240-306 written over the
trait-shaped entry points only (zk_fr_fft_in_place x7, zk_fr_batch_product_in_place, zk_fr_divide_by_vanishing_on_coset_in_place,
zk_msm_g1 x4, zk_msm_g2 x1, the host group helpers), with the proving key in HOST vectors.  Its 192 bytes must be the oracle's
known-trapdoor prediction at 2^10, 2^16 and 2^20 -- with the base-table cache (first call: uploads; then verified hits, window
multiples built beside the calls), without it, and with the key in a Rust-shaped {x, y, infinity} layout through zk_msm_*_strided.
examples/host_trait_collab_groth16.cpp is the same over E = MpcPairingEngine: P parties as threads, elements in the enum layouts of
MpcField / MpcG1Affine (both discriminant positions), zk_mpc_* entry points and the transport vtable only; its revealed bytes must
be the prediction on the SUMMED shares.  Then the cache's own contract through ctypes: content-addressed hits, a verified hit that
catches a table rewritten in place at an unsampled point, two tables taking turns at one address, drop, budget, tiny tables."""
import ctypes as C
import json
import subprocess

import os

import numpy as np
import pytest

import zkref as O
import zk_mpc_amd.convert as cv
from test_host_example import build

pytestmark = pytest.mark.gpu

TD = (2, 3, 5, 7, 11, 1, 1)          # alpha beta gamma delta tau g1_k g2_k: the composer's fixed toxic waste
R, S, W0, W1 = 13, 17, 3, 5


def predicted(log_d, R=R, S=S):
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
    # five slices: first proof 5 misses, then 15 hits -- every one of them confirmed against the caller's table in full
    assert (c["misses"], c["hits"], c["entries"], c["replaced"], c["uncached"]) == (5, 15, 5, 0, 0)
    assert c["verified"] == 15
    assert c["with_window_multiples"] == 5 and c["builds"] == 5   # (from 256 points on; built beside the calls, the stragglers by zk_bases_cache_sync)
    D = 1 << log_d
    tables = 96 * (2 * (D - 1) + 2 * D) + 192 * D
    assert c["uploaded_bytes"] == tables                                                 # every table became resident exactly once
    assert c["verified_bytes"] == 3 * tables                                             # ... and was compared once per later proof
    if log_d >= 16:
        assert proofs[-1]["ms"]["lib"] < proofs[0]["ms"]["lib"]
        # no call builds the window multiples of a LARGE table on the caller's time any more (round 5: the second proof of a 2^20 key
        # took 356 - 410 ms; a builder thread works in the caller's gaps now); tables of up to 2^16 points are built in the call that
        # earns them, several levels per launch and one normalisation (2^16: 42 ms for the whole key, once; round 5: 74)
        assert proofs[1]["ms"]["lib"] < (2.5 * proofs[-1]["ms"]["lib"] + 5 if log_d >= 20 else 80)


@pytest.mark.parametrize("mode", [("nocache",), ("cache", "strided"), ("nocache", "strided"), ("cache", "strided", "trust")])
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
        assert c["verified"] == (0 if "trust" in mode else 10)


def run_collab(exe, log_d, proofs, parties, *mode):
    r = subprocess.run([exe, str(log_d), str(proofs), str(parties), *mode], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.strip().splitlines()]
    return lines[:-1], lines[-1]


@pytest.mark.parametrize("log_d,parties,mode", [
    (10, 3, ("additive", "tagfirst")), (16, 3, ("additive", "tagfirst")), (18, 3, ("additive", "tagfirst")),
    (16, 2, ("spdz", "tagfirst")),
    (10, 3, ("additive", "taglast")), (10, 2, ("spdz", "taglast")), (12, 1, ("additive", "tagfirst")), (12, 8, ("additive", "tagfirst")),
    (12, 3, ("additive", "tagfirst", "trust")),
    (20, 1, ("additive", "tagfirst")), (20, 3, ("additive", "tagfirst")),         # BASELINE's own size (three provers on one device)
    (18, 2, ("spdz", "tagfirst")),
])
def test_collaborative_trait_path_reveals_the_predicted_proof(tmp_path, log_d, parties, mode):
    """create_proof::<MpcPairingEngine> unchanged: the composer follows src/groth16.rs:68-183,240-306 over
    Vec<MpcField> / &[MpcG1Affine] in their enum layouts and calls nothing but zk_mpc_fft_in_place x7, zk_mpc_batch_product_in_place
    (Beaver through the vtable), zk_mpc_divide_by_vanishing_on_coset_in_place, zk_mpc_msm_g1 x4 / _g2 x1 and the host group helpers.
    Every party must end with the same bytes (the program checks) and they must be create_proof on the summed shares."""
    exe = build(tmp_path, "host_trait_collab_groth16")
    nproofs = 3
    proofs, last = run_collab(exe, log_d, nproofs, parties, *mode)
    rr, ss = int.from_bytes(bytes.fromhex(last["r"]), "little"), int.from_bytes(bytes.fromhex(last["s"]), "little")
    want = predicted(log_d, rr, ss).hex()
    assert [p["proof"] for p in proofs] == [want] * nproofs
    spdz = mode[0] == "spdz"
    assert last["element_bytes"] == (72 if spdz else 40) and last["tag"] == ("last" if "taglast" in mode else "first")
    c = last["cache"]                                  # party 0's context: its key's five tables, found again by CONTENT in every later proof
    D = 1 << log_d
    if D >= 512:
        assert (c["misses"], c["hits"], c["entries"], c["replaced"], c["uncached"]) == (5, 5 * (nproofs - 1), 5, 0, 0)
        assert c["verified"] == (0 if "trust" in mode else 5 * (nproofs - 1))
        assert c["with_window_multiples"] == 5
    if parties > 1:
        assert proofs[0]["beaver_bytes_sent"] == (4 if spdz else 2) * 32 * D          # two masked operands (SPDZ: and their MAC checks)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _stats(ctx):
    out = np.zeros(10, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_bases_cache_stats(ctx.h, _p(out)))
    return dict(zip(("hits", "misses", "evictions", "replaced", "uncached", "entries", "pre", "resident", "uploaded", "budget"), [int(v) for v in out]))


def _stats2(ctx):
    out = np.zeros(4, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_bases_cache_stats2(ctx.h, _p(out)))
    return dict(zip(("verified", "verified_bytes", "builds", "building"), [int(v) for v in out]))


def test_cache_contract(ctx):
    """Content-addressed hits (another address, another layout: the same table), a sub-slice is its own table, a sampled point
    that changes is another table, an UNSAMPLED point that changes in place is caught by the verified hit (ADVICE r5 medium /
    VERDICT r5 weak 4 ii), trusted mode reads nothing but the sample, two tables that take turns at one address both stay and both
    get window multiples (VERDICT r5 missing 4), drop, budgets, tiny tables -- every result against the discrete-log identity."""
    n = 1000
    rng = O.Prng(515)
    ks = [rng.fr() for _ in range(n)]
    ks_b = [rng.fr() for _ in range(n)]
    sc = [rng.fr() for _ in range(n)]
    dk = ctx.upload(cv.fr_to_mont(ks))
    tab = ctx.fixed_base(dk.ptr, n, 1, cv.fr_to_mont([1])[0])
    pts = np.ascontiguousarray(tab.download())                            # (n, 12) host table: k_i G
    dkb = ctx.upload(cv.fr_to_mont(ks_b))
    tab_b = ctx.fixed_base(dkb.ptr, n, 1, cv.fr_to_mont([1])[0])
    pts_b = np.ascontiguousarray(tab_b.download())
    scal = cv.fr_to_mont(sc)
    want = lambda lo, m, kk=ks: O.g1_mul(O.G1_GEN, sum(s * k for s, k in zip(sc[:m], kk[lo:lo + m])) % O.R_MOD)
    msm = lambda arr, m: cv.g1_projective_to_affine(ctx.multi_scalar_mul_g1(arr, scal[:m]))
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 1 << 30, 1))
    ctx._ck(ctx.lib.zk_bases_cache_trust(ctx.h, 0))
    s0, v0 = _stats(ctx), _stats2(ctx)
    assert msm(pts, n) == want(0, n)
    assert msm(pts, n) == want(0, n)
    assert msm(pts.copy(), n) == want(0, n)                                # the same table at ANOTHER address: a hit (keyed by content)
    s1, v1 = _stats(ctx), _stats2(ctx)
    assert (s1["misses"] - s0["misses"], s1["hits"] - s0["hits"], s1["entries"]) == (1, 2, 1)
    assert v1["verified"] - v0["verified"] == 2 and v1["verified_bytes"] - v0["verified_bytes"] == 2 * 96 * n
    sub = pts[100:]                                                       # `&query[1..]`-style sub-slice: another length, another table
    assert msm(sub, 800) == want(100, 800)
    assert _stats(ctx)["entries"] == 2
    # the table changes IN PLACE at a sampled position (index 0 is always sampled): another fingerprint, another table, right answer
    old0 = pts[0].copy()
    pts[0] = pts[1]
    ks2 = [ks[1]] + ks[1:]
    assert msm(pts, n) == want(0, n, ks2)
    s2 = _stats(ctx)
    assert s2["misses"] - s1["misses"] == 2 and s2["entries"] == 3 and s2["replaced"] == s1["replaced"]
    pts[0] = old0
    assert msm(pts, n) == want(0, n)
    # ... and at a position the fingerprint does NOT sample (64 of 1000 points: indices k * 999 / 63 -- 7 is not one): the candidate
    # hit is compared in full, fails, the entry takes the new content and the sum is the new table's
    assert 7 not in [k * (n - 1) // 63 for k in range(64)]
    old7 = pts[7].copy()
    pts[7] = pts[8]
    ks3 = ks[:7] + [ks[8]] + ks[8:]
    s3 = _stats(ctx)
    assert msm(pts, n) == want(0, n, ks3)
    assert msm(pts, n) == want(0, n, ks3)                                  # (now a verified hit on the new content)
    s4 = _stats(ctx)
    assert s4["replaced"] - s3["replaced"] == 1 and s4["hits"] - s3["hits"] == 2 and s4["entries"] == s3["entries"]
    pts[7] = old7
    assert msm(pts, n) == want(0, n)                                       # and back: caught again
    assert _stats(ctx)["replaced"] - s3["replaced"] == 2
    # the same with window multiples on the entry (they are dropped with the stale content)
    for _ in range(2):
        assert msm(pts, n) == want(0, n)
    ctx._ck(ctx.lib.zk_bases_cache_sync(ctx.h))
    assert _stats(ctx)["pre"] >= 1
    pts[7] = pts[8]
    assert msm(pts, n) == want(0, n, ks3)
    pts[7] = old7
    assert msm(pts, n) == want(0, n)
    # trusted mode: a fingerprint match IS the hit, the caller's table is not read beyond the sample
    ctx._ck(ctx.lib.zk_bases_cache_trust(ctx.h, 1))
    v2 = _stats2(ctx)
    assert msm(pts, n) == want(0, n)
    assert _stats2(ctx)["verified"] == v2["verified"]
    ctx._ck(ctx.lib.zk_bases_cache_trust(ctx.h, 0))
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    assert _stats(ctx)["entries"] == 0 and _stats(ctx)["resident"] == 0
    # two tables of one length taking turns at ONE address (MpcGroup::all_public_or_shared's temporaries: A, B1, A, B1, ...): both
    # stay resident, every call after the first pair is a hit, both get their window multiples
    buf = np.empty_like(pts)
    s5 = _stats(ctx)
    for rnd in range(4):
        buf[:] = pts
        assert msm(buf, n) == want(0, n)
        buf[:] = pts_b
        assert msm(buf, n) == want(0, n, ks_b)
    ctx._ck(ctx.lib.zk_bases_cache_sync(ctx.h))
    s6 = _stats(ctx)
    assert (s6["misses"] - s5["misses"], s6["hits"] - s5["hits"], s6["entries"], s6["pre"], s6["replaced"] - s5["replaced"]) == (2, 6, 2, 2, 0)
    buf[:] = pts
    assert msm(buf, n) == want(0, n)                                       # from the window multiples
    buf[:] = pts_b
    assert msm(buf, n) == want(0, n, ks_b)
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    # a budget that holds one table (its device form and its packed copy): the second one pushes the first out
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 2 * 1000 * 96 + 10, 0))
    assert msm(pts, n) == want(0, n) and msm(sub, 800) == want(100, 800) and msm(pts, n) == want(0, n)
    s7 = _stats(ctx)
    assert s7["entries"] == 1 and s7["evictions"] >= 2
    # building the window multiples of one table pushes the other one out (the entry list shifts under the hit)
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    need = 20 * 900 * (256 + 96) + 900 * 240                                         # sub: 900 points, c = 13 -> 20 copies, limb slots + the packed copy + the build's scratch
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, need + 2 * 900 * 96 + 100_000, 1))      # room for sub's multiples and sub -- not for pts beside them
    assert msm(sub, 800) == want(100, 800) and msm(pts, n) == want(0, n)             # two plain tables resident (sub is the older one)
    assert _stats(ctx)["entries"] == 2
    assert msm(sub, 800) == want(100, 800)                                           # first hit on sub: its multiples need 6.5 MB -> pts goes
    ctx._ck(ctx.lib.zk_bases_cache_sync(ctx.h))
    s8 = _stats(ctx)
    assert s8["entries"] == 1 and s8["pre"] == 1
    assert msm(sub, 800) == want(100, 800) and msm(pts, n) == want(0, n)             # sub from its multiples; pts uploaded again
    # a table whose multiples can never fit evicts nothing
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 2 * (1000 + 900) * 96 + 50_000, 1))
    assert msm(sub, 800) == want(100, 800) and msm(pts, n) == want(0, n) and msm(pts, n) == want(0, n)
    ctx._ck(ctx.lib.zk_bases_cache_sync(ctx.h))
    s9 = _stats(ctx)
    assert s9["entries"] == 2 and s9["pre"] == 0 and s9["evictions"] == s8["evictions"]
    # smaller than any table: nothing is kept, everything still right
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 1024, 1))
    assert msm(pts, n) == want(0, n)
    assert _stats(ctx)["entries"] == 0
    # tiny tables are never cached
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 1 << 30, 1))
    assert msm(pts[:200], 200) == want(0, 200) and _stats(ctx)["entries"] == 0
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 64 << 30, 1))                        # (the session's context goes on with a roomy cache)
    tab.free(); dk.free(); tab_b.free(); dkb.free()


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


def test_cache_fuzz_against_the_discrete_log_identity(ctx):
    """A randomised walk over what a caller can do to the table cache -- six G1 / G2 tables of 300 .. 5 000 points presented packed,
    in the {x, y, infinity} layout and inside MpcGroup wrappers (both discriminant positions), through their own buffers and through
    copies, rewritten IN PLACE at random points (sampled or not) between calls, under a budget that forces evictions while window
    multiples are being built -- every sum against sum s_i k_i * G.  Whatever the cache decides (hit, verified hit, replacement,
    eviction, upload for one call), no call may return the sum over a stale table."""
    import zk_mpc_amd.api as A
    seed = int(os.environ.get("ZK_FUZZ_SEED", "7"))               # (tools/fuzz_trait_path.sh walks other seeds)
    rng = O.Prng(20261003 + seed)
    rs = np.random.RandomState(seed)
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_trust(ctx.h, 0))
    tables = []
    for group, n in ((1, 300), (1, 1000), (1, 1000), (1, 5000), (2, 300), (2, 700)):
        ks = [rng.fr() for _ in range(n)]
        dk = ctx.upload(cv.fr_to_mont(ks))
        tab = ctx.fixed_base(dk.ptr, n, group, cv.fr_to_mont([1])[0])
        pts = np.ascontiguousarray(tab.download())
        tab.free(); dk.free()
        tables.append({"group": group, "n": n, "ks": ks, "pts": pts})
    flay = A.MpcFieldLayout(False, False)
    # room for about three of the six tables with their multiples: evictions and skipped builds on the way
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 12 << 20, 1))
    for step in range(70):
        t = tables[rs.randint(len(tables))]
        group, n, pts, ks = t["group"], t["n"], t["pts"], t["ks"]
        if rs.rand() < 0.35:                                       # rewrite a point in place: point j becomes a copy of point i
            i, j = rs.randint(n), rs.randint(n)
            pts[j] = pts[i]
            ks[j] = ks[i]
        m = n if rs.rand() < 0.7 else rs.randint(n // 2, n)
        sc = [rng.fr() if rs.rand() < 0.9 else 0 for _ in range(m)]
        want = (O.g1_mul(O.G1_GEN, sum(s * k for s, k in zip(sc, ks)) % O.R_MOD) if group == 1
                else O.g2_mul(O.G2_GEN, sum(s * k for s, k in zip(sc, ks)) % O.R_MOD))
        to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
        src = pts if rs.rand() < 0.6 else pts.copy()               # its own buffer, or the same content elsewhere
        form = rs.randint(3)
        scal = cv.fr_to_mont(sc)
        if form == 0:                                              # packed
            got = (ctx.multi_scalar_mul_g1 if group == 1 else ctx.multi_scalar_mul_g2)(src, scal)
        else:                                                      # MpcGroup wrappers (Public), discriminant first / last; scalars as Public MpcField
            glay = A.mpc_group_layout(group, False, tag_last=(form == 2))
            sv = A.MpcVec(flay, m).set([False] * m, scal)
            got, _, all_pub = ctx.mpc_msm(group, A.mpc_wrap_points(src, glay), n, glay, sv, m)
            assert all_pub
        assert to_aff(got) == want, "step %d: table of %d G%d points, form %d" % (step, n, group, form)
    st = _stats(ctx)
    assert st["hits"] > 10 and st["replaced"] > 3 and st["evictions"] + st["misses"] > 6, st
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 64 << 30, 1))


def _spec_stats(ctx):
    out = np.zeros(3, dtype=np.uint64)
    ctx._ck(ctx.lib.zk_msm_speculate_stats(ctx.h, _p(out)))
    return dict(zip(("started", "taken", "dropped"), [int(v) for v in out]))


@pytest.mark.parametrize("n", [900, 40_000])
def test_msms_started_ahead_are_taken_only_for_the_same_scalars(ctx, n):
    """create_proof asks for A, B in G1, B in G2 over ONE `assignment`, one call behind the other (src/groth16.rs:137-160); the
    library learns the succession and starts the next two MSMs ahead.  A result started ahead may only be handed out for the table
    it was computed over AND scalars that are word for word the ones it ran on: a vector that differs at an unsampled element (same
    fingerprint), another order of tables, a table that left the cache or changed content in between -- every sum must be the right
    one, against the discrete-log identity.  n = 900: the jobs run beside the call (small); 40 000: behind it (serial)."""
    rng = O.Prng(8800 + n)
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 8 << 30, 1))
    ctx._ck(ctx.lib.zk_msm_speculate(ctx.h, 1))
    tabs = []
    for group in (1, 1, 2):
        ks = [rng.fr() for _ in range(n)]
        dk = ctx.upload(cv.fr_to_mont(ks))
        tb = ctx.fixed_base(dk.ptr, n, group, cv.fr_to_mont([1])[0])
        tabs.append((group, ks, np.ascontiguousarray(tb.download())))
        tb.free(); dk.free()

    def msm(k, sc, scal):
        group, ks, pts = tabs[k]
        got = (ctx.multi_scalar_mul_g1 if group == 1 else ctx.multi_scalar_mul_g2)(pts, scal)
        e = sum(s * kk for s, kk in zip(sc, ks)) % O.R_MOD
        want = O.g1_mul(O.G1_GEN, e) if group == 1 else O.g2_mul(O.G2_GEN, e)
        assert (cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine)(got) == want, k

    s0 = _spec_stats(ctx)
    for rnd in range(4):                                            # round 0 learns A -> B1 -> B2; rounds 1.. take two results each
        sc = [rng.fr() for _ in range(n)]
        scal = cv.fr_to_mont(sc)
        for k in (0, 1, 2):
            msm(k, sc, scal)
    s1 = _spec_stats(ctx)
    assert s1["started"] - s0["started"] == 6 and s1["taken"] - s0["taken"] == 6 and s1["dropped"] == s0["dropped"], (s0, s1)
    # the same fingerprint, other scalars: element 7 is not one of the 64 sampled ones
    assert 7 not in [k * (n - 1) // 63 for k in range(64)]
    sc = [rng.fr() for _ in range(n)]
    scal = cv.fr_to_mont(sc)
    msm(0, sc, scal)                                                # starts B1, B2 ahead on `sc`
    sc2 = list(sc); sc2[7] = (sc2[7] + 1) % O.R_MOD
    msm(1, sc2, cv.fr_to_mont(sc2))                                 # compared on the device: not the same vector -> dropped, computed afresh
    msm(2, sc2, cv.fr_to_mont(sc2))
    s2 = _spec_stats(ctx)
    # both jobs over `sc` are dropped; the call for B1 on `sc2` may start B2 ahead on `sc2`, which the next call then takes
    assert s2["dropped"] - s1["dropped"] == 2 and s2["taken"] - s1["taken"] <= 1, (s1, s2)
    # another order of tables (A, then B2 directly), twice: dropped, right, and the pattern is given up
    for _ in range(3):
        sc = [rng.fr() for _ in range(n)]
        scal = cv.fr_to_mont(sc)
        msm(0, sc, scal); msm(2, sc, scal); msm(1, sc, scal)
    # ... and earned back: the old order again, and after a few rounds its jobs are started ahead and taken again
    sb = _spec_stats(ctx)
    for _ in range(6):
        sc = [rng.fr() for _ in range(n)]
        scal = cv.fr_to_mont(sc)
        for k in (0, 1, 2):
            msm(k, sc, scal)
    assert _spec_stats(ctx)["taken"] - sb["taken"] >= 2, (sb, _spec_stats(ctx))
    # a table rewritten in place between the call that started a job over it and the call that asks for it
    for _ in range(2):
        sc = [rng.fr() for _ in range(n)]
        scal = cv.fr_to_mont(sc)
        for k in (0, 1, 2):
            msm(k, sc, scal)
    sc = [rng.fr() for _ in range(n)]
    scal = cv.fr_to_mont(sc)
    msm(0, sc, scal)
    group, ks, pts = tabs[1]
    pts[11] = pts[12]; ks[11] = ks[12]                              # (unsampled: the cache's verified hit replaces the entry, the job over the old content must not be used)
    msm(1, sc, scal)
    msm(2, sc, scal)
    # ... and one that left the cache
    sc = [rng.fr() for _ in range(n)]
    scal = cv.fr_to_mont(sc)
    msm(0, sc, scal)
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    msm(1, sc, scal); msm(2, sc, scal)
    # switched off: nothing is started
    ctx._ck(ctx.lib.zk_msm_speculate(ctx.h, 0))
    s3 = _spec_stats(ctx)
    for _ in range(2):
        sc = [rng.fr() for _ in range(n)]
        scal = cv.fr_to_mont(sc)
        for k in (0, 1, 2):
            msm(k, sc, scal)
    assert _spec_stats(ctx)["started"] == s3["started"]
    ctx._ck(ctx.lib.zk_msm_speculate(ctx.h, 1))
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 64 << 30, 1))


def test_fuzz_of_msms_started_ahead(ctx):
    """A randomised walk through what can happen between an MSM that was started ahead and the call it was started for: bursts of calls
    over one scalar vector in a recurring order (learnt), in other orders, with a table skipped; the scalars changed at ONE element
    between two calls of a burst; a table rewritten in place, dropped from the cache, evicted under a tight budget; the switch thrown
    mid-burst; shorter vectors.  Every sum against sum s_i k_i * G -- a result started ahead must never be handed out for anything but
    the table content and the scalars it was computed over."""
    seed = int(os.environ.get("ZK_FUZZ_SEED", "11"))
    rng = O.Prng(20261004 + seed)
    rs = np.random.RandomState(seed)
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_trust(ctx.h, 0))
    ctx._ck(ctx.lib.zk_msm_speculate(ctx.h, 1))
    tables = []
    for group, n in ((1, 1000), (1, 1000), (2, 1000), (1, 1200), (2, 700)):
        ks = [rng.fr() for _ in range(n)]
        dk = ctx.upload(cv.fr_to_mont(ks))
        tab = ctx.fixed_base(dk.ptr, n, group, cv.fr_to_mont([1])[0])
        pts = np.ascontiguousarray(tab.download())
        tab.free(); dk.free()
        tables.append({"group": group, "n": n, "ks": ks, "pts": pts})
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 24 << 20, 1))   # not all five with their multiples: evictions on the way
    orders = [(0, 1, 2)] * 5 + [(3, 0, 2), (0, 2), (1, 0, 4), (0, 1, 2, 3, 4)]
    s0 = _spec_stats(ctx)
    on = True
    for step in range(45):
        order = orders[rs.randint(len(orders))]
        if rs.rand() < 0.3:                                        # a transform's output as the scalars of the MSM right behind it
            a = rs.randint(0, 1 << 62, size=(1024, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
            h = ctx.coset_ifft_in_place(a, 10) if rs.rand() < 0.8 else ctx.ifft_in_place(a, 10)
            if rs.rand() < 0.3:
                h[rs.randint(1024), 0] ^= np.uint64(1)             # ... touched in between
            t = tables[3]                                          # 1 200 points: all 1 024 scalars count
            e = sum(int(s_) * kk for s_, kk in zip(cv.fr_from_mont(h), t["ks"])) % O.R_MOD
            assert cv.g1_projective_to_affine(ctx.multi_scalar_mul_g1(t["pts"], h)) == O.g1_mul(O.G1_GEN, e), "step %d, after a transform" % step
        m = 700 if rs.rand() < 0.8 else rs.randint(300, 700)
        sc = [rng.fr() if rs.rand() < 0.9 else 0 for _ in range(m)]
        for k in order:
            t = tables[k]
            group, n, pts, ks = t["group"], t["n"], t["pts"], t["ks"]
            r = rs.rand()
            if r < 0.12:                                           # the table changes in place (point j becomes a copy of point i)
                i, j = rs.randint(n), rs.randint(n)
                pts[j] = pts[i]; ks[j] = ks[i]
            elif r < 0.24:                                         # the scalars change at one element
                sc = list(sc); j = rs.randint(m); sc[j] = (sc[j] + 1 + rs.randint(5)) % O.R_MOD
            elif r < 0.28:
                ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
            elif r < 0.33:
                on = not on
                ctx._ck(ctx.lib.zk_msm_speculate(ctx.h, 1 if on else 0))
            e = sum(s * kk for s, kk in zip(sc, ks)) % O.R_MOD
            want = O.g1_mul(O.G1_GEN, e) if group == 1 else O.g2_mul(O.G2_GEN, e)
            src = pts if rs.rand() < 0.7 else pts.copy()
            got = (ctx.multi_scalar_mul_g1 if group == 1 else ctx.multi_scalar_mul_g2)(src, cv.fr_to_mont(sc))
            assert (cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine)(got) == want, \
                "step %d, table %d of %s, %d scalars" % (step, k, order, m)
    s1 = _spec_stats(ctx)
    if seed == 11:                                                 # (how often either end is walked is the seed's luck; the default's is known)
        assert s1["taken"] - s0["taken"] > 1 and s1["dropped"] - s0["dropped"] > 1, (s0, s1)
    ctx._ck(ctx.lib.zk_msm_speculate(ctx.h, 1))
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 64 << 30, 1))


@pytest.mark.parametrize("log_n", [9, 15])
def test_the_h_msm_started_at_the_end_of_the_transform(ctx, log_n):
    """`h = witness_map(..)` ends in coset_ifft_in_place(&mut ab) and the very next library call is multi_scalar_mul(&pk.h_query, &h)
    (src/groth16.rs:100-106, 296-305; D - 1 bases against D scalars: the min(len) rule).  The library learns that the output of a
    transform of this kind and size was the next MSM's scalar vector and starts that MSM when the transform ends; the result is
    released only if the scalars the call brings are the transform's output word for word.  Every sum against the discrete-log
    identity on the transform's OWN output; a caller that touches one element in between, or runs another kind of transform last,
    gets the right sum and no job's."""
    N = 1 << log_n
    n = N - 1
    rng = O.Prng(4400 + log_n)
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 8 << 30, 1))
    ctx._ck(ctx.lib.zk_msm_speculate(ctx.h, 1))
    ks = [rng.fr() for _ in range(n)]
    dk = ctx.upload(cv.fr_to_mont(ks))
    tb = ctx.fixed_base(dk.ptr, n, 1, cv.fr_to_mont([1])[0])
    pts = np.ascontiguousarray(tb.download())
    tb.free(); dk.free()

    def check(h, got):
        sc = cv.fr_from_mont(h[:n])
        e = sum(int(s) * kk for s, kk in zip(sc, ks)) % O.R_MOD
        assert cv.g1_projective_to_affine(got) == O.g1_mul(O.G1_GEN, e)

    rs = np.random.RandomState(log_n)
    def rand_vec():
        a = rs.randint(0, 1 << 62, size=(N, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
        return a

    s0 = _spec_stats(ctx)
    for rnd in range(5):                                          # the first round learns, the others take
        h = ctx.coset_ifft_in_place(rand_vec(), log_n)
        check(h, ctx.multi_scalar_mul_g1(pts, h))                 # N scalars against N - 1 bases
    s1 = _spec_stats(ctx)
    assert s1["taken"] - s0["taken"] == 4 and s1["dropped"] == s0["dropped"], (s0, s1)
    # one element changed between the transform and the MSM (not one of the 64 sampled ones)
    h = ctx.coset_ifft_in_place(rand_vec(), log_n)
    assert 5 not in [k * (N - 1) // 63 for k in range(64)] and 5 not in [k * (n - 1) // 63 for k in range(64)]
    h[5, 0] ^= np.uint64(1)
    check(h, ctx.multi_scalar_mul_g1(pts, h))
    s2 = _spec_stats(ctx)
    assert s2["dropped"] - s1["dropped"] == 1 and s2["taken"] == s1["taken"], (s1, s2)
    # another kind of transform last (nothing is started for it), the learned kind before it (its job is dropped)
    h = ctx.coset_ifft_in_place(rand_vec(), log_n)
    h = ctx.ifft_in_place(h, log_n)
    check(h, ctx.multi_scalar_mul_g1(pts, h))
    # ... and the old order again, until it is trusted again
    for rnd in range(5):
        h = ctx.coset_ifft_in_place(rand_vec(), log_n)
        check(h, ctx.multi_scalar_mul_g1(pts, h))
    assert _spec_stats(ctx)["taken"] > s2["taken"]
    ctx._ck(ctx.lib.zk_bases_cache_drop(ctx.h))
    ctx._ck(ctx.lib.zk_bases_cache_config(ctx.h, 64 << 30, 1))
