"""GPU parity: variable-base MSM on G1 / G2 against the oracle (Pippenger restatement == naive)."""
import numpy as np
import pytest

import zkref as O
import zk_mpc_amd.convert as cv
from helpers import mont1

pytestmark = pytest.mark.gpu


def g1_points(rng, n):
    return [O.g1_mul(O.G1_GEN, rng.fr()) for _ in range(n)]


def g2_points(rng, n):
    return [O.g2_mul(O.G2_GEN, rng.fr()) for _ in range(n)]


def run_g1(ctx, bases, scalars):
    out = ctx.multi_scalar_mul_g1(cv.g1_affine_to_array(bases), cv.fr_to_mont(scalars))
    return cv.g1_projective_to_affine(out)


def run_g2(ctx, bases, scalars):
    out = ctx.multi_scalar_mul_g2(cv.g2_affine_to_array(bases), cv.fr_to_mont(scalars))
    return cv.g2_projective_to_affine(out)


@pytest.mark.parametrize("n", [1, 2, 31, 32, 100, 700])
def test_msm_g1_random(ctx, n):
    rng = O.Prng(300 + n)
    bases, scalars = g1_points(rng, n), [rng.fr() for _ in range(n)]
    assert run_g1(ctx, bases, scalars) == O.msm_pippenger(bases, scalars, O.FqOps)


@pytest.mark.parametrize("n", [1, 3, 40, 200])
def test_msm_g2_random(ctx, n):
    rng = O.Prng(400 + n)
    bases, scalars = g2_points(rng, n), [rng.fr() for _ in range(n)]
    assert run_g2(ctx, bases, scalars) == O.msm_pippenger(bases, scalars, O.Fq2Ops)


def test_msm_edge_cases_g1(ctx):
    rng = O.Prng(55)
    P = g1_points(rng, 6)
    r = O.R_MOD
    cases = [
        ([], []),                                            # empty
        (P[:3], [0, 0, 0]),                                  # all-zero scalars -> infinity
        (P[:3], [1, 1, 1]),                                  # unit scalars (variable_base.rs:45-49 fast path)
        (P[:1] * 50, [7] * 50),                              # equal bases, equal scalars: P+P inside buckets
        ([P[0], O.g1_neg(P[0])], [5, 5]),                    # cancels to infinity
        ([P[0], O.g1_neg(P[0]), P[1]], [9, 9, 3]),
        ([None, P[1], None], [5, 6, 7]),                     # bases at infinity
        (P[:4], [r - 1, r - 2, 1, (r - 1) // 2]),            # extreme scalars
        (P[:5], [1 << 252, (1 << 253) % r, 2 ** 16 - 1, 2 ** 16, 2 ** 15]),  # window-boundary digits
        (P[:6], [3, 5]),                                     # more bases than scalars: min(len)
        (P[:2], [3, 5, 7, 9]),                               # more scalars than bases
    ]
    for bases, scalars in cases:
        n = min(len(bases), len(scalars))
        assert run_g1(ctx, bases, scalars) == O.msm_naive(bases[:n], scalars[:n], O.FqOps), (len(bases), scalars[:4])


def test_msm_edge_cases_g2(ctx):
    rng = O.Prng(56)
    P = g2_points(rng, 4)
    cases = [
        ([], []),
        (P[:2], [0, 0]),
        (P[:1] * 20, [11] * 20),
        ([P[0], O.g2_neg(P[0])], [5, 5]),
        ([None, P[1]], [5, 6]),
        (P[:3], [O.R_MOD - 1, 1, 2]),
    ]
    for bases, scalars in cases:
        assert run_g2(ctx, bases, scalars) == O.msm_naive(bases, scalars, O.Fq2Ops)


def test_msm_witness_like_scalars(ctx):
    """0/1-heavy scalars, as real witnesses are (booleans): one very heavy bucket."""
    rng = O.Prng(57)
    n = 600
    bases = g1_points(rng, 40) * 15
    scalars = [(rng.u64() & 1) if i % 7 else rng.fr() for i in range(n)]
    assert run_g1(ctx, bases, scalars) == O.msm_naive(bases, scalars, O.FqOps)


@pytest.mark.parametrize("group", [1, 2])
def test_msm_very_heavy_bucket(ctx, group):
    """Thousands of unit scalars: one bucket is cut into more than 32 segments, which the fold kernels add up with a
    whole block per bucket (k_fold / k_fold_g2pair, heavy part); a few buckets of 2 .. 32 segments take the light part."""
    rng = O.Prng(59 + group)
    n = 3000
    if group == 1:
        pts, run, ops = g1_points(rng, 25), run_g1, O.FqOps
    else:
        pts, run, ops = g2_points(rng, 25), run_g2, O.Fq2Ops
    bases = pts * (n // 25)
    scalars = [rng.fr() if i % 97 == 0 else (3 if i % 5 == 0 else 1) for i in range(n)]
    assert run(ctx, bases, scalars) == O.msm_naive(bases, scalars, ops)


def test_fixed_base_and_resident_msm(ctx):
    """zk_fixed_base_* (setup-side) against k_i * G, then a resident-bases MSM with an offset."""
    rng = O.Prng(58)
    n = 300
    ks = [rng.fr() for _ in range(n)]
    ks[0], ks[1] = 0, 1
    gk = rng.fr()
    dk = ctx.upload(cv.fr_to_mont(ks))
    b1 = ctx.fixed_base(dk.ptr, n, 1, cv.fr_to_mont([gk])[0])
    got = cv.g1_array_to_affine(b1.download())
    g = O.g1_mul(O.G1_GEN, gk)
    assert got == [O.g1_mul(g, k) for k in ks]
    b2 = ctx.fixed_base(dk.ptr, 40, 2, cv.fr_to_mont([gk])[0])
    g2 = O.g2_mul(O.G2_GEN, gk)
    assert cv.g2_array_to_affine(b2.download()) == [O.g2_mul(g2, k) for k in ks[:40]]
    sc = [rng.fr() for _ in range(n - 1)]
    ds = ctx.upload(cv.fr_to_mont(sc))
    out = cv.g1_projective_to_affine(ctx.msm_dev(b1, 1, ds.ptr, n - 1))
    # sum sc_i * ks_{i+1} * g
    e = sum(s * k for s, k in zip(sc, ks[1:])) % O.R_MOD
    assert out == O.g1_mul(g, e)


@pytest.mark.parametrize("log_n", [16, 20])
def test_msm_large_discrete_log_check(ctx, log_n):
    """Full-size check without a CPU MSM: bases = k_i * G (device fixed-base), so
    sum s_i (k_i G) must equal (sum s_i k_i mod r) * G -- an Fr inner product on the host."""
    n = 1 << log_n
    rs = np.random.RandomState(7 + log_n)

    def rand_mont(m):
        a = rs.randint(0, 1 << 62, size=(m, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a

    km, sm = rand_mont(n), rand_mont(n)
    dk, ds = ctx.upload(km), ctx.upload(sm)
    one = cv.fr_to_mont([1])[0]
    for group in (1, 2):
        m = n          # G2 at the full 2^20 as well: c = 20, k_accum_g2pair / k_reduce_g2pair with 16 virtual windows (the benched shape)
        bases = ctx.fixed_base(dk.ptr, m, group, one)
        out = ctx.msm_dev(bases, 0, ds.ptr, m)
        # inner product on the device too (vector mul), summed on the host in Python ints
        prod = ctx.alloc(m * 32)
        ctx.fr_vec_op_dev(0, dk.ptr, ds.ptr, prod.ptr, m)
        pr = ctx.download(prod, (m, 4))
        # sum of Montgomery residues is the Montgomery residue of the sum
        tot = 0
        for j in range(4):
            tot += int(pr[:, j].astype(object).sum()) << (64 * j)
        e = cv.fr_from_mont(cv.fr_raw([tot % O.R_MOD]))[0]
        to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
        want = O.g1_mul(O.G1_GEN, e) if group == 1 else O.g2_mul(O.G2_GEN, e)
        assert to_aff(out) == want
        # the same table with window multiples: one bucket set of 2^(c-1) buckets, cut into 2^15-bucket slices for the
        # reduce when c > 16 (2^20 points: c = 20, 16 slices; 2^18 points: c = 17, 2 slices)
        bases.precompute()
        c = ctx.lib.zk_bases_window_bits(bases.h)
        assert c == (20 if m >= (1 << 19) else 17 if m >= (1 << 16) else c) and c >= 13
        assert to_aff(ctx.msm_dev(bases, 0, ds.ptr, m)) == want
        assert to_aff(ctx.msm_dev(bases, 3, ds.ptr, m - 3)) == to_aff(ctx.msm_dev(bases, 3, ds.ptr, m - 3))
        bases.free()
        prod.free()


@pytest.mark.parametrize("group,logs", [(1, range(3, 20)), (2, range(3, 17))])
def test_msm_every_window_width(ctx, group, logs):
    """Every shape of the bucket grid of msm_reduce.cuh: n = 2^k - 1 walks the plan through c = 4 ... 16 (bucket sets of
    2^3 ... 2^15: row / column splits with rl = cl and rl = cl + 1, one to 64 windows, K x TW partial counts from 1 to 4) and,
    for tables that carry window multiples, through merged bucket sets up to 2^18; checked by the discrete-log identity."""
    rs = np.random.RandomState(4242 + group)
    top = 1 << (max(logs))
    km = rs.randint(0, 1 << 62, size=(top, 4), dtype=np.uint64); km[:, 3] &= np.uint64((1 << 60) - 1)
    sm = rs.randint(0, 1 << 62, size=(top, 4), dtype=np.uint64); sm[:, 3] &= np.uint64((1 << 60) - 1)
    dk, ds = ctx.upload(km), ctx.upload(sm)
    prod = ctx.alloc(top * 32)
    ctx.fr_vec_op_dev(0, dk.ptr, ds.ptr, prod.ptr, top)
    pr = ctx.download(prod, (top, 4))
    one = cv.fr_to_mont([1])[0]
    bases = ctx.fixed_base(dk.ptr, top, group, one)
    to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
    mul = (lambda e: O.g1_mul(O.G1_GEN, e)) if group == 1 else (lambda e: O.g2_mul(O.G2_GEN, e))

    def want(m):
        tot = sum(int(pr[:m, j].astype(object).sum()) << (64 * j) for j in range(4))
        return mul(cv.fr_from_mont(cv.fr_raw([tot % O.R_MOD]))[0])
    for k in logs:
        m = (1 << k) - 1
        assert to_aff(ctx.msm_dev(bases, 0, ds.ptr, m)) == want(m), (group, k)
    bases.precompute()                                   # one merged bucket set from here on (n >= 4096)
    for k in [x for x in logs if x >= 12]:
        m = (1 << k) - 1
        assert to_aff(ctx.msm_dev(bases, 0, ds.ptr, m)) == want(m), (group, k, "window multiples")
    bases.free(); prod.free(); dk.free(); ds.free()


@pytest.mark.parametrize("group,n", [(1, 256), (1, 300), (1, 1023), (1, 2047), (1, 4095), (2, 256), (2, 1000), (2, 3000)])
def test_window_multiples_of_small_tables(ctx, group, n):
    """Round 5: tables of 256 .. 4095 points carry window multiples too (windows of ~log2(n) + 2 bits, ONE bucket set: the host's
    Horner chain over ~30 windows was a third of a small proof).  Uniform, 0/1-heavy, all-equal and extreme scalars, a sub-range
    that still takes the merged path (>= n / 8 terms) and one that does not; discrete-log identity."""
    rs = np.random.RandomState(group * 1000 + n)
    km = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); km[:, 3] &= np.uint64((1 << 60) - 1)
    dk = ctx.upload(km)
    one = cv.fr_to_mont([1])[0]
    bases = ctx.fixed_base(dk.ptr, n, group, one)
    bases.precompute()
    c = ctx.lib.zk_bases_window_bits(bases.h)
    assert 9 <= c <= 16, c
    ks = cv.fr_from_mont(km)
    r = O.R_MOD
    gen_mul = (lambda e: O.g1_mul(O.G1_GEN, e)) if group == 1 else (lambda e: O.g2_mul(O.G2_GEN, e))
    to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
    sets = {"uniform": [int.from_bytes(rs.bytes(40), "little") % r for _ in range(n)],
            "bits": [int(rs.randint(0, 2)) if i % 9 else int.from_bytes(rs.bytes(40), "little") % r for i in range(n)],
            "equal": [12345678901234567890123] * n,
            "extreme": [(r - 1 - i) if i % 2 else (1 << (i % 253)) for i in range(n)],
            "zero": [0] * n}
    for name, sc in sets.items():
        ds = ctx.upload(cv.fr_to_mont(sc))
        for off, m in ((0, n), (3, n - 3), (n // 2, n // 4), (5, 10)):
            e = sum(s * k for s, k in zip(sc[:m], ks[off:off + m])) % r
            assert to_aff(ctx.msm_dev(bases, off, ds.ptr, m)) == gen_mul(e), (name, off, m, c)
        ds.free()
    bases.free(); dk.free()


def test_short_msm_over_a_large_table_of_window_multiples(ctx):
    """A 2^20-point table gets c = 20 (2^19 buckets, 13 windows); an MSM of 4096 .. 5041 terms over it has fewer than 2^16 digits
    and used to fall to the counting sort, whose one-block scan cannot take more than 2^16 buckets (ZK_ERR_ARG: ADVICE round 4).
    Sub-ranges of a resident key, Marlin commitments of a |H| = 4096 circuit over a large SRS and the crumbs of
    zk_groth16_prove_multi all land there.  Discrete-log identity, with and without an offset."""
    n = 1 << 20
    rs = np.random.RandomState(9090)
    km = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); km[:, 3] &= np.uint64((1 << 60) - 1)
    sm = rs.randint(0, 1 << 62, size=(8192, 4), dtype=np.uint64); sm[:, 3] &= np.uint64((1 << 60) - 1)
    dk, ds = ctx.upload(km), ctx.upload(sm)
    one = cv.fr_to_mont([1])[0]
    bases = ctx.fixed_base(dk.ptr, n, 1, one)
    bases.precompute()
    assert ctx.lib.zk_bases_window_bits(bases.h) == 20
    ks, sc = cv.fr_from_mont(km[:1 << 16]), cv.fr_from_mont(sm)
    for m, off in [(4096, 0), (4097, 0), (4097, 12345), (5041, 7), (5042, 1), (6000, 31)]:
        e = sum(s * k for s, k in zip(sc[:m], ks[off:off + m])) % O.R_MOD
        assert cv.g1_projective_to_affine(ctx.msm_dev(bases, off, ds.ptr, m)) == O.g1_mul(O.G1_GEN, e), (m, off)
    bases.free(); dk.free(); ds.free()


@pytest.mark.parametrize("group,n", [(1, 5000), (1, 1 << 16), (2, 6000), (1, (1 << 16) + 77), (2, (1 << 14) + 5)])
def test_msm_precomputed_window_multiples(ctx, group, n):
    """Resident bases with precomputed 2^(c w) multiples (one bucket set for all windows): random scalars,
    witness-like 0/1-heavy scalars, all-equal scalars (every point in one bucket per window), extreme scalars and
    an offset sub-range must give the same group element as the discrete-log identity.  The sizes sit on both sides of the
    2^16-digit line between the counting sort of small MSMs and the bucket sort of msm_sort.hip."""
    rs = np.random.RandomState(group * 100 + (n & 0xff))
    km = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    km[:, 3] &= np.uint64((1 << 60) - 1)
    dk = ctx.upload(km)
    one = cv.fr_to_mont([1])[0]
    bases = ctx.fixed_base(dk.ptr, n, group, one)
    bases.precompute()
    ks = cv.fr_from_mont(km)
    r = O.R_MOD

    def check(scalars, off=0):
        m = len(scalars)
        ds = ctx.upload(cv.fr_to_mont(scalars))
        out = ctx.msm_dev(bases, off, ds.ptr, m)
        e = sum(s * k for s, k in zip(scalars, ks[off:off + m])) % r
        if group == 1:
            assert cv.g1_projective_to_affine(out) == O.g1_mul(O.G1_GEN, e)
        else:
            assert cv.g2_projective_to_affine(out) == O.g2_mul(O.G2_GEN, e)
        ds.free()

    prng = O.Prng(n)
    check([prng.fr() for _ in range(n)])
    check([(prng.u64() & 1) if i % 5 else prng.fr() for i in range(n)])          # boolean-heavy witness
    check([12345678901234567890123] * n)                                          # all equal
    check([0] * n)                                                                # all zero: every digit is the 'none' key
    check([0] * (n - 3) + [5, 0, r - 1])                                          # nearly all zero
    check([r - 1, r - 2, 1, 0, (r - 1) // 2, 1 << 252, (1 << 16) - 1, 1 << 16] * (n // 8))
    check([prng.fr() for _ in range(n - 7)], off=7)                               # offset + shorter
    bases.free()


@pytest.mark.parametrize("n", [(1 << 16) + 100, (1 << 18) - 2])
def test_groth16_precomputed_key_matches_plain(ctx, n):
    """The same proof bytes with the proving key's window multiples (default for queries of >= 2^16 points; 2^18: c = 17,
    sliced reduce) and without them (ZK_PRECOMP=0), both equal to the known-trapdoor prediction of the C oracle."""
    import os
    import zkref_c as OC
    rng = O.Prng(n)
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(rng.fr()) for _ in range(7)]
    w0, w1, rr, ss = mont(rng.fr()), mont(rng.fr()), mont(rng.fr()), mont(rng.fr())
    dr = ctx.r1cs_mul_chain(n)
    dz = ctx.mul_chain_assignment_dev(n, w0, w1)
    os.environ["ZK_PRECOMP"] = "0"
    try:
        pk0 = ctx.groth16_setup(dr, *td)
    finally:
        del os.environ["ZK_PRECOMP"]
    pk1 = ctx.groth16_setup(dr, *td)
    assert ctx.lib.zk_bases_window_bits(pk0.query_bases("a_query").h) == 0
    assert ctx.lib.zk_bases_window_bits(pk1.query_bases("a_query").h) >= 15
    p0 = ctx.create_proof_dev(pk0, dr, dz.ptr, rr, ss)
    p1 = ctx.create_proof_dev(pk1, dr, dz.ptr, rr, ss)
    assert p0 == p1
    zarr = ctx.download(dz, (n + 3, 4))
    cr = OC.R1cs(2, n + 1, *OC.mul_chain_csr(n))
    assert p1 == OC.groth16_predict(cr, np.stack(td), zarr, OC.witness_map(cr, zarr), rr, ss)
    pk0.free(); pk1.free()


def test_msm_batch_pipeline(ctx):
    """zk_msm_batch_dev: more jobs than scratch slots, both groups, an empty job, offsets; equal to the single calls."""
    rng = O.Prng(4242)
    n = 300
    ks = [rng.fr() for _ in range(n)]
    sc = ctx.upload(cv.fr_to_mont(ks))
    b1 = ctx.fixed_base(sc.ptr, n, 1, mont1(1))
    b2 = ctx.fixed_base(sc.ptr, 64, 2, mont1(1))
    vecs = [ctx.upload(cv.fr_to_mont([rng.fr() for _ in range(n)])) for _ in range(4)]
    jobs = [(b1, 0, vecs[0].ptr, n), (b2, 0, vecs[1].ptr, 64), (b1, 5, vecs[2].ptr, 100), (b1, 0, vecs[3].ptr, 0),
            (b1, 0, vecs[3].ptr, n), (b2, 3, vecs[0].ptr, 61), (b1, 0, vecs[1].ptr, 1), (b1, 290, vecs[2].ptr, 10)]
    outs = ctx.msm_batch_dev(jobs)
    for (b, off, s, m), got in zip(jobs, outs):
        want = ctx.msm_dev(b, off, s, m)
        to_aff = cv.g1_projective_to_affine if b.group == 1 else cv.g2_projective_to_affine
        assert to_aff(got) == to_aff(want)


def test_msm_batch_over_window_multiples_with_offsets(ctx):
    """Pipelined batch over ONE resident table that carries window multiples (2^18 points: c = 17, sliced reduce), jobs
    with base offsets (the shifted powers of a degree-bounded KZG commitment) and different lengths: same group elements
    as the plain table gives."""
    n = 1 << 18
    rs = np.random.RandomState(99)

    def rand_mont(m):
        a = rs.randint(0, 1 << 62, size=(m, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    dk = ctx.upload(rand_mont(n))
    plain = ctx.fixed_base(dk.ptr, n, 1, mont1(1))
    pre = ctx.fixed_base(dk.ptr, n, 1, mont1(1))
    pre.precompute()
    assert ctx.lib.zk_bases_window_bits(pre.h) == 17 and ctx.lib.zk_bases_window_bits(plain.h) == 0
    vecs = [ctx.upload(rand_mont(n)) for _ in range(3)]
    spec = [(0, n, 0), (1000, n - 1000, 1), (5, 70000, 2), (n - 4096, 4096, 0), (12345, 1, 1)]
    got = ctx.msm_batch_dev([(pre, off, vecs[v].ptr, m) for off, m, v in spec])
    want = ctx.msm_batch_dev([(plain, off, vecs[v].ptr, m) for off, m, v in spec])
    for g, w in zip(got, want):
        assert cv.g1_projective_to_affine(g) == cv.g1_projective_to_affine(w)


def _mont_inner_product(ctx, dk, ds, m):
    """sum_i k_i s_i mod r (canonical integer) of two device vectors of Montgomery residues: the products on the device, the sum
    exact on the host in 32-bit halves (2^24 terms of < 2^32 stay below 2^56)."""
    prod = ctx.alloc(m * 32)
    ctx.fr_vec_op_dev(0, dk, ds, prod.ptr, m)
    pr = ctx.download(prod, (m, 4))
    prod.free()
    tot = 0
    for j in range(4):
        lo = int((pr[:, j] & np.uint64(0xFFFFFFFF)).sum(dtype=np.uint64))
        hi = int((pr[:, j] >> np.uint64(32)).sum(dtype=np.uint64))
        tot += (lo + (hi << 32)) << (64 * j)
    return cv.fr_from_mont(cv.fr_raw([tot % O.R_MOD]))[0]


@pytest.mark.parametrize("group,log_n", [(1, 22), (1, 24), (2, 22)])
def test_msm_benched_sizes_discrete_log_check(ctx, group, log_n):
    """The sizes bench.py's `micro` rows time (G1 2^22 / 2^24, G2 2^22), checked: over the plain table and over the same table
    with window multiples, bases = k_i G, sum s_i (k_i G) == (sum s_i k_i) G.  Also a base offset and a ragged length."""
    n = 1 << log_n
    rs = np.random.RandomState(70 + log_n + group)

    def rand_mont(m):
        a = rs.randint(0, 1 << 62, size=(m, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    km, sm = rand_mont(n), rand_mont(n)
    dk, ds = ctx.upload(km), ctx.upload(sm)
    del km, sm
    one = cv.fr_to_mont([1])[0]
    to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
    mul = (lambda e: O.g1_mul(O.G1_GEN, e)) if group == 1 else (lambda e: O.g2_mul(O.G2_GEN, e))
    bases = ctx.fixed_base(dk.ptr, n, group, one)
    want = mul(_mont_inner_product(ctx, dk.ptr, ds.ptr, n))
    assert to_aff(ctx.msm_dev(bases, 0, ds.ptr, n)) == want, "plain table"
    m = n - 12345                                        # ragged length behind an offset: bases[5 ...] x scalars[0 ...]
    want_off = mul(_mont_inner_product(ctx, dk.ptr + 5 * 32, ds.ptr, m))
    assert to_aff(ctx.msm_dev(bases, 5, ds.ptr, m)) == want_off, "plain table, offset"
    bases.precompute()
    assert ctx.lib.zk_bases_window_bits(bases.h) >= 13, "this device holds the window multiples of a 2^%d table" % log_n
    assert to_aff(ctx.msm_dev(bases, 0, ds.ptr, n)) == want, "window multiples"
    assert to_aff(ctx.msm_dev(bases, 5, ds.ptr, m)) == want_off, "window multiples, offset"
    bases.free(); dk.free(); ds.free()


def _adversarial_sets(rs, n):
    """The scalar sets of bench.py's adversarial rows (SURVEY 8d 'MSM micro'), plus small range-checked values."""
    a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    one = cv.fr_to_mont([1])[0]
    pick = rs.rand(n)
    z01 = a.copy()
    z01[pick < 0.45] = 0
    z01[(pick >= 0.45) & (pick < 0.9)] = one
    small = cv.fr_to_mont([int(v) for v in rs.randint(0, 1 << 16, size=4096)])
    sv = small[rs.randint(0, 4096, size=n)]
    sv[pick >= 0.9] = a[pick >= 0.9]
    minus_one = cv.fr_to_mont([O.R_MOD - 1])[0]
    pm = np.tile(one, (n, 1))
    pm[pick < 0.5] = minus_one                               # +1 / -1: both signs of one bucket
    return {"uniform": a, "all_zero": np.zeros((n, 4), dtype=np.uint64), "all_equal": np.tile(a[12345 % n:12345 % n + 1], (n, 1)),
            "all_one": np.tile(one, (n, 1)), "plus_minus_one": pm, "zero_one_heavy": z01, "small_values": np.ascontiguousarray(sv)}


@pytest.mark.parametrize("group,log_n", [(1, 20), (2, 18), (1, 12)])
def test_msm_adversarial_scalar_sets(ctx, group, log_n):
    """Witness-shaped scalar vectors (the reference's circuits are boolean-heavy: docs/benchmark.md:45-58; arkworks keeps a
    unit-scalar fast path for them, ec/src/msm/variable_base.rs:45-49): all zero, all one, +-1, all equal, 90 % zeros and ones,
    small range-checked values -- every one through the plain table and through window multiples, against the discrete-log
    identity.  These are the rows bench.py times."""
    n = 1 << log_n
    rs = np.random.RandomState(900 + log_n)
    km = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    km[:, 3] &= np.uint64((1 << 60) - 1)
    dk = ctx.upload(km)
    one = cv.fr_to_mont([1])[0]
    to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
    mul = (lambda e: O.g1_mul(O.G1_GEN, e)) if group == 1 else (lambda e: O.g2_mul(O.G2_GEN, e))
    plain = ctx.fixed_base(dk.ptr, n, group, one)
    pre = ctx.fixed_base(dk.ptr, n, group, one)
    pre.precompute()
    for name, arr in _adversarial_sets(rs, n).items():
        assert arr.shape == (n, 4), name
        ds = ctx.upload(np.ascontiguousarray(arr))
        want = mul(_mont_inner_product(ctx, dk.ptr, ds.ptr, n))
        assert to_aff(ctx.msm_dev(plain, 0, ds.ptr, n)) == want, (name, "plain table")
        assert to_aff(ctx.msm_dev(pre, 0, ds.ptr, n)) == want, (name, "window multiples")
        ds.free()
    plain.free(); pre.free(); dk.free()


@pytest.mark.parametrize("group,layout,note", [(1, 1, "packed"), (1, 2, "128-byte line"), (1, 3, "limbs"), (2, 1, "packed"), (1, 0, "limbs"),
                                               (2, 0, "packed")])
def test_window_multiple_layouts(ctx, group, layout, note):
    """The three memory layouts of a table's window multiples (zk_bases_precompute_as): packed, one 96-byte point per 128-byte
    line, and -- what the memory budget picks for G1 on this device -- 29-bit limbs with both signs, read by the accumulate
    kernel without unpacking or negation.  The same MSM through each, both signs of every digit, an offset sub-range."""
    n = (1 << 14) + 3
    rs = np.random.RandomState(31 + layout + 10 * group)
    km = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); km[:, 3] &= np.uint64((1 << 60) - 1)
    sm = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); sm[:, 3] &= np.uint64((1 << 60) - 1)
    dk, ds = ctx.upload(km), ctx.upload(sm)
    bases = ctx.fixed_base(dk.ptr, n, group, cv.fr_to_mont([1])[0])
    assert bases.precompute_note() == ""
    bases.precompute(layout)
    assert note in bases.precompute_note() and ctx.lib.zk_bases_window_bits(bases.h) >= 12
    to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
    mul = (lambda e: O.g1_mul(O.G1_GEN, e)) if group == 1 else (lambda e: O.g2_mul(O.G2_GEN, e))
    assert to_aff(ctx.msm_dev(bases, 0, ds.ptr, n)) == mul(_mont_inner_product(ctx, dk.ptr, ds.ptr, n))
    assert to_aff(ctx.msm_dev(bases, 9, ds.ptr, n - 9)) == mul(_mont_inner_product(ctx, dk.ptr + 9 * 32, ds.ptr, n - 9))
    if group == 2:
        with pytest.raises(Exception):
            other = ctx.fixed_base(dk.ptr, n, 2, cv.fr_to_mont([1])[0])
            try:
                other.precompute(3)                       # a G1 form
            finally:
                other.free()
    bases.free(); dk.free(); ds.free()


@pytest.mark.parametrize("group", [1, 2])
def test_lane_group_additions_special_cases(ctx, group):
    """Round 5: the additions of small (latency-bound) jobs run on two lane groups (ec_dual.cuh: G1 pairs in the group accumulate /
    fold / grid reduce, G2 quads in the accumulate / fold / grid reduce).  Tables with REPEATED points and their negatives force the
    branches random inputs never take: the same point twice in a bucket (mixed addition -> doubling), P and -P in a bucket
    (-> infinity), equal and opposite bucket sums in the folds and trees; single jobs and a batch of three (the group path)."""
    rng = O.Prng(900 + group)
    n = 512
    if group == 1:
        pts, neg, mul, add, to_arr, to_aff = g1_points(rng, 4), O.g1_neg, O.g1_mul, O.g1_add, cv.g1_affine_to_array, cv.g1_projective_to_affine
    else:
        pts, neg, mul, add, to_arr, to_aff = g2_points(rng, 4), O.g2_neg, O.g2_mul, O.g2_add, cv.g2_affine_to_array, cv.g2_projective_to_affine
    # base i: point (i % 4), negated when (i // 4) is odd; every 37th one is the point at infinity
    kind = [(i % 4, (i // 4) & 1, i % 37 == 36) for i in range(n)]
    table = [None if inf else (neg(pts[j]) if sgn else pts[j]) for j, sgn, inf in kind]
    bases = ctx.bases_upload(to_arr(table), group)
    bases.precompute()
    assert ctx.lib.zk_bases_window_bits(bases.h) >= 9
    r = O.R_MOD

    def expect(sc, off, m):
        tot = [0, 0, 0, 0]
        for s, (j, sgn, inf) in zip(sc[:m], kind[off:off + m]):
            if not inf:
                tot[j] = (tot[j] + (r - s if sgn else s)) % r
        acc = None
        for j in range(4):
            if tot[j]:
                acc = add(acc, mul(pts[j], tot[j]))
        return acc
    sets = {"equal": [0x1234567] * n,                                        # the same digit for every base: doublings and cancellations
            "pairs": [int(rng.fr()) if i % 8 < 4 else 0 for i in range(n)],   # P's scalar ...
            "uniform": [int(rng.fr()) for _ in range(n)],
            "small": [i % 3 for i in range(n)],
            "same": [5 if i % 8 == 0 else 0 for i in range(n)],              # one bucket: P0, P0, P0, ... (P + P in the accumulate loop)
            "cancel": [5 if i % 4 == 0 else 0 for i in range(n)]}            # one bucket: P0, -P0, P0, ... (-> infinity and back)
    sets["pairs"] = [sets["pairs"][i - 4] if i % 8 >= 4 else sets["pairs"][i] for i in range(n)]     # ... and the same one for -P
    dev = {k: ctx.upload(cv.fr_to_mont(v)) for k, v in sets.items()}
    for name, sc in sets.items():
        for off, m in ((0, n), (4, n - 4), (0, 300)):
            assert to_aff(ctx.msm_dev(bases, off, dev[name].ptr, m)) == expect(sc, off, m), (name, off, m)
    if group == 1:
        names = ("equal", "pairs", "same", "cancel")
        jobs = [(bases, 0, dev["equal"].ptr, n), (bases, 0, dev["pairs"].ptr, n), (bases, 0, dev["same"].ptr, n), (bases, 8, dev["cancel"].ptr, n - 8)]
        got = ctx.msm_batch_dev(jobs)
        for (b, off, _, m), g, name in zip(jobs, got, names):
            assert to_aff(g) == expect(sets[name], off, m), ("batch", name)
    for d in dev.values():
        d.free()
    bases.free()


def test_sort_graph_replay_reads_fresh_data(ctx):
    """core.hip::zk_graph_run: from the third sort with the same arguments on, the ~13 launches of a bucket sort are one captured graph.
    The same device buffers with NEW scalars every round (what a prover's queue does), an MSM of another size through the same scratch
    slot in between (its larger buffers move the scratch addresses: the generation count must retire the old graphs), and back."""
    import ctypes as C
    rs = np.random.RandomState(4711)
    n, big = 6000, 20000

    def rand_mont(m):
        a = rs.randint(0, 1 << 62, size=(m, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    km = rand_mont(big)
    dk = ctx.upload(km)
    bases = ctx.fixed_base(dk.ptr, big, 1, mont1(1))                 # plain table: n * W digits >= 2^16, the multi-launch sort
    ks = cv.fr_from_mont(km)
    r = O.R_MOD
    ds = ctx.alloc(big * 32)

    def run(m):
        sm = rand_mont(m)
        ctx._ck(ctx.lib.zk_memcpy_h2d(ctx.h, C.c_void_p(ds.ptr), sm.ctypes.data_as(C.c_void_p), sm.nbytes))     # same buffer, new contents
        sc = cv.fr_from_mont(sm)
        e = sum(s * k for s, k in zip(sc, ks[:m])) % r
        assert cv.g1_projective_to_affine(ctx.msm_dev(bases, 0, ds.ptr, m)) == O.g1_mul(O.G1_GEN, e), m
    for _ in range(5):
        run(n)                     # plain, plain, captured, replayed, replayed
    run(big)                       # larger scratch: addresses move
    for _ in range(4):
        run(n)
    ds.free(); bases.free(); dk.free()
