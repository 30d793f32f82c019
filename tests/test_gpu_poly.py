"""GPU parity: dense-polynomial kernels and KZG10 commit / open against the oracle."""
import numpy as np
import pytest

import marlin_ref as M
import zkref as O
import zk_mpc_amd.convert as cv
from helpers import mont1

pytestmark = pytest.mark.gpu


def up(ctx, vals):
    return ctx.upload(cv.fr_to_mont(vals) if len(vals) else np.zeros((1, 4), dtype=np.uint64))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4096, 4097, 70000])
def test_evaluate_and_divide_by_linear(ctx, n):
    rng = O.Prng(3000 + n)
    c = [rng.fr() for _ in range(n)]
    d = up(ctx, c)
    for z in (rng.fr(), 0, 1, O.R_MOD - 1):
        assert cv.fr_from_mont(ctx.poly_evaluate_dev(d.ptr, n, mont1(z)).reshape(1, 4)) == [O.poly_evaluate(c, z)]
        q = ctx.alloc(max(n - 1, 1) * 32)
        rem = ctx.poly_divide_by_linear_dev(d.ptr, n, mont1(z), q.ptr)
        wq, wr = O.poly_divide_with_q_and_r(c, [(-z) % O.R_MOD, 1])
        assert cv.fr_from_mont(rem.reshape(1, 4)) == [O.poly_evaluate(c, z)]
        if n > 1:
            assert cv.fr_from_mont(ctx.download(q, (n - 1, 4))) == wq
            assert wr == [O.poly_evaluate(c, z)]


def test_evaluate_batch(ctx):
    """zk_poly_evaluate_batch_dev: pairs of different lengths (incl. empty, one coefficient, spans that end on a block edge,
    more than 256 spans so that the join folds several per thread) and different points, one call."""
    rng = O.Prng(3100)
    sizes = [0, 1, 15, 16, 17, 4095, 4096, 4097, 8192, 70001, (1 << 20) + 4097 + 5]
    polys, cs, pts = [], [], []
    keep = []
    for n in sizes:
        c = [rng.fr() for _ in range(min(n, 70001))]
        if n > len(c):                      # long one: a short random head, zeros, a random tail (Horner in python stays cheap)
            c = c[:300] + [0] * (n - 600) + c[300:600]
        d = up(ctx, c)
        keep.append(d)
        polys.append((d.ptr, n))
        cs.append(c)
        pts.append(rng.fr() if n != 16 else 0)
    out = ctx.poly_evaluate_batch_dev(polys, cv.fr_to_mont(pts))
    got = cv.fr_from_mont(out)

    def horner(c, z):
        acc, i = 0, len(c)
        nz = [(j, v) for j, v in enumerate(c) if v]
        return sum(v * pow(z, j, O.R_MOD) for j, v in nz) % O.R_MOD
    assert got == [horner(c, z) for c, z in zip(cs, pts)]
    assert ctx.poly_evaluate_batch_dev([], np.zeros((0, 4), dtype=np.uint64)).shape == (0, 4)


def test_divide_by_linear_beyond_256_spans(ctx):
    """More than 256 spans of 4096 coefficients: the one-block suffix scan of the spans folds two per thread."""
    rng = O.Prng(3200)
    n = (1 << 20) + 4097 + 3
    c = [rng.fr() for _ in range(2000)] + [0] * (n - 4000) + [rng.fr() for _ in range(2000)]
    z = rng.fr()
    d, q = up(ctx, c), ctx.alloc(n * 32)
    rem = ctx.poly_divide_by_linear_dev(d.ptr, n, mont1(z), q.ptr)
    want, acc = [0] * (n - 1), 0
    for i in range(n - 1, 0, -1):
        acc = (c[i] + acc * z) % O.R_MOD
        want[i - 1] = acc
    assert cv.fr_from_mont(ctx.download(q, (n - 1, 4))) == want
    assert cv.fr_from_mont(rem.reshape(1, 4)) == [(c[0] + acc * z) % O.R_MOD]


def test_divide_by_root_of_domain(ctx):
    """z inside the evaluation domain (where an evaluate-and-interpolate division would divide by zero)."""
    rng = O.Prng(31)
    n = 300
    c = [rng.fr() for _ in range(n)]
    z = O.Domain(512).element(5)
    d, q = up(ctx, c), ctx.alloc(n * 32)
    ctx.poly_divide_by_linear_dev(d.ptr, n, mont1(z), q.ptr)
    assert cv.fr_from_mont(ctx.download(q, (n - 1, 4))) == O.poly_divide_with_q_and_r(c, [(-z) % O.R_MOD, 1])[0]


@pytest.mark.parametrize("n,log_dom", [(5, 3), (8, 3), (9, 3), (100, 5), (1000, 8), (3000, 10), (5000, 0), (5001, 1), (100000, 2), (70000, 15)])
def test_divide_by_vanishing(ctx, n, log_dom):
    rng = O.Prng(3100 + n)
    c = [rng.fr() for _ in range(n)]
    N = 1 << log_dom
    d, q, r = up(ctx, c), ctx.alloc(max(n, 1) * 32), ctx.alloc(N * 32)
    ctx.poly_divide_by_vanishing_dev(d.ptr, n, log_dom, q.ptr, r.ptr)
    wq, wr = M.divide_by_vanishing(c, N)       # O(n); cross-checked against the generic long division in test_oracle.py
    if n <= 1000 and n > N:
        assert (wq, wr) == O.poly_divide_with_q_and_r(c, [O.R_MOD - 1] + [0] * (N - 1) + [1])
    nq = max(n - N, 0)
    assert cv.fr_from_mont(ctx.download(r, (N, 4))) == (wr + [0] * N)[:N]
    if nq:
        assert cv.fr_from_mont(ctx.download(q, (nq, 4))) == wq


def test_batch_inversion_and_powers(ctx):
    rng = O.Prng(32)
    n = 5000
    v = [rng.fr() for _ in range(n)]
    v[0], v[17], v[n - 1] = 0, 0, 1
    d = up(ctx, v)
    ctx.batch_inversion_dev(d.ptr, n)
    assert cv.fr_from_mont(ctx.download(d, (n, 4))) == O.batch_inversion(v)
    b, s = rng.fr(), rng.fr()
    out = ctx.alloc(1000 * 32)
    ctx.fr_powers_dev(mont1(b), mont1(s), 1000, out.ptr)
    assert cv.fr_from_mont(ctx.download(out, (1000, 4))) == [s * pow(b, i, O.R_MOD) % O.R_MOD for i in range(1000)]


@pytest.mark.parametrize("na,nb", [(1, 1), (3, 5), (64, 64), (100, 29)])
def test_poly_mul(ctx, na, nb):
    rng = O.Prng(3200 + na)
    a, b = [rng.fr() for _ in range(na)], [rng.fr() for _ in range(nb)]
    da, db, out = up(ctx, a), up(ctx, b), ctx.alloc((na + nb) * 32)
    ctx.poly_mul_dev(da.ptr, na, db.ptr, nb, out.ptr)
    assert cv.fr_from_mont(ctx.download(out, (na + nb - 1, 4))) == O.poly_mul(a, b)


def test_kzg10_commit_open_check(ctx):
    """KZG10 with and without hiding: commitments / proofs equal the oracle's and satisfy the pairing check."""
    rng = O.Prng(33)
    deg = 40
    beta = rng.fr()
    pp = O.KzgParams(deg, beta, g_k=rng.fr(), gg_k=rng.fr(), h_k=rng.fr())
    pg = ctx.bases_upload(cv.g1_affine_to_array(pp.powers_of_g), 1)
    pgg = ctx.bases_upload(cv.g1_affine_to_array(pp.powers_of_gamma_g), 1)
    coeffs = [rng.fr() for _ in range(deg + 1)]
    blind = [rng.fr() for _ in range(3)]
    dc, dbl = up(ctx, coeffs), up(ctx, blind)
    z = rng.fr()
    v = O.poly_evaluate(coeffs, z)
    # no hiding
    c0 = cv.g1_projective_to_affine(ctx.kzg_commit_dev(pg, dc.ptr, deg + 1))
    assert c0 == O.kzg_commit(pp, coeffs)
    w0, _ = ctx.kzg_open_dev(pg, dc.ptr, deg + 1, mont1(z))
    w0 = cv.g1_projective_to_affine(w0)
    assert w0 == O.kzg_open(pp, coeffs, z)[0]
    assert O.kzg_check(pp, c0, z, v, w0)
    assert not O.kzg_check(pp, c0, z, (v + 1) % O.R_MOD, w0)
    # hiding bound 2 (blinding polynomial of degree 2)
    c1 = cv.g1_projective_to_affine(ctx.kzg_commit_dev(pg, dc.ptr, deg + 1, pgg, dbl.ptr, 3))
    assert c1 == O.kzg_commit(pp, coeffs, blind)
    w1, rv = ctx.kzg_open_dev(pg, dc.ptr, deg + 1, mont1(z), pgg, dbl.ptr, 3)
    w1, rv = cv.g1_projective_to_affine(w1), cv.fr_from_mont(rv.reshape(1, 4))[0]
    ow, orv = O.kzg_open(pp, coeffs, z, blind)
    assert (w1, rv) == (ow, orv)
    assert O.kzg_check(pp, c1, z, v, w1, rv)
