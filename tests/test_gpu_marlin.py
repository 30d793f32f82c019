"""GPU parity: the Marlin AHP prover rounds against oracle/marlin_ref.py, bit-exact polynomials on
small systems; at larger sizes the verifier's two sum-check equations (marlin/src/ahp/mod.rs:134-290) must evaluate to
zero on the device's polynomials, and KZG10 commitments / openings of the round polynomials must verify."""
import numpy as np
import pytest

import marlin_ref as M
import zkref as O
import zk_mpc_amd.convert as cv
from zk_mpc_amd import marlin as DM
from helpers import mont1

pytestmark = pytest.mark.gpu

LABELS_INDEX = [m + s for m in "abc" for s in ("_row", "_col", "_val", "_row_col")]


def build(ctx, n, seed):
    rng = O.Prng(seed)
    r1cs, z = O.mul_chain_r1cs(n, rng.fr(), rng.fr())
    sq, zz = M.pad_and_square(r1cs, z)
    dix = DM.Index(ctx, sq.num_instance, sq.num_witness, DM.Csr.from_rows(sq.a), DM.Csr.from_rows(sq.b), DM.Csr.from_rows(sq.c))
    return rng, r1cs, sq, zz, dix


def run_device(ctx, dix, zz, rnd, ch):
    st = DM.prover_init(dix, ctx.upload(cv.fr_to_mont(zz)))
    polys = dict(dix.polynomials())
    polys.update(DM.prover_first_round(st, cv.fr_to_mont(rnd)))
    polys.update(DM.prover_second_round(st, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"]))
    polys.update(DM.prover_third_round(st, ch["beta"]))
    return polys, st


@pytest.mark.parametrize("n", [3, 6, 13, 40])
def test_rounds_match_oracle(ctx, n):
    rng, r1cs, sq, zz, dix = build(ctx, n, 800 + n)
    oix = M.Index(sq)
    for label, want in oix.polynomials().items():
        assert DM.download_poly(ctx, dix.polynomials()[label]) == want, label
    md = M.mask_poly_degree(oix)
    rnd = [rng.fr() for _ in range(3 + md + 1)]
    ch = {k: rng.fr() for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma")}
    ost = M.prover_init(oix, zz)
    want = dict(M.prover_first_round(ost, rnd[0], rnd[1], rnd[2], rnd[3:]))
    want.update(M.prover_second_round(ost, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"]))
    want.update(M.prover_third_round(ost, ch["beta"]))
    got, _ = run_device(ctx, dix, zz, rnd, ch)
    for label, w in want.items():
        assert M.strip(DM.download_poly(ctx, got[label])) == M.strip(w), label


def test_unsatisfied_system_is_rejected(ctx):
    rng, r1cs, sq, zz, dix = build(ctx, 6, 811)
    zz[5] = (zz[5] + 1) % O.R_MOD
    md = 3 * dix.dom_h.size - 1
    rnd = [rng.fr() for _ in range(3 + md + 1)]
    ch = {k: rng.fr() for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma")}
    with pytest.raises(ValueError, match="sum over H"):
        run_device(ctx, dix, zz, rnd, ch)


@pytest.mark.parametrize("n", [1000, 5000])
def test_sumcheck_equations_hold_at_size(ctx, n):
    """Beyond what the Python oracle proves in seconds: the device's polynomials satisfy the verifier's equations."""
    rng, r1cs, sq, zz, dix = build(ctx, n, 820 + n)
    md = 3 * dix.dom_h.size - 1
    rnd = [rng.fr() for _ in range(3 + md + 1)]
    ch = {k: rng.fr() for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma")}
    polys, _ = run_device(ctx, dix, zz, rnd, ch)
    ev = lambda label, pt: cv.fr_from_mont(ctx.poly_evaluate_dev(polys[label].ptr, polys[label].n, mont1(pt)).reshape(1, 4))[0]
    info = M.IndexInfo(dix.num_constraints, dix.num_non_zero, dix.num_instance)
    pub = zz[1:r1cs.num_instance]
    outer, inner = M.sumcheck_equations(info, pub, ev, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"], ch["beta"], ch["gamma"])
    assert outer == 0 and inner == 0
    bad = [(pub[0] + 1) % O.R_MOD] + pub[1:]
    outer, _ = M.sumcheck_equations(info, bad, ev, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"], ch["beta"], ch["gamma"])
    assert outer != 0
    # degree bounds of the reference's assertions (prover.rs:545-546,707): the stripped lengths
    H, K = dix.dom_h.size, dix.dom_k.size
    assert len(M.strip(DM.download_poly(ctx, polys["g_1"]))) - 1 <= H - 2
    assert len(M.strip(DM.download_poly(ctx, polys["h_1"]))) - 1 <= 2 * H
    assert len(M.strip(DM.download_poly(ctx, polys["g_2"]))) - 1 <= K - 2


def test_round_polynomials_commit_and_open_with_kzg10(ctx):
    """Marlin::prove's PC::commit and opening of the nine prover polynomials (lib.rs:171-247,296-306) through KZG10 on the
    device: commitments equal the oracle's MSMs and each opening at beta / gamma passes the pairing check."""
    n = 6
    rng, r1cs, sq, zz, dix = build(ctx, n, 830)
    md = 3 * dix.dom_h.size - 1
    rnd = [rng.fr() for _ in range(3 + md + 1)]
    ch = {k: rng.fr() for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma")}
    polys, _ = run_device(ctx, dix, zz, rnd, ch)
    max_deg = max(p.n for p in polys.values())
    pp = O.KzgParams(max_deg, rng.fr(), g_k=rng.fr(), gg_k=rng.fr(), h_k=rng.fr())
    pg = ctx.bases_upload(cv.g1_affine_to_array(pp.powers_of_g), 1)
    for label, point in (("w", "beta"), ("z_a", "beta"), ("z_b", "beta"), ("mask_poly", "beta"), ("t", "beta"), ("g_1", "beta"),
                         ("h_1", "beta"), ("g_2", "gamma"), ("h_2", "gamma")):
        p = polys[label]
        coeffs = DM.download_poly(ctx, p)
        comm = cv.g1_projective_to_affine(ctx.kzg_commit_dev(pg, p.ptr, p.n))
        assert comm == O.kzg_commit(pp, coeffs), label
        z = ch[point]
        w, _ = ctx.kzg_open_dev(pg, p.ptr, p.n, mont1(z))
        assert O.kzg_check(pp, comm, z, O.poly_evaluate(coeffs, z), cv.g1_projective_to_affine(w)), label
    # the round's commitments as one pipelined batch, and a batched opening of several polynomials at one point
    labels = ["w", "z_a", "z_b", "mask_poly"]
    comms = DM.commit(ctx, pg, {l: polys[l] for l in labels})
    for l in labels:
        assert cv.g1_projective_to_affine(comms[l]) == O.kzg_commit(pp, DM.download_poly(ctx, polys[l])), l
    xi = rng.fr()
    (w_beta,) = DM.batch_open(ctx, pg, [([polys[l] for l in labels], ch["beta"])], xi)
    comb = []
    for i, l in enumerate(labels):
        comb = M.padd(comb, M.pscale(DM.download_poly(ctx, polys[l]), pow(xi, i, O.R_MOD)))
    c_comb = O.kzg_commit(pp, comb)
    assert O.kzg_check(pp, c_comb, ch["beta"], O.poly_evaluate(comb, ch["beta"]), cv.g1_projective_to_affine(w_beta))


def test_marlin_pc_commit_with_bounds_and_hiding(ctx):
    """MarlinKZG10::commit on the round oracles: hiding commitments (w, z_a, z_b, g_1) and the shifted commitments of the
    degree-bounded oracles (g_1: |H| - 2, g_2: |K| - 2) equal the oracle's MSMs over the (shifted) powers."""
    n = 6
    rng, r1cs, sq, zz, dix = build(ctx, n, 840)
    md = 3 * dix.dom_h.size - 1
    rnd = [rng.fr() for _ in range(3 + md + 1)]
    ch = {k: rng.fr() for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma")}
    polys, _ = run_device(ctx, dix, zz, rnd, ch)
    prover = {l: polys[l] for l in ("w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2")}
    max_deg = max(p.n for p in prover.values()) + 3
    pp = O.KzgParams(max_deg, rng.fr(), g_k=rng.fr(), gg_k=rng.fr(), h_k=rng.fr())
    pg = ctx.bases_upload(cv.g1_affine_to_array(pp.powers_of_g), 1)
    pgg = ctx.bases_upload(cv.g1_affine_to_array(pp.powers_of_gamma_g), 1)
    bounds = DM.oracle_bounds(dix)
    blinds_h, blinds_d = {}, {}
    for label, (bound, hiding) in bounds.items():
        if hiding is not None:
            b0 = [rng.fr() for _ in range(hiding + 1)]
            b1 = [rng.fr() for _ in range(hiding + 1)] if bound is not None else None
            blinds_h[label] = (b0, b1)
            up = lambda v: DM.DevPoly(ctx.upload(cv.fr_to_mont(v)), len(v))
            blinds_d[label] = (up(b0), up(b1) if b1 else None)
    got = DM.commit_marlin_pc(ctx, pg, pgg, prover, bounds, blinds_d)
    for label, p in prover.items():
        coeffs = DM.download_poly(ctx, p)
        bound, hiding = bounds[label]
        want = O.msm_naive(pp.powers_of_g, coeffs, O.FqOps)
        if hiding is not None:
            want = O.g1_add(want, O.msm_naive(pp.powers_of_gamma_g, blinds_h[label][0], O.FqOps))
        assert cv.g1_projective_to_affine(got[label]["comm"]) == want, label
        if bound is None:
            assert got[label]["shifted_comm"] is None
        else:
            ws = O.msm_naive(pp.powers_of_g[max_deg - bound:], coeffs, O.FqOps)
            if hiding is not None:
                ws = O.g1_add(ws, O.msm_naive(pp.powers_of_gamma_g, blinds_h[label][1], O.FqOps))
            assert cv.g1_projective_to_affine(got[label]["shifted_comm"]) == ws, label
    # a polynomial above its bound is refused
    with pytest.raises(ValueError, match="exceeds its bound"):
        DM.commit_marlin_pc(ctx, pg, pgg, {"g_2": prover["h_2"]}, {"g_2": (1, None)})
