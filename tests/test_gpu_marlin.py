"""GPU parity: the Marlin AHP prover rounds against oracle/marlin_ref.py, bit-exact polynomials on
small systems; at larger sizes the verifier's two sum-check equations (marlin/src/ahp/mod.rs:134-290) must evaluate to
zero on the device's polynomials, and KZG10 commitments / openings of the round polynomials must verify."""
import numpy as np
import pytest

import marlin_ref as M
import zkref as O
import zk_mpc_amd.convert as cv
import pyseq.marlin_seq as DM
from helpers import marlin_test_system, mont1

pytestmark = pytest.mark.gpu

LABELS_INDEX = [m + s for m in "abc" for s in ("_row", "_col", "_val", "_row_col")]


def build(ctx, n, seed):
    """n: an int (the mul-chain of SURVEY 8d: one term per row, unit coefficients, |K| = |H|) or "dense<k>" / "tiny<r>" (circuit-shaped:
    multi-term rows, non-unit coefficients, several public inputs, A / B / C of different density, |K| up to 4 |H|: helpers.py)."""
    rng = O.Prng(seed)
    r1cs, z = marlin_test_system(n, rng)
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


@pytest.mark.parametrize("n", [3, 6, 13, 40, "tiny5", "tiny11", "dense5", "dense6"])
def test_rounds_match_oracle(ctx, n):
    """Index polynomials (row / col / val / row_col of A*, B*, C*) and the nine round polynomials, coefficient for coefficient."""
    rng, r1cs, sq, zz, dix = build(ctx, n, 800 + (n if isinstance(n, int) else len(n) * 7 + int(n[-1])))
    if not isinstance(n, int):
        nnz = [sum(len(r) for r in m) for m in (sq.a, sq.b, sq.c)]
        assert max(nnz) > min(nnz) and dix.dom_k.size >= 2 * dix.dom_h.size and sq.num_instance >= 4     # the shape the mul-chain never has
        assert any(c not in (1,) for row in sq.a for c, _ in row)
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


@pytest.mark.parametrize("n", [1000, 5000, "dense10"])
def test_sumcheck_equations_hold_at_size(ctx, n):
    """Beyond what the Python oracle proves in seconds: the device's polynomials satisfy the verifier's equations."""
    rng, r1cs, sq, zz, dix = build(ctx, n, 820 + (n if isinstance(n, int) else 10))
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


# ---- Marlin as a proof: transcript, hiding commitments, open_combinations, the oracle's verifier ----------------------------------

def _oracle_keys(oix, beta, g_k, gg_k, h_k, extra=3):
    import marlin_full_ref as MF
    pp = O.KzgParams(MF.max_degree_for(oix) + extra, beta, g_k=g_k, gg_k=gg_k, h_k=h_k)
    return MF.Keys(oix, pp)


@pytest.mark.parametrize("n", [3, 6, 13, "tiny7", "dense5"])
def test_marlin_proof_bytes_equal_the_oracle_prover(ctx, n):
    """Marlin::prove on the device (marlin.py::prove: Fiat-Shamir through the library's FiatShamirRng<Blake2s>, MarlinKZG10
    commitments with hiding and degree bounds, open_combinations) against the oracle's Python prover from the same ChaCha20 prover
    rng: the same CanonicalSerialize bytes, the same challenges; the oracle's verifier accepts them and rejects a wrong input."""
    import fsrng_ref as FR
    import marlin_full_ref as MF
    from zk_mpc_amd.api import Rng
    rng, r1cs, sq, zz, dix = build(ctx, n, 4000 + (n if isinstance(n, int) else 90 + int(n[-1])))
    oix = M.Index(sq)
    beta, g_k, gg_k, h_k = rng.fr(), rng.fr(), rng.fr(), rng.fr()
    okeys = _oracle_keys(oix, beta, g_k, gg_k, h_k)
    srs = DM.UniversalSrs(ctx, okeys.max_degree, beta, g_k, gg_k)
    dkeys = DM.IndexKeys(dix, srs)
    assert dkeys.ivk_bytes() == okeys.ivk_bytes()
    seed = bytes((7 * i + (n if isinstance(n, int) else 5)) & 0xff for i in range(32))
    got = DM.prove(dkeys, ctx.upload(cv.fr_to_mont(zz)), Rng.from_seed(seed, 20))
    want = MF.prove(okeys, zz, FR.ChaChaRng(seed, 20))
    assert got.evaluations == want.evaluations
    assert got.serialize(ctx) == want.serialize()
    pub = zz[1:oix.num_instance]
    dev_as_oracle = MF.Proof([[(c.comm_aff, c.shifted_aff, c.shifted is not None) for c in rnd] for rnd in got.commitments],
                             got.evaluations, [(cv.g1_projective_to_affine(w), rv) for w, rv in got.pc_proof])
    assert MF.verify(okeys, pub, dev_as_oracle)
    assert not MF.verify(okeys, [(pub[0] + 1) % O.R_MOD] + pub[1:], dev_as_oracle)


@pytest.mark.parametrize("n", [1000, (1 << 14) - 3, "dense12", "dense16"])
def test_marlin_proof_verifies_at_size(ctx, n):
    """Beyond what the Python prover does in seconds: the oracle's VERIFIER (transcript re-derived from the proof, the two
    sum-check combinations, degree-bound adjustments, one pairing equation per query point) accepts the device's proof, given
    the device's index commitments -- which the small cases above tie to the oracle's own -- and rejects a wrong public input
    and a tampered evaluation."""
    import marlin_full_ref as MF
    from zk_mpc_amd.api import Rng
    rng, r1cs, sq, zz, dix = build(ctx, n, 5000 + ((n & 0xff) if isinstance(n, int) else int(n[5:])))
    if not isinstance(n, int):
        assert dix.dom_k.size == 4 * dix.dom_h.size == 4 << int(n[5:]) and dix.num_instance == 8
    beta, g_k, gg_k, h_k = rng.fr(), rng.fr(), rng.fr(), rng.fr()
    max_degree = DM.ahp_max_degree(dix) + 5
    srs = DM.UniversalSrs(ctx, max_degree, beta, g_k, gg_k)
    dkeys = DM.IndexKeys(dix, srs)
    proof = DM.prove(dkeys, ctx.upload(cv.fr_to_mont(zz)), Rng.from_seed(bytes(range(32)), 20), mask_on_device=(not isinstance(n, int) or n > 1000))
    if not isinstance(n, int):          # the one-call prover on the same index, same generator: the same bytes
        assert DM.prove_native(dkeys, ctx.upload(cv.fr_to_mont(zz)), Rng.from_seed(bytes(range(32)), 20), mask_on_device=True) == proof.serialize(ctx)

    class PP:                                        # what the verifier key holds (kzg10::VerifierKey), from the toxic waste
        pass
    pp = PP()
    pp.beta = beta
    pp.g, pp.gamma_g, pp.h = O.g1_mul(O.G1_GEN, g_k), O.g1_mul(O.G1_GEN, gg_k), O.g2_mul(O.G2_GEN, h_k)
    pp.beta_h = O.g2_mul(pp.h, beta)
    info = M.IndexInfo(dix.num_constraints, dix.num_non_zero, dix.num_instance)
    info.num_variables, info.num_constraints, info.num_non_zero = dix.num_variables, dix.num_constraints, dix.num_non_zero
    okeys = MF.Keys(info, pp, max_degree=max_degree, index_comms={l: dkeys.index_comms[l].comm_aff for l in MF.INDEX_LABELS})
    assert okeys.ivk_bytes() == dkeys.ivk_bytes()
    as_oracle = MF.Proof([[(c.comm_aff, c.shifted_aff, c.shifted is not None) for c in rnd] for rnd in proof.commitments],
                         proof.evaluations, [(cv.g1_projective_to_affine(w), rv) for w, rv in proof.pc_proof])
    pub = zz[1:dix.num_instance]
    assert MF.verify(okeys, pub, as_oracle)
    assert not MF.verify(okeys, [(pub[0] + 1) % O.R_MOD] + pub[1:], as_oracle)
    ev = list(proof.evaluations); ev[0] = (ev[0] + 1) % O.R_MOD
    assert not MF.verify(okeys, pub, MF.Proof(as_oracle.commitments, ev, as_oracle.pc_proof))
    assert len(proof.serialize(ctx)) == 8 + 3 * 8 + 9 * 49 + 2 * 48 + 8 + 7 * 32 + 8 + 3 + 8 + 2 * 49 + 32 + 1


@pytest.mark.parametrize("n", [3, 13, 1000, "tiny9", "dense8"])
def test_native_marlin_prove_equals_the_python_sequence(ctx, n):
    """zk_marlin_prove (one C++ entry point) emits the bytes of marlin.py::prove (the ~200-call sequence the collaborative
    provers build on) from the same prover rng -- and, through it, of the oracle's prover at the small sizes -- for the
    host-drawn and the device-sampled mask polynomial; an unsatisfied system is refused."""
    from zk_mpc_amd.api import Rng
    rng, r1cs, sq, zz, dix = build(ctx, n, 6000 + (n if isinstance(n, int) else 77 + int(n[-1])))
    srs = DM.UniversalSrs(ctx, DM.ahp_max_degree(dix) + 2, rng.fr(), rng.fr(), rng.fr())
    keys = DM.IndexKeys(dix, srs)
    z = ctx.upload(cv.fr_to_mont(zz))
    seed = bytes((3 * i + 1) & 0xff for i in range(32))
    for on_dev in (False, True):
        want = DM.prove(keys, z, Rng.from_seed(seed, 20), mask_on_device=on_dev).serialize(ctx)
        got = DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=on_dev)
        assert got == want
        assert DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=on_dev) == want      # scratch reuse across calls
    bad = list(zz); bad[dix.num_instance + 1] = (bad[dix.num_instance + 1] + 1) % O.R_MOD
    with pytest.raises(Exception, match="sum over H|divisible"):
        DM.prove_native(keys, ctx.upload(cv.fr_to_mont(bad)), Rng.from_seed(seed, 20))
