"""oracle/marlin_full_ref.py -- TEST INFRASTRUCTURE ONLY (imported by tests/ and bench.py's checker, never by the product).

Marlin as a PROOF on the CPU, restated from the reference with Python integers and naive MSMs (seconds for |H| <= 64):
  Marlin::{index, prove, verify}                      arkworks/marlin/src/lib.rs:100-442
  FiatShamirRng absorb order / to_bytes! encodings     lib.rs:161-164,187,213,236,296-300; marlin/src/data_structures.rs:36-43;
                                                       ahp/indexer.rs:44-50; ahp/prover.rs:76-83; ff/src/bytes.rs (Vec<T>: items back to
                                                       back, no length); ec/.../short_weierstrass_jacobian.rs:315-322 (x | y | infinity);
                                                       poly-commit/src/marlin/marlin_pc/data_structures.rs:252-263 (comm | bool | shifted)
  verifier messages / query set / linear combinations  ahp/verifier.rs:42-170, ahp/mod.rs:112-290 (marlin_ref.sumcheck_equations' terms)
  MarlinKZG10::{trim, commit, open, check}             poly-commit/src/marlin/marlin_pc/mod.rs:83-400
  Marlin::{open,check}_combinations, accumulate...     poly-commit/src/marlin/mod.rs:33-420
  batch_open / batch_check (per query point, BTree order)   poly-commit/src/lib.rs:240-312,399-458
  KZG10::{setup, commit, open, check}                  poly-commit/src/kzg10/mod.rs:44-343 (zkref.KzgParams / kzg_*)
  Proof / Commitment / kzg10::Proof CanonicalSerialize marlin/src/data_structures.rs:99-110; marlin_pc/data_structures.rs:242-250;
                                                       kzg10/data_structures.rs:532-538
Parity note: the reference holds no recorded Marlin proofs and cannot be run here, so the byte-level transcript is pinned only
by reading (the citations above) and by the generators' RFC vectors (tests/test_fsrng.py); what IS pinned end to end is the
protocol: verify() accepts the device prover's proof, rejects a wrong public input / a tampered proof, and prove() here
and the device prover emit identical bytes from the same rng.
"""
from __future__ import annotations

from typing import Dict, List

import fsrng_ref as FR
import marlin_ref as M
import zkref as O

P = O.R_MOD
Q = O.Q_MOD
INDEX_LABELS = [m + s for m in "abc" for s in ("_row", "_col", "_val", "_row_col")]
PROVER_LABELS = ["w", "z_a", "z_b", "mask_poly", "t", "g_1", "h_1", "g_2", "h_2"]
ROUNDS = [["w", "z_a", "z_b", "mask_poly"], ["t", "g_1", "h_1"], ["g_2", "h_2"]]
PROTOCOL_NAME = b"MARLIN-2019"


# ---- to_bytes! ----------------------------------------------------------------------------------------------------------------
def fr_bytes(v: int) -> bytes:
    return (v % P).to_bytes(32, "little")


def g1_bytes(pt) -> bytes:
    """GroupAffine::write: x | y | infinity; zero() is (0, 1, true)."""
    if pt is None:
        return (0).to_bytes(48, "little") + (1).to_bytes(48, "little") + b"\x01"
    return pt[0].to_bytes(48, "little") + pt[1].to_bytes(48, "little") + b"\x00"


def comm_bytes(c) -> bytes:
    """marlin_pc::Commitment::write: comm | shifted_exists | shifted_comm (or the empty commitment)."""
    comm, shifted, has_shift = c
    return g1_bytes(comm) + (b"\x01" if has_shift else b"\x00") + g1_bytes(shifted if has_shift else None)


# ---- keys -----------------------------------------------------------------------------------------------------------------------
def max_degree_for(index: M.Index) -> int:
    """AHPForR1CS::max_degree (ahp/mod.rs:75-97) with zk_bound = 1."""
    h, k = index.dom_h.size, index.dom_k.size
    return max(2 * h + 1 - 2, 3 * h + 2 - 3, h, 3 * k - 3)


def degree_bounds(index) -> Dict[str, int]:
    return {"g_1": index.dom_h.size - 2, "g_2": index.dom_k.size - 2}          # get_degree_bounds (ahp/mod.rs:100-110)


HIDING = {"w": 1, "z_a": 1, "z_b": 1, "g_1": 1}                                 # hiding bounds of the oracles (prover.rs:383-387,548-552)


class Keys:
    """(IndexProverKey, IndexVerifierKey) of Marlin::index over a KZG10 setup `pp` (zkref.KzgParams, max_degree >= the index's)."""

    def __init__(self, index, pp, max_degree=None, index_comms=None):
        """index: marlin_ref.Index (prover + verifier) or marlin_ref.IndexInfo-like with num_variables / num_constraints /
        num_non_zero / dom_h / dom_k / num_instance (verifier only, together with index_comms = {label: affine point}: the
        commitments of an index too large to commit to with Python MSMs; pp then only needs g, gamma_g, h, beta_h, beta)."""
        self.index, self.pp = index, pp
        self.max_degree = max_degree if max_degree is not None else len(pp.powers_of_g) - 1
        assert self.max_degree >= max_degree_for(index), "IndexTooLarge"
        self.bounds = degree_bounds(index)
        if index_comms is None:
            self.index_polys = index.polynomials()
            self.index_comms = {l: (O.kzg_commit(pp, self.index_polys[l]), None, False) for l in INDEX_LABELS}   # rng = None: no hiding
        else:
            self.index_polys = None
            self.index_comms = {l: (index_comms[l], None, False) for l in INDEX_LABELS}

    def ivk_bytes(self) -> bytes:
        ix = self.index
        out = ix.num_variables.to_bytes(8, "little") + ix.num_constraints.to_bytes(8, "little") + ix.num_non_zero.to_bytes(8, "little")
        return out + b"".join(comm_bytes(self.index_comms[l]) for l in INDEX_LABELS)

    def shift_power(self, bound: int):
        """VerifierKey::get_shift_power: powers_of_g[max_degree - bound] (from the toxic waste when the table is not held)."""
        if getattr(self.pp, "powers_of_g", None) is not None and len(self.pp.powers_of_g) > self.max_degree - bound:
            return self.pp.powers_of_g[self.max_degree - bound]
        return O.g1_mul(self.pp.g, pow(self.pp.beta, self.max_degree - bound, P))


# ---- MarlinKZG10::commit with the reference's rng order ---------------------------------------------------------------------------
def commit_round(keys: Keys, labels, polys, zk_rng):
    """PC::commit(ck, oracles, Some(zk_rng)): per oracle KZG10::commit (hiding: a random polynomial of degree hiding_bound + 1,
    three coefficients drawn in order), then, for a degree-bounded oracle, the commitment over the shifted powers with its
    own random polynomial.  Returns (commitments, randomness) keyed by label; a randomness is (blind, shifted_blind)."""
    pp = keys.pp
    comms, rands = {}, {}
    for l in labels:
        p = polys[l]
        hb = HIDING.get(l)
        blind = [zk_rng.next_fr() for _ in range(hb + 2)] if hb is not None else []
        comm = O.kzg_commit(pp, p, blind or None)
        shifted, sblind = None, None
        if l in keys.bounds:
            d = keys.bounds[l]
            assert len(M.strip(p)) - 1 <= d
            sblind = [zk_rng.next_fr() for _ in range(hb + 2)] if hb is not None else []
            shifted = O.msm_naive(pp.powers_of_g[keys.max_degree - d:], p, O.FqOps)
            if sblind:
                shifted = O.g1_add(shifted, O.msm_naive(pp.powers_of_gamma_g, sblind, O.FqOps))
        comms[l] = (comm, shifted, l in keys.bounds)
        rands[l] = (blind, sblind)
    return comms, rands


# ---- verifier messages ------------------------------------------------------------------------------------------------------------
def sample_outside(dom, fs) -> int:
    t = fs.next_fr()
    while dom.evaluate_vanishing_polynomial(t) == 0:
        t = fs.next_fr()
    return t


def linear_combinations(index, public_input, ch, ev):
    """construct_linear_combinations (ahp/mod.rs:112-290): label -> list of (coefficient, polynomial label or None for One),
    sorted by label.  ev(label) -> the evaluation the verifier has (or the prover computes) for the four single-polynomial
    combinations at their query points and the three denominators at gamma."""
    H, K = index.dom_h, index.dom_k
    alpha, eta_a, eta_b, eta_c, beta, gamma = (ch[k] for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma"))
    x = [1] + list(public_input)
    Xd = O.Domain(len(x))
    r_alpha_at_beta = M.eval_unnormalized_bivariate_lagrange_poly(H, alpha, beta)
    v_H_alpha, v_H_beta = H.evaluate_vanishing_polynomial(alpha), H.evaluate_vanishing_polynomial(beta)
    v_X_beta = Xd.evaluate_vanishing_polynomial(beta)
    z_b_beta, t_beta, g_1_beta = ev("z_b"), ev("t"), ev("g_1")
    x_beta = sum(l * xv for l, xv in zip(Xd.evaluate_all_lagrange_coefficients(beta), x)) % P
    lcs = {"z_b": [(1, "z_b")], "g_1": [(1, "g_1")], "t": [(1, "t")], "g_2": [(1, "g_2")]}
    lcs["outer_sumcheck"] = [(1, "mask_poly"),
                             (r_alpha_at_beta * ((eta_a + eta_c * z_b_beta) % P) % P, "z_a"),
                             (r_alpha_at_beta * eta_b % P * z_b_beta % P, None),
                             ((-t_beta * v_X_beta) % P, "w"),
                             ((-t_beta * x_beta) % P, None),
                             ((-v_H_beta) % P, "h_1"),
                             ((-beta * g_1_beta) % P, None)]
    ba = beta * alpha % P
    for m in "abc":
        lcs[m + "_denom"] = [(ba, None), ((-alpha) % P, m + "_row"), ((-beta) % P, m + "_col"), (1, m + "_row_col")]
    da, db, dc, g_2_gamma = ev("a_denom"), ev("b_denom"), ev("c_denom"), ev("g_2")
    v_K_gamma = K.evaluate_vanishing_polynomial(gamma)
    vv = v_H_alpha * v_H_beta % P
    b_expr = da * db % P * dc % P * ((gamma * g_2_gamma + t_beta * pow(K.size, -1, P)) % P) % P
    lcs["inner_sumcheck"] = [(eta_a * db % P * dc % P * vv % P, "a_val"), (eta_b * da % P * dc % P * vv % P, "b_val"),
                             (eta_c * db % P * da % P * vv % P, "c_val"), ((-b_expr) % P, None), ((-v_K_gamma) % P, "h_2")]
    return dict(sorted(lcs.items()))


QUERY = {"beta": ["g_1", "outer_sumcheck", "t", "z_b"], "gamma": ["a_denom", "b_denom", "c_denom", "g_2", "inner_sumcheck"]}
EVAL_LABELS = ["a_denom", "b_denom", "c_denom", "g_1", "g_2", "t", "z_b"]       # the proof's evaluations, sorted by label


class Proof:
    def __init__(self, commitments, evaluations, pc_proof):
        self.commitments = commitments          # [[(comm, shifted, has_shift), ...] per round]
        self.evaluations = evaluations          # EVAL_LABELS order
        self.pc_proof = pc_proof                # [(w, random_v or None)] for beta, gamma

    def serialize(self) -> bytes:
        """CanonicalSerialize of marlin::Proof (derive order: commitments, evaluations, prover_messages, pc_proof)."""
        u64 = lambda v: v.to_bytes(8, "little")
        out = u64(len(self.commitments))
        for rnd in self.commitments:
            out += u64(len(rnd))
            for comm, shifted, has in rnd:
                out += O.g1_serialize(comm) + (b"\x01" + O.g1_serialize(shifted) if has else b"\x00")
        out += u64(len(self.evaluations)) + b"".join(fr_bytes(e) for e in self.evaluations)
        out += u64(3) + b"\x00" * 3                                           # three EmptyMessage: Option::None each
        out += u64(len(self.pc_proof))
        for w, rv in self.pc_proof:
            out += O.g1_serialize(w) + (b"\x01" + fr_bytes(rv) if rv is not None else b"\x00")
        return out + b"\x00"                                                  # BatchLCProof.evals = None


def proof_deserialize(data: bytes) -> Proof:
    """CanonicalDeserialize of marlin::Proof (data_structures.rs:99-110; the inverse of Proof.serialize above): what a verifier on
    the other side of the wire does with the prover's bytes.  Points go through GroupAffine::deserialize (curve and subgroup
    checked), scalars must be canonical, nothing may trail."""
    pos = 0

    def take(n):
        nonlocal pos
        assert pos + n <= len(data), "truncated proof"
        out = data[pos:pos + n]
        pos += n
        return out

    def u64():
        return int.from_bytes(take(8), "little")

    def fr():
        v = int.from_bytes(take(32), "little")
        assert v < P, "non-canonical scalar"
        return v
    commitments = []
    for _ in range(u64()):
        rnd = []
        for _ in range(u64()):
            comm = O.g1_deserialize(take(48))
            has = take(1)[0]
            assert has in (0, 1)
            rnd.append((comm, O.g1_deserialize(take(48)) if has else None, bool(has)))
        commitments.append(rnd)
    evaluations = [fr() for _ in range(u64())]
    assert u64() == 3 and take(3) == b"\x00" * 3                   # three EmptyMessage
    pc_proof = []
    for _ in range(u64()):
        w = O.g1_deserialize(take(48))
        has = take(1)[0]
        assert has in (0, 1)
        pc_proof.append((w, fr() if has else None))
    assert take(1) == b"\x00" and pos == len(data)                 # BatchLCProof.evals = None, then the end
    return Proof(commitments, evaluations, pc_proof)


def transcript_challenges(keys_ivk_bytes: bytes, index, public_input, commitments, evaluations=None):
    """The verifier's side of the Fiat-Shamir transcript; returns (challenges, fs) with fs positioned after gamma (the caller
    absorbs the evaluations and draws the opening challenge)."""
    fs = FR.FiatShamirRng(PROTOCOL_NAME + keys_ivk_bytes + b"".join(fr_bytes(v) for v in public_input))
    ch = {}
    fs.absorb(b"".join(comm_bytes(c) for c in commitments[0]))
    ch["alpha"] = sample_outside(index.dom_h, fs)
    ch["eta_a"], ch["eta_b"], ch["eta_c"] = fs.next_fr(), fs.next_fr(), fs.next_fr()
    fs.absorb(b"".join(comm_bytes(c) for c in commitments[1]))
    ch["beta"] = sample_outside(index.dom_h, fs)
    fs.absorb(b"".join(comm_bytes(c) for c in commitments[2]))
    ch["gamma"] = fs.next_fr()
    return ch, fs


def prove(keys: Keys, full_assignment: List[int], zk_rng) -> Proof:
    """Marlin::prove.  zk_rng: an fsrng_ref.ChaChaRng (next_fr) -- the prover's randomness, drawn in the reference's order."""
    index, pp = keys.index, keys.pp
    st = M.prover_init(index, full_assignment)
    public_input = list(full_assignment[1:index.num_instance])
    fs = FR.FiatShamirRng(PROTOCOL_NAME + keys.ivk_bytes() + b"".join(fr_bytes(v) for v in public_input))
    polys = dict(keys.index_polys)
    rands = {l: ([], None) for l in INDEX_LABELS}
    comms = dict(keys.index_comms)
    # round 1 (prover.rs:311-404): F::rand x 3, then the mask polynomial's coefficients
    r = [zk_rng.next_fr() for _ in range(3)]
    mask = [zk_rng.next_fr() for _ in range(M.mask_poly_degree(index) + 1)]
    polys.update(M.prover_first_round(st, r[0], r[1], r[2], mask))
    c1, r1 = commit_round(keys, ROUNDS[0], polys, zk_rng)
    comms.update(c1); rands.update(r1)
    fs.absorb(b"".join(comm_bytes(c1[l]) for l in ROUNDS[0]))
    ch = {"alpha": sample_outside(index.dom_h, fs)}
    ch["eta_a"], ch["eta_b"], ch["eta_c"] = fs.next_fr(), fs.next_fr(), fs.next_fr()
    polys.update(M.prover_second_round(st, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"]))
    c2, r2 = commit_round(keys, ROUNDS[1], polys, zk_rng)
    comms.update(c2); rands.update(r2)
    fs.absorb(b"".join(comm_bytes(c2[l]) for l in ROUNDS[1]))
    ch["beta"] = sample_outside(index.dom_h, fs)
    polys.update(M.prover_third_round(st, ch["beta"]))
    c3, r3 = commit_round(keys, ROUNDS[2], polys, zk_rng)
    comms.update(c3); rands.update(r3)
    fs.absorb(b"".join(comm_bytes(c3[l]) for l in ROUNDS[2]))
    ch["gamma"] = fs.next_fr()
    point = {"beta": ch["beta"], "gamma": ch["gamma"]}
    # evaluations of the single-polynomial combinations and the denominators (lib.rs:279-294)
    ev_poly = lambda l, pt: M.evaluate(polys[l], pt)
    single = {"z_b": ev_poly("z_b", ch["beta"]), "g_1": ev_poly("g_1", ch["beta"]), "t": ev_poly("t", ch["beta"]),
              "g_2": ev_poly("g_2", ch["gamma"])}
    ba = ch["beta"] * ch["alpha"] % P
    for m in "abc":
        single[m + "_denom"] = (ba - ch["alpha"] * ev_poly(m + "_row", ch["gamma"]) - ch["beta"] * ev_poly(m + "_col", ch["gamma"])
                                + ev_poly(m + "_row_col", ch["gamma"])) % P
    lcs = linear_combinations(index, public_input, ch, lambda l: single[l])
    evaluations = [single[l] for l in EVAL_LABELS]
    fs.absorb(b"".join(fr_bytes(e) for e in evaluations))
    xi = fs.next_u128() % P
    # open_combinations (marlin/mod.rs:213-306) + batch_open per query point (lib.rs:240-312) + open (marlin_pc/mod.rs:245-340)
    pc_proof = []
    for pl in ("beta", "gamma"):
        z = point[pl]
        p_comb, r_comb, sw, sr, srw = [], [], [], [], []
        j = 0
        for label in QUERY[pl]:
            lc = lcs[label]
            poly, rnd, bound = [], [], None
            terms = [(c, l) for c, l in lc if l is not None]
            for c, l in terms:
                if len(lc) == 1 and l in keys.bounds:
                    assert c == 1
                    bound = keys.bounds[l]
                poly = M.padd(poly, M.pscale(polys[l], c))
                rnd = M.padd(rnd, M.pscale(rands[l][0], c))
            cj = pow(xi, j, P); j += 1
            p_comb = M.padd(p_comb, M.pscale(poly, cj))
            r_comb = M.padd(r_comb, M.pscale(rnd, cj))
            if bound is not None:
                src = terms[0][1]
                cj1 = pow(xi, j, P); j += 1
                wit, _ = O.poly_divide_with_q_and_r(poly, [(-z) % P, 1])
                sw.append((cj1, wit, bound))
                sb = rands[src][1] or []
                sr = M.padd(sr, M.pscale(sb, cj1))
                if sb:
                    srw = M.padd(srw, M.pscale(O.poly_divide_with_q_and_r(sb, [(-z) % P, 1])[0], cj1))
        hiding = any(v % P for v in r_comb)
        w, rv = O.kzg_open(pp, p_comb, z, r_comb if hiding else None)
        if sw:
            for cj1, wit, bound in sw:
                w = O.g1_add(w, O.msm_naive(pp.powers_of_g[keys.max_degree - bound:], M.pscale(wit, cj1), O.FqOps))
            if srw:
                w = O.g1_add(w, O.msm_naive(pp.powers_of_gamma_g, srw, O.FqOps))
            if rv is not None:
                rv = (rv + O.poly_evaluate(sr, z)) % P          # random_v.map(|v| v + shifted_random_v)
        pc_proof.append((w, rv))
    return Proof([[comms[l] for l in rnd] for rnd in ROUNDS], evaluations, pc_proof)


def verify(keys: Keys, public_input: List[int], proof: Proof) -> bool:
    """Marlin::verify (lib.rs:324-442) with MarlinKZG10::check_combinations; one pairing equation per query point."""
    index, pp = keys.index, keys.pp
    n_in = len(public_input) + 1
    public_input = list(public_input) + [0] * (max(len(public_input), O.Domain(n_in).size - 1) - len(public_input))
    ch, fs = transcript_challenges(keys.ivk_bytes(), index, public_input, proof.commitments)
    fs.absorb(b"".join(fr_bytes(e) for e in proof.evaluations))
    xi = fs.next_u128() % P
    given = dict(zip(EVAL_LABELS, proof.evaluations))
    comms = dict(keys.index_comms)
    for rnd, cs in zip(ROUNDS, proof.commitments):
        comms.update(dict(zip(rnd, cs)))
    for l, (_, shifted, has) in comms.items():
        if has != (l in keys.bounds):
            return False
    lcs = linear_combinations(index, public_input, ch, lambda l: given[l])
    point = {"beta": ch["beta"], "gamma": ch["gamma"]}
    ok = True
    for (pl, labels), (w, rv) in zip(QUERY.items(), proof.pc_proof):
        z = point[pl]
        acc, val = None, 0
        j = 0
        for label in labels:
            lc = lcs[label]
            v = given.get(label, 0)                                   # the two sum-checks evaluate to zero
            c_lc, s_lc, bound = None, None, None
            for c, l in lc:
                if l is None:
                    v = (v - c) % P                                    # constant terms move to the value (marlin/mod.rs:343-350)
                    continue
                comm, shifted, has = comms[l]
                if len(lc) == 1 and has:
                    if c != 1:
                        return False
                    bound = keys.bounds[l]
                    s_lc = shifted
                elif has:
                    return False                                       # EquationHasDegreeBounds
                c_lc = O.g1_add(c_lc, O.g1_mul(comm, c))
            cj = pow(xi, j, P); j += 1
            acc = O.g1_add(acc, O.g1_mul(c_lc, cj))
            val = (val + v * cj) % P
            if bound is not None:
                cj1 = pow(xi, j, P); j += 1
                adj = O.g1_add(s_lc, O.g1_neg(O.g1_mul(keys.shift_power(bound), v)))
                acc = O.g1_add(acc, O.g1_mul(adj, cj1))
        ok = ok and O.kzg_check(pp, acc, z, val, w, rv)
    return ok
