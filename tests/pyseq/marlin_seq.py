"""The Marlin prover as a round-by-round PYTHON sequence of C-ABI calls: test infrastructure.

Mirrors AHPForR1CS::{prover_init, prover_first_round, prover_second_round, prover_third_round}
(arkworks/marlin/src/ahp/prover.rs:216-716) and the commit / open calls of Marlin::prove (arkworks/marlin/src/lib.rs:152-319)
one call at a time -- the form a host that keeps the AHP rounds in its own language would write, and the form
the collaborative sequences of mpc_seq.py build on.  The PRODUCT is zk_marlin_prove / zk_marlin_prove_shared[_spdz] (one call,
csrc/marlin_prove.hip, reached through zk_mpc_amd.marlin.prove_native); this module is the second implementation the tests
compare those with, byte for byte.  Everything of zk_mpc_amd.marlin is re-exported, so `import pyseq.marlin_seq as DM` serves both.
"""
from __future__ import annotations

import numpy as np

from zk_mpc_amd import _lib
from zk_mpc_amd import convert as cv
from zk_mpc_amd.api import Context, DevBuf
from zk_mpc_amd.marlin import *          # noqa: F401,F403
from zk_mpc_amd.marlin import (_fr_bytes, _g1_to_bytes, _log2, _next_pow2)      # noqa: F401
from zk_mpc_amd.marlin import _FR_TWO_ADICITY, _FR_TWO_ADIC_ROOT_MONT      # noqa: F401


class ProverState:
    pass


def prover_init(index: Index, assignment_dev: DevBuf, shared: bool = False) -> ProverState:
    """prover.rs:216-309: z_A = A z, z_B = B z.  assignment_dev: the full (padded) assignment, instance first.
    shared: the assignment (and later the randomness) is this party's additive share; every step of the rounds is linear
    in it except the product z_A * z_B of round 2, which then goes through the caller's Beaver multiplication, and the
    zero tests, which go through the caller's open."""
    ctx, H = index.ctx, index.dom_h.size
    st = ProverState()
    st.shared = shared
    st.index, st.z = index, assignment_dev
    st.z_a, st.z_b = ctx.alloc(H * 32), ctx.alloc(H * 32)
    ctx.r1cs_matvec_dev(index.r1cs, 0, assignment_dev.ptr, st.z_a.ptr, H)
    ctx.r1cs_matvec_dev(index.r1cs, 1, assignment_dev.ptr, st.z_b.ptr, H)
    st.zk_bound = 1
    return st


def mask_poly_degree(index: Index) -> int:
    return 3 * index.dom_h.size + 2 * 1 - 3


def _blind_with_vanishing(ctx, poly: DevPoly, n: int, r_dev: int) -> DevPoly:
    """p + r (X^n - 1) for deg p < n: one more coefficient."""
    out = ctx.alloc((n + 1) * 32)
    ctx.memcpy_d2d(out.ptr, poly.ptr, n * 32)
    ctx.memcpy_d2d(out.ptr + 32 * n, r_dev, 32)
    ctx.fr_vec_op_dev(_lib.OP_SUB, out.ptr, r_dev, out.ptr, 1)
    return DevPoly(out, n + 1)


def prover_first_round(st: ProverState, randomness):
    """prover.rs:311-404.  randomness: 3 + mask_poly_degree + 1 field elements in the order the reference draws them
    (w, z_a, z_b blinders, then the mask polynomial's coefficients), as Montgomery limbs (n, 4) on the host or as a
    DevBuf already holding them."""
    ix = st.index
    ctx, H, X = ix.ctx, ix.dom_h, ix.dom_x
    n = H.size
    md = mask_poly_degree(ix)
    if isinstance(randomness, DevBuf):
        assert randomness.nbytes >= (3 + md + 1) * 32
        rnd = randomness
    else:
        assert randomness.shape == (3 + md + 1, 4)
        rnd = ctx.upload(randomness)
    # x(X): interpolation of the formatted input over X, then its evaluations over H
    xb = ctx.alloc(X.size * 32)
    ctx.memcpy_d2d(xb.ptr, st.z.ptr, X.size * 32)
    st.x_poly = X.ifft_in_place(ctx, xb)
    x_evals = H.fft(ctx, st.x_poly)
    d_iw, d_ix = ix.w_evals_index()
    w_evals, tmp = ctx.alloc(n * 32), ctx.alloc(n * 32)
    ctx.fr_gather_dev(st.z.ptr, d_iw.ptr, n, w_evals.ptr)
    ctx.fr_gather_dev(x_evals.ptr, d_ix.ptr, n, tmp.ptr)
    ctx.fr_vec_op_dev(_lib.OP_SUB, w_evals.ptr, tmp.ptr, w_evals.ptr, n)
    w_h = _blind_with_vanishing(ctx, H.ifft_in_place(ctx, w_evals), n, rnd.ptr)
    wq, wr = ctx.alloc(max(n + 1 - X.size, 1) * 32), ctx.alloc(X.size * 32)
    ctx.poly_divide_by_vanishing_dev(w_h.ptr, n + 1, X.log, wq.ptr, wr.ptr)
    if not st.shared and not ctx.fr_vec_is_zero_dev(wr.ptr, X.size):
        raise ValueError("w polynomial is not divisible by v_X")      # assert!(remainder.is_zero()), prover.rs:360
    st.w_poly = DevPoly(wq, n + 1 - X.size)
    za, zb = ctx.alloc(n * 32), ctx.alloc(n * 32)
    ctx.memcpy_d2d(za.ptr, st.z_a.ptr, n * 32)
    ctx.memcpy_d2d(zb.ptr, st.z_b.ptr, n * 32)
    st.z_a_poly = _blind_with_vanishing(ctx, H.ifft_in_place(ctx, za), n, rnd.ptr + 32)
    st.z_b_poly = _blind_with_vanishing(ctx, H.ifft_in_place(ctx, zb), n, rnd.ptr + 64)
    mask = ctx.alloc((md + 1) * 32)
    ctx.memcpy_d2d(mask.ptr, rnd.ptr + 96, (md + 1) * 32)
    mq, mr = ctx.alloc((md + 1) * 32), ctx.alloc(n * 32)
    ctx.poly_divide_by_vanishing_dev(mask.ptr, md + 1, H.log, mq.ptr, mr.ptr)
    ctx.fr_vec_op_dev(_lib.OP_SUB, mask.ptr, mr.ptr, mask.ptr, 1)     # mask[0] -= remainder[0]: sum over H becomes zero
    st.mask_poly = DevPoly(mask, md + 1)
    ctx.sync()
    return {"w": st.w_poly, "z_a": st.z_a_poly, "z_b": st.z_b_poly, "mask_poly": st.mask_poly}


def prover_second_round(st: ProverState, alpha: int, eta_a: int, eta_b: int, eta_c: int, batch_mul=None, open_is_zero=None):
    """prover.rs:438-565.  batch_mul(x_dev, y_dev, out_dev, n): element-wise product of two vectors of the prover's own
    values (default: the local product; over shares: FieldShare::batch_mul, as `DensePolynomial::mul` on MpcField does
    through batch_product_in_place).  open_is_zero(v_dev, n): whether the (shared) vector opens to zero."""
    ctx = st.index.ctx
    if batch_mul is None:
        batch_mul = lambda x, y, out, k: ctx.fr_vec_op_dev(_lib.OP_MUL, x, y, out, k)
    if open_is_zero is None:
        open_is_zero = lambda v, k: ctx.fr_vec_is_zero_dev(v, k)
    steps = second_round_steps(st, alpha, eta_a, eta_b, eta_c)
    req = next(steps)
    try:
        while True:
            if req[0] == "mul":
                batch_mul(*req[1:])
                req = steps.send(None)
            else:
                req = steps.send(open_is_zero(*req[1:]))
    except StopIteration as done:
        return done.value


def second_round_steps(st: ProverState, alpha: int, eta_a: int, eta_b: int, eta_c: int):
    """The second round as a generator that hands the two witness-dependent operations to its driver:
    yields ("mul", x_dev, y_dev, out_dev, n) for z_A * z_B on the multiplication domain, then ("zero", v_dev, n) and
    expects the answer (bool) to be sent back; returns the round's oracles.  Lets a SPDZ prover advance its share lane and
    its MAC lane in lock-step around one joint Beaver multiplication."""
    ix = st.index
    ctx, H, X, F = ix.ctx, ix.dom_h, ix.dom_x, ix.dom_h.F
    n = H.size
    m = HostField.m
    # r(alpha, X) on H: v_H(alpha) / (alpha - h)   (mod.rs:352-360)
    v_h_alpha = H.evaluate_vanishing_polynomial(alpha)
    ra = ctx.alloc(n * 32)
    ctx.fr_powers_dev(m(1), m(alpha), n, ra.ptr)                       # the constant vector alpha
    ctx.fr_vec_op_dev(_lib.OP_SUB, ra.ptr, H.elements().ptr, ra.ptr, n)
    ctx.batch_inversion_dev(ra.ptr, n)
    ctx.fr_vec_scale_dev(ra.ptr, m(v_h_alpha), ra.ptr, n)
    # t = sum_M eta_M M^T r  (calculate_t), interpolated over H
    t_ev, t_tmp = ctx.alloc(n * 32), ctx.alloc(n * 32)
    for which, eta in enumerate((eta_a, eta_b, eta_c)):
        ctx.r1cs_matvec_dev(ix.r1cs_t, which, ra.ptr, t_tmp.ptr, n)
        if which == 0:
            ctx.fr_vec_scale_dev(t_tmp.ptr, m(eta), t_ev.ptr, n)
        else:
            ctx.fr_vec_scale_dev(t_tmp.ptr, m(eta), t_tmp.ptr, n)
            ctx.fr_vec_op_dev(_lib.OP_ADD, t_ev.ptr, t_tmp.ptr, t_ev.ptr, n)
    st.t_poly = H.ifft_in_place(ctx, t_ev)
    r_alpha_poly = H.ifft_in_place(ctx, ra)
    # z = w v_X + x
    zp = ctx.alloc((n + 1) * 32)
    nw = st.w_poly.n
    ctx.dev_zero(zp.ptr, (n + 1) * 32)
    ctx.memcpy_d2d(zp.ptr + 32 * X.size, st.w_poly.ptr, nw * 32)
    ctx.fr_vec_op_dev(_lib.OP_SUB, zp.ptr, st.w_poly.ptr, zp.ptr, nw)
    ctx.fr_vec_op_dev(_lib.OP_ADD, zp.ptr, st.x_poly.ptr, zp.ptr, X.size)
    z_poly = DevPoly(zp, n + 1)
    # q_1 = mask + r_alpha * (eta_c z_a z_b + eta_a z_a + eta_b z_b) - t * z over one multiplication domain
    # (prover.rs:458-545; summed_z_m has 2n + 1 coefficients, so the domain is the 4n one the reference picks)
    mul = Domain(ctx, max(st.mask_poly.n, n + 2 * n + 1, n + z_poly.n))
    e_a, e_b = mul.fft(ctx, st.z_a_poly), mul.fft(ctx, st.z_b_poly)
    e_s = ctx.alloc(mul.size * 32)
    yield ("mul", e_a.ptr, e_b.ptr, e_s.ptr, mul.size)                # z_c = z_a z_b: the one product of two witness vectors
    ctx.fr_vec_scale_dev(e_s.ptr, m(eta_c), e_s.ptr, mul.size)
    ctx.fr_vec_scale_dev(e_a.ptr, m(eta_a), e_a.ptr, mul.size)
    ctx.fr_vec_op_dev(_lib.OP_ADD, e_s.ptr, e_a.ptr, e_s.ptr, mul.size)
    ctx.fr_vec_scale_dev(e_b.ptr, m(eta_b), e_b.ptr, mul.size)
    ctx.fr_vec_op_dev(_lib.OP_ADD, e_s.ptr, e_b.ptr, e_s.ptr, mul.size)
    e_r, e_z, e_t = mul.fft(ctx, r_alpha_poly), mul.fft(ctx, z_poly), mul.fft(ctx, st.t_poly)
    ctx.fr_vec_op_dev(_lib.OP_MUL, e_r.ptr, e_s.ptr, e_r.ptr, mul.size)      # public * own value: local
    ctx.fr_vec_op_dev(_lib.OP_MUL, e_z.ptr, e_t.ptr, e_z.ptr, mul.size)
    ctx.fr_vec_op_dev(_lib.OP_SUB, e_r.ptr, e_z.ptr, e_r.ptr, mul.size)
    q1 = mul.ifft_in_place(ctx, e_r)
    ctx.fr_vec_op_dev(_lib.OP_ADD, q1.ptr, st.mask_poly.ptr, q1.ptr, st.mask_poly.n)
    hq, hr = ctx.alloc((mul.size - n) * 32), ctx.alloc(n * 32)
    ctx.poly_divide_by_vanishing_dev(q1.ptr, mul.size, H.log, hq.ptr, hr.ptr)
    if not (yield ("zero", hr.ptr, 1)):
        raise ValueError("outer sum-check: the sum over H is not zero (unsatisfied constraint system)")
    st.first_msg = (alpha, eta_a, eta_b, eta_c)
    g_1 = DevPoly(hr, n - 1, 1)
    h_1 = DevPoly(hq, min(mul.size - n, 2 * n + 2 * st.zk_bound - 1))
    ctx.sync()
    return {"t": st.t_poly, "g_1": g_1, "h_1": h_1}


def prover_third_round(st: ProverState, beta: int):
    """prover.rs:583-716."""
    ix = st.index
    ctx, H, K, B, F = ix.ctx, ix.dom_h, ix.dom_k, ix.dom_b, ix.dom_h.F
    alpha, eta_a, eta_b, eta_c = st.first_msg
    m = HostField.m
    vv = F.mul(H.evaluate_vanishing_polynomial(alpha), H.evaluate_vanishing_polynomial(beta))
    etas = [m(eta_a), m(eta_b), m(eta_c)]
    on_k = [{k: v.ptr for k, v in ix.arith[n].evals_on_K.items()} for n in "abc"]
    on_b = [{k: v.ptr for k, v in ix.arith[n].evals_on_B.items()} for n in "abc"]
    f_ev = ctx.alloc(K.size * 32)
    ctx.marlin_round3_f_evals_dev(on_k, K.size, m(alpha), m(beta), etas, m(vv), f_ev.ptr)
    f = K.ifft_in_place(ctx, f_ev)
    g_2 = f.slice(1, K.size - 1)
    a_ev, b_ev = ctx.alloc(B.size * 32), ctx.alloc(B.size * 32)
    ctx.marlin_round3_ab_evals_dev(on_b, B.size, m(alpha), m(beta), etas, m(vv), a_ev.ptr, b_ev.ptr)
    # h_2 = (a - b f) / v_K.  a and b are only ever needed through a - b f, whose degree (<= 4|K| - 4) is below |B|: the
    # product is taken on B itself, where a and b already live as evaluations (the reference interpolates both and
    # multiplies the polynomials, prover.rs:680-698: five transforms of size |B| instead of two)
    if B.size >= 4 * K.size - 3:
        f_on_b = B.fft(ctx, f)
        ctx.fr_vec_op_dev(_lib.OP_MUL, b_ev.ptr, f_on_b.ptr, b_ev.ptr, B.size)
        ctx.fr_vec_op_dev(_lib.OP_SUB, a_ev.ptr, b_ev.ptr, a_ev.ptr, B.size)
        diff = B.ifft_in_place(ctx, a_ev)
        total = B.size
    else:
        # tiny K (|K| = 2: |B| = 4 < 4|K| - 3): the product does not fit B; the reference's way
        a_poly, b_poly = B.ifft_in_place(ctx, a_ev), B.ifft_in_place(ctx, b_ev)
        nb = min(B.size, 3 * K.size - 2)
        total = max(nb + K.size - 1, B.size)
        if total <= K.size:
            raise ValueError("degenerate K domain")
        bf = ctx.alloc(total * 32)
        ctx.dev_zero(bf.ptr, total * 32)
        ctx.poly_mul_dev(b_poly.ptr, nb, f.ptr, K.size, bf.ptr)
        ctx.fr_vec_scale_dev(bf.ptr, m(R_MOD - 1), bf.ptr, nb + K.size - 1)
        ctx.fr_vec_op_dev(_lib.OP_ADD, bf.ptr, a_poly.ptr, bf.ptr, B.size)
        diff = DevPoly(bf, total)
    hq, hr = ctx.alloc((total - K.size) * 32), ctx.alloc(K.size * 32)
    ctx.poly_divide_by_vanishing_dev(diff.ptr, total, K.log, hq.ptr, hr.ptr)
    if not ctx.fr_vec_is_zero_dev(hr.ptr, K.size):
        raise ValueError("inner sum-check: a - b f is not divisible by v_K")
    ctx.sync()
    return {"g_2": g_2, "h_2": DevPoly(hq, total - K.size)}



def linear_combination(ctx: Context, polys, coeffs) -> DevPoly:
    """sum_i coeffs[i] * polys[i] (the combined polynomial of a batched opening, poly-commit/src/lib.rs batch_open)."""
    n = max(p.n for p in polys if p is not None)
    out, tmp = ctx.alloc(n * 32), ctx.alloc(n * 32)
    ctx.dev_zero(out.ptr, n * 32)
    for p, k in zip(polys, coeffs):
        if p is None:            # a public polynomial on a non-leading party: its term belongs to the leader's share
            continue
        ctx.fr_vec_scale_dev(p.ptr, HostField.m(k), tmp.ptr, p.n)
        ctx.fr_vec_op_dev(_lib.OP_ADD, out.ptr, tmp.ptr, out.ptr, p.n)
    return DevPoly(out, n)



def commit_marlin_pc(ctx: Context, powers_g, powers_gamma_g, polys: dict, bounds: dict, blinds: dict = None) -> dict:
    """MarlinKZG10::commit (poly-commit/src/marlin/marlin_pc/mod.rs:172-243): per polynomial the KZG10 commitment, with
    hiding (plus MSM(powers_of_gamma_g, blinding polynomial), kzg10/mod.rs:171-199) where the oracle has a hiding bound,
    and for an oracle with degree bound d a second commitment to the same coefficients over the SHIFTED powers
    powers_of_g[max_degree - d ..] (marlin_pc/data_structures.rs shifted_powers), which is what enforces deg <= d.
    bounds: label -> (degree_bound or None, hiding_bound or None); blinds: label -> (blinding DevPoly, shifted blinding
    DevPoly or None), supplied by the caller's rng.  All MSMs of the call run as one pipelined batch.
    Returns label -> {"comm": G1, "shifted_comm": G1 or None}."""
    max_degree = len(powers_g) - 1
    jobs, slots = [], []
    for label, p in polys.items():
        bound, hiding = bounds.get(label, (None, None))
        jobs.append((powers_g, 0, p.ptr, p.n)); slots.append((label, "comm"))
        if hiding is not None and blinds and label in blinds:
            bl = blinds[label][0]
            jobs.append((powers_gamma_g, 0, bl.ptr, bl.n)); slots.append((label, "comm"))
        if bound is not None:
            if p.n - 1 > bound:
                raise ValueError("%s: degree exceeds its bound" % label)
            jobs.append((powers_g, max_degree - bound, p.ptr, p.n)); slots.append((label, "shifted_comm"))
            if hiding is not None and blinds and label in blinds and blinds[label][1] is not None:
                bl = blinds[label][1]
                jobs.append((powers_gamma_g, 0, bl.ptr, bl.n)); slots.append((label, "shifted_comm"))
    outs = ctx.msm_batch_dev(jobs)
    res = {label: {"comm": None, "shifted_comm": None} for label in polys}
    for (label, which), pt in zip(slots, outs):
        res[label][which] = pt if res[label][which] is None else ctx.g1_add(res[label][which], pt)
    return res



def batch_open(ctx: Context, powers_g, queries, opening_challenge: int):
    """KZG10 witnesses for several (polynomials, point) queries: per query p = sum_i xi^i p_i and
    w = commit((p - p(z)) / (X - z)) (kzg10/mod.rs:212-293 applied to the combination, as marlin_pc::batch_open does per
    query point); the witness MSMs of all queries run as one pipelined batch.  queries: [(polys, point), ...]."""
    F = HostField(ctx)
    jobs, keep = [], []
    for polys, point in queries:
        ks, k = [], 1
        for _ in polys:
            ks.append(k)
            k = F.mul(k, opening_challenge)
        comb = linear_combination(ctx, polys, ks)
        q = ctx.alloc(max(comb.n - 1, 1) * 32)
        ctx.poly_divide_by_linear_dev(comb.ptr, comb.n, HostField.m(point), q.ptr)
        keep += [comb, q]
        jobs.append((powers_g, 0, q.ptr, comb.n - 1))
    return ctx.msm_batch_dev(jobs)



class MarlinProof:
    """marlin::Proof (data_structures.rs:99-110): commitments per round, the evaluations sorted by label, three empty prover
    messages, one KZG10 proof (w, random_v) per query point (beta, gamma)."""

    def __init__(self, commitments, evaluations, pc_proof, challenges):
        self.commitments, self.evaluations, self.pc_proof, self.challenges = commitments, evaluations, pc_proof, challenges

    def serialize(self, ctx: Context) -> bytes:
        """CanonicalSerialize (derive order; Vec = u64 length + items, Option = one byte + item, points compressed)."""
        u64 = lambda v: v.to_bytes(8, "little")
        out = u64(len(self.commitments))
        for rnd in self.commitments:
            out += u64(len(rnd))
            for c in rnd:
                out += ctx.g1_serialize(c.comm) + (b"\x01" + ctx.g1_serialize(c.shifted) if c.shifted is not None else b"\x00")
        out += u64(len(self.evaluations)) + b"".join(_fr_bytes(e) for e in self.evaluations)
        out += u64(3) + b"\x00" * 3
        out += u64(len(self.pc_proof))
        for w, rv in self.pc_proof:
            out += ctx.g1_serialize(w) + (b"\x01" + _fr_bytes(rv) if rv is not None else b"\x00")
        return out + b"\x00"


def _sample_outside(dom: Domain, fs) -> int:
    """EvaluationDomain::sample_element_outside_domain (poly/src/domain/mod.rs:37-51)."""
    t = HostField.i(fs.next_fr())
    while dom.evaluate_vanishing_polynomial(t) == 0:
        t = HostField.i(fs.next_fr())
    return t


def _host_poly_eval(coeffs, x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


def _host_divide_by_linear(coeffs, z: int):
    """Quotient of p / (X - z) for a short host polynomial (the blinding polynomials: three coefficients)."""
    q, acc = [0] * max(len(coeffs) - 1, 0), 0
    for i in range(len(coeffs) - 1, 0, -1):
        acc = (coeffs[i] + acc * z) % R_MOD
        q[i - 1] = acc
    return q


def _draw_round_randomness(keys: IndexKeys, labels, zk_rng) -> dict:
    """The blinding polynomials of one PC::commit call, in the reference's order: label -> (blind, shifted blind)."""
    rands = {}
    for l in labels:
        hb = keys.hiding.get(l)
        blind = [HostField.i(v) for v in zk_rng.fill_fr(hb + 2)] if hb is not None else []
        sblind = None
        if l in keys.bounds:
            sblind = [HostField.i(v) for v in zk_rng.fill_fr(hb + 2)] if hb is not None else []
        rands[l] = (blind, sblind)
    return rands


def _commit_round(keys: IndexKeys, labels, polys: dict, zk_rng, rands: dict = None, raw: bool = False):
    """PC::commit(ck, oracles, Some(zk_rng)) (marlin_pc/mod.rs:172-243): the blinding polynomials are drawn oracle by oracle --
    three coefficients for a hiding bound of 1, a second set for the shifted commitment of a degree-bounded oracle -- and all
    MSMs of the round run as one pipelined batch.  Returns ({label: PcCommitment}, {label: (blind, shifted_blind)}).
    rands: randomness drawn beforehand (a SPDZ prover commits its share lane and its MAC lane under the same draws);
    raw: return the MSM results ({label: {"comm", "shifted_comm"}}) instead of PcCommitment objects (a collaborative prover
    reveals the sums over parties first)."""
    ctx, srs = keys.index.ctx, keys.srs
    if rands is None:
        rands = _draw_round_randomness(keys, labels, zk_rng)
    blinds, keep = {}, []
    for l in labels:
        blind, sblind = rands[l]
        if blind:
            db = ctx.upload(cv.fr_to_mont(blind))
            ds = ctx.upload(cv.fr_to_mont(sblind)) if sblind else None
            keep += [db, ds]
            blinds[l] = (DevPoly(db, len(blind)), DevPoly(ds, len(sblind)) if sblind else None)
    bounds = {l: (keys.bounds.get(l), keys.hiding.get(l)) for l in labels}
    res = commit_marlin_pc(ctx, srs.powers_g, srs.powers_gamma_g, {l: polys[l] for l in labels}, bounds, blinds)
    if raw:
        return res, rands
    return {l: PcCommitment(res[l]["comm"], res[l]["shifted_comm"]) for l in labels}, rands


def _linear_combinations(index: "Index", public_input, ch, ev):
    """AHPForR1CS::construct_linear_combinations (ahp/mod.rs:112-290): label -> [(coefficient, polynomial label or None for the
    constant term)], sorted by label; ev(label) -> the evaluation of the single-polynomial combinations / denominators."""
    F = index.dom_h.F
    H, K = index.dom_h, index.dom_k
    alpha, eta_a, eta_b, eta_c, beta, gamma = (ch[k] for k in ("alpha", "eta_a", "eta_b", "eta_c", "beta", "gamma"))
    neg = lambda v: (-v) % R_MOD
    x = [1] + list(public_input)
    nx = len(x)
    v_H_alpha, v_H_beta = H.evaluate_vanishing_polynomial(alpha), H.evaluate_vanishing_polynomial(beta)
    v_X_beta = (pow(beta, nx, R_MOD) - 1) % R_MOD
    # eval_unnormalized_bivariate_lagrange_poly (ahp/mod.rs:337-350): (v_H(alpha) - v_H(beta)) / (alpha - beta)
    r_alpha_at_beta = (v_H_alpha - v_H_beta) * pow((alpha - beta) % R_MOD, -1, R_MOD) % R_MOD if alpha != beta else \
        H.size * pow(alpha, H.size - 1, R_MOD) % R_MOD
    # x(beta) through the Lagrange coefficients of the input domain (radix2/mod.rs:116-165)
    wx = pow(int(HostField.i(_FR_TWO_ADIC_ROOT_MONT)), 1 << (_FR_TWO_ADICITY - _log2(nx)), R_MOD)
    if v_X_beta == 0:
        x_beta = next(xv for k, xv in enumerate(x) if pow(wx, k, R_MOD) == beta)
    else:
        x_beta, g = 0, 1
        for xv in x:
            x_beta = (x_beta + xv * (v_X_beta * pow(nx, -1, R_MOD) % R_MOD * g % R_MOD) % R_MOD * pow((beta - g) % R_MOD, -1, R_MOD)) % R_MOD
            g = g * wx % R_MOD
    z_b_beta, t_beta, g_1_beta = ev("z_b"), ev("t"), ev("g_1")
    lcs = {"z_b": [(1, "z_b")], "g_1": [(1, "g_1")], "t": [(1, "t")], "g_2": [(1, "g_2")]}
    lcs["outer_sumcheck"] = [(1, "mask_poly"),
                             (_mulmod((r_alpha_at_beta, (eta_a + eta_c * z_b_beta) % R_MOD)), "z_a"),
                             (_mulmod((r_alpha_at_beta, eta_b, z_b_beta)), None),
                             (neg(_mulmod((t_beta, v_X_beta))), "w"),
                             (neg(_mulmod((t_beta, x_beta))), None),
                             (neg(v_H_beta), "h_1"),
                             (neg(_mulmod((beta, g_1_beta))), None)]
    ba = beta * alpha % R_MOD
    for m in "abc":
        lcs[m + "_denom"] = [(ba, None), (neg(alpha), m + "_row"), (neg(beta), m + "_col"), (1, m + "_row_col")]
    da, db, dc, g_2_gamma = ev("a_denom"), ev("b_denom"), ev("c_denom"), ev("g_2")
    v_K_gamma = K.evaluate_vanishing_polynomial(gamma)
    vv = v_H_alpha * v_H_beta % R_MOD
    b_expr = _mulmod((da, db, dc, (gamma * g_2_gamma + t_beta * pow(K.size, -1, R_MOD)) % R_MOD))
    lcs["inner_sumcheck"] = [(_mulmod((eta_a, db, dc, vv)), "a_val"), (_mulmod((eta_b, da, dc, vv)), "b_val"),
                             (_mulmod((eta_c, db, da, vv)), "c_val"), (neg(b_expr), None), (neg(v_K_gamma), "h_2")]
    return dict(sorted(lcs.items()))


def _mulmod(xs) -> int:
    acc = 1
    for v in xs:
        acc = acc * v % R_MOD
    return acc


def prove(keys: IndexKeys, assignment_dev: DevBuf, zk_rng, mask_on_device: bool = False) -> MarlinProof:
    """Marlin::prove (lib.rs:152-319) on the device: the three AHP rounds, MarlinKZG10 commitments with hiding, the Fiat-Shamir
    transcript, the evaluations and open_combinations (marlin/mod.rs:213-306: one KZG10 proof per query point over the
    challenge-weighted combination of the linear combinations, degree-bounded oracles through their shifted witnesses).
    zk_rng: an api.Rng (the prover's randomness, drawn in the reference's order).
    mask_on_device: the 3 |H| coefficients of the mask polynomial (DensePolynomial::rand, prover.rs:371-376) are private prover
    randomness; drawn one by one from a host generator they cost more than the whole proof at 2^20 (0.4 s of scalar ChaCha).
    With this flag they are sampled on the device (zk_fr_random_dev, ChaCha20 under a 32-byte key taken from zk_rng): the
    same distribution, not the same stream -- proofs then differ from a reference run with the same seed, and verify alike."""
    from zk_mpc_amd.api import Rng
    index, srs = keys.index, keys.srs
    ctx = index.ctx
    m, ival = HostField.m, HostField.i
    st = prover_init(index, assignment_dev)
    ni = index.num_instance
    public_input = cv.fr_from_mont(ctx.download(assignment_dev.ptr + 32, (ni - 1, 4))) if ni > 1 else []
    fs = Rng.fiat_shamir(PROTOCOL_NAME + keys.ivk_bytes() + b"".join(_fr_bytes(v) for v in public_input))
    polys = dict(index.polynomials())
    rands = {l: ([], None) for l in INDEX_LABELS}
    comms = dict(keys.index_comms)
    ch = {}
    # ---- round 1: F::rand x 3 and the mask polynomial's coefficients, then the hiding commitments
    md = mask_poly_degree(index)
    if mask_on_device:
        rnd = ctx.alloc((3 + md + 1) * 32)
        head = ctx.upload(zk_rng.fill_fr(3))
        ctx.memcpy_d2d(rnd.ptr, head.ptr, 96)
        ctx.fr_random_dev(rnd.ptr + 96, md + 1, zk_rng.fill_bytes(32))
        ctx.sync()
    else:
        rnd = ctx.upload(zk_rng.fill_fr(3 + md + 1))
    r1 = prover_first_round(st, rnd)
    polys.update(r1)
    c1, q1 = _commit_round(keys, ROUND_LABELS[0], r1, zk_rng)
    comms.update(c1); rands.update(q1)
    fs.absorb(b"".join(c1[l].to_bytes() for l in ROUND_LABELS[0]))
    ch["alpha"] = _sample_outside(index.dom_h, fs)
    ch["eta_a"], ch["eta_b"], ch["eta_c"] = ival(fs.next_fr()), ival(fs.next_fr()), ival(fs.next_fr())
    # ---- round 2
    r2 = prover_second_round(st, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"])
    polys.update(r2)
    c2, q2 = _commit_round(keys, ROUND_LABELS[1], r2, zk_rng)
    comms.update(c2); rands.update(q2)
    fs.absorb(b"".join(c2[l].to_bytes() for l in ROUND_LABELS[1]))
    ch["beta"] = _sample_outside(index.dom_h, fs)
    # ---- round 3
    r3 = prover_third_round(st, ch["beta"])
    polys.update(r3)
    c3, q3 = _commit_round(keys, ROUND_LABELS[2], r3, zk_rng)
    comms.update(c3); rands.update(q3)
    fs.absorb(b"".join(c3[l].to_bytes() for l in ROUND_LABELS[2]))
    ch["gamma"] = ival(fs.next_fr())
    # ---- evaluations (lib.rs:279-294)
    ev_poly = lambda l, pt: ival(ctx.poly_evaluate_dev(polys[l].ptr, polys[l].n, m(pt)))
    single = {"z_b": ev_poly("z_b", ch["beta"]), "g_1": ev_poly("g_1", ch["beta"]), "t": ev_poly("t", ch["beta"]),
              "g_2": ev_poly("g_2", ch["gamma"])}
    ba = ch["beta"] * ch["alpha"] % R_MOD
    for mm in "abc":
        single[mm + "_denom"] = (ba - ch["alpha"] * ev_poly(mm + "_row", ch["gamma"]) - ch["beta"] * ev_poly(mm + "_col", ch["gamma"])
                                 + ev_poly(mm + "_row_col", ch["gamma"])) % R_MOD
    lcs = _linear_combinations(index, public_input, ch, lambda l: single[l])
    evaluations = [single[l] for l in EVAL_LABELS]
    fs.absorb(b"".join(_fr_bytes(e) for e in evaluations))
    xi = fs.next_u128() % R_MOD                                       # u128::rand(&mut fs_rng).into()  (lib.rs:300)
    ch["xi"] = xi
    # ---- open_combinations: per query point one polynomial sum_j xi^j LC_j, its blinding polynomial, and for a degree-bounded
    # oracle the witness of the oracle itself over the shifted powers with the next power of xi (marlin_pc/mod.rs:245-340)
    point = {"beta": ch["beta"], "gamma": ch["gamma"]}
    jobs, plan, keep = [], [], []
    for pl in ("beta", "gamma"):
        z = point[pl]
        terms, r_comb, shifted, sr = {}, [], [], []                   # terms: polynomial label -> accumulated coefficient
        j = 0
        for label in QUERY_SET[pl]:
            lc = [(c, l) for c, l in lcs[label] if l is not None]
            cj = pow(xi, j, R_MOD); j += 1
            for c, l in lc:
                terms[l] = (terms.get(l, 0) + c * cj) % R_MOD
                blind = rands[l][0]
                r_comb = [((r_comb[i] if i < len(r_comb) else 0) + (blind[i] if i < len(blind) else 0) * c % R_MOD * cj) % R_MOD
                          for i in range(max(len(r_comb), len(blind)))]
            if len(lcs[label]) == 1 and lc[0][1] in keys.bounds:
                src = lc[0][1]
                cj1 = pow(xi, j, R_MOD); j += 1
                shifted.append((src, cj1))
                sb = rands[src][1] or []
                sr = [((sr[i] if i < len(sr) else 0) + (sb[i] if i < len(sb) else 0) * cj1) % R_MOD for i in range(max(len(sr), len(sb)))]
        labels = list(terms)
        comb = linear_combination(ctx, [polys[l] for l in labels], [terms[l] for l in labels])
        q = ctx.alloc(max(comb.n - 1, 1) * 32)
        ctx.poly_divide_by_linear_dev(comb.ptr, comb.n, m(z), q.ptr)
        keep += [comb, q]
        these = [(srs.powers_g, 0, q.ptr, comb.n - 1)]
        hiding = any(r_comb)
        rv = None
        if hiding:
            rw = _host_divide_by_linear(r_comb, z)
            d = ctx.upload(cv.fr_to_mont(rw)); keep.append(d)
            these.append((srs.powers_gamma_g, 0, d.ptr, len(rw)))
            rv = _host_poly_eval(r_comb, z)
        srw = []
        for src, cj1 in shifted:
            p = polys[src]
            wq = ctx.alloc(max(p.n - 1, 1) * 32)
            ctx.poly_divide_by_linear_dev(p.ptr, p.n, m(z), wq.ptr)
            ctx.fr_vec_scale_dev(wq.ptr, m(cj1), wq.ptr, p.n - 1)
            keep.append(wq)
            these.append((srs.powers_g, srs.max_degree - keys.bounds[src], wq.ptr, p.n - 1))
            sb = rands[src][1] or []
            if sb:
                w1 = _host_divide_by_linear(sb, z)
                srw = [((srw[i] if i < len(srw) else 0) + w1[i] * cj1) % R_MOD for i in range(len(w1))]
        if srw:
            d = ctx.upload(cv.fr_to_mont(srw)); keep.append(d)
            these.append((srs.powers_gamma_g, 0, d.ptr, len(srw)))
        if shifted and rv is not None:
            rv = (rv + _host_poly_eval(sr, z)) % R_MOD
        plan.append((len(these), rv))
        jobs += these
    outs = ctx.msm_batch_dev(jobs)
    pc_proof, k = [], 0
    for cnt, rv in plan:
        w = outs[k]
        for o in outs[k + 1:k + cnt]:
            w = ctx.g1_add(w, o)
        k += cnt
        pc_proof.append((w, rv))
    ctx.sync()
    return MarlinProof([[comms[l] for l in rnd] for rnd in ROUND_LABELS], evaluations, pc_proof, ch)


