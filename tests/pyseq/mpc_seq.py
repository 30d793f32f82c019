"""The collaborative provers as PYTHON sequences of C-ABI calls over the parties' transports: test infrastructure.

create_proof over MpcField / MpcGroup (src/groth16.rs:68-183) and MpcMarlin::prove (src/marlin.rs:56) spelled out open by open --
additive (Party) and SPDZ (SpdzParty) -- on top of the share-level primitives of zk_mpc_amd.mpc (Beaver batch_mul, group scale,
opens, MAC checks).  The PRODUCT is one C-ABI call per proof (zk_groth16_prove_shared[_spdz], zk_marlin_prove_shared[_spdz]:
Party.*_native in zk_mpc_amd.mpc); these sequences are the second implementation the tests hold them to: same opened values, same
traffic, same bytes.  Everything of zk_mpc_amd.mpc is re-exported, so `import pyseq.mpc_seq as mpc` serves both.
"""
from __future__ import annotations

import os

import numpy as np

from zk_mpc_amd import mpc as _P
from zk_mpc_amd.mpc import *             # noqa: F401,F403
from zk_mpc_amd.mpc import MacCheckError  # noqa: F401


class Party(_P.Party):
    def create_proof_shared(self, pk, r1cs, z_share, r_share, s_share, triple=None, fused=True) -> bytes:
        """create_proof over additive shares (src/groth16.rs:68-183 with E = MpcPairingEngine).
        z_share: this party's share of the full assignment (device vector); r_share, s_share: (4,) uint64.
        Returns the revealed 192-byte proof (identical on every party).
        fused (default): the nine small opens of the three `scale` calls and of Proof::reveal travel in two collectives (every
        opened value is the same as in the reference's order; A is the opened s + x of the second scale).  fused=False keeps
        the reference's call order, one collective per open."""
        be = self.be
        D = be.domain_size(r1cs)
        P = self._pk_points(pk)
        # public point x shared scalar is local host arithmetic (0.4 ms per G1, 1.2 ms per G2 scalar multiplication): the three
        # that do not depend on the MSMs run on host threads under the device work (the library calls release the GIL)
        early = self._early(be.g1_mul, P["delta_g1"], r_share), self._early(be.g1_mul, P["delta_g1"], s_share), \
            self._early(be.g2_mul, P["delta_g2"], s_share)
        a, b, c = be.vec("wm_a", D), be.vec("wm_b", D), be.vec("wm_c", D)
        be.witness_map_pre(r1cs, z_share, a, b, c)                 # local: linear in the shares
        be.msms_begin(pk, r1cs, z_share)                           # the four MSMs over z run under the open and the second half below
        self.beaver_batch_mul(a, b, a, D, triple)                  # the one shared x shared vector product (:285)
        be.witness_map_post(r1cs, a, c)                            # h shares in `a`
        g1, g2 = be.msms(pk, r1cs, z_share, a)                     # party-local MSMs (multi_scale_pub_group)
        h_acc, l_acc, a_acc, b1_acc = g1[0], g1[1], g1[2], g1[3]
        pub1 = (lambda x: x) if self.leader else (lambda x: be.g1_zero())   # shift(): leader only
        pub2 = (lambda x: x) if self.leader else (lambda x: be.g2_zero())
        r_g1 = early[0].result()                                   # delta_g1 * r: public point * shared scalar, local
        if fused:
            y = be.fr_one() if self.leader else np.zeros(4, dtype=np.uint64)
            g_a = be.g1_add(be.g1_add(be.g1_add(r_g1, pub1(P["a0"])), a_acc), pub1(P["alpha_g1"]))
            g1_b = be.g1_add(be.g1_add(be.g1_add(early[1].result(), pub1(P["b0_g1"])), b1_acc), pub1(P["beta_g1"]))
            g2_b = be.g2_add(be.g2_add(be.g2_add(early[2].result(), pub2(P["b0_g2"])), g2), pub2(P["beta_g2"]))
            (oy_s, oy_r), (sx_rd, sx_a, sx_b), (B,) = self._open_many([be.fr_add(s_share, y), be.fr_add(r_share, y)], [r_g1, g_a, g1_b], [g2_b])
            parts = [self._early(self._scale_finish, sx, oy, y) for sx, oy in ((sx_rd, oy_s), (sx_a, oy_s), (sx_b, oy_r))]
            g_c = be.g1_add(parts[1].result(), parts[2].result())
            g_c = be.g1_add(g_c, be.g1_neg(parts[0].result()))
            g_c = be.g1_add(be.g1_add(g_c, l_acc), h_acc)
            C = self.reveal_g1(g_c)
            return be.g1_serialize(sx_a) + be.g2_serialize(B) + be.g1_serialize(C)
        r_s_delta = self.scale_g1(r_g1, s_share, lazy=True)        # :115
        g_a = be.g1_add(be.g1_add(be.g1_add(r_g1, pub1(P["a0"])), a_acc), pub1(P["alpha_g1"]))   # calculate_coeff
        s_g_a = self.scale_g1(g_a, s_share, lazy=True)             # :140
        s_g1 = early[1].result()
        g1_b = be.g1_add(be.g1_add(be.g1_add(s_g1, pub1(P["b0_g1"])), b1_acc), pub1(P["beta_g1"]))
        s_g2 = early[2].result()
        g2_b = be.g2_add(be.g2_add(be.g2_add(s_g2, pub2(P["b0_g2"])), g2), pub2(P["beta_g2"]))
        r_g1_b = self.scale_g1(g1_b, r_share, lazy=True)           # :161
        g_c = be.g1_add(s_g_a.result(), r_g1_b.result())           # :169-174
        g_c = be.g1_add(g_c, be.g1_neg(r_s_delta.result()))
        g_c = be.g1_add(g_c, l_acc)
        g_c = be.g1_add(g_c, h_acc)
        A, B, C = self.reveal_g1(g_a), self.reveal_g2(g2_b), self.reveal_g1(g_c)     # Proof::reveal
        return be.g1_serialize(A) + be.g2_serialize(B) + be.g1_serialize(C)


    # ---- collaborative Marlin (AHP rounds over additive shares) ----
    SHARED_POLYS = ("w", "z_a", "z_b", "mask_poly", "g_1", "h_1")     # the witness-dependent oracles; t, g_2, h_2 are public

    def marlin_prove_shared(self, index, powers_g, z_share, randomness_share, challenge_fn, triple_fn=None) -> dict:
        """Marlin::prove (arkworks/marlin/src/lib.rs:152-319) with F = MpcField over additive shares, on this party's GPU.

        Every step of the AHP rounds is linear in the witness except z_A * z_B in round 2 (`DensePolynomial::mul` on
        MpcField = FieldShare::batch_mul: one Beaver product of two vectors over the 4|H| multiplication domain); round 3
        involves public values only.  Commitments / evaluations / opening witnesses of witness-dependent polynomials are
        computed on the shares and revealed (`first_comms.publicize()`, `evaluations.publicize()` in the reference).

        z_share: this party's share of the padded assignment (device vector, instance part shared like the rest);
        randomness_share: this party's share of the prover's randomness (3 + 3|H| elements);
        challenge_fn(round, revealed_commitments) -> dict of challenges (the Fiat-Shamir transcript stays with the caller);
        triple_fn(n) -> (tx, ty, tz) device vectors, or None for the reference's DummyFieldTripleSource."""
        from . import marlin_seq as DM
        be = self.be
        ctx = be.ctx
        m = DM.HostField.m

        def reveal_some(comms):
            return {l: (self.reveal_g1(c) if l in self.SHARED_POLYS else c) for l, c in comms.items()}

        def open_is_zero(v, n):
            tmp = be.vec("marlin_open", n)
            be.open_vec(v, tmp, n)
            return be.is_zero_vec(tmp, n)

        def batch_mul(x, y, out, n):
            self.beaver_batch_mul(x, y, out, n, triple_fn(n) if triple_fn else None)

        st = DM.prover_init(index, z_share, shared=True)
        r1 = DM.prover_first_round(st, randomness_share)
        comms = reveal_some(DM.commit(ctx, powers_g, r1))
        ch = dict(challenge_fn(1, comms))
        r2 = DM.prover_second_round(st, ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"], batch_mul=batch_mul, open_is_zero=open_is_zero)
        comms.update(reveal_some(DM.commit(ctx, powers_g, r2)))
        ch.update(challenge_fn(2, comms))
        r3 = DM.prover_third_round(st, ch["beta"])
        comms.update(DM.commit(ctx, powers_g, r3))
        ch.update(challenge_fn(3, comms))
        polys = {**r1, **r2, **r3}
        ev = lambda l, pt: ctx.poly_evaluate_dev(polys[l].ptr, polys[l].n, m(pt))
        evals = {"g_1": self._open_fr(ev("g_1", ch["beta"])), "z_b": self._open_fr(ev("z_b", ch["beta"])),
                 "t": ev("t", ch["beta"]), "g_2": ev("g_2", ch["gamma"])}
        mine = lambda l: polys[l] if (l in self.SHARED_POLYS or self.leader) else None
        at_beta = [mine(l) for l in ("g_1", "z_b", "t", "mask_poly", "z_a", "w", "h_1")]
        ixp = index.polynomials()
        at_gamma = [polys["g_2"], polys["h_2"]] + [ixp[l] for l in sorted(ixp)]
        w_beta, w_gamma = DM.batch_open(ctx, powers_g, [(at_beta, ch["beta"]), (at_gamma, ch["gamma"])], ch["xi"])
        return {"commitments": comms, "evaluations": evals, "w_beta": self.reveal_g1(w_beta), "w_gamma": w_gamma, "challenges": ch}


    # ---- collaborative Marlin as a PROOF: transcript, hiding commitments, open_combinations over shares ----
    def marlin_prove_full(self, keys, z_share, zk_rng, triple_fn=None, mask_on_device=False):
        """MpcMarlin::prove (src/marlin.rs:56 -> arkworks/marlin/src/lib.rs:152-319 with F = MpcField) over additive shares:
        the complete proof, as `marlin.prove` emits it for one prover.  z_share: this party's share of the padded assignment
        (DevBuf; instance on the leader); zk_rng: this party's OWN generator -- every draw is a share (MpcField::rand), the
        effective randomness is the sum over parties.  The revealed proof is identical on every party and equal to the local
        proof on the summed inputs and summed randomness."""
        return _marlin_prove_full(self, keys, [z_share], zk_rng, triple_fn, spdz=False, mask_on_device=mask_on_device)


class SpdzParty(_P.SpdzParty, Party):
    def marlin_prove_shared_spdz(self, index, powers_g, z_share, randomness_share, challenge_fn, triple_fn=None) -> dict:
        """Marlin::prove over SPDZ shares (the `malicious` feature, BASELINE config 5 shape): the AHP rounds run on the
        share lane and on the MAC lane; the two lanes meet in the one Beaver multiplication of round 2 (SPDZ batch_mul:
        both opens MAC-checked) and in the MAC-checked opens of commitments, evaluations and the opening witness.
        z_share, randomness_share: (sh, mac) pairs; triple_fn(n) -> ((tx_sh, tx_mac), (ty..), (tz..)) or None (dummy)."""
        from . import marlin_seq as DM
        be = self.be
        ctx = be.ctx
        m = DM.HostField.m
        lanes = (0, 1)
        shared = Party.SHARED_POLYS

        def commit_open(polys2):
            c = [DM.commit(ctx, powers_g, polys2[lane]) for lane in lanes]
            return {l: (self.spdz_open_g1((c[0][l], c[1][l])) if l in shared else c[0][l]) for l in c[0]}

        st = [DM.prover_init(index, z_share[lane], shared=True) for lane in lanes]
        r1 = [DM.prover_first_round(st[lane], randomness_share[lane]) for lane in lanes]
        comms = commit_open(r1)
        ch = dict(challenge_fn(1, comms))
        steps = [DM.second_round_steps(st[lane], ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"]) for lane in lanes]
        req = [next(g) for g in steps]                                   # both lanes stop at z_A * z_B
        n_mul = req[0][4]
        self.spdz_beaver_batch_mul((req[0][1], req[1][1]), (req[0][2], req[1][2]), (req[0][3], req[1][3]), n_mul,
                                   triple_fn(n_mul) if triple_fn else None)
        req = [g.send(None) for g in steps]                              # ... and at the zero test of the sum over H
        opened = be.vec("marlin_open", req[0][2])
        self.spdz_open_vec((req[0][1], req[1][1]), opened, req[0][2])
        ok = be.is_zero_vec(opened, req[0][2])
        r2 = []
        for g in steps:
            try:
                g.send(ok)
                raise RuntimeError("second round did not finish")
            except StopIteration as done:
                r2.append(done.value)
        comms.update(commit_open(r2))
        ch.update(challenge_fn(2, comms))
        r3 = DM.prover_third_round(st[0], ch["beta"])                    # public values only: one lane suffices
        comms.update(DM.commit(ctx, powers_g, r3))
        ch.update(challenge_fn(3, comms))
        polys = [{**r1[lane], **r2[lane], **r3} for lane in lanes]
        ev = lambda lane, l, pt: ctx.poly_evaluate_dev(polys[lane][l].ptr, polys[lane][l].n, m(pt))
        evals = {l: self.spdz_open_fr((ev(0, l, ch["beta"]), ev(1, l, ch["beta"]))) for l in ("g_1", "z_b")}
        evals["t"], evals["g_2"] = ev(0, "t", ch["beta"]), ev(0, "g_2", ch["gamma"])
        # public polynomials enter a shared combination through shift(): on the leader, in both lanes (mac_share = 1 there)
        mine = lambda lane, l: polys[lane][l] if (l in shared or self.leader) else None
        ixp = index.polynomials()
        at_gamma = [polys[0]["g_2"], polys[0]["h_2"]] + [ixp[l] for l in sorted(ixp)]
        w = []
        for lane in lanes:
            at_beta = [mine(lane, l) for l in ("g_1", "z_b", "t", "mask_poly", "z_a", "w", "h_1")]
            w.append(DM.batch_open(ctx, powers_g, [(at_beta, ch["beta"])] + ([(at_gamma, ch["gamma"])] if lane == 0 else []), ch["xi"]))
        return {"commitments": comms, "evaluations": evals, "w_beta": self.spdz_open_g1((w[0][0], w[1][0])), "w_gamma": w[0][1],
                "challenges": ch}

    def create_proof_shared_spdz(self, pk, r1cs, z_share, r_share, s_share, triple=None, fused=True) -> bytes:
        """create_proof with E = MpcPairingEngine<_, SpdzPairingShare> (the `malicious` feature).
        z_share = (sh, mac) device vectors; r_share, s_share = (sh, mac) scalars.
        fused: as create_proof_shared -- the opens of the three scale calls and of B in one collective, their MAC checks in a
        second one, then C and its check."""
        be = self.be
        D = be.domain_size(r1cs)
        P = self._pk_points(pk)
        early = {(k, lane): self._early(fn, P[pt], sc[lane]) for lane in (0, 1)       # see create_proof_shared
                 for k, fn, pt, sc in (("r_g1", be.g1_mul, "delta_g1", r_share), ("s_g1", be.g1_mul, "delta_g1", s_share),
                                       ("s_g2", be.g2_mul, "delta_g2", s_share))}
        lanes = []
        for lane in (0, 1):
            a, b, c = be.vec("wm_a%d" % lane, D), be.vec("wm_b%d" % lane, D), be.vec("wm_c%d" % lane, D)
            be.witness_map_pre(r1cs, z_share[lane], a, b, c)
            lanes.append((a, b, c))
        A = (lanes[0][0], lanes[1][0])
        B = (lanes[0][1], lanes[1][1])
        be.msms_begin(pk, r1cs, z_share[0])                        # the share lane's four MSMs over z run under the opens below
        self.spdz_beaver_batch_mul(A, B, A, D, triple)
        msm = []
        for lane in (0, 1):
            be.witness_map_post(r1cs, lanes[lane][0], lanes[lane][2])
            msm.append(be.msms(pk, r1cs, z_share[lane], lanes[lane][0]))       # 2 x 5 MSMs (spdz.rs:482-488)
        pair = lambda f: tuple(f(lane) for lane in (0, 1))
        pub1 = (lambda x: x) if self.leader else (lambda x: be.g1_zero())      # shift: leader's sh; mac += mac_share * G
        pub2 = (lambda x: x) if self.leader else (lambda x: be.g2_zero())
        add1 = lambda u, v: (be.g1_add(u[0], v[0]), be.g1_add(u[1], v[1]))
        neg1 = lambda u: (be.g1_neg(u[0]), be.g1_neg(u[1]))
        h_acc, l_acc, a_acc, b1_acc = [pair(lambda lane, k=k: msm[lane][0][k]) for k in range(4)]
        b2_acc = pair(lambda lane: msm[lane][1])
        r_g1 = pair(lambda lane: early[("r_g1", lane)].result())
        if fused:
            y = be.fr_one() if self.leader else np.zeros(4, dtype=np.uint64)
            g_a = pair(lambda lane: be.g1_add(be.g1_add(be.g1_add(r_g1[lane], pub1(P["a0"])), a_acc[lane]), pub1(P["alpha_g1"])))
            g1_b = pair(lambda lane: be.g1_add(be.g1_add(be.g1_add(early[("s_g1", lane)].result(), pub1(P["b0_g1"])), b1_acc[lane]), pub1(P["beta_g1"])))
            g2_b = pair(lambda lane: be.g2_add(be.g2_add(be.g2_add(early[("s_g2", lane)].result(), pub2(P["b0_g2"])), b2_acc[lane]), pub2(P["beta_g2"])))
            sy = (be.fr_add(s_share[0], y), be.fr_add(s_share[1], y))              # o + y, y = from_add_shared(leader ? 1 : 0)
            ry = (be.fr_add(r_share[0], y), be.fr_add(r_share[1], y))
            (oy_s, oy_r), (sx_rd, sx_a, sx_b), (B,) = self._spdz_open_many([sy, ry], [r_g1, g_a, g1_b], [g2_b])
            parts = [self._early(self._scale_finish, sx, oy, y) for sx, oy in ((sx_rd, oy_s), (sx_a, oy_s), (sx_b, oy_r))]
            both = lambda f: (lambda t: (t, t))(f.result())                        # scale: both lanes hold the same value (key 1)
            g_c = add1(add1(add1(add1(both(parts[1]), both(parts[2])), neg1(both(parts[0]))), l_acc), h_acc)
            Cp = self.spdz_open_g1(g_c)
            return be.g1_serialize(sx_a) + be.g2_serialize(B) + be.g1_serialize(Cp)
        r_s_delta = self.spdz_scale_g1(r_g1, s_share, lazy=True)
        g_a = pair(lambda lane: be.g1_add(be.g1_add(be.g1_add(r_g1[lane], pub1(P["a0"])), a_acc[lane]), pub1(P["alpha_g1"])))
        s_g_a = self.spdz_scale_g1(g_a, s_share, lazy=True)
        s_g1 = pair(lambda lane: early[("s_g1", lane)].result())
        g1_b = pair(lambda lane: be.g1_add(be.g1_add(be.g1_add(s_g1[lane], pub1(P["b0_g1"])), b1_acc[lane]), pub1(P["beta_g1"])))
        s_g2 = pair(lambda lane: early[("s_g2", lane)].result())
        g2_b = pair(lambda lane: be.g2_add(be.g2_add(be.g2_add(s_g2[lane], pub2(P["b0_g2"])), b2_acc[lane]), pub2(P["beta_g2"])))
        r_g1_b = self.spdz_scale_g1(g1_b, r_share, lazy=True)
        g_c = add1(add1(add1(add1(s_g_a.result(), r_g1_b.result()), neg1(r_s_delta.result())), l_acc), h_acc)
        Ap, Bp, Cp = self.spdz_open_g1(g_a), self.spdz_open_g2(g2_b), self.spdz_open_g1(g_c)   # SpdzGroupShare::reveal
        return be.g1_serialize(Ap) + be.g2_serialize(Bp) + be.g1_serialize(Cp)


    def marlin_prove_full_spdz(self, keys, z_share, zk_rng, triple_fn=None, mask_on_device=False):
        """The same over SPDZ shares (the `malicious` feature; BASELINE config 5's prover): z_share = (share, MAC) DevBufs, every
        open MAC-checked.  The MAC lane of this party's fresh randomness is the share itself (key alpha = 1 on the leader:
        sum of MAC shares = sum of shares), as the reference's from_add_shared does."""
        return _marlin_prove_full(self, keys, list(z_share), zk_rng, triple_fn, spdz=True, mask_on_device=mask_on_device)



def _marlin_prove_full(party, keys, z_lanes, zk_rng, triple_fn, spdz: bool, mask_on_device: bool = False):
    from zk_mpc_amd import convert as cv
    from . import marlin_seq as DM
    from zk_mpc_amd.api import Rng
    be = party.be
    ctx = be.ctx
    index, srs = keys.index, keys.srs
    m, ival = DM.HostField.m, DM.HostField.i
    R_MOD = DM.R_MOD
    lanes = range(len(z_lanes))
    shared = Party.SHARED_POLYS
    leader = party.leader
    import os as _os
    import time as _time
    _laps, _t = [], [_time.perf_counter()]

    def lap(name):                                      # ZK_MPC_TIMING=1: host wall-clock laps on stderr
        if _os.environ.get("ZK_MPC_TIMING"):
            ctx.sync()
            now = _time.perf_counter()
            _laps.append("%s %.1f" % (name, (now - _t[0]) * 1e3))
            _t[0] = now

    def open_g1(pts):                                   # pts: one projective array per lane
        return party.spdz_open_g1(tuple(pts)) if spdz else party.reveal_g1(pts[0])

    def open_fr(vals):                                  # vals: one (4,) Montgomery array per lane
        return ival(party.spdz_open_fr(tuple(vals)) if spdz else party._open_fr(vals[0]))

    # the public input is the instance part of the assignment: shared as from_public (the leader holds it), opened for the transcript
    ni = index.num_instance
    pub = []
    if ni > 1:
        tmp = be.vec("marlin_pub", ni)
        if spdz:
            party.spdz_open_vec((z_lanes[0].ptr, z_lanes[1].ptr), tmp, ni)
        else:
            be.open_vec(z_lanes[0].ptr, tmp, ni)
        pub = cv.fr_from_mont(ctx.download(tmp, (ni, 4)))[1:]
    fs = Rng.fiat_shamir(DM.PROTOCOL_NAME + keys.ivk_bytes() + b"".join(DM._fr_bytes(v) for v in pub))
    st = [DM.prover_init(index, z, shared=True) for z in z_lanes]
    polys = [dict(index.polynomials()) for _ in lanes]
    rands = {l: ([], None) for l in DM.INDEX_LABELS}
    comms = dict(keys.index_comms)
    ch = {}

    def commit_round(labels, round_polys):
        """Shares of the commitments on every lane under the same draws, then the reveal of the witness-dependent ones
        (`comms.publicize()`, lib.rs:180,205,228); public oracles commit alike on every party."""
        rr = DM._draw_round_randomness(keys, labels, zk_rng)
        # public oracles (t, g_2, h_2) are the same on every lane: committed once
        res = [DM._commit_round(keys, labels if lane == 0 else [l for l in labels if l in shared], round_polys[lane], zk_rng, rands=rr,
                                raw=True)[0] for lane in lanes]
        out = {}
        for l in labels:
            if l in shared:
                c = open_g1([res[lane][l]["comm"] for lane in lanes])
                sc = open_g1([res[lane][l]["shifted_comm"] for lane in lanes]) if res[0][l]["shifted_comm"] is not None else None
            else:
                c, sc = res[0][l]["comm"], res[0][l]["shifted_comm"]
            out[l] = DM.PcCommitment(c, sc)
        rands.update(rr)
        comms.update(out)
        fs.absorb(b"".join(out[l].to_bytes() for l in labels))

    # ---- round 1: every draw is this party's share of the prover's randomness
    md = DM.mask_poly_degree(index)
    if mask_on_device:
        # this party's share of the mask polynomial sampled on the device under a key from its rng (marlin.py::prove:
        # 3 |H| draws from a host ChaCha generator take 0.19 s at 2^20, more than the rest of the proof)
        rnd = ctx.alloc((3 + md + 1) * 32)
        head = ctx.upload(zk_rng.fill_fr(3))
        ctx.memcpy_d2d(rnd.ptr, head.ptr, 96)
        ctx.fr_random_dev(rnd.ptr + 96, md + 1, zk_rng.fill_bytes(32))
        ctx.sync()
    else:
        rnd = ctx.upload(zk_rng.fill_fr(3 + md + 1))
    lap("init+rng")
    r1 = [DM.prover_first_round(st[lane], rnd) for lane in lanes]
    for lane in lanes:
        polys[lane].update(r1[lane])
    lap("round1")
    commit_round(DM.ROUND_LABELS[0], r1)
    lap("commit1")
    ch["alpha"] = DM._sample_outside(index.dom_h, fs)
    ch["eta_a"], ch["eta_b"], ch["eta_c"] = ival(fs.next_fr()), ival(fs.next_fr()), ival(fs.next_fr())
    # ---- round 2: the lanes advance in lock-step around ONE Beaver product and one opened zero test
    steps = [DM.second_round_steps(st[lane], ch["alpha"], ch["eta_a"], ch["eta_b"], ch["eta_c"]) for lane in lanes]
    req = [next(g) for g in steps]
    n_mul = req[0][4]
    if spdz:
        party.spdz_beaver_batch_mul((req[0][1], req[1][1]), (req[0][2], req[1][2]), (req[0][3], req[1][3]), n_mul,
                                    triple_fn(n_mul) if triple_fn else None)
    else:
        party.beaver_batch_mul(req[0][1], req[0][2], req[0][3], n_mul, triple_fn(n_mul) if triple_fn else None)
    req = [g.send(None) for g in steps]
    opened = be.vec("marlin_open", req[0][2])
    if spdz:
        party.spdz_open_vec((req[0][1], req[1][1]), opened, req[0][2])
    else:
        be.open_vec(req[0][1], opened, req[0][2])
    ok = be.is_zero_vec(opened, req[0][2])
    r2 = []
    for g in steps:
        try:
            g.send(ok)
            raise RuntimeError("second round did not finish")
        except StopIteration as done:
            r2.append(done.value)
    for lane in lanes:
        polys[lane].update(r2[lane])
    lap("round2")
    commit_round(DM.ROUND_LABELS[1], r2)
    lap("commit2")
    ch["beta"] = DM._sample_outside(index.dom_h, fs)
    # ---- round 3: public values only
    r3 = DM.prover_third_round(st[0], ch["beta"])
    for lane in lanes:
        polys[lane].update(r3)
    lap("round3")
    commit_round(DM.ROUND_LABELS[2], [r3 for _ in lanes])
    lap("commit3")
    ch["gamma"] = ival(fs.next_fr())
    # ---- evaluations: shared oracles are evaluated on the shares and opened (`evaluations.publicize()`)
    ev = lambda lane, l, pt: ctx.poly_evaluate_dev(polys[lane][l].ptr, polys[lane][l].n, m(pt))
    single = {l: open_fr([ev(lane, l, ch["beta"]) for lane in lanes]) for l in ("z_b", "g_1")}
    single["t"], single["g_2"] = ival(ev(0, "t", ch["beta"])), ival(ev(0, "g_2", ch["gamma"]))
    ba = ch["beta"] * ch["alpha"] % R_MOD
    for mm in "abc":
        single[mm + "_denom"] = (ba - ch["alpha"] * ival(ev(0, mm + "_row", ch["gamma"])) - ch["beta"] * ival(ev(0, mm + "_col", ch["gamma"]))
                                 + ival(ev(0, mm + "_row_col", ch["gamma"]))) % R_MOD
    lcs = DM._linear_combinations(index, pub, ch, lambda l: single[l])
    evaluations = [single[l] for l in DM.EVAL_LABELS]
    fs.absorb(b"".join(DM._fr_bytes(e) for e in evaluations))
    xi = fs.next_u128() % R_MOD
    ch["xi"] = xi
    lap("evals")
    # ---- open_combinations on the shares: the witness of a share combination is a share of the witness; public polynomials
    # enter a shared combination through shift(), i.e. on the leader (in both lanes: mac_share = 1 there)
    point = {"beta": ch["beta"], "gamma": ch["gamma"]}
    pc_proof, keep = [], []
    for pl in ("beta", "gamma"):
        z = point[pl]
        terms, shifted, j = {}, [], 0
        r_comb, sr = [], []
        for label in DM.QUERY_SET[pl]:
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
        any_shared = any(l in shared for l in labels)
        hiding = any(len(rands[l][0]) > 0 for l in labels)
        wit = []
        for lane in (lanes if any_shared else [0]):
            mine = [polys[lane][l] if (l in shared or leader or not any_shared) else None for l in labels]
            comb = DM.linear_combination(ctx, mine, [terms[l] for l in labels])
            q = ctx.alloc(max(comb.n - 1, 1) * 32)
            ctx.poly_divide_by_linear_dev(comb.ptr, comb.n, m(z), q.ptr)
            keep += [comb, q]
            jobs = [(srs.powers_g, 0, q.ptr, comb.n - 1)]
            if hiding:
                rw = DM._host_divide_by_linear(r_comb, z)
                d = ctx.upload(cv.fr_to_mont(rw)); keep.append(d)
                jobs.append((srs.powers_gamma_g, 0, d.ptr, len(rw)))
            srw = []
            for src, cj1 in shifted:
                pp = polys[lane][src]
                if src in shared or leader or not any_shared:
                    wq = ctx.alloc(max(pp.n - 1, 1) * 32)
                    ctx.poly_divide_by_linear_dev(pp.ptr, pp.n, m(z), wq.ptr)
                    ctx.fr_vec_scale_dev(wq.ptr, m(cj1), wq.ptr, pp.n - 1)
                    keep.append(wq)
                    jobs.append((srs.powers_g, srs.max_degree - keys.bounds[src], wq.ptr, pp.n - 1))
                sb = rands[src][1] or []
                if sb:
                    w1 = DM._host_divide_by_linear(sb, z)
                    srw = [((srw[i] if i < len(srw) else 0) + w1[i] * cj1) % R_MOD for i in range(len(w1))]
            if srw:
                d = ctx.upload(cv.fr_to_mont(srw)); keep.append(d)
                jobs.append((srs.powers_gamma_g, 0, d.ptr, len(srw)))
            outs = ctx.msm_batch_dev(jobs)
            w = outs[0]
            for o in outs[1:]:
                w = ctx.g1_add(w, o)
            wit.append(w)
        rv = None
        if hiding:
            rv_share = DM._host_poly_eval(r_comb, z)
            if shifted:
                rv_share = (rv_share + DM._host_poly_eval(sr, z)) % R_MOD
            rv = open_fr([m(rv_share) for _ in lanes])      # the MAC lane of fresh local randomness is the share itself
        w = open_g1(wit) if any_shared else wit[0]
        pc_proof.append((w, rv))
    ctx.sync()
    lap("open")
    if _laps:
        import sys as _sys
        print("marlin_prove_full ms: " + " ".join(_laps), file=_sys.stderr)
    return DM.MarlinProof([[comms[l] for l in rnd_labels] for rnd_labels in DM.ROUND_LABELS], evaluations, pc_proof, ch)

