"""GPU: N-party collaborative Groth16 on one device (parties = threads with their own zk_ctx, transport =
LocalNet, the analogue of the reference's LocalTestNet).  Parity: every party's revealed proof equals the
oracle's known-trapdoor prediction for the summed inputs, bit for bit (SURVEY 8c MPC parity)."""
import threading

import numpy as np
import pytest

import zkref as O
import zkref_c as OC
import zk_mpc_amd as Z
import zk_mpc_amd.convert as cv
import pyseq.mpc_seq as mpc
from helpers import td_mont, mont1
from oracle_backend import additive_shares

pytestmark = pytest.mark.gpu


def run_parties(n_parties, fn, devices=None):
    """devices: the HIP device of each party's context (default: all on device 0).  The party threads never select a device
    themselves: a context owns its device (ctx.hpp: ZkDeviceGuard at every entry point)."""
    nets = mpc.LocalNet.create(n_parties)
    out, err = [None] * n_parties, []

    def work(p):
        ctx = Z.Context(devices[p] if devices else 0, p, n_parties)
        try:
            out[p] = fn(p, ctx, nets[p])
        except Exception as e:  # pragma: no cover
            import traceback
            err.append("party %d: %s\n%s" % (p, e, traceback.format_exc()))
            try:
                nets[p].sh.barrier.abort()
            except Exception:
                pass
        finally:
            ctx.close()

    ts = [threading.Thread(target=work, args=(p,)) for p in range(n_parties)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, "\n".join(err)
    return out


@pytest.mark.parametrize("n_parties", [2, 3, 8])
def test_beaver_batch_mul_real_and_dummy_triples(n_parties):
    rng = O.Prng(900 + n_parties)
    n = 1000
    xs, ys = [rng.fr() for _ in range(n)], [rng.fr() for _ in range(n)]
    ta, tb = [rng.fr() for _ in range(n)], [rng.fr() for _ in range(n)]
    tc = [a * b % O.R_MOD for a, b in zip(ta, tb)]
    sh = {k: additive_shares(v, n_parties, rng) for k, v in dict(x=xs, y=ys, ta=ta, tb=tb, tc=tc).items()}

    def fn(p, ctx, net):
        party = mpc.Party(ctx, net=net)
        up = {k: ctx.upload(cv.fr_to_mont(v[p])) for k, v in sh.items()}
        out = ctx.alloc(n * 32)
        party.beaver_batch_mul(up["x"].ptr, up["y"].ptr, out.ptr, n, triple=(up["ta"].ptr, up["tb"].ptr, up["tc"].ptr))
        real = cv.fr_from_mont(ctx.download(out, (n, 4)))
        party.beaver_batch_mul(up["x"].ptr, up["y"].ptr, out.ptr, n)           # DummyFieldTripleSource
        dummy = cv.fr_from_mont(ctx.download(out, (n, 4)))
        return real, dummy

    res = run_parties(n_parties, fn)
    want = [a * b % O.R_MOD for a, b in zip(xs, ys)]
    assert [sum(c) % O.R_MOD for c in zip(*[r[0] for r in res])] == want
    assert [sum(c) % O.R_MOD for c in zip(*[r[1] for r in res])] == want
    assert all(v == 0 for v in res[1][1])      # dummy triples: everything lands on the leader (wire/field.rs:49-63)


@pytest.mark.parametrize("n_parties,n", [(3, 6), (3, 1000), (2, (1 << 12) - 2), (8, 301)])
def test_collaborative_prove(n_parties, n):
    """BASELINE config 3 shape (3-party additive, mul-chain) at test size."""
    rng = O.Prng(1000 + n)
    w0, w1 = rng.fr(), rng.fr()
    r1cs, z = O.mul_chain_r1cs(n, w0, w1)
    td = O.Trapdoor(rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr())
    r, s = rng.fr(), rng.fr()
    zs = additive_shares(z, n_parties, rng, public_prefix=2)
    rsh, ssh = O.additive_share(r, n_parties, rng), O.additive_share(s, n_parties, rng)
    tdm = td_mont(td)

    def fn(p, ctx, net):
        party = mpc.Party(ctx, net=net)
        dr = ctx.r1cs_mul_chain(n)
        pk = ctx.groth16_setup(dr, *[tdm[i] for i in range(7)])
        dz = ctx.upload(cv.fr_to_mont(zs[p]))
        proof = party.create_proof_shared(pk, dr, dz.ptr, mont1(rsh[p]), mont1(ssh[p]))
        sent = party.bytes_sent
        sent2 = party.bytes_sent
        assert party.create_proof_shared(pk, dr, dz.ptr, mont1(rsh[p]), mont1(ssh[p]), fused=False) == proof
        unfused = party.bytes_sent - sent2
        # the one-call C entry (zk_groth16_prove_shared; the transport reached through callbacks): same bytes, same traffic
        sent3 = party.bytes_sent
        assert party.create_proof_shared_native(pk, dr, dz.ptr, mont1(rsh[p]), mont1(ssh[p])) == proof
        assert party.bytes_sent - sent3 == sent
        return proof, (sent, unfused)

    res = run_parties(n_parties, fn)
    cr = OC.R1cs(2, n + 1, *OC.mul_chain_csr(n))
    zm = cv.fr_to_mont(z)
    want = OC.groth16_predict(cr, tdm, zm, OC.witness_map(cr, zm), mont1(r), mont1(s))
    assert all(pr == want for pr, _ in res)
    D = 1 << cr.domain_log
    # Appendix C traffic in the reference's order of opens; fused, s + y is opened once and A is the opened point of the second scale
    assert all(b == (2 * D * 32 + 2 * 32 + 3 * 144 + 288 + 144, 2 * D * 32 + 3 * (144 + 32) + 144 + 288 + 144) for _, b in res)


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs in one process")
@pytest.mark.parametrize("spdz", [False, True])
def test_parties_on_different_gpus_inside_one_process(spdz):
    """The reference's LocalTestNet shape (mpc-net/src/multi.rs:419-443): several parties as tasks of ONE process -- here each
    on its own GPU, Context(p, p, n), with threads that never call hipSetDevice.  Every entry point has to run on its
    context's device (scratch, streams, events, kernel launches); the revealed proof equals the local proof on the sums."""
    ndev = _device_count()
    n_parties, n = min(ndev, 3), 1000
    rng = O.Prng(4242 + int(spdz))
    w0, w1 = rng.fr(), rng.fr()
    r1cs, z = O.mul_chain_r1cs(n, w0, w1)
    td = O.Trapdoor(rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr())
    r, s = rng.fr(), rng.fr()
    zs = additive_shares(z, n_parties, rng, public_prefix=2)
    zm = additive_shares(z, n_parties, rng, public_prefix=2)          # independent sharing of alpha*z, alpha = 1
    rsh, ssh = O.additive_share(r, n_parties, rng), O.additive_share(s, n_parties, rng)
    rm, sm = O.additive_share(r, n_parties, rng), O.additive_share(s, n_parties, rng)
    tdm = td_mont(td)

    def fn(p, ctx, net):
        assert ctx.device == p
        dr = ctx.r1cs_mul_chain(n)
        pk = ctx.groth16_setup(dr, *[tdm[i] for i in range(7)])
        if spdz:
            party = mpc.SpdzParty(ctx, net=net)
            dz, dm = ctx.upload(cv.fr_to_mont(zs[p])), ctx.upload(cv.fr_to_mont(zm[p]))
            return party.create_proof_shared_spdz(pk, dr, (dz.ptr, dm.ptr), (mont1(rsh[p]), mont1(rm[p])), (mont1(ssh[p]), mont1(sm[p])))
        party = mpc.Party(ctx, net=net)
        dz = ctx.upload(cv.fr_to_mont(zs[p]))
        return party.create_proof_shared(pk, dr, dz.ptr, mont1(rsh[p]), mont1(ssh[p]))

    res = run_parties(n_parties, fn, devices=list(range(n_parties)))
    cr = OC.R1cs(2, n + 1, *OC.mul_chain_csr(n))
    zm = cv.fr_to_mont(z)
    want = OC.groth16_predict(cr, tdm, zm, OC.witness_map(cr, zm), mont1(r), mont1(s))
    assert all(pr == want for pr in res)


@pytest.mark.parametrize("n_parties,n", [(2, 100), (3, 5000), (8, 77)])
def test_collaborative_prove_spdz(n_parties, n):
    """The reference's `malicious` backend (SpdzFieldShare / SpdzGroupShare, key alpha = 1): two-lane shares, MAC-checked
    opens; the revealed proof equals the local proof, and a corrupted MAC share is detected."""
    rng = O.Prng(2000 + n)
    w0, w1 = rng.fr(), rng.fr()
    r1cs, z = O.mul_chain_r1cs(n, w0, w1)
    td = O.Trapdoor(rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr())
    r, s = rng.fr(), rng.fr()
    zs = additive_shares(z, n_parties, rng, public_prefix=2)
    zm = additive_shares(z, n_parties, rng, public_prefix=2)          # independent sharing of alpha*z, alpha = 1
    rsh, ssh = O.additive_share(r, n_parties, rng), O.additive_share(s, n_parties, rng)
    rm, sm = O.additive_share(r, n_parties, rng), O.additive_share(s, n_parties, rng)
    tdm = td_mont(td)

    def fn(p, ctx, net):
        party = mpc.SpdzParty(ctx, net=net)
        dr = ctx.r1cs_mul_chain(n)
        pk = ctx.groth16_setup(dr, *[tdm[i] for i in range(7)])
        dzs, dzm = ctx.upload(cv.fr_to_mont(zs[p])), ctx.upload(cv.fr_to_mont(zm[p]))
        rr, ss = (mont1(rsh[p]), mont1(rm[p])), (mont1(ssh[p]), mont1(sm[p]))
        sent0 = party.bytes_sent
        good = party.create_proof_shared_spdz(pk, dr, (dzs.ptr, dzm.ptr), rr, ss)
        sent = party.bytes_sent - sent0
        # the one-call C entry (zk_groth16_prove_shared_spdz): same bytes, same traffic; a bad MAC comes back as ZK_ERR_MAC
        sent0 = party.bytes_sent
        assert party.create_proof_shared_spdz_native(pk, dr, (dzs.ptr, dzm.ptr), rr, ss) == good
        assert party.bytes_sent - sent0 == sent
        bad = cv.fr_to_mont(zm[p])
        if p == n_parties - 1:
            bad[5, 0] ^= np.uint64(1)
        dbad = ctx.upload(bad)
        try:
            party.create_proof_shared_spdz(pk, dr, (dzs.ptr, dbad.ptr), rr, ss)
            caught = False
        except mpc.MacCheckError:
            caught = True
        try:
            party.create_proof_shared_spdz_native(pk, dr, (dzs.ptr, dbad.ptr), rr, ss)
            caught = False
        except mpc.MacCheckError:
            pass
        return good, caught

    res = run_parties(n_parties, fn)
    cr = OC.R1cs(2, n + 1, *OC.mul_chain_csr(n))
    zmnt = cv.fr_to_mont(z)
    want = OC.groth16_predict(cr, tdm, zmnt, OC.witness_map(cr, zmnt), mont1(r), mont1(s))
    assert all(g == want for g, _ in res)
    assert all(c for _, c in res)


def test_distnet_nccl_single_rank():
    """The bench's N>1 plumbing (torch.distributed/nccl transport over device tensors, share generation on the
    device) exercised with a 1-rank process group: the 1-party "collaborative" proof must equal the local one."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ctx = Z.Context(0, 0, 1)
    try:
        rng = O.Prng(77)
        n = 1000
        w0, w1 = rng.fr(), rng.fr()
        r, s = rng.fr(), rng.fr()
        td = O.Trapdoor(rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr())
        tdm = td_mont(td)
        dr = ctx.r1cs_mul_chain(n)
        pk = ctx.groth16_setup(dr, *[tdm[i] for i in range(7)])
        dz = ctx.mul_chain_assignment_dev(n, mont1(w0), mont1(w1))
        party = mpc.Party(ctx, dist)
        zshare = party.share_assignment_dev(dz, dr, seed=5)
        assert np.array_equal(ctx.download(zshare, (n + 3, 4)), ctx.download(dz, (n + 3, 4)))   # 1 party: share == value
        rs = party.share_scalars([r, s], seed=9)
        assert cv.fr_from_mont(np.stack(rs)) == [r, s]
        proof = party.create_proof_shared(pk, dr, zshare, rs[0], rs[1])
        assert proof == ctx.create_proof_dev(pk, dr, dz.ptr, mont1(r), mont1(s))
        # vector open through the nccl all-gather path
        a = ctx.upload(cv.fr_to_mont([rng.fr() for _ in range(64)]))
        out = party.be.vec("t_out", 64)
        party.be.open_vec(a.ptr, out, 64)
        assert np.array_equal(ctx.download(out, (64, 4)), ctx.download(a, (64, 4)))
        # ... and through the all-to-all + all-gather pattern used for three or more parties (RCCL all_to_all_single)
        party.net.open_pattern = "a2a"
        try:
            for m in (64, 61, 1):
                ctx.dev_zero(out, 64 * 32)
                party.be.open_vec(a.ptr, out, m)
                assert np.array_equal(ctx.download(out, (m, 4)), ctx.download(a, (m, 4)))
            proof2 = party.create_proof_shared(pk, dr, zshare, rs[0], rs[1])
            assert proof2 == proof
            # the same proof with the opens on the library's own RCCL communicator (ZK_TRANSPORT=native)
            os.environ["ZK_TRANSPORT"] = "native"
            try:
                party_n = mpc.Party(ctx, dist)
                assert party_n.be.native_open
                ctx.comm_set_open_pattern(2)
                assert party_n.create_proof_shared(pk, dr, zshare, rs[0], rs[1]) == proof
            finally:
                del os.environ["ZK_TRANSPORT"]
                ctx.comm_destroy()
            # king_share through RCCL's scatter (one rank: the only share is the value itself)
            ks = party.king_share_vec(a.ptr, 64, key32=bytes(range(32)))
            assert np.array_equal(ctx.download(ks, (64, 4)), ctx.download(a, (64, 4)))
        finally:
            party.net.open_pattern = None
    finally:
        ctx.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("n_parties,n", [(2, 6), (3, 13), (3, 500)])
def test_collaborative_marlin(n_parties, n):
    """Collaborative Marlin over additive shares (BASELINE config 5 shape at test size): revealed commitments, evaluations
    and opening witnesses equal the single-prover run on the summed witness and summed randomness; the opened evaluations
    satisfy the verifier's sum-check equations."""
    import marlin_ref as M
    import pyseq.marlin_seq as DM
    rng = O.Prng(1100 + n)
    r1cs, z = O.mul_chain_r1cs(n, rng.fr(), rng.fr())
    sq, zz = M.pad_and_square(r1cs, z)
    H = M.next_pow2(sq.num_constraints)
    n_rnd = 3 + 3 * H
    rnd = [rng.fr() for _ in range(n_rnd)]
    zs = additive_shares(zz, n_parties, rng)
    rs = additive_shares(rnd, n_parties, rng)
    chal = {1: {k: rng.fr() for k in ("alpha", "eta_a", "eta_b", "eta_c")}, 2: {"beta": rng.fr()}, 3: {"gamma": rng.fr(), "xi": rng.fr()}}
    beta_srs = rng.fr()
    a, b, c = DM.Csr.from_rows(sq.a), DM.Csr.from_rows(sq.b), DM.Csr.from_rows(sq.c)

    def setup(ctx):
        index = DM.Index(ctx, sq.num_instance, sq.num_witness, a, b, c)
        deg = 3 * max(index.dom_h.size, index.dom_k.size) + 2
        pw = ctx.alloc(deg * 32)
        ctx.fr_powers_dev(mont1(beta_srs), mont1(1), deg, pw.ptr)
        return index, ctx.fixed_base(pw.ptr, deg, 1, mont1(1))

    seen = [[] for _ in range(n_parties)]

    def fn(p, ctx, net):
        party = mpc.Party(ctx, net=net)
        index, powers_g = setup(ctx)
        def challenge_fn(rnd_no, comms):
            seen[p].append({l: cv.g1_projective_to_affine(c) for l, c in comms.items()})
            return chal[rnd_no]
        out = party.marlin_prove_shared(index, powers_g, ctx.upload(cv.fr_to_mont(zs[p])), cv.fr_to_mont(rs[p]), challenge_fn)
        return ({l: cv.g1_projective_to_affine(c) for l, c in out["commitments"].items()},
                {l: cv.fr_from_mont(np.asarray(v).reshape(1, 4))[0] for l, v in out["evaluations"].items()},
                cv.g1_projective_to_affine(out["w_beta"]), cv.g1_projective_to_affine(out["w_gamma"]))

    res = run_parties(n_parties, fn)
    assert all(r == res[0] for r in res)
    # single prover on the summed inputs
    ctx = Z.Context(0)
    try:
        index, powers_g = setup(ctx)
        st = DM.prover_init(index, ctx.upload(cv.fr_to_mont(zz)))
        polys = dict(DM.prover_first_round(st, cv.fr_to_mont(rnd)))
        c1 = chal[1]
        polys.update(DM.prover_second_round(st, c1["alpha"], c1["eta_a"], c1["eta_b"], c1["eta_c"]))
        polys.update(DM.prover_third_round(st, chal[2]["beta"]))
        comms = {l: cv.g1_projective_to_affine(v) for l, v in DM.commit(ctx, powers_g, polys).items()}
        beta, gamma, xi = chal[2]["beta"], chal[3]["gamma"], chal[3]["xi"]
        ev = lambda l, pt: cv.fr_from_mont(ctx.poly_evaluate_dev(polys[l].ptr, polys[l].n, mont1(pt)).reshape(1, 4))[0]
        evals = {"g_1": ev("g_1", beta), "z_b": ev("z_b", beta), "t": ev("t", beta), "g_2": ev("g_2", gamma)}
        ixp = index.polynomials()
        wb, wg = DM.batch_open(ctx, powers_g, [([polys[l] for l in ("g_1", "z_b", "t", "mask_poly", "z_a", "w", "h_1")], beta),
                                               ([polys["g_2"], polys["h_2"]] + [ixp[l] for l in sorted(ixp)], gamma)], xi)
        assert res[0] == (comms, evals, cv.g1_projective_to_affine(wb), cv.g1_projective_to_affine(wg))
        # the sum-check equations on the device's polynomials (pins the single-prover side as in test_gpu_marlin.py)
        allp = {**ixp, **polys}
        evf = lambda l, pt: cv.fr_from_mont(ctx.poly_evaluate_dev(allp[l].ptr, allp[l].n, mont1(pt)).reshape(1, 4))[0]
        info = M.IndexInfo(index.num_constraints, index.num_non_zero, index.num_instance)
        assert M.sumcheck_equations(info, zz[1:r1cs.num_instance], evf, c1["alpha"], c1["eta_a"], c1["eta_b"], c1["eta_c"], beta, gamma) == (0, 0)
    finally:
        ctx.close()
    # every party saw the same revealed commitments before each challenge
    assert all(s == seen[0] for s in seen)


@pytest.mark.parametrize("n_parties,n", [(2, 6), (3, 40), (8, 21)])
def test_collaborative_marlin_spdz(n_parties, n):
    """Marlin over SPDZ shares (two lanes, MAC-checked opens): the same revealed outputs as the additive run above on the
    same inputs, and a corrupted MAC share of the witness is detected."""
    import marlin_ref as M
    import pyseq.marlin_seq as DM
    rng = O.Prng(1200 + n)
    r1cs, z = O.mul_chain_r1cs(n, rng.fr(), rng.fr())
    sq, zz = M.pad_and_square(r1cs, z)
    H = M.next_pow2(sq.num_constraints)
    rnd = [rng.fr() for _ in range(3 + 3 * H)]
    zs, zm = additive_shares(zz, n_parties, rng), additive_shares(zz, n_parties, rng)       # key alpha = 1: mac shares sum to z
    rs, rm = additive_shares(rnd, n_parties, rng), additive_shares(rnd, n_parties, rng)
    chal = {1: {k: rng.fr() for k in ("alpha", "eta_a", "eta_b", "eta_c")}, 2: {"beta": rng.fr()}, 3: {"gamma": rng.fr(), "xi": rng.fr()}}
    beta_srs = rng.fr()
    a, b, c = DM.Csr.from_rows(sq.a), DM.Csr.from_rows(sq.b), DM.Csr.from_rows(sq.c)

    def setup(ctx):
        index = DM.Index(ctx, sq.num_instance, sq.num_witness, a, b, c)
        deg = 3 * max(index.dom_h.size, index.dom_k.size) + 2
        pw = ctx.alloc(deg * 32)
        ctx.fr_powers_dev(mont1(beta_srs), mont1(1), deg, pw.ptr)
        return index, ctx.fixed_base(pw.ptr, deg, 1, mont1(1))

    def flat(out):
        return ({l: cv.g1_projective_to_affine(c) for l, c in out["commitments"].items()},
                {l: cv.fr_from_mont(np.asarray(v).reshape(1, 4))[0] for l, v in out["evaluations"].items()},
                cv.g1_projective_to_affine(out["w_beta"]), cv.g1_projective_to_affine(out["w_gamma"]))

    def fn(p, ctx, net):
        party = mpc.SpdzParty(ctx, net=net)
        index, powers_g = setup(ctx)
        up = lambda v: ctx.upload(cv.fr_to_mont(v))
        spdz = flat(party.marlin_prove_shared_spdz(index, powers_g, (up(zs[p]), up(zm[p])), (cv.fr_to_mont(rs[p]), cv.fr_to_mont(rm[p])),
                                                   lambda k, comms: chal[k]))
        additive = flat(party.marlin_prove_shared(index, powers_g, up(zs[p]), cv.fr_to_mont(rs[p]), lambda k, comms: chal[k]))
        bad = cv.fr_to_mont(zm[p])
        if p == n_parties - 1:
            bad[3, 0] ^= np.uint64(1)
        try:
            party.marlin_prove_shared_spdz(index, powers_g, (up(zs[p]), ctx.upload(bad)), (cv.fr_to_mont(rs[p]), cv.fr_to_mont(rm[p])),
                                           lambda k, comms: chal[k])
            caught = False
        except mpc.MacCheckError:
            caught = True
        return spdz, additive, caught

    res = run_parties(n_parties, fn)
    assert all(r[0] == res[0][0] for r in res)
    assert res[0][0] == res[0][1]           # SPDZ run == additive run (which test_collaborative_marlin ties to the single prover)
    assert all(r[2] for r in res)


@pytest.mark.parametrize("n_parties,key32", [(2, None), (3, None), (3, bytes(range(32))), (8, None)])
def test_king_share_vector(n_parties, key32):
    """Reveal::king_share on a device vector: the leader's N - 1 uniform shares (ChaCha20 keyed from the OS CSPRNG by default,
    from a caller's key in tests) plus the residual, scattered; the shares sum to the secret, no single share equals it,
    two parties' masks differ, and two runs with the default key give different shares."""
    rng = O.Prng(1300 + n_parties)
    n = 777
    secret = [rng.fr() for _ in range(n)]

    def fn(p, ctx, net):
        party = mpc.Party(ctx, net=net)
        src = ctx.upload(cv.fr_to_mont(secret)) if p == 0 else None
        mine = party.king_share_vec(src.ptr if src else None, n, key32=key32)
        again = party.king_share_vec(src.ptr if src else None, n, key32=key32)
        return cv.fr_from_mont(ctx.download(mine, (n, 4))), cv.fr_from_mont(ctx.download(again, (n, 4)))

    res = run_parties(n_parties, fn)
    first = [r[0] for r in res]
    assert [sum(c) % O.R_MOD for c in zip(*first)] == secret
    assert [sum(c) % O.R_MOD for c in zip(*[r[1] for r in res])] == secret
    assert all(r != secret for r in first)
    assert first[0] != first[1]
    if key32 is None:
        assert all(r[0] != r[1] for r in res)       # a fresh key per call
    else:
        assert all(r[0] == r[1] for r in res)       # reproducible under a caller's key


def test_fr_random_uniform_and_keyed(ctx):
    """zk_fr_random_dev: element i = ChaCha20(key, block i, stream) reduced mod r.  Checked against the host block function
    (itself pinned to RFC 8439 in tests/test_fsrng.py), every value < r, distinct streams differ, and the values are spread
    over the whole of [0, r): the top three bits of value * 8 / r are uniform (chi-square over 8 bins, 2^16 samples), the
    largest value is within 0.1 % of r -- a mask confined to a sparse subset of F_r would fail both."""
    import ctypes as C
    n = 1 << 16
    key = bytes((7 * i + 1) & 0xff for i in range(32))
    out = ctx.alloc(n * 32)
    ctx.fr_random_dev(out.ptr, n, key, stream_id=5)
    raw = ctx.download(out, (n, 4))
    vals = cv.fr_from_mont(raw)
    assert all(0 <= v < O.R_MOD for v in vals)
    # element i against the host's block function: the device forms (lo + 2^256 hi) * 2^261 mod r (its internal Montgomery
    # form, 9 x 29 bits) and stores those words in the reference's layout (R = 2^256), i.e. the value wide * 2^5 mod r
    lib = ctx.lib
    for i in (0, 1, 12345, n - 1):
        w = (C.c_uint32 * 4)(i, 0, 5, 0)
        blk = (C.c_uint8 * 64)()
        assert lib.zk_chacha_block(key, w, 20, blk) == 0
        wide = int.from_bytes(bytes(blk), "little")
        assert vals[i] == wide * 32 % O.R_MOD
    bins = [0] * 8
    for v in vals:
        bins[v * 8 // O.R_MOD] += 1
    chi2 = sum((b - n / 8) ** 2 / (n / 8) for b in bins)
    assert chi2 < 30.0, bins                       # 7 degrees of freedom: P(chi2 > 30) ~ 1e-4
    assert max(vals) > O.R_MOD - O.R_MOD // 1000 and min(vals) < O.R_MOD // 1000
    assert len(set(vals)) == n
    out2 = ctx.alloc(n * 32)
    ctx.fr_random_dev(out2.ptr, n, key, stream_id=6)
    assert cv.fr_from_mont(ctx.download(out2, (16, 4))) != vals[:16]
    ctx.fr_random_dev(out2.ptr, n, key, stream_id=5)
    assert np.array_equal(ctx.download(out2, (n, 4)), raw)
    ctx.fr_random_dev(out2.ptr, 64)                                   # OS-keyed: just runs and differs
    assert cv.fr_from_mont(ctx.download(out2, (16, 4))) != vals[:16]


def test_native_rccl_open_single_rank(ctx):
    """The library's own transport (comm.hip: RCCL dlopen'ed, communicator from a unique id) with one rank: both open
    patterns return the vector itself, in place and out of place, for lengths with and without padding."""
    import os
    uid = ctx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(uid, 0, 1)
    try:
        rng = O.Prng(1400)
        vals = [rng.fr() for _ in range(1000)]
        v = ctx.upload(cv.fr_to_mont(vals))
        out = ctx.alloc(1000 * 32)
        for mode in (1, 2, 0):                              # all-gather, all-to-all of slices, by party count
            ctx.comm_set_open_pattern(mode)
            for m in (1000, 999, 1):
                ctx.dev_zero(out.ptr, 1000 * 32)
                ctx.open_sum_fr_dev(v.ptr, m, out.ptr)
                assert cv.fr_from_mont(ctx.download(out, (m, 4))) == vals[:m]
        w = ctx.upload(cv.fr_to_mont(vals))
        ctx.open_sum_fr_dev(w.ptr, 1000, w.ptr)           # in place
        assert cv.fr_from_mont(ctx.download(w, (1000, 4))) == vals
        with pytest.raises(Exception, match="already has a communicator"):
            ctx.comm_init(uid, 0, 1)
    finally:
        ctx.comm_destroy()


def _rccl_worker(rank, world, port, q):
    """One RCCL rank per GPU (needs >= `world` devices): both transports of the share-vector open (DistNet.open_sum over
    torch.distributed and the library's own communicator, zk_open_sum_fr_dev) in both patterns on lengths that do not divide
    by the party count, king_share through the scatter, and a collaborative Groth16 proof against the local proof on the
    summed inputs."""
    import os
    import sys
    import traceback
    try:
        import torch
        import torch.distributed as dist
        ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for pth in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            if pth not in sys.path:
                sys.path.insert(0, pth)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        ctx = Z.Context(rank, rank, world)
        rng = O.Prng(9000)                                   # same stream on every rank
        party = mpc.Party(ctx, dist)
        for n in (1, 7, 64, 1001, world * 500 + 3):
            vals = [[rng.fr() for _ in range(n)] for _ in range(world)]
            want = [sum(c) % O.R_MOD for c in zip(*vals)]
            v = ctx.upload(cv.fr_to_mont(vals[rank]))
            out = party.be.vec("o%d" % n, n)
            for pattern in (None, "allgather", "a2a"):
                party.net.open_pattern = pattern
                ctx.dev_zero(out, n * 32)
                party.be.open_vec(v.ptr, out, n)
                assert cv.fr_from_mont(ctx.download(out, (n, 4))) == want, ("torch", pattern, n)
            party.net.open_pattern = None
        box = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctx.comm_init(box[0], rank, world)
        for n in (1, 13, 999, world * 256 + 5):
            vals = [[rng.fr() for _ in range(n)] for _ in range(world)]
            want = [sum(c) % O.R_MOD for c in zip(*vals)]
            v = ctx.upload(cv.fr_to_mont(vals[rank]))
            o = ctx.alloc(n * 32)
            for pattern in (0, 1, 2):
                ctx.comm_set_open_pattern(pattern)
                ctx.dev_zero(o.ptr, n * 32)
                ctx.open_sum_fr_dev(v.ptr, n, o.ptr)
                ctx.sync()
                assert cv.fr_from_mont(ctx.download(o, (n, 4))) == want, ("native", pattern, n)
        ctx.comm_destroy()
        # collaborative proof == local proof on the summed inputs
        n = 500
        td = [mont1(rng.fr()) for _ in range(7)]
        w0, w1, r, s = rng.fr(), rng.fr(), rng.fr(), rng.fr()
        dr = ctx.r1cs_mul_chain(n)
        pk = ctx.groth16_setup(dr, *td)
        dz = ctx.mul_chain_assignment_dev(n, mont1(w0), mont1(w1))
        zshare = party.share_assignment_dev(dz, dr, seed=11)
        rs = party.share_scalars([r, s], seed=12)
        proof = party.create_proof_shared(pk, dr, zshare, rs[0], rs[1])
        assert proof == ctx.create_proof_dev(pk, dr, dz.ptr, mont1(r), mont1(s))
        secret = ctx.upload(cv.fr_to_mont([rng.fr() for _ in range(100)])) if rank == 0 else None
        mine = party.king_share_vec(secret.ptr if secret else None, 100)
        tot = party.be.vec("ks", 100)
        party.be.open_vec(mine, tot, 100)
        if rank == 0:
            assert np.array_equal(ctx.download(tot, (100, 4)), ctx.download(secret, (100, 4)))
        ctx.close()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_rccl_multi_rank_opens_and_proof(world):
    """Runs only on a box with at least `world` GPUs (the round's 1-GPU boxes skip it): MpcNet::broadcast_bytes semantics
    (mpc-net/src/multi.rs:469-525) over RCCL with more than one rank."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, torch.cuda.device_count()))
    import socket
    import torch.multiprocessing as mp
    from helpers import free_port
    port = free_port()
    c = mp.get_context("spawn")
    q = c.Queue()
    procs = [c.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=900) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(r[1] == "ok" for r in res), res


class _SumRng:
    """The single prover's view of P parties' generators: every draw is the sum of the parties' draws (MpcField::rand gives
    each party a share of the prover's randomness)."""

    def __init__(self, seeds):
        from zk_mpc_amd.api import Rng
        self.rngs = [Rng.from_seed(s, 20) for s in seeds]

    def fill_fr(self, n):
        tot = [0] * n
        for r in self.rngs:
            for i, v in enumerate(cv.fr_from_mont(r.fill_fr(n))):
                tot[i] = (tot[i] + v) % O.R_MOD
        return cv.fr_to_mont(tot) if n else np.zeros((0, 4), dtype=np.uint64)


@pytest.mark.parametrize("n_parties,n,spdz,mask_dev", [(2, 6, False, False), (3, 40, False, True), (2, 13, True, True), (8, 21, True, False),
                                                      (3, "dense5", False, False), (3, "dense6", True, False), (3, "tiny9", True, True)])
def test_collaborative_marlin_full_proof(n_parties, n, spdz, mask_dev):
    """MpcMarlin::prove as a PROOF (src/marlin.rs:56): each party runs the rounds on its shares with its own generator, the
    witness-dependent commitments / evaluations / opening witnesses are revealed (MAC-checked under SPDZ), the transcript
    runs on the revealed values.  Every party ends with the same bytes; they equal the single prover's proof on the summed
    inputs with the summed randomness; the oracle's Marlin::verify accepts them and rejects a wrong public input."""
    import marlin_full_ref as MF
    import marlin_ref as M
    import pyseq.marlin_seq as DM
    from zk_mpc_amd.api import Rng
    from helpers import marlin_test_system
    rng = O.Prng(7700 + (n if isinstance(n, int) else 50 + int(n[-1])) + n_parties)
    r1cs, z = marlin_test_system(n, rng)          # an int: the mul-chain; "dense<k>" / "tiny<r>": circuit-shaped rows, |K| > |H|, several inputs
    sq, zz = M.pad_and_square(r1cs, z)
    zs = additive_shares(zz, n_parties, rng, public_prefix=sq.num_instance)
    zm = additive_shares(zz, n_parties, rng, public_prefix=sq.num_instance)
    beta_srs, g_k, gg_k, h_k = rng.fr(), rng.fr(), rng.fr(), rng.fr()
    seeds = [bytes((11 * p + i) & 0xff for i in range(32)) for p in range(n_parties)]
    a, b, c = DM.Csr.from_rows(sq.a), DM.Csr.from_rows(sq.b), DM.Csr.from_rows(sq.c)

    def setup(ctx):
        index = DM.Index(ctx, sq.num_instance, sq.num_witness, a, b, c)
        srs = DM.UniversalSrs(ctx, DM.ahp_max_degree(index) + 3, beta_srs, g_k, gg_k)
        return DM.IndexKeys(index, srs)

    def fn(p, ctx, net):
        party = (mpc.SpdzParty if spdz else mpc.Party)(ctx, net=net)
        keys = setup(ctx)
        up = lambda v: ctx.upload(cv.fr_to_mont(v))
        if spdz:
            proof = party.marlin_prove_full_spdz(keys, (up(zs[p]), up(zm[p])), Rng.from_seed(seeds[p], 20), mask_on_device=mask_dev)
        else:
            proof = party.marlin_prove_full(keys, up(zs[p]), Rng.from_seed(seeds[p], 20), mask_on_device=mask_dev)
        # the same proof as ONE library call (zk_marlin_prove_shared[_spdz]) under the same seeds: byte for byte
        if spdz:
            native = party.marlin_prove_shared_spdz_native(keys, (up(zs[p]), up(zm[p])), Rng.from_seed(seeds[p], 20), mask_on_device=mask_dev)
        else:
            native = party.marlin_prove_shared_native(keys, up(zs[p]), Rng.from_seed(seeds[p], 20), mask_on_device=mask_dev)
        assert native == proof.serialize(ctx)
        return proof.serialize(ctx), proof.evaluations, [[(cc.comm_aff, cc.shifted_aff, cc.shifted is not None) for cc in rnd] for rnd in proof.commitments], \
            [(cv.g1_projective_to_affine(w), rv) for w, rv in proof.pc_proof]

    res = run_parties(n_parties, fn)
    assert all(r[0] == res[0][0] for r in res)
    ctx = Z.Context(0)
    try:
        keys = setup(ctx)
        if not mask_dev:        # with the mask sampled on the device the joint randomness is not the sum of the host streams
            local = DM.prove(keys, ctx.upload(cv.fr_to_mont(zz)), _SumRng(seeds))
            assert local.serialize(ctx) == res[0][0]
        oix = M.Index(sq)
        pp = O.KzgParams(keys.srs.max_degree, beta_srs, g_k=g_k, gg_k=gg_k, h_k=h_k)
        okeys = MF.Keys(oix, pp)
        as_oracle = MF.Proof(res[0][2], res[0][1], res[0][3])
        pub = zz[1:oix.num_instance]
        assert MF.verify(okeys, pub, as_oracle)
        assert not MF.verify(okeys, [(pub[0] + 1) % O.R_MOD] + pub[1:], as_oracle)
    finally:
        ctx.close()


@pytest.mark.parametrize("spdz", [False, True])
def test_native_collaborative_marlin_with_real_triples_and_tampering(spdz):
    """zk_marlin_prove_shared[_spdz] with REAL Beaver triples for the round-2 product (the dummy source is the reference's
    default; a deployment brings triples from the preprocessing phase): the proof verifies in the oracle.  Under SPDZ a
    party that lies about one MAC share of its assignment makes every party's call fail with MacCheckError."""
    import marlin_full_ref as MF
    import marlin_ref as M
    import pyseq.marlin_seq as DM
    from zk_mpc_amd.api import Rng
    n_parties, n = 3, 11
    rng = O.Prng(4242 + spdz)
    r1cs, z = O.mul_chain_r1cs(n, rng.fr(), rng.fr())
    sq, zz = M.pad_and_square(r1cs, z)
    zs = additive_shares(zz, n_parties, rng, public_prefix=sq.num_instance)
    zm = additive_shares(zz, n_parties, rng, public_prefix=sq.num_instance)
    beta_srs, g_k, gg_k, h_k = rng.fr(), rng.fr(), rng.fr(), rng.fr()
    seeds = [bytes((23 * p + i) & 0xff for i in range(32)) for p in range(n_parties)]
    a, b, c = DM.Csr.from_rows(sq.a), DM.Csr.from_rows(sq.b), DM.Csr.from_rows(sq.c)
    n_h = len(zz)
    n_mul = 1 << (3 * n_h + 1 - 1).bit_length()                 # the multiplication domain of round 2
    ta = [rng.fr() for _ in range(n_mul)]
    tb = [rng.fr() for _ in range(n_mul)]
    tc = [x * y % O.R_MOD for x, y in zip(ta, tb)]
    tr = {k: (additive_shares(v, n_parties, rng), additive_shares(v, n_parties, rng)) for k, v in (("a", ta), ("b", tb), ("c", tc))}

    def setup(ctx):
        index = DM.Index(ctx, sq.num_instance, sq.num_witness, a, b, c)
        srs = DM.UniversalSrs(ctx, DM.ahp_max_degree(index) + 3, beta_srs, g_k, gg_k)
        return DM.IndexKeys(index, srs)

    def fn(p, ctx, net, tamper=False):
        party = (mpc.SpdzParty if spdz else mpc.Party)(ctx, net=net)
        keys = setup(ctx)
        up = lambda v: ctx.upload(cv.fr_to_mont(v))
        bufs = {(k, lane): up(tr[k][lane][p]) for k in "abc" for lane in (0, 1)}
        mac = list(zm[p])
        if tamper and p == 1:
            mac[sq.num_instance + 2] = (mac[sq.num_instance + 2] + 1) % O.R_MOD
        if spdz:
            triple = tuple((bufs[(k, 0)].ptr, bufs[(k, 1)].ptr) for k in "abc")
            try:
                return party.marlin_prove_shared_spdz_native(keys, (up(zs[p]), up(mac)), Rng.from_seed(seeds[p], 20), triple=triple)
            except mpc.MacCheckError:
                return "mac"
        return party.marlin_prove_shared_native(keys, up(zs[p]), Rng.from_seed(seeds[p], 20), triple=tuple(bufs[(k, 0)].ptr for k in "abc"))

    res = run_parties(n_parties, fn)
    assert all(r == res[0] for r in res) and isinstance(res[0], bytes) and len(res[0]) > 900
    ctx = Z.Context(0)
    try:
        keys = setup(ctx)
        local = DM.prove(keys, ctx.upload(cv.fr_to_mont(zz)), _SumRng(seeds))     # triples do not change the opened values
        assert local.serialize(ctx) == res[0]
    finally:
        ctx.close()
    if spdz:
        assert run_parties(n_parties, lambda p, ctx, net: fn(p, ctx, net, tamper=True)) == ["mac"] * n_parties


# ---- several OS processes, one GPU, host-staged transport (gloo) ---------------------------------------------------------------
def _marlin_proc_worker(rank, world, port, spdz, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import marlin_ref as M
    import pyseq.marlin_seq as DM
    from zk_mpc_amd.api import Rng
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = None
    try:
        n = 9
        rng = O.Prng(8800 + world)                            # same seed on every rank: same circuit, same shares
        r1cs, z = O.mul_chain_r1cs(n, rng.fr(), rng.fr())
        sq, zz = M.pad_and_square(r1cs, z)
        zs = additive_shares(zz, world, rng, public_prefix=sq.num_instance)
        zm = additive_shares(zz, world, rng, public_prefix=sq.num_instance)
        beta_srs, g_k, gg_k, h_k = rng.fr(), rng.fr(), rng.fr(), rng.fr()
        ctx = Z.Context(0, rank, world)                       # every process its own context on the one GPU
        net = mpc.DistNet(dist)                               # CPU tensors over gloo: the opens are staged through host memory
        party = (mpc.SpdzParty if spdz else mpc.Party)(ctx, net=net)
        a, b, c = DM.Csr.from_rows(sq.a), DM.Csr.from_rows(sq.b), DM.Csr.from_rows(sq.c)
        index = DM.Index(ctx, sq.num_instance, sq.num_witness, a, b, c)
        srs = DM.UniversalSrs(ctx, DM.ahp_max_degree(index) + 3, beta_srs, g_k, gg_k)
        keys = DM.IndexKeys(index, srs)
        up = lambda v: ctx.upload(cv.fr_to_mont(v))
        seed = bytes((17 * rank + i) & 0xff for i in range(32))
        if spdz:
            proof = party.marlin_prove_full_spdz(keys, (up(zs[rank]), up(zm[rank])), Rng.from_seed(seed, 20))
        else:
            proof = party.marlin_prove_full(keys, up(zs[rank]), Rng.from_seed(seed, 20))
        if spdz:                                                # the one-call provers over the same transport: the same bytes
            native = party.marlin_prove_shared_spdz_native(keys, (up(zs[rank]), up(zm[rank])), Rng.from_seed(seed, 20))
        else:
            native = party.marlin_prove_shared_native(keys, up(zs[rank]), Rng.from_seed(seed, 20))
        assert native == proof.serialize(ctx)
        # the one-call collaborative Groth16 provers over the same transport: the library calls back into gloo for its two small
        # exchanges and (staged through host memory) for the vector opens
        gr = O.Prng(9900 + world)
        g_r1cs, g_z = O.mul_chain_r1cs(50, gr.fr(), gr.fr())
        td = O.Trapdoor(gr.fr(), gr.fr(), gr.fr(), gr.fr(), gr.fr(), gr.fr(), gr.fr())
        rr, ss = gr.fr(), gr.fr()
        gzs = additive_shares(g_z, world, gr, public_prefix=2)
        gzm = additive_shares(g_z, world, gr, public_prefix=2)
        rsh, ssh = O.additive_share(rr, world, gr), O.additive_share(ss, world, gr)
        rm, sm = O.additive_share(rr, world, gr), O.additive_share(ss, world, gr)
        tdm = td_mont(td)
        dr = ctx.r1cs_mul_chain(50)
        pk = ctx.groth16_setup(dr, *[tdm[i] for i in range(7)])
        dzs_, dzm_ = up(gzs[rank]), up(gzm[rank])          # (kept alive: a DevBuf frees its memory when it is collected)
        if spdz:
            g16 = party.create_proof_shared_spdz_native(pk, dr, (dzs_.ptr, dzm_.ptr), (mont1(rsh[rank]), mont1(rm[rank])),
                                                        (mont1(ssh[rank]), mont1(sm[rank])))
        else:
            g16 = party.create_proof_shared_native(pk, dr, dzs_.ptr, mont1(rsh[rank]), mont1(ssh[rank]))
        q.put((rank, proof.serialize(ctx), proof.evaluations,
               [[(cc.comm_aff, cc.shifted_aff, cc.shifted is not None) for cc in rnd] for rnd in proof.commitments],
               [(cv.g1_projective_to_affine(w), rv) for w, rv in proof.pc_proof], srs.max_degree, int(party.bytes_sent), g16))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERROR %s\n%s" % (e, traceback.format_exc())))
    finally:
        if ctx is not None:
            ctx.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,spdz", [(2, False), (3, False), (2, True), (3, True)])
def test_collaborative_marlin_across_os_processes(world, spdz):
    """MpcMarlin::prove (src/marlin.rs:56) with the parties as OS PROCESSES, as the reference runs them (one process per party
    over its TCP mesh, examples/bin_test_marlin.rs) -- here `world` processes on one GPU, torch.distributed/gloo between them,
    every open staged through host memory.  All parties end with the same proof bytes; the oracle's Marlin::verify accepts
    them and rejects a wrong public input."""
    import socket
    import torch.multiprocessing as mp
    import marlin_full_ref as MF
    import marlin_ref as M
    from helpers import free_port
    port = free_port()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_marlin_proc_worker, args=(r, world, port, spdz, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=60) for p in procs]
    assert all(len(r) > 2 for r in res), [r[1] for r in res if len(r) == 2]
    assert all(r[1] == res[0][1] for r in res) and len(res[0][1]) > 900
    rng = O.Prng(8800 + world)
    r1cs, z = O.mul_chain_r1cs(9, rng.fr(), rng.fr())
    sq, zz = M.pad_and_square(r1cs, z)
    additive_shares(zz, world, rng, public_prefix=sq.num_instance); additive_shares(zz, world, rng, public_prefix=sq.num_instance)
    beta_srs, g_k, gg_k, h_k = rng.fr(), rng.fr(), rng.fr(), rng.fr()
    oix = M.Index(sq)
    okeys = MF.Keys(oix, O.KzgParams(res[0][5], beta_srs, g_k=g_k, gg_k=gg_k, h_k=h_k))
    as_oracle = MF.Proof(res[0][3], res[0][2], res[0][4])
    pub = zz[1:oix.num_instance]
    assert MF.verify(okeys, pub, as_oracle)
    assert not MF.verify(okeys, [(pub[0] + 1) % O.R_MOD] + pub[1:], as_oracle)
    assert all(r[6] > 0 for r in res)
    # the Groth16 proof of zk_groth16_prove_shared[_spdz] across the same processes: the local proof on the summed inputs
    gr = O.Prng(9900 + world)
    g_r1cs, g_z = O.mul_chain_r1cs(50, gr.fr(), gr.fr())
    td = O.Trapdoor(gr.fr(), gr.fr(), gr.fr(), gr.fr(), gr.fr(), gr.fr(), gr.fr())
    rr, ss = gr.fr(), gr.fr()
    want = O.proof_serialize(*O.predict_proof(g_r1cs, O.ProvingKeyScalars(g_r1cs, td), g_z, rr, ss))
    assert all(r[7] == want for r in res)
