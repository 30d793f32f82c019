"""CPU: the N>1 path.  Two OS processes, torch.distributed/gloo on 127.0.0.1, run the product's
collaborative protocol (zk_mpc_amd/mpc.py: Beaver batch multiply with all-gather opens, group Beaver
scale, reveal) with the test-only oracle arithmetic backend.  The revealed proof must equal the local
proof on the summed inputs (MPC parity) -- bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    from helpers import free_port
    return free_port()


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import zkref as O
    import zk_mpc_amd.convert as cv
    import pyseq.mpc_seq as mpc
    from oracle_backend import OracleBackend, OraclePk, additive_shares

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = O.Prng(4242)                                  # same seed on every rank: same circuit, same shares
        r1cs, z = O.mul_chain_r1cs(11, rng.fr(), rng.fr())
        td = O.Trapdoor(rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr(), rng.fr())
        pks = O.ProvingKeyScalars(r1cs, td)
        pk = OraclePk(O.ProvingKey(pks))
        r, s = rng.fr(), rng.fr()
        zs = additive_shares(z, world, rng, public_prefix=r1cs.num_instance)
        rsh, ssh = O.additive_share(r, world, rng), O.additive_share(s, world, rng)
        net = mpc.DistNet(dist)
        be = OracleBackend(net, r1cs)
        party = mpc.Party(net=net, backend=be)
        # (1) vector Beaver with a REAL (non-dummy) triple
        n = 16
        xs, ys = [rng.fr() for _ in range(n)], [rng.fr() for _ in range(n)]
        ta, tb = [rng.fr() for _ in range(n)], [rng.fr() for _ in range(n)]
        tc = [a * b % O.R_MOD for a, b in zip(ta, tb)]
        sh = lambda v: additive_shares(v, world, rng)[rank]
        X, Y = be.put("x", cv.fr_to_mont(sh(xs))), be.put("y", cv.fr_to_mont(sh(ys)))
        T = tuple(be.put(nm, cv.fr_to_mont(sh(v))) for nm, v in (("ta", ta), ("tb", tb), ("tc", tc)))
        out = be.vec("out", n)
        party.beaver_batch_mul(X, Y, out, n, triple=T)
        parts = [cv.fr_from_mont(a) for a in net.all_gather_small(be.store[out])]
        assert [sum(c) % O.R_MOD for c in zip(*parts)] == [a * b % O.R_MOD for a, b in zip(xs, ys)]
        # (2) dummy triple (the reference's DummyFieldTripleSource)
        party.beaver_batch_mul(X, Y, out, n)
        parts = [cv.fr_from_mont(a) for a in net.all_gather_small(be.store[out])]
        assert [sum(c) % O.R_MOD for c in zip(*parts)] == [a * b % O.R_MOD for a, b in zip(xs, ys)]
        # (3) the collaborative prover
        Z = be.put("z", cv.fr_to_mont(zs[rank]))
        proof = party.create_proof_shared(pk, r1cs, Z, cv.fr_to_mont([rsh[rank]])[0], cv.fr_to_mont([ssh[rank]])[0])
        want = O.proof_serialize(*O.predict_proof(r1cs, pks, z, r, s))
        assert proof == want, "revealed %d-party proof differs from the local proof on the summed inputs" % world
        # the reference's order of opens (one collective each) gives the same bytes as the fused opens (the default)
        assert party.create_proof_shared(pk, r1cs, Z, cv.fr_to_mont([rsh[rank]])[0], cv.fr_to_mont([ssh[rank]])[0], fused=False) == want
        assert party.bytes_sent >= 2 * be.dom.size * 32
        # (4) SPDZ (malicious backend): two-lane shares, MAC-checked opens; mac shares are independent sharings (key alpha = 1)
        sp = mpc.SpdzParty(net=net, backend=be)
        zm = additive_shares(z, world, rng, public_prefix=r1cs.num_instance)
        Zs = (be.put("zs", cv.fr_to_mont(zs[rank])), be.put("zm", cv.fr_to_mont(zm[rank])))
        m1 = lambda v: cv.fr_to_mont([v])[0]
        rm, sm = O.additive_share(r, world, rng), O.additive_share(s, world, rng)
        proof2 = sp.create_proof_shared_spdz(pk, r1cs, Zs, (m1(rsh[rank]), m1(rm[rank])), (m1(ssh[rank]), m1(sm[rank])))
        assert proof2 == want, "SPDZ proof differs"
        assert sp.create_proof_shared_spdz(pk, r1cs, Zs, (m1(rsh[rank]), m1(rm[rank])), (m1(ssh[rank]), m1(sm[rank])), fused=False) == want
        # a corrupted MAC share must be caught by the next open
        bad = np.array(be.store[Zs[1]], copy=True)
        if rank == world - 1:
            bad[3, 0] ^= np.uint64(1)
        Zbad = (Zs[0], be.put("zbad", bad))
        caught = False
        try:
            sp.create_proof_shared_spdz(pk, r1cs, Zbad, (m1(rsh[rank]), m1(rm[rank])), (m1(ssh[rank]), m1(sm[rank])))
        except mpc.MacCheckError:
            caught = True
        assert caught, "corrupted MAC share was not detected"
        # (5) the vector open of the GPU transport path (DistNet.open_sum: all-to-all of slices, local sum, all-gather of the
        # summed slices; all-gather-then-sum for 2 parties) on CPU tensors, incl. lengths that do not divide by the world size
        import torch
        import zkref_c as OC
        bufs = {}

        def buffer(name, nbytes):
            if (name, nbytes) not in bufs:
                bufs[(name, nbytes)] = torch.empty(nbytes // 8, dtype=torch.int64)
            return bufs[(name, nbytes)]

        def sum_parties(gathered, n_parts, m, dst):
            a = gathered.numpy().view(np.uint64)[: n_parts * m * 4].reshape(n_parts, m, 4)
            acc = a[0]
            for part in a[1:]:
                acc = OC.fr_vec_op(1, np.ascontiguousarray(acc), np.ascontiguousarray(part))
            dst[: m * 4] = torch.from_numpy(np.ascontiguousarray(acc).view(np.int64).reshape(-1))
        for n_open in (1, 7, 16, 17, 100):
            vals = [[rng.fr() for _ in range(n_open)] for _ in range(world)]
            mine = torch.from_numpy(cv.fr_to_mont(vals[rank]).view(np.int64).reshape(-1).copy())
            want_sum = [sum(c) % O.R_MOD for c in zip(*vals)]
            for mode in ("a2a", "allgather", None):
                net.open_pattern = mode                       # the same choice on every rank (a DistNet constructor argument)
                res = net.open_sum(mine, n_open, sum_parties, buffer)
                got = cv.fr_from_mont(res.numpy().view(np.uint64).reshape(-1, 4)[:n_open])
                assert got == want_sum, "open_sum(%s) wrong for n=%d" % (mode, n_open)
        net.open_pattern = None
        # (6) king_share: the leader splits a vector and scatters the shares (transport scatter); they sum to the vector
        secret = [rng.fr() for _ in range(23)]
        S = be.put("secret", cv.fr_to_mont(secret)) if rank == 0 else None
        mine = party.king_share_vec(S, 23, key32=bytes([5] * 32))
        parts = [cv.fr_from_mont(a) for a in net.all_gather_small(be.store[mine])]
        assert [sum(c) % O.R_MOD for c in zip(*parts)] == secret
        assert parts[0] != secret or world == 1
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_collaborative_prove_gloo(world):
    """world = 8 is BASELINE config 5's party count (SPDZ shares, the all-to-all open with 8 slices on lengths that do not
    divide by 8: mpc-net/src/multi.rs:469-525 ordering by party id, share/spdz.rs:177-196)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
