#!/usr/bin/env python3
"""Where the collaborative Groth16 path spends its time on one rank (nccl transport, world size 1): witness-map halves,
Beaver product, the five MSMs, and the host-side group Beaver / reveals, each bracketed by a device sync."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import zk_mpc_amd as Z  # noqa: E402
import zk_mpc_amd.convert as cv  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pyseq.mpc_seq as mpc  # noqa: E402  (the Python sequences: test infrastructure)

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = Z.Context(0, 0, 1)
mont = lambda v: cv.fr_to_mont([v])[0]
n = (1 << 20) - 2
r1cs = ctx.r1cs_mul_chain(n)
pk = ctx.groth16_setup(r1cs, *[mont(i + 11) for i in range(7)])
z = ctx.mul_chain_assignment_dev(n, mont(3), mont(5))
party = mpc.Party(ctx, dist)
zs = party.share_assignment_dev(z, r1cs, seed=1)
rs = party.share_scalars([7, 9], seed=2)
be = party.be
D = be.domain_size(r1cs)
for it in range(4):
    ctx.sync()
    t = [time.perf_counter()]

    def lap():
        ctx.sync()
        torch.cuda.synchronize()
        t.append(time.perf_counter())
    a, b, c = be.vec("wm_a", D), be.vec("wm_b", D), be.vec("wm_c", D)
    be.witness_map_pre(r1cs, zs, a, b, c); lap()
    party.beaver_batch_mul(a, b, a, D, None); lap()
    be.witness_map_post(r1cs, a, c); lap()
    g1, g2 = be.msms(pk, r1cs, zs, a); lap()
    P = be.pk_points(pk); lap()
    r_g1 = be.g1_mul(P["delta_g1"], rs[0]); lap()
    x = party.scale_g1(r_g1, rs[1]); lap()
    y = party.reveal_g1(x); lap()
    proof = party.create_proof_shared(pk, r1cs, zs, rs[0], rs[1]); lap()
    names = ["wm_pre", "beaver", "wm_post", "msms", "pk_points", "g1_mul(host)", "scale_g1", "reveal_g1", "whole proof"]
    print({k: round((t[i + 1] - t[i]) * 1e3, 2) for i, k in enumerate(names)})
dist.destroy_process_group()
