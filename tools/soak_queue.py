#!/usr/bin/env python3
"""Soak test of the announced-next-proof queue (run on an MI355X from the repo root): for circuit sizes on both sides of the
small-job scheduling boundary, 400 proofs each in random order with random announcements (some true, some not), over satisfying
and 0/1-heavy arbitrary assignments; every proof must equal the same assignment proved alone.  Round 4: 3 200 proofs, 0 differ."""
import os, sys, random, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import zk_mpc_amd as Z, zk_mpc_amd.convert as cv
import zkref as O
ctx = Z.Context(0)
rnd = random.Random(5)
rng = O.Prng(99)
mont = lambda v: cv.fr_to_mont([v])[0]
bad = 0; total = 0
t0 = time.time()
for n in [100, 1000, 5000, (1 << 14) - 2, (1 << 15) + 7, (1 << 16) - 2, (1 << 16) + 50, (1 << 18) - 2]:
    td = [mont(rng.fr()) for _ in range(7)]
    dr = ctx.r1cs_mul_chain(n)
    pk = ctx.groth16_setup(dr, *td)
    zs = [ctx.mul_chain_assignment_dev(n, mont(rng.fr()), mont(rng.fr())) for _ in range(2)]
    rs_np = np.random.RandomState(n)
    for dens in (0.9, 0.5):          # arbitrary (unsatisfying) assignments, 0/1-heavy: heavy buckets, split segments, folds
        a = rs_np.randint(0, 1 << 62, size=(n + 3, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
        pick = rs_np.rand(n + 3)
        a[pick < dens] = 0
        a[(pick >= dens / 2) & (pick < dens)] = mont(1)
        a[0] = mont(1)
        zs.append(ctx.upload(np.ascontiguousarray(a)))
    rs = [(mont(rng.fr()), mont(rng.fr())) for _ in range(4)]
    ref = []
    for z, (r, s) in zip(zs, rs):
        ref.append(ctx.create_proof_dev(pk, dr, z.ptr, r, s)); ctx.sync()
    nxt = None
    for it in range(400):
        k = nxt if (nxt is not None and rnd.random() < 0.8) else rnd.randrange(4)
        nxt = rnd.randrange(4) if rnd.random() < 0.7 else None
        ctx.groth16_hint_next_dev(zs[nxt].ptr if nxt is not None else None)
        p = ctx.create_proof_dev(pk, dr, zs[k].ptr, *rs[k])
        total += 1
        if p != ref[k]:
            bad += 1
            print("MISMATCH n=%d it=%d k=%d" % (n, it, k), flush=True)
    ctx.groth16_hint_next_dev(None)
    pk.free()
    print("n=%d done, %d proofs, %d bad, %.1f s" % (n, total, bad, time.time() - t0), flush=True)
print("SOAK", "FAILED" if bad else "ok", total, bad)
