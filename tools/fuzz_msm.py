#!/usr/bin/env python3
"""Randomised MSM check on an MI355X (run from the repo root): random lengths 1 .. 2^17 (tile boundaries of the bucket sort, every
window width), base offsets, G1 and G2, plain tables and window multiples, scalar mixtures (uniform, zeros, ones, +-1, small
values, repeated values).  bases = k_i G, so sum s_i (k_i G) must equal (sum s_i k_i) G, the inner product computed from the
device's own Fr products and summed exactly on the host (tests/test_gpu_msm.py::_mont_inner_product)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import zk_mpc_amd as Z, zk_mpc_amd.convert as cv
import zkref as O
from test_gpu_msm import _mont_inner_product

def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    ctx = Z.Context(0)
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    one = cv.fr_to_mont([1])[0]
    minus_one = cv.fr_to_mont([O.R_MOD - 1])[0]
    NMAX = 1 << 17
    k = rs.randint(0, 1 << 62, size=(NMAX + 64, 4), dtype=np.uint64); k[:, 3] &= np.uint64((1 << 60) - 1)
    dk = ctx.upload(k)
    tables = {}
    for group in (1, 2):
        b = ctx.fixed_base(dk.ptr, NMAX + 64, group, one)
        p = ctx.fixed_base(dk.ptr, NMAX + 64, group, one)
        p.precompute()
        tables[group] = (b, p)
    bad = 0
    t0 = time.time()
    for it in range(iters):
        group = 1 if rs.rand() < 0.7 else 2
        n = int(rs.choice([rs.randint(1, 64), rs.randint(1, 5000), rs.randint(1000, NMAX), 1024 * rs.randint(1, 64) + rs.randint(-2, 3)]))
        n = max(1, min(n, NMAX))
        off = int(rs.randint(0, 64))
        a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
        pick = rs.rand(n)
        kind = rs.randint(0, 6)
        if kind == 1: a[pick < 0.9] = 0; a[(pick >= 0.45) & (pick < 0.9)] = one
        elif kind == 2: a[pick < 0.5] = one; a[pick >= 0.5] = minus_one
        elif kind == 3: a[:] = a[rs.randint(0, n)]
        elif kind == 4:
            sm = cv.fr_to_mont([int(v) for v in rs.randint(0, 1 << 20, size=257)]); a = np.ascontiguousarray(sm[rs.randint(0, 257, size=n)])
        elif kind == 5: a[pick < 0.97] = 0
        ds = ctx.upload(np.ascontiguousarray(a))
        ip = _mont_inner_product(ctx, dk.ptr + off * 32, ds.ptr, n)
        to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
        want = O.g1_mul(O.G1_GEN, ip) if group == 1 else O.g2_mul(O.G2_GEN, ip)      # the oracle's scalar multiplication
        for name, tb in zip(("plain", "window multiples"), tables[group]):
            got = to_aff(ctx.msm_dev(tb, off, ds.ptr, n))
            if got != want:
                bad += 1
                print("MISMATCH it=%d group=%d n=%d off=%d kind=%d table=%s" % (it, group, n, off, kind, name), flush=True)
        ds.free()
    # round 5: SMALL tables (256 .. 8192 points) with their own window multiples -- one bucket set, segments of 8, the single-block
    # sort, and, through zk_msm_batch_dev, the group launches (2 .. 7 jobs over one table: one accumulate launch, one launch per
    # reduce level) -- every job against the inner product, the batch against the single calls
    small = 0
    for it in range(max(iters // 3, 10)):
        group = 1 if rs.rand() < 0.75 else 2
        nt = int(rs.choice([256, 257, 1000, 1024, 3000, 4095, 4096, 8192, rs.randint(256, 8192)]))
        off_t = int(rs.randint(0, 1000))
        tb = ctx.fixed_base(dk.ptr + off_t * 32, nt, group, one)
        tb.precompute()
        to_aff = cv.g1_projective_to_affine if group == 1 else cv.g2_projective_to_affine
        gen = (lambda e: O.g1_mul(O.G1_GEN, e)) if group == 1 else (lambda e: O.g2_mul(O.G2_GEN, e))
        jobs, wants, keep = [], [], []
        for j in range(int(rs.randint(1, 8))):
            n = int(rs.choice([nt, nt - 1, max(1, nt // 2), max(1, nt // 8), int(rs.randint(1, nt + 1))]))
            off = int(rs.randint(0, nt - n + 1))
            a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
            kind = rs.randint(0, 5)
            pick = rs.rand(n)
            if kind == 1: a[pick < 0.9] = 0; a[(pick >= 0.45) & (pick < 0.9)] = one
            elif kind == 2: a[:] = a[rs.randint(0, n)]
            elif kind == 3: a[pick < 0.5] = one; a[pick >= 0.5] = minus_one
            ds = ctx.upload(np.ascontiguousarray(a))
            keep.append(ds)
            jobs.append((tb, off, ds.ptr, n))
            wants.append(gen(_mont_inner_product(ctx, dk.ptr + (off_t + off) * 32, ds.ptr, n)))
        singles = [to_aff(ctx.msm_dev(*j)) for j in jobs]
        batch = [to_aff(x) for x in ctx.msm_batch_dev(jobs)]
        small += 1
        if singles != wants or batch != wants:
            bad += 1
            print("MISMATCH small it=%d group=%d table=%d jobs=%s single_ok=%s batch_ok=%s" % (it, group, nt, [(j[1], j[3]) for j in jobs], singles == wants, batch == wants), flush=True)
        for ds in keep: ds.free()
        tb.free()
    print("FUZZ %s: %d cases + %d small-table batches, %d mismatches, %.1f s" % ("FAILED" if bad else "ok", iters, small, bad, time.time() - t0))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
