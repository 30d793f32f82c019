#!/usr/bin/env python3
"""The HBM-bound kernels of the path against the HBM roofline: Fr vector ops (3 x 32 B per element), the sum over
parties of an open (32 (P + 1) B per element), the Beaver combine, the CSR mat-vec of the mul-chain system, the random share
sampler.  Prints microseconds and GB/s of ALGORITHMIC bytes and the fraction of 8 TB/s.  Run on an MI355X."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_mpc_amd as Z
import zk_mpc_amd.convert as cv

PEAK = 8000.0   # GB/s (MI355X_MICROARCH.md)


def timed(ctx, fn, reps=20):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


def main():
    ctx = Z.Context(0)
    rs = np.random.RandomState(3)
    for lg in (20, 22, 24):
        n = 1 << lg
        a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        da, db, dc = ctx.upload(a), ctx.upload(a[::-1].copy()), ctx.alloc(n * 32)
        rows = []
        for name, op in (("vec_add", 1), ("vec_mul", 0)):
            dt = timed(ctx, lambda: ctx.fr_vec_op_dev(op, da.ptr, db.ptr, dc.ptr, n))
            rows.append((name, 96 * n, dt))
        dt = timed(ctx, lambda: ctx.fr_vec_scale_dev(da.ptr, cv.fr_to_mont([12345])[0], dc.ptr, n))
        rows.append(("vec_scale", 64 * n, dt))
        dt = timed(ctx, lambda: ctx.beaver_combine_dev(da.ptr, db.ptr, dc.ptr, n))
        rows.append(("beaver_combine (dummy triple)", 96 * n, dt))
        dt = timed(ctx, lambda: ctx.fr_random_dev(dc.ptr, n, b"k" * 32))
        rows.append(("fr_random (ChaCha20 per element)", 32 * n, dt))
        for P in (3, 8):
            m = n // P
            dt = timed(ctx, lambda: ctx.fr_sum_parties_dev(da.ptr, P, m, dc.ptr))
            rows.append(("sum_parties P=%d" % P, 32 * (P + 1) * m, dt))
        if lg == 20:
            r1cs = ctx.r1cs_mul_chain(n - 2)
            dt = timed(ctx, lambda: ctx.r1cs_matvec_dev(r1cs, 0, da.ptr, dc.ptr, n))
            rows.append(("spmv (mul-chain A, 1 non-zero per row)", 64 * (n - 2) + 16 * (n - 2), dt))
        for name, nbytes, dt in rows:
            print(json.dumps({"kernel": name, "log_n": lg, "us": round(dt * 1e6, 1), "algorithmic_GBps": round(nbytes / dt / 1e9, 1),
                              "frac_of_hbm_peak": round(nbytes / dt / 1e9 / PEAK, 3)}))
        da.free(); db.free(); dc.free()


if __name__ == "__main__":
    main()
