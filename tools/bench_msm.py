#!/usr/bin/env python3
"""MSM / NTT micro-benchmarks: Mscalar/s for G1 and G2 at n = 2^16 .. 2^24, NTT GB/s for log n = 16 .. 24.
Bases = k_i * G from the device fixed-base kernel, scalars = arbitrary residues < 2^252.  Run on an MI355X."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_mpc_amd as Z
import zk_mpc_amd.convert as cv

def main():
    ctx = Z.Context(0)
    rs = np.random.RandomState(1)
    out = {"msm": [], "ntt": []}
    max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n_max = 1 << max_log
    a = rs.randint(0, 1 << 62, size=(n_max, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
    d = ctx.upload(a)
    one = cv.fr_to_mont([1])[0]
    for group, top in ((1, max_log), (2, max_log - 2)):
        bases = ctx.fixed_base(d.ptr, 1 << top, group, one)
        ctx.sync()
        for lg in reversed(range(16, top + 1, 2)):      # largest first: the scratch arena is sized once
            n = 1 << lg
            for _ in range(4):                      # the first calls at a new size grow the scratch arena
                ctx.msm_dev(bases, 0, d.ptr, n)
            ctx.sync()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.msm_dev(bases, 0, d.ptr, n)
            ctx.sync()
            dt = (time.perf_counter() - t0) / reps
            out["msm"].append({"group": "G%d" % group, "log_n": lg, "ms": round(dt * 1e3, 3), "mscalar_per_s": round(n / dt / 1e6, 1)})
        if group == 1:
            # adversarial scalar sets: all-equal, all-zero, 0/1-heavy like a real witness (90 % of the scalars 0 or 1)
            n = 1 << min(20, top)
            sets = {}
            sets["all_equal"] = np.tile(a[12345:12346], (n, 1))
            sets["all_zero"] = np.zeros((n, 4), dtype=np.uint64)
            z01 = a[:n].copy()
            pick = rs.rand(n)
            z01[pick < 0.9] = 0
            z01[(pick >= 0.45) & (pick < 0.9), 0] = 1
            one_m = cv.fr_to_mont([1])[0]
            z01[(pick >= 0.45) & (pick < 0.9)] = one_m
            sets["zero_one_heavy"] = z01
            for name, arr in sets.items():
                dd = ctx.upload(np.ascontiguousarray(arr))
                ctx.msm_dev(bases, 0, dd.ptr, n); ctx.sync()
                t0 = time.perf_counter()
                for _ in range(3):
                    ctx.msm_dev(bases, 0, dd.ptr, n)
                ctx.sync()
                dt = (time.perf_counter() - t0) / 3
                out["msm"].append({"group": "G1", "log_n": 20, "scalars": name, "ms": round(dt * 1e3, 3), "mscalar_per_s": round(n / dt / 1e6, 1)})
                dd.free()
        bases.free()
    for lg in range(16, max_log + 1, 2):
        n = 1 << lg
        for inv, cos in ((0, 0), (1, 1)):
            ctx.ntt_dev(d.ptr, lg, inv, cos); ctx.sync()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.ntt_dev(d.ptr, lg, inv, cos)
            ctx.sync()
            dt = (time.perf_counter() - t0) / reps
            out["ntt"].append({"log_n": lg, "inverse": inv, "coset": cos, "us": round(dt * 1e6, 1),
                               "algorithmic_GBps": round(2 * 32 * n / dt / 1e9, 1)})
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
