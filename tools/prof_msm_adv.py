#!/usr/bin/env python3
"""One adversarial scalar set through the G1 MSM, for a kernel trace:  rocprofv3 --kernel-trace --stats -- python3 tools/prof_msm_adv.py SET [LOG] [pre]
SET: uniform | all_zero | all_equal | zero_one_heavy | small_values.  `pre`: the table carries window multiples (one bucket set)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_mpc_amd as Z
import zk_mpc_amd.convert as cv


def scalar_set(name, n, rs):
    a = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    one = cv.fr_to_mont([1])[0]
    if name == "uniform":
        return a
    if name == "all_zero":
        return np.zeros((n, 4), dtype=np.uint64)
    if name == "all_equal":
        return np.tile(a[12345:12346], (n, 1))
    pick = rs.rand(n)
    if name == "zero_one_heavy":
        a[pick < 0.45] = 0
        a[(pick >= 0.45) & (pick < 0.9)] = one
        return a
    if name == "small_values":          # bytes and 16-bit values, like range-checked witness columns
        small = cv.fr_to_mont([int(v) for v in rs.randint(0, 1 << 16, size=4096)])
        idx = rs.randint(0, 4096, size=n)
        b = small[idx]
        b[pick >= 0.9] = a[pick >= 0.9]
        return np.ascontiguousarray(b)
    raise SystemExit("unknown set " + name)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    pre = len(sys.argv) > 3 and sys.argv[3] == "pre"
    n = 1 << lg
    ctx = Z.Context(0)
    rs = np.random.RandomState(1)
    k = rs.randint(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    k[:, 3] &= np.uint64((1 << 60) - 1)
    dk = ctx.upload(k)
    bases = ctx.fixed_base(dk.ptr, n, 1, cv.fr_to_mont([1])[0])
    if pre:
        bases.precompute()
    d = ctx.upload(np.ascontiguousarray(scalar_set(name, n, rs)))
    for _ in range(3):
        ctx.msm_dev(bases, 0, d.ptr, n)
    ctx.sync()
    reps = []
    for _ in range(10):
        t0 = time.perf_counter()
        ctx.msm_dev(bases, 0, d.ptr, n)
        reps.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"set": name, "log_n": lg, "window_multiples": pre, "ms_median": round(float(np.median(reps)), 3),
                      "ms_all": [round(x, 3) for x in reps]}))


if __name__ == "__main__":
    main()
