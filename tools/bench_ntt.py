#!/usr/bin/env python3
"""NTT micro-benchmark: microseconds and algorithmic GB/s (one read + one write of the vector) for
log n = 12 .. 24, forward and coset-inverse.  Run on an MI355X:  python tools/bench_ntt.py [max_log] [reps] [min_log]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zk_mpc_amd as Z


def main():
    ctx = Z.Context(0)
    rs = np.random.RandomState(1)
    max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    min_log = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    a = rs.randint(0, 1 << 62, size=(1 << max_log, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    d = ctx.upload(a)
    for lg in range(min_log, max_log + 1, 2):
        n = 1 << lg
        for inv, cos in ((0, 0), (1, 0), (0, 1), (1, 1)):
            ctx.ntt_dev(d.ptr, lg, inv, cos)
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.ntt_dev(d.ptr, lg, inv, cos)
            ctx.sync()
            dt = (time.perf_counter() - t0) / reps
            print(json.dumps({"log_n": lg, "inverse": inv, "coset": cos, "us": round(dt * 1e6, 1),
                              "algorithmic_GBps": round(2 * 32 * n / dt / 1e9, 1)}))


if __name__ == "__main__":
    main()
