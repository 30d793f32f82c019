#!/usr/bin/env python3
"""Throughput of the SHE ring kernels (row a15) on one GPU: Ciphertext::mul and Encodedtext::mul over random data.

Prints one JSON line per (N, batch): products/s, Fq753 Montgomery products/s and the fraction of the measured
v_mad_u64_u32 issue peak (tools/ubench_int.hip: ~32 T/s) that the 1 352 mads of each product account for."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zk_mpc_amd.api import Context  # noqa: E402

MAD_PEAK = 32.0e12
MADS_PER_MMUL = 2 * 26 * 26


def mmuls_ct_mul(n):
    lg = n.bit_length() - 1
    if n & (n - 1) or n < 4:
        return 4 * n * n + 3 * n
    return 7 * (n // 2) * lg + 4 * n + 3 * n


def mmuls_et_mul(n):
    lg = n.bit_length() - 1
    if n & (n - 1) or n < 4:
        return n * n + n
    return 3 * (n // 2) * lg + n + n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="3,64,256,1024,4096,16384")
    ap.add_argument("--elems", type=int, default=1 << 21, help="coefficients per operand polynomial set (batch = elems / N)")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    ctx = Context(0)
    rng = np.random.default_rng(7)
    for n in [int(x) for x in args.sizes.split(",")]:
        batch = max(1, args.elems // n)
        if n == 3:
            batch = min(batch, 1 << 16)
        # random 768-bit words with the top bits cleared are valid (< q) Montgomery representations
        def rnd(k):
            a = rng.integers(0, 1 << 63, size=(k, 12), dtype=np.uint64)
            a[:, 11] &= np.uint64((1 << 46) - 1)
            return a
        x, y = ctx.upload(rnd(batch * 3 * n)), ctx.upload(rnd(batch * 3 * n))
        out = ctx.alloc(batch * 3 * n * 96)
        for name, fn, cnt in (("ciphertext_mul", lambda: ctx.ciphertext_mul_dev(x.ptr, y.ptr, out.ptr, n, batch), mmuls_ct_mul(n)),
                              ("encodedtext_mul", lambda: ctx.encodedtext_mul_dev(x.ptr, y.ptr, out.ptr, n, batch), mmuls_et_mul(n))):
            fn()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
            ctx.sync()
            dt = (time.perf_counter() - t0) / args.reps
            mm = cnt * batch / dt
            print(json.dumps({"op": name, "N": n, "batch": batch, "ms": round(dt * 1e3, 3), "products_per_s": round(batch / dt, 1),
                              "fq753_mmul_per_s": round(mm, 1), "frac_of_mad_peak": round(mm * MADS_PER_MMUL / MAD_PEAK, 3),
                              "coeff_bytes_per_s": round(batch * n * 96 * (9 if name == "ciphertext_mul" else 3) / dt, 1)}), flush=True)
        del x, y, out


if __name__ == "__main__":
    main()
