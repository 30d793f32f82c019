#!/usr/bin/env python3
"""Small proofs as a throughput workload: T host threads, each with its own context (own key copy, own streams and scratch), proving
back to back on ONE GPU.  A small proof leaves most of the chip idle for most of its 0.6 ms; contexts are independent, so the proofs
of different threads overlap.  python tools/bench_small_throughput.py <log_constraints> <threads,...> [proofs per thread] [chain 0|1]
(chain: zk_groth16_chain_fronts -- one context does better with it, several sharing a GPU without)"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import zk_mpc_amd as Z  # noqa: E402
import zk_mpc_amd.convert as cv  # noqa: E402


def seeded_fr(i):
    return (0x9E3779B97F4A7C15 * (i + 1) ** 3 + 12345) % cv.R_MOD if hasattr(cv, "R_MOD") else (0x9E3779B97F4A7C15 * (i + 1) ** 3 + 12345)


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    threads = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4").split(",")]
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    chain = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    n = (1 << L) - 2
    mont = lambda v: cv.fr_to_mont([v])[0]
    td = [mont(1000 + i) for i in range(1, 8)]
    for T in threads:
        parties = []
        for t in range(T):
            ctx = Z.Context(0)
            ctx.groth16_chain_fronts(bool(chain))
            r1cs = ctx.r1cs_mul_chain(n)
            pk = ctx.groth16_setup(r1cs, *td)
            zs = [ctx.mul_chain_assignment_dev(n, mont(100 + 10 * q + t), mont(101 + 10 * q + t)) for q in range(4)]
            rs = (mont(200 + t), mont(201 + t))
            for i in range(6):
                ctx.create_proof_dev(pk, r1cs, zs[i % 4].ptr, *rs)
            parties.append((ctx, r1cs, pk, zs, rs))
        barrier = threading.Barrier(T + 1)
        proofs = [None] * T

        def work(t):
            ctx, r1cs, pk, zs, rs = parties[t]
            barrier.wait()
            for i in range(K):
                ctx.groth16_hint_next_dev(zs[(i + 1) % 4].ptr)
                proofs[t] = ctx.create_proof_dev(pk, r1cs, zs[i % 4].ptr, *rs)
            barrier.wait()
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        for x in th:
            x.start()
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        dt = time.perf_counter() - t0
        for x in th:
            x.join()
        print(json.dumps({"groth16_log": L, "contexts": T, "chain_fronts": chain, "proofs": T * K, "ms_per_proof_aggregate": round(dt / (T * K) * 1e3, 4),
                          "proofs_per_s": round(T * K / dt, 1), "ms_per_proof_per_context": round(dt / K * 1e3, 3)}), flush=True)
        for ctx, r1cs, pk, zs, rs in parties:
            pk.free()
        del parties


if __name__ == "__main__":
    main()
