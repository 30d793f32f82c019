#!/usr/bin/env python3
"""Host wall-clock laps of zk_marlin_prove's phases at small sizes (zk_set_profiling: "marlin.<phase>" timers, a lap includes
the device work the host waited for) next to the whole-call time without profiling.  python tools/marlin_laps.py 10,12,14"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zk_mpc_amd import marlin as DM  # noqa: E402
from zk_mpc_amd.api import Context, Rng  # noqa: E402


def main():
    logs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,12,14").split(",")]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctx = Context(0)
    m = DM.HostField.m
    for L in logs:
        n = (1 << L) - 3
        ni, nw, a, b, c = DM.mul_chain_system(ctx, n)
        index = DM.Index(ctx, ni, nw, a, b, c)
        srs = DM.UniversalSrs(ctx, DM.ahp_max_degree(index) + 5, 0x1234567, 3, 7)
        keys = DM.IndexKeys(index, srs)
        z = ctx.mul_chain_assignment_dev(n, m(3), m(5))
        seed = bytes(range(32))
        ctx.pooling = True
        for _ in range(4):
            DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=True)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=True)
        ctx.sync()
        plain = (time.perf_counter() - t0) / reps * 1e3
        ctx.set_profiling(True)
        ctx.timers()
        t0 = time.perf_counter()
        for _ in range(reps):
            DM.prove_native(keys, z, Rng.from_seed(seed, 20), mask_on_device=True)
        ctx.sync()
        prof = (time.perf_counter() - t0) / reps * 1e3
        tm = ctx.timers()
        ctx.set_profiling(False)
        ctx.pooling = False
        ctx.drop_pool()
        print(json.dumps({"marlin_log": L, "ms": round(plain, 3), "ms_profiled": round(prof, 3),
                          "laps_ms": {k[7:]: round(v[0] / reps, 3) for k, v in tm.items() if k.startswith("marlin.")},
                          "device_ms": {k: [round(v[0] / reps, 3), v[1] // reps] for k, v in tm.items() if not k.startswith("marlin.")}}), flush=True)
        del index, srs, keys, z


if __name__ == "__main__":
    main()
