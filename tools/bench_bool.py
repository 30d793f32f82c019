#!/usr/bin/env python3
"""The boolean-heavy leg of bench.py alone (for rocprofv3 --kernel-trace: `-- python3 tools/bench_bool.py [log_n]`)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
import zk_mpc_amd as Z  # noqa: E402
import zk_mpc_amd.convert as cv  # noqa: E402


def main():
    log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = Z.Context(0)
    td = [cv.fr_to_mont([B.seeded_fr(i)])[0] for i in range(1, 8)]
    print(json.dumps(B.boolean_heavy_leg(ctx, log_n, td, os.cpu_count() or 1)))
    ctx.close()


if __name__ == "__main__":
    main()
