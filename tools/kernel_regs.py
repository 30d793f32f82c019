#!/usr/bin/env python3
"""Register / spill / scratch / LDS figures of every kernel in hipcc device assembly (the .amdhsa metadata block).

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S zk-mpc_amd/csrc/msm.hip -o /tmp/msm.s
  python tools/kernel_regs.py /tmp/msm.s [substring]
"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if pat in name:
        print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>4s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} "
              f"scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>6s}")
