#!/usr/bin/env python3
"""Opcode histogram per basic block of one kernel in hipcc device assembly (where do the issue slots of an issue-bound kernel go).

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -S zk-mpc_amd/csrc/msm.hip -o /tmp/msm.s
  python tools/isa_mix.py /tmp/msm.s k_accumIN2zk7FqField [min_block_size]
"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 40
m = re.search(r"^(\S*%s\S*):[^\n]*\n" % re.escape(pat), txt, re.M)
assert m, "kernel not found"
body = txt[m.end():txt.index(".Lfunc_end", m.end())]
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in (x.strip() for x in body.split("\n")):
    if not l or l.startswith(";"):
        continue
    if re.match(r"^\.LBB\d+_\d+:", l):
        cur = l.split(":")[0]
        blocks[cur] = []
        continue
    if l.startswith("."):
        continue
    blocks[cur].append(l.split()[0])
print(m.group(1)[:110])
for b, ins in blocks.items():
    if len(ins) < minsz:
        continue
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    mad = c.get("v_mad_u64_u32", 0)
    print("%-10s %5d instr, %5d VALU, %5d mad64 (%.3f of VALU)" % (b, len(ins), valu, mad, mad / max(valu, 1)))
    print("           " + ", ".join("%s %d" % kv for kv in c.most_common(16)))
