#!/usr/bin/env python3
"""Instruction-class histogram of one kernel from hipcc's device assembly (static count).

  hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S zk-mpc_amd/csrc/msm.hip -o /tmp/msm.s
  python tools/isa_hist.py /tmp/msm.s k_accum [--block largest|all] [--rates profiles/r1_ubench_int.txt]

Prints the mnemonic histogram of the whole kernel and of its largest basic block (for k_accum<G1> that block is the
mixed addition's main path: one trip per bucket entry), and prices the block with the measured issue rates of
tools/ubench_int.hip: classes are "half" (~32-37 T lane-ops/s: v_mad_u64_u32, v_mul_lo_u32, v_add_co_u32, v_cndmask,
v_alignbit, v_lshrrev_b64, ...) and "full" (~67 T: v_add_u32, v_sub_u32, v_and_b32, 32-bit shifts, v_xor, v_or, v_mov).
The mix-weighted ceiling is  mads / (sum_i count_i / rate_i)  relative to the v_mad_u64_u32 rate.
"""
from __future__ import annotations

import collections
import json
import re
import sys

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_lshlrev_b32",
        "v_ashrrev_i32", "v_mov_b32", "v_not_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_max_u32", "v_min_u32",
        "v_max_i32", "v_min_i32"}


def kernel_body(path: str, name: str):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l):
            start = i
            break
    if start is None:
        raise SystemExit("kernel %s not found" % name)
    body = []
    for l in lines[start + 1:]:
        if l.startswith("\t.section") or l.startswith(".Lfunc_end") or re.match(r"^_Z\w+:", l):
            break
        body.append(l)
    return lines[start], body


def blocks(body):
    cur, out = [], []
    for l in body:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s):
                if cur:
                    out.append(cur)
                cur = []
            continue
        op = s.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        cur.append(op)
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
            out.append(cur)
            cur = []
    if cur:
        out.append(cur)
    return out


def classify(op: str) -> str:
    if op.startswith("v_mad_u64_u32"):
        return "mad64"
    if op.startswith("v_"):
        base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
        return "valu_full" if base in FULL else "valu_half"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def summarise(ops, title):
    h = collections.Counter(ops)
    cls = collections.Counter()
    for op, c in h.items():
        cls[classify(op)] += c
    valu = cls["mad64"] + cls["valu_half"] + cls["valu_full"]
    print("== %s: %d instructions, %d VALU (%d v_mad_u64_u32 = %.1f %%, %d other half-rate, %d full-rate), %d SALU, %d VMEM, %d LDS"
          % (title, len(ops), valu, cls["mad64"], 100.0 * cls["mad64"] / max(valu, 1), cls["valu_half"], cls["valu_full"],
             cls["salu"], cls["vmem"], cls["lds"]))
    for op, c in h.most_common(28):
        print("   %-24s %6d  %s" % (op, c, classify(op)))
    # price: half-rate class at the mad rate (1 slot), full-rate at half a slot
    slots = cls["mad64"] + cls["valu_half"] + 0.5 * cls["valu_full"]
    if cls["mad64"]:
        print("   issue slots (half-rate = 1, full-rate = 0.5): %.0f -> mix-weighted ceiling of the mad fraction = %.3f"
              % (slots, cls["mad64"] / slots))
    return {"instructions": len(ops), "valu": valu, "mad64": cls["mad64"], "valu_half_other": cls["valu_half"],
            "valu_full": cls["valu_full"], "salu": cls["salu"], "vmem": cls["vmem"], "lds": cls["lds"],
            "mix_ceiling": (cls["mad64"] / slots) if slots else None, "top": dict(h.most_common(40))}


def main():
    path, name = sys.argv[1], sys.argv[2]
    head, body = kernel_body(path, name)
    bl = blocks(body)
    allops = [op for b in bl for op in b]
    out = {"kernel": head.rstrip(":"), "whole": summarise(allops, "whole kernel (static)")}
    big = max(bl, key=len)
    out["largest_block"] = summarise(big, "largest basic block")
    if "--main" in sys.argv:          # --main 19,20: the basic blocks (by index, see --list) that make up the hot path
        idx = [int(x) for x in sys.argv[sys.argv.index("--main") + 1].split(",")]
        out["main_path"] = summarise([op for i in idx for op in bl[i]], "main path = blocks %s" % idx)
        out["main_path"]["blocks"] = idx
    if "--list" in sys.argv:
        for i, b in enumerate(bl):
            c = collections.Counter(classify(o) for o in b)
            print("   block %3d: %5d instructions %s ends with %s" % (i, len(b), dict(c), b[-1]))
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
