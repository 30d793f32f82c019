#!/usr/bin/env python3
"""Aggregate rocprofv3 PMC passes on the instruction mix and occupancy into per-kernel, per-launch averages:
   profiles/rNN_pmc_valu.json.  usage: pmc_valu.py <out.json> <pass_dir> [<pass_dir> ...]
Each pass is `rocprofv3 --kernel-trace --pmc <counters> --output-format csv` of the same bench.py command (separate passes:
the counters do not all fit one).  Values are summed over the instances (XCDs / SEs) of a dispatch, then averaged over the
launches of a kernel.  Derived: valu_wave_insts_per_s = SQ_INSTS_VALU / kernel duration (from the same pass's kernel trace),
to be read against the measured integer issue peak of 32.2 T lane-ops/s = 0.503 T wave-instructions/s (tools/ubench_int.hip)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_traffic as PT          # the hash of the kernel sources a counter file belongs to (bench.py refuses a file from other sources)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if "rocprim" in name:
        return "rocprim::radix_sort (onesweep)"
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    if not m:
        return name[:40]
    return m.group(1) + (m.group(2) or "").replace("zk::FqField", "G1").replace("zk::Fq2Field", "G2")


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum, launches]
    dur = defaultdict(lambda: [0.0, 0])
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            per = defaultdict(float)
            names = {}
            for row in csv.DictReader(open(f)):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
            for (disp, cname), v in per.items():
                a = acc[short(names[disp])][cname]
                a[0] += v
                a[1] += 1
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                a = dur[short(row["Kernel_Name"])]
                a[0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
                a[1] += 1
    res = {}
    for k, cs in acc.items():
        e = {c: v[0] / v[1] for c, v in cs.items() if v[1]}
        e["launches_sampled"] = max(v[1] for v in cs.values())
        if dur[k][1]:
            e["avg_duration_ms_under_pmc"] = dur[k][0] / dur[k][1] * 1e3
            if "SQ_INSTS_VALU" in e:
                e["valu_wave_insts_per_s"] = e["SQ_INSTS_VALU"] / (dur[k][0] / dur[k][1])
                e["frac_of_int_issue_peak"] = e["valu_wave_insts_per_s"] / (32.16e12 / 64)
        res[k] = e
    json.dump({"_note": "rocprofv3 --pmc passes (separate runs) over bench.py --steps 3 --warmup 1 --no-hint; per-launch averages, "
                        "counter values summed over XCDs / shader engines.  Kernels run serialised under counter collection, so "
                        "durations are those of a kernel alone on the chip.", "kernel_sources_sha256": PT.sources_sha256(),
               "kernel_sources": PT.KERNEL_SOURCES, "kernels": res}, open(out, "w"), indent=1)
    for k in ("k_accum<G1>", "k_accum_g2pair<2>", "k_ntt_pass<0>", "k_reduce<G1>", "k_reduce_g2pair", "rocprim::radix_sort (onesweep)"):
        if k in res:
            print(k, {a: (round(b, 4) if b < 10 else round(b)) for a, b in res[k].items()})


if __name__ == "__main__":
    main()
