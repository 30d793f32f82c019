#!/usr/bin/env python3
"""Aggregate two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace, csv output) into
per-kernel HBM bytes per launch: profiles/rNN_pmc_traffic.json.

Units and correction as in /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in KiB;
on gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane loads, so fetched bytes = 2 x FETCH_SIZE (calibrated on
k_vec_op: 2 x 32 MiB read -> FETCH_SIZE 32 MiB).  hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# what the accumulate kernels are made of: a traffic file belongs to ONE state of these sources (bench.py refuses another)
KERNEL_SOURCES = ["msm.hip", "msm_sort.hip", "msm_g2pair.hip", "msm_reduce.cuh", "msm_digits.cuh", "fixed_base.hip", "ec.cuh", "ec_dual.cuh", "fp29.cuh"]


def sources_sha256():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "zk-mpc_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    if not m:
        return name[:40]
    base, targ = m.group(1), m.group(2) or ""
    targ = targ.replace("zk::FqField", "G1").replace("zk::Fq2Field", "G2")
    return base + targ


def load(dirname, counter):
    acc = defaultdict(lambda: [0.0, 0])
    files = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)
    if not files:
        sys.exit("no counter_collection.csv under " + dirname)
    per_dispatch = defaultdict(float)
    names = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            key = (f, row["Dispatch_Id"])
            per_dispatch[key] += float(row["Counter_Value"])      # one row per XCD / instance: sum them
            names[key] = row["Kernel_Name"]
    for key, v in per_dispatch.items():
        a = acc[short(names[key])]
        a[0] += v
        a[1] += 1
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        fb = f / nf * 1024 if nf else 0.0
        wb = w / nw * 1024 if nw else 0.0
        kernels[k] = {"launches_sampled": nf or nw, "FETCH_SIZE_bytes": int(fb), "fetch_bytes_corrected": int(2 * fb),
                      "write_bytes": int(wb), "hbm_bytes": int(2 * fb + wb)}
    json.dump({"kernel_sources_sha256": sources_sha256(), "kernel_sources": KERNEL_SOURCES,
               "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes). " + note +
               " Values are per-launch averages in bytes; fetch_bytes_corrected = 2 x FETCH_SIZE (gfx950 correction, "
               "MI355X_MICROARCH.md HBM section); hbm_bytes = fetch_bytes_corrected + write_bytes.", "kernels": kernels},
              open(out, "w"), indent=1)
    for k in ("k_accum<G1>", "k_accum<G2>", "k_scatter", "k_vec_op<0>", "k_ntt_pass"):
        if k in kernels:
            print(k, kernels[k])


if __name__ == "__main__":
    main()
