#!/usr/bin/env python3
"""How much of a window of a rocprofv3 --kernel-trace CSV the GPU spends under-filled: at every instant the threads of all
kernels in flight are summed; an instant counts as "filled" when that sum reaches `fill` threads (default 256 CUs x 512).
Prints the filled / under-filled / idle split and the kernels that own the under-filled time (latency-bound launches: scans'
upper levels, single-block reductions, host synchronisation).

usage: trace_underfill.py kernel_trace.csv [window_ms=90] [fill_threads=131072] [--period=KERNEL[:idx]]"""
import csv, re, sys, collections

f = sys.argv[1]
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
win = float(pos[1]) if len(pos) > 1 else 90.0
fill = int(pos[2]) if len(pos) > 2 else 131072
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+|rocprim\w*|\w+)(<|\()", r["Kernel_Name"])
    g = int(r["Grid_Size_X"]) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30], g))
period = [a for a in sys.argv if a.startswith("--period=")]
if period:        # --period=KERNEL[:idx]: the window runs from the idx-th launch of KERNEL (default -3) to the next one
    name, _, idx = period[0][9:].partition(":")
    starts = sorted(s for s, _, k, _ in rows if k == name)
    i = int(idx) if idx else -3
    lo, hi = starts[i], starts[i + 1]
    rows = [r for r in rows if lo <= r[0] < hi]
else:
    end = max(e for _, e, _, _ in rows)
    rows = [r for r in rows if r[0] >= end - win * 1e6]
ev = []
for i, (s, e, k, g) in enumerate(rows):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort()
live = set()
t_prev = ev[0][0]
filled = under = idle = 0
owners = collections.Counter()
for t, d, i in ev:
    dt = t - t_prev
    if dt > 0:
        threads = sum(rows[j][3] for j in live)
        if not live:
            idle += dt
        elif threads >= fill:
            filled += dt
        else:
            under += dt
            for j in live:
                owners[rows[j][2] + " grid=%d" % rows[j][3]] += dt / len(live)
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
    t_prev = t
span = filled + under + idle
print("window %.2f ms: filled %.2f ms (%.1f %%), under-filled %.2f ms (%.1f %%), idle %.2f ms (%.1f %%)" % (
    span / 1e6, filled / 1e6, 100 * filled / span, under / 1e6, 100 * under / span, idle / 1e6, 100 * idle / span))
print("under-filled time by kernel:")
for k, v in owners.most_common(16):
    print("  %-44s %8.2f ms" % (k, v / 1e6))
