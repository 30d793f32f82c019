#!/usr/bin/env python3
"""Idle time of the GPU inside a window of a rocprofv3 --kernel-trace CSV: union of the kernel intervals against the span, and the
largest gaps with the kernels on either side (host synchronisation points show up here).

usage: trace_gaps.py kernel_trace.csv [window_ms=90] [min_gap_us=30] [--period=KERNEL[:idx]]\nThe window ends at the last kernel of the trace, or is one period of a once-per-job kernel."""
import csv, re, sys

f = sys.argv[1]
pos = [a for a in sys.argv[1:] if not a.startswith("--")]
win = float(pos[1]) if len(pos) > 1 else 90.0
min_gap = float(pos[2]) if len(pos) > 2 else 30.0
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_\w+|rocprim\w*|\w+)(<|\()", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else r["Kernel_Name"][:30]))
rows.sort()
period = [a for a in sys.argv if a.startswith("--period=")]
if period:        # --period=KERNEL[:idx]: the window runs from the idx-th launch of KERNEL (default -3) to the next one
    name, _, idx = period[0][9:].partition(":")
    starts = sorted(s for s, _, k in rows if k == name)
    i = int(idx) if idx else -3
    lo, hi = starts[i], starts[i + 1]
    rows = [r for r in rows if lo <= r[0] < hi]
    end = hi
else:
    end = max(e for _, e, _ in rows)
    rows = [r for r in rows if r[0] >= end - win * 1e6]
span = end - rows[0][0]
busy, cur_s, cur_e, gaps, last_name = 0, rows[0][0], rows[0][1], [], rows[0][2]
by_kernel = {}
for s, e, k in rows:
    by_kernel[k] = by_kernel.get(k, 0) + (e - s)
    if s > cur_e:
        gaps.append((s - cur_e, cur_e - rows[0][0], last_name, k))
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e:
        last_name = k
busy += cur_e - cur_s
print("window %.2f ms: busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps" % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
print("kernel time by name (sum of durations, overlapping kernels counted each):")
for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])[:14]:
    print("  %-34s %8.2f ms" % (k, v / 1e6))
print("gaps >= %.0f us:" % min_gap)
for g, at, a, b in sorted(gaps, reverse=True):
    if g / 1e3 >= min_gap:
        print("  %8.1f us at +%7.2f ms   after %-28s before %s" % (g / 1e3, at / 1e6, a, b))
