#!/usr/bin/env python3
"""Print the kernel timeline of the last proof in a rocprofv3 --kernel-trace CSV (start, end, duration, queue, stream)."""
import csv, glob, re, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/prof*/*/*_kernel_trace.csv'))[-1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    n = r['Kernel_Name']; m = re.search(r'(k_\w+)(<[^>]*>)?', n)
    r['k'] = (m.group(1) + ('.g2' if 'Fq2' in n else '.g1' if 'FqField' in n else '')) if m else n[:30]
acc = [i for i, r in enumerate(rows) if r['k'] == 'k_accum.g2']
t0 = rows[acc[-1]]['s'] - 4_000_000
sel = sorted([r for r in rows if r['s'] >= t0], key=lambda r: r['s'])
base = sel[0]['s']
for r in sel:
    if (r['e'] - r['s']) / 1e3 > thr or 'accum' in r['k']:
        print(f"{(r['s']-base)/1e6:8.3f} {(r['e']-base)/1e6:8.3f} dur={(r['e']-r['s'])/1e3:9.1f}us q={r['Queue_Id']} st={r['Stream_Id']} {r['k']} grid={r['Grid_Size_X']}")
print("span ms", (max(r['e'] for r in sel) - base) / 1e6)
