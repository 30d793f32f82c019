#!/usr/bin/env python3
"""Print the kernel timeline of one Groth16 proof from a rocprofv3 --kernel-trace CSV (start, end, duration, queue).

usage: trace_timeline.py [kernel_trace.csv] [min_us=120] [which=-8]
`which` indexes the k_accum<G2> launches (one per proof): bench.py ends with 6 standalone G2 MSMs, so -8 is a proof
from the timed region."""
import csv, glob, re, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/prof*/*/*_kernel_trace.csv'))[-1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
which = int(sys.argv[3]) if len(sys.argv) > 3 else -8
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    n = r['Kernel_Name']; m = re.search(r'(k_\w+)(<[^>]*>)?', n)
    r['k'] = (m.group(1) + ('.g2' if 'Fq2' in n else '.g1' if 'FqField' in n else '')) if m else n[:30]
    if r['k'].endswith('_g2pair'): r['k'] = r['k'][:-7] + '.g2'      # the lane-pair G2 kernels of msm_g2pair.hip
    if 'RedG2Pair' in n: r['k'] += '.g2'                             # msm_reduce.cuh kernels by point policy
    elif 'RedG1' in n: r['k'] += '.g1'
    if 'rocprim' in n: r['k'] = 'rocprim.radix_sort'                 # onesweep histogram / digit passes (msm_sort.hip)
acc = [i for i, r in enumerate(rows) if r['k'] == 'k_accum.g2']
t0 = rows[acc[which]]['s'] - 5_000_000
t1 = rows[acc[which]]['s'] + 30_000_000
sel = sorted([r for r in rows if t0 <= r['s'] <= t1], key=lambda r: r['s'])
base = sel[0]['s']
for r in sel:
    d = (r['e'] - r['s']) / 1e3
    if d > thr or 'accum' in r['k']:
        print(f"{(r['s']-base)/1e6:8.3f} {(r['e']-base)/1e6:8.3f} dur={d:9.1f}us q={r['Queue_Id']} {r['k']}")
