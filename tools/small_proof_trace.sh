cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/small
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/small -o m --output-format csv -- python3 bench.py --log-constraints ${1:-10} --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-predict --no-micro > gpurun_out/small.log 2>&1
python3 - <<'PY'
import csv,re,glob
rows=[]
for r in csv.DictReader(open(glob.glob('gpurun_out/small/**/m_kernel_trace.csv',recursive=True)[0])):
    m=re.search(r"(k_\w+|rocprim\w*|\w+)(<|\()", r["Kernel_Name"])
    nm=m.group(1) if m else r["Kernel_Name"][:30]
    if 'Fq2' in r["Kernel_Name"] or 'g2pair' in r["Kernel_Name"]: nm+='.g2'
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),nm, int(r['Grid_Size_X']), r['Queue_Id']))
try:
    for r in csv.DictReader(open(glob.glob('gpurun_out/small/**/m_memory_copy_trace.csv',recursive=True)[0])):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction",""), 0, '-'))
except Exception as e: print(e)
rows.sort()
acc=[i for i,r in enumerate(rows) if r[2].startswith('k_accum') and r[2].endswith('.g2')]
i0=acc[8]
t0=rows[i0][0]-600_000
out=open('gpurun_out/r5_small_timeline.txt','w')
for s,e,k,g,q in rows:
    if t0 <= s <= t0+3_000_000:
        out.write("%8.1f %8.1f %7.1f us q=%s %-22s grid=%d\n"%((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3,q,k,g))
PY
tail -2 gpurun_out/small.log | cut -c 1-400
