cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/msmall
rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/msmall -o m --output-format csv -- python3 bench.py --marlin --log-constraints ${1:-10} --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/msmall.log 2>&1
python3 - <<'PY'
import csv,re,glob
rows=[]
for r in csv.DictReader(open(glob.glob('gpurun_out/msmall/**/m_kernel_trace.csv',recursive=True)[0])):
    m=re.search(r"(k_\w+|\w+)(<|\()", r["Kernel_Name"])
    nm=m.group(1) if m else r["Kernel_Name"][:30]
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),nm, r['Queue_Id']))
for r in csv.DictReader(open(glob.glob('gpurun_out/msmall/**/m_memory_copy_trace.csv',recursive=True)[0])):
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY_"+r.get("Direction","")[12:], '-'))
rows.sort()
# a proof = from one k_spmv burst to the next: find starts of "k_fr_random"/first kernel after long gap... use k_spmv occurrences
marks=[i for i,r in enumerate(rows) if r[2]=='k_spmv']
starts=[marks[0]]
for a,b in zip(marks,marks[1:]):
    if rows[b][0]-rows[a][0] > 2_000_000: starts.append(b)
i0,i1=starts[-3],starts[-2]
t0=rows[i0][0]
out=open('gpurun_out/r5_marlin_small_timeline.txt','w')
prev_end=t0; busy=0; n=0
for s,e,k,q in rows[i0:i1]:
    gap=(s-prev_end)/1e3
    out.write("%9.1f %8.1f us gap %7.1f q=%s %s\n"%((s-t0)/1e3,(e-s)/1e3,gap if gap>0 else 0,q,k))
    if e>prev_end: busy+= (e-max(s,prev_end))/1e3; prev_end=e
    n+=1
out.write("launches %d busy %.1f us window %.1f us\n"%(n,busy,(rows[i1][0]-t0)/1e3))
print("launches %d busy %.1f us window %.1f us"%(n,busy,(rows[i1][0]-t0)/1e3))
PY
