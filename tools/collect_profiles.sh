#!/bin/bash
# round-5 evidence: kernel stats of the default bench command, HBM traffic (two PMC passes), VALU instruction counts
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5prof; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-micro --no-predict"
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- $B > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- $B > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/valu -o v --output-format csv -- $B > $O/valu.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write gpurun_out/r5_pmc_traffic.json "bench.py --steps 10 --warmup 3 (queue of 4, hint), 2^20 - 2 constraints; collected in round 5 after the last change to the kernel sources listed." | head -5
python3 tools/pmc_valu.py gpurun_out/r5_pmc_valu.json $O/valu 2>&1 | head -5
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r5_kernel_stats.csv; head -12 gpurun_out/r5_kernel_stats.csv
du -sh $O; rm -rf $O/fetch $O/write $O/valu $O/stats
