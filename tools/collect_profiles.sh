#!/bin/bash
# round-6 evidence: kernel stats of the default bench command, HBM traffic (two PMC passes), VALU instruction counts; the composed
# trait paths (plain and collaborative) call by call
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r6prof; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-micro --no-predict"
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- $B > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- $B > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/valu -o v --output-format csv -- $B > $O/valu.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write gpurun_out/r6_pmc_traffic.json "bench.py --steps 10 --warmup 3 (host-witness queue of 4, next one announced), 2^20 - 2 constraints; collected in round 6 after the last change to the kernel sources listed." | head -5
python3 tools/pmc_valu.py gpurun_out/r6_pmc_valu.json $O/valu 2>&1 | head -5
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r6_kernel_stats.csv; head -12 gpurun_out/r6_kernel_stats.csv
du -sh $O; rm -rf $O/fetch $O/write $O/valu $O/stats
# the composed trait paths, every call
( cd examples/_bin
  ./host_trait_groth16 20 9
  ./host_trait_groth16 20 9 cache strided
  ./host_trait_groth16 20 9 cache strided trust
  ./host_trait_groth16 20 3 nocache
  for lg in 10 12 14 16; do ./host_trait_groth16 $lg 7 cache strided; done ) > gpurun_out/r6_trait_path.jsonl 2>&1
( cd examples/_bin
  ./host_trait_collab_groth16 20 8 1 additive tagfirst verify sync2
  ./host_trait_collab_groth16 18 8 3 additive tagfirst verify sync2
  ./host_trait_collab_groth16 20 8 3 additive tagfirst verify sync2
  ./host_trait_collab_groth16 18 8 2 spdz tagfirst verify sync2
  ./host_trait_collab_groth16 18 8 3 additive tagfirst verify
  for lg in 10 12 14 16; do ./host_trait_collab_groth16 $lg 7 3 additive tagfirst verify sync2; done ) > gpurun_out/r6_trait_path_collab.jsonl 2>&1
