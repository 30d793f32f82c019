#!/bin/bash
# Round-end evidence, run on the GPU box from the repo root:  bash tools/collect_profiles.sh <tag>
# Writes under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -u
T=${1:-r1}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $O/${T}_bench_n1.json 2> $O/${T}_bench_n1.err
rocprofv3 --kernel-trace --stats -d $O/${T}_trace -o t --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-predict > $O/${T}_trace.log 2>&1
python3 tools/trace_timeline.py $O/${T}_trace/t_kernel_trace.csv 60 7 > $O/${T}_timeline.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${T}_pmc_fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-predict > $O/${T}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${T}_pmc_write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-predict > $O/${T}_pmc_write.log 2>&1
python3 tools/pmc_traffic.py $O/${T}_pmc_fetch $O/${T}_pmc_write $O/${T}_pmc_traffic.json "bench.py --steps 3 --warmup 1, 2^20 Groth16 proofs." > $O/${T}_pmc_traffic.txt 2>&1
for p in "a:SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_WAVES" "b:SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY" "c:MeanOccupancyPerCU" "d:SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM"; do
  tag=${p%%:*}; ctr=${p#*:}
  rocprofv3 --kernel-trace --pmc $ctr -d $O/${T}_pmc_$tag -o v --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-predict --no-hint > $O/${T}_pmc_$tag.log 2>&1
done
python3 tools/pmc_valu.py $O/${T}_pmc_valu.json $O/${T}_pmc_a $O/${T}_pmc_b $O/${T}_pmc_c $O/${T}_pmc_d > $O/${T}_pmc_valu.txt 2>&1
python3 tools/bench_marlin.py --logs 16,18,20,21,22 > $O/${T}_marlin_bench.jsonl 2> $O/${T}_marlin_bench.err
rocprofv3 --kernel-trace --stats -d $O/${T}_marlin_trace -o m --output-format csv -- python3 tools/bench_marlin.py --logs 20 --reps 3 > $O/${T}_marlin_trace.log 2>&1
python3 tools/bench_msm.py 24 > $O/${T}_micro_msm_ntt.json 2> $O/${T}_micro.err
python3 tools/bench_she.py > $O/${T}_she_bench.jsonl 2> $O/${T}_she.err
python3 tools/bench_ntt.py 24 20 10 > $O/${T}_ntt_bench.jsonl 2> $O/${T}_ntt.err
python3 tools/bench_vec.py > $O/${T}_vec_bench.jsonl 2> $O/${T}_vec.err
for L in 16 18 20; do python3 bench.py --marlin --log-constraints $L --steps 4 --warmup 1 2>> $O/${T}_marlin_prove.err; done > $O/${T}_marlin_prove.jsonl
rocprofv3 --kernel-trace -d $O/${T}_marlin_prove_trace -o m --output-format csv -- python3 bench.py --marlin --log-constraints 20 --steps 3 --warmup 1 > $O/${T}_marlin_prove_trace.log 2>&1
python3 tools/trace_underfill.py $O/${T}_marlin_prove_trace/m_kernel_trace.csv --period=k_denoms_k:-4 > $O/${T}_marlin_underfill.txt 2>&1
python3 tools/trace_gaps.py $O/${T}_marlin_prove_trace/m_kernel_trace.csv 0 100 --period=k_denoms_k:-4 >> $O/${T}_marlin_underfill.txt 2>&1
for c in "SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS"; do d=$O/${T}_nttpmc_$(echo $c | cut -c4-16 | tr ' ' '_'); rocprofv3 --kernel-trace --pmc $c -d $d -o p --output-format csv -- python3 tools/bench_ntt.py 20 2 20 > $d.log 2>&1; done
[ -x tools/_bin/ubench_chain ] && tools/_bin/ubench_chain > $O/${T}_ubench_chain.txt 2>&1
ls $O | grep "^${T}_"
