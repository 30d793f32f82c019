#!/bin/bash
# small-proof sweep: Groth16 at 2^10..2^16 (queued / isolated / phases) and Marlin at 2^10..2^14
cd /root/repo
out=gpurun_out/${1:-small_proofs}.jsonl; rm -f $out
for L in 10 12 14 16; do
  python bench.py --log-constraints $L --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-micro 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'groth16_log': $L, 'ms_queued': d['ms_per_step'], 'median': d['ms_per_step_median'], 'isolated_ms': d['isolated_proof_ms'], 'match': d.get('proof_matches_prediction'), 'phases': d['phases_ms_per_proof']}))" >> $out
done
for L in 10 12 14; do
  python bench.py --marlin --log-constraints $L --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'marlin_log': $L, 'ms': d['ms_per_step'], 'verifier': d.get('oracle_verifier_accepts')}))" >> $out
done
cat $out
