cd /root/repo
for L in 15 16 17 18 19; do
  python bench.py --log-constraints $L --steps 20 --warmup 4 --no-cpu-baseline --no-extras --no-micro 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'groth16_log': $L, 'ms_queued': d['ms_per_step'], 'median': d['ms_per_step_median'], 'isolated_ms': d['isolated_proof_ms'], 'match': d.get('proof_matches_prediction'), 'phases': d['phases_ms_per_proof']}))"
done
