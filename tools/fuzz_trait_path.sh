#!/bin/bash
# the two randomised walks of tests/test_gpu_trait_path.py (the table cache; the MSMs started ahead) under other seeds
# usage (on an MI355X, from the repo root): tools/fuzz_trait_path.sh [first_seed=100] [count=20]
first=${1:-100}; count=${2:-20}; bad=0
for ((s = first; s < first + count; s++)); do
  if ZK_FUZZ_SEED=$s python3 -m pytest tests/test_gpu_trait_path.py -m gpu -x -q -k "fuzz" > /tmp/fuzz_$s.log 2>&1; then echo "seed $s ok: $(tail -1 /tmp/fuzz_$s.log)"; else echo "seed $s FAILED"; tail -30 /tmp/fuzz_$s.log; bad=$((bad + 1)); fi
done
echo "FUZZ trait path: $count seeds, $bad failed"
