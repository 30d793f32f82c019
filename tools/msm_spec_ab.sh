#!/bin/bash
# A/B of the MSMs started ahead (ZK_MSM_SPEC=0/1) on the plain composed trait path at every size, and on 3 parties sharing one GPU
# -> profiles/r6_msm_spec_ab.jsonl
cd "$GRAFT_REPO_ROOT/examples/_bin"
for lg in 10 12 14 16 17 18 20; do
  for sp in 0 1; do
    ZK_MSM_SPEC=$sp ./host_trait_groth16 $lg 9 cache strided | python3 -c "
import sys,json,statistics
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
pr=[r for r in rows if 'ms' in r]
st=pr[-3:]
med=lambda k: round(statistics.median(p['ms'][k] for p in st),3)
print(json.dumps({'log_d': $lg, 'spec': $sp, 'lib': med('lib'), 'msm_a': med('msm_a'), 'msm_b1': med('msm_b1'), 'msm_b2': med('msm_b2'), 'same_bytes': len(set(p['proof'] for p in pr))==1, 'all_lib': [round(p['ms']['lib'],1) for p in pr]}))"
  done
done
for sp in 0 1; do ZK_MSM_SPEC=$sp ./host_trait_collab_groth16 18 8 3 additive tagfirst verify sync2 | python3 -c "
import sys,json,statistics
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
pr=[r for r in rows if 'ms' in r]
print(json.dumps({'collab_p3_2p18_spec': $sp, 'lib': [round(p['ms']['lib'],1) for p in pr], 'max': [round(p['ms_max_lib'],1) for p in pr]}))"; done
