#!/bin/bash
cd /root/repo
rm -f gpurun_out/r5_trait_small.jsonl
for args in "10 8 cache packed" "12 8 cache packed" "14 8 cache packed" "16 8 cache packed" "20 7 cache packed"; do
  echo "== $args" >> gpurun_out/r5_trait_small.jsonl
  examples/_bin/host_trait_groth16 $args >> gpurun_out/r5_trait_small.jsonl 2>&1
done
