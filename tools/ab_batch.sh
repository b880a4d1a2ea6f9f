#!/bin/bash
# Dev tool: A/B of two builds on the batched-filter workloads:  tools/ab_batch.sh libA.so libB.so
for lib in "$1" "$2" "$1" "$2"; do
  echo "== $lib"; BAYESSSM_AMD_LIB=$lib timeout -k 10 300 python tools/diag_batch_overhead.py 2>&1 | grep -v amdgpu.ids | grep "T=100"
  BAYESSSM_AMD_LIB=$lib timeout -k 10 300 python tools/bench_batch.py 200 2>&1 | grep -v amdgpu.ids | grep "F=  512"
done
