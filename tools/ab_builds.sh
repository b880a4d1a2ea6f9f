#!/bin/bash
# Dev tool: A/B of two builds of the library on one GPU box:  tools/ab_builds.sh libA.so libB.so
for r in 1 2 3; do
  for lib in "$1" "$2"; do
    BAYESSSM_AMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmmh --no-batch --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['sweep']['us_per_observation'],2), [round(v['avg_us'],2) for v in list(d['kernels'].values())[:6]])"
  done
done
