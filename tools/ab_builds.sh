#!/bin/bash
# Dev tool: A/B of several builds of the library on ONE GPU box:  tools/ab_builds.sh libA.so libB.so [libC.so ...]
# (alternating runs, three rounds: box-to-box variance is ~2 %, run-to-run on one box ~0.3 %)
for r in 1 2 3; do
  for lib in "$@"; do
    BAYESSSM_AMD_LIB=$lib timeout -k 10 300 python bench.py --steps ${AB_STEPS:-10} --warmup 2 --no-cpu-baseline --no-pmmh --no-batch --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s' % '$lib'.split('/')[-1], round(d['sweep']['us_per_observation'],2), d.get('loglike_last_run'), [round(v['avg_us'],2) for v in list(d['kernels'].values())[:6]])"
  done
done
