#!/bin/bash
# GPU call: the randomized stress tools on the current build, different seeds per call.  usage: bash tools/gpu/stress.sh TAG SEED [SECONDS]
TAG=${1:-s}; SEED=${2:-1}; SEC=${3:-150}
OUT=gpurun_out/stress_${TAG}.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout -k 10 $((SEC + 120)) "$@" 2>/dev/null | tail -2 >> $OUT || echo "   (exit $?)" >> $OUT; }
run python tools/stress_resample.py $SEED $SEC
run python tools/stress_oracle.py $((SEED + 1)) $SEC
echo "== the same with the one-launch-per-observation path forced at every size (BAYESSSM_AMD_FUSED=2)" >> $OUT
BAYESSSM_AMD_FUSED=2 timeout -k 10 $((SEC + 120)) python tools/stress_oracle.py $((SEED + 2)) $SEC 2>/dev/null | tail -2 >> $OUT
run python tools/stress_batch.py $((SEED + 3)) $SEC
run python tools/stress_large.py $((SEED + 4)) $SEC
run python tools/stress_fused.py $((SEED + 5)) $SEC
cat $OUT
