#!/bin/bash
# rocprofv3 passes on the final build: kernel trace + stats of the bench's C2 workload, then FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (no trace domains next to --pmc).  usage: bash tools/gpu/profile.sh TAG
TAG=${1:-p}
OUT=gpurun_out/prof_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-pmmh --no-batch --no-configs > $OUT/kt.log 2>&1 || { tail -5 $OUT/kt.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 bench.py --steps 1 --warmup 0 --T 60 --no-cpu-baseline --no-profile --no-pmmh --no-batch --no-configs > $OUT/f.log 2>&1 || { tail -5 $OUT/f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 bench.py --steps 1 --warmup 0 --T 60 --no-cpu-baseline --no-profile --no-pmmh --no-batch --no-configs > $OUT/w.log 2>&1 || { tail -5 $OUT/w.log; exit 1; }
# the multi-launch path of the same build (option fused = 0): kernel trace only
BAYESSSM_AMD_FUSED=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_multi -o kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-pmmh --no-batch --no-configs > $OUT/kt_multi.log 2>&1 || { tail -5 $OUT/kt_multi.log; exit 1; }
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
ls -R $OUT | grep -c csv
f=$(find $OUT -name "*kernel_stats.csv" | sort | tail -1)
if [ -n "$f" ]; then cut -c1-160 "$f" | sed -n 1,8p; fi
