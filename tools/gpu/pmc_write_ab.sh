#!/bin/bash
# Dev tool: WRITE_SIZE per kernel for several builds / paths (one --pmc pass each, no trace domains).
# usage: bash tools/gpu/pmc_write_ab.sh TAG lib1.so[:fused] ...
TAG=$1; shift
OUT=gpurun_out/pmcw_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
for spec in "$@"; do
  lib=${spec%%:*}; fused=${spec##*:}; [ "$fused" = "$spec" ] && fused=1
  name=$(basename $lib .so)_f$fused
  BAYESSSM_AMD_LIB=$PWD/$lib BAYESSSM_AMD_FUSED=$fused rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$name -o w -- python3 bench.py --steps 1 --warmup 0 --T 60 --no-cpu-baseline --no-profile --no-pmmh --no-batch --no-configs > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
  python3 - "$OUT/$name/w_counter_collection.csv" "$name" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "WRITE_SIZE": continue
    k = r["Kernel_Name"].split("(")[0][-40:]
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:6]:
    print("%-24s %-42s WRITE_SIZE %10.1f KiB per launch x %d" % (sys.argv[2], k, v / n, n))
PY
done
