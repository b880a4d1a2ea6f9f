#!/bin/bash
# Dev tool: FETCH_SIZE / WRITE_SIZE per kernel at C5's size (N = 2^22, stratified), separate --pmc passes.  usage: bash tools/gpu/pmc_c5.sh TAG
TAG=${1:-c5}
OUT=gpurun_out/pmcc5_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/$ctr -o c -- python3 bench.py --steps 1 --warmup 0 --T 40 --particles 4194304 --resample-fn stratified --no-cpu-baseline --no-profile --no-pmmh --no-batch --no-configs > $OUT/$ctr.log 2>&1 || { tail -5 $OUT/$ctr.log; exit 1; }
  python3 - "$OUT/$ctr/c_counter_collection.csv" $ctr <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"].split("(")[0][-44:]
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
    print("%-12s %-46s %10.1f KiB per launch x %d%s" % (sys.argv[2], k, v / n, n, "   (x 2 on gfx950)" if sys.argv[2] == "FETCH_SIZE" else ""))
PY
done
