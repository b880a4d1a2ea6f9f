#!/bin/bash
# Dev tool: store flavours at C5's size: WRITE_SIZE per kernel and us per observation for several builds.  usage: bash tools/gpu/c5_store_ab.sh lib1.so lib2.so ...
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  OUT=gpurun_out/c5st_$name
  mkdir -p $OUT
  BAYESSSM_AMD_LIB=$PWD/$lib rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -o c -- python3 bench.py --steps 1 --warmup 0 --T 40 --particles 4194304 --resample-fn stratified --no-cpu-baseline --no-profile --no-pmmh --no-batch --no-configs > $OUT/w.log 2>&1 || { tail -3 $OUT/w.log; continue; }
  python3 - "$OUT/w/c_counter_collection.csv" $name <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "WRITE_SIZE": continue
    k = r["Kernel_Name"].split("(")[0][-40:]
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
print(sys.argv[2], "WRITE_SIZE MiB per launch:", {k[-34:]: round(v / n / 1024, 1) for k, (v, n) in acc.items() if n >= 40})
PY
  BAYESSSM_AMD_LIB=$PWD/$lib timeout -k 10 200 python tools/ab_c5.py 2>/dev/null | head -2 | sed "s/^/$name  /"
done
