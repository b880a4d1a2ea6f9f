#!/bin/bash
# Dev tool: SQ counters of the C2 bench's kernels (separate --pmc passes, no trace domains).  usage: bash tools/gpu/pmc_sq.sh TAG
TAG=${1:-sq}
OUT=gpurun_out/pmcsq_${TAG}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_IFETCH_LEVEL SQ_CYCLES SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o s -- python3 bench.py --steps 1 --warmup 0 --T 60 --no-cpu-baseline --no-profile --no-pmmh --no-batch --no-configs > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; continue; }
  python3 - "$OUT/p$i/s_counter_collection.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-32:]
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    if "k_obs" in k or "k_apply" in k:
        print(k, {c: round(v / n) for c, (v, n) in d.items()})
PY
done
