#!/bin/bash
# GPU call: the whole -m gpu suite, then the default bench line.  usage: bash tools/gpu/full.sh TAG
TAG=${1:-f}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${TAG}_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR|E  )" gpurun_out/${TAG}_tests.log | head -20; exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rc=$?
if [ $rc -ne 0 ]; then tail -5 gpurun_out/${TAG}_bench.err; exit $rc; fi
python - <<PY
import json
j=json.load(open("gpurun_out/${TAG}_bench.json"))
print("us/obs %.2f  value %.3e  pmmh it/s %.2f  c5 us/obs %.1f  c4 %.3e  inflight %s" % (j["sweep"]["us_per_observation"], j["value"], j["pmmh_chains"]["iters_per_sec"], j["c5_chains"]["us_per_observation"], j["c4"]["particle_steps_per_s"], {k: round(v["particle_steps_per_s"]/1e9,1) for k,v in j["runs_in_flight"].items() if k!="note"}))
for k,v in j["kernels"].items():
    if v["launches"]>10: print("   %-34s %.2f" % (k, v["avg_us"]))
PY
