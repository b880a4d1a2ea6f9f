#!/bin/bash
# GPU call for kernel iterations: parity tests first (resampler, filter, batch, full size), then a short bench line and the
# stage stamps of the DEV build.  usage: bash tools/gpu/quick.sh TAG [pytest-args]
TAG=${1:-q}; shift
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_resample.py tests/test_gpu_filter.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py tests/test_gpu_sir.py -m gpu -q -x "$@" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -4 gpurun_out/${TAG}_tests.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR|E  )" gpurun_out/${TAG}_tests.log | head -20; exit $rc; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-batch --no-pmmh > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rc=$?
if [ $rc -ne 0 ]; then tail -5 gpurun_out/${TAG}_bench.err; exit $rc; fi
python - <<PY
import json
j=json.load(open("gpurun_out/${TAG}_bench.json"))
print("us/obs %.2f  value %.3e" % (j["sweep"]["us_per_observation"], j["value"]))
for k,v in j["kernels"].items():
    if v["launches"]>10: print("   %-34s %.2f" % (k, v["avg_us"]))
PY
if [ -f bayesssm_amd/libbayesssm_amd_dev.so ]; then
export BAYESSSM_AMD_LIB=$PWD/bayesssm_amd/libbayesssm_amd_dev.so
timeout -k 10 120 python tools/diag_stamps_pf.py > gpurun_out/${TAG}_stamps_pf.txt 2>&1 && \
timeout -k 10 120 python tools/diag_stamps_head.py > gpurun_out/${TAG}_stamps_head.txt 2>&1
grep -v amdgpu.ids gpurun_out/${TAG}_stamps_pf.txt
fi
