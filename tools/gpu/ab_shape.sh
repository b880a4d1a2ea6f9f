#!/bin/bash
# Experiment: the scan workgroup as NT x EL = 512 x 4 instead of 256 x 8 (same 2048 terms, same block records).
# Build first:  make -C bayesssm_amd/csrc NT=512 EL=4 OUT=../libbayesssm_amd_nt512.so
# usage: bash tools/gpu/ab_shape.sh TAG
TAG=${1:-shape}
mkdir -p gpurun_out
export BAYESSSM_AMD_LIB=$PWD/bayesssm_amd/libbayesssm_amd_nt512.so
timeout -k 10 600 python -m pytest tests/test_gpu_resample.py tests/test_gpu_filter.py tests/test_gpu_batch.py tests/test_gpu_fullsize.py tests/test_gpu_sir.py tests/test_gpu_fold.py -m gpu -q > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
grep -E "^(FAILED|ERROR)" gpurun_out/${TAG}_tests.log | head -20
for lib in libbayesssm_amd_nt512.so libbayesssm_amd.so libbayesssm_amd_nt512.so libbayesssm_amd.so; do
export BAYESSSM_AMD_LIB=$PWD/bayesssm_amd/$lib
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-batch --no-pmmh 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('$lib us/obs %.2f' % j['sweep']['us_per_observation'], {k.split('(')[0][:22]: round(v['avg_us'],2) for k,v in j['kernels'].items() if v['launches']>10})
" || exit 1
done
