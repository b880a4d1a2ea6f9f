#!/bin/bash
# stage stamps only (DEV build)
TAG=${1:-s}
export BAYESSSM_AMD_LIB=$PWD/bayesssm_amd/libbayesssm_amd_dev.so
timeout -k 10 120 python tools/diag_stamps_pf.py > gpurun_out/${TAG}_stamps_pf.txt 2>&1 && \
timeout -k 10 120 python tools/diag_stamps_head.py > gpurun_out/${TAG}_stamps_head.txt 2>&1
grep -v amdgpu.ids gpurun_out/${TAG}_stamps_pf.txt
