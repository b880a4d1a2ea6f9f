#!/bin/bash
# GPU call: full -m gpu suite, the default bench line, stage stamps of the DEV build.  A step that is killed at its limit stops the call.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2a_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r2a_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
rc=$?; tail -c 600 gpurun_out/r2a_bench.json
if [ $rc -ne 0 ]; then tail -5 gpurun_out/r2a_bench.err; exit $rc; fi
export BAYESSSM_AMD_LIB=$PWD/bayesssm_amd/libbayesssm_amd_dev.so
timeout -k 10 120 python tools/diag_stamps_pf.py > gpurun_out/r2a_stamps_pf.txt 2>&1 && \
timeout -k 10 120 python tools/diag_stamps_head.py > gpurun_out/r2a_stamps_head.txt 2>&1 && \
timeout -k 10 120 python tools/diag_stamps.py > gpurun_out/r2a_stamps.txt 2>&1
cat gpurun_out/r2a_stamps_pf.txt
