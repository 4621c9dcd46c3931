#!/bin/bash
timeout -k 10 900 python -m pytest tests/test_gpu_basis_front.py tests/test_gpu_fusion.py -x -q -m gpu > gpurun_out/r05_tests27.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r05_tests27.log
[ $rc -eq 0 ] || exit 1
QCX_FUZZ_SECONDS=150 QCX_FUZZ_SEED=31337 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -s > gpurun_out/r05_fuzz_31337.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r05_fuzz_31337.log | cut -c1-200
timeout -k 10 300 python tools/experiments/probe_attempts.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_attempts3.txt
