#!/bin/bash
# round 5: the plan cache behind circuit fronts too -- parity, then attempt times over n
timeout -k 10 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_basis_front.py tests/test_host_driver.py -x -q -m gpu > gpurun_out/r05_tests23.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r05_tests23.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/experiments/probe_attempts.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_attempts.txt
QCX_FUZZ_SECONDS=200 QCX_FUZZ_SEED=9191 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -s > gpurun_out/r05_fuzz_9191.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r05_fuzz_9191.log | cut -c1-200
