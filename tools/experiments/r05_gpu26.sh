#!/bin/bash
# round 5: a measurement's bookkeeping launches (one copy back, self-cleaning ticket / candidate count) -- parity, attempt times
timeout -k 10 900 python -m pytest tests/test_gpu_measure.py tests/test_gpu_sharded_c.py tests/test_gpu_basis_front.py tests/test_gpu_maxsize.py -x -q -m gpu > gpurun_out/r05_tests26.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/r05_tests26.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/experiments/probe_attempts.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_attempts2.txt
QCX_FUZZ_SECONDS=120 QCX_FUZZ_SEED=555 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -s > gpurun_out/r05_fuzz_555.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r05_fuzz_555.log | cut -c1-200
