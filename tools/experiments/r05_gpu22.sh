#!/bin/bash
# round 5: strict gates for non-finite states (K9) -- parity; plus the suites that cover this half-day's refactors
timeout -k 10 900 python -m pytest tests/test_gpu_nonfinite.py -x -q -m gpu > gpurun_out/r05_tests22.log 2>&1; rc=$?; echo "nonfinite tests rc=$rc"; tail -15 gpurun_out/r05_tests22.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_basis_front.py tests/test_gpu_gates.py -x -q -m gpu > gpurun_out/r05_tests22b.log 2>&1; rc=$?; echo "front/gates tests rc=$rc"; tail -4 gpurun_out/r05_tests22b.log
