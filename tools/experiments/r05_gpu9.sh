#!/bin/bash
# round 5: the branch-free tolerance step and the k_camodc cleanup, checked and timed
python -m pytest tests/test_gpu_fusion.py tests/test_gpu_gates.py -x -q -m gpu > gpurun_out/r05_tests9.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_tests9.log
timeout -k 10 200 python tools/experiments/run_iqft_modes.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/experiments/run_iqft_modes.py 2>&1 | grep -v amdgpu.ids
