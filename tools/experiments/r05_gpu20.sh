#!/bin/bash
# round 5: the plan cache -- parity and timing
timeout -k 10 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_basis_front.py -x -q -m gpu > gpurun_out/r05_tests20.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r05_tests20.log
[ $rc -eq 0 ] || exit 1
for a in "" "fuse_plan_cache=0"; do echo "== $a"; timeout -k 10 200 python tools/experiments/run_iqft_modes.py $a; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_plan_cache.txt
