#!/bin/bash
# round 5: k_gen_cols -- the waves' column assignment rotates with the tile number
timeout -k 10 900 python -m pytest tests/test_gpu_basis_front.py -x -q -m gpu > gpurun_out/r05_tests16.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r05_tests16.log
[ $rc -eq 0 ] || exit 1
for a in "" "fuse_cols_waves=6" "fuse_cols_waves=8"; do echo "== shor $a"; timeout -k 10 120 python tools/run_shor_modes.py $a; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_gen_cols_rot.txt
