#!/bin/bash
# round 5: the expanding store of a compact chain's last pass -- parity and timing (with k_meas_fast in the same build)
timeout -k 10 900 python -m pytest tests/test_gpu_basis_front.py tests/test_gpu_measure.py -x -q -m gpu > gpurun_out/r05_tests12.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r05_tests12.log
[ $rc -eq 0 ] && for a in "" "fuse_expand_fused=0"; do echo "== shor $a"; timeout -k 10 120 python tools/run_shor_modes.py $a; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_expand_fused.txt
[ $rc -eq 0 ] && bash tools/prof_cmd.sh r05g_camodc trace sq sq2 fetch write -- tools/run_camodc.py
