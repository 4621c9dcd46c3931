#!/bin/bash
# round 5: k_camodc_wave -- parity, timing; then the whole GPU suite on this tree
timeout -k 10 600 python -m pytest tests/test_gpu_gates.py -x -q -m gpu > gpurun_out/r05_tests13.log 2>&1; rc=$?; echo "gates tests rc=$rc"; tail -5 gpurun_out/r05_tests13.log
[ $rc -eq 0 ] || exit 1
for a in "cam_wave=1" "cam_wave=0"; do echo "== $a"; timeout -k 10 120 python tools/run_camodc.py $a; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_camodc_wave.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_i.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -5 gpurun_out/r05_gputests_i.log
