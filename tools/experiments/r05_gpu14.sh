#!/bin/bash
# round 5: k_camodc_wave with whole-line nontemporal traffic against rows < C only and against the LDS tile kernel
timeout -k 10 600 python -m pytest tests/test_gpu_gates.py -x -q -m gpu > gpurun_out/r05_tests14.log 2>&1; rc=$?; echo "gates tests rc=$rc"; tail -5 gpurun_out/r05_tests14.log
[ $rc -eq 0 ] || exit 1
for a in "cam_wave=1 cam_wave_lines=1" "cam_wave=1 cam_wave_lines=0" "cam_wave=0" "cam_wave=1 cam_wave_lines=1 cam_wave_cap=8192" "cam_wave=1 cam_wave_lines=1 cam_wave_cap=32768" "cam_wave=1 cam_wave_lines=1 cam_skip=0"; do echo "== $a"; timeout -k 10 120 python tools/run_camodc.py $a; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_camodc_wave2.txt
