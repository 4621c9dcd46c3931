#!/bin/bash
# round 5: k_meas_fast (the scan's events from a candidate list) -- parity and timing
timeout -k 10 600 python -m pytest tests/test_gpu_measure.py -x -q -m gpu > gpurun_out/r05_tests11.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/r05_tests11.log
[ $rc -eq 0 ] && timeout -k 10 300 python tools/experiments/probe_meas.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_probe_meas11.txt
