#!/bin/bash
# round 5, second session: interleaved tile streams -- k_camodc sweep, defaults of the fused passes, parity tests of what changed
for s in 0 1 2 3 4; do echo "== cam_streams_log2=$s"; timeout -k 10 120 python tools/run_camodc.py cam_streams_log2=$s 2>&1 | grep -v amdgpu.ids | tail -6; done > gpurun_out/r05_cam_streams.txt 2>&1
cat gpurun_out/r05_cam_streams.txt | awk '/==/{print} /control/{s+=$NF*0; print}' | tail -40
timeout -k 10 200 python tools/experiments/probe_streams.py -1 0 3 > gpurun_out/r05_streams2.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_streams2.txt
timeout -k 10 600 python -m pytest tests/test_gpu_gates.py tests/test_gpu_fusion.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r05_tests28.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_tests28.log
