#!/bin/bash
# round 5: workgroups per launch of the k_fused_x8 passes, exact and tolerance (fuse_x8_cap)
for a in "fuse_x8_cap=65536" "fuse_x8_cap=32768" "fuse_x8_cap=16384" "fuse_x8_cap=8192" "fuse_x8_cap=4096" "fuse_x8_cap=2048"; do echo "== $a"; timeout -k 10 200 python tools/experiments/run_iqft_modes.py $a; done 2>&1 | grep -v "amdgpu.ids\|shor30" | tee gpurun_out/r05_x8_cap.txt
