#!/bin/bash
# round 5: s_setprio around the memory phases of k_fused_x8 (fill issue / store at priority 3, rounds at 0)
timeout -k 10 200 python tools/experiments/run_iqft_modes.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/experiments/run_iqft_modes.py 2>&1 | grep -v amdgpu.ids
