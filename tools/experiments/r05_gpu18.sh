#!/bin/bash
# round 5: k_gen_cols -- workgroups per launch (the prologue of a workgroup is long: masks staged, the hot combination's residue walked)
for a in "fuse_grid_cap=1536" "fuse_grid_cap=3072" "fuse_grid_cap=6144" "fuse_grid_cap=12288" "fuse_grid_cap=24576" "fuse_grid_cap=65536" "fuse_grid_cap=0"; do echo "== shor $a"; timeout -k 10 120 python tools/run_shor_modes.py $a; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_gen_cols_grid.txt
