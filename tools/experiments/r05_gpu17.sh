#!/bin/bash
# round 5: k_gen_cols alone under the kernel trace: full, gates skipped (fuse_dbg=1), stores skipped (2), both (3)
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_gc$d -- python3 $GRAFT_REPO_ROOT/tools/run_shor_modes.py fuse_dbg=$d > $GRAFT_REPO_ROOT/gpurun_out/prof_gc$d.log 2>&1
  echo "== fuse_dbg=$d"; f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_gc$d/*/*kernel_stats.csv | head -1); grep -E "k_gen_cols|k_fused_x8|k_expand" $f | cut -c1-160
done
