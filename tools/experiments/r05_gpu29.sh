#!/bin/bash
# kernel trace of small attempts: where the 77 / 90 us at n = 16 / 20 go
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_small
mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/tools/experiments/run_attempts_small.py > $OUT/run.log 2> $OUT/run.err)
python3 $REPO/tools/experiments/trace_gaps.py $OUT/trace > $REPO/gpurun_out/r05_small_attempt_gaps.txt 2>&1
cat $REPO/gpurun_out/r05_small_attempt_gaps.txt
find $OUT -type f \( -name "*.db" -o -name "*.rocpd" \) -delete
