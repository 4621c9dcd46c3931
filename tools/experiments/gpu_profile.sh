#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and the two HBM-traffic PMC passes of bench.py.
# rocprofv3 gets the program itself after "--" (no env/bash -c hop), counters in their own passes.
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${1:-r01}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
echo trace done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err
echo write done
cd $REPO
find $OUT -name "*.csv" | head -20
# fused workloads: kernel trace of the tuning script (H sweep n=30, IQFT n=28, Shor n=30 with fusion on)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fused -- python3 $REPO/tools/experiments/tune_fuse.py --geoms 11:4 --out $OUT/tune_fuse_under_trace.json > $OUT/fused_under_trace.log 2> $OUT/trace_fused.err
echo fused trace done
cd $REPO
