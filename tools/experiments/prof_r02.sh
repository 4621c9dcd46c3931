#!/bin/bash
# Runs on the GPU box (via gpurun).  rocprofv3 evidence for the kernels that are DEFAULT since round 1's last
# commits (k_fused_rounds) and for the per-gate kernels k_phase / k_camodc, which had no counters.
#   usage: tools/prof_r02.sh <tag> [fused|gates|bench|sq ...]     (default: all)
# rocprofv3 gets the program itself after "--"; counters in their own passes (--kernel-trace only next to --pmc).
set +e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02a}
shift || true
WHAT=${*:-fused gates sq bench}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WHAT; do
case $w in
fused)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fused -- python3 $REPO/tools/experiments/tune_fuse.py --geoms 11:4 --out $OUT/tune_fuse_under_trace.json > $OUT/fused_under_trace.log 2> $OUT/trace_fused.err
    echo fused trace done
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_fused -- python3 $REPO/tools/experiments/tune_fuse.py --geoms 11:4 --out $OUT/tune_fuse_under_fetch.json > $OUT/fused_under_fetch.log 2> $OUT/pmc_fetch_fused.err
    echo fused fetch done
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_fused -- python3 $REPO/tools/experiments/tune_fuse.py --geoms 11:4 --out $OUT/tune_fuse_under_write.json > $OUT/fused_under_write.log 2> $OUT/pmc_write_fused.err
    echo fused write done
    ;;
sq)
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq_fused -- python3 $REPO/tools/experiments/tune_fuse.py --geoms 11:4 --out $OUT/tune_fuse_under_sq.json > $OUT/fused_under_sq.log 2> $OUT/pmc_sq_fused.err
    echo fused sq done
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq2_fused -- python3 $REPO/tools/experiments/tune_fuse.py --geoms 11:4 --out $OUT/tune_fuse_under_sq2.json > $OUT/fused_under_sq2.log 2> $OUT/pmc_sq2_fused.err
    echo fused sq2 done
    ;;
gates)
    python3 $REPO/tools/experiments/probe_gates.py --out $OUT/probe_gates.json > $OUT/probe_gates.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gates -- python3 $REPO/tools/experiments/probe_gates.py --out $OUT/probe_gates_under_trace.json > $OUT/gates_under_trace.log 2> $OUT/trace_gates.err
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_gates -- python3 $REPO/tools/experiments/probe_gates.py --out $OUT/probe_gates_under_fetch.json > $OUT/gates_under_fetch.log 2> $OUT/pmc_fetch_gates.err
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_gates -- python3 $REPO/tools/experiments/probe_gates.py --out $OUT/probe_gates_under_write.json > $OUT/gates_under_write.log 2> $OUT/pmc_write_gates.err
    echo gates done
    ;;
bench)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
    echo trace done
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fused > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err
    echo fetch done
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fused > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err
    echo write done
    ;;
esac
done
cd $REPO
# keep what travels back small: the per-dispatch CSVs are all the summaries need
find $OUT -name "*.csv" -size +20M -delete
find $OUT -type f \( -name "*.db" -o -name "*.rocpd" \) -delete
du -sh $OUT
