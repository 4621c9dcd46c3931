#!/bin/bash
# Runs on the GPU box (via gpurun).  Round 4 rocprofv3 evidence, everything summarised into profiles/ by the script itself:
#   bench  kernel trace + FETCH_SIZE / WRITE_SIZE passes of bench.py as the driver runs it -> <tag>_bench_kernel_trace.json (rows
#          per kernel AND per grid size: the n = 30 launches of k_h_pair have their own row), <tag>_hbm_traffic.json
#   tol    SQ counters of the n = 28 inverse QFT and the n = 30 Shor circuit, exact next to tolerance mode (tools/run_iqft_modes.py)
#   usage: tools/prof_r04.sh <tag> [bench|tol ...]
# rocprofv3 gets the program itself after "--"; counters in their own passes (--kernel-trace only next to --pmc).
set +e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04f}
shift || true
WHAT=${*:-bench tol}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in $WHAT; do
case $w in
bench)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
    echo trace done
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err
    echo fetch done
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err
    echo write done
    (cd $REPO && python3 tools/summarize_prof.py $OUT/trace ${TAG}_bench_kernel_trace > $OUT/summary_trace.txt 2>&1;
     python3 tools/make_traffic.py $OUT/pmc_fetch $OUT/pmc_write $TAG > $OUT/summary_traffic.txt 2>&1;
     cp $OUT/bench_under_trace.json profiles/${TAG}_bench_under_kernel_trace.json;
     cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) profiles/${TAG}_bench_kernel_stats.csv)
    ;;
tol)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_tol -- python3 $REPO/tools/experiments/run_iqft_modes.py > $OUT/tol_under_trace.log 2> $OUT/trace_tol.err
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq_tol -- python3 $REPO/tools/experiments/run_iqft_modes.py > $OUT/tol_under_sq.log 2> $OUT/pmc_sq_tol.err
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq2_tol -- python3 $REPO/tools/experiments/run_iqft_modes.py > $OUT/tol_under_sq2.log 2> $OUT/pmc_sq2_tol.err
    echo tol done
    (cd $REPO && python3 tools/summarize_prof.py $OUT/trace_tol ${TAG}_tol_kernel_trace > $OUT/summary_tol_trace.txt 2>&1;
     python3 tools/summarize_prof.py $OUT/pmc_sq_tol ${TAG}_tol_pmc_sq > $OUT/summary_tol_sq.txt 2>&1;
     python3 tools/summarize_prof.py $OUT/pmc_sq2_tol ${TAG}_tol_pmc_sq2 > $OUT/summary_tol_sq2.txt 2>&1)
    ;;
esac
done
cd $REPO
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/ 2>/dev/null
find $OUT -name "*.csv" -size +20M -delete
find $OUT -type f \( -name "*.db" -o -name "*.rocpd" \) -delete
du -sh $OUT
