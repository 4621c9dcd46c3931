#!/bin/bash
# Runs on the GPU box (via gpurun).  Round 5's rocprofv3 evidence, summarised into profiles/ by the script itself:
#   bench  kernel trace + FETCH_SIZE / WRITE_SIZE passes of bench.py as the driver runs it -> <tag>_bench_kernel_trace.json (rows per
#          kernel AND per grid size: the n = 30 launches of k_h_pair have their own row), <tag>_hbm_traffic.json
#   iqft   kernel trace + two SQ counter passes of the n = 28 inverse QFT, bit-exact default (k_fused_x8)   -> <tag>_iqft_{trace,sq,sq2}.json
#   tol    the same in the tolerance mode (k_fused_x8<.., TOL>)                                              -> <tag>_tol_{trace,sq,sq2}.json
#   shor   kernel trace of the n = 30 Shor circuit, exact x 3 then tolerance x 3, + the launch sequence      -> <tag>_shor_trace.json, <tag>_shor_pass_sequence.txt
#   usage: tools/prof_r05.sh <tag> [bench|iqft|tol|shor ...]
# rocprofv3 gets the program itself after "--" (python3 <script>); counters in their own passes (--kernel-trace only next to --pmc).
set +e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05f}
shift || true
WHAT=${*:-bench iqft tol shor}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES"
three() {       # three NAME SCRIPT: trace + two SQ passes of one script
    local name=$1 script=$2
    (cd /tmp && export TMPDIR=/tmp
     rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -- python3 $REPO/$script > $OUT/${name}_trace.log 2> $OUT/${name}_trace.err
     rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/${name}_sq -- python3 $REPO/$script > $OUT/${name}_sq.log 2> $OUT/${name}_sq.err
     rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/${name}_sq2 -- python3 $REPO/$script > $OUT/${name}_sq2.log 2> $OUT/${name}_sq2.err)
    (cd $REPO && for p in trace sq sq2; do python3 tools/summarize_prof.py $OUT/${name}_$p ${TAG}_${name}_$p > $OUT/summary_${name}_$p.txt 2>&1; done
     python3 tools/trace_seq.py $OUT/${name}_trace > profiles/${TAG}_${name}_pass_sequence.txt 2>/dev/null; grep -h "ms" $OUT/${name}_trace.log >> profiles/${TAG}_${name}_pass_sequence.txt)
    echo "$name done"
}
for w in $WHAT; do
case $w in
bench)
    (cd /tmp && export TMPDIR=/tmp
     rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err
     echo trace done
     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_pmc_fetch.json 2> $OUT/pmc_fetch.err
     echo fetch done
     rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench_under_pmc_write.json 2> $OUT/pmc_write.err
     echo write done)
    (cd $REPO && python3 tools/summarize_prof.py $OUT/trace ${TAG}_bench_kernel_trace > $OUT/summary_trace.txt 2>&1;
     python3 tools/make_traffic.py $OUT/pmc_fetch $OUT/pmc_write $TAG > $OUT/summary_traffic.txt 2>&1;
     cp $OUT/bench_under_trace.json profiles/${TAG}_bench_under_kernel_trace.json;
     cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) profiles/${TAG}_bench_kernel_stats.csv)
    ;;
iqft) three iqft tools/run_iqft_exact.py ;;
tol)  three tol tools/run_iqft_tol.py ;;
shor) three shor tools/run_shor_modes.py ;;
esac
done
cd $REPO
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/ 2>/dev/null
find $OUT -name "*.csv" -size +20M -delete
find $OUT -type f \( -name "*.db" -o -name "*.rocpd" \) -delete
du -sh $OUT
