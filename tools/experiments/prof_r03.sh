#!/bin/bash
# Runs on the GPU box (via gpurun).  Round 3 rocprofv3 evidence:
#   bench  kernel trace + FETCH_SIZE / WRITE_SIZE passes of bench.py as the driver runs it (headline k_h_pair, the DEFAULT fused
#          sweep kernel k_fused_rounds<1024,12,..>, the configs object: exact and tolerance-mode inverse QFT, the Shor circuit
#          with its basis-state front k_basis_front)
#   tol    SQ counters of the n = 28 inverse QFT, tolerance mode next to the exact mode (tools/run_iqft_modes.py)
#   usage: tools/prof_r03.sh <tag> [bench|tol ...]
# rocprofv3 gets the program itself after "--"; counters in their own passes (--kernel-trace only next to --pmc).
set +e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03a}
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
    ;;
tol)
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_tol -- python3 $REPO/tools/experiments/run_iqft_modes.py > $OUT/tol_under_trace.log 2> $OUT/trace_tol.err
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq_tol -- python3 $REPO/tools/experiments/run_iqft_modes.py > $OUT/tol_under_sq.log 2> $OUT/pmc_sq_tol.err
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq2_tol -- python3 $REPO/tools/experiments/run_iqft_modes.py > $OUT/tol_under_sq2.log 2> $OUT/pmc_sq2_tol.err
    echo tol done
    ;;
esac
done
cd $REPO
find $OUT -name "*.csv" -size +20M -delete
find $OUT -type f \( -name "*.db" -o -name "*.rocpd" \) -delete
du -sh $OUT
