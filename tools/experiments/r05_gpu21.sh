#!/bin/bash
# round 5: randomised campaign at n = 20 .. 26
SEED=${1:-$RANDOM}
echo "master seed $SEED"
QCX_FUZZ_NMIN=20 QCX_FUZZ_NMAX=26 QCX_FUZZ_SECONDS=${2:-840} QCX_FUZZ_SEED=$SEED timeout -k 10 1100 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -s > gpurun_out/r05_fuzz_huge_$SEED.log 2>&1; rc=$?
echo "fuzz rc=$rc"; grep -c "^case" gpurun_out/r05_fuzz_huge_$SEED.log; grep -c "shards=[248]" gpurun_out/r05_fuzz_huge_$SEED.log; tail -4 gpurun_out/r05_fuzz_huge_$SEED.log | cut -c1-300
