#!/bin/bash
# round 5: randomised campaign, larger registers (n = 17 .. 25), sharded registers included
SEED=${1:-$RANDOM}
echo "master seed $SEED"
QCX_FUZZ_NMIN=17 QCX_FUZZ_NMAX=25 QCX_FUZZ_SECONDS=${2:-600} QCX_FUZZ_SEED=$SEED timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -s > gpurun_out/r05_fuzz_big_$SEED.log 2>&1; rc=$?
echo "fuzz rc=$rc"; grep -c "^case" gpurun_out/r05_fuzz_big_$SEED.log; grep -c "shards=[248]" gpurun_out/r05_fuzz_big_$SEED.log; tail -8 gpurun_out/r05_fuzz_big_$SEED.log | cut -c1-600
