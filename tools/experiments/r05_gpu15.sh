#!/bin/bash
# round 5: randomised campaign against the oracle (tests/test_gpu_fuzz.py) with a fresh master seed
SEED=${1:-$RANDOM}
echo "master seed $SEED"
QCX_FUZZ_SECONDS=${2:-420} QCX_FUZZ_SEED=$SEED timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -s > gpurun_out/r05_fuzz_$SEED.log 2>&1; rc=$?
echo "fuzz rc=$rc"; grep -c "^case" gpurun_out/r05_fuzz_$SEED.log; tail -25 gpurun_out/r05_fuzz_$SEED.log | cut -c1-400
