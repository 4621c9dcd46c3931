#!/bin/bash
# round 5, final tree: the whole GPU suite, then bench.py as the driver runs it
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_final.log 2>&1; rc=$?; echo "suite rc=$rc"; tail -4 gpurun_out/r05_gputests_final.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 900 python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r05_bench_final.json
