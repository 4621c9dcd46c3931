#!/bin/bash
# The first run on a node with more than one MI355X: everything multi-GPU in this repo has so far only run on virtual
# shards of ONE GPU and over gloo on the CPU.  This script runs the multi-GPU surface in a fixed order,
# each step as its OWN process with its own log and exit code, cheapest and most diagnostic first, and stops at the
# first failure (a failed or hung GPU step must not be followed by another one).
#
#   usage: tools/first_multigpu_lease.sh [N_GPUS=8] [OUT=gpurun_out/lease]
#
# Steps (logs in $OUT/NN_name.log, verdicts in $OUT/summary.txt):
#   01 devices        rocm-smi / hipDeviceCanAccessPeer matrix as the C host sees it
#   02 selfcheck      bench.py --gpus 2 at n_local = 22, rehearsal size: the exchange self-check picks overlap/sync/pairwise
#   03 selfcheck_pw   the same with QCX_SHARD_EXCHANGE=pairwise (north_star's literal form)
#   04 sharded_c      tests/test_gpu_sharded_c.py     one-process C host: peer stores, staged copies, relays, compact companion
#   05 sharded_mp     tests/test_gpu_sharded_multiproc.py   one process per GPU over RCCL
#   06 bench_2/4/8    bench.py --gpus N --steps 5 --warmup 1 at full size (config 4 and 5 ride on the line)
#   07 bench_c        tools/bench_sharded_c.py        the C host's sweep / config 4 / config 5
#   08 bench_8_pw     bench.py --gpus N with the pairwise exchange, for the comparison DESIGN.md s5 asks for
set -u
N=${1:-8}
OUT=${2:-gpurun_out/lease}
cd "$(dirname "$0")/.." || exit 1
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
: > "$OUT/summary.txt"

step() {          # step NAME TIMEOUT_S cmd...
    local name=$1 tmo=$2; shift 2
    echo "== $name: $*" | tee -a "$OUT/summary.txt"
    timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "   exit $rc" | tee -a "$OUT/summary.txt"
    if [ $rc -ne 0 ]; then
        tail -n 30 "$OUT/$name.log"
        echo "STOPPED at $name (exit $rc): read $OUT/$name.log before starting anything else on the GPUs" | tee -a "$OUT/summary.txt"
        exit $rc
    fi
}

step 01_devices 120 python3 -c "
import quantumcomputer_amd as qc, ctypes as C
lib = qc.lib()
n = C.c_int(0); lib.qcx_device_count(C.byref(n)); print('devices', n.value)
print('spread for $N shards:', qc.spread_devices($N))
"
step 02_selfcheck 600 python3 bench.py --gpus 2 --steps 1 --warmup 1 --n-local 22 --no-cpu-baseline
step 03_selfcheck_pw 600 env QCX_SHARD_EXCHANGE=pairwise python3 bench.py --gpus 2 --steps 1 --warmup 1 --n-local 22 --no-cpu-baseline
step 04_sharded_c 1200 python3 -m pytest tests/test_gpu_sharded_c.py -x -q -m gpu
step 05_sharded_mp 1200 python3 -m pytest tests/test_gpu_sharded_multiproc.py -x -q -m gpu
for g in 2 4 8; do
    [ "$g" -le "$N" ] || continue
    step "06_bench_$g" 1200 python3 bench.py --gpus "$g" --steps 5 --warmup 1
done
step 07_bench_c 1200 python3 tools/bench_sharded_c.py --shards "$N"
step 08_bench_${N}_pw 1200 env QCX_SHARD_EXCHANGE=pairwise python3 bench.py --gpus "$N" --steps 5 --warmup 1 --no-cpu-baseline
echo "ALL STEPS GREEN" | tee -a "$OUT/summary.txt"
