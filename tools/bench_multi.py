#!/usr/bin/env python3
"""BASELINE.json configurations 4 and 5 on a multi-GPU node (one rank per GPU):

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29600 \
        tools/bench_multi.py --config 4          # n=32 over 8 GPUs: H on each global qubit vs a local one
  ... tools/bench_multi.py --config 5            # n=30 Shor N=21 a=2 (L=25, M=5) + measurement, end to end

Rank 0 prints one JSON object.  (Not part of bench.py: the driver's scaling run uses the H sweep.)"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--fused", action="store_true", help="run gate lists through the fused-pass scheduler")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    import quantumcomputer_amd as qc
    from quantumcomputer_amd.sharded import ShardedRegister
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    qc.lib()
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    k = world.bit_length() - 1

    def timed(fn):
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dist.barrier()
        return time.perf_counter() - t0

    out = {"config": a.config, "n_gpus": world}
    if a.config == 4:
        n = a.n or 32
        reg = ShardedRegister(n, 0, fusion=a.fused)
        reg.fill_random(1); reg.synchronize()
        nl = reg.n_local
        res = {}
        res["local_h_ms"] = timed(lambda: (reg.hadamard_gate(nl - 8), reg.synchronize())) * 1e3
        for q in range(n - 1, nl - 1, -1):                     # every global qubit: one exchange each
            reg.fill_random(1); reg.synchronize()
            res[f"global_h_q{q}_ms"] = timed(lambda q=q: (reg.hadamard_gate(q), reg.synchronize())) * 1e3
        shard_bytes = 16.0 * (1 << nl)
        g = [v for kk, v in res.items() if kk.startswith("global")]
        out.update(n=n, shard_GiB=shard_bytes / 2**30, results=res,
                   note="a global H = pack + all-to-all of (W-1)/W of the shard + the local gate")
        if g and min(g) > res["local_h_ms"]:
            out["exchange_GBps_per_gpu"] = shard_bytes * (world - 1) / world / ((min(g) - res["local_h_ms"]) * 1e-3) / 1e9
    else:
        L, M, Cn, aa = 25, 5, 21, 2
        reg = ShardedRegister(L, M, fusion=a.fused)

        def circuit():
            reg.reset_register(); reg.quantum_computation(Cn, aa); reg.synchronize()
        circuit()
        dt = timed(circuit)
        nrm = reg.norm2()
        rng = qc.Rng(12345)
        t0 = time.perf_counter(); idx = reg.measure_state(rng.uniform()); tm = time.perf_counter() - t0
        xt = 0
        for p in range(L):
            xt |= ((idx >> (L + M - 1 - p)) & 1) << p
        w = xt / float(1 << L)
        gates = 3 * L + L * (L - 1) // 2
        out.update(n=L + M, gates=gates, fused=bool(a.fused), circuit_seconds=dt, amplitude_updates_per_s=gates * 2.0 ** (L + M) / dt,
                   exchanges=reg.exchanges, measure_seconds=tm, measured_index=idx, omega=w,
                   nearest_sixth=min((abs(w - j / 6.0), j) for j in range(7))[1], total_probability=nrm)
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
