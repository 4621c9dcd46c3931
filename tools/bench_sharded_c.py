#!/usr/bin/env python3
"""The C-ABI sharded register (qcx_register_create_sharded, one process) at full size.  On a one-GPU box all shards sit on
device 0 (--devices 0): that measures what the sharded SCHEDULE costs (gate lists per shard, pack + trade passes as
local copies) next to the unsharded register, and checks windows of the 16 GiB result against the oracle.  On a
multi-GPU node leave --devices at its default (shard r on device r): the trades are then peer stores over xGMI.

  sweep   n-qubit H sweep (config 2 shape), one launch per gate and as fused passes
  shor    Shor N=21 a=2 L=n-5 M=5 circuit + measurement (config 5 shape)
  config4 H on each shard-id qubit vs a local one (config 4 shape)
usage: bench_sharded_c.py [-n 30] [--shards 8] [--devices 0] [--what sweep,shor,config4] [--out file.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumcomputer_amd as qc  # noqa: E402


def timed(reg, fn, reps=2):
    best = 1e9
    for _ in range(reps):
        reg.synchronize()
        t0 = time.perf_counter()
        fn(); reg.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def windows_match(reg, want_fn, n, count=1 << 12):
    """compare a few windows of the state with want_fn(first, count) -> float64 array (oracle side)"""
    ok = True
    for first in (0, (1 << n) // 3, (1 << n) - count):
        got = reg.read(first, count)
        ok = ok and np.array_equal(got.view(np.uint64), want_fn(first, count).view(np.uint64))
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=30)
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--devices", default="", help="comma list; empty = shard r on device r")
    ap.add_argument("--what", default="sweep,shor,config4")
    ap.add_argument("--out", default="gpurun_out/bench_sharded_c.json")
    ap.add_argument("--tune", default="", help="qcx_tune_set keys, k=v,k=v (e.g. fuse_compact=0)")
    a = ap.parse_args()
    if a.tune:
        qc.tune(**{k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(","))})
    devs = [int(x) for x in a.devices.split(",")] if a.devices else None
    n, W = a.n, a.shards
    k = W.bit_length() - 1
    out = {"n": n, "shards": W, "devices": devs or list(range(W))}
    what = a.what.split(",")
    if "sweep" in what:
        res = {}
        for label, shards in (("unsharded", 1), ("sharded", W)):
            with qc.Register(n, 0, shards=shards, devices=devs if shards > 1 else None) as reg:
                for mode, fusion in (("per_gate", -1), ("fused", 1)):
                    reg.set_fusion(fusion)
                    reg.fill_random(1)
                    sweep = lambda: [qc.hadamard_gate(q, reg) for q in range(n)]
                    sweep(); reg.synchronize()
                    e0 = reg.sharded_stats()[0]
                    dt = timed(reg, sweep)
                    res[f"{label}_{mode}"] = dict(ms=dt * 1e3, amplitude_updates_per_s=n * 2.0 ** n / dt,
                                                  exchanges_per_sweep=(reg.sharded_stats()[0] - e0) / 2)
                    print(f"sweep n={n} {label:9s} {mode:8s}: {dt * 1e3:8.2f} ms  {n * 2.0 ** n / dt:.3e} upd/s", flush=True)
        out["sweep"] = res
    if "shor" in what:
        L, M = n - 5, 5
        res = {}
        for label, shards in (("unsharded", 1), ("sharded", W)):
            with qc.Register(L, M, shards=shards, devices=devs if shards > 1 else None) as reg:
                run = lambda: (qc.reset_register(reg), qc.quantum_computation(21, 2, reg))
                run(); reg.synchronize()
                e0 = reg.sharded_stats()[0]
                dt = timed(reg, run)
                ex = (reg.sharded_stats()[0] - e0) / 2
                nrm = reg.norm2()
                t0 = time.perf_counter(); idx = qc.measure_state(reg, 0.37); tm = time.perf_counter() - t0
                res[label] = dict(circuit_ms=dt * 1e3, exchanges=ex, norm=nrm, measure_ms=tm * 1e3, measured_index=idx,
                                  omega=qc.read_omega(idx, reg))
                print(f"shor n={n} {label:9s}: circuit {dt * 1e3:8.2f} ms, {ex} exchanges, norm {nrm!r}, measure {tm * 1e3:.2f} ms -> {idx}", flush=True)
        res["same_measured_index"] = res["unsharded"]["measured_index"] == res["sharded"]["measured_index"]
        out["shor"] = res
    if "config4" in what:
        res = {}
        with qc.Register(n, 0, shards=W, devices=devs) as reg:
            reg.set_fusion(-1)
            reg.fill_random(1)
            nl = n - k
            res["local_h_ms"] = timed(reg, lambda: qc.hadamard_gate(nl - 8, reg)) * 1e3
            for q in range(n - 1, nl - 1, -1):
                reg.fill_random(1); reg.synchronize()
                res[f"global_h_q{q}_ms"] = timed(reg, lambda q=q: qc.hadamard_gate(q, reg), reps=1) * 1e3
            g = [v for kk, v in res.items() if kk.startswith("global")]
            sent = 16.0 * 2.0 ** nl * (W - 1) / W
            res["exchange_GBps_per_shard"] = sent / ((min(g) - res["local_h_ms"]) * 1e-3) / 1e9 if min(g) > res["local_h_ms"] else None
            print("config4:", json.dumps(res), flush=True)
        out["config4"] = res
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
