#!/usr/bin/env python3
"""Fused-pass throughput for the three workloads, over tile geometries (T = tile bits, c = contiguous
low bits).  usage: tune_fuse.py [--quick]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402


def timed(reg, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        reg.synchronize()
        t0 = time.perf_counter()
        fn(); reg.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/tune_fuse.json")
    ap.add_argument("--geoms", default="12:4,12:5,12:6,11:4,11:5,11:3,12:3,10:4")
    ap.add_argument("--tune", default="", help="extra tunables, e.g. fuse_rounds_occ=7,fuse_grid_cap=16384")
    a = ap.parse_args()
    geoms = [tuple(int(x) for x in g.split(":")) for g in a.geoms.split(",")]
    if a.tune:
        qc.tune(**{k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(","))})
    out = {}
    n = 30
    with qc.Register(n, 0) as reg:
        reg.fill_random(1)
        reg.set_fusion(True)
        for T, c in geoms:
            qc.tune(fuse_T=T, fuse_c=c)
            p0 = reg.fusion_stats()[0]
            dt = timed(reg, lambda: [qc.hadamard_gate(q, reg) for q in range(n)])
            passes = (reg.fusion_stats()[0] - p0) // 3
            out[f"sweep30_T{T}_c{c}"] = dict(seconds=dt, passes=passes, amplitude_updates_per_s=n * 2.0 ** n / dt,
                                             gbs_per_pass=passes * 32 * 2.0 ** n / dt / 1e9)
            print(f"H-sweep n=30 T={T} c={c}: {dt * 1e3:7.2f} ms, {passes} passes, {n * 2.0 ** n / dt:.3e} upd/s, "
                  f"{passes * 32 * 2.0 ** n / dt / 1e9:6.0f} GB/s per pass", flush=True)
    n = 28
    with qc.Register(n, 0) as reg:
        reg.fill_random(1)
        reg.set_fusion(True)
        for T, c in geoms:
            qc.tune(fuse_T=T, fuse_c=c)
            p0 = reg.fusion_stats()[0]
            dt = timed(reg, lambda: qc.inverse_QFT(reg), reps=2)
            passes = (reg.fusion_stats()[0] - p0) // 2
            out[f"iqft28_T{T}_c{c}"] = dict(seconds=dt, passes=passes, amplitude_updates_per_s=406 * 2.0 ** n / dt)
            print(f"IQFT n=28 T={T} c={c}: {dt * 1e3:7.2f} ms, {passes} passes, {406 * 2.0 ** n / dt:.3e} upd/s", flush=True)
    L, M = 25, 5
    with qc.Register(L, M) as reg:
        reg.set_fusion(True)
        for T, c in geoms[:4]:
            qc.tune(fuse_T=T, fuse_c=c)
            p0 = reg.fusion_stats()[0]

            def run():
                qc.reset_register(reg); qc.quantum_computation(21, 2, reg)
            dt = timed(reg, run, reps=2)
            passes = (reg.fusion_stats()[0] - p0) // 2
            nrm = reg.norm2()
            out[f"shor30_T{T}_c{c}"] = dict(seconds=dt, passes=passes, amplitude_updates_per_s=375 * 2.0 ** 30 / dt, norm=nrm)
            print(f"Shor n=30 T={T} c={c}: {dt * 1e3:7.2f} ms, {passes} passes, {375 * 2.0 ** 30 / dt:.3e} upd/s, norm {nrm}", flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
