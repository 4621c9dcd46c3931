#!/usr/bin/env python3
"""Where does a fused pass spend its time?  Runs the n=30 fused H sweep (and optionally the n=28 IQFT) with the
diagnostic knob fuse_dbg: bit 0 skips the gates of a rounds pass, bit 1 its stores, bit 2 its tile fill -- the results
are wrong by design, only the times mean something.  usage: probe_pass.py [--geoms 11:4,12:3] [--tune k=v,...]"""
import argparse
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
    ap.add_argument("--geoms", default="11:4")
    ap.add_argument("--tune", default="")
    ap.add_argument("--iqft", action="store_true")
    a = ap.parse_args()
    if a.tune:
        qc.tune(**{k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(","))})
    n = 30
    names = {0: "full", 1: "no gates", 2: "no stores", 3: "fill only", 4: "no fill", 5: "stores only", 6: "gates only (no HBM)"}
    with qc.Register(n, 0) as reg:
        reg.fill_random(1)
        reg.set_fusion(True)
        for g in a.geoms.split(","):
            T, c = (int(x) for x in g.split(":"))
            qc.tune(fuse_T=T, fuse_c=c)
            for dbg in (0, 1, 2, 3, 4, 5, 6):
                qc.tune(fuse_dbg=dbg)
                p0 = reg.fusion_stats()[0]
                dt = timed(reg, lambda: [qc.hadamard_gate(q, reg) for q in range(n)])
                passes = (reg.fusion_stats()[0] - p0) // 3
                print(f"sweep30 T={T} c={c} {names[dbg]:22s}: {dt * 1e3:7.2f} ms  {passes} passes  {dt * 1e3 / passes:6.2f} ms/pass", flush=True)
            qc.tune(fuse_dbg=0)
            reg.fill_random(1)
    if a.iqft:
        n = 28
        with qc.Register(n, 0) as reg:
            reg.fill_random(1)
            reg.set_fusion(True)
            for dbg in (0, 1, 2, 6):
                qc.tune(fuse_dbg=dbg)
                dt = timed(reg, lambda: qc.inverse_QFT(reg), reps=2)
                print(f"iqft28 {names[dbg]:22s}: {dt * 1e3:7.2f} ms", flush=True)
            qc.tune(fuse_dbg=0)


if __name__ == "__main__":
    main()
