#!/usr/bin/env python3
"""Tolerance mode (qcx_set_fusion(reg, 2)) timings: n=28 inverse_QFT and n=30 Shor circuit over tile geometries,
occupancy targets and the diagnostic knob fuse_dbg (1 = gates skipped: what the memory pipeline alone costs).
usage: tune_tol.py [--geoms 11:4,12:4,...] [--out file.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402


def timed(reg, fn, reps=3):
    best, passes = 1e30, 0
    for _ in range(reps):
        reg.synchronize()
        p0 = reg.fusion_stats()[0]
        reg.timer_start(); fn(); ms = reg.timer_stop()
        if ms < best:
            best, passes = ms, reg.fusion_stats()[0] - p0
    return best, passes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/tune_tol.json")
    ap.add_argument("--geoms", default="11:4,12:4,12:3,10:4,11:3,12:5")
    ap.add_argument("--occ", default="6,8")
    ap.add_argument("--shor", action="store_true")
    a = ap.parse_args()
    geoms = [tuple(int(x) for x in g.split(":")) for g in a.geoms.split(",")]
    out = {}
    n = 28
    with qc.Register(n, 0) as reg:
        reg.set_fusion(2)
        reg.fill_random(1)
        for T, c in geoms:
            for occ in [int(x) for x in a.occ.split(",")]:
                for dbg in (0, 1):
                    qc.tune(fuse_T=T, fuse_c=c, fuse_tol_occ=occ, fuse_dbg=dbg)
                    qc.inverse_QFT(reg)
                    ms, passes = timed(reg, lambda: qc.inverse_QFT(reg))
                    out[f"iqft28_T{T}_c{c}_occ{occ}_dbg{dbg}"] = dict(ms=ms, passes=passes)
                    print(f"IQFT n=28 tol T={T} c={c} occ={occ} dbg={dbg}: {ms:7.3f} ms, {passes} passes, "
                          f"{passes * 32 * 2.0 ** n / ms / 1e6:6.0f} GB/s per pass", flush=True)
        qc.tune(fuse_dbg=0)
    if a.shor:
        L, M = 25, 5
        with qc.Register(L, M) as reg:
            for mode in (0, 2):
                reg.set_fusion(mode)
                for T, c in geoms[:3]:
                    qc.tune(fuse_T=T, fuse_c=c, fuse_tol_occ=6)

                    def run():
                        qc.reset_register(reg); qc.quantum_computation(21, 2, reg)
                    run()
                    ms, passes = timed(reg, run, reps=2)
                    out[f"shor30_mode{mode}_T{T}_c{c}"] = dict(ms=ms, passes=passes, norm=reg.norm2())
                    print(f"Shor n=30 mode {mode} T={T} c={c}: {ms:7.2f} ms, {passes} passes", flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
