#!/usr/bin/env python3
"""Sweep the launch forms of the Hadamard kernel (K1) on one MI355X and print GB/s per target qubit.

All variants run interleaved in ONE process on the same buffer (cdna_hip_programming.md s5.4 rule 24).
Algorithmic bytes per gate = 32 * 2^n (16 B read + 16 B written per amplitude, in place).

usage: python tools/tune_h.py [-n 30] [--reps 3] [--out gpurun_out/tune_h.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumcomputer_amd as qc  # noqa: E402

VARIANTS = {
    "pair_p1": dict(h_variant=1, h_ppt=1, h_nt=0, h_grid_cap=0),
    "pair_p2": dict(h_variant=1, h_ppt=2, h_nt=0, h_grid_cap=0),
    "pair_p4": dict(h_variant=1, h_ppt=4, h_nt=0, h_grid_cap=0),
    "pair_p8": dict(h_variant=1, h_ppt=8, h_nt=0, h_grid_cap=0),
    "pair_p4_nt": dict(h_variant=1, h_ppt=4, h_nt=1, h_grid_cap=0),
    "pair_p2_nt": dict(h_variant=1, h_ppt=2, h_nt=1, h_grid_cap=0),
    "pair_p4_g2048": dict(h_variant=1, h_ppt=4, h_nt=0, h_grid_cap=2048),
    "pair_p4_g4096": dict(h_variant=1, h_ppt=4, h_nt=0, h_grid_cap=4096),
    "pair_p4_g8192": dict(h_variant=1, h_ppt=4, h_nt=0, h_grid_cap=8192),
    "pair_p2_g4096": dict(h_variant=1, h_ppt=2, h_nt=0, h_grid_cap=4096),
    "wave_r4": dict(h_variant=2, h_wave_r=4, h_nt=0, h_grid_cap=0, h_ppt=4),
    "wave_r8": dict(h_variant=2, h_wave_r=8, h_nt=0, h_grid_cap=0, h_ppt=4),
    "wave_r4_nt": dict(h_variant=2, h_wave_r=4, h_nt=1, h_grid_cap=0, h_ppt=4),
    "wave_r4_g4096": dict(h_variant=2, h_wave_r=4, h_nt=0, h_grid_cap=4096, h_ppt=4),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--variants", default="")
    ap.add_argument("--qubits", default="")
    ap.add_argument("--out", default="gpurun_out/tune_h.json")
    a = ap.parse_args()
    names = [v for v in a.variants.split(",") if v] or list(VARIANTS)
    qubits = [int(x) for x in a.qubits.split(",") if x] or list(range(a.n))
    gbytes = 32.0 * (1 << a.n) / 1e9
    res = {k: {} for k in names}
    with qc.Register(a.n, 0) as reg:
        reg.fill_random(1)
        reg.synchronize()
        for q in qubits:                       # warm-up pass (page mapping, clocks)
            qc.hadamard_gate(q, reg)
        reg.synchronize()
        for q in qubits:
            for rep in range(a.reps):
                for k in names:
                    v = VARIANTS[k]
                    if v["h_variant"] == 2 and q >= (9 if v.get("h_wave_r") == 8 else 8):
                        continue
                    qc.tune(**v)
                    reg.timer_start()
                    qc.hadamard_gate(q, reg)
                    ms = reg.timer_stop()
                    res[k].setdefault(q, []).append(ms)
            line = f"q={q:2d} " + " ".join(
                f"{k}={gbytes / (min(res[k][q]) * 1e-3):7.0f}" for k in names if q in res[k])
            print(line, flush=True)
        print("norm2 after sweep:", reg.norm2())
    summary = {}
    for k in names:
        if res[k]:
            per_q = {q: gbytes / (min(v) * 1e-3) for q, v in res[k].items()}
            summary[k] = dict(min_gbs=min(per_q.values()), mean_gbs=sum(per_q.values()) / len(per_q), per_q=per_q)
            print(f"{k:16s} min {summary[k]['min_gbs']:7.0f} GB/s  mean {summary[k]['mean_gbs']:7.0f} GB/s over {len(per_q)} qubits")
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(n=a.n, gbytes_per_gate=gbytes, ms=res, summary=summary), f)


if __name__ == "__main__":
    main()
