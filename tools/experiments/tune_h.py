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

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

def _v(variant, ppt=1, nt=3, cap=0, r=4, wc=0, blk=256, sl=0, wb=256):
    return dict(h_variant=variant, h_ppt=ppt, h_nt=nt, h_grid_cap=cap, h_wave_r=r, h_wc=wc, h_block=blk,
                h_streams_log2=sl, h_wave_block=wb, h_skew=0)


VARIANTS = {"auto": dict(h_variant=0)}
for _sl in (1, 2, 3):
    for _skew in (0, 1237, 4099, 30011, 262147):
        for _ppt in (1, 2):
            d = _v(1, ppt=_ppt, blk=64, sl=_sl); d["h_skew"] = _skew
            VARIANTS[f"p{_ppt}_s{_sl}_k{_skew}"] = d

# capped (grid-stride) grids: fewer, longer-lived workgroups walking the pairs in order
for _cap in (2048, 4096, 8192, 16384, 32768, 65536, 262144):
    for _blk in (64, 256):
        VARIANTS[f"cap{_cap}_b{_blk}"] = _v(1, ppt=1, blk=_blk, cap=_cap)


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
                    if v["h_variant"] == 2 and q >= {8: 9, 4: 8, 2: 7}[v.get("h_wave_r", 4)]:
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
