#!/usr/bin/env python3
"""Sweep launch forms of the controlled-phase kernel (K2).  Reports GB/s on the algorithmic minimum
32 * 2^(n-2) bytes (only the control=target=1 quarter changes).  usage: tune_phase.py [-n 30]"""
import argparse
import itertools
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/tune_phase.json")
    a = ap.parse_args()
    n = a.n
    variants = {}
    for apt, blk, nt, sl in itertools.product((1, 2, 4), (64, 256), (0, 1), (0, 1, 2, 3)):
        if sl in (1, 2) and not (apt == 1 and blk == 64):
            continue
        variants[f"a{apt}_b{blk}_nt{nt}_s{sl}"] = dict(ph_apt=apt, ph_block=blk, ph_nt=nt, ph_streams_log2=sl)
    pairs = [(n - 1, n - 2), (n - 1, n - 8), (n - 1, 20), (n - 1, 12), (n - 1, 7), (n - 1, 4), (n - 1, 2), (n - 1, 0),
             (20, 12), (12, 5), (8, 3), (21, 20), (5, 4), (3, 1)]
    gb = 32.0 * (1 << (n - 2)) / 1e9
    res = {k: {} for k in variants}
    with qc.Register(n, 0) as reg:
        reg.fill_random(1)
        for c, t in pairs:
            qc.c_phase_shift_gate(c, t, 0.3, reg)
        reg.synchronize()
        for c, t in pairs:
            for rep in range(a.reps):
                for k, v in variants.items():
                    qc.tune(**v)
                    reg.timer_start()
                    qc.c_phase_shift_gate(c, t, math.pi / 8, reg)
                    res[k].setdefault(f"{c},{t}", []).append(reg.timer_stop())
            best = sorted(((gb / (min(res[k][f"{c},{t}"]) * 1e-3), k) for k in variants), reverse=True)[:4]
            print(f"c={c:2d} t={t:2d} " + "  ".join(f"{k}={g:6.0f}" for g, k in best), flush=True)
    summ = {k: {p: gb / (min(v) * 1e-3) for p, v in r.items()} for k, r in res.items()}
    rank = sorted(((sum(1 / x for x in s.values()), k) for k, s in summ.items()))
    for tot, k in rank[:8]:
        print(f"{k:18s} harmonic-mean GB/s {len(pairs) / tot:7.0f}  min {min(summ[k].values()):6.0f}")
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(dict(n=n, min_gbytes=gb, gbs=summ), open(a.out, "w"))


if __name__ == "__main__":
    main()
