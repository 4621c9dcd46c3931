#!/usr/bin/env python3
"""n=28 inverse_QFT in tolerance mode over pass-kernel launch knobs: tile order swizzle (fuse_swz), grid cap, geometry."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402


def timed(reg, fn, reps=3):
    best = 1e30
    for _ in range(reps):
        reg.synchronize()
        reg.timer_start(); fn(); best = min(best, reg.timer_stop())
    return best


n = 28
with qc.Register(n, 0) as reg:
    reg.set_fusion(2)
    reg.fill_random(1)
    for (T, c), swz, cap, dbg in itertools.product([(10, 4), (11, 4)], [0, 1, 2, 3, 4], [24576, 8192, 65536, 0], [1, 0]):
        qc.tune(fuse_T=T, fuse_c=c, fuse_tol_occ=6, fuse_swz=swz, fuse_grid_cap=cap, fuse_dbg=dbg)
        qc.inverse_QFT(reg)
        ms = timed(reg, lambda: qc.inverse_QFT(reg))
        print(f"T={T} c={c} swz={swz} cap={cap:6d} dbg={dbg}: {ms:7.3f} ms", flush=True)
