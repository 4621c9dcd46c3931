#!/usr/bin/env python3
"""Fused n=30 Hadamard sweep over launch knobs of the rounds kernel (tile order swizzle, grid cap, occupancy)."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

n = 30
with qc.Register(n, 0) as reg:
    reg.set_fusion(1)
    reg.fill_random(1)

    def sweep():
        for q in range(n):
            qc.hadamard_gate(q, reg)

    for swz, cap, occ, dbg in itertools.product([0, 1, 2, 3, 4, 5], [24576, 8192], [8], [0, 1]):
        qc.tune(fuse_swz=swz, fuse_grid_cap=cap, fuse_rounds_occ=occ, fuse_dbg=dbg)
        sweep(); reg.synchronize()
        best = 1e9
        for _ in range(3):
            reg.timer_start(); sweep(); best = min(best, reg.timer_stop())
        print(f"sweep30 swz={swz} cap={cap} occ={occ} dbg={dbg}: {best:7.3f} ms", flush=True)
