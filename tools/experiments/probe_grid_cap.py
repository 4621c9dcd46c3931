#!/usr/bin/env python3
"""fused phase passes over the number of workgroups of the rounds kernels (fuse_grid_cap) and of the radix-8 tolerance kernel
(fuse_q3_cap): n = 28 inverse QFT and n = 30 Shor circuit, exact and tolerance mode"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402


def best(reg, fn, reps=3):
    fn(); reg.synchronize()
    b = 1e9
    for _ in range(reps):
        reg.timer_start(); fn(); b = min(b, reg.timer_stop())
    return b


with qc.Register(25, 5) as reg, qc.Register(28, 0) as r28:
    r28.fill_random(1)

    def shor():
        qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.flush()
    for cap, q3cap in ((24576, 65536), (65536, 65536), (262144, 65536), (0, 65536), (24576, 16384), (24576, 262144), (24576, 0)):
        qc.tune(fuse_grid_cap=cap, fuse_q3_cap=q3cap)
        out = []
        for mode in (0, 2):
            reg.set_fusion(mode); r28.set_fusion(mode)
            out.append(f"mode {mode}: IQFT28 {best(r28, lambda: qc.inverse_QFT(r28)):6.3f}  Shor30 {best(reg, shor):6.3f}")
        print(f"rounds cap {cap or 'per tile':>8}  q3 cap {q3cap or 'per tile':>8} | " + " | ".join(out), flush=True)
