#!/usr/bin/env python3
"""n = 30 Shor circuit and n = 28 inverse QFT under different tile geometries of the phase passes (exact and tolerance mode)"""
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


geoms = [dict(), dict(fuse_T_phase=11, fuse_c_phase=4), dict(fuse_T_phase=12, fuse_c_phase=4), dict(fuse_T_phase=12, fuse_c_phase=3),
         dict(fuse_T_phase=11, fuse_c_phase=3), dict(fuse_T_phase=10, fuse_c_phase=3)]
keys = ("fuse_T_phase", "fuse_c_phase", "fuse_T", "fuse_c")
old = {k: qc.lib().qcx_tune_get(k.encode()) for k in keys}
with qc.Register(25, 5) as reg, qc.Register(28, 0) as r28:
    for g in geoms:
        qc.tune(**old); qc.tune(**g)
        for mode in (0,):
            reg.set_fusion(mode)
            def shor():
                qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.flush()
            p0 = reg.fusion_stats()[0]
            t = best(reg, shor)
            passes = (reg.fusion_stats()[0] - p0) // 4
            r28.set_fusion(mode); r28.fill_random(1)
            p0 = r28.fusion_stats()[0]
            t2 = best(r28, lambda: qc.inverse_QFT(r28))
            print(f"{g}: mode {mode}: n=30 Shor {t:7.3f} ms ({passes} passes incl. front)   n=28 IQFT {t2:7.3f} ms ({(r28.fusion_stats()[0] - p0) // 4} passes)", flush=True)
qc.tune(**old)
