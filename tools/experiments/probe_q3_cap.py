#!/usr/bin/env python3
"""chained radix-8 passes over the number of workgroups: n = 30 fused Hadamard sweep (exact), n = 28 tolerance inverse QFT and the
n = 30 tolerance Shor circuit"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402


def best(reg, fn, reps=4):
    fn(); reg.synchronize()
    b = 1e9
    for _ in range(reps):
        reg.timer_start(); fn(); b = min(b, reg.timer_stop())
    return b


with qc.Register(30, 0) as reg, qc.Register(28, 0) as r28:
    reg.set_fusion(1); reg.fill_random(1)
    r28.set_fusion(2); r28.fill_random(1)

    def sweep():
        for q in range(30):
            qc.hadamard_gate(q, reg)
        reg.flush()
    for cap in (2048, 3072, 4096, 8192, 16384, 32768, 65536, 0):
        qc.tune(fuse_q3_cap_exact=cap, fuse_q3_cap=cap)
        print(f"workgroups {cap or 'one per tile':>12}: n=30 sweep {best(reg, sweep):7.3f} ms   n=28 tolerance IQFT {best(r28, lambda: qc.inverse_QFT(r28)):7.3f} ms", flush=True)
