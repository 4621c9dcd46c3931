#!/usr/bin/env python3
"""n = 30 (L = 25, M = 5): twelve stand-alone controlled modular multiplies (C = 21, the ladder's first multipliers, controls spread
over the L register), one launch per gate -- a fixed launch order for rocprofv3 passes over k_camodc"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumcomputer_amd as qc  # noqa: E402

if len(sys.argv) > 1:
    qc.tune(**{k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1:])})
L, M, Cn = 25, 5, 21
with qc.Register(L, M) as reg:
    reg.set_fusion(-1)
    qc.reset_register(reg)
    for l in range(M, L + M):
        qc.hadamard_gate(l, reg)
    x = 2
    for rep in range(2):
        for ctl in (5, 9, 14, 19, 24, 29):
            reg.timer_start(); qc.c_amodc_gate(Cn, x, ctl, reg); ms = reg.timer_stop()
            print(f"c_amodc C={Cn} A={x} control {ctl}: {ms:.3f} ms", flush=True)
            x = x * x % Cn
