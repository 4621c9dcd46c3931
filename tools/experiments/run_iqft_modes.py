#!/usr/bin/env python3
"""n = 28 inverse QFT three times in the exact fused mode (0) and three times in the tolerance mode (2), in that order,
and the n = 30 Shor circuit (reset + quantum_computation) twice per mode -- a fixed launch order for rocprofv3 passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

if len(sys.argv) > 1:
    qc.tune(**{k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1:])})
with qc.Register(28, 0) as reg:
    for mode in (0, 2):
        reg.set_fusion(mode)
        reg.fill_random(1)
        for _ in range(3):
            reg.timer_start(); qc.inverse_QFT(reg); ms = reg.timer_stop()
        print(f"iqft28 mode {mode}: {ms:.3f} ms", flush=True)
with qc.Register(25, 5) as reg:
    for mode in (0, 2):
        reg.set_fusion(mode)
        for _ in range(2):
            reg.timer_start(); qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.flush(); ms = reg.timer_stop()
        print(f"shor30 mode {mode}: {ms:.3f} ms", flush=True)
