#!/usr/bin/env python3
"""n = 28 inverse QFT four times in the tolerance mode -- a fixed launch order for rocprofv3 passes (tools/trace_seq.py)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumcomputer_amd as qc  # noqa: E402

if len(sys.argv) > 1:
    qc.tune(**{k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1:])})
with qc.Register(28, 0) as reg:
    reg.set_fusion(2)
    reg.fill_random(1)
    for _ in range(4):
        reg.timer_start(); qc.inverse_QFT(reg); ms = reg.timer_stop()
    print(f"iqft28 tolerance: {ms:.3f} ms", flush=True)
