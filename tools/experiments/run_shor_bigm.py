#!/usr/bin/env python3
"""n = 30 Shor circuit with a large M register (C = 8191, M = 13, L = 17), exact mode x 3 -- a fixed launch order for tools/trace_seq.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

if len(sys.argv) > 1:
    qc.tune(**{k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1:])})
with qc.Register(17, 13) as reg:
    reg.set_fusion(0)
    for _ in range(3):
        reg.timer_start(); qc.reset_register(reg); qc.quantum_computation(8191, 3, reg); reg.flush(); ms = reg.timer_stop()
    print(f"shor M=13 exact: {ms:.3f} ms", flush=True)
