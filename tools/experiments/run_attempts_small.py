#!/usr/bin/env python3
"""60 period-finding attempts at n = 16 and n = 20 (M = 5, N = 21) -- a fixed launch order for a rocprofv3 kernel trace
(tools/experiments/trace_gaps.py prints the kernels of one attempt with their durations and the gaps between them)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

for L in (11, 15):
    with qc.Register(L, 5) as reg:
        for k in range(60):
            qc.reset_register(reg); qc.quantum_computation(21, 2, reg); qc.measure_state(reg, 0.05 + 0.9 * k / 60)
