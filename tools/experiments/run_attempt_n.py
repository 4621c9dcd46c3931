#!/usr/bin/env python3
"""twenty period-finding attempts at a given n (M = 5, N = 21): a fixed launch order for a kernel trace"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
with qc.Register(n - 5, 5) as reg:
    for k in range(20):
        qc.reset_register(reg); qc.quantum_computation(21, 2, reg); qc.measure_state(reg, 0.05 + 0.04 * k)
