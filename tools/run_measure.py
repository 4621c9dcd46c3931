#!/usr/bin/env python3
"""one measurement of an n=30 Shor state (profiling target)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumcomputer_amd as qc
with qc.Register(25, 5) as reg:
    reg.set_fusion(True)
    rng = qc.Rng(12345)
    for _ in range(2):
        qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.synchronize()
        t0 = time.perf_counter(); idx = qc.measure_state(reg, rng); dt = time.perf_counter() - t0
        print("measure n=30: %.2f ms -> %d" % (dt * 1e3, idx))
