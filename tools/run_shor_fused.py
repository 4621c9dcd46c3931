#!/usr/bin/env python3
"""one fused Shor circuit at n=30 (profiling target)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quantumcomputer_amd as qc
with qc.Register(25, 5) as reg:
    reg.set_fusion(True)
    for _ in range(2):
        qc.reset_register(reg); reg.synchronize(); t0 = time.perf_counter()
        qc.quantum_computation(21, 2, reg); reg.synchronize()
        print("fused Shor n=30: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
