#!/usr/bin/env python3
"""one fused IQFT at n qubits (profiling target)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
with qc.Register(n, 0) as reg:
    reg.fill_random(1); reg.set_fusion(True)
    for _ in range(2):
        reg.synchronize(); t0 = time.perf_counter(); qc.inverse_QFT(reg); reg.synchronize()
        print("fused IQFT n=%d: %.2f ms" % (n, (time.perf_counter() - t0) * 1e3))
