#!/usr/bin/env python3
"""one period-finding attempt (reset + quantum_computation + measure_state), wall clock in a warm process, with the scan's result
written straight into pinned host memory (meas_host_out = 1) and with the copy back on the stream (0); same measured indices"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

for (C, L, M, a) in ((15, 3, 4, 7), (15, 8, 4, 7), (21, 11, 5, 2), (21, 15, 5, 2), (21, 19, 5, 2), (21, 25, 5, 2)):
    row, idx = [], []
    for ho in (1, 0, 1, 0):
        qc.tune(meas_host_out=ho)
        with qc.Register(L, M) as reg:
            reps = 400 if L < 17 else (40 if L < 23 else 10)
            def attempt(r):
                qc.reset_register(reg); qc.quantum_computation(C, a, reg); return qc.measure_state(reg, r)
            attempt(0.3); attempt(0.6)
            reg.synchronize()
            t0 = time.perf_counter()
            got = [attempt(0.05 + 0.9 * k / reps) for k in range(reps)]
            row.append((time.perf_counter() - t0) / reps * 1e6)
            idx.append(got)
    print(f"n={L + M:2d}: host_out=1 {row[0]:9.1f} {row[2]:9.1f} us   copy back {row[1]:9.1f} {row[3]:9.1f} us   {'same indices' if all(i == idx[0] for i in idx) else 'DIFFERENT INDICES'}", flush=True)
