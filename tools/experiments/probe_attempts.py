#!/usr/bin/env python3
"""one period-finding attempt (reset + quantum_computation(21, 2) + measure_state) at L = 6 .. 25, M = 5: wall-clock per attempt in a
warm process, with and without the plan cache"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

M, C, a = 5, 21, 2
extra = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1:])}
for L in (6, 9, 11, 13, 15, 17, 19, 21, 23, 25):
    row = []
    for cache in (1, 0):
        qc.tune(fuse_plan_cache=cache, **extra)
        with qc.Register(L, M) as reg:
            reps = 200 if L < 17 else (40 if L < 23 else 10)
            def attempt(r):
                qc.reset_register(reg); qc.quantum_computation(C, a, reg); return qc.measure_state(reg, r)
            attempt(0.3); attempt(0.6)
            reg.synchronize()
            t0 = time.perf_counter()
            for k in range(reps):
                attempt(0.05 + 0.9 * k / reps)
            row.append((time.perf_counter() - t0) / reps * 1e6)
    print(f"n={L + M:2d}: {row[0]:10.1f} us per attempt with the plan cache, {row[1]:10.1f} us without", flush=True)
