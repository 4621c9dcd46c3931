#!/usr/bin/env python3
"""How fast do the gate kernels run on a state that fits the 256-MiB Infinity Cache?  Per-gate Hadamards and the fused
sweep at n = 20 .. 27 (16 MiB .. 2 GiB): if cache-resident passes run well above the HBM rate, a sweep at n = 30 could be
blocked into cache-sized chunks (several tile passes per HBM round trip)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

for n in (20, 21, 22, 23, 24, 25, 26, 27):
    with qc.Register(n, 0) as reg:
        reg.fill_random(1)
        bytes_gate = 32.0 * (1 << n)
        for q in (4, 10, n - 1):
            for _ in range(3):
                qc.hadamard_gate(q, reg)
            reps = 40
            reg.timer_start()
            for _ in range(reps):
                qc.hadamard_gate(q, reg)
            ms = reg.timer_stop() / reps
            print(f"n={n} H({q:2d}) per gate: {ms * 1e3:8.1f} us  {bytes_gate / ms / 1e6:8.0f} GB/s", flush=True)
        reg.set_fusion(1)
        for _ in range(2):
            for q in range(n):
                qc.hadamard_gate(q, reg)
        reg.synchronize()
        p0 = reg.fusion_stats()[0]
        reps = 10
        reg.timer_start()
        for _ in range(reps):
            for q in range(n):
                qc.hadamard_gate(q, reg)
            reg.flush()
        ms = reg.timer_stop() / reps
        passes = (reg.fusion_stats()[0] - p0) / reps
        print(f"n={n} fused sweep: {ms:8.3f} ms  {passes:.0f} passes  {passes * bytes_gate / ms / 1e6:8.0f} GB/s per pass", flush=True)
