#!/usr/bin/env python3
"""controlled modular multiply with an M register beyond the LDS tile (M > 12, K3b: staged in place) at n = 30: HIP-event times,
rate on the bytes the path moves (64 B per rewritten amplitude) and on SURVEY s8(d)'s algorithmic 32 * 2^(n-1) B."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

extra = dict(kv.split("=") for kv in sys.argv[1:])
n = int(extra.pop("n", 30))
if extra:
    qc.tune(**{k: int(v) for k, v in extra.items()})
for M, C in ((5, 21), (9, 509), (10, 1021), (11, 2039), (12, 4093), (13, 8191), (14, 16381), (16, 65521), (18, 262139), (20, 1048573)):
    L = n - M
    with qc.Register(L, M) as reg:
        reg.set_fusion(-1)
        reg.fill_random(3)
        ctl = M + 2
        best = 1e9
        for rep in range(4):
            reg.timer_start(); qc.c_amodc_gate(C, 7, ctl, reg); best = min(best, reg.timer_stop())
        rewritten = (1 << (n - M - 1)) * C
        moved = (64 if M > 12 else 32) * rewritten
        print(f"n={n} M={M:2d} C={C:8d}: {best:7.3f} ms   moved {moved / 1e9:6.2f} GB at {moved / best / 1e6:6.0f} GB/s   "
              f"algorithmic 32*2^(n-1) at {32 * (1 << (n - 1)) / best / 1e6:6.0f} GB/s", flush=True)

# the whole Shor circuit with a large M register: C = 8191 (M = 13), L = n - 13; the front (Hadamard layer + the L multiplies) is
# one write pass (k_basis_front_big), the inverse QFT runs in fused passes
for mode in (0, 2):
    with qc.Register(n - 13, 13) as reg:
        reg.set_fusion(mode)
        best = 1e9
        for rep in range(3):
            reg.timer_start(); qc.reset_register(reg); qc.quantum_computation(8191, 3, reg); reg.flush(); best = min(best, reg.timer_stop())
        print(f"n={n} Shor C=8191 a=3 L={n - 13} M=13 mode {mode}: {best:7.3f} ms  fronts/gates fused {reg.fusion_stats()}  |norm-1| {abs(reg.norm2() - 1):.1e}", flush=True)
