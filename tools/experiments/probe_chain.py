#!/usr/bin/env python3
"""fused workloads with and without CHAINED passes (qcx_tune fuse_chain): n = 30 Hadamard sweep, n = 28 inverse QFT (exact
and tolerance mode), n = 30 Shor N = 21 circuit (exact and tolerance mode); HIP-event times, best of a few."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

extra = dict(kv.split("=") for kv in sys.argv[1:])
if extra:
    qc.tune(**{k: int(v) for k, v in extra.items()})


def best(reg, fn, reps=4):
    fn(); reg.synchronize()
    b = 1e9
    for _ in range(reps):
        reg.timer_start(); fn(); b = min(b, reg.timer_stop())
    return b


for chain in (1,):
    qc.tune(fuse_chain=chain)
    with qc.Register(30, 0) as reg:
        reg.set_fusion(1); reg.fill_random(1)
        def sweep():
            for q in range(30):
                qc.hadamard_gate(q, reg)
            reg.flush()
        t = best(reg, sweep)
        print(f"chain={chain} n=30 fused sweep       : {t:7.3f} ms", flush=True)
    with qc.Register(28, 0) as reg:
        for mode in (0, 2):
            reg.set_fusion(mode); reg.fill_random(1)
            t = best(reg, lambda: qc.inverse_QFT(reg))
            print(f"chain={chain} n=28 inverse QFT mode {mode} : {t:7.3f} ms", flush=True)
    with qc.Register(25, 5) as reg:
        for mode in (0, 2):
            reg.set_fusion(mode)
            def shor():
                qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.flush()
            t = best(reg, shor)
            print(f"chain={chain} n=30 Shor circuit mode {mode}: {t:7.3f} ms", flush=True)
