"""one period-finding attempt (reset + circuit + measurement) at the reference's own sizes, 300 attempts each: wall time per
attempt; run under rocprofv3 --kernel-trace to see the launches behind it"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc
for (L, M, Cn, a) in ((3, 4, 15, 7), (5, 5, 21, 2), (8, 4, 15, 7)):
    rng = qc.Rng(1)
    with qc.Register(L, M) as reg:
        for _ in range(20):
            qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); qc.measure_state(reg, rng)
        reg.synchronize(); t0 = time.perf_counter()
        N = 300
        for _ in range(N):
            qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); qc.measure_state(reg, rng)
        dt = (time.perf_counter() - t0) / N
        print(f"n={L + M}: {dt * 1e6:.1f} us per attempt", flush=True)
