"""latency of one period-finding attempt (reset + circuit + measurement) at small register sizes
   usage: python tools/probe_shots.py [meas_min_log2 ...]"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import quantumcomputer_amd as qc
thresholds = [int(x) for x in sys.argv[1:]] or [qc.lib().qcx_tune_get(b"meas_min_log2")]
for th in thresholds:
    qc.tune(meas_min_log2=th)
    for (L, M, Cn, a) in ((3, 4, 15, 7), (8, 4, 15, 7), (8, 5, 21, 2), (9, 5, 21, 2), (10, 5, 21, 2), (11, 5, 21, 2), (13, 5, 21, 2), (15, 5, 21, 2)):
        rng = qc.Rng(1)
        with qc.Register(L, M) as reg:
            for _ in range(20):
                qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); qc.measure_state(reg, rng)
            reg.synchronize(); t0 = time.perf_counter()
            N = 200
            for _ in range(N):
                qc.reset_register(reg); qc.quantum_computation(Cn, a, reg)
            reg.synchronize(); tc = (time.perf_counter() - t0) / N
            t0 = time.perf_counter()
            for _ in range(N):
                qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); qc.measure_state(reg, rng)
            dt = (time.perf_counter() - t0) / N
            print(f"meas_min_log2={th:2d} n={L+M:2d}: {dt*1e6:8.1f} us per attempt, of which measurement {(dt-tc)*1e6:8.1f} us", flush=True)
