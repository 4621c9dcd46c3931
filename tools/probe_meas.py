import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import quantumcomputer_amd as qc
for (L, M) in ((9, 5), (11, 5), (15, 5)):
    rng = qc.Rng(1)
    with qc.Register(L, M) as reg:
        qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.synchronize()
        st = reg.read()
        t0 = time.perf_counter()
        for _ in range(200):
            reg.write(st)          # restore (H2D copy) then measure
            qc.measure_state(reg, rng)
        t1 = time.perf_counter()
        for _ in range(200):
            reg.write(st)
        t2 = time.perf_counter()
        print(f"n={L+M}: measure {((t1-t0)-(t2-t1))/200*1e6:.1f} us", flush=True)
