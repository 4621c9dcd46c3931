import sys, time
sys.path.insert(0, '/root/repo')
import quantumcomputer_amd as qc
with qc.Register(25, 5) as reg:
    for mode in (0, 2):
        reg.set_fusion(mode)
        for rep in range(4):
            reg.synchronize()
            t0 = time.perf_counter()
            qc.reset_register(reg); qc.quantum_computation(21, 2, reg)
            t1 = time.perf_counter()
            idx = qc.measure_state(reg, 0.37)
            t2 = time.perf_counter()
        print(f"mode {mode}: host time of reset + quantum_computation (asynchronous) {1e3*(t1-t0):.3f} ms; measure_state returns after {1e3*(t2-t1):.3f} ms more", flush=True)
