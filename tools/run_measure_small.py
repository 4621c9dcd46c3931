import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import quantumcomputer_amd as qc
rng = qc.Rng(1)
with qc.Register(13, 5) as reg:
    for _ in range(60):
        qc.reset_register(reg); qc.quantum_computation(21, 2, reg); qc.measure_state(reg, rng)
