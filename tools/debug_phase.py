import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import quantumcomputer_amd as qc
from oracle import binding as ob
def check(n, c, t, th, **tune):
    if tune: qc.tune(**tune)
    a = ob.random_state(n, 11)
    with qc.Register(n - 5, 5) as reg:
        reg.write(a); qc.c_phase_shift_gate(c, t, th, reg); got = reg.read()
    w = a.copy(); ob.cphase(w, n, c, t, th)
    d = got.view(np.uint64) != w.view(np.uint64)
    return int(d.sum()), (np.nonzero(d)[0][:4] // 2).tolist(), float(np.abs(got - w).max())
th = 0.20966817126512538
print("n=18 (6,2):", check(18, 6, 2, th))
print("n=18 (6,2) s0:", check(18, 6, 2, th, ph_streams_log2=0))
print("n=18 (6,2) blk256:", check(18, 6, 2, th, ph_streams_log2=-1, ph_block=256, ph_apt=4))
qc.tune(ph_block=64, ph_apt=1)
print("n=18 (7,2):", check(18, 7, 2, th), " (6,3):", check(18, 6, 3, th), " (2,6):", check(18, 2, 6, th))
print("n=15 (6,2):", check(15, 6, 2, th), " n=20 (6,2):", check(20, 6, 2, th))
print("theta pi/8:", check(18, 6, 2, 0.39269908169872414))
