import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import quantumcomputer_amd as qc
from oracle import binding as ob
from test_gpu_fusion import random_program, run_both, bits
L, M, Cn, seed = 13, 5, 21, 11
rs = np.random.RandomState(seed * 7 + L)
prog = random_program(rs, L + M, M, Cn, 90)[:13]
print(prog)
def ok(p):
    got, want, st = run_both(qc, ob, L, M, Cn, p, 11)
    return np.array_equal(bits(got), bits(want)), st
print("full:", ok(prog))
# minimise: try removing each gate
cur = list(prog)
changed = True
while changed:
    changed = False
    for i in range(len(cur)):
        trial = cur[:i] + cur[i+1:]
        if trial and not ok(trial)[0]:
            cur = trial; changed = True; break
print("minimal failing:", cur, ok(cur))
for T, c in ((11, 4), (12, 4), (10, 4), (11, 5), (11, 6), (9, 4)):
    qc.tune(fuse_T=T, fuse_c=c)
    print("T,c", T, c, ok(cur))
