#!/usr/bin/env python3
"""randomised fused-vs-oracle stress (bit-exact); prints failing seeds"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import quantumcomputer_amd as qc
from oracle import binding as ob
from test_gpu_fusion import random_program, run_both, bits
bad = []
for pruns in (1, 0):
    for rounds in (1, 0):
        qc.tune(fuse_T_phase=10 if pruns else 0, fuse_rounds=rounds)
        for (L, M, Cn) in ((16, 4, 15), (13, 5, 21), (12, 0, 1), (14, 6, 35)):
            for seed in range(int(os.environ.get('STRESS_SEEDS', '12'))):
                rs = np.random.RandomState(seed * 7 + L)
                prog = random_program(rs, L + M, M, Cn, 90)
                got, want, _ = run_both(qc, ob, L, M, Cn, prog, 11)
                if not np.array_equal(bits(got), bits(want)):
                    nbad = int((bits(got) != bits(want)).sum())
                    bad.append((pruns, rounds, L, M, seed, nbad))
print("failures:", bad)
if bad:
    pruns, rounds, L, M, seed, _ = bad[0]
    Cn = {4: 15, 5: 21, 0: 1, 6: 35}[M]
    qc.tune(fuse_T_phase=10 if pruns else 0, fuse_rounds=rounds)
    rs = np.random.RandomState(seed * 7 + L)
    prog = random_program(rs, L + M, M, Cn, 90)
    # bisect the program length
    lo, hi = 0, len(prog)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        got, want, _ = run_both(qc, ob, L, M, Cn, prog[:mid], 11)
        if np.array_equal(bits(got), bits(want)): lo = mid
        else: hi = mid
    print("first failing prefix length", hi, "last gates:", prog[max(0, hi - 6):hi])
