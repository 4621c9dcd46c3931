"""GPU: seeded random programs over all three gate kinds (random angles, random C / a^x, random controls and
targets, every register size from 1 to 18 qubits), per-gate and fused, always the oracle's bits; plus random
measurement draws.  This is the net that caught the sin-vs-sincos last-bit difference."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def program(rs, n, M, length):
    prog = []
    for _ in range(length):
        kind = rs.randint(0, 10)
        if kind < 4 or n < 2:
            prog.append(("h", int(rs.randint(0, n))))
        elif kind < 8 or M == 0 or n - M < 1:
            c, t = rs.choice(n, 2, replace=False)
            theta = float(rs.uniform(-10, 10)) if rs.randint(0, 2) else math.pi / (1 << int(rs.randint(1, 30)))
            prog.append(("p", int(c), int(t), theta))
        else:
            Cn = int(rs.randint(2, (1 << M) + 1))
            prog.append(("c", Cn, int(rs.randint(0, 1 << 20)), int(rs.randint(0, n))))
    return prog


@pytest.mark.parametrize("fusion", [False, True])
@pytest.mark.parametrize("seed", range(24))
def test_random_programs(qc, ob, seed, fusion):
    rs = np.random.RandomState(1000 + seed)
    for _ in range(7):
        n = int(rs.randint(1, 19))
        M = int(rs.randint(0, min(n, 9) + 1))
        L = n - M
        prog = program(rs, n, M, int(rs.randint(5, 70)))
        want = ob.random_state(n, int(rs.randint(1, 10 ** 6)))
        with qc.Register(L, M) as reg:
            reg.write(want)
            reg.set_fusion(fusion)
            for g in prog:
                if g[0] == "h":
                    qc.hadamard_gate(g[1], reg); ob.hadamard(want, n, g[1])
                elif g[0] == "p":
                    qc.c_phase_shift_gate(g[1], g[2], g[3], reg); ob.cphase(want, n, g[1], g[2], g[3])
                else:
                    qc.c_amodc_gate(g[1], g[2], g[3], reg); ob.camodc(want, n, M, g[1], g[2], g[3])
            got = reg.read()
            assert np.array_equal(bits(got), bits(want)), f"seed={seed} n={n} M={M} fusion={fusion} prog={prog}"
            # a measurement on whatever state came out (not normalised in general: gcd(A, C) > 1 sums, C > 2^M)
            total = float((want.reshape(-1, 2) ** 2).sum())
            r = float(rs.uniform(0, 1.05 * total))
            w2 = want.copy()
            assert qc.measure_state(reg, r) == ob.measure(w2, n, r)
            assert np.array_equal(bits(reg.read()), bits(w2))


def qft_like_program(rs, n, M, length):
    """programs dominated by what the tolerance mode and the circuit front act on: Hadamards followed by runs of phases on
    the same qubit (ascending, descending and random target order, repeated targets), stray gates of every kind between"""
    prog = []
    while len(prog) < length:
        kind = rs.randint(0, 10)
        if kind < 6 or M == 0:
            l = int(rs.randint(0, n))
            prog.append(("h", l))
            targets = [k for k in rs.permutation(n)[: int(rs.randint(0, n))] if k != l]
            if rs.randint(0, 3) == 0:
                targets = sorted(targets, reverse=bool(rs.randint(0, 2)))
            for k in targets:
                theta = math.pi / float(1 << abs(l - int(k))) if rs.randint(0, 2) else float(rs.uniform(-3, 3))
                prog.append(("p", l, int(k), theta))
        elif kind < 8:
            c, t = rs.choice(n, 2, replace=False)
            prog.append(("p", int(c), int(t), float(rs.uniform(-3, 3))))
        else:
            Cn = int(rs.randint(2, (1 << M) + 1))
            prog.append(("c", Cn, int(rs.randint(1, 1 << 16)), int(rs.randint(M, n))))
    return prog


@pytest.mark.parametrize("seed", range(16))
def test_random_qft_like_programs_in_every_mode(qc, ob, seed):
    """the same program through modes -1, 0, 1 (bit for bit) and 2 (1e-12), starting from a reset, a measured basis state or a
    dense state: the lazy reset / collapse, the circuit front, the fast rounds and their fall-backs all get traffic"""
    rs = np.random.RandomState(4000 + seed)
    for _ in range(4):
        n = int(rs.randint(6, 19))
        M = int(rs.choice([0, 0, 3, 4, 5]))
        M = min(M, n - 1)
        prog = qft_like_program(rs, n, M, int(rs.randint(10, 120)))
        start = int(rs.randint(0, 3))
        r0 = float(rs.uniform(0, 1))
        states = {}
        for mode in (-1, 0, 1, 2):
            with qc.Register(n - M, M) as reg:
                reg.set_fusion(mode)
                if start == 0:
                    qc.reset_register(reg)
                elif start == 1:
                    reg.fill_random(seed); qc.measure_state(reg, r0)
                else:
                    reg.fill_random(seed)
                for g in prog:
                    if g[0] == "h":
                        qc.hadamard_gate(g[1], reg)
                    elif g[0] == "p":
                        qc.c_phase_shift_gate(g[1], g[2], g[3], reg)
                    else:
                        qc.c_amodc_gate(g[1], g[2], g[3], reg)
                states[mode] = reg.read()
        if start == 0:
            want = np.zeros(2 << n); ob.reset(want, n)
        else:
            want = ob.fill_random(n, seed)
            if start == 1:
                ob.measure(want, n, r0)
        for g in prog:
            if g[0] == "h":
                ob.hadamard(want, n, g[1])
            elif g[0] == "p":
                ob.cphase(want, n, g[1], g[2], g[3])
            else:
                ob.camodc(want, n, M, g[1], g[2], g[3])
        for mode in (-1, 0, 1):
            assert np.array_equal(bits(states[mode]), bits(want)), f"seed={seed} n={n} M={M} mode={mode} start={start}"
        d = states[2] - want
        scale = max(1.0, float(np.max(np.abs(want))))
        assert float(np.max(np.hypot(d[0::2], d[1::2]))) <= 1e-12 * scale, f"seed={seed} n={n} M={M} start={start}"
