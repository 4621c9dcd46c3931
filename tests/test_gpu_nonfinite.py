"""GPU: states that hold Inf / NaN (or components so large that a circuit overflows).  The reference's mat-vec multiplies every
stored triplet out -- the zero imaginary parts of the matrix entries and the 1 of an identity row included (Q:393-413) -- so a
non-finite component poisons the other component of its amplitude at EVERY gate, wherever the gate "does nothing".  A register
that is handed such values switches to the strict gate kernels (K9, csrc/qcx_kernels.h) and must then give what the oracle gives:
the same finite values and infinities bit for bit, a NaN wherever the oracle has a NaN (payload and sign of a NaN are the
hardware's business: x86 makes the negative default NaN, the GPU the positive one)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same_with_nans(got, want, what):
    got = np.ascontiguousarray(got, dtype=np.float64); want = np.ascontiguousarray(want, dtype=np.float64)
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn, wn), f"{what}: NaN pattern differs at {np.nonzero(gn != wn)[0][:8]} ({int(gn.sum())} vs {int(wn.sum())} NaNs)"
    g, w = got[~gn].view(np.uint64), want[~wn].view(np.uint64)
    bad = np.nonzero(g != w)[0]
    assert bad.size == 0, f"{what}: {bad.size} non-NaN doubles differ, first {got[~gn][bad[0]]!r} vs {want[~wn][bad[0]]!r}"


def poisoned_state(ob, n, seed, kinds):
    rs = np.random.RandomState(seed)
    a = ob.random_state(n, 100 + seed)
    where = rs.choice(2 << n, size=len(kinds), replace=False)
    for w, k in zip(where, kinds):
        a[w] = k
    return a


CASES = [
    ("one inf", [math.inf]),
    ("inf, -inf, nan", [math.inf, -math.inf, math.nan]),
    ("nan only", [math.nan]),
    ("huge finite: overflows on the way", [1e308, -1.7e308, 8e307]),
    ("mixed", [math.inf, math.nan, 1e300, -math.inf, 1e-320, -0.0]),
]


@pytest.mark.parametrize("name,kinds", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("L,M,C,a", [(7, 4, 15, 7), (8, 5, 21, 2), (11, 0, 1, 1), (6, 6, 35, 2)])
def test_gates_on_non_finite_states_match_the_oracle(qc, ob, L, M, C, a, name, kinds):
    n = L + M
    state = poisoned_state(ob, n, n * 7 + len(kinds), kinds)
    rs = np.random.RandomState(n + len(kinds))
    for fusion in (0, 1, -1, 2):                   # whatever the mode: a poisoned register runs strict passes
        want = state.copy()
        with qc.Register(L, M) as reg:
            reg.set_fusion(fusion)
            reg.write(state)
            same_with_nans(reg.read(), want, f"{name}: write/read")
            # single gates of every kind, on every kind of qubit
            for q in (0, n - 1, int(rs.randint(0, n))):
                qc.hadamard_gate(q, reg); ob.hadamard(want, n, q)
            same_with_nans(reg.read(), want, f"{name}: after Hadamards (fusion {fusion})")
            for _ in range(4):
                c, t = (int(v) for v in rs.choice(n, 2, replace=False))
                th = float(rs.uniform(-3, 3))
                qc.c_phase_shift_gate(c, t, th, reg); ob.cphase(want, n, c, t, th)
            same_with_nans(reg.read(), want, f"{name}: after phases (fusion {fusion})")
            if M:
                for ctl in (M, n - 1, 0 if M > 1 else M):      # above the M register, at the top, INSIDE the M register
                    x = int(rs.randint(1, 4 * C))
                    qc.c_amodc_gate(C, x, ctl, reg); ob.camodc(want, n, M, C, x, ctl)
                same_with_nans(reg.read(), want, f"{name}: after modular multiplies (fusion {fusion})")
                # the whole-circuit calls
                qc.inverse_QFT(reg); ob.iqft(want, n, M)
                same_with_nans(reg.read(), want, f"{name}: after inverse_QFT (fusion {fusion})")
                qc.quantum_computation(C, a, reg); ob.quantum_computation(want, n, M, C, a)
                same_with_nans(reg.read(), want, f"{name}: after quantum_computation (fusion {fusion})")
            else:
                qc.inverse_QFT(reg); ob.iqft(want, n, M)
                same_with_nans(reg.read(), want, f"{name}: after inverse_QFT (fusion {fusion})")
            # measurement: the reference's running sum turns NaN / Inf and the comparison with r decides as IEEE says
            for r in (0.3, 0.999):
                w2 = want.copy()
                reg.write(want)
                got = qc.measure_state(reg, r)
                assert got == ob.measure(w2, n, r), (name, fusion, r)
                same_with_nans(reg.read(), w2, f"{name}: collapsed state")


def test_a_reset_or_a_collapse_ends_the_strict_mode(qc, ob):
    """the strict passes are for poisoned states only: after reset_register (or a measurement's collapse, or fill_random) the
    register is back on the fused engine -- same bits as a register that never saw a NaN -- and a later finite write keeps it there"""
    L, M, C, a = 9, 5, 21, 2
    n = L + M
    bad = poisoned_state(ob, n, 5, [math.nan, math.inf])
    with qc.Register(L, M) as reg, qc.Register(L, M) as clean:
        reg.write(bad)
        qc.hadamard_gate(3, reg)
        p0 = reg.fusion_stats()[0]
        for r_ in (reg, clean):
            qc.reset_register(r_); qc.quantum_computation(C, a, r_)
        assert np.array_equal(reg.read().view(np.uint64), clean.read().view(np.uint64))
        assert reg.fusion_stats()[0] > p0, "fused passes ran again"
        want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a)
        assert np.array_equal(reg.read().view(np.uint64), want.view(np.uint64))
        # poisoned again, then a measurement: the collapsed state is finite, the next circuit is fused and exact
        reg.write(bad)
        qc.hadamard_gate(0, reg)
        w = bad.copy(); ob.hadamard(w, n, 0)
        idx = qc.measure_state(reg, 0.5)
        assert idx == ob.measure(w, n, 0.5)
        for l in range(M, n):
            qc.hadamard_gate(l, reg); ob.hadamard(w, n, l)
        qc.inverse_QFT(reg); ob.iqft(w, n, M)
        assert np.array_equal(reg.read().view(np.uint64), w.view(np.uint64))
        # a finite state written over everything
        fin = ob.random_state(n, 77)
        reg.fill_random(1); reg.write(fin)
        qc.inverse_QFT(reg); w3 = fin.copy(); ob.iqft(w3, n, M)
        assert np.array_equal(reg.read().view(np.uint64), w3.view(np.uint64))
