"""CPU: the host logic of the C-ABI sharded register (qcx_register_create_sharded, csrc/qcx_sharded.inc.h) without a GPU.
A dry-run register (devices = [-1]) schedules exactly what a real one would launch -- gate lists per layout, pack passes,
trades, the restoration of the identity layout -- and hands the steps back as text; tests/sharded_replay.py applies
them to a full state vector with the oracle's gates and the result must equal the oracle's own run of the same gate
list in logical order, bit for bit.  (The GPU suite, tests/test_gpu_sharded_c.py, then shows that the kernels do what
the replay does.)"""
import math
import random

import numpy as np
import pytest

from sharded_replay import replay


def run_dry(qc, ob, L, M, shards, program, seed=3):
    n = L + M
    k = shards.bit_length() - 1
    want = ob.fill_random(n, seed)
    got = want.copy()
    with qc.Register(L, M, shards=shards, devices=[-1]) as reg:
        assert reg.shards == shards
        for g in program:
            if g[0] == "h":
                qc.hadamard_gate(g[1], reg); ob.hadamard(want, n, g[1])
            elif g[0] == "p":
                qc.c_phase_shift_gate(g[1], g[2], g[3], reg); ob.cphase(want, n, g[1], g[2], g[3])
            else:
                qc.c_amodc_gate(g[1], g[2], g[3], reg); ob.camodc(want, n, M, g[1], g[2], g[3])
        reg.flush()
        mid_layout = reg.sharded_layout()
        reg.sharded_restore_identity()
        assert reg.sharded_layout() == list(range(n))
        trace = reg.sharded_trace()
        exchanges, packs = reg.sharded_stats()
    got, counts = replay(trace, n, k, M, got, ob)
    assert counts["trade"] == exchanges
    return got, want, counts, mid_layout


@pytest.mark.parametrize("shards,n", [(2, 10), (4, 12), (8, 16)])
def test_cyclic_hadamard_sweeps_cost_one_exchange_each(qc, ob, shards, n):
    prog = [("h", q) for _ in range(3) for q in range(n)]
    got, want, counts, _ = run_dry(qc, ob, n, 0, shards, prog)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    # look-ahead eviction: one trade per sweep (+ at most two to restore the identity layout), not one per global H
    assert 3 <= counts["trade"] <= 3 + 2


@pytest.mark.parametrize("shards,L,M", [(2, 6, 5), (4, 8, 5), (8, 12, 5)])
def test_shor_circuit_schedule(qc, ob, shards, L, M):
    n = L + M
    prog = [("h", l) for l in range(M, n)]
    atox = 2 % 21
    for l in range(M, n):
        prog.append(("c", 21, atox, l)); atox = atox * atox % 21
    for l in range(n - 1, M - 1, -1):
        prog.append(("h", l))
        for kk in range(l - 1, M - 1, -1):
            prog.append(("p", l, kk, math.pi / float(1 << (l - kk))))
    got, want, counts, _ = run_dry(qc, ob, L, M, shards, prog)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert counts["gates"] == len(prog)


@pytest.mark.parametrize("shards,n,M", [(2, 9, 0), (4, 12, 3), (4, 13, 5), (8, 16, 4)])
def test_random_programs(qc, ob, shards, n, M):
    rnd = random.Random(100 * shards + n)
    for trial in range(6):
        prog = []
        for _ in range(rnd.randrange(20, 90)):
            t = rnd.random()
            if t < 0.5:
                prog.append(("h", rnd.randrange(n)))
            elif t < 0.85 or M == 0:
                c, tq = rnd.sample(range(n), 2)
                prog.append(("p", c, tq, rnd.uniform(-3.0, 3.0)))
            else:
                Cn = rnd.randrange(2, (1 << M) + 1)
                prog.append(("c", Cn, rnd.randrange(1, 200), rnd.randrange(M, n)))
        got, want, counts, _ = run_dry(qc, ob, n - M, M, shards, prog, seed=trial)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (shards, n, M, trial)


def test_layout_after_a_global_hadamard_and_argument_checks(qc, ob):
    n, shards = 12, 4
    got, want, counts, layout = run_dry(qc, ob, n, 0, shards, [("h", n - 1)])
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert layout[n - 1] < n - 2 and sorted(layout) == list(range(n))          # the target came local, a permutation stays one
    for bad in (3, 5, 32):
        with pytest.raises(qc.QcxError):
            qc.Register(n, 0, shards=bad, devices=[-1])
    with pytest.raises(qc.QcxError):
        qc.Register(4, 0, shards=8, devices=[-1])                               # too small for that many shards
    with qc.Register(n, 0, shards=2, devices=[-1]) as reg:
        with pytest.raises(qc.QcxError):
            qc.hadamard_gate(n, reg)
        with pytest.raises(qc.QcxError):
            reg.norm2()                                                         # a dry run has no amplitudes: loud, not silent


@pytest.mark.parametrize("slices_log2,overlap", [(3, 1), (2, 1), (1, 1), (0, 1), (3, 0)])
def test_exchange_windows_in_the_schedule(qc, ob, monkeypatch, slices_log2, overlap):
    """with spectator bits a trade sits between a pre-window (gates resolved under the old layout) and a post-window
    (new layout); the replay applies them in trace order -- any mistake in which layout a window is resolved under, or in
    where the trade zone sits below the spectators, breaks the bits"""
    monkeypatch.setenv("QCX_SHARD_SLICES_LOG2", str(slices_log2))
    monkeypatch.setenv("QCX_SHARD_OVERLAP", str(overlap))
    rnd = random.Random(5 + slices_log2)
    n, M, shards = 15, 3, 4
    for trial in range(4):
        prog = []
        for _ in range(100):
            t = rnd.random()
            if t < 0.45:
                prog.append(("h", rnd.choice([n - 1, n - 2, rnd.randrange(n)])))
            elif t < 0.85:
                c, tq = rnd.sample(range(n), 2)
                prog.append(("p", c, tq, rnd.uniform(-3.0, 3.0)))
            else:
                prog.append(("c", rnd.randrange(2, 9), rnd.randrange(1, 50), rnd.randrange(M, n)))
        got, want, counts, _ = run_dry(qc, ob, n - M, M, shards, prog, seed=trial)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (slices_log2, overlap, trial)
        assert counts["trade"] >= 2
