"""GPU: BASELINE.json's configs at their FULL sizes, every one a one-GPU box can hold.

  config 2   n = 26 Hadamard sweep, whole 1-GiB state against the oracle, one launch per gate and as fused passes
  config 3   tolerance mode with a DENSE input against the whole-state oracle (n = 26), next to the exact default
  config 4   n = 32 (64 GiB) sharded over 8 shards -- virtual shards on the visible GPUs: on a one-GPU box all eight sit on
             device 0, the code path of an 8-GPU node except that the peer stores stay on the device -- H on the three
             global qubits (one at a time and all three behind ONE exchange) and on a local one, windows against the
             oracle's twin of the synthetic input
  config 5   n = 30 Shor N = 21 (L = 25, M = 5) on 8 shards: the circuit front against the oracle's per-index chains, the
             whole circuit against the unsharded register AND the oracle's final state (windows, norm, measured index), exact and
             tolerance modes; and the WHOLE n = 28 / n = 30 final state of the unsharded register against the oracle

Reference: qc_shor.c:442-484 (hadamard_gate), 678-690 (inverse_QFT), 712-737 (quantum_computation), 272-306 (measure_state).
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W = 13
THREADS = min(16, os.cpu_count() or 1)
TOL = 1e-12


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def max_delta(a, b):
    d = np.asarray(a) - np.asarray(b)
    return float(np.max(np.hypot(d[0::2], d[1::2])))


def oracle_pick(ob, want, n, r):
    """the index Q:283-292 selects for the draw r, WITHOUT the collapse (ob.measure overwrites its argument: the n = 30
    oracle state is shared between tests)"""
    hit, idx, _ = ob.measure_range(want, 0, 1 << n, (1 << n) - 1, 0.0, r)
    return idx if hit else (1 << n) - 1


def chunks_equal(reg, want, n, chunk_log=24):
    """compare the whole register with `want` (interleaved float64) chunk by chunk: the read-back buffer stays small"""
    step = 1 << min(chunk_log, n)
    for s in range(0, 1 << n, step):
        if not np.array_equal(bits(reg.read(s, step)), bits(want[2 * s:2 * (s + step)])):
            return s
    return -1


# ---- config 2: n = 26 Hadamard sweep, every amplitude ------------------------------------------------------------------
@pytest.mark.parametrize("mode", [-1, 1], ids=["one launch per gate", "fused passes"])
def test_config2_n26_sweep_whole_state(qc, ob, mode):
    n = 26
    want = ob.fill_random(n, 26)
    for q in range(n):
        ob.hadamard(want, n, q, THREADS)
    with qc.Register(n, 0) as reg:
        reg.set_fusion(mode)
        reg.fill_random(26)
        p0 = reg.norm2()
        for q in range(n):
            qc.hadamard_gate(q, reg)                                       # Q:442-484, q = 0 .. 25 in order
        assert abs(reg.norm2() - p0) < 1e-12 * p0
        bad = chunks_equal(reg, want, n)
        assert bad < 0, f"n=26 sweep (mode {mode}): first differing chunk at {bad}"


# ---- config 3 twin: tolerance mode on a dense input against the whole-state oracle ---------------------------------------
@pytest.mark.parametrize("n", [26])
def test_config3_iqft_dense_input_tolerance_vs_oracle(qc, ob, n):
    want = ob.fill_random(n, 33)
    ob.iqft(want, n, 0, THREADS)                                           # Q:678-690 with M = 0
    scale = math.sqrt(6.0 / (1 << n))
    with qc.Register(n, 0) as reg:
        reg.set_fusion(2)
        reg.fill_random(33)
        qc.inverse_QFT(reg)
        worst = 0.0
        step = 1 << 24
        for s in range(0, 1 << n, step):
            worst = max(worst, max_delta(reg.read(s, step), want[2 * s:2 * (s + step)]))
        assert worst <= TOL * scale * 4, worst                              # amplitudes ~ scale; north_star allows 1e-10 absolute
        reg.set_fusion(0)                                                   # and the exact default on the same input: same bits
        reg.fill_random(33)
        qc.inverse_QFT(reg)
        assert chunks_equal(reg, want, n) < 0


# ---- config 4: n = 32 over 8 shards ----------------------------------------------------------------------------------------
N4 = int(os.environ.get("QCX_TEST_N4", "32"))


@pytest.fixture(scope="module")
def reg4(qc):
    reg = qc.Register(N4, 0, shards=8, devices=qc.spread_devices(8))
    yield reg
    reg.close()


def _h_window(ob, n, seed, q, s):
    """(want, reader) for the result of H(q) on fill_random(seed) in the window pair at s"""
    if q < W - 1:
        mini = ob.fill_random(n, seed, s, 1 << W)
        ob.hadamard(mini, W, q)
        return mini, [(s, 1 << W)]
    h = 1 << (W - 1)
    mini = np.concatenate([ob.fill_random(n, seed, s, h), ob.fill_random(n, seed, s | (1 << q), h)])
    ob.hadamard(mini, W, W - 1)
    return mini, [(s, h), (s | (1 << q), h)]


@pytest.mark.parametrize("q", [N4 - 1, N4 - 2, N4 - 3, 17, 2])
def test_config4_n32_hadamard_on_global_and_local_qubits(qc, ob, reg4, q):
    n = N4
    seed = 4000 + q
    rs = np.random.RandomState(q)
    ex0, _ = reg4.sharded_stats()
    reg4.fill_random(seed)
    qc.hadamard_gate(q, reg4)
    half = W - 1
    picks = {0, (1 << (n - half)) - 1} | set(int(x) for x in rs.randint(0, 1 << (n - half), 6))
    for p in sorted(picks):
        s = ((p << half) & ~(1 << q)) if q >= half else ((p << half) & ~((1 << W) - 1))
        want, spans = _h_window(ob, n, seed, q, s)
        got = np.concatenate([reg4.read(a, c) for a, c in spans])
        assert np.array_equal(bits(got), bits(want)), f"n={n} H({q}) window at {s}"
    ex1, _ = reg4.sharded_stats()
    if q >= n - 3:
        assert ex1 > ex0                                                    # the gate crossed the shard boundary: an exchange ran
    reg4.sharded_restore_identity()


def test_config4_n32_three_global_hadamards_behind_one_exchange(qc, ob, reg4):
    """H on q = n-1, n-2, n-3 queued together: the scheduler trades all three shard-id bits at once.  An output window
    depends on the 8 input windows that differ in the top three bits: a 15-qubit oracle register (12 low bits + 3 top)."""
    n = N4
    seed = 4321
    reg4.fill_random(seed)
    ex0, _ = reg4.sharded_stats()
    for q in (n - 1, n - 2, n - 3):
        qc.hadamard_gate(q, reg4)
    reg4.flush()
    ex1, _ = reg4.sharded_stats()
    assert ex1 - ex0 == 1
    h = 1 << (W - 1)
    rs = np.random.RandomState(7)
    for p in sorted({0, (1 << (n - 3 - (W - 1))) - 1} | set(int(x) for x in rs.randint(0, 1 << (n - 3 - (W - 1)), 4))):
        s = p << (W - 1)
        tops = [s | (t << (n - 3)) for t in range(8)]
        mini = np.concatenate([ob.fill_random(n, seed, a, h) for a in tops])
        for b in (W + 1, W, W - 1):                                        # top bit first, as issued
            ob.hadamard(mini, W + 2, b)
        got = np.concatenate([reg4.read(a, h) for a in tops])
        assert np.array_equal(bits(got), bits(mini)), f"window family at {s}"
    reg4.sharded_restore_identity()


def test_config4_n32_sweep_keeps_the_norm(qc, reg4):
    n = N4
    reg4.set_fusion(1)
    reg4.fill_random(99)
    p0 = reg4.norm2()
    for q in range(n):
        qc.hadamard_gate(q, reg4)
    p1 = reg4.norm2()
    assert abs(p1 - p0) < 1e-12 * p0
    reg4.sharded_restore_identity()


# ---- config 5: n = 30 Shor N = 21 on 8 shards -----------------------------------------------------------------------------
@pytest.fixture(scope="module")
def shor30_want(ob):
    """the oracle's final state of the n = 30 Shor N = 21 circuit (OpenMP pairwise form: 375 gates over 16 GiB, ~45 s), built
    once for the sharded tests and the whole-state test below"""
    if os.environ.get("QCX_TEST_SKIP_WHOLE30"):
        return None
    L, M, Cn, a = 25, 5, 21, 2
    n = L + M
    want = np.zeros(2 << n)
    ob.reset(want, n)
    ob.quantum_computation(want, n, M, Cn, a, threads=THREADS)
    return want


@pytest.mark.parametrize("mode", [1, 2], ids=["exact", "tolerance"])
def test_config5_n30_shor_on_8_shards(qc, ob, mode, shor30_want):
    L, M, Cn, a = 25, 5, 21, 2
    n = L + M
    rs = np.random.RandomState(50 + mode)
    wins = sorted({0, (1 << n) - (1 << W)} | {int(v) << W for v in rs.randint(0, 1 << (n - W), 6)})
    with qc.Register(L, M, shards=8, devices=qc.spread_devices(8)) as sh:
        sh.set_fusion(mode)
        # the circuit front (Q:720-731): Hadamard layer + multiply ladder, against the oracle's per-index chain
        qc.reset_register(sh)
        for l in range(M, n):
            qc.hadamard_gate(l, sh)
        atox = a % Cn
        for l in range(M, n):
            qc.c_amodc_gate(Cn, atox, l, sh)
            atox = atox * atox % Cn
        assert abs(sh.norm2() - 1.0) < 1e-12
        for s in wins:
            assert np.array_equal(bits(sh.read(s, 1 << W)), bits(ob.shor_front_window(n, M, Cn, a, s, 1 << W))), (mode, s)
        # the whole circuit next to the unsharded register (itself checked against the oracle below and in test_gpu_fullsize.py)
        with qc.Register(L, M) as one:
            one.set_fusion(mode if mode == 2 else 0)
            for reg in (sh, one):
                qc.reset_register(reg); qc.quantum_computation(Cn, a, reg)
            assert abs(sh.norm2() - 1.0) < 1e-12
            scale = 2.0 ** -(L // 2)
            for s in wins:
                g, w = sh.read(s, 1 << W), one.read(s, 1 << W)
                if mode == 1:
                    assert np.array_equal(bits(g), bits(w)), s
                else:
                    assert max_delta(g, w) <= TOL * scale, s
                if shor30_want is not None:                                   # and against the ORACLE's final state itself
                    o = shor30_want[2 * s:2 * (s + (1 << W))]
                    if mode == 1:
                        assert np.array_equal(bits(g), bits(o)), s
                    else:
                        assert max_delta(g, o) <= TOL * scale, s
            for r in (0.123456789, 0.5, 0.987654321):
                for reg in (sh, one):
                    qc.reset_register(reg); qc.quantum_computation(Cn, a, reg)
                i_sh, i_one = qc.measure_state(sh, r), qc.measure_state(one, r)
                assert i_sh == i_one, (mode, r)
                if shor30_want is not None:
                    assert i_sh == oracle_pick(ob, shor30_want, n, r), (mode, r)
                w = qc.read_omega(i_sh, sh)
                assert min(abs(w - k / 6.0) for k in range(7)) < 2.0 ** -8            # period 6: the draw lands on (the skirt of) a peak at k/6
        ex, _ = sh.sharded_stats()
        assert 1 <= ex                                                       # the inverse QFT's Hadamards on shard-id qubits traded


def test_config5_n30_tolerance_front_vs_oracle(qc, ob):
    """the tolerance mode's n = 30 run checked against the ORACLE where the oracle reaches: the circuit front"""
    L, M, Cn, a = 25, 5, 21, 2
    n = L + M
    rs = np.random.RandomState(77)
    with qc.Register(L, M) as reg:
        reg.set_fusion(2)
        qc.reset_register(reg)
        for l in range(M, n):
            qc.hadamard_gate(l, reg)
        atox = a % Cn
        for l in range(M, n):
            qc.c_amodc_gate(Cn, atox, l, reg)
            atox = atox * atox % Cn
        for s in sorted({0, (1 << n) - (1 << W)} | {int(v) << W for v in rs.randint(0, 1 << (n - W), 6)}):
            assert np.array_equal(bits(reg.read(s, 1 << W)), bits(ob.shor_front_window(n, M, Cn, a, s, 1 << W))), s


@pytest.mark.parametrize("L", [23, 25], ids=["n=28", "n=30"])
def test_config5_shor_whole_final_state_vs_oracle(qc, ob, L, shor30_want):
    """every amplitude of the Shor N = 21 circuit's final state against the oracle (OpenMP pairwise form on the host:
    375 gates over 16 GiB at n = 30), exact mode bit for bit, tolerance mode to 1e-12 of the amplitude scale, and the
    same measured index for the same draw"""
    M, Cn, a = 5, 21, 2
    n = L + M
    if L == 25 and os.environ.get("QCX_TEST_SKIP_WHOLE30"):
        pytest.skip("QCX_TEST_SKIP_WHOLE30")
    if L == 25:
        want = shor30_want
    else:
        want = np.zeros(2 << n)
        ob.reset(want, n)
        ob.quantum_computation(want, n, M, Cn, a, threads=THREADS)
    with qc.Register(L, M) as reg:
        qc.reset_register(reg); qc.quantum_computation(Cn, a, reg)
        assert chunks_equal(reg, want, n) < 0
        reg.set_fusion(2)
        qc.reset_register(reg); qc.quantum_computation(Cn, a, reg)
        worst, step = 0.0, 1 << 24
        for s in range(0, 1 << n, step):
            worst = max(worst, max_delta(reg.read(s, step), want[2 * s:2 * (s + step)]))
        assert worst <= TOL * 2.0 ** -(L // 2), worst
        r = 0.6180339887
        assert qc.measure_state(reg, r) == oracle_pick(ob, want, n, r)
