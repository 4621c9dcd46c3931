"""GPU: the exact parallel measurement scan (K4c) picks the SAME index as the reference's strictly
sequential cumulative sum (oracle), on inputs built to break a naive parallel prefix sum: ties at
half an ulp, binade crossings inside blocks, subnormal partial sums, spikes larger than the running
sum, sparse states, r on and next to partial sums."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(params=[0, 8, 9, 10, 13], ids=lambda b: f"record=2^{b}" if b else "record=auto")
def force_parallel(qc, request):
    """the exact parallel form (K4c: one read of the state, look-back, tree walk) even on small registers, with every record
    size (0: chosen from the register size)"""
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in ("meas_parallel", "meas_min_log2", "meas_block_log")}
    qc.tune(meas_parallel=1, meas_min_log2=10, meas_block_log=request.param)
    yield
    qc.tune(**old)


def last_stats(qc):
    s, b = C.c_uint(0), C.c_uint(0)
    qc.lib().qcx_measure_last_stats(C.byref(s), C.byref(b))
    return s.value, b.value


def check(qc, ob, n, a, rvals):
    a = np.ascontiguousarray(a, dtype=np.float64)
    with qc.Register(n, 0) as reg:
        for r in rvals:
            w = a.copy()
            want = ob.measure(w, n, float(r))
            reg.write(a)
            got = qc.measure_state(reg, float(r))
            assert got == want, f"n={n} r={r!r}: got {got}, want {want}"
            assert np.array_equal(bits(reg.read()), bits(w))


def partial_sum_rs(a, rs, k=8):
    p = (a.reshape(-1, 2) ** 2).sum(axis=1)
    cum = np.cumsum(p)              # not the sequential roundings, but lands r next to real partial sums
    picks = rs.randint(0, cum.size, k)
    out = []
    for i in picks:
        out += [float(cum[i]), float(np.nextafter(cum[i], 0.0)), float(np.nextafter(cum[i], 2.0))]
    return out


@pytest.mark.parametrize("n", [14, 17, 20])
def test_random_dense_states(qc, ob, force_parallel, n):
    rs = np.random.RandomState(n)
    a = ob.random_state(n, 40 + n)
    check(qc, ob, n, a, [0.0, 1.0, 0.5, 1e-9, 0.999999999] + list(rs.uniform(0, 1, 10)) + partial_sum_rs(a, rs))
    slow, blocks = last_stats(qc)
    blog = qc.lib().qcx_tune_get(b"meas_block_log") or min(13, max(8, (n - 1) // 2))     # auto: from the register size
    blog = min(blog, 11)                            # a record is one wave's 2^8 .. 2^11 amplitudes
    assert blocks == -(-((1 << n) - 1) // (1 << blog)) and slow <= 80


def test_uniform_superposition_no_rounding(qc, ob, force_parallel):
    n = 18
    a = np.zeros(2 << n); a[0::2] = 2.0 ** (-n / 2)
    k = [0, 1, 2, 1000, (1 << n) - 2, (1 << n) - 1]
    check(qc, ob, n, a, [x / float(1 << n) for x in k] + [0.3, 0.7, 1.0, 1.5])


def test_sparse_states(qc, ob, force_parallel):
    n = 18
    a = np.zeros(2 << n); a[2] = 1.0                               # the reset state
    check(qc, ob, n, a, [0.0, 0.3, 1.0, 1.1])
    b = np.zeros(2 << n); b[2 * 200001] = 0.6; b[2 * 200001 + 1] = 0.8   # one amplitude far inside
    check(qc, ob, n, b, [1e-300, 0.3, 1.0])
    c = np.zeros(2 << n)                                           # all zero: falls through to the last index
    check(qc, ob, n, c, [0.5])
    d = np.zeros(2 << n); d[-2] = 1.0                              # weight only on the excluded last index
    check(qc, ob, n, d, [0.5, 0.0])


def test_half_ulp_ties_round_to_even(qc, ob, force_parallel):
    """after p0 = 1 every further p = 2^-53 is exactly half an ulp: the sequential sum never moves,
    any pairwise/tree sum would; then odd multiples (3 * 2^-53) alternate up/down"""
    n = 16
    a = np.zeros(2 << n); a[0] = 1.0
    a[2::2] = 2.0 ** -27; a[3::2] = 2.0 ** -27                     # p = 2^-54 + 2^-54 = 2^-53
    check(qc, ob, n, a, [1.0, 1.0 + 2.0 ** -52, 1.0 + 2.0 ** -40, 0.5])
    b = a.copy(); b[2::2] = 2.0 ** -26; b[3::2] = 2.0 ** -27 * math.sqrt(2)   # mixed: mostly non-ties, some ties
    check(qc, ob, n, b, [1.0 + 2.0 ** -45, 1.0 + 2.0 ** -38, 1.00000001])
    c = np.zeros(2 << n); c[0] = 1.0
    c[2::4] = 2.0 ** -27; c[3::4] = 2.0 ** -27                     # ties on every other element only
    c[4::4] = 2.0 ** -26
    check(qc, ob, n, c, [1.0 + 2.0 ** -44, 1.0 + 2.0 ** -41, 1.0 + 2.0 ** -39])


def test_binade_crossings_inside_blocks(qc, ob, force_parallel):
    n = 16
    i = np.arange(1 << n, dtype=np.float64)
    a = np.zeros(2 << n)
    a[0::2] = 2.0 ** (-30 + i / 4096.0)                            # p grows 2x every 2048 elements
    tot = float(((a[0::2]) ** 2).sum())
    check(qc, ob, n, a, [tot * f for f in (1e-12, 1e-6, 0.01, 0.3, 0.9, 0.999999, 1.0, 1.01)])
    rs = np.random.RandomState(3)
    b = ob.random_state(n, 9) * np.repeat(10.0 ** rs.uniform(-9, 0, 1 << n), 2)   # wild dynamic range
    tb = float((b ** 2).sum())
    check(qc, ob, n, b, [tb * f for f in (1e-15, 1e-9, 1e-3, 0.2, 0.5, 0.99)])


def test_subnormal_start_and_spikes(qc, ob, force_parallel):
    n = 15
    a = ob.random_state(n, 13)
    a[0:64] = 1e-160                                                # p = 2e-320: subnormal partial sums
    a[2 * 5000] = 0.9                                               # a spike larger than the running sum
    a[2 * 20000 + 1] = -0.7
    check(qc, ob, n, a, [1e-322, 4e-320, 1e-300, 0.05, 0.5, 0.81, 1.2, 1.4, 5.0])


@pytest.mark.parametrize("knobs", [dict(meas_fast=0), dict(meas_dbg=2), dict(meas_fast=1)], ids=["walk-alone", "hand-over", "fast"])
@pytest.mark.parametrize("blog", [8, 11])
def test_event_list_and_walk_agree_with_the_oracle(qc, ob, knobs, blog):
    """k_meas_fast (the events of the scan from the candidate list), the tree walk alone, and the hand-over from one to the other
    in the middle of a scan (meas_dbg bit 1: at the third candidate) all pick the reference's index"""
    keys = ("meas_parallel", "meas_min_log2", "meas_block_log", "meas_fast", "meas_dbg")
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in keys}
    qc.tune(meas_parallel=1, meas_min_log2=10, meas_block_log=blog, **knobs)
    try:
        n = 17
        rs = np.random.RandomState(blog)
        a = ob.random_state(n, 77)
        check(qc, ob, n, a, [0.0, 1.0, 0.5, 1e-9, 3e-6, 0.999999999] + list(rs.uniform(0, 1, 6)) + partial_sum_rs(a, rs, 4))
        if knobs.get("meas_fast", 1) and not knobs.get("meas_dbg"):
            slow, blocks = last_stats(qc)
            assert blocks == (1 << n) >> blog and 0 < slow <= 80, (slow, blocks)
        i = np.arange(1 << n, dtype=np.float64)
        b = np.zeros(2 << n)
        b[0::2] = 2.0 ** (-30 + i / 8192.0)                            # p doubles every 4096 elements: crossings inside records
        tot = float(((b[0::2]) ** 2).sum())
        check(qc, ob, n, b, [tot * f for f in (1e-12, 1e-6, 0.01, 0.3, 0.9, 0.999999, 1.0, 1.01)])
        c = np.zeros(2 << n); c[2 * 70000] = 0.6; c[2 * 70000 + 1] = 0.8   # sparse: one amplitude far inside
        check(qc, ob, n, c, [1e-300, 0.3, 1.0])
        d = ob.random_state(n, 5)
        d[0:64] = 1e-160; d[2 * 5000] = 0.9; d[2 * 90000 + 1] = -0.7      # subnormal start, spikes larger than the running sum
        check(qc, ob, n, d, [1e-322, 4e-320, 0.05, 0.5, 0.81, 1.2, 1.4, 5.0])
    finally:
        qc.tune(**old)


@pytest.mark.parametrize("n", [24, 27])
def test_parallel_equals_sequential_scan_on_the_gpu(qc, n):
    """larger than the CPU comfortably checks: the single-wave sequential kernel is the arbiter"""
    rs = np.random.RandomState(n)
    with qc.Register(n, 0) as reg:
        for r in [0.25, 0.9999] + list(rs.uniform(0, 1, 2)):
            reg.fill_random(50 + n)
            qc.tune(meas_parallel=0)
            want = qc.measure_state(reg, float(r))
            reg.fill_random(50 + n)
            qc.tune(meas_parallel=1)
            got = qc.measure_state(reg, float(r))
            assert got == want
    slow, blocks = last_stats(qc)
    assert slow <= 120, (slow, blocks)


def test_shor_state_n30_measurement_is_fast_and_periodic(qc):
    import time
    L, M = 25, 5
    rng = qc.Rng(12345)
    with qc.Register(L, M) as reg:
        qc.reset_register(reg); qc.quantum_computation(21, 2, reg); reg.synchronize()
        t0 = time.perf_counter()
        idx = qc.measure_state(reg, rng)
        dt = time.perf_counter() - t0
        w = qc.read_omega(idx, reg)
    assert min(abs(w - k / 6.0) for k in range(7)) < 2.0 ** -20
    assert dt < 2.0, f"measurement took {dt:.2f} s"
    print("n=30 measure seconds:", dt, "slow/blocks:", last_stats(qc))


@pytest.mark.parametrize("n", [3, 9, 13, 18, 22])
def test_total_probability_is_the_references_sequential_sum(qc, ob, n):
    """check_normalisation (testing_and_debug.c:28-37) adds |amp|^2 in index order; qcx_total_probability returns that
    sum bit for bit (the exact measurement scan run to the end), on a plain and on a sharded register"""
    with qc.Register(n, 0) as reg:
        reg.fill_random(n)
        for q in (0, n - 1):
            qc.hadamard_gate(q, reg)
        want = ob.fill_random(n, n)
        for q in (0, n - 1):
            ob.hadamard(want, n, q)
        assert reg.total_probability() == ob.norm2(want, n)
        assert abs(reg.norm2() - ob.norm2(want, n)) < 1e-13
    if n >= 13:
        with qc.Register(n, 0, shards=4, devices=qc.spread_devices(4)) as reg:
            reg.fill_random(n)
            for q in (0, n - 1):
                qc.hadamard_gate(q, reg)
            assert reg.total_probability() == ob.norm2(want, n)
