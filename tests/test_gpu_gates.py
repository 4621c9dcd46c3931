"""GPU parity tests proper: every gate of the hot path through the C ABI (libqcx.so) against the
CPU oracle on the same seeded inputs, BIT-EXACT (uint64 views of the doubles are compared, so even
the sign of a zero has to agree)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bits_equal(got, want, what=""):
    g, w = bits(got), bits(want)
    if not np.array_equal(g, w):
        bad = np.nonzero(g != w)[0]
        raise AssertionError(f"{what}: {bad.size}/{g.size} doubles differ; first at {bad[0]}: "
                             f"got {got[bad[0]]!r} want {want[bad[0]]!r}")


def sparse_random_state(ob, n, seed):
    """random state with exact zeros and repeated magnitudes sprinkled in (cancellation -> +0 cases)"""
    a = ob.random_state(n, seed)
    rs = np.random.RandomState(seed)
    v = a.reshape(-1, 2)
    k = max(1, v.shape[0] // 4)
    v[rs.randint(0, v.shape[0], k)] = 0.0
    if v.shape[0] >= 4:
        v[1] = v[0]; v[3] = -v[2]
    return np.ascontiguousarray(a)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 8, 9, 10, 11, 13, 16])
def test_hadamard_every_qubit_bit_exact(qc, ob, n):
    for seed, maker in ((7, ob.random_state), (11, lambda nn, ss: sparse_random_state(ob, nn, ss))):
        for q in range(n):
            a = maker(n, seed + q)
            with qc.Register(n, 0) as reg:
                reg.write(a)
                qc.hadamard_gate(q, reg)
                got = reg.read()
            want = a.copy(); ob.hadamard(want, n, q)
            assert_bits_equal(got, want, f"H n={n} q={q}")


@pytest.mark.parametrize("variant", [
    dict(h_variant=1, h_ppt=1, h_block=256, h_nt=0), dict(h_variant=1, h_ppt=8, h_nt=3, h_block=128, h_wc=1),
    dict(h_variant=1, h_ppt=2, h_block=64, h_streams_log2=3), dict(h_variant=1, h_ppt=1, h_block=512, h_streams_log2=1),
    dict(h_variant=2, h_wave_r=2, h_wave_block=256, h_streams_log2=2), dict(h_variant=2, h_wave_r=4, h_wave_block=64, h_nt=0),
    dict(h_variant=2, h_wave_r=8, h_nt=3, h_streams_log2=1), dict(h_variant=1, h_ppt=4, h_grid_cap=3, h_block=256)])
def test_hadamard_all_kernel_variants(qc, ob, variant):
    """every launch form of K1 (pair form / wave-tile shuffle form, nontemporal or not, capped grid)"""
    defaults = {k: qc.lib().qcx_tune_get(k.encode()) for k in ("h_variant", "h_ppt", "h_nt", "h_grid_cap", "h_wave_r", "h_wave_block", "h_block", "h_streams_log2", "h_wc")}
    try:
        qc.tune(**variant)
        for n in (9, 12, 14):
            for q in range(n):
                a = ob.random_state(n, 100 + q)
                with qc.Register(n, 0) as reg:
                    reg.write(a); qc.hadamard_gate(q, reg); got = reg.read()
                want = a.copy(); ob.hadamard(want, n, q)
                assert_bits_equal(got, want, f"H {variant} n={n} q={q}")
    finally:
        qc.tune(**defaults)


@pytest.mark.parametrize("n", [2, 3, 6, 10, 12, 15])
def test_cphase_bit_exact(qc, ob, n):
    rs = np.random.RandomState(n)
    pairs = [(c, t) for c in range(n) for t in range(n) if c != t]
    if len(pairs) > 40:
        pairs = [pairs[i] for i in rs.choice(len(pairs), 40, replace=False)]
    for k, (c, t) in enumerate(pairs):
        theta = math.pi / (1 << (1 + k % 12)) if k % 3 else rs.uniform(-7, 7)
        a = ob.random_state(n, 300 + k)
        with qc.Register(n, 0) as reg:
            reg.write(a); qc.c_phase_shift_gate(c, t, theta, reg); got = reg.read()
        want = a.copy(); ob.cphase(want, n, c, t, theta)
        assert_bits_equal(got, want, f"CPHASE n={n} c={c} t={t} theta={theta}")


CAMODC_CASES = [
    # (L, M, C, atox, ctl)
    (3, 4, 15, 7, 4), (3, 4, 15, 7 ** 2, 5), (3, 4, 15, 4, 6),        # Shor N=15 ladder
    (5, 5, 21, 2, 5), (5, 5, 21, 16, 7), (5, 5, 21, 2 ** 16, 9),      # Shor N=21
    (4, 6, 33, 7, 8), (2, 10, 1000, 999, 11), (2, 12, 4093, 1234, 12), (1, 12, 4096, 4095, 12),
    (3, 4, 15, 5, 4), (3, 4, 15, 3, 6), (3, 4, 15, 0, 5), (3, 4, 15, 15, 5),   # gcd(A, C) > 1: many-to-one sums
    (4, 5, 21, 7, 6), (3, 6, 36, 6, 7),
    (4, 3, 15, 7, 4), (4, 3, 21, 2, 5),                                # C > 2^M: only bits < M of f' kept
    (3, 5, 21, 2, 2), (3, 5, 21, 10, 0), (3, 4, 15, 7, 3),             # control inside the M register
    (12, 4, 15, 7, 15), (13, 5, 21, 4, 17), (9, 5, 21, 2, 10),         # control above / inside the LDS tile
    # M > 12: in place through the staging buffer, K3b (closed form, many-to-one, control inside M, C > 2^M)
    (2, 13, 8191, 1234, 14), (3, 13, 5000, 7, 13), (1, 14, 16000, 3, 14), (2, 13, 8190, 6, 13),
    (2, 13, 6000, 12, 5), (2, 13, 9000, 7, 14),
]


@pytest.mark.parametrize("L,M,C,atox,ctl", CAMODC_CASES)
def test_camodc_bit_exact(qc, ob, L, M, C, atox, ctl):
    n = L + M
    a = ob.random_state(n, 500 + ctl)
    want = a.copy(); ob.camodc(want, n, M, C, atox, ctl)
    for fusion in (False, True):
        with qc.Register(L, M) as reg:
            reg.write(a); reg.set_fusion(fusion)
            qc.c_amodc_gate(C, atox, ctl, reg)
            qc.hadamard_gate(n - 1, reg)                   # something after it
            got = reg.read()
        w2 = want.copy(); ob.hadamard(w2, n, n - 1)
        assert_bits_equal(got, w2, f"C_AMODC L={L} M={M} C={C} atox={atox} ctl={ctl} fusion={fusion}")


@pytest.mark.parametrize("L,M,C,atox,ctl", [(7, 13, 8191, 1234, 15), (6, 14, 16001, 3, 19), (7, 13, 5000, 10, 13), (6, 13, 8000, 7, 4),
                                            (5, 15, 32749, 2, 17), (6, 13, 9000, 7, 16)])
def test_camodc_large_m_in_batches(qc, ob, L, M, C, atox, ctl):
    """M > 12 with more than one 2^M-block per control value: the staged in-place kernels (K3b) with a staging buffer of
    one block, of a few blocks (uneven last batch) and of everything at once -- every amplitude vs the oracle"""
    n = L + M
    a = ob.random_state(n, 900 + ctl)
    want = a.copy(); ob.camodc(want, n, M, C, atox, ctl)
    old = qc.lib().qcx_tune_get(b"cam_stage_mb")
    try:
        for mb in (1, 3, 1024):            # 2^13 rows = 128 KiB: 1 MiB holds 8 blocks, 3 MiB 24, ...
            qc.tune(cam_stage_mb=mb)
            with qc.Register(L, M) as reg:
                reg.write(a)
                qc.c_amodc_gate(C, atox, ctl, reg)
                assert_bits_equal(reg.read(), want, f"C_AMODC L={L} M={M} C={C} atox={atox} ctl={ctl} stage={mb} MiB")
    finally:
        qc.tune(cam_stage_mb=old)


def test_shor_circuit_with_a_large_m_register(qc, ob):
    """the whole circuit of Q:712-737 with M = 13 (C = 8191, L = 9): queued, the multiplies stand-alone between fused passes"""
    L, M, Cn, a = 9, 13, 8191, 3
    n = L + M
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, Cn, a)
    for mode in (1, -1):
        with qc.Register(L, M) as reg:
            reg.set_fusion(mode)
            qc.reset_register(reg)
            qc.quantum_computation(Cn, a, reg)
            assert_bits_equal(reg.read(), want, f"Shor C={Cn} L={L} M={M} mode={mode}")


def test_reset_register(qc):
    with qc.Register(5, 4) as reg:
        reg.write(np.full(2 << 9, 0.25))
        qc.reset_register(reg)
        s = reg.read()
    want = np.zeros(2 << 9); want[2] = 1.0
    assert_bits_equal(s, want, "reset")


@pytest.mark.parametrize("mode", [-1, 0], ids=["one-launch-per-gate", "circuit-as-fused-passes"])
@pytest.mark.parametrize("L,M,C,a", [(3, 4, 15, 7), (4, 4, 15, 7), (5, 5, 21, 2), (5, 5, 33, 7), (8, 4, 15, 7), (6, 6, 35, 2)])
def test_shor_circuit_bit_exact(qc, ob, L, M, C, a, mode):
    n = L + M
    with qc.Register(L, M) as reg:
        reg.set_fusion(mode)
        qc.reset_register(reg)
        qc.quantum_computation(C, a, reg)
        got = reg.read()
        total = reg.norm2()
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a)
    assert_bits_equal(got, want, f"Shor circuit C={C} L={L} M={M} a={a}")
    assert abs(total - 1.0) < 1e-13


def test_shor_circuit_reference_intpow_mode(qc, ob):
    """the reference's wrapped INT_POW (a=7, L=8: 7^128 overflows) reproduced on request"""
    L, M, C, a = 8, 4, 15, 7
    n = L + M
    with qc.Register(L, M) as reg:
        reg.set_fusion(-1)
        qc.reset_register(reg); qc.quantum_computation(C, a, reg, ref_intpow=True); got = reg.read()
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a, ref_intpow=True)
    assert_bits_equal(got, want, "ref-intpow circuit")


@pytest.mark.parametrize("mode", [-1, 0], ids=["one-launch-per-gate", "circuit-as-fused-passes"])
def test_iqft_schedule_full_register(qc, ob, mode):
    n = 12
    a = ob.random_state(n, 77)
    with qc.Register(n, 0) as reg:
        reg.set_fusion(mode)
        reg.write(a); qc.inverse_QFT(reg); got = reg.read()
    want = a.copy(); ob.iqft(want, n, 0)
    assert_bits_equal(got, want, "IQFT schedule n=12")


def test_measurement_histogram_matches_reference_seed(qc, ob):
    """C=15 L=3 M=4 a=7, MT19937 seed 12345, 500 shots -> 123/113/127/137"""
    L, M = 3, 4
    rng = qc.Rng(12345)
    hist = {}
    with qc.Register(L, M) as reg:
        for _ in range(500):
            qc.reset_register(reg)
            qc.quantum_computation(15, 7, reg)
            idx = qc.measure_state(reg, rng)
            w = qc.read_omega(idx, reg)
            hist[w] = hist.get(w, 0) + 1
            s = reg.read()
            assert s[2 * idx] == 1.0 and np.count_nonzero(s) == 1
    assert hist == {0.0: 123, 0.25: 113, 0.5: 127, 0.75: 137}


@pytest.mark.parametrize("n", [1, 5, 6, 7, 11, 14])
def test_measure_index_bit_exact_against_sequential_sum(qc, ob, n):
    a = ob.random_state(n, 900 + n)
    rs = np.random.RandomState(n)
    cum = np.cumsum((a.reshape(-1, 2) ** 2).sum(axis=1))
    rvals = [0.0, 1.0 - 2 ** -53, 0.999999, 0.5, float(cum[0]), float(cum[min(3, cum.size - 1)]),
             float(np.nextafter(cum[cum.size // 2], 2.0))] + list(rs.uniform(0, 1, 12))
    for r in rvals:
        want_state = a.copy(); want = ob.measure(want_state, n, r)
        with qc.Register(n, 0) as reg:
            reg.write(a); got = qc.measure_state(reg, r); got_state = reg.read()
        assert got == want, f"n={n} r={r!r}: got {got} want {want}"
        assert_bits_equal(got_state, want_state, "collapsed state")


def test_bad_arguments(qc):
    from quantumcomputer_amd import QcxError
    with qc.Register(3, 2) as reg:
        with pytest.raises(QcxError):
            qc.hadamard_gate(5, reg)
        with pytest.raises(QcxError):
            qc.c_phase_shift_gate(1, 1, 0.3, reg)
        with pytest.raises(QcxError):
            qc.c_amodc_gate(15, 7, 9, reg)
    with pytest.raises(QcxError):
        qc.Register(40, 40)


@pytest.mark.parametrize("skip,ntl", [(0, 0), (1, 0), (1, 1), (0, 1)])
def test_camodc_store_and_fill_variants(qc, ob, skip, ntl):
    """k_camodc with and without reading the lines above row C, with default and nontemporal stores of the completely
    rewritten lines: the same bits (C just above / below line boundaries, non-coprime multipliers, controls right above
    the M register, inside and above the tile, every M from 3 to 7)"""
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in ("cam_skip", "cam_nt_lines")}
    qc.tune(cam_skip=skip, cam_nt_lines=ntl)
    try:
        for (L, M, C, atox, ctl) in [(11, 5, 21, 2, 9), (11, 5, 24, 5, 15), (10, 6, 33, 7, 6), (9, 5, 7, 3, 13), (12, 4, 15, 10, 5),
                                     (8, 7, 100, 30, 14), (14, 5, 32, 3, 18), (14, 5, 17, 4, 12), (11, 5, 21, 16, 5), (12, 4, 15, 7, 4),
                                     (12, 4, 13, 6, 5), (10, 6, 64, 5, 7), (10, 6, 61, 17, 15), (11, 3, 7, 3, 3), (9, 3, 8, 5, 11), (6, 3, 5, 2, 4)]:
            n = L + M
            a = ob.random_state(n, 600 + ctl)
            want = a.copy(); ob.camodc(want, n, M, C, atox, ctl)
            with qc.Register(L, M) as reg:
                reg.write(a); reg.set_fusion(-1)
                qc.c_amodc_gate(C, atox, ctl, reg)
                assert_bits_equal(reg.read(), want, f"C_AMODC skip={skip} ntl={ntl} L={L} M={M} C={C} atox={atox} ctl={ctl}")
    finally:
        qc.tune(**old)
