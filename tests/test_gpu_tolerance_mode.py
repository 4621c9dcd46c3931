"""GPU: the opt-in TOLERANCE MODE (qcx_set_fusion(reg, 2), K6t in csrc/qcx_kernels.h): runs of controlled phases that
share a qubit are merged into one diagonal (SURVEY s8(f)-2, Q:682-689) -- one complex multiply per amplitude, FMA
allowed.  NOT bit-exact by design; the bound written here is

    max |amplitude - oracle amplitude|  <=  1e-12     for the circuits below (states of norm 1),

two orders inside what north_star allows (1e-10).  The default modes (-1, 0, 1) stay bit-exact and are tested elsewhere;
this file also checks that mode 2 leaves them alone, that the measurement histogram of the reference's seeded run is
unchanged, and the n = 28 inverse-QFT (BASELINE config 3) on basis states against the oracle's per-index chains."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-12
W = 13


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def max_delta(a, b):
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return float(np.max(np.hypot(d[0::2], d[1::2])))


@pytest.fixture()
def tune_guard(qc):
    keys = ("fuse_T", "fuse_c", "fuse_grid_cap", "fuse_T_phase", "fuse_c_phase", "fuse_phase_ratio", "fuse_tol_occ", "fuse_hsweep_T", "fuse_tol_T", "fuse_q3")
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in keys}
    yield
    qc.tune(**old)


@pytest.mark.parametrize("n,M", [(10, 0), (12, 0), (14, 4), (16, 0), (16, 5), (11, 5)])
@pytest.mark.parametrize("geom", [None, (10, 4), (11, 4), (12, 3), (12, 4)], ids=["default", "T10c4", "T11c4", "T12c3", "T12c4"])
def test_inverse_qft_within_tolerance(qc, ob, tune_guard, n, M, geom):
    if geom:
        if geom[0] > n:
            pytest.skip("tile larger than the register")
        qc.tune(fuse_T=geom[0], fuse_c=geom[1], fuse_T_phase=0, fuse_tol_T=0, fuse_q3=0)
    with qc.Register(n - M, M) as reg:
        reg.set_fusion(qc.FUSION_TOLERANCE)
        reg.fill_random(21)
        n0 = reg.norm2()
        qc.inverse_QFT(reg)
        got = reg.read()
        assert abs(reg.norm2() - n0) < 1e-13
        passes = reg.fusion_stats()[0]
    want = ob.fill_random(n, 21); ob.iqft(want, n, M, 8)
    assert max_delta(got, want) <= TOL
    if n - M >= 4:
        assert passes >= 1 and not np.array_equal(bits(got), bits(want))      # the merged arithmetic really ran


@pytest.mark.parametrize("occ", [6, 8])
@pytest.mark.parametrize("C,L,M,a", [(15, 3, 4, 7), (15, 8, 4, 7), (21, 9, 5, 2), (35, 6, 6, 2), (21, 11, 5, 2)])
def test_shor_circuit_within_tolerance(qc, ob, tune_guard, C, L, M, a, occ):
    qc.tune(fuse_tol_occ=occ)
    n = L + M
    with qc.Register(L, M) as reg:
        reg.set_fusion(2)
        qc.reset_register(reg)
        qc.quantum_computation(C, a, reg)
        got = reg.read()
        assert abs(reg.norm2() - 1.0) < 1e-13
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a, threads=8)
    assert max_delta(got, want) <= TOL


@pytest.mark.parametrize("seed", range(8))
def test_random_programs_within_tolerance(qc, ob, tune_guard, seed):
    """phase runs that share a qubit (targets may repeat), phases that share none, Hadamards, modular multiplies; controls
    on register bits, lane bits and outside the tile; several geometries and a grid that walks many tiles"""
    rs = np.random.RandomState(300 + seed)
    n, M = int(rs.randint(10, 19)), int(rs.choice([0, 3, 4, 5]))
    Cn = int(rs.randint(3, 1 << M)) if M else 1
    T, c = [(10, 4), (11, 4), (12, 3), (11, 2), (12, 4), (10, 6), (11, 4), (12, 0)][seed]
    qc.tune(fuse_T=min(T, n), fuse_c=min(c, T, n), fuse_grid_cap=[0, 5][seed % 2] or 24576, fuse_tol_T=[0, 10][seed % 2], fuse_q3=[1, 0, 1, 1][seed % 4])
    want = ob.fill_random(n, seed)
    with qc.Register(n - M, M) as reg:
        reg.set_fusion(2)
        reg.fill_random(seed)
        for _ in range(int(rs.randint(30, 70))):
            k = rs.randint(0, 10)
            if k < 3:
                q = int(rs.randint(0, n)); qc.hadamard_gate(q, reg); ob.hadamard(want, n, q, 8)
            elif k < 8 or M == 0:
                ctl = int(rs.randint(0, n))
                for _ in range(int(rs.randint(1, 10))):
                    t = int(rs.randint(0, n))
                    if t == ctl:
                        continue
                    th = float(rs.uniform(-3, 3))
                    qc.c_phase_shift_gate(ctl, t, th, reg); ob.cphase(want, n, ctl, t, th, 8)
            else:
                atox, ctl = int(rs.randint(1, 4 * Cn)), int(rs.randint(M, n))
                qc.c_amodc_gate(Cn, atox, ctl, reg); ob.camodc(want, n, M, Cn, atox, ctl)
        got = reg.read()
    assert max_delta(got, want) <= TOL, (n, M, T, c)


@pytest.mark.parametrize("n,M", [(12, 0), (12, 3), (20, 0), (21, 5), (24, 0), (20, 4)])
def test_radix8_rounds_within_tolerance(qc, ob, n, M):
    """registers where 2^12 tiles with radix-8 fast rounds (k_fused_q3) save a pass: that kernel runs, against the oracle"""
    with qc.Register(n - M, M) as reg:
        reg.set_fusion(2)
        reg.fill_random(8)
        p0 = reg.fusion_stats()[0]
        qc.inverse_QFT(reg)
        got = reg.read()
        passes = reg.fusion_stats()[0] - p0
    want = ob.fill_random(n, 8); ob.iqft(want, n, M, 8)
    assert max_delta(got, want) <= TOL
    assert passes == (n - max(M, 4) + 7) // 8          # 8 hot bits per pass


def test_seeded_histogram_is_unchanged(qc):
    """C=15 L=3 M=4 a=7, MT19937 seed 12345, 500 shots -> 123/113/127/137, in tolerance mode too"""
    L, M = 3, 4
    rng = qc.Rng(12345)
    hist = {}
    with qc.Register(L, M) as reg:
        reg.set_fusion(2)
        for _ in range(500):
            qc.reset_register(reg)
            qc.quantum_computation(15, 7, reg)
            w = qc.read_omega(qc.measure_state(reg, rng), reg)
            hist[w] = hist.get(w, 0) + 1
    assert hist == {0.0: 123, 0.25: 113, 0.5: 127, 0.75: 137}


def test_seeded_measurements_agree_with_the_exact_mode(qc):
    """a register large enough for fused passes (n = 14): 40 seeded shots give the same indices in both modes"""
    L, M = 9, 5
    picks = []
    for mode in (0, 2):
        rng = qc.Rng(2024)
        with qc.Register(L, M) as reg:
            reg.set_fusion(mode)
            out = []
            for _ in range(40):
                qc.reset_register(reg); qc.quantum_computation(21, 2, reg)
                out.append(qc.measure_state(reg, rng))
            picks.append(out)
    assert picks[0] == picks[1]


def test_switching_back_restores_the_exact_bits(qc, ob):
    n = 14
    with qc.Register(n, 0) as reg:
        for mode, exact in ((2, False), (1, True), (0, True), (2, False), (-1, True)):
            reg.set_fusion(mode)
            reg.fill_random(4)
            qc.inverse_QFT(reg)
            got = reg.read()
            want = ob.fill_random(n, 4); ob.iqft(want, n, 0, 8)
            assert max_delta(got, want) <= TOL
            assert np.array_equal(bits(got), bits(want)) == exact, mode


def _basis_state(qc, reg, x):
    qc.reset_register(reg)
    if x != 1:
        reg.write(np.array([0.0, 0.0]), first=1)
        reg.write(np.array([1.0, 0.0]), first=x)


def test_config3_iqft_n28_on_basis_states_within_tolerance(qc, ob):
    """BASELINE config 3 at full size in tolerance mode: windows of the 2^28 result against the oracle's per-index chains
    (orc_basis_iqft_window).  Amplitudes have magnitude 2^-14; the bound is relative to that."""
    n = 28
    rs = np.random.RandomState(29)
    with qc.Register(n, 0) as reg:
        reg.set_fusion(2)
        for x in (0, (1 << n) - 1, 0x5A5A5A5 & ((1 << n) - 1), (1 << (n - 1)) | 1):
            _basis_state(qc, reg, x)
            qc.inverse_QFT(reg)
            assert abs(reg.norm2() - 1.0) < 1e-12
            starts = {0, (1 << n) - (1 << W), x & ~((1 << W) - 1)} | {int(v) << W for v in rs.randint(0, 1 << (n - W), 4)}
            for s in sorted(starts):
                got = reg.read(s, 1 << W)
                assert max_delta(got, ob.basis_iqft_window(x, n, 0, s, 1 << W)) <= TOL * 2.0 ** -14, (x, s)
        assert reg.fusion_stats()[0] == 4 * 3                  # three radix-8 passes per transform


def test_config3_iqft_n28_dense_input_close_to_the_exact_mode(qc):
    n = 28
    rs = np.random.RandomState(4)
    with qc.Register(n, 0) as a, qc.Register(n, 0) as b:
        a.fill_random(11); b.fill_random(11)
        n0 = a.norm2()
        a.set_fusion(2)
        qc.inverse_QFT(a); qc.inverse_QFT(b)
        assert abs(a.norm2() - n0) < 1e-12
        for s in sorted({0, (1 << n) - (1 << W)} | {int(v) << W for v in rs.randint(0, 1 << (n - W), 8)}):
            assert max_delta(a.read(s, 1 << W), b.read(s, 1 << W)) <= TOL * 2.0 ** -13, s


def test_config5_shor_n30_tolerance_against_the_exact_mode(qc):
    """BASELINE config 5 on one GPU at full size: the n = 30 Shor N = 21 circuit in tolerance mode next to the bit-exact
    default (itself checked against the oracle in tests/test_gpu_fullsize.py / test_gpu_basis_front.py): windows agree to
    1e-12 of the amplitude scale, norm is kept, the same uniform draw measures the same index"""
    L, M, Cn, a = 25, 5, 21, 2
    n = L + M
    rs = np.random.RandomState(30)
    with qc.Register(L, M) as ex, qc.Register(L, M) as tl:
        tl.set_fusion(2)
        for reg in (ex, tl):
            qc.reset_register(reg); qc.quantum_computation(Cn, a, reg)
        assert abs(tl.norm2() - 1.0) < 1e-12
        scale = 2.0 ** -(L // 2)
        for s in sorted({0, (1 << n) - (1 << W)} | {int(v) << W for v in rs.randint(0, 1 << (n - W), 8)}):
            assert max_delta(tl.read(s, 1 << W), ex.read(s, 1 << W)) <= TOL * scale, s
        for r in (0.123456789, 0.5, 0.987654321):
            for reg in (ex, tl):
                qc.reset_register(reg); qc.quantum_computation(Cn, a, reg)
            assert qc.measure_state(ex, r) == qc.measure_state(tl, r)
