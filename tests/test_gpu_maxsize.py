"""GPU: the fused passes, the generated circuit front and the one-read measurement at the LARGEST registers one MI355X holds.

test_gpu_fullsize.py checks single gates up to n = 34 (QCX_TEST_NFULL).  This module drives the multi-pass machinery at
QCX_TEST_NMAX qubits (default 32 = 64 GiB; 33 = 128 GiB still has room for the second buffer of the chained passes, 34 =
256 GiB does not and must fall back to passes in place):

* `qcx_inverse_QFT` through the default fused path on basis states -- windows bit for bit against the oracle's per-index
  evaluation (orc_basis_iqft_window, pinned to the whole-state oracle in tests/test_oracle_pinning.py);
* Shor N = 21 (M = 5, L = NMAX - 5): the front windows against the oracle's closed form, then the whole circuit -- norm, the
  measured index the same through two runs of the scan that share no binade guess (record sizes 2^11 and 2^8), the period visible in omega;
* a fused Hadamard sweep applied twice (identity within rounding), norm kept;
* measure_state on a dense random state: the same index from both record sizes for r over the whole range.

Everything is a window / property check: no state of this size visits the host.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_MAX = int(os.environ.get("QCX_TEST_NMAX", "32"))
W = 13


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def chain_stats(qc, reg):
    v = C.c_ulong(0)
    qc.lib().qcx_chain_stats(reg._h, C.byref(v))
    return int(v.value)


def scan(qc, reg, n, r):
    """the measurement scan without the collapse (qcx_shard_measure_scan on the register's buffer): the index measure_state
    would return for the draw r.  (Taking the device pointer pins the register to passes in place from then on.)"""
    found, index, cum = C.c_int(0), C.c_uint64(0), C.c_double(0.0)
    reg.flush()
    st = qc.lib().qcx_shard_measure_scan(reg.device_pointer(), n, 0, (1 << n) - 1, 0.0, float(r), C.byref(found), C.byref(index),
                                         C.byref(cum), None)
    assert st == 0
    return int(index.value) if found.value else (1 << n) - 1


@pytest.fixture
def tune_guard(qc):
    keys = ("meas_block_log", "fuse_chain", "fuse_gen", "fuse_zskip")
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in keys}
    yield
    qc.tune(**old)


def test_iqft_on_basis_states_default_path(qc, ob):
    n = N_MAX
    rs = np.random.RandomState(n)
    with qc.Register(n, 0) as reg:
        for x in ((1 << n) - 1, (0x5A5A5A5A5 & ((1 << n) - 1)) | 1):
            qc.reset_register(reg)
            reg.write(np.array([0.0, 0.0]), first=1)
            reg.write(np.array([1.0, 0.0]), first=x)
            qc.inverse_QFT(reg)
            assert abs(reg.norm2() - 1.0) < 1e-12
            starts = {0, (1 << n) - (1 << W), x & ~((1 << W) - 1)} | {int(v) << W for v in rs.randint(0, 1 << (n - W), 4)}
            for s in sorted(starts):
                assert np.array_equal(bits(reg.read(s, 1 << W)), bits(ob.basis_iqft_window(x, n, 0, s, 1 << W))), (x, s)


def test_shor_front_and_whole_circuit(qc, ob, tune_guard):
    L, M, Cn, a = N_MAX - 5, 5, 21, 2
    n = L + M
    rs = np.random.RandomState(7)
    with qc.Register(L, M) as reg:
        # the front alone (Hadamard layer + modular-multiply ladder), queued and flushed by the norm
        qc.reset_register(reg)
        for l in range(M, n):
            qc.hadamard_gate(l, reg)
        atox = a % Cn
        for l in range(M, n):
            qc.c_amodc_gate(Cn, atox, l, reg)
            atox = atox * atox % Cn
        assert abs(reg.norm2() - 1.0) < 1e-12
        for s in sorted({0, (1 << n) - (1 << W)} | {int(v) << W for v in rs.randint(0, 1 << (n - W), 6)}):
            assert np.array_equal(bits(reg.read(s, 1 << W)), bits(ob.shor_front_window(n, M, Cn, a, s, 1 << W))), s
        # the whole circuit (front generated inside the first pass where the planner can), then both measurement scans
        qc.reset_register(reg)
        qc.quantum_computation(Cn, a, reg)
        assert abs(reg.norm2() - 1.0) < 1e-11
        picks = []
        for r in (0.0, 1e-7, 0.25, 0.5, 0.77, 0.999999, 1.0):
            qc.tune(meas_block_log=11)          # two runs of the scan that share no binade guess: other records,
            i1 = scan(qc, reg, n, r)          # other look-back groups
            qc.tune(meas_block_log=8)
            i0 = scan(qc, reg, n, r)
            assert i1 == i0, (r, i1, i0)
            picks.append(i1)
        # period 6: omega of every drawn index sits next to a multiple of 1/6
        for i in picks[1:-1]:
            w = qc.read_omega(i, reg)
            assert min(abs(w - k / 6.0) for k in range(7)) < 2.0 ** -8, (i, w)
        # and the collapsing measurement itself
        qc.tune(meas_block_log=0)
        assert qc.measure_state(reg, 0.5) == picks[3]
        assert abs(reg.norm2() - 1.0) == 0.0


def test_fused_sweep_twice_is_identity(qc, ob):
    n = N_MAX
    rs = np.random.RandomState(11)
    scale = math.sqrt(6.0 / (1 << n))
    with qc.Register(n, 0) as reg:
        reg.fill_random(91)
        p0 = reg.norm2()
        reg.set_fusion(1)
        for rep in range(2):
            for q in range(n):
                qc.hadamard_gate(q, reg)
            p1 = reg.norm2()                                  # (flushes the queue: one fused sweep)
            assert abs(p1 - p0) < 1e-12 * p0
        for s in [0, (1 << n) - (1 << W)] + [int(x) << W for x in rs.randint(0, 1 << (n - W), 4)]:
            a = ob.fill_random(n, 91, s, 1 << W)
            assert np.max(np.abs(reg.read(s, 1 << W) - a)) < 64 * n * 2.3e-16 * scale
        ch = chain_stats(qc, reg)
        if n <= 33:
            assert ch >= 2, ch                                # room for the second buffer: the sweeps went through chained passes
        else:
            assert ch == 0, ch                                # 256 GiB: no second buffer, passes in place


def test_measure_dense_random_state_both_scans(qc, tune_guard):
    n = N_MAX
    with qc.Register(n, 0) as reg:
        reg.fill_random(5)
        tot = reg.norm2()
        last = -1
        for f in (0.0, 1e-9, 0.1, 0.5, 0.9, 0.999999999, 1.0, 1.5):
            r = f * tot
            qc.tune(meas_block_log=11)
            i1 = scan(qc, reg, n, r)
            qc.tune(meas_block_log=8)
            i0 = scan(qc, reg, n, r)
            assert i1 == i0, (f, i1, i0)
            assert i1 >= last                                  # the cumulative sum is monotone in r
            last = i1
        assert last == (1 << n) - 1                           # r beyond the total: falls through to the last index
