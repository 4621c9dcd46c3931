"""GPU: state files (qcx_state_save / qcx_state_load): round trip, header checks, corruption detection."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_save_load_round_trip_and_host_reader(qc, ob, tmp_path):
    L, M = 9, 4
    n = L + M
    st = ob.random_state(n, 21)
    st[6] = -0.0                                   # bits, not values
    p = tmp_path / "state.qcx"
    with qc.Register(L, M) as reg:
        reg.write(st)
        qc.hadamard_gate(7, reg)
        want = reg.read()
        reg.save(p)
        qc.reset_register(reg)
        reg.load(p)
        assert np.array_equal(bits(reg.read()), bits(want))
    assert os.path.getsize(p) == 64 + 16 * (1 << n)
    fl, fm, amps = qc.load_state_file(p)
    assert (fl, fm) == (L, M) and np.array_equal(bits(np.array(amps)), bits(want))


def test_queued_gates_are_flushed_before_saving(qc, ob, tmp_path):
    n = 12
    st = ob.random_state(n, 3)
    with qc.Register(n, 0) as reg:
        reg.write(st); reg.set_fusion(True)
        for q in (1, 5, 9):
            qc.hadamard_gate(q, reg); ob.hadamard(st, n, q)
        reg.save(tmp_path / "s.qcx")
    _, _, amps = qc.load_state_file(tmp_path / "s.qcx")
    assert np.array_equal(bits(np.array(amps)), bits(st))


def test_load_rejects_wrong_shape_and_corruption(qc, ob, tmp_path):
    p = tmp_path / "a.qcx"
    with qc.Register(6, 4) as reg:
        reg.fill_random(5); reg.save(p)
    with qc.Register(7, 3) as other:                                   # same n, different registers
        with pytest.raises(qc.QcxError):
            other.load(p)
    with qc.Register(6, 4) as reg:
        raw = bytearray(open(p, "rb").read())
        raw[64 + 100] ^= 1
        open(p, "wb").write(raw)
        with pytest.raises(qc.QcxError, match="truncated or corrupt"):
            reg.load(p)
        open(p, "wb").write(raw[:200])
        with pytest.raises(qc.QcxError):
            reg.load(p)
        open(p, "wb").write(b"not a state file" * 8)
        with pytest.raises(qc.QcxError):
            reg.load(p)
        with pytest.raises(qc.QcxError):
            reg.load(tmp_path / "missing.qcx")


def test_large_state_streams_in_pieces(qc, tmp_path):
    """2^23 amplitudes = 128 MiB: two staging pieces"""
    n = 23
    p = tmp_path / "big.qcx"
    with qc.Register(n, 0) as reg:
        reg.fill_random(9)
        a = reg.read(0, 1 << 12); b = reg.read((1 << n) - 4096, 4096)
        reg.save(p)
        qc.reset_register(reg)
        reg.load(p)
        assert np.array_equal(bits(reg.read(0, 1 << 12)), bits(a)) and np.array_equal(bits(reg.read((1 << n) - 4096, 4096)), bits(b))
        assert abs(reg.norm2() - 1.0) < 1e-3
