"""GPU: the circuit front on a basis state as ONE write pass (K0b, k_basis_front).  reset_register / the collapse of a
measurement are lazy: the next flush writes the basis state together with the longest queue prefix of the shape
"Hadamards on distinct qubits, then controlled modular multiplies" -- the front of quantum_computation (Q:720-731) -- in
closed form, with the reference's own roundings.  Everything here is bit for bit against the oracle."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture()
def tune_guard(qc):
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in ("fuse_front",)}
    yield
    qc.tune(**old)


@pytest.mark.parametrize("C,L,M,a", [(15, 3, 4, 7), (15, 8, 4, 7), (21, 9, 5, 2), (21, 14, 5, 2), (35, 7, 6, 2), (15, 12, 4, 11), (33, 10, 6, 7), (21, 6, 5, 2), (21, 3, 5, 2),
                                     (8191, 9, 13, 3), (16381, 7, 14, 2)])       # M > 12: a workgroup per block (k_basis_front_big)
@pytest.mark.parametrize("mode", [0, 1])
def test_shor_circuit_front_is_one_write_pass(qc, ob, C, L, M, a, mode):
    n = L + M
    with qc.Register(L, M) as reg:
        reg.set_fusion(mode)
        for shot in range(2):
            p0 = reg.fusion_stats()
            qc.reset_register(reg)
            qc.quantum_computation(C, a, reg)
            got = reg.read()
            want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a, threads=8)
            assert np.array_equal(bits(got), bits(want)), (shot,)
            # 2L gates went into the front (tiny registers: the one-thread-per-amplitude form of the kernel)
            if True:
                assert 3 * L + L * (L - 1) // 2 - 2 <= reg.fusion_stats()[1] - p0[1] <= 3 * L + L * (L - 1) // 2     # (a lone last gate may run stand-alone)
            idx = qc.measure_state(reg, 0.3 + 0.2 * shot)
            assert idx == ob.measure(want, n, 0.3 + 0.2 * shot)
            assert np.array_equal(bits(reg.read()), bits(want))          # the lazy collapse, materialised by the read


@pytest.mark.parametrize("seed", range(10))
def test_random_fronts_from_random_basis_states(qc, ob, tune_guard, seed):
    """basis state = a measured index (bits set inside the Hadamard set give minus signs), Hadamards on a random subset
    (also inside the M register: then no multiply joins), multiplies with coprime and non-coprime factors, controls in
    and outside the Hadamard set, C up to 2^M; then more gates of every kind"""
    rs = np.random.RandomState(500 + seed)
    n, M = int(rs.randint(5, 17)), int(rs.choice([0, 2, 4, 5]))
    want = ob.fill_random(n, seed)
    with qc.Register(n - M, M) as reg:
        reg.set_fusion(int(rs.choice([0, 1])))
        reg.fill_random(seed)
        r = float(rs.uniform(0, 1))
        assert qc.measure_state(reg, r) == ob.measure(want, n, r)         # collapse: lazy on the product side
        lo = M if rs.rand() < 0.7 else 0
        hs = [int(q) for q in rs.permutation(np.arange(lo, n))[: int(rs.randint(0, n - lo + 1))]]
        for q in hs:
            qc.hadamard_gate(q, reg); ob.hadamard(want, n, q)
        if M:
            for _ in range(int(rs.randint(0, 8))):
                Cn = int(rs.randint(2, (1 << M) + 1)); A = int(rs.randint(1, 3 * Cn)); ctl = int(rs.randint(M, n))
                qc.c_amodc_gate(Cn, A, ctl, reg); ob.camodc(want, n, M, Cn, A, ctl)
        for _ in range(int(rs.randint(0, 12))):
            k = rs.randint(0, 3)
            if k == 0:
                q = int(rs.randint(0, n)); qc.hadamard_gate(q, reg); ob.hadamard(want, n, q)
            elif k == 1 or M == 0:
                c, t = (int(x) for x in rs.choice(n, 2, replace=False)); th = float(rs.uniform(-3, 3))
                qc.c_phase_shift_gate(c, t, th, reg); ob.cphase(want, n, c, t, th)
            else:
                Cn = int(rs.randint(2, (1 << M) + 1)); A = int(rs.randint(1, 3 * Cn)); ctl = int(rs.randint(M, n))
                qc.c_amodc_gate(Cn, A, ctl, reg); ob.camodc(want, n, M, Cn, A, ctl)
        assert np.array_equal(bits(reg.read()), bits(want)), (n, M, hs)


def test_front_off_and_strict_mode_give_the_same_bits(qc, ob, tune_guard):
    L, M, C, a = 11, 5, 21, 2
    n = L + M
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a, threads=8)
    for front, mode in ((0, 0), (1, 0), (1, -1), (1, 2)):
        qc.tune(fuse_front=front)
        with qc.Register(L, M) as reg:
            reg.set_fusion(mode)
            qc.reset_register(reg)
            qc.quantum_computation(C, a, reg)
            got = reg.read()
        if mode == 2:
            d = got - want
            assert float(np.max(np.hypot(d[0::2], d[1::2]))) <= 1e-12
        else:
            assert np.array_equal(bits(got), bits(want)), (front, mode)


def test_lazy_reset_is_visible_to_every_observer(qc, ob, tmp_path):
    n = 12
    with qc.Register(n, 0) as reg:
        reg.fill_random(3)
        qc.reset_register(reg)                                  # nothing written yet
        assert reg.norm2() == 1.0 and reg.total_probability() == 1.0
        s = reg.read(); assert s[2] == 1.0 and np.count_nonzero(s) == 1
        qc.reset_register(reg)
        reg.write(np.array([0.5, 0.25]), first=7)               # partial write on top of the (materialised) reset state
        s = reg.read(); assert s[2] == 1.0 and s[14] == 0.5 and s[15] == 0.25 and np.count_nonzero(s) == 3
        reg.fill_random(5); qc.reset_register(reg); reg.fill_random(6)      # a fill after a lazy reset wins
        assert np.array_equal(bits(reg.read()), bits(ob.fill_random(n, 6)))
        qc.reset_register(reg); p = str(tmp_path / "s.qcx"); reg.save(p)
        reg.fill_random(1); reg.load(p)
        s = reg.read(); assert s[2] == 1.0 and np.count_nonzero(s) == 1
        qc.reset_register(reg)
        assert qc.measure_state(reg, 0.99) == 1                 # measuring the basis state gives it back
        assert reg.device_pointer() != 0
        s = reg.read(); assert s[2] == 1.0 and np.count_nonzero(s) == 1


def test_config5_front_n30_is_one_pass(qc, ob):
    """n = 30: reset + 25 H + 25 multiplies through the front, windows against the oracle's per-index chain"""
    L, M, Cn, a = 25, 5, 21, 2
    n = L + M
    W = 13
    rs = np.random.RandomState(6)
    with qc.Register(L, M) as reg:
        reg.set_fusion(1)
        qc.reset_register(reg)
        for l in range(M, n):
            qc.hadamard_gate(l, reg)
        atox = a % Cn
        for l in range(M, n):
            qc.c_amodc_gate(Cn, atox, l, reg); atox = atox * atox % Cn
        p0 = reg.fusion_stats()[0]
        assert abs(reg.norm2() - 1.0) < 1e-12
        assert reg.fusion_stats()[0] - p0 == 1                  # the whole front: one write pass
        for s in sorted({0, (1 << n) - (1 << W)} | {int(v) << W for v in rs.randint(0, 1 << (n - W), 6)}):
            assert np.array_equal(bits(reg.read(s, 1 << W)), bits(ob.shor_front_window(n, M, Cn, a, s, 1 << W))), s


def _gen_fronts(qc, reg):
    import ctypes as C
    out = C.c_ulong(0)
    qc.lib().qcx_gen_stats(reg._h, C.byref(out))
    return out.value


def _gen_cols(qc, reg):
    import ctypes as C
    out = C.c_ulong(0)
    qc.lib().qcx_gen_cols_stats(reg._h, C.byref(out))
    return out.value


def _compact(qc, reg):
    import ctypes as C
    out = C.c_ulong(0)
    qc.lib().qcx_compact_stats(reg._h, C.byref(out))
    return out.value


@pytest.mark.parametrize("mode", [0, 2], ids=["exact", "tolerance"])
@pytest.mark.parametrize("C,L,M,a", [(21, 9, 5, 2), (21, 14, 5, 2), (15, 12, 4, 11), (33, 10, 6, 7), (35, 17, 6, 2), (21, 16, 5, 16), (255, 9, 8, 2), (255, 12, 8, 2), (15, 13, 4, 7)])
def test_front_generated_inside_the_first_pass(qc, ob, C, L, M, a, mode):
    """round 4: when a fused pass follows the circuit front, the front is not written at all -- the pass generates its tiles
    (GenFront) instead of reading them.  Same bits as with the separate write pass (fuse_gen = 0) and as the oracle; from
    reset states and from measured basis states (minus signs); the tolerance mode's passes generate the same front."""
    n = L + M
    old = qc.lib().qcx_tune_get(b"fuse_gen")
    oldz = qc.lib().qcx_tune_get(b"fuse_zskip")
    oldc = qc.lib().qcx_tune_get(b"fuse_gen_cols")
    oldk = qc.lib().qcx_tune_get(b"fuse_compact")
    try:
        outs = []
        # (zskip: waves whose share of a tile is all +0 skip the rounds; cols: the generated pass keeps only the populated
        #  columns of the four lowest M-register bits, a wave per column -- k_gen_cols, n >= 14)
        #  compact: the whole flush on a compact copy of the state, [L register][orbit column] -- compact_chain)
        for gen, zskip, cols, compact in ((1, 1, 1, 1), (0, 1, 1, 1), (1, 0, 1, 0), (1, 1, 0, 1), (1, 1, 1, 0)):
            qc.tune(fuse_gen=gen, fuse_zskip=zskip, fuse_gen_cols=cols, fuse_compact=compact)
            with qc.Register(L, M) as reg:
                reg.set_fusion(mode)
                g0 = _gen_fronts(qc, reg)
                c0 = _gen_cols(qc, reg)
                k0 = _compact(qc, reg)
                qc.reset_register(reg); qc.quantum_computation(C, a, reg)
                first = reg.read()
                idx = qc.measure_state(reg, 0.41)                       # collapse: a basis state with bits inside the Hadamard set
                for l in range(M, n):
                    qc.hadamard_gate(l, reg)
                qc.inverse_QFT(reg)
                second = reg.read()
                outs.append((first, idx, second))
                assert (_gen_fronts(qc, reg) - g0 >= 1) == bool(gen)
                assert (_gen_cols(qc, reg) - c0 >= 1) == bool(gen and cols)     # (both modes: the tolerance mode takes the exact first pass)
                if (C, L, M) in ((21, 14, 5), (35, 17, 6), (21, 16, 5), (255, 12, 8), (15, 13, 4)):
                    assert (_compact(qc, reg) - k0 >= 1) == bool(gen and cols and compact)
        want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a, threads=8)
        if mode == 0:
            assert all(np.array_equal(bits(o[0]), bits(want)) for o in outs)
        else:
            assert all(float(np.max(np.abs(o[0] - want))) <= 1e-12 for o in outs)
        assert outs[0][1] == outs[1][1] == outs[2][1] == outs[3][1] == outs[4][1] == ob.measure(want, n, 0.41)
        w2 = want                                                        # collapsed by ob.measure
        for l in range(M, n):
            ob.hadamard(w2, n, l, 8)
        ob.iqft(w2, n, M, 8)
        if mode == 0:
            assert all(np.array_equal(bits(o[2]), bits(w2)) for o in outs)
        else:
            assert all(float(np.max(np.abs(o[2] - w2))) <= 1e-12 for o in outs)
    finally:
        qc.tune(fuse_gen=old, fuse_zskip=oldz, fuse_gen_cols=oldc, fuse_compact=oldk)


@pytest.mark.parametrize("mode", [0, 2], ids=["exact", "tolerance"])
@pytest.mark.parametrize("C,L,M,a", [(21, 14, 5, 2), (35, 13, 6, 2), (15, 13, 4, 7), (255, 12, 8, 2), (32, 12, 5, 31)])
def test_measurement_on_the_compact_form(qc, ob, C, L, M, a, mode):
    """a whole-circuit entry point may leave the state in the compact form of its chain ([L register][orbit column]):
    measure_state scans it there -- the amplitudes it leaves out are +0 -- and must return the index the reference's scan of
    the whole register returns, for draws over the whole range (r = 0 stops at index 0 whatever it holds; C = 32, a = 31 with
    M = 5 puts the register's last index, which the scan never examines, on the orbit {1, 31}); every other observer sees the expanded state."""
    n = L + M
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a, threads=8)
    tot = float((want ** 2).sum())
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in ("fuse_compact", "fuse_compact_lazy")}
    try:
        for lazy in (1, 0):
            qc.tune(fuse_compact=1, fuse_compact_lazy=lazy)
            with qc.Register(L, M) as reg:
                reg.set_fusion(mode)
                k0 = _compact(qc, reg)
                for r in (0.0, 1e-12, 0.1, 0.37, 0.5, 0.93, 0.999999, tot, 1.0 - 1e-16, 1.5):
                    qc.reset_register(reg)
                    qc.quantum_computation(C, a, reg)
                    idx = qc.measure_state(reg, r)
                    w = want.copy()
                    if mode == 0:
                        assert idx == ob.measure(w, n, r), (lazy, r)
                    else:                                   # the tolerance mode's sums differ in the last bits: the draw must fall next to the same boundary
                        cum = np.cumsum((want.reshape(-1, 2) ** 2).sum(axis=1))
                        lo = int(np.searchsorted(cum, r - 1e-9)); hi = int(np.searchsorted(cum, r + 1e-9))
                        assert (r <= 0.0 and idx == 0) or lo <= idx <= max(hi, lo) or idx == (1 << n) - 1, (lazy, r, idx, lo, hi)
                    got = reg.read()                        # the collapsed state, materialised by the read
                    assert got[2 * idx] == 1.0 and float(np.abs(got).sum()) == 1.0
                assert _compact(qc, reg) - k0 == 10
                # not a measurement: the observer sees the expanded state
                qc.reset_register(reg)
                qc.quantum_computation(C, a, reg)
                assert abs(reg.norm2() - tot) < 1e-12
                got = reg.read()
                if mode == 0:
                    assert np.array_equal(bits(got), bits(want))
                else:
                    assert float(np.max(np.abs(got - want))) <= 1e-12
                # gates queued behind a compact state run on the expanded one
                qc.reset_register(reg)
                qc.quantum_computation(C, a, reg)
                qc.hadamard_gate(n - 1, reg); qc.hadamard_gate(0, reg)
                w2 = want.copy(); ob.hadamard(w2, n, n - 1, 8); ob.hadamard(w2, n, 0, 8)
                got = reg.read()
                assert np.array_equal(bits(got), bits(w2)) if mode == 0 else float(np.max(np.abs(got - w2))) <= 1e-12
    finally:
        qc.tune(**old)


def _expanding(qc, reg):
    import ctypes as C
    out = C.c_ulong(0)
    qc.lib().qcx_expanding_store_stats(reg._h, C.byref(out))
    return out.value


@pytest.mark.parametrize("mode", [0, 2], ids=["exact", "tolerance"])
@pytest.mark.parametrize("C,L,M,a", [(21, 16, 5, 2), (21, 19, 5, 2), (15, 16, 4, 7), (35, 17, 6, 2), (255, 15, 8, 2), (32, 15, 5, 31), (21, 15, 5, 16)])
def test_last_pass_of_a_compact_chain_stores_the_real_register(qc, ob, C, L, M, a, mode):
    """round 5: when the state is wanted in the register, the last pass of a compact chain (a k_fused_x8 pass) writes the real
    register itself -- (l << M) | orbit[j] for column j, +0 everywhere else -- and k_expand_compact does not run.  Same bits as
    with the separate expansion (fuse_expand_fused = 0) and as the oracle, from a reset state and from a measured basis state."""
    n = L + M
    old = qc.lib().qcx_tune_get(b"fuse_expand_fused")
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a, threads=8)
    # does the chain's last pass qualify?  The planner alone, as compact_chain calls it (mode | 8), on the virtual register
    # [L register][orbit column]: a k_fused_x8 pass (records FUSE_ROUND8 = 9 / FUSE_QROUND3 = 8 on a 2^12 tile) holding the column bits
    orbit = sorted({pow(a, e, C) for e in range(4 * C)})
    cb = max(2, (len(orbit) - 1).bit_length())
    nv = L + cb
    descs = []
    for l in range(nv - 1, cb - 1, -1):
        descs.append((0, l, 0, 0.0, 0.0, 0, 0))
        for k in range(l - 1, cb - 1, -1):
            c_, s_ = qc.polar(math.pi / float(1 << (l - k)))
            descs.append((1, 0, (1 << l) | (1 << k), c_, s_, 0, 0))
    acts, recs, _ = qc.fusion_plan(nv, cb, descs, (2 if mode == 2 else 1) | 4 | 8)
    last = acts[-1]
    kinds = {recs[last.rec_off + k].type & 0xFF for k in range(last.nops)} if last.fused else set()
    qualifies = int(len(acts) > 1 and last.fused and last.T == 12 and (9 in kinds or (8 in kinds and last.diag_cnt > 0)) and list(last.st_pos[:cb]) == list(range(cb)))
    # (35, 17, 6): the plan ends in a stand-alone gate -- the separate expansion runs, also a case worth having
    assert qualifies or mode == 2 or (C, L, M) == (35, 17, 6), "the exact chains of these sizes end in a k_fused_x8 pass"
    try:
        outs = []
        for fused in (1, 0):
            qc.tune(fuse_expand_fused=fused)
            with qc.Register(L, M) as reg:
                reg.set_fusion(mode)
                reg.fill_random(3)                                   # stale amplitudes everywhere: every +0 must be written
                e0, k0 = _expanding(qc, reg), _compact(qc, reg)
                qc.reset_register(reg); qc.quantum_computation(C, a, reg)
                first = reg.read()
                assert _compact(qc, reg) - k0 == 1
                assert _expanding(qc, reg) - e0 == fused * qualifies
                idx = qc.measure_state(reg, 0.77)                    # collapse, then the Hadamard layer and the inverse QFT again: minus signs in the front
                for l in range(M, n):
                    qc.hadamard_gate(l, reg)
                qc.inverse_QFT(reg)
                second = reg.read()
                outs.append((first, idx, second))
        w = want.copy()
        assert outs[0][1] == outs[1][1] and (mode != 0 or outs[0][1] == ob.measure(w, n, 0.77))
        if mode == 0:
            assert np.array_equal(bits(outs[0][0]), bits(want)) and np.array_equal(bits(outs[1][0]), bits(want))
            for l in range(M, n):
                ob.hadamard(w, n, l, 8)
            ob.iqft(w, n, M, 8)
            assert np.array_equal(bits(outs[0][2]), bits(w)) and np.array_equal(bits(outs[1][2]), bits(w))
        else:
            assert float(np.max(np.abs(outs[0][0] - want))) <= 1e-12 and float(np.max(np.abs(outs[1][0] - want))) <= 1e-12
            assert float(np.max(np.abs(outs[0][2] - outs[1][2]))) <= 1e-12
    finally:
        qc.tune(fuse_expand_fused=old)


@pytest.mark.parametrize("mode", [1, 2], ids=["exact (every gate queued)", "tolerance"])
@pytest.mark.parametrize("seed", range(6))
def test_compact_chain_under_random_programs(qc, ob, seed, mode):
    """behind the front ANY program of Hadamards and controlled phases on the L register qualifies for a compact chain, not
    only the inverse QFT's schedule: random programs (the planner then produces stand-alone gates, single-Hadamard rounds,
    several chains ...) against the oracle; a second batch of gates, queued behind the compact result, runs on the expanded state"""
    rs = np.random.RandomState(4000 + seed)
    C, M, a = [(21, 5, 2), (15, 4, 7), (35, 6, 2), (255, 8, 2), (33, 6, 7), (21, 5, 16)][seed]
    L = int(rs.randint(11, 15))
    n = L + M
    want = np.zeros(2 << n); ob.reset(want, n)
    with qc.Register(L, M) as reg:
        reg.set_fusion(mode)
        k0 = _compact(qc, reg)
        qc.reset_register(reg)
        for l in range(M, n):
            qc.hadamard_gate(l, reg); ob.hadamard(want, n, l, 8)
        x = a % C
        for l in range(M, n):
            qc.c_amodc_gate(C, x, l, reg); ob.camodc(want, n, M, C, x, l, 8); x = (x * x) % C
        for batch in range(2):
            for _ in range(int(rs.randint(30, 90))):
                if rs.randint(0, 3) == 0:
                    q = int(rs.randint(M, n)); qc.hadamard_gate(q, reg); ob.hadamard(want, n, q, 8)
                else:
                    c, t = (int(v) for v in rs.choice(np.arange(M, n), 2, replace=False))
                    th = float(rs.uniform(-3.2, 3.2)) if rs.randint(0, 2) else math.pi / (1 << int(rs.randint(1, 12)))
                    qc.c_phase_shift_gate(c, t, th, reg); ob.cphase(want, n, c, t, th, 8)
            got = reg.read()
            if mode == 1:
                assert np.array_equal(bits(got), bits(want)), (seed, batch)
            else:
                assert float(np.max(np.abs(got - want))) <= 1e-12, (seed, batch)
            if batch == 0 and L + max(2, (len({pow(a, e, C) for e in range(4 * C)}) - 1).bit_length()) >= 14:
                assert _compact(qc, reg) - k0 == 1, "the first flush ran as a compact chain"
