"""GPU: the gate queue + fused LDS-tile passes give the SAME BITS as the per-gate kernels and the
oracle: random gate programs, whole Shor circuits, the IQFT ladder, every tile geometry."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture()
def tune_guard(qc):
    keys = ("fuse_T", "fuse_c", "fuse_grid_cap", "fuse_max_queue", "fuse_rounds", "fuse_ldsdma", "fuse_camruns",
            "fuse_T_phase", "fuse_c_phase", "fuse_phase_ratio", "fuse_rounds_occ", "fuse_hsweep_T", "fuse_hsweep_c")
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in keys}
    yield
    qc.tune(**old)


def random_program(rs, n, M, Cn, length):
    prog = []
    for _ in range(length):
        k = rs.randint(0, 10)
        if k < 4:
            prog.append(("h", int(rs.randint(0, n))))
        elif k < 9 or M == 0:
            c, t = rs.choice(n, 2, replace=False)
            prog.append(("p", int(c), int(t), float(rs.uniform(-3, 3)) if k % 2 else math.pi / (1 << int(rs.randint(1, 12)))))
        else:
            prog.append(("c", int(rs.randint(1, 4 * Cn)), int(rs.randint(M, n))))
    return prog


def run_both(qc, ob, L, M, Cn, prog, seed, fusion=True):
    n = L + M
    want = ob.random_state(n, seed)
    with qc.Register(L, M) as reg:
        reg.write(want)
        reg.set_fusion(fusion)
        for g in prog:
            if g[0] == "h":
                qc.hadamard_gate(g[1], reg); ob.hadamard(want, n, g[1])
            elif g[0] == "p":
                qc.c_phase_shift_gate(g[1], g[2], g[3], reg); ob.cphase(want, n, g[1], g[2], g[3])
            else:
                qc.c_amodc_gate(Cn, g[1], g[2], reg); ob.camodc(want, n, M, Cn, g[1], g[2])
        got = reg.read()
        stats = reg.fusion_stats()
    return got, want, stats


@pytest.mark.parametrize("L,M,Cn", [(3, 2, 3), (5, 4, 15), (9, 4, 15), (10, 5, 21), (13, 5, 21), (12, 0, 1), (10, 10, 1000)])
def test_random_programs_fused_bit_exact(qc, ob, L, M, Cn):
    rs = np.random.RandomState(L * 31 + M)
    for trial in range(3):
        prog = random_program(rs, L + M, M, Cn, 60)
        got, want, stats = run_both(qc, ob, L, M, Cn, prog, 70 + trial)
        assert np.array_equal(bits(got), bits(want)), f"L={L} M={M} trial {trial}"
    if L + M >= 8:
        assert stats[0] > 0 and stats[1] >= stats[0]


@pytest.mark.parametrize("rounds,dma,tphase", [(1, 1, 10), (1, 1, 0), (0, 1, 10), (0, 0, 0), (1, 0, 0), (1, 1, 11), (1, 1, 12)])
def test_kernel_forms(qc, ob, tune_guard, rounds, dma, tphase):
    """rounds / per-gate op form, LDS-DMA / register fill, and the tile size of phase-dominated passes (0: same kernel
    as the rest; ratio 1 so that the random programs reach it)"""
    qc.tune(fuse_rounds=rounds, fuse_ldsdma=dma, fuse_T_phase=tphase, fuse_phase_ratio=1)
    rs = np.random.RandomState(rounds * 8 + dma * 2 + tphase)
    for (L, M, Cn) in ((13, 5, 21), (16, 4, 15), (20, 0, 1)):
        prog = random_program(rs, L + M, M, Cn, 90)
        got, want, _ = run_both(qc, ob, L, M, Cn, prog, 11)
        assert np.array_equal(bits(got), bits(want)), (L, M)


@pytest.mark.parametrize("T,c", [(8, 2), (9, 4), (10, 6), (11, 3), (12, 3), (12, 4), (12, 6), (12, 0), (10, 10)])
def test_every_tile_geometry(qc, ob, tune_guard, T, c):
    qc.tune(fuse_T=T, fuse_c=c)
    rs = np.random.RandomState(T * 16 + c)
    L, M, Cn = 11, 5, 21
    prog = random_program(rs, L + M, M, Cn, 80)
    got, want, _ = run_both(qc, ob, L, M, Cn, prog, 5)
    assert np.array_equal(bits(got), bits(want))
    qc.tune(fuse_grid_cap=3)                       # grid-stride over tiles
    got, want, _ = run_both(qc, ob, L, M, Cn, prog, 6)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("L,M,C,a", [(3, 4, 15, 7), (8, 4, 15, 7), (9, 5, 21, 2), (6, 6, 35, 2), (14, 5, 21, 2)])
def test_shor_circuit_fused(qc, ob, L, M, C, a):
    n = L + M
    with qc.Register(L, M) as reg:
        reg.set_fusion(True)
        qc.reset_register(reg)
        qc.quantum_computation(C, a, reg)
        got = reg.read()
        passes, gates = reg.fusion_stats()
    want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a, threads=8)
    assert np.array_equal(bits(got), bits(want))
    if n >= 8:
        assert gates == 3 * L + L * (L - 1) // 2 or passes > 0
        assert passes < gates / 3


def test_iqft_and_sweep_fused(qc, ob):
    n = 16
    a = ob.random_state(n, 3)
    with qc.Register(n, 0) as reg:
        reg.write(a); reg.set_fusion(True)
        qc.inverse_QFT(reg)
        for q in range(n):
            qc.hadamard_gate(q, reg)
        got = reg.read()
    want = a.copy(); ob.iqft(want, n, 0, threads=8)
    for q in range(n):
        ob.hadamard(want, n, q, 8)
    assert np.array_equal(bits(got), bits(want))


def test_queue_semantics(qc, ob):
    """reads flush; reset drops pending gates; measurement sees the fused state; toggling fusion flushes"""
    n = 10
    a = ob.random_state(n, 8)
    with qc.Register(n, 0) as reg:
        reg.write(a); reg.set_fusion(True)
        qc.hadamard_gate(3, reg); qc.hadamard_gate(9, reg)
        w = a.copy(); ob.hadamard(w, n, 3); ob.hadamard(w, n, 9)
        assert np.array_equal(bits(reg.read(5, 100)), bits(w[10:210]))
        qc.hadamard_gate(1, reg)
        qc.reset_register(reg)                      # the queued H(1) must not touch the fresh state
        s = reg.read(); z = np.zeros(2 << n); z[2] = 1.0
        assert np.array_equal(bits(s), bits(z))
        reg.write(a)
        qc.hadamard_gate(0, reg); qc.c_phase_shift_gate(9, 2, 0.4, reg)
        w = a.copy(); ob.hadamard(w, n, 0); ob.cphase(w, n, 9, 2, 0.4)
        r = 0.37
        assert qc.measure_state(reg, r) == ob.measure(w, n, r)
        reg.write(a)
        qc.hadamard_gate(5, reg)
        reg.set_fusion(False)                       # flushes
        qc.hadamard_gate(6, reg)
        w = a.copy(); ob.hadamard(w, n, 5); ob.hadamard(w, n, 6)
        assert np.array_equal(bits(reg.read()), bits(w))


def test_fullsize_fused_sweep_matches_unfused_windows(qc):
    """n = 28: the fused sweep and the per-gate sweep leave identical bits (spot windows)"""
    n = 28
    with qc.Register(n, 0) as r1, qc.Register(n, 0) as r2:
        r1.fill_random(2); r2.fill_random(2)
        r2.set_fusion(True)
        for q in range(n):
            qc.hadamard_gate(q, r1); qc.hadamard_gate(q, r2)
        for k in range(n - 2, n - 12, -1):
            th = math.pi / (1 << (n - 1 - k))
            qc.c_phase_shift_gate(n - 1, k, th, r1); qc.c_phase_shift_gate(n - 1, k, th, r2)
        for s in (0, 12345 << 13, (1 << n) - (1 << 13), 1 << 27):
            assert np.array_equal(bits(r1.read(s, 1 << 13)), bits(r2.read(s, 1 << 13)))
        assert r2.fusion_stats()[0] <= 6


@pytest.mark.parametrize("camruns", [1, 0])
def test_runs_of_modular_multiplies_fold_exactly(qc, ob, tune_guard, camruns):
    """consecutive C_AMODC gates (the Shor ladder) are folded into one gather per tile: same bits as one by one;
    controls inside and outside the tile, identity multipliers, a non-coprime gate breaking the run"""
    qc.tune(fuse_camruns=camruns)
    for (L, M, Cn, mults) in ((14, 5, 21, [2, 4, 16, 4, 16, 1, 2, 20, 3, 5, 10, 8, 11, 13]),
                              (11, 4, 15, [7, 4, 1, 13, 5, 2, 8, 11, 14, 7, 3]),
                              (12, 8, 255, [2, 4, 16, 1, 7, 254, 128, 13, 17, 3, 64, 32])):
        n = L + M
        want = ob.random_state(n, 21)
        with qc.Register(L, M) as reg:
            reg.write(want); reg.set_fusion(True)
            qc.hadamard_gate(n - 1, reg); ob.hadamard(want, n, n - 1)
            for k, a in enumerate(mults):
                ctl = M + (k * 5) % L
                qc.c_amodc_gate(Cn, a, ctl, reg); ob.camodc(want, n, M, Cn, a, ctl)
            qc.c_phase_shift_gate(n - 1, M, 0.3, reg); ob.cphase(want, n, n - 1, M, 0.3)
            got = reg.read()
        assert np.array_equal(bits(got), bits(want)), (L, M, Cn)


def _run_gpu(qc, L, M, Cn, state, prog, fusion):
    with qc.Register(L, M) as reg:
        reg.write(state)
        reg.set_fusion(fusion)
        for g in prog:
            if g[0] == "h":
                qc.hadamard_gate(g[1], reg)
            elif g[0] == "p":
                qc.c_phase_shift_gate(g[1], g[2], g[3], reg)
            else:
                qc.c_amodc_gate(Cn, g[1], g[2], reg)
        return reg.read()


def _run_oracle(ob, n, M, Cn, state, prog):
    want = state.copy()
    for g in prog:
        if g[0] == "h":
            ob.hadamard(want, n, g[1])
        elif g[0] == "p":
            ob.cphase(want, n, g[1], g[2], g[3])
        else:
            ob.camodc(want, n, M, Cn, g[1], g[2])
    return want


def _sparse_state_with_negative_zeros(ob, rs, n, seed):
    st = ob.random_state(n, seed)
    v = st.reshape(-1, 2)
    v[rs.rand(v.shape[0]) < 0.6] = 0.0                      # most amplitudes exactly zero ...
    neg = rs.rand(v.shape[0], 2) < 0.3                      # ... and a third of the zero components are -0
    v[(v == 0.0) & neg] = -0.0
    assert np.signbit(v[v == 0.0]).any()
    return st


@pytest.mark.parametrize("L,M,Cn", [(9, 4, 15), (13, 5, 21), (14, 0, 1)])
def test_sparse_and_negative_zero_states(qc, ob, L, M, Cn):
    """States with exact zeros and -0 entries (a caller can write them).  The reference's mat-vec rewrites -- and so
    canonicalises -- every amplitude at every gate (Q:393-413); the GPU paths canonicalise a caller-written state once,
    before the first gate that runs on it, and every gate kernel keeps the state canonical: same BITS as the oracle, zero
    signs included, fused and per gate."""
    n = L + M
    rs = np.random.RandomState(1000 + n)
    for trial in range(4):
        st = _sparse_state_with_negative_zeros(ob, rs, n, 300 + trial)
        prog = random_program(rs, n, M, Cn, 12 if trial < 2 else 70)     # short programs keep untouched amplitudes around
        plain = _run_gpu(qc, L, M, Cn, st, prog, False)
        fused = _run_gpu(qc, L, M, Cn, st, prog, True)
        want = _run_oracle(ob, n, M, Cn, st, prog)
        assert np.array_equal(bits(fused), bits(want)), f"L={L} M={M} trial {trial}: fused != oracle"
        assert np.array_equal(bits(plain), bits(want)), f"L={L} M={M} trial {trial}: per-gate != oracle"


def test_phase_only_programs_canonicalise_untouched_negative_zeros(qc, ob):
    """no H in the program, so most amplitudes are never touched by a gate kernel -- the reference still turns their -0
    components into +0 at the first gate (its mat-vec writes 0 + 1.0 * x everywhere).  Same bits here: the written state is
    canonicalised once before the first gate.  Without a gate nothing is changed: a read returns the bits that were written."""
    n = 13
    rs = np.random.RandomState(77)
    st = ob.random_state(n, 5)
    v = st.reshape(-1, 2)
    v[rs.rand(v.shape[0]) < 0.5] = -0.0
    prog = []
    for _ in range(40):
        c, t = rs.choice(n, 2, replace=False)
        prog.append(("p", int(c), int(t), float(rs.uniform(-3, 3))))
    plain = _run_gpu(qc, n, 0, 1, st, prog, False)
    fused = _run_gpu(qc, n, 0, 1, st, prog, True)
    want = _run_oracle(ob, n, 0, 1, st, prog)
    assert not np.signbit(want[want == 0.0]).any()
    assert np.array_equal(bits(fused), bits(want)) and np.array_equal(bits(plain), bits(want))
    with qc.Register(n, 0) as reg:
        reg.write(st)
        assert np.array_equal(bits(reg.read()), bits(st))                     # no gate: the caller's bits, -0 included
        for shards in (1,):
            qc.c_phase_shift_gate(3, 1, 0.25, reg)
            w1 = st.copy(); ob.cphase(w1, n, 3, 1, 0.25)
            assert np.array_equal(bits(reg.read()), bits(w1))
    with qc.Register(n, 0, shards=4, devices=qc.spread_devices(4)) as reg:    # the sharded register keeps the same rule
        reg.write(st)
        qc.c_phase_shift_gate(n - 1, 1, 0.25, reg)
        w1 = st.copy(); ob.cphase(w1, n, n - 1, 1, 0.25)
        assert np.array_equal(bits(reg.read()), bits(w1))


@pytest.mark.parametrize("count", [63, 64, 65, 128, 129, 200])
def test_long_phase_runs(qc, ob, count):
    """runs are cut at 64 gates (one ballot word per run): exactly 64, one more, several runs, with controls inside
    the tile, in the lane bits, in the wave bits and outside the tile"""
    n = 16
    rs = np.random.RandomState(count)
    st = ob.random_state(n, 13)
    want = st.copy()
    with qc.Register(n, 0) as reg:
        reg.write(st); reg.set_fusion(True)
        qc.hadamard_gate(7, reg); ob.hadamard(want, n, 7)
        for k in range(count):
            t = int(rs.choice([0, 1, 2, 3, 5, 6, 9, 11, 12, 14, 15]))
            th = float(rs.uniform(-3, 3))
            qc.c_phase_shift_gate(7, t, th, reg); ob.cphase(want, n, 7, t, th)
        qc.hadamard_gate(9, reg); ob.hadamard(want, n, 9)
        got = reg.read()
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("occ", [0, 6, 7, 8])
def test_rounds_kernel_builds(qc, ob, tune_guard, occ):
    """k_fused_rounds built for 6 / 7 / 8 waves per SIMD (with and without the modular-multiply code), 0 = the general
    k_fused kernel: same bits"""
    qc.tune(fuse_rounds_occ=occ)
    rs = np.random.RandomState(occ + 40)
    for (L, M, Cn) in ((13, 5, 21), (17, 0, 1), (15, 4, 15)):
        prog = random_program(rs, L + M, M, Cn, 80)
        got, want, _ = run_both(qc, ob, L, M, Cn, prog, 17)
        assert np.array_equal(bits(got), bits(want)), (occ, L, M)


def test_pure_hadamard_sweep_takes_the_three_pass_geometry_and_keeps_the_bits(qc, tune_guard):
    """a queue of nothing but Hadamards is planned on 2^12-amplitude tiles with 128-B runs when that saves passes (30
    qubits: 3 instead of 4); same bits as one launch per gate, at full size"""
    n = 28
    with qc.Register(n, 0) as r1, qc.Register(n, 0) as r2:
        r1.fill_random(6); r2.fill_random(6)
        r2.set_fusion(True)
        p0 = r2.fusion_stats()[0]
        for q in range(n):
            qc.hadamard_gate(q, r1); qc.hadamard_gate(q, r2)
        r2.flush()
        assert r2.fusion_stats()[0] - p0 == 3              # 12 + 9 + 7 gates
        for s in (0, 4321 << 13, (1 << n) - (1 << 13), 1 << 26):
            assert np.array_equal(bits(r1.read(s, 1 << 13)), bits(r2.read(s, 1 << 13)))
        qc.tune(fuse_hsweep_T=0)                             # the switch: back to the default geometry
        p0 = r2.fusion_stats()[0]
        for q in range(n):
            qc.hadamard_gate(q, r1); qc.hadamard_gate(q, r2)
        r2.flush()
        assert r2.fusion_stats()[0] - p0 == 4
        assert np.array_equal(bits(r1.read(0, 1 << 13)), bits(r2.read(0, 1 << 13)))


# ---- chained passes (round 4): runs of passes through the register's second buffer -------------------------------------------
@pytest.fixture()
def chain_guard(qc):
    keys = ("fuse_chain", "fuse_chain_min_n", "fuse_chain_dir", "fuse_T", "fuse_c", "fuse_T_phase", "fuse_phase_ratio")
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in keys}
    yield
    qc.tune(**old)


def chained_passes(qc, reg):
    import ctypes as C
    out = C.c_ulong(0)
    qc.lib().qcx_chain_stats(reg._h, C.byref(out))
    return out.value


@pytest.mark.parametrize("mode", [1, 2], ids=["exact", "tolerance"])
@pytest.mark.parametrize("L,M,Cn", [(14, 0, 1), (17, 0, 1), (11, 4, 15), (13, 5, 21), (21, 0, 1)])
def test_chained_passes_give_the_same_bits(qc, ob, chain_guard, L, M, Cn, mode):
    """every chained pass reads one buffer under one layout and writes the other under another one; sweeps, the
    inverse-QFT schedule, Shor circuits and random programs against the oracle (bit for bit in the exact mode, 1e-12 in the
    tolerance mode), every observer in between (reads, norm, measurement) sees the identity layout"""
    n = L + M
    qc.tune(fuse_chain=1, fuse_chain_min_n=13)
    rs = np.random.RandomState(n * 7 + mode)
    threads = 8
    with qc.Register(L, M) as reg:
        reg.set_fusion(mode)
        # (fuse_chain_dir: the gathered side of a chained pass -- its stores (0), its reads (1), chosen by the kind of chain (-1))
        for geom in (dict(fuse_T=11, fuse_c=4, fuse_chain_dir=-1), dict(fuse_T=10, fuse_c=4, fuse_chain_dir=0), dict(fuse_T=12, fuse_c=3, fuse_chain_dir=1),
                     dict(fuse_T=11, fuse_c=4, fuse_chain_dir=1), dict(fuse_T=11, fuse_c=4, fuse_chain_dir=0)):
            qc.tune(**geom)
            want = ob.fill_random(n, 21)
            reg.fill_random(21)
            for _ in range(2):
                for q in range(n):
                    qc.hadamard_gate(q, reg); ob.hadamard(want, n, q, threads)
            assert abs(reg.norm2() - ob.norm2(want, n)) < 1e-12
            qc.inverse_QFT(reg); ob.iqft(want, n, M, threads)
            prog = random_program(rs, n, M, Cn, 80)
            for g in prog:
                if g[0] == "h":
                    qc.hadamard_gate(g[1], reg); ob.hadamard(want, n, g[1], threads)
                elif g[0] == "p":
                    qc.c_phase_shift_gate(g[1], g[2], g[3], reg); ob.cphase(want, n, g[1], g[2], g[3], threads)
                else:
                    qc.c_amodc_gate(Cn, g[1], g[2], reg); ob.camodc(want, n, M, Cn, g[1], g[2], threads)
            got = reg.read()
            if mode == 1:
                assert np.array_equal(bits(got), bits(want)), geom
            else:
                assert float(np.max(np.abs(got - want))) <= 1e-12, geom
        assert chained_passes(qc, reg) >= 4
        if M:
            qc.reset_register(reg); qc.quantum_computation(Cn, 2 if Cn == 21 else 7, reg)
            w2 = np.zeros(2 << n); ob.reset(w2, n); ob.quantum_computation(w2, n, M, Cn, 2 if Cn == 21 else 7, threads=threads)
            r = 0.37
            assert qc.measure_state(reg, r) == ob.measure(w2, n, r)
        # a register whose buffer pointer has been handed out works in place from then on (the pointer stays valid)
        before = chained_passes(qc, reg)
        p0 = reg.device_pointer()
        reg.fill_random(5); w3 = ob.fill_random(n, 5)
        for q in range(n):
            qc.hadamard_gate(q, reg); ob.hadamard(w3, n, q, threads)
        got = reg.read()
        assert reg.device_pointer() == p0 and chained_passes(qc, reg) == before
        assert np.array_equal(bits(got), bits(w3)) if mode == 1 else float(np.max(np.abs(got - w3))) <= 1e-12


# ---- round 5, second half: which tile a workgroup takes (fuse_stream_tile) ----------------------------------------------------
@pytest.fixture()
def streams_guard(qc):
    keys = ("fuse_streams_log2", "fuse_streams_pos", "fuse_chain", "fuse_chain_min_n", "fuse_q3_cap_exact", "fuse_x8_cap", "fuse_grid_cap")
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in keys}
    yield
    qc.tune(**old)


@pytest.mark.parametrize("mode", [1, 2], ids=["exact", "tolerance"])
@pytest.mark.parametrize("s,pos1", [(-1, -1), (0, 0), (1, 0), (3, 0), (3, 4), (3, 1), (4, 3), (6, 9), (15, 31)])
def test_tile_order_changes_no_bit(qc, ob, streams_guard, s, pos1, mode):
    """the slot -> tile map of the fused kernels (interleaved streams, the XCD number inside the tile number) is a permutation of
    the tiles for every setting, clamped on small registers, also under persistent grids: Hadamard sweeps (k_fused_q3), the
    inverse QFT (k_fused_x8, exact walk and tolerance round) and a Shor circuit (k_fused_rounds behind the front) against the
    oracle"""
    qc.tune(fuse_streams_log2=s, fuse_streams_pos=pos1, fuse_chain=1, fuse_chain_min_n=13)
    threads = 8
    for (L, M, Cn, a), caps in (((18, 0, 1, 1), (0, 24)), ((13, 0, 1, 1), (0, 3)), ((13, 5, 21, 2), (0, 40))):
        n = L + M
        for cap in caps:                                   # 0: one workgroup per tile; else persistent workgroups that walk the tiles
            qc.tune(fuse_q3_cap_exact=cap, fuse_x8_cap=cap or 65536, fuse_grid_cap=cap)
            with qc.Register(L, M) as reg:
                reg.set_fusion(mode)
                want = ob.fill_random(n, 33); reg.fill_random(33)
                for q in range(n):
                    qc.hadamard_gate(q, reg); ob.hadamard(want, n, q, threads)
                reg.flush()
                qc.inverse_QFT(reg); ob.iqft(want, n, M, threads)
                got = reg.read()
                if mode == 1:
                    assert np.array_equal(bits(got), bits(want)), (n, cap)
                else:
                    assert float(np.max(np.abs(got - want))) <= 1e-12, (n, cap)
                if M:
                    qc.reset_register(reg); qc.quantum_computation(Cn, a, reg)
                    w2 = np.zeros(2 << n); ob.reset(w2, n); ob.quantum_computation(w2, n, M, Cn, a, threads=threads)
                    got = reg.read()
                    if mode == 1:
                        assert np.array_equal(bits(got), bits(w2)), (n, cap)
                    else:
                        assert float(np.max(np.abs(got - w2))) <= 1e-12, (n, cap)


# ---- round 5: the exact walk on 8 amplitudes per thread (k_fused_x8) and the tolerance round on the same shell ---------------
@pytest.fixture()
def x8_guard(qc):
    keys = ("fuse_x8", "fuse_x8_T", "fuse_x8_c", "fuse_x8_map", "fuse_x8_min_tiles_log2", "fuse_x8_ratio", "fuse_x8_cap", "fuse_x8t",
            "fuse_chain", "fuse_chain_min_n", "fuse_phase_ratio", "fuse_gen")
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in keys}
    yield
    qc.tune(**old)


@pytest.mark.parametrize("T,c", [(12, 4), (12, 3), (11, 4), (11, 3), (10, 4), (10, 2)])
@pytest.mark.parametrize("xmap", [1, 0], ids=["maps", "ascending"])
def test_exact_walk_on_8_amplitudes_every_tile_size(qc, ob, x8_guard, T, c, xmap):
    """k_fused_x8 at every tile size it is built for (2^10, 2^11, 2^12), with the planner's thread maps (wave-private parts, no
    barriers between rounds) and with plain ascending maps, in place and chained, few tiles and many, grid-stride and one
    workgroup per tile: random programs (all register patterns of a run, long runs, rounds with one or two Hadamards) and the
    inverse-QFT schedule, bit for bit against the oracle"""
    qc.tune(fuse_x8=1, fuse_x8_T=T, fuse_x8_c=c, fuse_x8_map=xmap, fuse_x8_min_tiles_log2=0, fuse_x8_ratio=0, fuse_chain_min_n=13)
    rs = np.random.RandomState(100 * T + 10 * c + xmap)
    for L, M, Cn, chain, cap in ((13, 0, 1, 1, 65536), (16, 0, 1, 1, 7), (12, 4, 15, 0, 65536), (T, 0, 1, 1, 65536), (17, 0, 1, 0, 65536)):
        n = L + M
        qc.tune(fuse_chain=chain, fuse_x8_cap=cap)
        prog = [g for g in random_program(rs, n, M, Cn, 90) if g[0] != "c"]
        # a long run behind one Hadamard (cut at 63 gates) and phases whose two qubits are both register bits of a round
        top = n - 1
        prog += [("h", top)] + [("p", top, int(rs.randint(M, top)), float(rs.uniform(-3, 3))) for _ in range(70)]
        prog += [("h", top - 1), ("p", top, top - 1, 0.3), ("h", top - 2), ("p", top - 1, top - 2, 0.4), ("p", top, top - 2, 0.5)]
        got, want, stats = run_both(qc, ob, L, M, Cn, prog, 90 + n)
        assert np.array_equal(bits(got), bits(want)), (T, c, xmap, L, M, chain, cap)
    # the schedule of Q:678-690
    n = 16
    qc.tune(fuse_chain=1, fuse_x8_cap=65536)
    want = ob.random_state(n, 5)
    with qc.Register(n, 0) as reg:
        reg.write(want)
        qc.inverse_QFT(reg)
        ob.iqft(want, n, 0, 4)
        assert np.array_equal(bits(reg.read()), bits(want))


@pytest.mark.parametrize("C,L,M,a", [(21, 12, 5, 2), (15, 13, 4, 7)])
def test_exact_walk_generates_the_circuit_front(qc, ob, x8_guard, C, L, M, a):
    """GenFront inside a k_fused_x8 pass (the tile generated under the swizzle): Shor circuits whose first pass behind the front
    is a phase-carrying pass, with the compact chain switched off so that the generated fill of THIS kernel runs"""
    old = qc.lib().qcx_tune_get(b"fuse_compact")
    try:
        qc.tune(fuse_compact=0, fuse_gen_cols=0, fuse_x8_min_tiles_log2=0)
        n = L + M
        want = np.zeros(2 << n); ob.reset(want, n); ob.quantum_computation(want, n, M, C, a, threads=4)
        with qc.Register(L, M) as reg:
            qc.reset_register(reg); qc.quantum_computation(C, a, reg)
            assert np.array_equal(bits(reg.read()), bits(want))
    finally:
        qc.tune(fuse_compact=old, fuse_gen_cols=1)


# ---- round 5: the plan cache -------------------------------------------------------------------------------------------------
def _plan_hits(qc, reg):
    import ctypes as C
    out = C.c_ulong(0)
    qc.lib().qcx_plan_cache_stats(reg._h, C.byref(out))
    return out.value


@pytest.mark.parametrize("n,M,Cn,a", [(17, 5, 21, 2), (13, 4, 15, 7), (9, 4, 15, 7)], ids=["n=17: compact chains", "n=13: front generated in the first pass", "n=9: front written by its own pass"])
@pytest.mark.parametrize("mode", [0, 1, 2], ids=["whole-circuit calls", "every gate queued", "tolerance"])
def test_plan_cache_reuses_a_plan_only_for_identical_inputs(qc, ob, mode, n, M, Cn, a):
    """a flush whose inputs are those of the last one (shape, mode, knobs, front, gate list) reuses its plan and records; anything
    else plans afresh.  Same bits as without the cache, on different states, across interleaved other flushes."""
    L = n - M
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in ("fuse_plan_cache", "fuse_T")}
    try:
        outs = {}
        for cache in (1, 0):
            qc.tune(fuse_plan_cache=cache)
            res = []
            with qc.Register(L, M) as reg:
                reg.set_fusion(mode)
                h0 = _plan_hits(qc, reg)
                # (1) the inverse QFT on three different dense states: one plan, two hits
                for seed in (3, 4, 5):
                    reg.fill_random(seed); qc.inverse_QFT(reg); res.append(reg.read())
                h1 = _plan_hits(qc, reg)
                # (2) a different list in between (a Hadamard sweep), then the inverse QFT again: planned afresh, then a hit
                reg.fill_random(6)
                for q in range(n):
                    qc.hadamard_gate(q, reg)
                res.append(reg.read())
                reg.fill_random(7); qc.inverse_QFT(reg); res.append(reg.read())
                reg.fill_random(8); qc.inverse_QFT(reg); res.append(reg.read())
                h2 = _plan_hits(qc, reg)
                # (3) a knob changes the plan: no hit
                qc.tune(fuse_T=10 if old["fuse_T"] != 10 else 11)
                reg.fill_random(9); qc.inverse_QFT(reg); res.append(reg.read())
                h3 = _plan_hits(qc, reg)
                qc.tune(fuse_T=old["fuse_T"])
                # (4) period-finding attempts: the same circuit behind a reset, measured (compact chains); then read
                picks = []
                for r in (0.11, 0.52, 0.93):
                    qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); picks.append(qc.measure_state(reg, r))
                qc.reset_register(reg); qc.quantum_computation(Cn, a, reg); res.append(reg.read())
                h4 = _plan_hits(qc, reg)
                # (5) a collapse to another basis state gives another front: the plan of the reset state must not be reused blindly
                for l in range(M, n):
                    qc.hadamard_gate(l, reg)
                qc.inverse_QFT(reg); res.append(reg.read())
                if cache:
                    # (mode 0 queues only the whole-circuit calls: the sweep in between runs gate by gate and leaves the cached plan alone)
                    assert h1 - h0 == 2 and h2 - h1 == (2 if mode == 0 else 1) and h3 == h2, (h0, h1, h2, h3)
                    assert h4 - h3 >= 3, (h3, h4)
                else:
                    assert h4 == h0
            outs[cache] = (res, picks)
        for x, y in zip(outs[1][0], outs[0][0]):
            assert np.array_equal(bits(x), bits(y))
        assert outs[1][1] == outs[0][1]
        # and against the oracle: the first inverse QFT and the circuit
        want = ob.fill_random(n, 3); ob.iqft(want, n, M, 8)
        w2 = np.zeros(2 << n); ob.reset(w2, n); ob.quantum_computation(w2, n, M, Cn, a, threads=8)
        if mode == 2:
            assert float(np.max(np.abs(outs[1][0][0] - want))) <= 1e-12 and float(np.max(np.abs(outs[1][0][7] - w2))) <= 1e-12
        else:
            assert np.array_equal(bits(outs[1][0][0]), bits(want)) and np.array_equal(bits(outs[1][0][7]), bits(w2))
            assert outs[1][1][0] == ob.measure(w2.copy(), n, 0.11)
    finally:
        qc.tune(**old)
