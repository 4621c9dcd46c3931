"""CPU: the fused-pass PLANNER (csrc/qcx_fuse.inc.h, host code) and the record formats the pass kernels interpret.
`qcx_fusion_plan` returns the plan without touching a GPU; tests/fuse_emulator.py applies its records with the kernels'
arithmetic; the result must be the oracle's state, bit for bit.  The GPU suite then only has to show that the kernels
do what the emulator does (tests/test_gpu_fusion.py)."""
import math

import numpy as np
import pytest

import fuse_emulator as emu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


TUNE_KEYS = ("fuse_T", "fuse_c", "fuse_rounds", "fuse_camruns", "fuse_T_phase", "fuse_c_phase", "fuse_phase_ratio", "fuse_tol_T", "fuse_q3",
             "fuse_x8", "fuse_x8_T", "fuse_x8_c", "fuse_x8_map", "fuse_x8_min_tiles_log2")


@pytest.fixture()
def tune_guard(qc):
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in TUNE_KEYS}
    # (by default registers of fewer than four tiles keep the radix-4 kernels; the emulator tests run on small registers and want the
    #  walk on 8 amplitudes wherever the planner would take it at full size)
    qc.tune(fuse_x8_min_tiles_log2=0)
    yield
    qc.tune(**old)


def random_program(rs, n, M, Cn, length):
    """(descs for the planner, steps for the oracle)"""
    descs, steps = [], []
    for _ in range(length):
        k = rs.randint(0, 10)
        if k < 4:
            q = int(rs.randint(0, n))
            descs.append((0, q, 0, 0.0, 0.0, 0, 0)); steps.append(("h", q))
        elif k < 9 or M == 0:
            c, t = (int(x) for x in rs.choice(n, 2, replace=False))
            th = float(rs.uniform(-3, 3)) if k % 2 else math.pi / (1 << int(rs.randint(1, 12)))
            steps.append(("p", c, t, th))
            descs.append((1, 0, (1 << c) | (1 << t), 0.0, 0.0, 0, 0))
        else:
            atox, ctl = int(rs.randint(1, 4 * Cn)), int(rs.randint(M, n))
            descs.append((2, ctl, 0, 0.0, 0.0, Cn, atox % Cn)); steps.append(("c", atox, ctl))
    return descs, steps


def fill_polar(qc, descs, steps):
    out = []
    for d, s in zip(descs, steps):
        if s[0] == "p":
            c, sn = qc.polar(s[3])
            d = (1, 0, d[2], c, sn, 0, 0)
        out.append(d)
    return out


def oracle_run(ob, state, n, M, Cn, steps):
    for s in steps:
        if s[0] == "h":
            ob.hadamard(state, n, s[1])
        elif s[0] == "p":
            ob.cphase(state, n, s[1], s[2], s[3])
        else:
            ob.camodc(state, n, M, Cn, s[1], s[2])


CASES = [(12, 0, 1), (13, 4, 15), (14, 5, 21), (11, 4, 15), (12, 6, 35)]
TUNES = [dict(), dict(fuse_T=10, fuse_c=4), dict(fuse_T=12, fuse_c=3), dict(fuse_T=9, fuse_c=4), dict(fuse_rounds=0),
         dict(fuse_T_phase=10, fuse_phase_ratio=1), dict(fuse_T_phase=12, fuse_c_phase=2, fuse_phase_ratio=1), dict(fuse_camruns=0),
         # round 5: phase-dominated bit-exact passes take the walk on 8 amplitudes per thread (FUSE_ROUND8) by default; the
         # radix-4 walk stays selectable
         dict(fuse_x8=0, fuse_phase_ratio=1), dict(fuse_x8_T=11, fuse_x8_c=3, fuse_phase_ratio=1), dict(fuse_x8_T=10, fuse_phase_ratio=1),
         dict(fuse_x8_map=0, fuse_phase_ratio=1), dict(fuse_x8_T=12, fuse_x8_c=2, fuse_phase_ratio=0)]


@pytest.mark.parametrize("tune", TUNES, ids=lambda t: ",".join(f"{k}={v}" for k, v in t.items()) or "default")
@pytest.mark.parametrize("n,M,Cn", CASES)
def test_planned_records_reproduce_the_oracle(qc, ob, tune_guard, n, M, Cn, tune):
    qc.tune(**tune)
    rs = np.random.RandomState(n * 100 + M + len(tune) * 7)
    for trial in range(2):
        descs, steps = random_program(rs, n, M, Cn, 70)
        descs = fill_polar(qc, descs, steps)
        actions, recs, nrec = qc.fusion_plan(n, M, descs)
        state = ob.random_state(n, 40 + trial)
        want = state.copy()
        oracle_run(ob, want, n, M, Cn, steps)
        totals = emu.run_plan(state, n, M, descs, actions, recs, ob)
        assert np.array_equal(bits(state), bits(want)), f"n={n} M={M} {tune} trial {trial}: {totals}"
        assert totals["passes"] >= 1


def iqft_descs(qc, n, M):
    descs = []
    for l in range(n - 1, M - 1, -1):
        descs.append((0, l, 0, 0.0, 0.0, 0, 0))
        for k in range(l - 1, M - 1, -1):
            c, s = qc.polar(math.pi / float(1 << (l - k)))
            descs.append((1, 0, (1 << l) | (1 << k), c, s, 0, 0))
    return descs


def test_plan_of_the_n28_iqft(qc, tune_guard):
    """config 3: 28 H + 378 phases.  Round 5: 3 passes of the exact walk on 8 amplitudes per thread (2^12 tiles, 8 hot bits next to
    c = 4; the last one the contiguous low tile with 12 Hadamards); with fuse_x8 = 0 the radix-4 plan of rounds 2-4: 4 passes,
    the three phase-dominated ones on 2^10 tiles"""
    n = 28
    descs = iqft_descs(qc, n, 0)
    actions, recs, nrec = qc.fusion_plan(n, 0, descs)
    assert len(descs) == 406 and [a.fused for a in actions] == [1, 1, 1]
    assert sum(a.ngates for a in actions) == 406
    assert [(a.T, a.c, a.nopipe) for a in actions] == [(12, 4, 1), (12, 4, 1), (12, 4, 1)]
    hot = [[a.hbit[j] for j in range(a.nh)] for a in actions]
    assert hot[0] == list(range(20, 28)) and hot[1] == list(range(12, 20)) and hot[2] == list(range(4, 12))
    for a in actions:
        assert a.rounds_form == 1
        R = [recs[a.rec_off + k] for k in range(a.nops)]
        assert all((r.type & 0xFF) in (emu.FUSE_ROUND8, emu.FUSE_H, emu.FUSE_PRUN, emu.FUSE_PHASE) for r in R)
        assert sum(1 for r in R if (r.type & 0xFF) == emu.FUSE_ROUND8) == (4 if a is actions[2] else 3)     # three Hadamards per round
        runs = [r.type >> 16 for r in R if (r.type & 0xFF) == emu.FUSE_PRUN]
        assert runs and max(runs) <= 63
        assert sum(runs) + sum(1 for r in R if (r.type & 0xFF) == emu.FUSE_H) == a.ngates
    qc.tune(fuse_x8=0)
    actions, recs, nrec = qc.fusion_plan(n, 0, descs)
    assert [a.fused for a in actions] == [1, 1, 1, 1]
    assert [(a.T, a.c, a.nopipe) for a in actions] == [(10, 4, 1), (10, 4, 1), (10, 4, 1), (11, 4, 0)]
    hot = [[a.hbit[j] for j in range(a.nh)] for a in actions]
    assert hot[0] == [22, 23, 24, 25, 26, 27] and hot[1] == [16, 17, 18, 19, 20, 21] and hot[2] == [10, 11, 12, 13, 14, 15]
    for a in actions:
        assert a.rounds_form == 1
        R = [recs[a.rec_off + k] for k in range(a.nops)]
        runs = [r.type >> 16 for r in R if (r.type & 0xFF) == emu.FUSE_PRUN]
        assert runs and max(runs) <= 64
        assert sum(runs) + sum(1 for r in R if (r.type & 0xFF) == emu.FUSE_H) == a.ngates


@pytest.mark.parametrize("n,M", [(12, 0), (14, 0), (15, 4), (16, 5)])
@pytest.mark.parametrize("tune", [dict(), dict(fuse_x8_map=0), dict(fuse_x8_T=11, fuse_x8_c=3), dict(fuse_x8_T=10)],
                         ids=lambda t: ",".join(f"{k}={v}" for k, v in t.items()) or "default")
def test_radix8_exact_rounds_on_the_iqft_schedule(qc, ob, tune_guard, n, M, tune):
    """the schedule of Q:678-690 through FUSE_ROUND8 records against the oracle, bit for bit; every round's thread map is a
    permutation of its non-register tile bits, and phases whose target is a register bit of the round run on whole waves"""
    qc.tune(**tune)
    descs = iqft_descs(qc, n, M)
    steps = []
    for l in range(n - 1, M - 1, -1):
        steps.append(("h", l))
        for k in range(l - 1, M - 1, -1):
            steps.append(("p", l, k, math.pi / float(1 << (l - k))))
    actions, recs, nrec = qc.fusion_plan(n, M, descs)
    kinds = {recs[a.rec_off + k].type & 0xFF for a in actions if a.fused for k in range(a.nops)}
    assert emu.FUSE_ROUND8 in kinds
    state = ob.random_state(n, 9)
    want = state.copy()
    oracle_run(ob, want, n, M, 1, steps)
    totals = emu.run_plan(state, n, M, descs, actions, recs, ob)
    assert np.array_equal(bits(state), bits(want)), totals
    assert totals["h"] == n - M and totals["maps"]


def test_radix8_thread_maps_of_the_n28_iqft(qc, tune_guard):
    """the three passes of config 3: in the two passes with outside hot bits the wave number rides on three of the four filler
    bits for the WHOLE pass (no barrier between rounds: header bit 25), in the last pass (12 Hadamards on the tile's own 12
    bits) it moves once; every round's lane order is free of LDS bank conflicts under the kernel's swizzle; and the lanes
    take the bits that fewest gates test"""
    n = 28
    descs = iqft_descs(qc, n, 0)
    actions, recs, nrec = qc.fusion_plan(n, 0, descs)
    nobar = []
    for a in actions:
        R = [recs[a.rec_off + k] for k in range(a.nops)]
        heads = [r for r in R if (r.type & 0xFF) == emu.FUSE_ROUND8]
        nobar.append([(h.a >> 25) & 1 for h in heads])
        for h in heads:
            import struct
            tmap = struct.unpack("<Q", struct.pack("<d", h.c))[0]
            tb = [(tmap >> (4 * k)) & 15 for k in range(9)]
            assert emu.x8_lane_conflicts(tb[:6]) == (1, 1), (tb, emu.x8_lane_conflicts(tb[:6]))
        if a is not actions[2]:
            for h in heads:
                tmap = struct.unpack("<Q", struct.pack("<d", h.c))[0]
                assert all(((tmap >> (4 * k)) & 15) < 4 for k in (6, 7, 8)), "the waves of a pass with outside hot bits sit on its fillers"
    assert nobar[0] == [0, 1, 1] and nobar[1] == [0, 1, 1] and sum(nobar[2]) >= 2, nobar
    qc.tune(fuse_x8_map=0)
    actions, recs, nrec = qc.fusion_plan(n, 0, descs)
    assert not any((recs[a.rec_off + k].a >> 25) & 1 for a in actions for k in range(a.nops) if (recs[a.rec_off + k].type & 0xFF) == emu.FUSE_ROUND8)


def test_plan_of_the_n30_shor_circuit(qc, tune_guard):
    """config 5 on one GPU: 25 H, 25 controlled multiplies (one folded run), 25 H + 300 phases"""
    L, M, Cn, a = 25, 5, 21, 2
    n = L + M
    descs = [(0, l, 0, 0.0, 0.0, 0, 0) for l in range(M, n)]
    x = a % Cn
    for l in range(M, n):
        descs.append((2, l, 0, 0.0, 0.0, Cn, x)); x = (x * x) % Cn
    descs += iqft_descs(qc, n, M)
    actions, recs, nrec = qc.fusion_plan(n, M, descs)
    assert len(descs) == 375 and all(a.fused for a in actions) and sum(a.ngates for a in actions) == 375
    assert len(actions) <= 8
    camruns = [recs[a.rec_off + k] for a in actions for k in range(a.nops) if (recs[a.rec_off + k].type & 0xFF) == emu.FUSE_CAMRUN]
    assert len(camruns) == 1 and (camruns[0].a & 0xFFFF) == 25


def test_single_gates_and_oversized_multiplies_stay_stand_alone(qc, tune_guard):
    n, M = 13, 4
    c, s = qc.polar(0.3)
    descs = [(0, 5, 0, 0.0, 0.0, 0, 0)]
    actions, _, _ = qc.fusion_plan(n, M, descs)
    assert len(actions) == 1 and not actions[0].fused                    # alone: its tuned kernel
    # C > 2^M (the reference's undersized-M case, table form) never joins a pass
    descs = [(0, 5, 0, 0.0, 0.0, 0, 0), (1, 0, 0b110000, c, s, 0, 0), (2, 6, 0, 0.0, 0.0, 21, 2), (0, 7, 0, 0.0, 0.0, 0, 0),
             (1, 0, 0b11, c, s, 0, 0)]
    actions, _, _ = qc.fusion_plan(n, M, descs)
    assert [(a.fused, a.first_gate, a.ngates) for a in actions] == [(1, 0, 2), (0, 2, 1), (1, 3, 2)]


@pytest.mark.parametrize("count", [63, 64, 65, 128, 129, 200])
def test_long_phase_runs_are_cut_at_64_gates(qc, ob, tune_guard, count):
    """a run holds at most 64 gates (one ballot word): longer stretches of equal-selection phases become several runs"""
    n = 13
    rs = np.random.RandomState(count)
    steps = [("h", 7)]
    for k in range(count):
        steps.append(("p", 7, int(rs.choice([0, 1, 2, 3, 5, 6, 9, 11, 12])), float(rs.uniform(-3, 3))))
    steps.append(("h", 9))
    descs = []
    for s in steps:
        if s[0] == "h":
            descs.append((0, s[1], 0, 0.0, 0.0, 0, 0))
        else:
            c, sn = qc.polar(s[3])
            descs.append((1, 0, (1 << s[1]) | (1 << s[2]), c, sn, 0, 0))
    actions, recs, nrec = qc.fusion_plan(n, 0, descs)
    runs = [recs[k].type >> 16 for k in range(nrec) if (recs[k].type & 0xFF) == emu.FUSE_PRUN]
    assert sum(runs) == count and max(runs) <= 64 and len(runs) >= -(-count // 64)
    state = ob.random_state(n, 9)
    want = state.copy()
    oracle_run(ob, want, n, 0, 1, steps)
    emu.run_plan(state, n, 0, descs, actions, recs, ob)
    assert np.array_equal(bits(state), bits(want))


# ---- tolerance mode (qcx_set_fusion(reg, 2)): merged diagonals, NOT bit-exact -----------------------------------------
TOL = 1e-12       # max |delta amplitude| against the oracle on these n <= 14 circuits (north_star allows 1e-10)


def shor_descs(qc, Cn, L, M, a):
    """the gate list of quantum_computation (Q:712-737) with exact modular powers"""
    n = L + M
    descs = [(0, l, 0, 0.0, 0.0, 0, 0) for l in range(M, n)]
    x = a % Cn
    for l in range(M, n):
        descs.append((2, l, 0, 0.0, 0.0, Cn, x)); x = (x * x) % Cn
    return descs + iqft_descs(qc, n, M)


def max_delta(a, b):
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return float(np.max(np.hypot(d[0::2], d[1::2])))


@pytest.mark.parametrize("n,M", [(12, 0), (13, 0), (14, 4), (11, 5)])
@pytest.mark.parametrize("T,c", [(10, 4), (11, 4), (12, 3), (12, 4)])
def test_tolerance_mode_iqft_merges_every_run_into_one_diagonal(qc, ob, tune_guard, n, M, T, c):
    if T > n:
        pytest.skip("tile larger than the register")
    qc.tune(fuse_T=T, fuse_c=c, fuse_T_phase=0, fuse_tol_T=0, fuse_q3=0)         # the geometry under test, not the planner's own choice
    descs = iqft_descs(qc, n, M)
    L = n - M
    acts, recs, _ = qc.fusion_plan(n, M, descs, mode=2)
    state = ob.fill_random(n, 11)
    want = state.copy()
    ob.iqft(want, n, M)
    tot = emu.run_plan(state, n, M, descs, acts, recs, ob)
    assert max_delta(state, want) <= TOL
    assert not np.array_equal(bits(state), bits(want)) or L < 3          # it really is the other arithmetic
    # every H but the last is followed by a run of phases sharing its qubit: one diagonal each (a diagonal that lands in a
    # round of another shape gets its phases back; a lone last H may run stand-alone)
    assert max(L - 3, 0) <= tot["diags"] <= max(L - 1, 0) and L - 1 <= tot["h"] <= L
    exact_acts, _, _ = qc.fusion_plan(n, M, descs, mode=1)
    assert sum(a.fused for a in acts) <= sum(a.fused for a in exact_acts)


@pytest.mark.parametrize("Cn,L,M,a", [(15, 8, 4, 7), (21, 9, 5, 2), (33, 8, 6, 5)])
def test_tolerance_mode_shor_circuit(qc, ob, tune_guard, Cn, L, M, a):
    n = L + M
    descs = shor_descs(qc, Cn, L, M, a)
    acts, recs, _ = qc.fusion_plan(n, M, descs, mode=2)
    state = np.zeros(2 << n); ob.reset(state, n)
    want = state.copy()
    ob.quantum_computation(want, n, M, Cn, a)
    tot = emu.run_plan(state, n, M, descs, acts, recs, ob)
    assert max_delta(state, want) <= TOL and tot["diags"] >= L - 3
    # the measurement distribution is untouched at this level: same cumulative probabilities to 1e-12
    p0 = np.cumsum(state[0::2] ** 2 + state[1::2] ** 2); p1 = np.cumsum(want[0::2] ** 2 + want[1::2] ** 2)
    assert np.max(np.abs(p0 - p1)) < 1e-12


@pytest.mark.parametrize("seed", range(6))
def test_tolerance_mode_random_programs(qc, ob, tune_guard, seed):
    """random programs with plenty of phase runs sharing a qubit (and phases that share none), every gate kind, several
    geometries; repeated targets inside a run multiply their factors"""
    rs = np.random.RandomState(100 + seed)
    n, M, Cn = int(rs.randint(10, 15)), int(rs.choice([0, 3, 4])), 0
    if M:
        Cn = int(rs.randint(3, 1 << M))
    T, c = [(10, 4), (11, 4), (12, 3), (11, 2)][seed % 4]
    if T > n:
        T = n
    qc.tune(fuse_T=T, fuse_c=min(c, T))
    descs, steps = [], []
    for _ in range(int(rs.randint(30, 80))):
        k = rs.randint(0, 10)
        if k < 3:
            q = int(rs.randint(0, n)); descs.append((0, q, 0, 0.0, 0.0, 0, 0)); steps.append(("h", q))
        elif k < 8 or M == 0:
            ctl = int(rs.randint(0, n))
            for _ in range(int(rs.randint(1, 9))):                         # a run sharing ctl (targets may repeat)
                t = int(rs.randint(0, n))
                if t == ctl:
                    continue
                th = float(rs.uniform(-3, 3))
                steps.append(("p", ctl, t, th)); descs.append((1, 0, (1 << ctl) | (1 << t), 0.0, 0.0, 0, 0))
        else:
            atox, ctl = int(rs.randint(1, 4 * Cn)), int(rs.randint(M, n))
            descs.append((2, ctl, 0, 0.0, 0.0, Cn, atox % Cn)); steps.append(("c", atox, ctl))
    descs = fill_polar(qc, descs, steps)
    acts, recs, _ = qc.fusion_plan(n, M, descs, mode=2)
    state = ob.fill_random(n, seed)
    want = state.copy()
    oracle_run(ob, want, n, M, Cn, steps)
    emu.run_plan(state, n, M, descs, acts, recs, ob)
    assert max_delta(state, want) <= TOL, (n, M, T, c)


def test_tolerance_mode_plan_of_the_n28_iqft(qc, tune_guard):
    """config 3 in tolerance mode: 28 H + 378 phases -> THREE passes of radix-8 fast rounds on 2^12 tiles (8 hot bits each,
    k_fused_q3), 27 diagonals; with the radix-8 form off: four passes of radix-4 fast rounds on 2^10 tiles"""
    n = 28
    descs = iqft_descs(qc, n, 0)
    actions, recs, nrec = qc.fusion_plan(n, 0, descs, mode=2)
    assert [a.fused for a in actions] == [1, 1, 1] and sum(a.ngates for a in actions) == 406
    assert [(a.T, a.c) for a in actions] == [(12, 4)] * 3
    assert sum(a.diag_cnt for a in actions) == 27
    for a in actions:
        kinds = {recs[a.rec_off + k].type & 0xFF for k in range(0, a.nops, 2)}
        assert kinds == {emu.FUSE_QROUND3}, kinds
    qc.tune(fuse_q3=0)
    actions, recs, nrec = qc.fusion_plan(n, 0, descs, mode=2)
    assert [(a.T, a.c) for a in actions] == [(10, 4)] * 4 and sum(a.diag_cnt for a in actions) == 27
    for a in actions:
        assert {recs[a.rec_off + k].type & 0xFF for k in range(0, a.nops, 2)} == {emu.FUSE_QROUND}


@pytest.mark.parametrize("n,M", [(12, 0), (12, 3), (20, 0), (21, 5)])
def test_tolerance_mode_radix8_rounds(qc, ob, tune_guard, n, M):
    """registers where the radix-8 form saves a pass (2^12 tiles against 2^10 / 2^11): emulated against the oracle"""
    descs = iqft_descs(qc, n, M)
    acts, recs, _ = qc.fusion_plan(n, M, descs, mode=2)
    state = ob.fill_random(n, 5)
    want = state.copy(); ob.iqft(want, n, M)
    tot = emu.run_plan(state, n, M, descs, acts, recs, ob)
    assert max_delta(state, want) <= TOL
    q3 = [a for a in acts if a.fused and (recs[a.rec_off].type & 0xFF) == emu.FUSE_QROUND3]
    assert len(q3) == (n - max(M, 4) + 7) // 8, [(a.T, a.c) for a in acts]


def test_mode_1_plan_is_unchanged_by_the_tolerance_code(qc, ob, tune_guard):
    """the bit-exact planner's output carries no diagonal: same records through either entry point"""
    descs = iqft_descs(qc, 13, 0)
    a1, r1, n1 = qc.fusion_plan(13, 0, descs)
    a2, r2, n2 = qc.fusion_plan(13, 0, descs, mode=1)
    assert n1 == n2 and all(x.diag_cnt == 0 for x in a1)
    assert bytes(memoryview(r1).cast("B"))[:32 * n1] == bytes(memoryview(r2).cast("B"))[:32 * n2]


@pytest.mark.parametrize("n", [21])
@pytest.mark.parametrize("mode", [1, 2])
def test_hadamard_sweep_takes_the_exact_radix8_form(qc, ob, tune_guard, n, mode):
    """an all-Hadamard queue on 2^12 tiles (12 + 9 + 9 hot bits at n = 30): radix-8 rounds with EXACT butterflies
    (k_fused_q3<.., EXACT>), bit for bit -- in the tolerance mode too, there is nothing to merge"""
    descs = [(0, q, 0, 0.0, 0.0, 0, 0) for q in range(n)]
    acts, recs, _ = qc.fusion_plan(n, 0, descs, mode=mode)
    state = ob.fill_random(n, 2)
    want = state.copy()
    for q in range(n):
        ob.hadamard(want, n, q, 8)
    emu.run_plan(state, n, 0, descs, acts, recs, ob)
    assert np.array_equal(bits(state), bits(want))
    fused = [a for a in acts if a.fused]
    assert fused and all((recs[a.rec_off].type & 0xFF) == emu.FUSE_QROUND3 and a.diag_cnt == 0 and (a.T, a.c) == (12, 3) for a in fused)
    qc.tune(fuse_q3=0)
    acts, recs, _ = qc.fusion_plan(n, 0, descs, mode=mode)
    assert all((recs[a.rec_off].type & 0xFF) == emu.FUSE_ROUND for a in acts if a.fused)


@pytest.mark.parametrize("seed", range(6))
def test_tolerance_mode_shard_level_lists(qc, ob, tune_guard, seed):
    """what a sharded register hands to qcx_shard_run_fused_mode(2, ...): phases whose other qubit lives in the shard id arrive
    with ONE mask bit (the shard's bit is 1) -- they join the diagonal of their run as its constant factor"""
    rs = np.random.RandomState(700 + seed)
    n = int(rs.randint(11, 15))
    descs, ops = [], []
    for l in range(n - 1, max(n - 9, 0), -1):
        descs.append((0, l, 0, 0.0, 0.0, 0, 0)); ops.append(("h", l))
        for k in range(l - 1, -1, -1):
            c, s = qc.polar(math.pi / float(1 << (l - k)))
            descs.append((1, 0, (1 << l) | (1 << k), c, s, 0, 0)); ops.append(("p2", l, k, c, s))
        for _ in range(int(rs.randint(0, 4))):                         # targets in the shard id: a phase on l alone
            th = float(rs.uniform(-1, 1)); c, s = qc.polar(th)
            descs.append((1, 0, 1 << l, c, s, 0, 0)); ops.append(("p1", l, c, s))
    acts, recs, _ = qc.fusion_plan(n, 0, descs, mode=2)
    state = ob.fill_random(n, seed)
    z = state[0::2] + 1j * state[1::2]
    idx = np.arange(1 << n)
    for o in ops:                                                      # independent complex128 evaluation (tolerance compare)
        if o[0] == "h":
            b = 1 << o[1]; lo = idx[(idx & b) == 0]; a0, a1 = z[lo].copy(), z[lo | b].copy()
            z[lo], z[lo | b] = (a0 + a1) * emu.SQRT1_2, (a0 - a1) * emu.SQRT1_2
        elif o[0] == "p2":
            sel = ((idx >> o[1]) & 1).astype(bool) & ((idx >> o[2]) & 1).astype(bool); z[sel] *= complex(o[3], o[4])
        else:
            sel = ((idx >> o[1]) & 1).astype(bool); z[sel] *= complex(o[2], o[3])
    class OneBitOracle:            # run_plan only needs the oracle for stand-alone gates: none of the one-bit phases stays alone here
        def __getattr__(self, name):
            return getattr(ob, name)
    tot = emu.run_plan(state, n, 0, descs, acts, recs, OneBitOracle())
    got = state[0::2] + 1j * state[1::2]
    assert float(np.max(np.abs(got - z))) <= 1e-12
    assert tot["diags"] >= 6


# ---- chained passes (round 4): the same records under another numbering, other addresses -----------------------------------
CHAIN_KEYS = TUNE_KEYS + ("fuse_chain", "fuse_chain_min_n", "fuse_chain_dir", "fuse_hsweep_T", "fuse_hsweep_c")


@pytest.fixture()
def chain_guard(qc):
    old = {k: qc.lib().qcx_tune_get(k.encode()) for k in CHAIN_KEYS}
    qc.tune(fuse_chain=1, fuse_chain_min_n=13, fuse_x8_min_tiles_log2=0)
    yield
    qc.tune(**old)


def sweep_descs(n, reps=1, order=None):
    return [(0, q, 0, 0.0, 0.0, 0, 0) for _ in range(reps) for q in (order or range(n))]


@pytest.mark.parametrize("mode", [1, 2], ids=["exact", "tolerance"])
@pytest.mark.parametrize("n,M,Cn", [(14, 0, 1), (16, 0, 1), (15, 4, 15), (16, 5, 21)])
def test_chained_plans_reproduce_the_oracle(qc, ob, chain_guard, n, M, Cn, mode):
    """plans of a register with a second buffer (mode | 4): sweeps, the inverse-QFT schedule, Shor circuits and random
    programs.  The emulator applies the records on logical indices (bit-exact vs the oracle in mode 1, 1e-12 in mode 2);
    check_chain_addressing follows the kernels' address arithmetic through every pass."""
    rs = np.random.RandomState(1000 + n + mode)
    programs = []
    programs.append(("sweep", sweep_descs(n, 2), [("h", q) for _ in range(2) for q in range(n)]))
    d = iqft_descs(qc, n, M)
    st = []
    for l in range(n - 1, M - 1, -1):
        st.append(("h", l))
        for k in range(l - 1, M - 1, -1):
            st.append(("p", l, k, math.pi / float(1 << (l - k))))
    programs.append(("iqft", d, st))
    for trial in range(2):
        dd, ss = random_program(rs, n, M, Cn, 120)
        programs.append((f"random{trial}", fill_polar(qc, dd, ss), ss))
    total_chained = 0
    for name, descs, steps in programs:
        # (fuse_chain_dir: which side of a chained pass is the gathered one -- stores (0), reads (1), by the kind of chain (-1))
        for tune in (dict(), dict(fuse_T=10, fuse_c=4, fuse_chain_dir=0), dict(fuse_T=12, fuse_c=3, fuse_chain_dir=1), dict(fuse_chain_dir=1),
                     dict(fuse_chain_dir=0)):
            qc.tune(fuse_T=11, fuse_c=4, fuse_chain_dir=-1)
            qc.tune(**tune)
            actions, recs, nrec = qc.fusion_plan(n, M, descs, mode | 4)
            total_chained += emu.check_chain_addressing(n, actions)
            state = ob.random_state(n, 3)
            want = state.copy()
            oracle_run(ob, want, n, M, Cn, steps)
            emu.run_plan(state, n, M, descs, actions, recs, ob)
            if mode == 1:
                assert np.array_equal(bits(state), bits(want)), (name, tune)
            else:
                assert float(np.max(np.abs(state - want))) <= 1e-12, (name, tune)
    assert total_chained >= 6, "the planner never chained anything"


def test_chain_of_the_n30_sweep_and_the_n28_iqft(qc, chain_guard):
    """what the chained plans look like at full size (addresses only: no state): n = 30 sweep = 3 passes, every one reads
    whole tiles (the first from the identity layout: its tile is the 12 low bits), stores runs of 2^3 amplitudes, and the last
    one stores the identity layout; n = 28 inverse QFT, tolerance mode: 3 radix-8 passes"""
    qc.tune(fuse_chain_min_n=20)
    n = 30
    actions, recs, nrec = qc.fusion_plan(n, 0, sweep_descs(n), 1 | 4)
    assert [a.fused for a in actions] == [1, 1, 1] and [a.chained for a in actions] == [1, 1, 1]
    for k, a in enumerate(actions):
        assert a.T == 12
        assert [int(x) for x in a.in_pos[:12]] == list(range(12)), "every pass of the chain reads contiguous tiles"
        run = 0
        while run < 12 and a.st_pos[run] == run:
            run += 1
        assert run == (3 if k < 2 else 3), (k, run)            # stores: 128-byte runs
    last = actions[-1]
    out_of = {int(last.tl[int(last.st_loc[j])]): int(last.st_pos[j]) for j in range(12)}
    assert all(q == p for q, p in out_of.items()), "the last pass of a chain stores the identity layout"
    n = 28
    actions, recs, nrec = qc.fusion_plan(n, 0, iqft_descs(qc, n, 0), 2 | 4)
    assert [a.fused for a in actions] == [1, 1, 1] and [a.chained for a in actions] == [1, 1, 1]
    assert [int(x) for x in actions[1].in_pos[:12]] == list(range(12)) and [int(x) for x in actions[2].in_pos[:12]] == list(range(12))


@pytest.mark.parametrize("mode", [1, 2], ids=["exact", "tolerance"])
def test_chained_tables_cover_every_tile_bit_at_full_size(qc, chain_guard, mode):
    """round 4's advisor finding: at n >= 30 a scattered Hadamard list can need more than 16 run-length segments for the tile
    number in the by-output order; the planner validated the by-input order and launched the other one with stale tables
    (18 free bits never deposited -> every tile at base 0).  Random H-heavy lists at n = 30 ... 36: every fused action's
    segment lists deposit all n - T bits and the layouts chain up to the identity (followed symbolically)."""
    rs = np.random.RandomState(77 + mode)
    total = 0
    small_ok = 0
    for trial in range(400):
        n = int(rs.randint(30, 37))
        cnt = int(rs.randint(20, 140))
        descs = []
        for _ in range(cnt):
            if mode == 2 and rs.rand() < 0.2:
                c, t = (int(x) for x in rs.choice(n, 2, replace=False))
                th = float(rs.uniform(-3, 3))
                descs.append((1, 0, (1 << c) | (1 << t), math.cos(th), math.sin(th), 0, 0))
            else:
                descs.append((0, int(rs.randint(0, n)), 0, 0.0, 0.0, 0, 0))
        for tune in (dict(), dict(fuse_chain_dir=0), dict(fuse_chain_dir=1)):
            qc.tune(fuse_chain_dir=-1)
            qc.tune(**tune)
            actions, recs, nrec = qc.fusion_plan(n, 0, descs, mode | 4)
            total += emu.check_chain_layouts(n, actions)
    assert total > 1000, total
    # the symbolic check agrees with the enumerating one where both run
    for n in (14, 16):
        descs = sweep_descs(n, 2)
        actions, recs, nrec = qc.fusion_plan(n, 0, descs, mode | 4)
        assert emu.check_chain_layouts(n, actions) == emu.check_chain_addressing(n, actions)
        small_ok += 1
    assert small_ok == 2


# ---- compact chains (round 4): the gate list of an inverse QFT on the VIRTUAL register [L register][orbit column] -------------
@pytest.mark.parametrize("mode", [1, 2], ids=["exact", "tolerance"])
@pytest.mark.parametrize("C,L,M,a", [(21, 9, 5, 2), (15, 10, 4, 7), (35, 8, 6, 2)])
def test_compact_register_premise(qc, ob, chain_guard, C, L, M, a, mode):
    """what compact_chain (csrc/qcx_fuse.inc.h) relies on, without a GPU: behind the circuit front the state lives on the
    orbit's residues; gathered into [L register][column] it is a register of L + cb qubits whose gate list is the inverse
    QFT's with every qubit shifted by M - cb, and running THAT through the planner's passes (chained, as on the GPU) and
    spreading the result back gives the bits the oracle gets on the whole register (1e-12 in the tolerance mode)."""
    n = L + M
    state = np.zeros(2 << n); ob.reset(state, n)
    for l in range(M, n):
        ob.hadamard(state, n, l)
    x = a % C
    for l in range(M, n):
        ob.camodc(state, n, M, C, x, l); x = (x * x) % C
    orbit = sorted({pow(a, e, C) for e in range(4 * C)})
    amp = state.reshape(-1, 2)
    low = np.arange(1 << n) & ((1 << M) - 1)
    assert not np.any(amp[~np.isin(low, orbit)]), "amplitudes off the orbit are exactly zero"
    cb = 2
    while (1 << cb) < len(orbit):
        cb += 1
    assert cb + 2 <= M
    nv = L + cb
    comp = np.zeros((1 << nv, 2))
    for j, f in enumerate(orbit):
        comp[(np.arange(1 << L) << cb) | j] = amp[(np.arange(1 << L) << M) | f]
    comp = comp.reshape(-1).copy()
    vdescs = iqft_descs(qc, nv, cb)                       # the inverse QFT of the virtual register: same angles, shifted qubits
    rdescs = iqft_descs(qc, n, M)
    assert [(d[0], d[3], d[4]) for d in vdescs] == [(d[0], d[3], d[4]) for d in rdescs]
    qc.tune(fuse_T=11, fuse_c=4)
    actions, recs, nrec = qc.fusion_plan(nv, cb, vdescs, mode | 4)
    emu.check_chain_addressing(nv, actions)
    emu.run_plan(comp, nv, cb, vdescs, actions, recs, ob)
    want = state.copy(); ob.iqft(want, n, M)
    got = np.zeros((1 << n, 2))
    cc = comp.reshape(-1, 2)
    for j, f in enumerate(orbit):
        got[(np.arange(1 << L) << M) | f] = cc[(np.arange(1 << L) << cb) | j]
    got = got.reshape(-1)
    if mode == 1:
        assert np.array_equal(bits(got), bits(want))
    else:
        assert float(np.max(np.abs(got - want))) <= 1e-12
    assert not np.any(cc[(np.arange(1 << nv) & ((1 << cb) - 1)) >= len(orbit)]), "unused columns stay zero"


# ---- round 5: the expanding store of a compact chain's last pass (FusePass::xp_*, k_fused_x8) -----------------------------------
@pytest.mark.parametrize("mode", [1, 2], ids=["exact", "tolerance"])
@pytest.mark.parametrize("C,L,M,a", [(21, 25, 5, 2), (21, 19, 5, 2), (15, 16, 4, 7), (255, 15, 8, 2), (35, 20, 6, 2)])
def test_expanding_store_tables_address_the_real_register(qc, chain_guard, C, L, M, a, mode):
    """what compact_chain hands the last pass of a chain so that it writes the REAL register itself, without a GPU: the planner's
    last action on the virtual register [L register][orbit column] (mode | 8), the host's tables for it (qcx_expand_store_plan),
    and the kernel's store loop restated here.  Every store index of a tile must land on the real index of the amplitude it
    carries -- (L part << M) | orbit[column] for the threads whose f lies on the orbit, a +0 everywhere else -- and the stores of
    a tile must cover its 2^(T - cb + M) real amplitudes exactly once."""
    import ctypes as Ct
    orbit = sorted({pow(a, e, C) for e in range(4 * C)})
    cb = max(2, (len(orbit) - 1).bit_length())
    nv = L + cb
    acts, recs, _ = qc.fusion_plan(nv, cb, iqft_descs(qc, nv, cb), mode | 4 | 8)
    last = acts[-1]
    st_pos = (Ct.c_ubyte * 16)(*last.st_pos); st_loc = (Ct.c_ubyte * 16)(*last.st_loc)
    xp_pos, xp_loc, xp_col = (Ct.c_ubyte * 24)(), (Ct.c_ubyte * 24)(), (Ct.c_ubyte * 4)()
    ok = Ct.c_int(-1)
    assert qc.lib().qcx_expand_store_plan(last.T, st_pos, st_loc, M, cb, xp_pos, xp_loc, xp_col, Ct.byref(ok)) == 0
    kinds = {recs[last.rec_off + k].type & 0xFF for k in range(last.nops)} if last.fused else set()
    if not (last.fused and last.T == 12 and list(last.st_pos[:cb]) == list(range(cb))):
        assert ok.value == 0 or not last.fused
        pytest.skip("the plan's last action does not qualify for the expanding store")
    assert ok.value == 1 and kinds & {8, 9}
    T, LB = 12, 9                                           # 2^12 tile, 512 threads
    tl = list(last.tl[:T])                                  # tile-local bit j carries virtual qubit tl[j]
    ebits = T - cb + M
    e = np.arange(1 << ebits, dtype=np.int64)
    thread, it = e & 511, e >> LB
    f = thread & ((1 << M) - 1)
    off = f.copy(); loc = np.zeros_like(e)
    for i in range(M, LB):                                  # the thread's bits above f
        on = (thread >> i) & 1
        off |= on << xp_pos[i]; loc |= on << xp_loc[i]
    for b in range(ebits - LB):                             # the iteration's bits
        on = (it >> b) & 1
        off |= on << xp_pos[LB + b]; loc |= on << xp_loc[LB + b]
    col_of = {v: j for j, v in enumerate(orbit)}
    col = np.array([col_of.get(int(v), -1) for v in range(1 << M)])[f]
    live = col >= 0
    for b in range(cb):
        loc |= np.where(live, ((col >> b) & 1) << xp_col[b], 0)
    # (1) the stores of a tile cover its real amplitudes exactly once: the offsets are 2^ebits distinct values made of the M low
    #     bits and the real positions of the tile's L-register qubits
    real_bits = sorted([q - cb + M for q in tl if q >= cb])
    span = set(range(M)) | set(real_bits)
    assert len(np.unique(off)) == e.size and all(int(o) & ~sum(1 << p for p in span) == 0 for o in off[:: 97])
    # (2) a live store carries the tile element whose qubits spell the same L part, in its column
    want = np.zeros_like(e)
    for j in range(T):
        q = tl[j]
        bitj = (loc >> j) & 1
        if q >= cb:
            want |= bitj << (q - cb + M)
    want_live = want | np.array(orbit + [0] * 16)[np.clip(col, 0, None)]
    assert np.array_equal(off[live], want_live[live])
    colbits = np.zeros_like(e)
    for j in range(T):
        if tl[j] < cb:
            colbits |= ((loc >> j) & 1) << tl[j]
    assert np.array_equal(colbits[live], col[live])
    # (3) and every element of the compact tile that holds an orbit column leaves exactly once
    assert len(np.unique(loc[live])) == int(live.sum()) == len(orbit) << (T - cb)
