"""CPU emulator of the fused-pass kernels' record interpreter (TEST INFRASTRUCTURE, numpy).

`qcx_fusion_plan` (include/qcx.h) returns what the planner in csrc/qcx_fuse.inc.h would hand to the GPU: actions (fused
passes / stand-alone gates) and the 32-byte records of every pass.  This module applies those records to a small state
vector with the arithmetic the kernels in csrc/qcx_kernels.h use -- same products and sums in the same order, the
"+ 0.0" canonicalisation where the kernels put it (per gate in the plain form; once per round / run in the ROUNDS
form) -- so that the CPU-only test suite can check planner + record format + kernel semantics against the oracle
without a GPU.  Everything is expressed on LOGICAL amplitude indices: tile-local bit j is qubit act.tl[j] (in place on the
identity layout: j for j < c, hbit[j - c] above).  Where a chained pass reads and writes its tiles in memory is checked
separately (check_chain_addressing): it never changes what the records mean.
"""
import struct

import numpy as np

FUSE_H, FUSE_PHASE, FUSE_CAMODC, FUSE_ROUND, FUSE_PRUN, FUSE_CAMRUN, FUSE_DIAG, FUSE_QROUND, FUSE_QROUND3, FUSE_ROUND8 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
SQRT1_2 = 0.70710678118654752440


class Pass:
    def __init__(self, n, act):
        self.n = n
        self.c = act.c
        self.hbit = [int(act.hbit[j]) for j in range(act.nh)]
        self.T = act.T
        assert self.T == self.c + len(self.hbit)
        # the qubit that is tile-local bit j: act.tl (a chained pass orders its tile by the layout it reads; in place on the
        # identity layout this is the low c bits followed by the hot bits)
        self.tl = [int(act.tl[j]) for j in range(self.T)]
        assert sorted(self.tl) == sorted(list(range(self.c)) + self.hbit)
        if not getattr(act, "chained", 0):
            assert self.tl == list(range(self.c)) + self.hbit
        self.idx = np.arange(1 << n, dtype=np.uint64)
        tile_mask = 0
        for b in self.tl:
            tile_mask |= 1 << b
        self.base = self.idx & np.uint64(~tile_mask & ((1 << 64) - 1))

    def gbit(self, j):
        return self.tl[j]

    def bit(self, j_local):
        return ((self.idx >> np.uint64(self.gbit(j_local))) & np.uint64(1)).astype(bool)

    def local_all_set(self, mloc):
        ok = np.ones(self.idx.shape, dtype=bool)
        j = 0
        while mloc >> j:
            if (mloc >> j) & 1:
                assert j < self.T, "tile-local mask bit outside the tile"
                ok &= self.bit(j)
            j += 1
        return ok

    def outside_all_set(self, mext):
        m = np.uint64(mext)
        return (self.base & m) == m


def _h(re, im, gb, canon):
    """H on global bit gb: the kernels' butterfly (t0 +/- t1, each product s * x rounded on its own)"""
    n_amp = re.shape[0]
    i0 = np.arange(n_amp, dtype=np.int64)
    i0 = i0[(i0 >> gb) & 1 == 0]
    i1 = i0 | (1 << gb)
    t0r, t0i, t1r, t1i = SQRT1_2 * re[i0], SQRT1_2 * im[i0], SQRT1_2 * re[i1], SQRT1_2 * im[i1]
    ar, ai, br, bi = t0r + t1r, t0i + t1i, t0r - t1r, t0i - t1i
    if canon:
        ar, ai, br, bi = ar + 0.0, ai + 0.0, br + 0.0, bi + 0.0
    re[i0], im[i0], re[i1], im[i1] = ar, ai, br, bi


def _rotate(re, im, sel, cc, ss, canon):
    x, y = re[sel], im[sel]
    nx = (cc * x) - (ss * y)
    ny = (cc * y) + (ss * x)
    if canon:
        nx, ny = nx + 0.0, ny + 0.0
    re[sel], im[sel] = nx, ny


def _cam_extra(rec):
    return struct.unpack("<IIII", struct.pack("<dd", rec.c, rec.s))       # C, d, Cd, inv


def _camodc_step(P, re, im, rec):
    """fuse_camodc_step: closed form of the controlled modular multiply on 2^M blocks inside the tile"""
    M = rec.a & 0xFF
    ctl_local = ((rec.a >> 8) & 0xFF) - 1
    Cn, d, Cd, inv = _cam_extra(rec)
    on = P.outside_all_set(rec.mask)
    if ctl_local >= 0:
        on = on & P.bit(ctl_local)
    idx = P.idx.astype(np.int64)
    f = idx & ((1 << M) - 1)
    act = on & (f < Cn)
    nr, ni = re.copy(), im.copy()
    blk0 = idx - f
    if d == 1:
        src = blk0 + (f * inv) % Cn
        sr, si = np.zeros_like(re), np.zeros_like(im)
        sr[act] += re[src[act]]; si[act] += im[src[act]]
    else:
        sr, si = np.zeros_like(re), np.zeros_like(im)
        live = act & (f % d == 0)
        s0 = ((f // d) * inv) % Cd
        for q in range(d):
            src = blk0 + s0 + q * Cd
            sr[live] += re[src[live]]; si[live] += im[src[live]]
    nr[act], ni[act] = sr[act], si[act]
    re[:], im[:] = nr, ni


def _camrun(P, re, im, recs, i, blob):
    """FUSE_CAMRUN: a run of permutation-type multiplies folded into one gather through per-gate byte tables"""
    hdr = recs[i]
    cnt, cpad = hdr.a & 0xFFFF, hdr.a >> 16
    tabs = int(hdr.mask)
    M = recs[i + 1].a & 0xFF
    Cn = _cam_extra(recs[i + 1])[0]
    x = np.ones(P.idx.shape, dtype=np.int64)
    for g in range(cnt):
        r = recs[i + 1 + g]
        assert (r.type & 0xFF) == FUSE_CAMODC and _cam_extra(r)[1] == 1 and _cam_extra(r)[0] == Cn
        cond = P.outside_all_set(r.mask)
        cl = ((r.a >> 8) & 0xFF) - 1
        if cl >= 0:
            cond = cond & P.bit(cl)
        tb = np.frombuffer(blob[tabs + g * cpad: tabs + (g + 1) * cpad], dtype=np.uint8).astype(np.int64)
        x = np.where(cond, tb[x], x)
    idx = P.idx.astype(np.int64)
    f = idx & ((1 << M) - 1)
    wr = (x != 1) & (f < Cn)
    src = (idx - f) + (x * f) % Cn
    nr, ni = re.copy(), im.copy()
    nr[wr] = 0.0 + re[src[wr]]
    ni[wr] = 0.0 + im[src[wr]]
    re[:], im[:] = nr, ni
    return cnt


def _diag(P, re, im, rec, area, nd):
    """FUSE_DIAG (tolerance mode, K6t): amplitudes with the control bit set are multiplied by
    E_out(tile) * G0[local bits 0-3] * G1[4-7] * G2[8-11]; the tables come from the pass's table area (complex128 units)"""
    a = rec.a
    cl, slot, groups = (a & 0xFF) - 1, (a >> 8) & 0xFF, (a >> 16) & 7
    assert slot < nd
    sel = P.outside_all_set(rec.mask)
    if cl >= 0:
        sel = sel & P.bit(cl)
    tloc = struct.unpack("<Q", struct.pack("<d", rec.c))[0]
    # tile-local element index of every amplitude
    e = np.zeros(P.idx.shape, dtype=np.int64)
    for j in range(P.T):
        e |= P.bit(j).astype(np.int64) << j
    assert tloc < (1 << P.T) and all(((tloc >> (4 * g)) & 15) == 0 for g in range(3) if not (groups >> g) & 1)
    G = area[3 * nd + 48 * slot: 3 * nd + 48 * (slot + 1)]
    info = struct.unpack("<8I", area[3 * slot: 3 * slot + 2].tobytes())
    field_off, present = info[:5], info[5]
    F = np.full(P.idx.shape, area[3 * slot + 2], dtype=np.complex128)        # the diagonal's constant factor (kc, ks)
    base = P.base.astype(np.int64)
    for f in range(5):
        if (present >> f) & 1:
            F = F * area[field_off[f] + ((base >> (8 * f)) & 255)]
    for g in range(3):
        if (groups >> g) & 1:
            F = F * G[16 * g + ((e >> (4 * g)) & 15)]
        else:
            assert np.all(G[16 * g: 16 * g + 16] == 1.0)
    z = (re + 1j * im)
    z[sel] = z[sel] * F[sel]
    re[:], im[:] = z.real, z.imag


def apply_pass(state, n, act, recs):
    """one fused pass; state = interleaved (re, im) float64 array of 2 * 2^n, modified in place"""
    re, im = state[0::2].copy(), state[1::2].copy()
    P = Pass(n, act)
    r0 = act.rec_off
    R = [recs[r0 + k] for k in range(act.rec_cnt)]
    blob = b""
    if act.table_bytes:
        raw = b"".join(bytes(memoryview(recs[r0 + act.table_rec_off + k]).cast("B")) for k in range(act.rec_cnt - act.table_rec_off))
        blob = raw[:act.table_bytes]
    nops = act.nops
    nd = getattr(act, "diag_cnt", 0)
    area = None
    if nd:
        raw = b"".join(bytes(memoryview(recs[r0 + act.diag_rec_off + k]).cast("B")) for k in range(act.rec_cnt - act.diag_rec_off))
        area = np.frombuffer(raw, dtype=np.complex128)
    i = 0
    stats = dict(rounds=0, runs=0, run_gates=0, h=0, diags=0)
    if not act.rounds_form:
        while i < nops:
            r = R[i]
            t = r.type & 0xFF
            if t == FUSE_H:
                _h(re, im, P.gbit(r.a), True); stats["h"] += 1
            elif t == FUSE_PHASE:
                _rotate(re, im, P.outside_all_set(r.mask) & P.local_all_set(r.a), r.c, r.s, True)
            elif t == FUSE_CAMODC:
                _camodc_step(P, re, im, r)
            else:
                raise AssertionError(f"record type {t} in a plain gate list")
            i += 1
    else:
        while i < nops:
            r = R[i]
            t = r.type & 0xFF
            if t == FUSE_ROUND:
                rb0, rb1, has_h, cnt = r.a & 0xFF, (r.a >> 8) & 0xFF, (r.a >> 16) & 1, int(r.mask)
                assert rb0 < rb1 < P.T and cnt >= 1
                qreg = P.bit(rb0).astype(np.int64) | (P.bit(rb1).astype(np.int64) << 1)
                stats["rounds"] += 1
                saw_h = False
                o, oend = i + 1, i + cnt
                while o <= oend:
                    it = R[o].type
                    kind, code, rc = it & 0xFF, (it >> 8) & 0xFF, it >> 16
                    if kind == FUSE_H:
                        assert (code & 32) and rc == 0
                        _h(re, im, P.gbit(rb1 if code & 1 else rb0), False)
                        saw_h = True; stats["h"] += 1
                        o += 1
                    elif kind == FUSE_PRUN:
                        rsel, canon = code & 15, bool(code & 16)
                        assert 1 <= rc <= 64 and rsel and not (code & 32)
                        assert nd or canon == (not has_h), "a run canonicalises its own zeros exactly when its round has no H"
                        touched = np.zeros(P.idx.shape, dtype=bool)
                        for g in range(rc):
                            gr = R[o + 1 + g]
                            assert (gr.type & 0xFF) == FUSE_PHASE and ((gr.type >> 8) & 15) == rsel
                            assert not (gr.a & ((1 << rb0) | (1 << rb1))), "register bits stay out of the lane mask"
                            sel = P.outside_all_set(gr.mask) & P.local_all_set(gr.a) & (((rsel >> qreg) & 1) == 1)
                            _rotate(re, im, sel, gr.c, gr.s, False)
                            touched |= sel
                        if canon:
                            # on ALL amplitudes held in the run's registers (round 4: the state is canonical whenever gates
                            # run, so "+ 0.0" changes nothing where no gate of the run rotated)
                            whole = ((rsel >> qreg) & 1) == 1
                            re[whole] += 0.0; im[whole] += 0.0
                        stats["runs"] += 1; stats["run_gates"] += rc
                        o += 1 + rc
                    elif kind == FUSE_DIAG and nd:
                        _diag(P, re, im, R[o], area, nd); stats["diags"] += 1
                        o += 1
                    else:
                        raise AssertionError(f"item kind {kind} inside a round")
                assert o == oend + 1 and saw_h == bool(has_h)
                if has_h:
                    re += 0.0; im += 0.0
                i += 1 + cnt
            elif t == FUSE_ROUND8:
                # the exact walk on 8 amplitudes per thread (k_fused_x8): three register bits, runs name their registers by pattern
                rb = [r.a & 0xFF, (r.a >> 8) & 0xFF, (r.a >> 16) & 0xFF]
                has_h, cnt = (r.a >> 24) & 1, int(r.mask)
                assert rb[0] < rb[1] < rb[2] < P.T and cnt >= 1 and not nd
                # the thread-bit map: the T - 3 non-register tile bits, each exactly once, lane bits first
                tmap = struct.unpack("<Q", struct.pack("<d", r.c))[0]
                tb = [(tmap >> (4 * k)) & 15 for k in range(P.T - 3)]
                assert sorted(tb + rb) == list(range(P.T)), ("thread map", tb, rb)
                # bit 25: no barrier in front of this round -- legal only when the wave number (thread bits 6 and up) rides on the same
                # tile bits as in the previous round: every wave then reads what it wrote itself and nothing else
                wave_bits = tuple(tb[6:])
                if (r.a >> 25) & 1:
                    assert i > 0 and getattr(P, "x8_wave_bits", None) == wave_bits, "a round without a barrier must keep the waves on their tile bits"
                    stats["nobarrier"] = stats.get("nobarrier", 0) + 1
                P.x8_wave_bits = wave_bits
                stats.setdefault("maps", []).append((tuple(rb), tuple(tb)))
                qreg = P.bit(rb[0]).astype(np.int64) | (P.bit(rb[1]).astype(np.int64) << 1) | (P.bit(rb[2]).astype(np.int64) << 2)
                pat_bits = {0: 0, 1: 1, 2: 2, 3: 4, 4: 3, 5: 5, 6: 6}        # pattern -> the register bits a gate's mask held
                stats["rounds"] += 1
                saw_h = False
                o, oend = i + 1, i + cnt
                while o <= oend:
                    it = R[o].type
                    kind, code, rc = it & 0xFF, (it >> 8) & 0xFF, it >> 16
                    if kind == FUSE_H:
                        assert (code & 32) and rc == 0 and (code & 3) < 3
                        _h(re, im, P.gbit(rb[code & 3]), False)
                        saw_h = True; stats["h"] += 1
                        o += 1
                    elif kind == FUSE_PRUN:
                        pat, canon = code & 7, bool(code & 16)
                        assert 1 <= rc <= 63 and pat in pat_bits and not (code & 32)
                        assert canon == (not has_h), "a run canonicalises its own zeros exactly when its round has no H"
                        need = pat_bits[pat]
                        held = (qreg & need) == need                     # the registers of the pattern
                        for g in range(rc):
                            gr = R[o + 1 + g]
                            assert (gr.type & 0xFF) == FUSE_PHASE and ((gr.type >> 8) & 7) == pat
                            assert not (gr.a & ((1 << rb[0]) | (1 << rb[1]) | (1 << rb[2]))), "register bits stay out of the lane mask"
                            # bits 48 .. of the outside mask: conditions on the tile bits the wave number rides on (thread bits 6 ..)
                            wloc = 0
                            for w in range(len(wave_bits)):
                                if (int(gr.mask) >> (48 + w)) & 1:
                                    wloc |= 1 << wave_bits[w]
                            assert not (int(gr.mask) >> (48 + len(wave_bits))) and not (gr.a & sum(1 << b for b in wave_bits)), "wave-bit conditions live in the outside mask"
                            ext = int(gr.mask) & ((1 << 48) - 1)
                            _rotate(re, im, P.outside_all_set(ext) & P.local_all_set(gr.a | wloc) & held, gr.c, gr.s, False)
                        if canon:
                            re[held] += 0.0; im[held] += 0.0
                        stats["runs"] += 1; stats["run_gates"] += rc
                        o += 1 + rc
                    else:
                        raise AssertionError(f"item kind {kind} inside a radix-8 exact round")
                assert o == oend + 1 and saw_h == bool(has_h)
                if has_h:
                    re += 0.0; im += 0.0
                i += 1 + cnt
            elif t == FUSE_QROUND and nd:
                # tolerance mode fast round: H(x) [D(x)] [H(y) [D(y)]] on the two register bits, two step words in the next record
                rb0, rb1, ns = r.a & 0xFF, (r.a >> 8) & 0xFF, (r.a >> 16) & 0xFF
                assert rb0 < rb1 < P.T and int(r.mask) == 1 and 1 <= ns <= 2
                steps = [R[i + 1].type, R[i + 1].a]
                assert (steps[1] == 0xFFFFFFFF) == (ns == 1)
                stats["rounds"] += 1
                for sw in steps[:ns]:
                    hb, ob = (rb1, rb0) if sw & 1 else (rb0, rb1)
                    _h(re, im, P.gbit(hb), False); stats["h"] += 1
                    if sw & 2:
                        slot, groups, t_other = (sw >> 8) & 0xFF, (sw >> 16) & 7, (sw >> 19) & 1
                        # the same diagonal through the generic record: control = the H's bit, local target mask from the tables
                        G = area[3 * nd + 48 * slot: 3 * nd + 48 * (slot + 1)]
                        tloc = 0
                        for lb in range(12):
                            if G[16 * (lb >> 2) + (1 << (lb & 3))] != 1.0:
                                tloc |= 1 << lb
                        assert ((tloc >> ob) & 1) <= t_other, "a target on the other register bit must be flagged"
                        if not t_other:
                            assert G[16 * (ob >> 2) + (1 << (ob & 3))] == 1.0
                        class Rec: pass
                        rec = Rec(); rec.a = (hb + 1) | (slot << 8) | (groups << 16); rec.mask = 0
                        rec.c = struct.unpack("<d", struct.pack("<Q", tloc))[0]
                        _diag(P, re, im, rec, area, nd); stats["diags"] += 1
                i += 2
            elif t == FUSE_QROUND3:
                # tolerance mode, radix-8 fast round (k_fused_q3): up to three steps H(x) [D(x)] on three register bits
                rb = [r.a & 0xFF, (r.a >> 8) & 0xFF, (r.a >> 16) & 0xFF]
                ns = (r.a >> 24) & 3
                assert rb[0] < rb[1] < rb[2] < P.T and int(r.mask) == 1 and 1 <= ns <= 3
                # round 5: the header also carries k_fused_x8's thread map (c) and "no barrier in front of this round" (bit 28 of a)
                tmap = struct.unpack("<Q", struct.pack("<d", r.c))[0]
                tb = [(tmap >> (4 * k)) & 15 for k in range(P.T - 3)]
                if P.T == 12:
                    assert sorted(tb + rb) == list(range(P.T)), ("thread map", tb, rb)
                    wave_bits = tuple(tb[6:])
                    if (r.a >> 28) & 1:
                        assert i > 0 and getattr(P, "x8_wave_bits", None) == wave_bits, "a round without a barrier must keep the waves on their tile bits"
                        stats["nobarrier"] = stats.get("nobarrier", 0) + 1
                    P.x8_wave_bits = wave_bits
                    stats.setdefault("maps", []).append((tuple(rb), tuple(tb)))
                steps = [R[i + 1].type, R[i + 1].a, int(R[i + 1].mask) & 0xFFFFFFFF]
                assert sorted(sw & 3 for sw in steps) == [0, 1, 2], "the three step words name the three register bits"
                stats["rounds"] += 1
                for sw in steps[:ns]:
                    Rr = sw & 3
                    hb = rb[Rr]
                    others = [rb[k] for k in range(3) if k != Rr]
                    _h(re, im, P.gbit(hb), False); stats["h"] += 1
                    if sw & 4:
                        assert nd, "a diagonal in a pass without tables"
                        slot, groups = (sw >> 8) & 0xFF, (sw >> 16) & 7
                        G = area[3 * nd + 48 * slot: 3 * nd + 48 * (slot + 1)]
                        tloc = 0
                        for lb in range(12):
                            if G[16 * (lb >> 2) + (1 << (lb & 3))] != 1.0:
                                tloc |= 1 << lb
                        for k, ob in enumerate(others):
                            flagged = (sw >> (19 + k)) & 1
                            assert ((tloc >> ob) & 1) <= flagged, "a target on another register bit must be flagged"
                            if not flagged:
                                assert G[16 * (ob >> 2) + (1 << (ob & 3))] == 1.0
                        class Rec: pass
                        rec = Rec(); rec.a = (hb + 1) | (slot << 8) | (groups << 16); rec.mask = 0
                        rec.c = struct.unpack("<d", struct.pack("<Q", tloc))[0]
                        _diag(P, re, im, rec, area, nd); stats["diags"] += 1
                if not nd:
                    re += 0.0; im += 0.0         # the exact form (Hadamards only): canonical zeros (the kernel: once per pass, same values)
                i += 2
            elif t == FUSE_CAMRUN:
                i += 1 + _camrun(P, re, im, R, i, blob)
            elif t == FUSE_CAMODC:
                _camodc_step(P, re, im, r)
                i += 1
            else:
                raise AssertionError(f"record type {t} between rounds")
    state[0::2], state[1::2] = re, im
    return stats


def run_plan(state, n, M, descs, actions, recs, ob):
    """apply a whole plan: fused passes through the emulator, stand-alone gates through the oracle"""
    covered = 0
    totals = dict(passes=0, standalone=0, rounds=0, runs=0, run_gates=0, h=0, diags=0)
    for act in actions:
        assert act.first_gate == covered, "actions cover the gate list in order without gaps"
        covered += act.ngates
        if act.fused:
            st = apply_pass(state, n, act, recs)
            totals["passes"] += 1
            for k, v in st.items():
                if k == "maps":
                    totals.setdefault("maps", []).extend(v)
                elif k == "nobarrier":
                    totals["nobarrier"] = totals.get("nobarrier", 0) + v
                else:
                    totals[k] += v
        else:
            typ, q, mask, c, s, Cn, A = descs[act.first_gate]
            totals["standalone"] += 1
            if typ == 0:
                ob.hadamard(state, n, q)
            elif typ == 1:
                bits = [b for b in range(n) if (mask >> b) & 1]
                assert len(bits) == 2
                _rotate_oracle_pair(state, n, bits, c, s)
            else:
                ob.camodc(state, n, M, Cn, A, q)
    assert covered == len(descs)
    return totals


def _rotate_oracle_pair(state, n, bits, c, s):
    re, im = state[0::2].copy(), state[1::2].copy()
    idx = np.arange(1 << n, dtype=np.int64)
    sel = (((idx >> bits[0]) & 1) == 1) & (((idx >> bits[1]) & 1) == 1)
    _rotate(re, im, sel, c, s, True)
    state[0::2], state[1::2] = re, im


def _deposit(t, segs, nseg):
    x = 0
    for k in range(nseg):
        src, dst, ln = segs[4 * k], segs[4 * k + 1], segs[4 * k + 2]
        x |= ((t >> src) & ((1 << ln) - 1)) << dst
    return x


def _spread(e, pos, T):
    off = 0
    for j in range(T):
        off |= ((e >> j) & 1) << int(pos[j])
    return off


def check_chain_addressing(n, actions):
    """Follow the ADDRESSES of a plan's fused passes exactly as the kernels compute them (FusePass tables): a buffer that
    holds, at every physical index, the LOGICAL index of the amplitude stored there.  Every pass must (a) find in each tile
    exactly the amplitudes of one logical tile, element e at the local index the records assume (act.tl), (b) see the logical
    base index it tests controls against, (c) store a permutation; and after the last pass of every chain the layout must be
    the identity again.  Returns the number of chained passes."""
    cur = np.arange(1 << n, dtype=np.int64)          # identity layout
    chained = 0
    for act in actions:
        if not act.fused:
            assert np.array_equal(cur, np.arange(1 << n)), "a stand-alone gate needs the identity layout"
            continue
        T = act.T
        tl = [int(act.tl[j]) for j in range(T)]
        e = np.arange(1 << T, dtype=np.int64)
        in_off = np.zeros(1 << T, dtype=np.int64)
        lg_off = np.zeros(1 << T, dtype=np.int64)
        for j in range(T):
            in_off |= ((e >> j) & 1) << int(act.in_pos[j])
            lg_off |= ((e >> j) & 1) << tl[j]
        # store order: position j of the store index eo belongs to local bit st_loc[j], lands at output bit st_pos[j]
        ld = np.zeros(1 << T, dtype=np.int64)
        st = np.zeros(1 << T, dtype=np.int64)
        for j in range(T):
            ld |= ((e >> j) & 1) << int(act.st_loc[j])
            st |= ((e >> j) & 1) << int(act.st_pos[j])
        assert sorted(int(x) for x in act.st_loc[:T]) == list(range(T))
        out = np.full(1 << n, -1, dtype=np.int64) if act.chained else None
        ntiles = 1 << (n - T)
        for t in range(ntiles):
            b_in = _deposit(t, bytes(act.seg_in), act.nseg_in)
            b_lg, b_out = b_in, b_in
            if act.chained:
                b_lg = _deposit(t, bytes(act.seg_lg), act.nseg_lg)
                b_out = _deposit(t, bytes(act.seg_out), act.nseg_out)
            tile = cur[b_in | in_off]                 # what the fill brings to local index e
            assert np.array_equal(tile, b_lg | lg_off), f"tile {t}: the fill does not deliver the logical tile the records assume"
            if act.chained:
                assert np.all(out[b_out | st] == -1)
                out[b_out | st] = tile[ld]
            else:
                assert act.nseg_out == 0 and np.array_equal(np.asarray(act.st_pos[:T]), np.asarray(act.in_pos[:T]))
        if act.chained:
            assert np.all(out >= 0), "the stores of a chained pass cover the buffer exactly once"
            cur = out
            chained += 1
    assert np.array_equal(cur, np.arange(1 << n)), "a plan must leave the identity layout behind"
    return chained


def _seg_map(segs, nseg):
    """tile-number bit k -> index bit it is deposited at, from a run-length segment list"""
    m = {}
    for k in range(nseg):
        src, dst, ln = segs[4 * k], segs[4 * k + 1], segs[4 * k + 2]
        for j in range(ln):
            assert src + j not in m, "a tile-number bit is deposited twice"
            m[src + j] = dst + j
    return m


def check_chain_layouts(n, actions):
    """check_chain_addressing for registers too large to enumerate: the same address arithmetic followed SYMBOLICALLY -- the
    layout is a permutation qubit -> physical index bit, every table of a pass is one bit to one bit.  Every fused pass must
    deposit all n - T tile-number bits (input, and for a chained pass output and logical base), find its tile's qubits where
    the layout has them, store a bijection, and every chain must end on the identity.  Returns the number of chained passes."""
    lay = list(range(n))                              # lay[q] = physical bit of qubit q in the current buffer
    chained = 0
    for act in actions:
        if not act.fused:
            assert lay == list(range(n)), "a stand-alone gate needs the identity layout"
            continue
        T = act.T
        tl = [int(act.tl[j]) for j in range(T)]
        m_in = _seg_map(bytes(act.seg_in), act.nseg_in)
        assert sorted(m_in) == list(range(n - T)), ("seg_in covers", len(m_in), "of", n - T, "tile-number bits")
        for j in range(T):
            assert int(act.in_pos[j]) == lay[tl[j]], "tile-local bit j is not where the input layout has its qubit"
        if not act.chained:
            free = sorted(set(range(n)) - {lay[q] for q in tl})
            assert sorted(m_in.values()) == free
            assert lay == list(range(n)), "an in-place pass works on the identity layout"
            continue
        m_out = _seg_map(bytes(act.seg_out), act.nseg_out)
        m_lg = _seg_map(bytes(act.seg_lg), act.nseg_lg)
        assert sorted(m_out) == list(range(n - T)), ("seg_out covers", len(m_out), "of", n - T)
        assert sorted(m_lg) == list(range(n - T)), ("seg_lg covers", len(m_lg), "of", n - T)
        new = [None] * n
        for k in range(n - T):
            q = m_lg[k]                               # the qubit tile-number bit k stands for
            assert q not in tl and lay[q] == m_in[k], "input and logical deposits of a tile-number bit disagree"
            new[q] = m_out[k]
        assert sorted(int(x) for x in act.st_loc[:T]) == list(range(T))
        for j in range(T):
            new[tl[int(act.st_loc[j])]] = int(act.st_pos[j])
        assert sorted(new) == list(range(n)), "the stores of a chained pass are not a permutation of the index bits"
        lay = new
        chained += 1
    assert lay == list(range(n)), "a plan must leave the identity layout behind"
    return chained


def x8_swz(e):
    """LDS slot of tile-local element e in k_fused_x8 (csrc/qcx_kernels.h): the two upper nibbles folded onto the lowest"""
    return e ^ ((e >> 4) & 15) ^ ((e >> 8) & 15)


def x8_lane_conflicts(tb):
    """worst multiplicity of a 16-byte bank group among the lane groups the LDS serves together, for a round whose lane bit k rides
    on tile bit tb[k]: (reads, writes); (1, 1) = conflict-free.  Groups as in the MI355X guide's LDS table."""
    slot = []
    for l in range(64):
        e = 0
        for k in range(6):
            e |= ((l >> k) & 1) << tb[k]
        slot.append(x8_swz(e))
    rg = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    worst_r = max(max(np.bincount([slot[32 * h + l] & 15 for l in g], minlength=16)) for h in range(2) for g in rg)
    worst_w = max(max(np.bincount([slot[8 * g + k] & 7 for k in range(8)], minlength=8)) for g in range(8))
    return int(worst_r), int(worst_w)
