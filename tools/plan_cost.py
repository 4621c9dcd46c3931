"""Instruction-count model of a bit-exact plan (host only; no GPU): for every pass and round of the rounds form, how many
FP64 wave-instructions the item walk issues against the count a machine with every lane useful would need.

  python tools/plan_cost.py [n] [M]      (the IQFT of an n-qubit register whose low M qubits are untouched; default 28 0)

Per gate of a run the walk issues 6 * popcount(rsel) products and sums (4 v_mul_f64 + 2 v_add_f64 per rotated register)
on every wave that has at least one live lane, plus 2 mask instructions when the gate has lane-bit members.  A wave is
skipped when a wave-uniform member bit (wave bits of the round, bits outside the tile) is 0.  Ideal: rotations * 6 / 64.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import quantumcomputer_amd._lib as qc            # noqa: E402
import fuse_emulator as emu                      # noqa: E402
from test_fusion_plan import iqft_descs          # noqa: E402


def popcount(x):
    return bin(x).count("1")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    descs = iqft_descs(qc, n, M)
    acts, recs, _ = qc.fusion_plan(n, M, descs, mode=mode)
    tot_issue = tot_ideal = tot_mask = tot_h = 0.0
    for pi, a in enumerate(acts):
        if not a.fused or not a.rounds_form:
            print(f"pass {pi}: not a rounds-form pass (fused={a.fused})")
            continue
        T = a.T
        R = [recs[a.rec_off + k] for k in range(a.nops)]
        ntiles = 1 << (n - T)
        i = 0
        p_issue = p_ideal = p_mask = p_h = 0.0
        cls = {}
        nrounds = 0
        while i < a.nops:
            r = R[i]
            t = r.type & 0xFF
            assert t in (emu.FUSE_ROUND, emu.FUSE_ROUND8), t
            x8 = t == emu.FUSE_ROUND8
            rb0, rb1, cnt = r.a & 0xFF, (r.a >> 8) & 0xFF, int(r.mask)
            nrounds += 1
            if x8:              # the exact walk on 8 amplitudes per thread: three register bits, the thread map in the header
                import struct
                tmap = struct.unpack("<Q", struct.pack("<d", r.c))[0]
                rest = [(tmap >> (4 * k)) & 15 for k in range(T - 3)]
                waves_per_tile = (1 << T) // 512
            else:
                rest = [j for j in range(T) if j not in (rb0, rb1)]
                waves_per_tile = (1 << T) // 256
            lane_bits, wave_bits = rest[:6], rest[6:]
            lane_m = sum(1 << j for j in lane_bits)
            wave_m = sum(1 << j for j in wave_bits)
            o, oend = i + 1, i + cnt
            while o <= oend:
                it = R[o].type
                kind, code, rc = it & 0xFF, (it >> 8) & 0xFF, it >> 16
                if kind == emu.FUSE_H:
                    p_h += (32 if x8 else 16) * waves_per_tile * ntiles           # 4 products + 4 sums per pair of registers
                    o += 1
                    continue
                assert kind == emu.FUSE_PRUN
                rsel = code & 15
                nreg_run = {0: 8, 1: 4, 2: 4, 3: 4, 4: 2, 5: 2, 6: 2}[code & 7] if x8 else popcount(rsel)
                for g in range(rc):
                    gr = R[o + 1 + g]
                    nl = popcount(gr.a & lane_m)
                    nw = popcount(gr.a & wave_m)
                    no = popcount(int(gr.mask))
                    nreg = nreg_run
                    waves = waves_per_tile * ntiles / (1 << (nw + no))        # waves that issue the gate
                    p_issue += waves * 6 * nreg
                    p_mask += waves * 2 * (1 if nl else 0)
                    p_ideal += waves * 6 * nreg / (1 << nl)
                    key = (nreg, nl)
                    cls[key] = cls.get(key, 0) + 1
                o += 1 + rc
            i += 1 + cnt
        hot = [a.hbit[j] for j in range(a.nh)]
        print(f"pass {pi}: T={T} hot={hot} rounds={nrounds} gates={a.ngates}  FP64 issued {p_issue:.3e}  ideal {p_ideal:.3e}  "
              f"(x{p_issue / max(p_ideal, 1):.2f})  mask {p_mask:.3e}  H {p_h:.3e}")
        print("         gates by (registers rotated, lane-bit members): " + ", ".join(f"{k}: {v}" for k, v in sorted(cls.items())))
        tot_issue += p_issue; tot_ideal += p_ideal; tot_mask += p_mask; tot_h += p_h
    print(f"total: rotations issued {tot_issue:.3e}, ideal {tot_ideal:.3e} (x{tot_issue / tot_ideal:.2f}); masks {tot_mask:.3e}; H {tot_h:.3e}; "
          f"all {tot_issue + tot_mask + tot_h:.3e} = x{(tot_issue + tot_mask + tot_h) / (tot_ideal + tot_h):.2f} of ideal")
    rate = 1024 * 2.0e9 / 4            # wave-instructions per second: 1024 SIMDs, 4 cycles each, ~2.0 GHz under FP64 load
    print(f"at {rate:.2e} FP64 wave-instructions/s: issued {1e3 * (tot_issue + tot_mask + tot_h) / rate:.2f} ms, ideal {1e3 * (tot_ideal + tot_h) / rate:.2f} ms")


if __name__ == "__main__":
    main()
