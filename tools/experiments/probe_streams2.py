#!/usr/bin/env python3
"""fused Hadamard sweep over the tile-order experiments: streams x position of the stream number (argv: register sizes)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402


def best(reg, fn, reps=5):
    fn(); reg.synchronize()
    b = 1e9
    for _ in range(reps):
        reg.timer_start(); fn(); b = min(b, reg.timer_stop())
    return b


for n in [int(a) for a in sys.argv[1:]] or [30]:
    with qc.Register(n, 0) as reg:
        reg.set_fusion(1); reg.fill_random(1)

        def sweep():
            for q in range(n):
                qc.hadamard_gate(q, reg)
            reg.flush()
        dim = 1 << n
        ref = None
        for s, pos in ((0, 0), (3, 0), (2, 3), (2, 4), (3, 2), (3, 3), (3, 4), (3, 5), (3, 6), (4, 3), (4, 4), (4, 5), (5, 4), (6, 4), (0, 0), (3, 4)):
            qc.tune(fuse_streams_log2=s, fuse_streams_pos=pos)
            reg.fill_random(1); sweep()
            w = b"".join(reg.read(off, 4096).tobytes() for off in (0, dim // 3, dim // 2 + 12345, dim - 4096))
            ref = ref or w
            print(f"n={n} s={s} pos={pos - 1 if pos else 'top':>3}: {best(reg, sweep):7.3f} ms{'' if w == ref else ' WRONG'}", flush=True)
