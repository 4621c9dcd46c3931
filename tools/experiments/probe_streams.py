#!/usr/bin/env python3
"""fused passes over the number of interleaved tile streams (fuse_streams_log2, fuse_stream_tile in qcx_kernels.h): n = 30 fused
Hadamard sweep, n = 28 inverse QFT exact and tolerance, n = 30 Shor circuit exact and tolerance; results compared bit for bit
against s = 0 (a checksum over the whole state) so that a number from a wrong pass cannot be read as a gain"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402


def best(reg, fn, reps=4):
    fn(); reg.synchronize()
    b = 1e9
    for _ in range(reps):
        reg.timer_start(); fn(); b = min(b, reg.timer_stop())
    return b


def window_bits(reg):
    """a few windows of the state as bytes (the whole-state comparisons are the suite's; this guards the probe)"""
    dim = 1 << reg.num_qubits
    return b"".join(reg.read(off, 4096).tobytes() for off in (0, dim // 3, dim // 2 + 12345, dim - 4096))


levels = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(0, 0), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0)]
with qc.Register(30, 0) as reg, qc.Register(28, 0) as r28, qc.Register(25, 5) as shor:
    def sweep():
        for q in range(30):
            qc.hadamard_gate(q, reg)
        reg.flush()

    def shor_circuit():
        qc.reset_register(shor); qc.quantum_computation(21, 2, shor); shor.flush()

    ref = {}
    for s, pos1 in levels:               # pos1 = position of the stream number in the tile number + 1; 0 = on top
        qc.tune(fuse_streams_log2=s, fuse_streams_pos=pos1)
        row = [f"s={s} pos={pos1 - 1 if pos1 else 'top'}"]
        reg.set_fusion(1); reg.fill_random(1); sweep(); w = window_bits(reg)
        ok = ref.setdefault("sweep", w) == w
        row.append(f"sweep30 {best(reg, sweep):7.3f} ms{'' if ok else ' WRONG'}")
        for mode, name in ((0, "exact"), (2, "tol")):
            r28.set_fusion(mode); r28.fill_random(1); qc.inverse_QFT(r28); w = window_bits(r28)
            ok = ref.setdefault("iqft" + name, w) == w
            row.append(f"iqft28 {name} {best(r28, lambda: qc.inverse_QFT(r28)):7.3f} ms{'' if ok else ' WRONG'}")
        for mode, name in ((0, "exact"), (2, "tol")):
            shor.set_fusion(mode); shor_circuit(); w = window_bits(shor)
            ok = ref.setdefault("shor" + name, w) == w
            row.append(f"shor30 {name} {best(shor, shor_circuit, 3):7.3f} ms{'' if ok else ' WRONG'}")
        print("   ".join(row), flush=True)
