#!/usr/bin/env python3
"""diagnostic: the measurement scan on the n = 34 Shor state (tests/test_gpu_maxsize.py at QCX_TEST_NMAX=34), status and error text"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quantumcomputer_amd as qc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 34
L, M = n - 5, 5
with qc.Register(L, M) as reg:
    qc.reset_register(reg)
    qc.quantum_computation(21, 2, reg)
    print("norm", reg.norm2(), flush=True)
    reg.flush()
    ptr = reg.device_pointer()
    for ho in (1, 0):
        for blog in (11, 8):
            for r in (0.0, 0.5):
                qc.tune(meas_block_log=blog, meas_host_out=ho)
                found, index, cum = C.c_int(0), C.c_uint64(0), C.c_double(0.0)
                st = qc.lib().qcx_shard_measure_scan(ptr, n, 0, (1 << n) - 1, 0.0, float(r), C.byref(found), C.byref(index), C.byref(cum), None)
                print(f"host_out={ho} blog={blog} r={r}: status {st} found {found.value} index {index.value} cum {cum.value} err '{qc.lib().qcx_last_error().decode() if st else ''}'", flush=True)
