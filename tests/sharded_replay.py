"""Replays the step list of a DRY-RUN sharded register (qcx_register_create_sharded with devices[0] = -1, text from
qcx_sharded_trace) on the CPU: the whole state is held as one array in PHYSICAL index order (shard id = top k bits), the
gates are the oracle's, applied at the physical bit positions the schedule names; pack / permute / trade steps are the
index-bit permutations the kernels k_swap_bits / k_pack_push perform (`trade z`: the k shard-id bits change places with
the local bits z .. z+k-1).  If the scheduler's bookkeeping is right, the
array after `restore identity` equals the oracle's result of the same gate list in logical order, bit for bit."""
import numpy as np


def _swap_bits_index(n, pairs):
    """index array src such that new[j] = old[src[j]] for the transpositions applied in order"""
    idx = np.arange(1 << n, dtype=np.int64)
    src = idx.copy()
    # data moved by s_1, ..., s_m in that order: new[j] = old[s_1(s_2(...s_m(j)))]
    for a, b in reversed(pairs):
        x = ((src >> a) ^ (src >> b)) & 1
        src = src ^ ((x << a) | (x << b))
    return src


def _permute(state, n, pairs):
    v = state.reshape(-1, 2)
    return np.ascontiguousarray(v[_swap_bits_index(n, pairs)]).reshape(-1)


def _phase(state, n, pa, pb, c, s):
    v = state.reshape(-1, 2)
    idx = np.arange(1 << n, dtype=np.int64)
    sel = (((idx >> pa) & 1) & ((idx >> pb) & 1)).astype(bool)
    re, im = v[sel, 0].copy(), v[sel, 1].copy()
    v[sel, 0] = ((c * re) - (s * im)) + 0.0          # Q:409: separately rounded products, then the reference's "0 +"
    v[sel, 1] = ((c * im) + (s * re)) + 0.0          # Q:412


def replay(trace, n, k, M, state, ob):
    """apply the trace to `state` (float64, 2 * 2^n, physical order); returns (state, counts of step kinds)"""
    n_local = n - k
    counts = {"ops": 0, "pack": 0, "trade": 0, "permute": 0, "gates": 0}
    for line in trace.splitlines():
        t = line.split()
        if not t:
            continue
        if t[0] == "ops":
            counts["ops"] += 1
        elif t[0] == "h":
            assert int(t[1]) < n_local, "H on a shard-id bit must have been preceded by a trade"
            ob.hadamard(state, n, int(t[1])); counts["gates"] += 1
        elif t[0] == "p":
            _phase(state, n, int(t[1]), int(t[2]), float.fromhex(t[3]), float.fromhex(t[4])); counts["gates"] += 1
        elif t[0] == "c":
            pc = int(t[3])
            assert pc >= M
            ob.camodc(state, n, M, int(t[1]), int(t[2]), pc); counts["gates"] += 1
        elif t[0] in ("pack", "permute"):
            pairs = [tuple(int(x) for x in p.split(":")) for p in t[1:]]
            assert all(a < n_local and b < n_local for a, b in pairs)
            if pairs:
                state = _permute(state, n, pairs)
            counts[t[0]] += 1
        elif t[0] == "trade":
            zone_lo = int(t[1])                    # the k trade-zone bits sit below the spectator bits of a slice
            assert zone_lo + k <= n_local
            state = _permute(state, n, [(zone_lo + j, n_local + j) for j in range(k)])
            counts["trade"] += 1
        elif t[0] == "reset":
            ob.reset(state, n)
        else:
            raise ValueError(f"unknown step {line!r}")
    return state, counts
