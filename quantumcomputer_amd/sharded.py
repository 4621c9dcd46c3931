"""State vector sharded over the GPUs of one node, one process per GPU (torch.distributed; the
"nccl" backend is RCCL over xGMI on ROCm).

Layout: rank r of W = 2^k holds the 2^(n-k) amplitudes whose top k index bits are r.
  * H on a local qubit, every controlled phase (diagonal) and the controlled modular multiply
    (permutes only the low M bits) need NO communication: a global control bit is a per-rank
    constant -- ranks where it is 0 skip the gate.
  * H on a global qubit is the one exchange step.  Instead of pairwise half-shard swaps (one xGMI
    link each) the k rank bits are exchanged with the top k local bits in ONE all-to-all, which
    drives all 7 links of every GPU at once; the logical->physical qubit map records the swap and
    later gates are translated through it.  Measurement and read-back restore the identity map.

All arithmetic runs in libqcx.so through the shard-level C ABI (`HipEngine`); torch only owns the
device buffers, the stream and the collective.  The engine is injectable so that the host logic
(who skips, which bits, which chunks) can be exercised on CPU with the gloo backend in tests.
"""
import ctypes as C
import math

import numpy as np
import torch
import torch.distributed as dist

from ._lib import check, lib


class HipEngine:
    """shard-level C ABI on torch CUDA(HIP) tensors; launches on torch's current stream"""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipEngine needs a GPU tensor device; there is no CPU fallback")
        lib()

    @staticmethod
    def _s():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def reset(self, t, n_local, holds_one):
        check(lib().qcx_shard_reset(self._p(t), n_local, int(holds_one), self._s()), "qcx_shard_reset")

    def fill_random(self, t, n_local, first_global, seed, scale):
        check(lib().qcx_shard_fill_random(self._p(t), n_local, first_global, seed, scale, self._s()), "qcx_shard_fill_random")

    def hadamard(self, t, n_local, q):
        check(lib().qcx_shard_hadamard(self._p(t), n_local, q, self._s()), "qcx_shard_hadamard")

    def phase(self, t, n_local, mask, c, s):
        check(lib().qcx_shard_phase(self._p(t), n_local, mask, c, s, self._s()), "qcx_shard_phase")

    def camodc(self, t, n_local, M, Cn, A, ctl_local):
        check(lib().qcx_shard_camodc(self._p(t), n_local, M, Cn, A, ctl_local, self._s()), "qcx_shard_camodc")

    def norm2(self, t, n_local):
        out = C.c_double(0.0)
        check(lib().qcx_shard_norm2(self._p(t), n_local, C.byref(out), self._s()), "qcx_shard_norm2")
        return out.value

    def measure_scan(self, t, n_local, first_global, last_excluded, cum_in, r):
        found, idx, cum = C.c_int(0), C.c_uint64(0), C.c_double(0.0)
        check(lib().qcx_shard_measure_scan(self._p(t), n_local, first_global, last_excluded, cum_in, r,
                                           C.byref(found), C.byref(idx), C.byref(cum), self._s()), "qcx_shard_measure_scan")
        return bool(found.value), int(idx.value), float(cum.value)

    def collapse(self, t, n_local, local_index):
        check(lib().qcx_shard_collapse(self._p(t), n_local, local_index, self._s()), "qcx_shard_collapse")


class ShardedRegister:
    """Register (qc_shor.c:194-203) sharded by its top log2(world) qubits."""

    def __init__(self, L_size, M_size, device=None, group=None, engine=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        k = int(math.log2(self.world))
        if (1 << k) != self.world:
            raise ValueError("world size must be a power of two")
        self.k = k
        self.L_size, self.M_size = int(L_size), int(M_size)
        self.num_qubits = self.L_size + self.M_size
        self.num_states = 1 << self.num_qubits
        self.n_local = self.num_qubits - k
        if self.n_local < max(2 * k, 1) or self.M_size > self.n_local - k:
            raise ValueError("register too small for this many ranks (need n - k >= 2k and M <= n - 2k)")
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.engine = engine if engine is not None else HipEngine(self.device)
        self.bufs = [torch.zeros(2 << self.n_local, dtype=torch.float64, device=self.device) for _ in range(2 if k else 1)]
        self.cur = 0
        self.swapped = False          # True: rank bits and top-k local bits are exchanged
        self.exchanges = 0            # all-to-alls performed (statistics)

    # -- layout -------------------------------------------------------------------------------
    @property
    def shard(self):
        return self.bufs[self.cur]

    def phys(self, q):
        """physical bit position of logical qubit q"""
        if self.swapped and self.k:
            if self.n_local - self.k <= q < self.n_local:
                return q + self.k
            if q >= self.n_local:
                return q - self.k
        return q

    def _rank_bit(self, pq):
        return (self.rank >> (pq - self.n_local)) & 1

    def _toggle(self):
        """swap the k rank bits with the top k local bits: one all-to-all of equal chunks
        (chunk c of rank r <-> chunk r of rank c), result lands in the other buffer"""
        if not self.k:
            return
        src, dst = self.bufs[self.cur], self.bufs[self.cur ^ 1]
        dist.all_to_all_single(dst, src, group=self.group)
        self.cur ^= 1
        self.swapped = not self.swapped
        self.exchanges += 1

    def _identity(self):
        if self.swapped:
            self._toggle()

    # -- gates ----------------------------------------------------------------------------------
    def reset_register(self):
        self.swapped = False
        self.engine.reset(self.shard, self.n_local, self.rank == 0)

    def fill_random(self, seed):
        self.swapped = False
        self.engine.fill_random(self.shard, self.n_local, self.rank << self.n_local, int(seed),
                                math.sqrt(6.0 / float(self.num_states)))

    def hadamard_gate(self, q):
        if not 0 <= q < self.num_qubits:
            raise ValueError("bad qubit")
        if self.phys(q) >= self.n_local:
            self._toggle()
        self.engine.hadamard(self.shard, self.n_local, self.phys(q))

    def c_phase_shift_gate(self, c, t, theta):
        if c == t or not (0 <= c < self.num_qubits and 0 <= t < self.num_qubits):
            raise ValueError("bad qubit")
        mask = 0
        for pq in (self.phys(c), self.phys(t)):
            if pq >= self.n_local:
                if not self._rank_bit(pq):
                    return                        # this rank's amplitudes all have the bit at 0
            else:
                mask |= 1 << pq
        self.engine.phase(self.shard, self.n_local, mask, 1.0 * math.cos(theta), 1.0 * math.sin(theta))

    def c_amodc_gate(self, Cn, atox, ctl):
        if not 0 <= ctl < self.num_qubits:
            raise ValueError("bad qubit")
        pc = self.phys(ctl)
        if pc >= self.n_local:
            if not self._rank_bit(pc):
                return
            pc = -1
        self.engine.camodc(self.shard, self.n_local, self.M_size, Cn, int(atox % Cn), pc)

    def inverse_QFT(self):                               # qc_shor.c:678-690
        for l in range(self.L_size + self.M_size - 1, self.M_size - 1, -1):
            self.hadamard_gate(l)
            for kk in range(l - 1, self.M_size - 1, -1):
                self.c_phase_shift_gate(l, kk, math.pi / float(1 << (l - kk)))

    def quantum_computation(self, Cn, a):                # qc_shor.c:712-737, exact modular powers
        lo = self.num_qubits - self.L_size
        for l in range(lo, self.num_qubits):
            self.hadamard_gate(l)
        atox = a % Cn
        for l in range(lo, self.num_qubits):
            self.c_amodc_gate(Cn, atox, l)
            atox = (atox * atox) % Cn
        self.inverse_QFT()

    # -- measurement, reductions, read-back ---------------------------------------------------------
    def norm2(self):
        t = torch.tensor([self.engine.norm2(self.shard, self.n_local)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, group=self.group)
        return float(t.item())

    def measure_state(self, r):
        """qc_shor.c:272-306 over the shards: the sequential cumulative sum is handed from rank to
        rank in index order, so the selected index is the one the unsharded scan would pick."""
        self._identity()
        last_excluded = self.num_states - 1
        msg = torch.zeros(3, dtype=torch.float64, device=self.device)
        cum, found, idx = 0.0, False, last_excluded
        for rk in range(self.world):
            if self.rank == rk:
                f, i, cum_out = self.engine.measure_scan(self.shard, self.n_local, rk << self.n_local, last_excluded, cum, float(r))
                msg[0], msg[1], msg[2] = float(f), float(i), cum_out        # indices < 2^53: exact in a double
            dist.broadcast(msg, src=rk if self.group is None else dist.get_global_rank(self.group, rk), group=self.group)
            m = msg.tolist()
            cum = m[2]
            if m[0] != 0.0:
                found, idx = True, int(m[1])
                break
        owner = idx >> self.n_local
        self.engine.collapse(self.shard, self.n_local, idx & ((1 << self.n_local) - 1) if owner == self.rank else -1)
        return idx

    def local_numpy(self):
        """this rank's shard in the IDENTITY layout, as a numpy array of 2*2^n_local doubles"""
        self._identity()
        return self.shard.detach().cpu().numpy().copy()

    def gather(self):
        """whole state on every rank (tests only: 16 * 2^n bytes)"""
        self._identity()
        parts = [torch.empty_like(self.shard) for _ in range(self.world)]
        dist.all_gather(parts, self.shard, group=self.group)
        return torch.cat(parts).cpu().numpy()
