"""State vector sharded over the GPUs of one node, one process per GPU (torch.distributed; the
"nccl" backend is RCCL over xGMI on ROCm).

Layout: rank r of W = 2^k holds the 2^(n-k) amplitudes whose top k PHYSICAL index bits
are r.  A logical->physical qubit permutation is kept on the host.
  * H on a qubit that is physically local, every controlled phase (diagonal) and the controlled
    modular multiply (permutes only the low M bits, which never move) need NO communication: a
    control that sits in the rank id is a per-rank constant -- ranks where it is 0 skip the gate.
  * H on a qubit that is physically global is the one exchange step.  Instead of pairwise half-shard
    swaps (each rides ONE xGMI link) all k rank bits are traded at once for k local bits with ONE
    all-to-all, which drives all 7 links of every GPU together:
        pack   (local, qcx_shard_swap_bits): bring the k local bits to be given up to the top k
               local positions -- out of place, so it doubles as the send-buffer fill;
        trade  (all_to_all_single): chunk c of rank r <-> chunk r of rank c.
    Gates are queued (like the single-GPU fusion queue) and executed on flush, so the choice of WHICH
    local qubits to give up can look ahead: the k candidates whose next use as an H target lies
    furthest in the future are evicted (Belady).  A cyclic H sweep then costs one exchange per sweep,
    the Shor circuit two in total.
  * measurement, norm and read-back flush the queue; measurement and read-back also restore the
    identity layout.
  * PAIRWISE form of the exchange (`exchange="pairwise"` / QCX_SHARD_EXCHANGE=pairwise; SURVEY s8(e)'s literal
    form, kept as the fallback should the all-to-all misbehave on a node): ONE rank bit j is traded for ONE
    local bit.  The local bit is brought to the top of the (slice of the) shard by the pack pass, so that the
    half to give away is contiguous; rank r sends it to r xor 2^j and receives the partner's matching half
    (dist.batch_isend_irecv = ncclSend/ncclRecv in one group under RCCL; 2^(k-1) disjoint pairs run at once),
    the qubit map is relabelled -- nothing is sent back.  Same gate kernels, same bits.

All arithmetic runs in libqcx.so through the shard-level C ABI (`HipEngine`); torch only owns the
device buffers, the stream and the collective.  The engine is injectable so that the host logic can be
exercised on CPU with the gloo backend in tests.
"""
import ctypes as C
import math
import os

import torch
import torch.distributed as dist

from ._lib import GateDesc, check, lib, polar

MIN_EVICT_POS = 6          # never trade away index bits below this: runs of the pack pass stay >= 1 KiB


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

    def run_ops(self, t, n_local, M, descs, mode=1):
        """a list of (type, q, mask, c, s, C, A) tuples through the fusion scheduler (qcx_shard_run_fused_mode: 1 = bit-exact
        passes, 2 = the tolerance mode's passes)"""
        if not descs:
            return
        arr = (GateDesc * len(descs))()
        for i, d in enumerate(descs):
            arr[i].type, arr[i].q, arr[i].mask, arr[i].c, arr[i].s, arr[i].C, arr[i].A = d
        check(lib().qcx_shard_run_fused_mode(int(mode), self._p(t), n_local, M, len(descs), C.cast(arr, C.c_void_p), self._s()), "qcx_shard_run_fused_mode")

    def basis_front(self, t, n_local, first_global, n, M, basis, descs):
        """this shard's part of the basis state, fused with the closed-form front of `descs` (global qubit numbers);
        returns the number of gates consumed (qcx_shard_basis_front)"""
        arr = (GateDesc * max(len(descs), 1))()
        for i, d in enumerate(descs):
            arr[i].type, arr[i].q, arr[i].mask, arr[i].c, arr[i].s, arr[i].C, arr[i].A = d
        used = C.c_uint(0)
        check(lib().qcx_shard_basis_front(self._p(t), n_local, first_global, n, M, basis, len(descs), C.cast(arr, C.c_void_p),
                                          C.byref(used), self._s()), "qcx_shard_basis_front")
        return int(used.value)

    # compact circuits: the front's orbit, the front in the compact form, the expansion
    def compact_plan(self, n, M, basis, descs):
        """(gates of the closed-form front, column bits, orbit) -- orbit empty: no compact form"""
        arr = (GateDesc * max(len(descs), 1))()
        for i, d in enumerate(descs):
            arr[i].type, arr[i].q, arr[i].mask, arr[i].c, arr[i].s, arr[i].C, arr[i].A = d
        used, cb, ncols = C.c_uint(0), C.c_uint(0), C.c_uint(0)
        orbit = (C.c_uint16 * 16)()
        check(lib().qcx_compact_plan(n, M, basis, len(descs), C.cast(arr, C.c_void_p), C.byref(used), C.byref(cb), C.byref(ncols), orbit), "qcx_compact_plan")
        return used.value, cb.value, [int(orbit[j]) for j in range(ncols.value)]

    def compact_front(self, t, n_local_compact, first_global, n, M, basis, descs, cb, orbit):
        arr = (GateDesc * max(len(descs), 1))()
        for i, d in enumerate(descs):
            arr[i].type, arr[i].q, arr[i].mask, arr[i].c, arr[i].s, arr[i].C, arr[i].A = d
        ob = (C.c_uint16 * 16)(*orbit)
        check(lib().qcx_shard_compact_front(self._p(t), n_local_compact, first_global, n, M, basis, len(descs), C.cast(arr, C.c_void_p),
                                            cb, len(orbit), ob, self._s()), "qcx_shard_compact_front")

    def expand_compact(self, tc, t, n_local, M, cb, orbit):
        ob = (C.c_uint16 * 16)(*orbit)
        check(lib().qcx_shard_expand_compact(self._p(tc), self._p(t), n_local, M, cb, len(orbit), ob, self._s()), "qcx_shard_expand_compact")

    def swap_bits(self, src, dst, n_local, pos_a, pos_b):
        m = len(pos_a)
        a = (C.c_uint * max(m, 1))(*pos_a)
        b = (C.c_uint * max(m, 1))(*pos_b)
        check(lib().qcx_shard_swap_bits(self._p(src), self._p(dst), n_local, m, a, b, self._s()), "qcx_shard_swap_bits")

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


class _Done:
    """a finished collective (host-staged exchange)"""

    def wait(self):
        return True


_DONE = _Done()


class _PairWork:
    """the send + receive of one pairwise half swap; wait() also moves the received half next to the kept one"""

    def __init__(self, reqs, finish):
        self.reqs, self.finish = reqs, finish

    def wait(self):
        for r in self.reqs:
            r.wait()
        if self.finish is not None:
            self.finish()
            self.finish = None
        return True


class ShardedRegister:
    """Register (qc_shor.c:194-203) sharded by its top log2(world) physical index bits."""

    def __init__(self, L_size, M_size, device=None, group=None, engine=None, max_queue=8192, slices_log2=None,
                 dry_run=False, fusion=True, exchange=None):
        self.group = group
        # the form of the exchange step: "alltoall" (all k rank bits traded at once) or "pairwise" (one rank bit per swap)
        self.exchange_form = (exchange or os.environ.get("QCX_SHARD_EXCHANGE", "alltoall")).lower()
        if self.exchange_form not in ("alltoall", "pairwise"):
            raise ValueError("exchange must be 'alltoall' or 'pairwise'")
        self.pair_swaps = 0               # pairwise half-shard swaps performed
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        k = int(math.log2(self.world))
        if (1 << k) != self.world:
            raise ValueError("world size must be a power of two")
        self.k = k
        self.L_size, self.M_size = int(L_size), int(M_size)
        self.num_qubits = n = self.L_size + self.M_size
        self.num_states = 1 << n
        self.n_local = n - k
        # sigma spectator bits -> 2^sigma slices for the overlapped exchange (QCX_SHARD_SLICES_LOG2, default 3)
        sigma = int(os.environ.get("QCX_SHARD_SLICES_LOG2", "3")) if slices_log2 is None else int(slices_log2)
        if not k:
            sigma = 0
        while True:
            self.min_evict = max(MIN_EVICT_POS if self.n_local - sigma - 2 * k >= MIN_EVICT_POS else 0, self.M_size)
            if sigma == 0 or self.n_local - sigma - self.min_evict >= 2 * k:
                break
            sigma -= 1
        if self.n_local < 1 or (k and self.n_local - self.min_evict < 2 * k):
            raise ValueError("register too small for this many ranks (need n_local - max(M, 6) >= 2 log2(world))")
        self.sigma = sigma
        self.slice_bits = self.n_local - sigma
        self.zone_lo = self.slice_bits - k
        self.overlap = os.environ.get("QCX_SHARD_OVERLAP", "1") != "0"
        # async_exchange False: every all-to-all is issued synchronously (async_op=False) -- the conservative fallback
        # (QCX_SHARD_OVERLAP=0 selects it together with no overlap; bench.py reports which mode ran)
        self.async_exchange = self.overlap
        self.overlapped_gates = 0
        # run each window's gate list through the fused-pass scheduler (qcx_shard_run_fused): same bits, fewer HBM passes.
        # False: one kernel launch per gate (bench.py --gpus N uses that to stay comparable with its N = 1 headline)
        self.fusion = bool(fusion)
        self.fusion_mode = 2 if (fusion is not True and fusion is not False and int(fusion) == 2) else 1     # 2: tolerance mode per shard
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.engine = engine if engine is not None else HipEngine(self.device)
        backend = dist.get_backend(group) if hasattr(dist, "get_backend") else ""     # (test doubles may not have it)
        self._host_staged = not dry_run and self.device.type == "cuda" and self.world > 1 and backend == "gloo"
        self.dry_run = bool(dry_run)              # schedule only (tools/model_sharded.py): no amplitude storage
        if self.dry_run:
            self.bufs = [("buf", 0), ("buf", 1)]
        else:
            self.bufs = [torch.zeros(2 << self.n_local, dtype=torch.float64, device=self.device) for _ in range(2 if k else 1)]
        self.cur = 0
        self.perm = list(range(n))        # logical qubit -> physical index bit
        self.inv = list(range(n))         # physical index bit -> logical qubit
        self.queue = []
        self.max_queue = max_queue
        self.exchanges = 0                # all-to-alls performed
        self.fronts = 0                   # circuit fronts written as one pass (qcx_shard_basis_front)
        self.compact_circuits = 0         # flushes that ran on the companion register of compact circuits (_try_compact)
        self.compact_measures = 0         # measurements that scanned the companion register instead of the expanded state
        self._comp = None                 # that companion: L + cb qubits on the same ranks, created on first use
        self._compact = None              # (cb, orbit) while the result of a compact circuit still sits on the companion (round 5:
                                          # expanded when something other than measure_state looks at the state -- or never)
        self._slices_log2 = slices_log2
        self._basis_pending = False
        self.pack_passes = 0              # local pack passes performed
        self.profile = None               # set to [] to collect (gate, ms, exchanged) per executed gate (cuda only)

    # -- layout -------------------------------------------------------------------------------
    @property
    def shard(self):
        return self.bufs[self.cur]

    def phys(self, q):
        return self.perm[q]

    def _rank_bit(self, pq):
        return (self.rank >> (pq - self.n_local)) & 1

    def _set_phys(self, logical, pos):
        self.perm[logical] = pos
        self.inv[pos] = logical

    # -- slices ---------------------------------------------------------------------------------
    # physical local index bits:  [ 0 .. zone_lo )  [ zone_lo .. slice_bits ) = trade zone (k bits)
    #                             [ slice_bits .. n_local ) = spectator bits (sigma bits, never traded)
    # A slice = fixed spectator bits = one contiguous 2^slice_bits range of the shard.  The exchange with the
    # rank id is done slice by slice (pack -> all-to-all), which lets the gates before/after it run on the
    # slices that are not in flight: exchange and compute overlap.
    def _views(self, buf):
        if self.dry_run:
            return [(buf, s) for s in range(1 << self.sigma)]
        w = 2 << self.slice_bits
        return [buf[s * w:(s + 1) * w] for s in range(1 << self.sigma)]

    def _plan_give(self, give):
        """transpositions (application order) that bring local position give[j] to trade-zone slot j"""
        pos, swaps = list(give), []
        for j in range(self.k):
            t, p = self.zone_lo + j, pos[j]
            if p != t:
                swaps.append((p, t))
                for jj in range(j + 1, self.k):
                    if pos[jj] == t:
                        pos[jj] = p
        return swaps

    def _book(self, swaps, trade):
        for a, b in swaps:
            la, lb = self.inv[a], self.inv[b]
            self._set_phys(la, b); self._set_phys(lb, a)
        if trade:
            for j in range(self.k):
                lt, lr = self.inv[self.zone_lo + j], self.inv[self.n_local + j]
                self._set_phys(lt, self.n_local + j); self._set_phys(lr, self.zone_lo + j)

    def _move_slice(self, sidx, swaps, src_buf, dst_buf, async_op):
        """pack (optional) + all-to-all of one slice; returns (work, buffer holding the result)"""
        src, dst = self._views(src_buf)[sidx], self._views(dst_buf)[sidx]
        if swaps:
            rev = list(reversed(swaps))            # dst[j] = src[s_1(...s_m(j))]: kernel applies its list in array order
            self.engine.swap_bits(src, dst, self.slice_bits, [x[0] for x in rev], [x[1] for x in rev])
            src, dst, out = dst, src, src_buf
        else:
            out = dst_buf
        if self._host_staged:
            # gloo has no device all-to-all: stage through the host (rehearsals of the multi-process path on one GPU,
            # tests/test_gpu_sharded_multiproc.py).  Production is nccl (= RCCL), device to device.
            src_h = src.cpu()
            dst_h = torch.empty_like(src_h)
            dist.all_to_all_single(dst_h, src_h, group=self.group)
            dst.copy_(dst_h)
            return _DONE, out
        work = dist.all_to_all_single(dst, src, group=self.group, async_op=async_op)
        return (work if work is not None else _DONE), out

    def _trade_now(self, give):
        """whole-shard exchange without any overlap (used when restoring the identity layout)"""
        swaps = self._plan_give(give)
        src_buf, dst_buf = self.bufs[self.cur], self.bufs[self.cur ^ 1]
        for sidx in range(1 << self.sigma):
            self._move_slice(sidx, swaps, src_buf, dst_buf, False)
        if not swaps:                      # with a pack pass the data travels there and back: same buffer
            self.cur ^= 1
        self._book(swaps, True)
        self.exchanges += 1
        self.pack_passes += 1 if swaps else 0

    # -- the pairwise form: rank bit j <-> the top bit of the slice --------------------------------------
    def _plan_pair(self, give):
        """the transposition (if any) that brings local position `give` to the top of the slice"""
        top = self.slice_bits - 1
        return [(give, top)] if give != top else []

    def _book_pair(self, swaps, j):
        for a, b in swaps:
            la, lb = self.inv[a], self.inv[b]
            self._set_phys(la, b); self._set_phys(lb, a)
        top = self.slice_bits - 1
        lt, lr = self.inv[top], self.inv[self.n_local + j]
        self._set_phys(lt, self.n_local + j); self._set_phys(lr, top)

    def _peer(self, j):
        partner = self.rank ^ (1 << j)
        return partner if self.group is None else dist.get_global_rank(self.group, partner)

    def _swap_slice(self, sidx, swaps, src_buf, dst_buf, j, async_op):
        """pack (optional) + half swap of one slice with rank ^ 2^j; returns (work, buffer holding the result).
        With b = this rank's bit j and x = the slice's top bit: an amplitude (rank bit b, top bit x) belongs, after the
        swap, to the rank whose bit j is x, at top bit b.  So the half x = 1 - b leaves for the partner, and the partner's
        half x' = b arrives -- at top bit 1 - b (the partner's old rank bit), i.e. exactly where the half that left was."""
        src, dst = self._views(src_buf)[sidx], self._views(dst_buf)[sidx]
        if swaps:
            rev = list(reversed(swaps))
            self.engine.swap_bits(src, dst, self.slice_bits, [x[0] for x in rev], [x[1] for x in rev])
            keep_buf, keep, other = dst_buf, dst, src
        else:
            keep_buf, keep, other = src_buf, src, dst
        if self.dry_run:
            return dist.pair_swap(keep, other, async_op), keep_buf
        b = (self.rank >> j) & 1
        w = 1 << self.slice_bits                    # doubles per half: 2 * 2^(slice_bits - 1)
        lo = (1 - b) * w
        send, recv = keep[lo:lo + w], other[lo:lo + w]
        peer = self._peer(j)
        if self._host_staged:                       # gloo with GPU tensors (rehearsals on one GPU): stage through the host
            send_h = send.cpu()
            recv_h = torch.empty_like(send_h)
            for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, send_h, peer, self.group), dist.P2POp(dist.irecv, recv_h, peer, self.group)]):
                r.wait()
            send.copy_(recv_h)
            return _DONE, keep_buf
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, send, peer, self.group), dist.P2POp(dist.irecv, recv, peer, self.group)])
        work = _PairWork(reqs, lambda: send.copy_(recv))
        if not async_op:
            work.wait()
        return work, keep_buf

    def _pair_now(self, give, j):
        """whole-shard pairwise swap without any overlap (restoring the identity layout)"""
        swaps = self._plan_pair(give)
        src_buf, dst_buf = self.bufs[self.cur], self.bufs[self.cur ^ 1]
        out = src_buf
        for sidx in range(1 << self.sigma):
            _, out = self._swap_slice(sidx, swaps, src_buf, dst_buf, j, False)
        if out is dst_buf:
            self.cur ^= 1
        self._book_pair(swaps, j)
        self.exchanges += 1
        self.pair_swaps += 1
        self.pack_passes += 1 if swaps else 0

    def _local_permute(self, swaps):
        """transpositions among local positions on the whole shard (8 per out-of-place pass)"""
        for lo in range(0, len(swaps), 8):
            part = swaps[lo:lo + 8]
            src, dst = self.bufs[self.cur], self.bufs[self.cur ^ 1]
            rev = list(reversed(part))
            self.engine.swap_bits(src, dst, self.n_local, [x[0] for x in rev], [x[1] for x in rev])
            self.cur ^= 1
            self.pack_passes += 1
            self._book(part, False)

    def _next_use(self, logical, start):
        """index in the queue of the next H on `logical` at or after `start` (len(queue) + 1 = never)"""
        for i in range(start, len(self.queue)):
            g = self.queue[i]
            if g[0] == "h" and g[1] == logical:
                return i
        return len(self.queue) + 1

    def _choose_give(self, at, count=None):
        """the k (or `count`) local positions (outside the spectator bits) whose qubits are H targets latest (Belady)"""
        cand = list(range(self.min_evict, self.slice_bits))
        cand.sort(key=lambda p: (-self._next_use(self.inv[p], at), -p))
        return sorted(cand[:self.k if count is None else count])

    def _identity(self):
        """restore logical == physical (the order measurement and read-back need)"""
        self.flush()
        n, nl, k = self.num_qubits, self.n_local, self.k
        if self.perm == list(range(n)):
            return
        G = list(range(nl, n))
        if k and self.exchange_form == "pairwise":
            for j in range(k):
                g = nl + j
                if self.perm[g] == g:
                    continue
                if self.perm[g] >= nl:
                    # the rightful owner of rank bit j sits in a later rank slot: bring it local first (the stranger that
                    # goes there instead is dealt with when that slot's turn comes)
                    cand = [p for p in range(self.slice_bits - 1, self.min_evict - 1, -1) if self.inv[p] not in G]
                    self._pair_now(cand[0], self.perm[g] - nl)
                self._pair_now(self.perm[g], j)
        elif k and any(self.perm[g] != g for g in G):
            if any(self.perm[g] >= nl for g in G):
                # some rightful rank-id qubits sit in the rank id but in the wrong slot / beside strangers:
                # one trade brings the whole rank id local (giving up positions that hold none of G)
                cand = [p for p in range(self.slice_bits - 1, self.min_evict - 1, -1) if self.inv[p] not in G][:k]
                self._trade_now(cand)
            self._trade_now([self.perm[nl + j] for j in range(k)])     # rank bit j <- logical qubit nl + j
        swaps, perm, inv = [], list(self.perm), list(self.inv)
        for q in range(nl):
            a = perm[q]
            if a != q:
                swaps.append((a, q))
                other = inv[q]
                perm[q], perm[other] = q, a
                inv[q], inv[a] = q, other
        self._local_permute(swaps)
        assert self.perm == list(range(n)), self.perm

    # -- queue ----------------------------------------------------------------------------------
    def _push(self, g):
        self.queue.append(g)
        if len(self.queue) >= self.max_queue:
            self.flush()

    def _resolve(self, g):
        """queue entry -> physical operation under the CURRENT layout"""
        if g[0] == "h":
            return ("h", self.perm[g[1]])
        if g[0] == "p":
            return ("p", self.perm[g[1]], self.perm[g[2]], g[3], g[4])
        return ("c", g[1], g[2], self.perm[g[3]])

    def _bit_outside(self, pq, nbits, sidx):
        """value of physical index bit pq >= nbits for the amplitudes of this view: spectator bit of the slice
        or bit of the rank id"""
        if pq >= self.n_local:
            return (self.rank >> (pq - self.n_local)) & 1
        return (sidx >> (pq - nbits)) & 1

    def _run(self, op, view, nbits, sidx):
        """execute one resolved operation on `view` = 2^nbits consecutive amplitudes of the shard"""
        ev0 = None
        if self.profile is not None and self.device.type == "cuda":
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True); ev0.record()
        if op[0] == "h":
            assert op[1] < nbits
            self.engine.hadamard(view, nbits, op[1])
        elif op[0] == "p":
            mask, skip = 0, False
            for pq in (op[1], op[2]):
                if pq >= nbits:
                    skip = skip or not self._bit_outside(pq, nbits, sidx)   # all amplitudes here have the bit at 0
                else:
                    mask |= 1 << pq
            if not skip:
                self.engine.phase(view, nbits, mask, op[3], op[4])
        else:
            pc = op[3]
            if pc >= nbits:
                if self._bit_outside(pc, nbits, sidx):
                    self.engine.camodc(view, nbits, self.M_size, op[1], op[2], -1)
            else:
                self.engine.camodc(view, nbits, self.M_size, op[1], op[2], pc)
        if ev0 is not None:
            ev1.record()
            self.profile.append((op[0], op[1] if op[0] == "h" else -1, ev0, ev1, nbits))

    def _run_ops(self, ops, view, nbits, sidx):
        """a list of resolved operations on one view: one by one, or as ONE fused gate list"""
        if not ops:
            return
        if not (self.fusion and hasattr(self.engine, "run_ops")):
            for op in ops:
                self._run(op, view, nbits, sidx)
            return
        descs = []
        for op in ops:
            if op[0] == "h":
                assert op[1] < nbits
                descs.append((0, op[1], 0, 0.0, 0.0, 0, 0))
            elif op[0] == "p":
                mask, skip = 0, False
                for pq in (op[1], op[2]):
                    if pq >= nbits:
                        skip = skip or not self._bit_outside(pq, nbits, sidx)
                    else:
                        mask |= 1 << pq
                if not skip:
                    descs.append((1, 0, mask, op[3], op[4], 0, 0))
            else:
                pc = op[3]
                if pc >= nbits:
                    if self._bit_outside(pc, nbits, sidx):
                        descs.append((2, 0xFFFFFFFF, 0, 0.0, 0.0, op[1], op[2]))
                else:
                    descs.append((2, pc, 0, 0.0, 0.0, op[1], op[2]))
        ev0 = None
        if self.profile is not None and self.device.type == "cuda":
            ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True); ev0.record()
        if self.fusion_mode == 2:
            self.engine.run_ops(view, nbits, self.M_size, descs, 2)
        else:
            self.engine.run_ops(view, nbits, self.M_size, descs)
        if ev0 is not None:
            ev1.record()
            self.profile.append(("f", -1, ev0, ev1, nbits))

    def _sliceable(self, g):
        """may this queued gate run slice by slice under the current layout?  (an H must not target a spectator
        bit, nor a qubit of the rank id)"""
        return g[0] != "h" or self.perm[g[1]] < self.slice_bits

    def _compact_local(self):
        """this rank's own view of whether the queue can run on the companion register: None, or (used, cb, orbit, descs, rest)
        with the companion in self._comp.  Never raises and runs no collective: the ranks compare notes in _try_compact."""
        eng = self.engine
        if os.environ.get("QCX_SHARD_COMPACT", "1") == "0":
            return None
        n, M = self.num_qubits, self.M_size
        descs = []
        for g in self.queue:
            if g[0] == "h":
                descs.append((0, g[1], 0, 0.0, 0.0, 0, 0))
            elif g[0] == "c":
                descs.append((2, g[3], 0, 0.0, 0.0, g[1], g[2]))
            else:
                break
        if not descs or self.n_local < M + 6:
            return None
        try:
            used, cb, orbit = eng.compact_plan(n, M, 1, descs)
        except Exception:
            return None
        if not orbit or not used or used >= len(self.queue):
            return None
        rest = self.queue[used:]
        for g in rest:
            if not ((g[0] == "h" and g[1] >= M) or (g[0] == "p" and g[1] >= M and g[2] >= M)):
                return None
        comp = self._comp
        if comp is None or comp.M_size != cb:
            try:
                comp = ShardedRegister(self.L_size, cb, device=self.device, group=self.group, engine=eng, max_queue=self.max_queue,
                                       slices_log2=self._slices_log2, fusion=(2 if self.fusion_mode == 2 else True),
                                       exchange=self.exchange_form)
            except Exception:                      # too small for this many ranks (ValueError), or no memory for its two buffers
                return None
            comp._try_compact = lambda: False      # (never a compact circuit of its own)
            self._comp = comp
        return used, cb, orbit, descs, rest

    def _try_compact(self):
        """Behind the circuit front the M register reads one of the residues of the multiply ladder's orbit; when nothing else
        in the queue touches it the whole queue runs on a COMPANION register of L + cb qubits -- [L register][orbit column],
        the same ranks, 2^(M - cb) times smaller: fused passes and all-to-alls alike -- and every rank expands its part into the
        real register (the C host: sh_compact; one GPU: compact_chain).  The companion's flush runs collectives, so the ranks
        must take the SAME decision: each forms its own (it depends on per-process state -- QCX_SHARD_COMPACT, the library's
        tune values, whether the companion's buffers could be allocated) and one small all-reduce (min) makes it common; a
        rank that cannot, vetoes for all.  True: done."""
        eng = self.engine
        if not (self.fusion and hasattr(eng, "compact_plan") and self.queue) or self.dry_run:
            return False
        plan = self._compact_local()
        if self.world > 1:
            cb = plan[1] if plan else 0
            t = torch.tensor([1.0 if plan else 0.0, float(cb), -float(cb)], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            v = t.tolist()
            if v[0] != 1.0 or v[1] != -v[2]:
                return False
        if not plan:
            return False
        used, cb, orbit, descs, rest = plan
        n, M = self.num_qubits, self.M_size
        comp = self._comp
        comp.queue = []
        comp.perm, comp.inv = list(range(comp.num_qubits)), list(range(comp.num_qubits))
        comp._basis_pending = False
        ex0, pp0, og0, ps0 = comp.exchanges, comp.pack_passes, comp.overlapped_gates, comp.pair_swaps
        eng.compact_front(comp.shard, comp.n_local, self.rank << self.n_local, n, M, 1, descs[:used], cb, orbit)
        sh = M - cb
        comp.queue = [("h", g[1] - sh) if g[0] == "h" else ("p", g[1] - sh, g[2] - sh, g[3], g[4]) for g in rest]
        comp.flush()
        comp._identity()
        self._compact = (cb, list(orbit))          # the real register is written by _expand_compact -- or never
        self.exchanges += comp.exchanges - ex0
        self.pack_passes += comp.pack_passes - pp0
        self.overlapped_gates += comp.overlapped_gates - og0
        self.pair_swaps += comp.pair_swaps - ps0
        self._basis_pending = False
        self.queue = []
        self.fronts += 1
        self.compact_circuits += 1
        return True

    def _expand_compact(self):
        """the real register from the companion's compact form (every rank expands its own part; no communication)"""
        if self._compact is None:
            return
        cb, orbit = self._compact
        self._compact = None
        self.engine.expand_compact(self._comp.shard, self.shard, self.n_local, self.M_size, cb, orbit)

    def flush(self, keep_compact=False):
        if self._compact is not None:            # an earlier flush left the state on the companion register
            if keep_compact and not self.queue and not getattr(self, "_basis_pending", False):
                return
            if getattr(self, "_basis_pending", False):
                self._compact = None             # (a reset came after it: the compact form is history)
            else:
                self._expand_compact()
        if getattr(self, "_basis_pending", False) and self._try_compact():
            if not keep_compact:
                self._expand_compact()           # (only measure_state leaves the result on the companion)
            return
        if getattr(self, "_basis_pending", False):
            self._materialize_basis()
        if not self.queue:
            return
        q, nl, S = self.queue, self.n_local, 1 << self.sigma
        i = 0
        while i < len(q):
            x = next((j for j in range(i, len(q)) if q[j][0] == "h" and self.perm[q[j][1]] >= nl), None)
            if x is None:
                self._run_ops([self._resolve(g) for g in q[i:]], self.shard, nl, 0)
                break
            # gates that can share the pipeline with the exchange at x: a run before it ...
            a = x
            if S > 1 and self.overlap:
                while a > i and self._sliceable(q[a - 1]):
                    a -= 1
            self._run_ops([self._resolve(g) for g in q[i:a]], self.shard, nl, 0)
            pre_ops = [self._resolve(g) for g in q[a:x]]            # resolved under the layout before the trade
            pair_j = -1
            if self.exchange_form == "pairwise":                    # one rank bit: the one the Hadamard at x needs
                pair_j = self.perm[q[x][1]] - nl
                swaps = self._plan_pair(self._choose_give(x, 1)[0])
                self._book_pair(swaps, pair_j)
            else:
                swaps = self._plan_give(self._choose_give(x))
                self._book(swaps, True)
            # ... and a run after it, under the new layout
            b = x
            while b < len(q) and self._sliceable(q[b]) and not (q[b][0] == "h" and self.perm[q[b][1]] >= nl):
                b += 1
                if not (S > 1 and self.overlap) and b > x:           # no overlap: only the gate that needed the trade
                    break
            if S > 1 and self.overlap and b < len(q) and q[b][0] == "h" and self.perm[q[b][1]] >= nl:
                # the run ends at the NEXT exchange: leave its second half to that exchange's pre-window, so
                # that both transfers have gates to hide behind
                b = x + max(1, (b - x + 1) // 2)
            post_ops = [self._resolve(g) for g in q[x:b]]
            src_buf, dst_buf = self.bufs[self.cur], self.bufs[self.cur ^ 1]
            src_views = self._views(src_buf)
            works, outs = [None] * S, [None] * S
            for sidx in range(S):
                self._run_ops(pre_ops, src_views[sidx], self.slice_bits, sidx)
                if pair_j >= 0:
                    works[sidx], outs[sidx] = self._swap_slice(sidx, swaps, src_buf, dst_buf, pair_j, self.async_exchange)
                else:
                    works[sidx], outs[sidx] = self._move_slice(sidx, swaps, src_buf, dst_buf, self.async_exchange)
                if sidx >= 1:
                    works[sidx - 1].wait()
                    view = self._views(outs[sidx - 1])[sidx - 1]
                    self._run_ops(post_ops, view, self.slice_bits, sidx - 1)
            works[S - 1].wait()
            view = self._views(outs[S - 1])[S - 1]
            self._run_ops(post_ops, view, self.slice_bits, S - 1)
            if outs[0] is dst_buf:
                self.cur ^= 1
            self.exchanges += 1
            self.pair_swaps += 1 if pair_j >= 0 else 0
            self.pack_passes += 1 if swaps else 0
            self.overlapped_gates += len(pre_ops) + len(post_ops)
            i = b
        self.queue = []

    def synchronize(self):
        self.flush()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    # -- gates ----------------------------------------------------------------------------------
    def reset_register(self):
        self.queue = []                                   # pending gates act on a state that is being overwritten
        self._compact = None                              # (and so does a compact result nobody looked at)
        n = self.num_qubits
        self.perm, self.inv = list(range(n)), list(range(n))
        if self.fusion and hasattr(self.engine, "basis_front"):
            self._basis_pending = True                    # lazily: written at the next flush, fused with the circuit front (K0b)
            return
        self._basis_pending = False
        self.engine.reset(self.shard, self.n_local, self.rank == 0)

    def _materialize_basis(self):
        """a lazily pending reset: this rank's part of |0...01>, together with the queue's closed-form front (Hadamards on
        distinct qubits -- shard-id qubits included --, then controlled modular multiplies): no exchange.  Every rank sees
        the same queue and drops the same number of gates."""
        descs = []
        for g in self.queue:
            if g[0] == "h":
                descs.append((0, g[1], 0, 0.0, 0.0, 0, 0))
            elif g[0] == "c":
                descs.append((2, g[3], 0, 0.0, 0.0, g[1], g[2]))
            else:
                break
        used = self.engine.basis_front(self.shard, self.n_local, self.rank << self.n_local, self.num_qubits, self.M_size, 1, descs)
        self._basis_pending = False        # only after the write was launched: a failed launch leaves the reset pending
        if used:
            self.queue = self.queue[used:]
            self.fronts += 1

    def fill_random(self, seed):
        self._basis_pending = False
        self._compact = None
        self.queue = []
        n = self.num_qubits
        self.perm, self.inv = list(range(n)), list(range(n))
        self.engine.fill_random(self.shard, self.n_local, self.rank << self.n_local, int(seed),
                                math.sqrt(6.0 / float(self.num_states)))

    def hadamard_gate(self, q):
        if not 0 <= q < self.num_qubits:
            raise ValueError("bad qubit")
        self._push(("h", q))

    def c_phase_shift_gate(self, c, t, theta):
        if c == t or not (0 <= c < self.num_qubits and 0 <= t < self.num_qubits):
            raise ValueError("bad qubit")
        cs, sn = polar(theta)                      # glibc sincos, like the single-GPU path and the reference
        self._push(("p", c, t, cs, sn))

    def c_amodc_gate(self, Cn, atox, ctl):
        if not 0 <= ctl < self.num_qubits:
            raise ValueError("bad qubit")
        self._push(("c", int(Cn), int(atox % Cn), ctl))

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
        self.flush()
        t = torch.tensor([self.engine.norm2(self.shard, self.n_local)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, group=self.group)
        return float(t.item())

    def measure_state(self, r):
        """qc_shor.c:272-306 over the shards: the sequential cumulative sum is handed from rank to
        rank in index order, so the selected index is the one the unsharded scan would pick.
        Round 5: behind a compact circuit the scan runs on the companion register's shards -- the amplitudes the compact form
        leaves out are +0 and add nothing to the running sum, the compact order is the index order -- and the 16 * 2^n bytes
        of the real register are never written (the collapse replaces them)."""
        self.flush(keep_compact=True)
        if self._compact is not None:
            idx = self._measure_compact(float(r))
            if idx is not None:
                self._compact = None
                self.compact_measures += 1
                owner = idx >> self.n_local
                self.engine.collapse(self.shard, self.n_local, idx & ((1 << self.n_local) - 1) if owner == self.rank else -1)
                return idx
            self._expand_compact()               # (a hit in a padding column: the premise broke -- scan the expanded register)
        self._identity()
        last_excluded = self.num_states - 1
        msg = torch.zeros(3, dtype=torch.float64, device=self.device)
        cum, idx = 0.0, last_excluded
        for rk in range(self.world):
            if self.rank == rk:
                f, i, cum_out = self.engine.measure_scan(self.shard, self.n_local, rk << self.n_local, last_excluded, cum, float(r))
                msg[0], msg[1], msg[2] = float(f), float(i), cum_out        # indices < 2^53: exact in a double
            dist.broadcast(msg, src=rk if self.group is None else dist.get_global_rank(self.group, rk), group=self.group)
            m = msg.tolist()
            cum = m[2]
            if m[0] != 0.0:
                idx = int(m[1])
                break
        owner = idx >> self.n_local
        self.engine.collapse(self.shard, self.n_local, idx & ((1 << self.n_local) - 1) if owner == self.rank else -1)
        return idx

    def _measure_compact(self, r):
        """Q:283-292 on the companion register; the real index, or None when the compact premise does not hold"""
        comp = self._comp
        cb, orbit = self._compact
        M, n = self.M_size, self.num_qubits
        if r <= 0.0:
            return 0                             # the reference stops at index 0 whatever it holds
        last_excl = 1 << comp.num_qubits         # compact elements whose real index is below 2^n - 1
        if orbit[-1] == (1 << M) - 1:
            last_excl = (((1 << (n - M)) - 1) << cb) | (len(orbit) - 1)
        msg = torch.zeros(3, dtype=torch.float64, device=self.device)
        cum, cidx = 0.0, None
        for rk in range(self.world):
            if self.rank == rk:
                f, i, cum_out = self.engine.measure_scan(comp.shard, comp.n_local, rk << comp.n_local, last_excl, cum, r)
                msg[0], msg[1], msg[2] = float(f), float(i), cum_out
            dist.broadcast(msg, src=rk if self.group is None else dist.get_global_rank(self.group, rk), group=self.group)
            m = msg.tolist()
            cum = m[2]
            if m[0] != 0.0:
                cidx = int(m[1])
                break
        if cidx is None:
            return self.num_states - 1           # Q:283 fall-through
        col = cidx & ((1 << cb) - 1)
        if col >= len(orbit):
            return None
        return ((cidx >> cb) << M) | orbit[col]

    def local_numpy(self):
        """this rank's shard in the IDENTITY layout, as a numpy array of 2*2^n_local doubles"""
        self._identity()
        return self.shard.detach().cpu().numpy().copy()

    def gather(self):
        """whole state on every rank (tests only: 16 * 2^n bytes)"""
        self._identity()
        mine = self.shard.cpu() if self._host_staged else self.shard          # gloo gathers host tensors only
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        return torch.cat(parts).cpu().numpy()
