#!/usr/bin/env python3
"""Dry-run of the sharded H-sweep schedule at the real sizes (no GPU, no amplitudes): the ShardedRegister
runs with a recording engine and a stub torch.distributed, and a simple timeline model (compute stream +
communication stream, events as in sharded.flush) estimates the sweep time and the weak-scaling
efficiency.  Model inputs: per-gate HBM rate and per-link xGMI rate (defaults from this round's
measurements / the MI355X guide).  It checks the SCHEDULE (exchanges per sweep, what overlaps what) --
the numbers are estimates, not measurements.

usage: model_sharded.py [--n-local 30] [--sweeps 6] [--gate-gbs 6430] [--link-gbs 70]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantumcomputer_amd import sharded  # noqa: E402


class Work:
    def __init__(self, tl, done):
        self.tl, self.done = tl, done

    def wait(self):
        self.tl.compute = max(self.tl.compute, self.done)


class Timeline:
    """two in-order streams; a collective starts when both the data it reads is ready (compute stream position
    at enqueue time) and the comm stream is free"""

    def __init__(self, world, gate_gbs, link_gbs, n_local):
        self.compute = 0.0
        self.comm = 0.0
        self.world, self.gate_gbs, self.link_gbs, self.n_local = world, gate_gbs, link_gbs, n_local
        self.gate_ms = self.a2a_ms = 0.0
        self.launches = 0

    def kernel(self, nbits, frac=1.0):
        dt = 32.0 * (1 << nbits) * frac / (self.gate_gbs * 1e9) * 1e3 + 0.004
        self.compute += dt; self.gate_ms += dt; self.launches += 1

    def a2a(self, nbits, async_op):
        # each rank sends (W-1)/W of the slice, one chunk per peer, each over its own link
        per_link = 16.0 * (1 << nbits) / self.world
        dt = per_link / (self.link_gbs * 1e9) * 1e3 + 0.05
        start = max(self.compute, self.comm)
        self.comm = start + dt; self.a2a_ms += dt
        if not async_op:
            self.compute = self.comm
        return Work(self, self.comm)


class ModelEngine:
    def __init__(self, tl): self.tl = tl
    def reset(self, *a): pass
    def fill_random(self, *a): pass
    def hadamard(self, view, nbits, q): self.tl.kernel(nbits)
    def phase(self, view, nbits, mask, c, s): self.tl.kernel(nbits, 0.25)
    def camodc(self, view, nbits, *a): self.tl.kernel(nbits, 0.5)
    def swap_bits(self, src, dst, nbits, a, b): self.tl.kernel(nbits)
    def norm2(self, *a): return 1.0


class StubDist:
    def __init__(self, world, tl): self.world, self.tl = world, tl
    def get_world_size(self, group=None): return self.world
    def get_rank(self, group=None): return 0
    def all_to_all_single(self, dst, src, group=None, async_op=False):
        return self.tl.a2a(self.cur_bits, async_op)
    def pair_swap(self, keep, other, async_op=False):
        # the pairwise form (QCX_SHARD_EXCHANGE=pairwise): half of the slice over ONE link, both directions at once
        tl = self.tl
        dt = 16.0 * (1 << self.cur_bits) / 2 / (tl.link_gbs * 1e9) * 1e3 + 0.05
        start = max(tl.compute, tl.comm)
        tl.comm = start + dt; tl.a2a_ms += dt
        if not async_op:
            tl.compute = tl.comm
        return Work(tl, tl.comm)


def run(world, n_local, sweeps, gate_gbs, link_gbs, slices_log2, overlap):
    k = world.bit_length() - 1
    n = n_local + k
    tl = Timeline(world, gate_gbs, link_gbs, n_local)
    stub = StubDist(world, tl)
    real = sharded.dist
    sharded.dist = stub
    os.environ["QCX_SHARD_OVERLAP"] = "1" if overlap else "0"
    try:
        reg = sharded.ShardedRegister(n, 0, device="cpu", engine=ModelEngine(tl), slices_log2=slices_log2, dry_run=True)
        stub.cur_bits = reg.slice_bits
        per_sweep = []
        for s in range(sweeps):                     # flush per sweep: the look-ahead sees one sweep, like a
            t0 = max(tl.compute, tl.comm)           # driver that synchronises every step would
            for q in range(n):
                reg.hadamard_gate(q)
            if s == sweeps - 1 or True:
                reg.flush()
            per_sweep.append(max(tl.compute, tl.comm) - t0)
        return dict(world=world, n=n, sigma=reg.sigma, exchanges=reg.exchanges, packs=reg.pack_passes,
                    overlapped_gates=reg.overlapped_gates, per_sweep_ms=[round(x, 1) for x in per_sweep],
                    gate_ms=tl.gate_ms, a2a_ms=tl.a2a_ms)
    finally:
        sharded.dist = real


def c_path_striping_model(n_local, gate_gbs, link_gbs, gpus_on_node=8):
    """The one-process sharded register (csrc/qcx_sharded.inc.h): a trade is ONE pack+push kernel per GPU; a pair of shards
    moves S = 16 * 2^(n_local - k) bytes each way.  Without relays that rides the pair's one direct link.  With R relay GPUs
    (the GPUs of the node that hold no shard) the direct link carries the share a = (W-1)/(R+W-1) and each relay link
    1/(R+W-1) of each of the W-1 chunks; a relayed stripe costs two hops, done one after the other (push, then forward).
    MODEL, not a measurement: link_gbs per direction per link, no protocol overhead."""
    print("one-process sharded register, time of one trade (model):")
    for W in (2, 4, 8):
        k = W.bit_length() - 1
        S = 16.0 * 2.0 ** (n_local - k)
        direct_ms = S / (link_gbs * 1e9) * 1e3
        pack_ms = 32.0 * 2.0 ** n_local / (gate_gbs * 1e9) * 1e3           # the same pass run locally (HBM-bound floor)
        R = gpus_on_node - W
        line = f"  W={W}: chunk {S / 2**30:.1f} GiB per pair, direct link only {max(direct_ms, pack_ms):7.1f} ms"
        if R > 0:
            a = (W - 1) / (R + W - 1)
            hop_ms = a * S / (link_gbs * 1e9) * 1e3                         # every link out of a GPU carries a * S in phase A ...
            line += f"; {R} relays: direct share {a:.3f}, two phases of {hop_ms:6.1f} ms = {max(2 * hop_ms, pack_ms):7.1f} ms ({direct_ms / (2 * hop_ms):.2f}x)"
        else:
            line += "; no idle GPU to relay through (all links already busy)"
        print(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-local", type=int, default=30)
    ap.add_argument("--sweeps", type=int, default=6)
    ap.add_argument("--gate-gbs", type=float, default=6430.0)
    ap.add_argument("--link-gbs", type=float, default=70.0)
    ap.add_argument("--c-path", action="store_true", help="only the striping model of the one-process sharded register")
    a = ap.parse_args()
    if a.c_path:
        c_path_striping_model(a.n_local, a.gate_gbs, a.link_gbs)
        return
    base = None
    for world in (1, 2, 4, 8):
        for sl, ov in ((3, True), (0, False)):
            if world == 1 and not ov:
                continue
            r = run(world, a.n_local, a.sweeps, a.gate_gbs, a.link_gbs, sl, ov)
            steady = sum(r["per_sweep_ms"][1:]) / (a.sweeps - 1)
            upd = r["n"] * 2.0 ** r["n"] / (steady * 1e-3)
            if world == 1:
                base = upd
            print(f"W={world} n={r['n']} slices={1 << r['sigma']} overlap={ov}: sweeps {r['per_sweep_ms']} ms, "
                  f"exchanges {r['exchanges']}, packs {r['packs']}, gates in windows {r['overlapped_gates']}, "
                  f"steady {steady:.1f} ms/sweep -> {upd:.3e} upd/s, weak-scaling efficiency {upd / (world * base):.2f}")


if __name__ == "__main__":
    main()
