// qcx_kernels.h -- hand-written gfx950 (CDNA4) kernels for the gate-application
// path of qc_shor.c.  HBM-bound streaming kernels on the in-place amplitude
// vector; wave64 everywhere.  No matrix is built and no mat-vec is run: each
// kernel computes, per amplitude, exactly the sums the reference's COO
// mat-vec (Q:396-413) would have produced for that gate, in the same order
// and with the same roundings (compile with -ffp-contract=off).
//
// Bit-exactness notes (proved against the oracle in tests/):
//   reference output  = 0 + term0 (+ term1), term = (mr*cr)-(mi*ci) with mi = 0
//   kernel output     = (t0 (+|-) t1) + 0.0
// identical for every finite input, zero signs included: the only effect of
// the reference's "0 +" and "- 0*x" is to turn a -0 result into +0, which the
// trailing "+ 0.0" reproduces (it is not foldable under IEEE rules).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace qcx {

typedef double amp_t __attribute__((ext_vector_type(2)));   // (re, im): one 16-B global access

#define QCX_SQRT1_2 0.70710678118654752440   /* M_SQRT1_2, Q:210-213 */

__device__ __forceinline__ uint64_t insert_zero(uint64_t p, unsigned b)
{
    const uint64_t low = ((uint64_t)1 << b) - 1;
    return ((p & ~low) << 1) | (p & low);
}

template <bool NT> __device__ __forceinline__ amp_t ld(const amp_t *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(amp_t *p, amp_t v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// one Hadamard butterfly: a = amplitude with target bit 0, b = with target bit 1
__device__ __forceinline__ void h_butterfly(amp_t &a, amp_t &b)
{
    const double s = QCX_SQRT1_2;
    const double t0r = s * a.x, t0i = s * a.y, t1r = s * b.x, t1i = s * b.y;
    a.x = (t0r + t1r) + 0.0;  a.y = (t0i + t1i) + 0.0;
    b.x = (t0r - t1r) + 0.0;  b.y = (t0i - t1i) + 0.0;
}

// ---------------------------------------------------------------------------
// K1a  Hadamard, pair form, any target qubit.  Thread p owns the pair
// (i0, i0 | 2^q) with i0 = p with a zero inserted at bit q.  For q >= 6 both
// streams are 1-KiB-per-wave-instruction coalesced; below that the two loads of
// a wave interleave inside one 2-KiB window.  PPT pairs per thread are loaded
// before any is stored (2*PPT independent 16-B loads in flight per lane).
// ---------------------------------------------------------------------------
template <int PPT, bool NTL, bool NTS, bool WC, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_h_pair(amp_t *__restrict__ amp, unsigned q, uint64_t npairs,
                                                    unsigned glog, unsigned slog)
{
    const uint64_t low = ((uint64_t)1 << q) - 1, bit = (uint64_t)1 << q;
    const uint64_t step = (uint64_t)gridDim.x * (BLOCK * PPT);
    // WC: a wave's PPT loads are consecutive 64-pair runs (per-wave contiguous); otherwise the
    // block's waves interleave (k-th load of the block covers BLOCK consecutive pairs)
    const unsigned toff = WC ? ((threadIdx.x >> 6) * (64 * PPT) + (threadIdx.x & 63u)) : threadIdx.x;
    constexpr unsigned kstride = WC ? 64 : BLOCK;
    // block -> tile map: with slog > 0 the 2^glog tiles are dealt as 2^slog interleaved streams, so
    // blocks b, b + 2^slog, ... walk one contiguous region (slog = 3: one stream per XCD under the
    // round-robin block placement; speed only, any placement is correct)
    uint64_t tile0 = blockIdx.x;
    if (slog & 0xffu) {
        // streams; optional skew (slog >> 8): stream j walks its segment rotated by j * skew tiles, so that the
        // concurrently active windows are NOT a power-of-two distance apart
        const unsigned sl = slog & 0xffu, skew = slog >> 8;
        const uint64_t j = blockIdx.x & ((1u << sl) - 1u), pos = blockIdx.x >> sl, seg = (uint64_t)1 << (glog - sl);
        tile0 = (j << (glog - sl)) | ((pos + j * skew) & (seg - 1));
    }
    for (uint64_t base = tile0 * (BLOCK * PPT); base < npairs; base += step) {
        amp_t a[PPT], b[PPT];
        uint64_t i0[PPT];
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const uint64_t p = base + (uint64_t)k * kstride + toff;
            i0[k] = ((p & ~low) << 1) | (p & low);
            if (p < npairs) { a[k] = ld<NTL>(amp + i0[k]); b[k] = ld<NTL>(amp + i0[k] + bit); }
        }
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const uint64_t p = base + (uint64_t)k * kstride + toff;
            if (p < npairs) {
                h_butterfly(a[k], b[k]);
                st<NTS>(amp + i0[k], a[k]);
                st<NTS>(amp + i0[k] + bit, b[k]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K1b  Hadamard, wave-tile form for low target qubits (q < 6 + log2 R).
// A wave owns 64*R consecutive amplitudes as R registers x 64 lanes, every
// load/store a fully coalesced 1 KiB.  Partners sit in another lane (q < 6:
// wavefront xor-shuffle butterfly) or in another register of the same lane.
// ---------------------------------------------------------------------------
__device__ __forceinline__ amp_t shfl_xor_amp(amp_t v, int lane_mask)
{
    amp_t r;
    r.x = __shfl_xor(v.x, lane_mask, 64);
    r.y = __shfl_xor(v.y, lane_mask, 64);
    return r;
}

template <int Q, int R>
__device__ __forceinline__ void h_wave_tile(amp_t (&r)[R], unsigned lane)
{
    const double s = QCX_SQRT1_2;
    if constexpr (Q < 6) {
        const bool upper = (lane >> Q) & 1u;
#pragma unroll
        for (int k = 0; k < R; k++) {
            amp_t t; t.x = s * r[k].x; t.y = s * r[k].y;
            const amp_t o = shfl_xor_amp(t, 1 << Q);
            // lower lane: t + o ; upper lane: o - t   (o is the bit-0 partner's product)
            r[k].x = (upper ? (o.x - t.x) : (t.x + o.x)) + 0.0;
            r[k].y = (upper ? (o.y - t.y) : (t.y + o.y)) + 0.0;
        }
    } else {
        constexpr int D = 1 << (Q - 6);
#pragma unroll
        for (int k = 0; k < R; k++)
            if ((k & D) == 0) h_butterfly(r[k], r[k + D]);
    }
}

template <int Q, int R, bool NTL, bool NTS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_h_wave(amp_t *__restrict__ amp, uint64_t ntiles, unsigned glog, unsigned slog)
{
    // tile = 64*R amplitudes, one per wave per iteration; slog: stream-interleaved block order as in k_h_pair
    const unsigned lane = threadIdx.x & 63u;
    uint64_t blk = blockIdx.x;
    if (slog) blk = ((uint64_t)(blockIdx.x & ((1u << slog) - 1u)) << (glog - slog)) | (blockIdx.x >> slog);
    const uint64_t wave = (blk * BLOCK + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * BLOCK) >> 6;
    for (uint64_t t = wave; t < ntiles; t += nwaves) {
        amp_t *base = amp + t * (64 * R) + lane;
        amp_t r[R];
#pragma unroll
        for (int k = 0; k < R; k++) r[k] = ld<NTL>(base + k * 64);
        h_wave_tile<Q, R>(r, lane);
#pragma unroll
        for (int k = 0; k < R; k++) st<NTS>(base + k * 64, r[k]);
    }
}

// ---------------------------------------------------------------------------
// K2  controlled phase: multiply by (c + i s) every amplitude whose index has
// all NB mask bits set (b0 < b1).  Only that 1/2^NB of the vector is touched.
// ---------------------------------------------------------------------------
template <int NB, int APT, bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_phase(amp_t *__restrict__ amp, unsigned b0, unsigned b1,
                                                   double c, double s, uint64_t count, unsigned glog, unsigned slog)
{
    const uint64_t step = (uint64_t)gridDim.x * (BLOCK * APT);
    uint64_t tile0 = blockIdx.x;              // stream-interleaved tile order as in k_h_pair
    if (slog) tile0 = ((uint64_t)(blockIdx.x & ((1u << slog) - 1u)) << (glog - slog)) | (blockIdx.x >> slog);
    for (uint64_t base = tile0 * (BLOCK * APT); base < count; base += step) {
        amp_t v[APT];
        uint64_t idx[APT];
#pragma unroll
        for (int k = 0; k < APT; k++) {
            const uint64_t p = base + (uint64_t)k * BLOCK + threadIdx.x;
            uint64_t i = p;
            if (NB >= 1) i = insert_zero(i, b0) | ((uint64_t)1 << b0);
            if (NB >= 2) i = insert_zero(i, b1) | ((uint64_t)1 << b1);
            idx[k] = i;
            if (p < count) v[k] = ld<NT>(amp + i);
        }
#pragma unroll
        for (int k = 0; k < APT; k++) {
            const uint64_t p = base + (uint64_t)k * BLOCK + threadIdx.x;
            if (p < count) {
                amp_t o;
                o.x = ((c * v[k].x) - (s * v[k].y)) + 0.0;      // Q:409
                o.y = ((c * v[k].y) + (s * v[k].x)) + 0.0;      // Q:412
                st<NT>(amp + idx[k], o);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// K2b  controlled phase whose mask has a bit below 3, i.e. inside a 128-B line (8 amplitudes).  k_phase above gives
// every lane one TOUCHED amplitude, so for such masks a wave instruction reads 16 B out of every 32 or 64: twice to four
// times the requests for the same lines.  Here a lane owns one amplitude of every touched LINE (the mask bits >= 3
// are squeezed out of the numbering, the bits < 3 stay free): whole-line coalesced loads, the rotation only where the
// low mask bits are set, and stores of the rotated lanes only -- except at 16- or 32-B granularity (mask bit 0 or 1),
// where the whole line is stored back (untouched amplitudes with the bits they had): masked 32-B stores measured
// 4.0 ms against 2.6 ms at n = 30.
// NH = number of mask bits >= 3 (0, 1 or 2): h0 < h1.
// ---------------------------------------------------------------------------
template <int NH, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_phase_lines(amp_t *__restrict__ amp, unsigned h0, unsigned h1, unsigned lowmask, int store_all,
                                                         double c, double s, uint64_t count, unsigned glog, unsigned slog)
{
    const uint64_t step = (uint64_t)gridDim.x * BLOCK;
    uint64_t tile0 = blockIdx.x;
    if (slog) tile0 = ((uint64_t)(blockIdx.x & ((1u << slog) - 1u)) << (glog - slog)) | (blockIdx.x >> slog);
    for (uint64_t p = tile0 * BLOCK + threadIdx.x; p < count; p += step) {
        uint64_t i = p;
        if (NH >= 1) i = insert_zero(i, h0) | ((uint64_t)1 << h0);
        if (NH >= 2) i = insert_zero(i, h1) | ((uint64_t)1 << h1);
        const amp_t v = __builtin_nontemporal_load(amp + i);
        const bool hit = ((unsigned)i & lowmask) == lowmask;
        amp_t o = v;
        if (hit) {
            o.x = ((c * v.x) - (s * v.y)) + 0.0;      // Q:409
            o.y = ((c * v.y) + (s * v.x)) + 0.0;      // Q:412
        }
        if (hit || store_all) __builtin_nontemporal_store(o, amp + i);
    }
}

// ---------------------------------------------------------------------------
// K3  controlled modular multiply on the low M bits (Q:595-660).  Pure data
// movement: destination g (< C) of every 2^M-block with the control bit set
// receives the sum, in ascending source order, of the amplitudes at the
// sources f < C with (A*f) mod C == g.  With d = gcd(A, C) those are
//     f = f0 + t*(C/d),  t = 0..d-1,  f0 = ((g/d) * inv) mod (C/d),  d | g
// (inv = (A/d)^-1 mod C/d, computed on the host); for the usual coprime case
// d = 1 and the gate is a permutation.  A tile of 2^logT >= 2^M amplitudes is
// staged in LDS so the update is in place with coalesced global traffic; the
// amplitudes whose control bit is 0 are never touched (camodc_tile).
// ---------------------------------------------------------------------------

struct CamodcParams {
    unsigned M, logT;
    int      ctl;        // local bit index, or -1: control lives in the rank id and is 1
    unsigned C, d, Cd, inv;
    uint64_t ntiles;     // tiles to process (tiles hold control-set amplitudes only when ctl >= M)
    unsigned skip, ntl;  // k_camodc: do not read the lines above row C; store completely rewritten lines nontemporally
};

// base of tile tt.  A tile is 2^logT amplitudes that all have the control bit SET: a control at or
// above the M register is squeezed out of the tile numbering (whether it lies above the tile or inside it), so the
// control-clear half of the vector is never read.  Runs stay >= 2^M amplitudes (the control is never below M here).
__device__ __forceinline__ amp_t *camodc_tile(amp_t *amp, const CamodcParams &P, uint64_t tt)
{
    if (P.ctl >= (int)P.logT) {
        const unsigned b = (unsigned)P.ctl - P.logT;
        return amp + ((insert_zero(tt, b) | ((uint64_t)1 << b)) << P.logT);
    }
    if (P.ctl >= (int)P.M) return amp + (tt << (P.logT + 1));
    return amp + (tt << P.logT);
}
// offset of tile element e from camodc_tile(): with the control inside the tile, e with a 1 inserted at the control bit
__device__ __forceinline__ unsigned camodc_elem(bool squeeze, unsigned ctl, unsigned e)
{
    if (!squeeze) return e;
    const unsigned low = (1u << ctl) - 1u;
    return ((e & ~low) << 1) | (1u << ctl) | (e & low);
}

// (A variant that stored every amplitude of the tile back as whole nontemporal lines -- 32 B per amplitude instead of 16 + 16 * C / 2^M,
// no partial-line writes -- measured 2.9-3.4 against 2.8-2.9 ms in round 3 and was removed in round 5.)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_camodc(amp_t *__restrict__ amp, CamodcParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qcx_lds_raw[];
    amp_t *tile = reinterpret_cast<amp_t *>(qcx_lds_raw);
    const unsigned T = 1u << P.logT, blkmask = (1u << P.M) - 1u;
    const bool squeeze = P.ctl >= (int)P.M && P.ctl < (int)P.logT;
    // the permutation case (gcd(A, C) = 1): the source row of every destination row, tabulated once per workgroup -- one
    // 32-bit modulo per residue instead of two divisions and a modulo per amplitude (round 4: 2.85 -> 2.4 ms at n = 30 together
    // with the skipped lines)
    unsigned short *lut = reinterpret_cast<unsigned short *>(tile + T);
    const bool perm = P.d == 1u && P.M <= 12u;
    if (perm) for (unsigned f = threadIdx.x; f <= blkmask; f += BLOCK) lut[f] = (unsigned short)(f < P.C ? (f * P.inv) % P.C : f);
    for (uint64_t tt = blockIdx.x; tt < P.ntiles; tt += gridDim.x) {
        amp_t *g = camodc_tile(amp, P, tt);
        // (fill through registers with the DEFAULT cache policy, measured at n = 30: 2.9 ms per gate; an LDS-DMA fill 3.0-3.2 ms;
        // nontemporal loads 4.1-4.5 ms -- the moved elements are rewritten as PARTIAL lines a moment later and those
        // writes must still find their lines in L2)
        // Rows f >= C of a 2^M block are neither sources nor destinations (Q:631-634): the 128-B lines that hold nothing
        // else are not even read (round 4; C = 21, M = 5: one line in four).  lineC = C rounded up to whole lines.
        const unsigned lineC = P.skip ? ((P.C + 7u) & ~7u) : (1u << P.M);
        for (unsigned e = threadIdx.x; e < T; e += BLOCK)
            if ((e & blkmask) < lineC) tile[e] = g[camodc_elem(squeeze, (unsigned)P.ctl, e)];
        __syncthreads();
        const unsigned fullC = P.ntl ? (P.C & ~7u) : 0u;                 // rows below this sit in lines that are rewritten completely
        for (unsigned e = threadIdx.x; e < T; e += BLOCK) {
            const unsigned f = e & blkmask;
            bool on = true;
            if (P.ctl >= 0 && P.ctl < (int)P.M) on = (e >> P.ctl) & 1u;
            if (!on || f >= P.C) continue;                     // identity rows (Q:611-613, Q:631-634)
            amp_t acc; acc.x = 0.0; acc.y = 0.0;
            if (perm) { const amp_t sv = tile[(e - f) + lut[f]]; acc.x += sv.x; acc.y += sv.y; }
            else if (f % P.d == 0) {
                unsigned src = ((f / P.d) * P.inv) % P.Cd;            // < C^2 <= 2^32: 32-bit arithmetic is exact (host checks)
                const amp_t *blk = tile + (e - f);
                for (unsigned t = 0; t < P.d; t++, src += P.Cd) { acc.x += blk[src].x; acc.y += blk[src].y; }
            }
            // whole rewritten lines leave as nontemporal stores, the partly rewritten last line of a block through L2
            if (f < fullC && P.ctl >= (int)P.M) __builtin_nontemporal_store(acc, g + camodc_elem(squeeze, (unsigned)P.ctl, e));
            else g[camodc_elem(squeeze, (unsigned)P.ctl, e)] = acc;
        }
        __syncthreads();
    }
}

// (Round 5 tried the permutation case WITHOUT the LDS tile for blocks that fit a wave (M <= 6): lane l loads its amplitude, the
// destination lanes fetch theirs by shuffle, four tiles of 64 amplitudes per wave in flight, no barrier.  Slower: 2.52 ms per gate
// at n = 30 against 2.33 for k_camodc (2.42 when it writes whole lines with nontemporal traffic); the counters of k_camodc --
// 71 % of wave cycles waiting, traffic exactly the touched lines -- and this experiment say the same thing: the gate is bound by
// the memory side's handling of 3-lines-in-4 read-modify-write traffic, not by the workgroups' load -> barrier -> store chains.
// profiles/r05g_camodc_*.json, profiles/r05_camodc_wave_experiment.txt.  Removed again.)
// generic table form (C > 2^M, or 32-bit wrap in A*f): CSR of sources per
// destination low-bits value, built on the host exactly as Q:619-647 maps them.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_camodc_table(amp_t *__restrict__ amp, CamodcParams P,
                                                          const uint32_t *__restrict__ off,
                                                          const uint32_t *__restrict__ srcs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qcx_lds_raw[];
    amp_t *tile = reinterpret_cast<amp_t *>(qcx_lds_raw);
    const unsigned T = 1u << P.logT, blkmask = (1u << P.M) - 1u;
    const bool squeeze = P.ctl >= (int)P.M && P.ctl < (int)P.logT;
    for (uint64_t tt = blockIdx.x; tt < P.ntiles; tt += gridDim.x) {
        amp_t *g = camodc_tile(amp, P, tt);
        for (unsigned e = threadIdx.x; e < T; e += BLOCK) tile[e] = g[camodc_elem(squeeze, (unsigned)P.ctl, e)];
        __syncthreads();
        for (unsigned e = threadIdx.x; e < T; e += BLOCK) {
            const unsigned f = e & blkmask;
            // (a control inside the M register is encoded in the table: the host passes ctl = -1 then)
            const amp_t *blk = tile + (e - f);
            amp_t acc; acc.x = 0.0; acc.y = 0.0;
            for (uint32_t k = off[f]; k < off[f + 1]; k++) { acc.x += blk[srcs[k]].x; acc.y += blk[srcs[k]].y; }
            g[camodc_elem(squeeze, (unsigned)P.ctl, e)] = acc;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K0 helpers
// ---------------------------------------------------------------------------
__global__ void k_set_one(amp_t *amp, uint64_t index)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { amp_t v; v.x = 1.0; v.y = 0.0; amp[index] = v; }
}

// synthetic input: counter-based pseudo-random amplitudes, component k of the whole vector is
//   ((splitmix64(seed + k) >> 11) * 2^-53 - 0.5) * scale      (k = 2*index + {0: re, 1: im})
// every operation exact or singly rounded, so the CPU twin in oracle/ gives the same bits and a
// window of a 16-GiB state can be checked without ever holding it on the host.
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_fill_random(amp_t *__restrict__ amp, uint64_t count, uint64_t first_global,
                                                     uint64_t seed, double scale)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        const uint64_t k = 2 * (first_global + i);
        amp_t v;
        v.x = ((double)(splitmix64(seed + k) >> 11) * 0x1p-53 - 0.5) * scale;
        v.y = ((double)(splitmix64(seed + k + 1) >> 11) * 0x1p-53 - 0.5) * scale;
        amp[i] = v;
    }
}

// ---------------------------------------------------------------------------
// K0c  canonical zeros.  The reference's mat-vec rewrites EVERY amplitude at every gate as 0 + 1.0 * x (+ ...) (Q:393-413),
// which turns a -0 component into +0 even where the gate is the identity.  The gate kernels only touch the amplitudes a
// gate acts on, so they rely on an invariant instead: the state holds no -0 whenever a gate runs.  Reset, collapse, the
// synthetic fill and every gate kernel keep it (each result ends in "+ 0.0"); a state WRITTEN by the caller
// (qcx_state_write / qcx_state_load) may break it, and is passed through this kernel once before the next gate.
// Reads everything, stores only the amplitudes that change.
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// K9  STRICT gates for states that hold non-finite (or overflow-prone) amplitudes (round 5).  The gate kernels above give the
// reference's bits for every FINITE state while skipping what the reference's mat-vec spends on nothing: the products with
// the matrix entries' zero imaginary parts, and the rows whose entry is 1.  With an Inf or NaN in the state those are not
// nothing: 0 * Inf = NaN poisons the other component of the SAME amplitude, through an identity row too (Q:409-412 runs
// over every stored triplet).  A register that was handed such values (qcx_state_write / _load; the host sets
// qcx_register::nonfinite from a scan of what was written) runs every gate through these kernels instead -- one plain pass
// per gate over ALL amplitudes with the mat-vec's own four products and sums per triplet (Q:396-413 applied to
// the triplets each gate stores, Q:456-481 / Q:529-562 / Q:612-657) -- until a reset, a fill or a measurement replaces the state.  z and one arrive as
// kernel arguments so that no compiler ever folds a product with them.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_strict_h(amp_t *__restrict__ amp, unsigned n, unsigned q, double s, double z)
{
    const uint64_t half = (uint64_t)1 << (n - 1), bitq = (uint64_t)1 << q, low = bitq - 1;
    for (uint64_t p = (uint64_t)blockIdx.x * 256u + threadIdx.x; p < half; p += (uint64_t)gridDim.x * 256u) {
        const uint64_t i0 = ((p & ~low) << 1) | (p & low), i1 = i0 | bitq;
        const amp_t a = amp[i0], b = amp[i1];
        double lo_r = 0.0, lo_i = 0.0, hi_r = 0.0, hi_i = 0.0;
        lo_r += (s * a.x) - (z * a.y);    lo_i += (s * a.y) + (z * a.x);        // row i0, column i0
        lo_r += (s * b.x) - (z * b.y);    lo_i += (s * b.y) + (z * b.x);        // row i0, column i1
        hi_r += (s * a.x) - (z * a.y);    hi_i += (s * a.y) + (z * a.x);        // row i1, column i0
        hi_r += (-s * b.x) - (z * b.y);   hi_i += (-s * b.y) + (z * b.x);       // row i1, column i1
        amp_t lo, hi; lo.x = lo_r; lo.y = lo_i; hi.x = hi_r; hi.y = hi_i;
        amp[i0] = lo; amp[i1] = hi;
    }
}

__global__ __launch_bounds__(256) void k_strict_phase(amp_t *__restrict__ amp, uint64_t count, uint64_t both, double er, double ei, double one, double z)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256u) {
        const amp_t v = amp[i];
        double nr = 0.0, ni = 0.0;
        if ((i & both) == both) { nr += (er * v.x) - (ei * v.y);   ni += (er * v.y) + (ei * v.x); }
        else                    { nr += (one * v.x) - (z * v.y);   ni += (one * v.y) + (z * v.x); }
        amp_t o; o.x = nr; o.y = ni;
        amp[i] = o;
    }
}

// one workgroup per 2^M-block (M <= 12: the block and the destination of every row sit in LDS); destination g sums its sources
// in ascending row order, every row of the block is written (the identity rows too: their entry 1 is a triplet like any other)
__global__ __launch_bounds__(256) void k_strict_camodc(amp_t *__restrict__ amp, unsigned n, unsigned M, unsigned C, unsigned A, int ctl, double one, double z)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qcx_lds_raw[];
    amp_t *blk = reinterpret_cast<amp_t *>(qcx_lds_raw);
    const unsigned B = 1u << M;
    unsigned short *dst_on = reinterpret_cast<unsigned short *>(blk + B);         // destination of row f when its control reads 1
    for (unsigned f = threadIdx.x; f < B; f += 256u) dst_on[f] = (unsigned short)(f < C ? ((unsigned)(A * f) % C) & (B - 1u) : f);   // Q:645-647
    const uint64_t nblk = (uint64_t)1 << (n - M);
    for (uint64_t b = blockIdx.x; b < nblk; b += gridDim.x) {
        amp_t *g = amp + (b << M);
        __syncthreads();
        for (unsigned f = threadIdx.x; f < B; f += 256u) blk[f] = g[f];
        __syncthreads();
        const int on = (ctl >= (int)M) ? (int)(((b << M) >> ctl) & 1u) : -1;      // -1: the control is a bit of the row
        for (unsigned d = threadIdx.x; d < B; d += 256u) {
            double ar = 0.0, ai = 0.0;
            for (unsigned f = 0; f < B; f++) {
                const bool cbit = on >= 0 ? on != 0 : ((f >> ctl) & 1u) != 0;
                const unsigned to = cbit ? dst_on[f] : f;
                if (to == d) { const amp_t v = blk[f]; ar += (one * v.x) - (z * v.y); ai += (one * v.y) + (z * v.x); }
            }
            amp_t o; o.x = ar; o.y = ai;
            g[d] = o;
        }
    }
}

// does [amp, amp + count) hold a component that is not finite, or so large (>= 2^500) that a circuit could overflow from it?
__global__ __launch_bounds__(256) void k_scan_nonfinite(const amp_t *__restrict__ amp, uint64_t count, unsigned *flag)
{
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256u) {
        const amp_t v = amp[i];
        bad |= !(fabs(v.x) < 0x1p500) || !(fabs(v.y) < 0x1p500);
    }
    if (__ballot(bad) && (threadIdx.x & 63u) == 0u) atomicOr(flag, 1u);
}

__global__ __launch_bounds__(256) void k_canon_zeros(amp_t *__restrict__ amp, uint64_t count)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        amp_t v = __builtin_nontemporal_load(amp + i);
        const bool neg0 = (uint64_t)__double_as_longlong(v.x) == 0x8000000000000000ULL || (uint64_t)__double_as_longlong(v.y) == 0x8000000000000000ULL;
        if (neg0) { v.x += 0.0; v.y += 0.0; amp[i] = v; }
    }
}

// ---------------------------------------------------------------------------
// K5  total probability.  Deterministic two-stage tree (fixed grid), not the
// reference's sequential order (T:28-37) -- a check value, compared with a
// tolerance.
// ---------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_norm_partial(const amp_t *__restrict__ amp, uint64_t count, double *partial)
{
    __shared__ double red[BLOCK / 64];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < count; i += (uint64_t)gridDim.x * BLOCK) {
        const amp_t v = amp[i];
        acc += v.x * v.x + v.y * v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < BLOCK / 64; w++) t += red[w];
        partial[blockIdx.x] = t;
    }
}

__global__ void k_norm_final(const double *partial, unsigned nparts, double *out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (unsigned i = 0; i < nparts; i++) t += partial[i];
        *out = t;
    }
}

// ---------------------------------------------------------------------------
// K4  measurement scan, exact form: the reference's strictly sequential
// cumulative sum (Q:283-292) carried by ONE wave.  Lanes fetch 64 amplitudes
// at a time (coalesced) and the 64 additions are chained in index order; lane
// k ends each round holding the running sum through element k, so the first
// lane with cum >= r is the reference's answer bit for bit.
// ---------------------------------------------------------------------------
struct MeasureOut {
    int      found;
    uint64_t index;
    double   cum;
    unsigned stats[2];          // K4c: [records looked at closely, records] of the scan (qcx_measure_last_stats): one copy back with the result
};

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

__global__ __launch_bounds__(64) void k_measure_scan(const amp_t *__restrict__ amp, uint64_t count,
                                                     double cum_in, double r, MeasureOut *out)
{
    const unsigned lane = threadIdx.x;
    double cum = cum_in;
    amp_t nxt; nxt.x = 0.0; nxt.y = 0.0;
    if (lane < count) nxt = amp[lane];
    for (uint64_t base = 0; base < count; base += 64) {
        const amp_t v = nxt;
        const uint64_t ni = base + 64 + lane;
        nxt.x = 0.0; nxt.y = 0.0;
        if (ni < count) nxt = amp[ni];                       // prefetch the next round
        const double p = v.x * v.x + v.y * v.y;              // gsl_complex_abs2 (Q:286); 0 past the end
        double run = cum;
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const double pj = readlane_f64(p, j);
            run = run + ((int)lane >= j ? pj : 0.0);         // + 0.0 leaves a non-negative sum unchanged
        }
        const bool hit = (base + lane < count) && (run >= r);
        const unsigned long long m = __ballot(hit);
        if (m) {
            const int first = __builtin_ctzll(m);
            if ((int)lane == first) { out->found = 1; out->index = base + lane; out->cum = run; }
            return;
        }
        cum = readlane_f64(run, 63);
    }
    if (lane == 0) { out->found = 0; out->index = 0; out->cum = cum; }
}

// ---------------------------------------------------------------------------
// K4  exact AND parallel measurement scan: the arithmetic fact K4c below rests on.
//
// The reference's cumulative sum cum_i = fl(cum_{i-1} + p_i) is order dependent, so a parallel
// prefix sum does not give the same decisions.  But while the running sum stays inside one binade
// [2^e, 2^(e+1)) it is an integer K (53 bits) times the fixed ulp u = 2^(e-52), and adding p >= 0
// rounds to   K + floor(p/u) + [frac(p/u) > 1/2]        (a tie, frac == 1/2, rounds to even).
// So for a block of amplitudes with no tie and no element >= 2^e, the sequential sum over the block is
// exactly K + S with S = sum_i (floor(p_i/u) + [frac > 1/2]) -- an ORDINARY integer sum -- provided
// K + S < 2^53 (no binade crossing).  A block whose binade guess holds costs a few integer operations; any
// other block (binade crossing, wrong guess, tie, start of the sum, the block in which cum reaches r) is redone
// with a strictly sequential scan.  The result is the reference's index, bit for bit, for every input; only the
// speed depends on the data.  (Round 3's two-read form of this, K4b -- tree block sums for the guesses, then the
// increments, then a one-wave chain over the records: 7 ms at n = 30 -- was removed in round 5; K4c reads once.)
// ---------------------------------------------------------------------------
// amplitudes per record = 2^blog, a launch parameter (8..11; host: meas_block_log): small registers want small records
enum : uint32_t { MEAS_ALLZERO = 1u << 16, MEAS_TIE = 1u << 17, MEAS_BIG = 1u << 18, MEAS_EUNK = 1u << 19 };
// MEAS_RHIT (round 5): the APPROXIMATE running sum reaches r inside this record (with a margin) -- a hint for k_meas_fast's
// list of records to look at closely; it decides nothing
enum : uint32_t { MEAS_RHIT = 1u << 20 };

struct MeasBlock {
    uint64_t S;        // integer increment of the block in units of the assumed ulp (saturated)
    uint32_t meta;     // low 11 bits: assumed biased exponent of the running sum at block start; flags above
    uint32_t pad;
};

__device__ __forceinline__ double prob_of(amp_t v) { return v.x * v.x + v.y * v.y; }    // gsl_complex_abs2

// ---------------------------------------------------------------------------
// K4c  measurement scan, exact and parallel, in ONE read of the state (round 4).
//
//   1. k_meas_onepass   a workgroup of four waves takes four consecutive RECORDS of 2^RLOG amplitudes (one per wave, kept
//                       as |amp|^2 in registers), publishes the (tree) sum of the four, obtains the approximate sum of
//                       everything before it by a DECOUPLED LOOK-BACK, and every wave computes the integer increment S of
//                       its record under the binade that prefix falls in -- one pass over HBM.  Workgroup ids are handed
//                       out by an atomic ticket, so a workgroup only ever waits for workgroups that are already running.
//                       The XCDs' L2s are not coherent, so every poll is a round trip to the memory side, and ~1300
//                       workgroups run at once: polling their slots one by one costs more traffic than it hides (measured:
//                       +0.45 ms per 16 GiB with 64 slots per poll, +2.3 ms with 512).  The look-back therefore has two
//                       levels: the last member of every group of 64 workgroups publishes the group's sum as soon as it has seen
//                       the other 63, and a later workgroup reads (a) the slots of the earlier members of its own group and
//                       (b) ONE entry per earlier group -- its sum, or the running total through it.
//                       The look-back only feeds the GUESS: the order of the approximate additions may differ from run to
//                       run; the walk below validates every guess against the exact running sum, so the measured index
//                       never depends on it;
//   2. k_meas_groups    sums of 64, 64^2, ... consecutive records (a few tiny launches);
//   3. k_meas_fast      (round 5, further down) the EVENTS of the scan from a list of candidate records the groups launch wrote:
//                       8 waves, stretch sums and record data prepared in parallel, ~1 us of serial work per event;
//   4. k_meas_walk      the fallback behind it, and the whole of step 3 in round 4:
//                       one wave descends that tree with the EXACT running sum: 64 entries per step, whole groups at a
//                       time while they are plain (integer additions inside one binade are associative); only a record with
//                       a tie, an oversized element, a wrong guess, a binade crossing or the crossing of r is redone: as one
//                       integer addition per 1024 amplitudes where that holds, per 64 below that, and the strictly
//                       sequential chain only for the 64 amplitudes in which something happens.
// ---------------------------------------------------------------------------
typedef unsigned long long meas_slot_t;      // bits of a double; all ones = not published yet

struct MeasLookback {
    meas_slot_t *agg, *incl;                 // per workgroup: its sum / the running total through it
    meas_slot_t *gsum, *gincl;               // per group of 64 workgroups: the group's sum (published by its last member) / the running total through it
    unsigned    *ticket;
};

__device__ __forceinline__ uint64_t meas_inc(uint64_t b, int e, uint32_t &flags)    // b = bits of p > 0; increment of the running sum in ulps of binade e
{
    // branch-free (the callers unroll this 32 times): p/u = mp * 2^-sh, rounded to nearest; a tie (fraction exactly 1/2) is
    // flagged, as is an element at or above the binade (sh <= 0).  sh = 53: quotient 0, remainder mp against half = 2^52;
    // sh >= 54: half > mp, the increment is 0 (the shift is clamped to 63 for the far-away exponents).
    const int ex = (int)((b >> 52) & 0x7ff);
    const uint64_t mp = (b & 0xfffffffffffffULL) | (ex ? (uint64_t)1 << 52 : 0);       // subnormal: no implicit bit
    const int sh = e - (ex ? ex : 1);
    const bool big = sh <= 0;
    const unsigned shc = (unsigned)(sh < 1 ? 1 : (sh > 63 ? 63 : sh));
    const uint64_t rem = mp & (((uint64_t)1 << shc) - 1), half = (uint64_t)1 << (shc - 1);
    const uint64_t inc = (mp >> shc) + (rem > half ? 1u : 0u);
    flags |= big ? (uint32_t)MEAS_BIG : (rem == half ? (uint32_t)MEAS_TIE : 0u);
    return big ? (uint64_t)1 << 54 : inc;
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return readlane_f64(v, 0);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), lane) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(v & 0xffffffffu), lane);
}
__device__ __forceinline__ meas_slot_t meas_poll(const meas_slot_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void meas_post(meas_slot_t *p, double v) { __hip_atomic_store(p, (meas_slot_t)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one look-back window: lane l holds the slot at distance l (vi = a running total, or all ones; va = a plain sum, or all
// ones; lanes at or beyond `lanes` hold nothing).  1: a running total was found and `run` is complete; 0: all `lanes`
// sums consumed, go on further back; -1: a slot before the first running total is still unpublished
__device__ __forceinline__ int meas_window(unsigned lane, unsigned lanes, meas_slot_t vi, meas_slot_t va, double &run)
{
    const bool in = lane < lanes;
    const bool has_i = in && vi != ~0ULL, has_a = has_i || (in && va != ~0ULL);
    const unsigned long long mi = __ballot(has_i), ma = __ballot(has_a);
    const unsigned first = mi ? (unsigned)__builtin_ctzll(mi) : lanes;          // slots before it must have published their sums
    const unsigned long long need = first >= 64 ? ~0ULL : (((unsigned long long)1 << first) - 1);
    if ((ma & need) != need) return -1;
    double c = 0.0;
    if (lane < first) c = __longlong_as_double((long long)va);
    else if (lane == first && mi) c = __longlong_as_double((long long)vi);
    run += wave_sum_f64(c);
    return mi ? 1 : 0;
}

template <int RLOG>
__global__ __launch_bounds__(256) void k_meas_onepass(const amp_t *__restrict__ amp, uint64_t count, double cum_in,
                                                       MeasLookback LB, MeasBlock *out, unsigned spin_limit, unsigned dbg, double r)
{
    constexpr unsigned PER = (1u << RLOG) / 64u;            // amplitudes per lane: 4 .. 32
    __shared__ unsigned s_blk;
    __shared__ double red[4];
    __shared__ uint64_t s_pref;
    if (threadIdx.x == 0) s_blk = atomicAdd(LB.ticket, 1u);
    __syncthreads();
    const unsigned blk = s_blk, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t rec = (uint64_t)blk * 4u + wave;
    const uint64_t base = rec << RLOG;
    double p[PER];
    // loads in batches of 8 (8 x 16 B in flight per lane); the record's probabilities stay in registers
    constexpr unsigned BATCH = PER < 8u ? PER : 8u;
    if (((uint64_t)(blk + 1u) << (RLOG + 2)) <= count) {            // (all workgroups but the last: no per-load bounds test)
#pragma unroll
        for (unsigned kb = 0; kb < PER; kb += BATCH) {
            amp_t v[BATCH];
#pragma unroll
            for (unsigned g = 0; g < BATCH; g++) v[g] = __builtin_nontemporal_load(amp + base + (uint64_t)(kb + g) * 64u + lane);
#pragma unroll
            for (unsigned g = 0; g < BATCH; g++) p[kb + g] = prob_of(v[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (unsigned k = 0; k < PER; k++) {            // (unrolled too: a run-time index would put p[] into scratch memory)
            const uint64_t i = base + (uint64_t)k * 64u + lane;
            double pk = 0.0;
            if (i < count) pk = prob_of(amp[i]);
            p[k] = pk;
        }
    }
    double acc = 0.0;
#pragma unroll
    for (unsigned k = 0; k < PER; k++) acc += p[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        double t = (red[0] + red[1]) + (red[2] + red[3]);
        if (!(t < __builtin_inf())) t = __builtin_inf();          // NaN / Inf input: every later guess is "unknown" (exact path)
        const unsigned g = blk >> 6, q = blk & 63u;
        if (lane == 0) meas_post(&LB.agg[blk], t);
        double run = 0.0;
        int state = (dbg & 1u) ? 1 : 0;                            // dbg bit 0 (diagnostics): no look-back, every guess from cum_in alone
        if (state) run = cum_in;
        unsigned spins = 0;
        // (a) the earlier members of this workgroup's own group, nearest first.  The LAST member of a group (q = 63) waits for
        // all 63 sums whether or not a running total turns up among them: it publishes the group's sum (no atomics: float
        // atomic adds from every workgroup plus a release-ordered count cost 6 ms per 16 GiB when tried)
        while (state == 0 && q > 0) {
            meas_slot_t vi = ~0ULL, va = ~0ULL;
            if (lane < q) { vi = meas_poll(&LB.incl[blk - 1u - lane]); if (vi == ~0ULL || q == 63u) va = meas_poll(&LB.agg[blk - 1u - lane]); }
            bool ready = true;
            if (q == 63u) {
                ready = __ballot(lane < q && va == ~0ULL) == 0ULL;
                if (ready) {
                    double gs = wave_sum_f64(lane < q ? __longlong_as_double((long long)va) : 0.0) + t;
                    if (!(gs < __builtin_inf())) gs = __builtin_inf();
                    if (lane == 0) meas_post(&LB.gsum[g], gs);
                }
            }
            const int w = ready ? meas_window(lane, q, vi, va, run) : -1;
            if (w >= 0) { state = w; break; }
            if (++spins > spin_limit) { state = -1; break; }
            __builtin_amdgcn_s_sleep(32);
        }
        // (b) whole groups before it, nearest first; "group -1" holds cum_in
        int64_t gj = (int64_t)g - 1;
        while (state == 0) {
            const int64_t gi = gj - (int64_t)lane;
            meas_slot_t vi = ~0ULL, va = ~0ULL;
            unsigned lanes = 64;
            if (gj < 63) lanes = (unsigned)(gj + 2);               // groups gj .. 0, then the virtual one
            if (gi == -1) vi = (meas_slot_t)__double_as_longlong(cum_in);
            else if (gi >= 0) { vi = meas_poll(&LB.gincl[gi]); if (vi == ~0ULL) va = meas_poll(&LB.gsum[gi]); }
            const int w = meas_window(lane, lanes, vi, va, run);
            if (w > 0) { state = 1; break; }
            if (w == 0) { gj -= 64; continue; }
            if (++spins > spin_limit) { state = -1; break; }       // (never seen; a bound, so that nothing can hang)
            __builtin_amdgcn_s_sleep(32);
        }
        if (state < 0) run = __builtin_inf();
        if (lane == 0) {
            double through = run + t;
            if (!(through < __builtin_inf())) through = __builtin_inf();
            meas_post(&LB.incl[blk], through);
            if (q == 63u) meas_post(&LB.gincl[g], through);
            s_pref = (uint64_t)__double_as_longlong(run);
        }
    }
    __syncthreads();
    double pref = __longlong_as_double((long long)s_pref);
    for (unsigned w = 0; w < wave; w++) pref += red[w];           // the waves before this one in the workgroup
    const int e = (int)(((uint64_t)__double_as_longlong(pref) >> 52) & 0x7ff);
    uint64_t S = 0;
    uint32_t flags = (e == 0 || e == 0x7ff) ? (uint32_t)MEAS_EUNK : 0u;
    if (pref <= r * (1.0 + 1e-9) && pref + red[wave] >= r * (1.0 - 1e-9)) flags |= (uint32_t)MEAS_RHIT;
    bool nonzero = false;
    const uint64_t SAT = (uint64_t)1 << 54;
#pragma unroll
    for (unsigned k = 0; k < PER; k++) {
        const uint64_t b = (uint64_t)__double_as_longlong(p[k]);
        uint32_t f1 = 0;
        const uint64_t inc = meas_inc(b, e, f1);
        nonzero |= b != 0;
        S += b ? inc : 0;
        flags |= b ? f1 : 0u;
        if (S > SAT) S = SAT;
    }
    const unsigned long long any_nz = __ballot(nonzero);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        S += (uint64_t)__shfl_down((unsigned long long)S, o, 64);
        flags |= (uint32_t)__shfl_down((int)flags, o, 64);
    }
    if (lane == 0) {
        if (S > SAT) S = SAT;
        MeasBlock mb; mb.S = S; mb.meta = (uint32_t)e | (flags & 0x7fff0000u) | (any_nz ? 0u : (uint32_t)MEAS_ALLZERO); mb.pad = 0;
        out[rec] = mb;
    }
}

// one level up: record g = 64 consecutive records of the level below (one wave each).  A group is PLAIN when all its
// non-zero members are unflagged and were computed under the same binade; anything else sets MEAS_EUNK = "descend".
// cands (the launch over the records themselves only): the records k_meas_fast has to look at closely -- flagged ones, and both
// sides of every change of the binade guess (a crossing lies in one of the two)
#define QCX_MEAS_CAND_CAP 192u
struct MeasCands { unsigned count, pad; unsigned ticket[4]; unsigned list[QCX_MEAS_CAND_CAP]; };    // ticket: the look-back's workgroup counter (a fixed address whatever the scan's size)

__global__ __launch_bounds__(64) void k_meas_groups(const MeasBlock *__restrict__ in, unsigned nin, MeasBlock *__restrict__ outg, MeasCands *cands,
                                                    meas_slot_t *look, unsigned nslots)
{
    const unsigned lane = threadIdx.x, i = blockIdx.x * 64u + lane;
    // (the launch over the records also puts the look-back's slots back to "not published" for the next scan -- k_meas_onepass is
    //  through with them: one memset launch less per measurement.  The slots only feed guesses: a stale one costs time, never a result)
    if (look) {
        const unsigned per = (nslots + gridDim.x - 1u) / gridDim.x;
        for (unsigned k = lane; k < per; k += 64u) { const unsigned idx = blockIdx.x * per + k; if (idx < nslots) look[idx] = ~0ULL; }
    }
    MeasBlock m; m.S = 0; m.meta = MEAS_ALLZERO; m.pad = 0;
    if (i < nin) m = in[i];
    const bool zero = (m.meta & MEAS_ALLZERO) != 0;
    if (cands) {
        const int e_me = (int)(m.meta & 0x7ffu);
        int e_next = __shfl_down(e_me, 1, 64), e_prev = __shfl_up(e_me, 1, 64);
        if (lane == 63u && i + 1u < nin) e_next = (int)(in[i + 1u].meta & 0x7ffu);
        if (lane == 0u) e_prev = i > 0u ? (int)(in[i - 1u].meta & 0x7ffu) : e_me;
        if (i + 1u >= nin) e_next = e_me;
        const bool c = i < nin && ((!zero && (m.meta & (MEAS_TIE | MEAS_BIG | MEAS_EUNK | MEAS_RHIT))) || e_next != e_me || e_prev != e_me);
        const unsigned long long cm = __ballot(c);
        if (cm) {
            unsigned base = 0;
            if (lane == 0u) base = atomicAdd(&cands->count, (unsigned)__builtin_popcountll(cm));
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
            const unsigned slot = base + (unsigned)__builtin_popcountll(cm & (((unsigned long long)1 << lane) - 1ULL));
            if (c && slot < QCX_MEAS_CAND_CAP) cands->list[slot] = i;
        }
    }
    const unsigned long long nz = __ballot(!zero);
    MeasBlock g; g.S = 0; g.meta = MEAS_ALLZERO; g.pad = 0;
    if (nz) {
        const int e_ref = __builtin_amdgcn_readlane((int)(m.meta & 0x7ffu), __builtin_ctzll(nz));
        const bool bad = !zero && ((m.meta & (MEAS_TIE | MEAS_BIG | MEAS_EUNK)) || (int)(m.meta & 0x7ffu) != e_ref);
        unsigned long long tot = zero ? 0ULL : (unsigned long long)m.S;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, 64);
        tot = readlane_u64(tot, 0);
        if (tot > ((uint64_t)1 << 54)) tot = (uint64_t)1 << 54;
        g.S = tot;
        g.meta = (uint32_t)e_ref | (__ballot(bad) ? (uint32_t)MEAS_EUNK : 0u);
    }
    if (lane == 0) outg[blockIdx.x] = g;
}

// exact scan of one record.  Batches of 1024 amplitudes (16 per lane, the next batch's loads in flight meanwhile): when a
// whole batch has no tie / oversized element and stays inside the binade and below r, it is ONE integer addition (one
// wave reduction).  Otherwise the longest prefix of its 16 groups of 64 for which that holds is found by bisection (the sums
// are monotone), the one group in which something happens runs the strictly sequential chain, and the rest of the batch
// starts over under the binade the chain ended in.
__device__ __forceinline__ bool meas_int_value(double cum, double r, unsigned long long inc, uint32_t fl, double &cn)
{
    // all lanes: inc = this lane's integer increments under the binade of cum, fl = its flags.  true: cn = the exact running
    // sum after them (every addition stayed inside the binade and below r)
    const uint64_t cb = (uint64_t)__double_as_longlong(cum);
    const int ec = (int)((cb >> 52) & 0x7ff);
    if (ec == 0 || ec == 0x7ff) return false;
    if (__ballot(fl != 0) != 0ULL) return false;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) inc += __shfl_down(inc, o, 64);
    const uint64_t tot = readlane_u64(inc, 0);                                  // <= 1024 increments < 2^53 each: no overflow
    const uint64_t Kn = ((cb & 0xfffffffffffffULL) | ((uint64_t)1 << 52)) + tot;
    if (Kn >= ((uint64_t)1 << 53)) return false;
    cn = __longlong_as_double((long long)(((uint64_t)ec << 52) | (Kn & 0xfffffffffffffULL)));
    return !(cn >= r);
}

__device__ __forceinline__ bool wave_exact_block(const amp_t *__restrict__ amp, uint64_t first, uint64_t len,
                                                 double &cum, double r, uint64_t *hit_index, double *hit_cum)
{
    const unsigned lane = threadIdx.x & 63u;
    constexpr int G = 16;                                   // groups of 64 per batch
    double nx[G];
#pragma unroll
    for (int g = 0; g < G; g++) { const uint64_t o = (uint64_t)g * 64 + lane; nx[g] = 0.0; if (o < len) nx[g] = prob_of(amp[first + o]); }
    for (uint64_t base0 = 0; base0 < len; base0 += 64 * G) {
        double pv[G];
#pragma unroll
        for (int g = 0; g < G; g++) pv[g] = nx[g];
#pragma unroll
        for (int g = 0; g < G; g++) {                       // prefetch the next batch
            const uint64_t o = base0 + 64 * G + (uint64_t)g * 64 + lane;
            nx[g] = 0.0;
            if (o < len) nx[g] = prob_of(amp[first + o]);
        }
        int a = 0;                                          // groups [0, a) of the batch are done
        const int gmax = (int)min((uint64_t)G, (len - base0 + 63) / 64);
        while (a < gmax) {
            // this lane's increments of the groups still to do, under the binade the running sum is in now
            const int ec = (int)(((uint64_t)__double_as_longlong(cum) >> 52) & 0x7ff);
            uint64_t inc[G]; uint32_t fl[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const uint64_t pb = (uint64_t)__double_as_longlong(pv[g]);
                uint32_t f1 = 0;
                const uint64_t i1 = meas_inc(pb, ec, f1);
                inc[g] = pb ? i1 : 0; fl[g] = pb ? f1 : 0u;
            }
            auto value = [&](int lo, int hi, double &cn) {  // the groups [lo, hi) as one integer addition
                unsigned long long sI = 0; uint32_t sF = 0;
#pragma unroll
                for (int g = 0; g < G; g++) { const bool in = g >= lo && g < hi; sI += in ? inc[g] : 0; sF |= in ? fl[g] : 0u; }
                return meas_int_value(cum, r, sI, sF, cn);
            };
            double cn = cum;
            int take = a;                                   // groups [a, take) can be taken at once
            if (!(cum >= r)) {
                if (value(a, gmax, cn)) take = gmax;
                else {
                    int lo = a, hi = gmax - 1;              // the longest good prefix ends in [lo, hi]
                    double cbest = cum;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        double c2;
                        if (value(a, mid, c2)) { lo = mid; cbest = c2; } else hi = mid - 1;
                    }
                    take = lo; cn = cbest;
                }
            }
            if (take > a) { cum = cn; a = take; }
            if (a >= gmax) break;
            // group a: the strictly sequential chain
            double p = pv[0];                               // pv[a] without a run-time register index
#pragma unroll
            for (int q = 1; q < G; q++) p = (q == a) ? pv[q] : p;
            const uint64_t base = base0 + (uint64_t)a * 64;
            double run = cum;
#pragma unroll
            for (int jj = 0; jj < 64; jj++) {
                const double pj = readlane_f64(p, jj);
                run = run + ((int)lane >= jj ? pj : 0.0);
            }
            const bool hit = (base + lane < len) && (run >= r);
            const unsigned long long m = __ballot(hit);
            if (m) {
                const int firstl = __builtin_ctzll(m);
                *hit_index = first + base + (uint64_t)firstl;
                *hit_cum = readlane_f64(run, firstl);
                return true;
            }
            cum = readlane_f64(run, 63);
            a++;
        }
    }
    return false;
}

struct MeasLevels {
    const MeasBlock *lv[5];          // lv[0]: the records, lv[L]: sums of 64^L consecutive records
    unsigned n[5];
    int top;                         // highest level present
};

// (Round 5 staged the upper levels of the tree in LDS for the walk: no change, 3.71 against 3.72 ms on a dense n = 30 state --
// the ~30 us per event are spent in the exact rescans of the record the event lies in and in the steps' arithmetic on ONE wave,
// not in the latency of the level reads.  Not kept.)
// resume: what k_meas_fast (below) left -- nothing to do (state 0), or "go on at record b with the exact running sum cum"; nullptr:
// the whole scan is this kernel's
struct MeasResume { uint32_t state, slow; uint64_t b; double cum; };

__global__ __launch_bounds__(64) void k_meas_walk(const amp_t *__restrict__ amp, uint64_t count, MeasLevels T,
                                                  double cum_in, double r, MeasureOut *out, unsigned *stats, unsigned rlog, const MeasResume *resume,
                                                  MeasCands *cands, meas_slot_t *look, unsigned nslots)
{
    const unsigned lane = threadIdx.x;
    const unsigned n0 = T.n[0];
    if (look) for (unsigned k = lane; k < nslots; k += 64u) look[k] = ~0ULL;      // (a scan without a groups launch: at most 34 slots)
    // the last kernel of a scan leaves the look-back's ticket and the candidate count at zero for the next scan (two memset
    // launches less per measurement: ~9 us of the ~100 an n = 20 attempt takes; the host falls back to memsets after a failed call)
    if (lane == 0) { cands->count = 0u; cands->ticket[0] = cands->ticket[1] = cands->ticket[2] = cands->ticket[3] = 0u; }
    double cum = cum_in;
    unsigned slow = 0;
    uint64_t b = 0;                                   // next record
    if (resume) {
        if (resume->state == 0u) return;
        b = resume->b; cum = resume->cum; slow = resume->slow;
    }
    int cap = T.top;                                  // highest level the next step may use
    uint64_t hi = 0; double hc = 0.0;
    // r already reached before the first addition (Q:289 with cum_0 >= cum_in): the first examined element is the answer
    if (b == 0 && cum_in >= r) {
        const uint64_t len = min((uint64_t)1 << rlog, count);
        slow++;
        if (wave_exact_block(amp, 0, len, cum, r, &hi, &hc)) {
            if (lane == 0) { out->found = 1; out->index = hi; out->cum = hc; if (stats) { stats[0] = slow; stats[1] = n0; } }
            return;
        }
        b = 1;
    }
    while (b < n0) {
        int L = 0;
        while (L < cap && (b & (((uint64_t)1 << (6 * (L + 1))) - 1)) == 0) L++;       // the highest level b is aligned to
        const MeasBlock *rec = T.lv[L];
        const unsigned nL = T.n[L];
        const unsigned shift = 6u * (unsigned)L;
        const uint64_t i0 = b >> shift;
        const unsigned lim = L == T.top ? 64u : 64u - (unsigned)(i0 & 63u);       // up to the next boundary of the level above
        MeasBlock mine; mine.S = 0; mine.meta = MEAS_ALLZERO; mine.pad = 0;
        if (lane < lim && i0 + lane < nL) mine = rec[i0 + lane];
        const uint64_t cb = (uint64_t)__double_as_longlong(cum);
        const int ec = (int)((cb >> 52) & 0x7ff);
        const bool ecok = ec != 0 && ec != 0x7ff;
        const bool zero = (mine.meta & MEAS_ALLZERO) != 0;
        const bool plain = zero || (ecok && !(mine.meta & (MEAS_TIE | MEAS_BIG | MEAS_EUNK)) && (int)(mine.meta & 0x7ffu) == ec);
        // inclusive sums of the increments over the lanes (saturated entries stay below 2^54 each: no overflow in 64 of them)
        unsigned long long inc = zero ? 0ULL : (unsigned long long)mine.S;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += up;
        }
        const uint64_t Kn = ((cb & 0xfffffffffffffULL) | ((uint64_t)1 << 52)) + inc;
        bool ok = plain && lane < lim;
        double cn = cum;                                 // the exact running sum through this lane's entry, if ok
        if (ok && (!zero || inc != 0)) {
            // (an all-zero entry adds nothing: its value is the one through the last non-zero lane before it)
            ok = ecok && Kn < ((uint64_t)1 << 53);
            if (ok) { cn = __longlong_as_double((long long)(((uint64_t)ec << 52) | (Kn & 0xfffffffffffffULL))); ok = !(cn >= r); }
        }
        const unsigned long long bad = ~__ballot(ok);
        const unsigned adv = bad ? (unsigned)__builtin_ctzll(bad) : 64u;        // entries taken in this step (monotone sums: a good lane validates all before it)
        if (adv) { cum = readlane_f64(cn, (int)adv - 1); b += (uint64_t)adv << shift; }
        if (adv >= lim) { cap = T.top; continue; }
        if (L > 0) { cap = L - 1; continue; }                                    // the entry at b needs a closer look
        // level 0: record b the slow way
        slow++;
        const uint64_t first = b << rlog;
        if (first >= count) break;
        const uint64_t len = min((uint64_t)1 << rlog, count - first);
        if (wave_exact_block(amp, first, len, cum, r, &hi, &hc)) {
            if (lane == 0) { out->found = 1; out->index = hi; out->cum = hc; if (stats) { stats[0] = slow; stats[1] = n0; } }
            return;
        }
        b++;
        cap = T.top;
    }
    if (lane == 0) { out->found = 0; out->index = 0; out->cum = cum; if (stats) { stats[0] = slow; stats[1] = n0; } }
}

// ---------------------------------------------------------------------------
// K4c, step 3a (round 5): the events of the scan WITHOUT the one-wave tree walk.  On a dense state the walk above spends ~34 us per
// record it has to redo (a binade crossing every time the running sum doubles: 32 of them at n = 30, 1.1 ms next to a 2.8 ms read)
// -- the descent to the record and back up, a dependent load per level, and the redo itself, all on one wave.  But WHERE the
// events are is known beforehand from the approximate prefixes: k_meas_groups lists the flagged records and both sides of
// every change of the binade guess (MeasCands).  One workgroup of 8 waves takes the list in index order, candidate k on wave
// k mod 8, in two phases:
//   parallel (no dependence on the running sum): the integer sum of the stretch of plain records between candidate k-1 and
//     candidate k -- straight from the tree: at most 63 entries at either end per level, every address known up front, all
//     loads in flight together -- and the candidate's own amplitudes: lane l keeps 2^rlog / 64 CONSECUTIVE probabilities in
//     registers, with their integer increments under the stretch's binade and the next one;
//   serial (wave k waits for wave k-1's exact running sum in LDS): validate the stretch against it (same binade, no overflow,
//     below r), then the record: a prefix scan of the lanes' increments finds the first lane in which something happens
//     (crossing, tie, oversized element, r); THAT lane adds its few probabilities one by one -- the reference's own
//     additions, no shuffles -- and the lanes behind it go on under the binade it ended in.  About a microsecond.
// Every decision is validated against the exact running sum exactly as in the walk, so the index is the reference's for any
// input; whatever does not go as the list predicts (a stretch that fails, more than QCX_MEAS_CAND_CAP candidates, r reached
// before the first addition) is handed to k_meas_walk with the exact state (MeasResume) -- slower, same result.
// ---------------------------------------------------------------------------
struct MeasStretch { uint64_t tot; int es; bool ok; };          // es = -1: nothing but zeros

__device__ __forceinline__ void meas_stretch_take(const MeasBlock &m, bool valid, MeasStretch &st)
{
    const bool zero = !valid || (m.meta & MEAS_ALLZERO) != 0;
    const unsigned long long nz = __ballot(!zero);
    if (!nz) return;
    if (st.es < 0) st.es = __builtin_amdgcn_readlane((int)(m.meta & 0x7ffu), __builtin_ctzll(nz));
    const bool bad = !zero && ((m.meta & (MEAS_TIE | MEAS_BIG | MEAS_EUNK)) || (int)(m.meta & 0x7ffu) != st.es);
    if (__ballot(bad)) st.ok = false;
    unsigned long long t = zero ? 0ULL : (unsigned long long)m.S;                  // <= 2^54 each
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
    st.tot += readlane_u64(t, 0);                                                  // <= 10 x 64 entries: no overflow
}

// the records [a, b) as ONE integer addition, from the tree (every load issued before the first is used)
__device__ __forceinline__ MeasStretch meas_stretch(const MeasLevels &T, uint64_t a, uint64_t b)
{
    const unsigned lane = threadIdx.x & 63u;
    MeasBlock mL[5], mR[5];
    bool vL[5], vR[5];
    uint64_t ua = a, ub = b;
#pragma unroll
    for (int L = 0; L < 5; L++) {
        vL[L] = vR[L] = false;
        mL[L].S = mR[L].S = 0; mL[L].meta = mR[L].meta = MEAS_ALLZERO; mL[L].pad = mR[L].pad = 0;
        if (L > T.top || ua >= ub) continue;
        if (L == T.top) {                               // (at most 64 entries up here)
            vL[L] = lane < ub - ua;
            if (vL[L]) mL[L] = T.lv[L][ua + lane];
            ua = ub;
            continue;
        }
        const uint64_t lb = min(ub, (ua + 63u) & ~(uint64_t)63);          // up to the next group boundary
        vL[L] = lane < lb - ua;
        if (vL[L]) mL[L] = T.lv[L][ua + lane];
        if (lb < ub) {
            const uint64_t ra = ub & ~(uint64_t)63;                         // >= lb: the whole groups in between belong to the level above
            vR[L] = lane < ub - ra;
            if (vR[L]) mR[L] = T.lv[L][ra + lane];
            ua = lb >> 6; ub = ra >> 6;
        } else ua = ub;
    }
    MeasStretch st; st.tot = 0; st.es = -1; st.ok = true;
#pragma unroll
    for (int L = 0; L < 5; L++) { meas_stretch_take(mL[L], vL[L], st); meas_stretch_take(mR[L], vR[L], st); }
    return st;
}

// this lane's `per` consecutive probabilities as one integer increment under binade e (saturated), and their flags
__device__ __forceinline__ void meas_lane_inc(const double (&p)[32], unsigned per, int e, uint64_t &S, uint32_t &fl)
{
    S = 0; fl = 0;
#pragma unroll
    for (unsigned j = 0; j < 32; j++) {
        if (j < per) {
            const uint64_t pb = (uint64_t)__double_as_longlong(p[j]);
            uint32_t f1 = 0;
            const uint64_t i1 = meas_inc(pb, e, f1);
            S += pb ? i1 : 0; fl |= pb ? f1 : 0u;
        }
    }
    if (S > ((uint64_t)1 << 54)) S = (uint64_t)1 << 54;
}

// one record, exactly, from the exact running sum cum (< r on entry).  true: r was reached at *hit_index; false: cum = the running sum behind it
__device__ __forceinline__ bool meas_record_exact(const double (&p)[32], unsigned per, uint64_t first, double &cum, double r,
                                                  int eA, uint64_t SA, uint32_t flA, int eB, uint64_t SB, uint32_t flB,
                                                  uint64_t *hit_index, double *hit_cum)
{
    const unsigned lane = threadIdx.x & 63u;
    unsigned done = 0;                                   // lanes [0, done) are behind us
    while (done < 64u) {
        const uint64_t cb = (uint64_t)__double_as_longlong(cum);
        const int ec = (int)((cb >> 52) & 0x7ff);
        const bool ecok = ec != 0 && ec != 0x7ff;
        uint64_t S = 0; uint32_t fl = MEAS_EUNK;
        if (ecok) {
            if (ec == eA) { S = SA; fl = flA; }
            else if (ec == eB) { S = SB; fl = flB; }
            else meas_lane_inc(p, per, ec, S, fl);
        }
        const bool in = lane >= done;
        unsigned long long inc = in ? (unsigned long long)S : 0ULL;           // <= 2^54 each: no overflow over 64 lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long up = __shfl_up(inc, o, 64);
            if ((int)lane >= o) inc += up;
        }
        const unsigned long long flagged = __ballot(in && fl != 0u);
        const unsigned firstfl = flagged ? (unsigned)__builtin_ctzll(flagged) : 64u;
        const uint64_t Kn = ((cb & 0xfffffffffffffULL) | ((uint64_t)1 << 52)) + inc;
        bool ok = in && ecok && lane < firstfl && Kn < ((uint64_t)1 << 53);
        double cn = cum;
        if (ok) { cn = __longlong_as_double((long long)(((uint64_t)ec << 52) | (Kn & 0xfffffffffffffULL))); ok = !(cn >= r); }
        const unsigned long long bad = ~__ballot(ok) & ~(done ? (((unsigned long long)1 << done) - 1ULL) : 0ULL);
        const unsigned L = bad ? (unsigned)__builtin_ctzll(bad) : 64u;       // monotone sums: a good lane validates all before it
        if (L > done) cum = readlane_f64(cn, (int)L - 1);
        if (L >= 64u) break;
        // lane L: the reference's own additions, one by one (every lane runs them on its own numbers; lane L's count)
        double run = cum, hc = 0.0;
        int hj = -1;
#pragma unroll
        for (unsigned j = 0; j < 32; j++) {
            if (j < per) {
                run = run + p[j];
                if (hj < 0 && run >= r) { hj = (int)j; hc = run; }
            }
        }
        const int hjL = __builtin_amdgcn_readlane(hj, (int)L);
        if (hjL >= 0) {
            *hit_index = first + (uint64_t)L * per + (uint64_t)hjL;
            *hit_cum = readlane_f64(hc, (int)L);
            return true;
        }
        cum = readlane_f64(run, (int)L);
        done = L + 1u;
    }
    return false;
}

__global__ __launch_bounds__(512) void k_meas_fast(const amp_t *__restrict__ amp, uint64_t count, MeasLevels T, double cum_in, double r,
                                                   MeasureOut *out, unsigned *stats, unsigned rlog, const MeasCands *cands,
                                                   MeasResume *resume, unsigned dbg)
{
    __shared__ unsigned s_raw[QCX_MEAS_CAND_CAP], s_sorted[QCX_MEAS_CAND_CAP + 1];
    __shared__ unsigned long long s_cum[QCX_MEAS_CAND_CAP + 1];
    __shared__ unsigned s_flag[QCX_MEAS_CAND_CAP + 1];
    __shared__ unsigned s_stop;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const unsigned n0 = T.n[0], ncand = cands->count;
    if (ncand > QCX_MEAS_CAND_CAP || cum_in >= r) {                 // (uniform) not this kernel's case: the walk takes everything
        if (threadIdx.x == 0) { resume->state = 1u; resume->slow = 0u; resume->b = 0; resume->cum = cum_in; }
        return;
    }
    for (unsigned t = threadIdx.x; t < ncand; t += blockDim.x) s_raw[t] = cands->list[t];
    for (unsigned t = threadIdx.x; t <= ncand; t += blockDim.x) s_flag[t] = 0u;
    if (threadIdx.x == 0) s_stop = 0u;
    __syncthreads();
    for (unsigned t = threadIdx.x; t < ncand; t += blockDim.x) {   // rank sort (distinct values)
        const unsigned v = s_raw[t];
        unsigned rank = 0;
        for (unsigned j = 0; j < ncand; j++) rank += s_raw[j] < v ? 1u : 0u;
        s_sorted[rank] = v;
    }
    __syncthreads();
    const unsigned per = (1u << rlog) >> 6;                           // 4 .. 32 probabilities per lane
    // (no workgroup barrier below: waves leave at different times)
    for (unsigned k = wave; k <= ncand; k += nwaves) {                // k = ncand: the stretch behind the last candidate
        const uint64_t a = k ? (uint64_t)s_sorted[k - 1u] + 1u : 0u;
        const bool has_rec = k < ncand;
        const uint64_t b = has_rec ? (uint64_t)s_sorted[k] : (uint64_t)n0;
        // ---- parallel phase -------------------------------------------------------------------------------------------------
        const MeasStretch st = meas_stretch(T, a, b);
        double p[32];
        const uint64_t first = b << rlog;
        int eA = -1, eB = -1;
        uint64_t SA = 0, SB = 0; uint32_t flA = 0, flB = 0;
        if (has_rec) {
#pragma unroll
            for (unsigned j = 0; j < 32; j++) {
                p[j] = 0.0;
                if (j < per) { const uint64_t i = first + (uint64_t)lane * per + j; if (i < count) p[j] = prob_of(amp[i]); }
            }
            eA = st.es >= 0 ? st.es : (int)(T.lv[0][b].meta & 0x7ffu);     // the binade the record most likely starts in
            if (eA > 0 && eA < 0x7fe) {
                eB = eA + 1;
                meas_lane_inc(p, per, eA, SA, flA);
                meas_lane_inc(p, per, eB, SB, flB);
            } else eA = -1;
        } else {
#pragma unroll
            for (unsigned j = 0; j < 32; j++) p[j] = 0.0;
        }
        // ---- serial phase: the exact running sum behind candidate k - 1 ------------------------------------------------------
        double cum = cum_in;
        if (k) {
            unsigned f = 0;
            while (true) {
                f = __hip_atomic_load(&s_flag[k - 1u], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (f || __hip_atomic_load(&s_stop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!f) return;                                           // somebody before us finished the scan (or handed it over)
            cum = __longlong_as_double((long long)__hip_atomic_load(&s_cum[k - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        }
        bool fail = (dbg & 2u) && k == 2u;                            // (diagnostics: hand over at the third candidate)
        // the stretch [a, b): plain records under the binade of cum, no overflow, below r
        if (!fail && st.es >= 0) {
            const uint64_t cb = (uint64_t)__double_as_longlong(cum);
            const int ec = (int)((cb >> 52) & 0x7ff);
            const uint64_t Kn = ((cb & 0xfffffffffffffULL) | ((uint64_t)1 << 52)) + st.tot;
            if (!st.ok || ec == 0 || ec == 0x7ff || ec != st.es || Kn >= ((uint64_t)1 << 53)) fail = true;
            else {
                const double cn = __longlong_as_double((long long)(((uint64_t)ec << 52) | (Kn & 0xfffffffffffffULL)));
                if (cn >= r) fail = true; else cum = cn;
            }
        } else if (!fail && !st.ok) fail = true;
        if (fail) {                                                   // the walk goes on from the start of the stretch
            if (lane == 0) {
                resume->state = 1u; resume->slow = k; resume->b = a; resume->cum = cum;
                __hip_atomic_store(&s_stop, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            return;
        }
        if (has_rec) {
            uint64_t hi = 0; double hc = 0.0;
            if (meas_record_exact(p, per, first, cum, r, eA, SA, flA, eB, SB, flB, &hi, &hc)) {
                if (lane == 0) {
                    out->found = 1; out->index = hi; out->cum = hc;
                    if (stats) { stats[0] = k + 1u; stats[1] = n0; }
                    resume->state = 0u;
                    __hip_atomic_store(&s_stop, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                return;
            }
            if (lane == 0) {
                __hip_atomic_store(&s_cum[k], (unsigned long long)__double_as_longlong(cum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&s_flag[k], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else if (lane == 0) {                                       // the end of the records: r was never reached
            out->found = 0; out->index = 0; out->cum = cum;
            if (stats) { stats[0] = ncand; stats[1] = n0; }
            resume->state = 0u;
        }
    }
}

// ---------------------------------------------------------------------------
// K6  fused multi-gate pass.  A workgroup stages a TILE of 2^T amplitudes in
// LDS -- the c lowest index bits (contiguous runs of 2^c amplitudes = 16*2^c bytes) plus nh = T - c
// arbitrary higher bits hbit[0..nh) -- applies a whole list of gates to it and writes it back: one
// HBM round trip for the lot.  Every gate performs exactly the arithmetic of its stand-alone
// kernel, in the order the gates were issued, so results are bit-identical to the unfused path:
//   H on a tile bit           butterfly between LDS slots (same t0 +/- t1 + 0.0 form)
//   controlled phase          any control/target: a bit outside the tile is a per-tile constant (the gate
//                             is skipped or applies to the whole tile); consecutive phases stay in registers
//   controlled modular mult.  when the M register is inside the tile (closed form of k_camodc)
// ---------------------------------------------------------------------------
enum : uint32_t { FUSE_H = 0, FUSE_PHASE = 1, FUSE_CAMODC = 2 };

struct FuseOp {                 // 32 bytes
    uint32_t type;
    uint32_t a;                 // H: tile-local bit;  CAMODC: M
    uint64_t mask;              // PHASE: global index bits that must all be 1
    double   c, s;              // PHASE: cos, sin.   CAMODC: reinterpreted below
};
struct FuseCamExtra {           // overlays c, s of a CAMODC op (16 bytes)
    uint32_t C, d, Cd, inv;
};
struct FuseSeg { uint8_t src, dst, len, pad; };
#define QCX_MAX_SEG 16
struct FusePass {
    uint32_t nops, T, c, nh;
    int32_t  cam_ctl_local[4];  // [0]: 1 = ROUNDS form; [1]: bytes of folded-multiply tables, [2]: their record offset in ops,
                                // [3]: byte offset of the table area behind the lut in LDS
    uint8_t  hbit[16];          // global bit carried by tile-local bit c + j, ascending
    uint32_t xm_off, xm_cnt;    // LDS copy of the records' outside-tile masks: byte offset behind the lut, entries (0 = none)
    uint32_t has_cam, dbg;      // the pass holds modular multiplies (selects the kernel variant); dbg: diagnostics only (tools/experiments/probe_pass.py):
                                // bit 0 skip the gates, bit 1 skip the stores, bit 2 skip the tile fill -- results are then wrong by design
    uint32_t dg_cnt, dg_rec_off;// tolerance mode: merged diagonals of the pass (0 = none), record offset of their table area in ops
    uint32_t dg_lds_off, dg_slim;// byte offset of their LDS area behind the lut; 1: every round of the pass is a fast round; 2: radix-8 fast rounds
    double   tol_scale;         // radix-8 passes: M_SQRT1_2 ^ (Hadamards of the pass), applied once when a tile is stored
    // General tile addressing (round 4).  A pass reads the tile whose local bit j is PHYSICAL index bit in_pos[j] of the input
    // buffer (ascending in j: in_pos[j] = j for the first c) and stores it where the pass's output layout puts those bits --
    // in place on the identity layout (the default: out = in), or OUT OF PLACE into the second buffer under another
    // logical -> physical map (a CHAINED pass: the host chooses the map so that the next pass reads whole contiguous tiles and
    // only this pass's stores are gathered; the last pass of a chain stores the identity layout again).  The tile NUMBER's
    // bits are spread over the free positions of the three index spaces by run-length segments: input, output, and LOGICAL
    // (the gate records test control qubits outside the tile against the logical base index).
    uint8_t  chained, nseg_in, nseg_out, nseg_lg;
    uint8_t  in_pos[16];        // physical (input) index bit of tile-local bit j
    uint8_t  st_loc[16];        // store order: the j-th lowest OUTPUT position of the tile's bits belongs to tile-local bit st_loc[j] ...
    uint8_t  st_pos[16];        // ... and is output index bit st_pos[j] (ascending in j)
    FuseSeg  seg_in[QCX_MAX_SEG], seg_out[QCX_MAX_SEG], seg_lg[QCX_MAX_SEG];
    // gen = 1: the pass does not READ its tiles: the register is a basis state that has not been written yet (lazy reset /
    // collapse), and the tile -- that state after the closed-form circuit front, see GenFront -- is generated in LDS
    uint32_t gen, gen_rec_off, gen_lds_off;
    uint32_t gen_tab_bytes;     // gen = 3: bytes of k_gen_cols' residue -> column table in LDS (at gen_lds_off behind the columns)
    // zskip = W > 0 (W = log2 of the waves per workgroup): the wave number is mapped onto the tile-local bits zb[0..W) -- bits no
    // Hadamard of the pass targets and no multiply moves -- instead of onto the highest thread bits, and a wave whose
    // amplitudes are ALL +0 skips the rounds (gates map +0 to +0: same bits).  The host asks for it behind a circuit front:
    // there only the residues of the multiply ladder's orbit are populated among the 2^M low index values (C = 21, a = 2:
    // six of 32), so whole waves of a tile hold nothing.  Correct for any state: a wave that finds a non-zero bit works.
    uint8_t  zskip, zb[3];
    uint32_t zpad;
    uint64_t zlist;             // the zb[] ascending, one per byte
    // xp_on (round 5): the LAST pass of a compact chain stores the REAL register (k_fused_x8 only): its output index is the virtual
    // register's [L-register bits][column], cb column bits; the amplitude of L-part l and column j belongs at real index
    // (l << M) | orbit[j], every other amplitude of the 2^M-block is +0 -- the pass writes all 2^M of them, 2^(M - cb) times its
    // tile, and k_expand_compact (one more read and write of the compact form) does not run.  The expanded store order: bits
    // 0 .. M-1 of a store index = the M-register value f, the bits above = the tile's store-order bits cb .. T-1;
    // xp_pos[i] / xp_loc[i] (i >= M) = REAL index bit / tile-local bit of store-index bit i; xp_colloc[b] = tile-local bit of column bit b
    uint8_t  xp_on, xp_M, xp_cb, xp_ncols;
    uint8_t  xp_pos[24], xp_loc[24], xp_colloc[4];
    uint16_t xp_orbit[16];
};

// Which tile a workgroup takes (round 5, second half).  Workgroup slot b (blockIdx, or the b-th turn of a persistent workgroup) used
// to take tile b; now the 2^s low bits of b -- under round-robin placement the XCD the workgroup runs on, for s >= 3 -- are moved
// to bit position `pos` of the tile number: t = [b's bits above s + pos][b mod 2^s][the `pos` bits of b above s].  s = bits 12-15 of
// FusePass::dbg, pos + 1 = bits 16-20 (0: on top, the per-gate kernels' "interleaved streams" of h_plan); launch_pass sets both
// (fuse_streams_log2 / fuse_streams_pos).  Why: a chained memory-bound pass stores runs of 2^c amplitudes (128 B at c = 3), and the
// tiles whose runs are NEIGHBOURS in the output are consecutive tile numbers (planner: tiles numbered by output position).  With
// t = b, eight neighbouring runs leave from eight different XCDs, i.e. through eight different L2s; with s = 3, pos = 3 one XCD
// takes the eight tiles whose runs make up 1 KiB of the output, back to back.  n = 30 fused Hadamard sweep 19.1-19.7 -> 18.0-18.4 ms
// (same box each pair), n = 28 tolerance inverse QFT 5.02 -> 4.94; the FP64-bound exact walk does not care.  The gain swings with
// pos (pos = 2, 4: none or a loss; 1, 3, 5: the gain) and needs s >= 3 (profiles/r05_streams.txt; the microbenchmark behind it,
// a pass without gates: tools/experiments/tile_shell.hip, profiles/r05_tile_shell.txt).  Which tile a workgroup takes changes
// nothing about what it does to it: same bits.
__device__ __forceinline__ uint64_t fuse_stream_tile(uint64_t b, unsigned tiles_log2, unsigned slog, unsigned pos1)
{
    const unsigned pos = pos1 ? min(pos1 - 1u, tiles_log2 - slog) : tiles_log2 - slog;
    const uint64_t r = b >> slog, sb = b & ((1u << slog) - 1u);
    return (r & (((uint64_t)1 << pos) - 1u)) | (sb << pos) | ((r >> pos) << (pos + slog));
}

// The circuit front on a basis state (K0b, BasisFront below) evaluated per TILE of the first pass behind it (round 4): the
// front's separate write pass and the first pass's read disappear.  Same closed form, same bits: after the front, the
// 2^M-block of an amplitude index i (its bits >= M) is populated iff its non-Hadamard bits equal the basis state's, and then
// holds ONE amplitude +/- v, at the residue  f = f0 * prod over the set control bits of A_g  mod C  (one modulus for the whole
// ladder, f0 < C: the products commute; anything else takes the separate write pass).  Per tile: the controls outside the
// tile give a factor from <= 5 byte-indexed tables, the 2^h combinations of the tile-local bits >= M a table built once per
// workgroup; an element then costs one LDS look-up and a compare.
struct GenFront {
    uint64_t basis;
    uint64_t fixed_out, sign_out;           // outside-tile parts of BasisFront::fixed_mask / sign_mask
    double   v;
    uint32_t M, ncam, C, f0;                // C = 0: no multiplies (every populated block holds f0)
    uint32_t cmpmask;                       // the low-M bits that must equal the residue (Hadamards on M-register bits free them)
    uint32_t lowout_mask;                   // low-M bits that lie outside the tile (taken from the tile's base index)
    uint32_t sfm, sbv;                      // slot space: which slot bits are fixed, and to what
    uint32_t h, present;                    // slot bits; which of the five byte tables hold a control
    uint8_t  slotbit[16], lowbit[16], signbit[16];      // per tile-local bit: slot bit index / low-M bit position (0xff: none) / in the sign mask
    uint8_t  camloc[64];                    // per multiply: slot bit of its control, 0xff = outside the tile (in the byte tables)
    uint32_t camA[64];
    uint16_t tabP[5][256];                  // product of the A_g whose control is a set bit of byte f of the base index, mod C
    // the COMPACT form (FusePass::gen = 3, k_gen_cols on the virtual register of a compact chain): the pass's index space is
    // [L-register bits][column number], cb column bits; column j holds the amplitudes whose M register reads orbit[j]
    uint32_t cb, ncols, sgn_slots;          // sgn_slots: which of the tile's hot bits lie in the sign mask
    uint32_t Cinv;                          // floor(2^32 / C): x mod C without a division (k_gen_cols; x < 2^32)
    uint16_t orbit[16];                     // the populated M-register values, ascending
};

// per-thread / per-k constant of the generated fill: slot bits | low bits << 12 | sign parity << 24 of a tile-local element index
__device__ __forceinline__ uint32_t gen_pack(unsigned e, const GenFront *G, unsigned T)
{
    uint32_t w = 0;
    for (unsigned j = 0; j < T; j++) {
        if (!((e >> j) & 1u)) continue;
        if (G->slotbit[j] != 0xff) w |= 1u << G->slotbit[j];
        if (G->lowbit[j] != 0xff) w |= 1u << (12u + G->lowbit[j]);
        if (G->signbit[j]) w ^= 1u << 24;
    }
    return w;
}

// once per workgroup: the slot table (residue factor of the tile-local controls, times f0; 0xffff = block not populated)
template <int BLOCK>
__device__ __forceinline__ void gen_setup(const GenFront *G, unsigned short *phot)
{
    const unsigned nslot = 1u << G->h;
    for (unsigned s = threadIdx.x; s < nslot; s += BLOCK) {
        unsigned x = G->f0;
        if ((s & G->sfm) != G->sbv) x = 0xffffu;
        else if (G->C)
            for (unsigned g = 0; g < G->ncam; g++)
                if (G->camloc[g] != 0xff && ((s >> G->camloc[g]) & 1u)) x = (x * G->camA[g]) % G->C;
        phot[s] = (unsigned short)x;
    }
}

// per tile: res[s] = the residue of slot s for this tile (0xffff: not populated), then the tile itself
template <int BLOCK, unsigned EPT>
__device__ __forceinline__ void gen_tile(const GenFront *G, amp_t *tile, const unsigned short *phot, unsigned short *res,
                                         uint64_t base, uint32_t packT, const uint32_t (&packK)[EPT])
{
    const unsigned nslot = 1u << G->h;
    const bool pop_out = (base & G->fixed_out) == (G->basis & G->fixed_out);
    for (unsigned s = threadIdx.x; s < nslot; s += BLOCK) {
        unsigned x = phot[s];
        if (!pop_out) x = 0xffffu;
        else if (x != 0xffffu && G->C) {
#pragma unroll
            for (unsigned f = 0; f < 5; f++)
                if ((G->present >> f) & 1u) x = (x * (unsigned)G->tabP[f][(unsigned)((base >> (8u * f)) & 255u)]) % G->C;
        }
        res[s] = (unsigned short)x;
    }
    __syncthreads();
    const uint32_t lowbase = (uint32_t)base & G->lowout_mask;
    const uint32_t parbase = (uint32_t)__builtin_popcountll(base & G->sign_out) & 1u;
    const double vp = G->v, vm = -G->v;
#pragma unroll
    for (unsigned k = 0; k < EPT; k++) {
        const uint32_t w = packT ^ packK[k];
        const unsigned f = res[w & 0xfffu];
        const uint32_t low = ((w >> 12) & 0xfffu) | lowbase;
        amp_t a; a.x = 0.0; a.y = 0.0;
        if (f != 0xffffu && ((low ^ f) & G->cmpmask) == 0u) a.x = (((w >> 24) ^ parbase) & 1u) ? vm : vp;
        tile[k * BLOCK + threadIdx.x] = a;
    }
}

// index with the tile number's bits deposited by segments: bits [src, src + len) of t go to [dst, dst + len)
__device__ __forceinline__ uint64_t fuse_deposit(uint64_t t, const FuseSeg *seg, unsigned nseg)
{
    uint64_t x = 0;
    for (unsigned k = 0; k < nseg; k++) x |= ((t >> seg[k].src) & (((uint64_t)1 << seg[k].len) - 1)) << seg[k].dst;
    return x;
}
// offset of a tile-local element index under a bit -> position table (linear in its bits)
__device__ __forceinline__ uint64_t fuse_spread(unsigned e, const uint8_t *pos, unsigned T)
{
    uint64_t off = 0;
    for (unsigned j = 0; j < T; j++) off |= (uint64_t)((e >> j) & 1u) << pos[j];
    return off;
}

// One controlled modular multiply on an LDS-resident tile whose low M local bits are the M register.
//   op.a    = M | (ctl_local + 1) << 8      (0 in the high part: the control is outside the tile)
//   op.mask = global-index mask of an outside control (0 if the control is tile-local)
// The destination -> source map of the permutation case (d = 1) is tabulated once per step by the first 2^M
// threads (one 32-bit modulo each) instead of one modulo per amplitude; `lut` lives behind the tile buffers.
template <int BLOCK, unsigned EPTC>
__device__ __forceinline__ void fuse_camodc_step(amp_t *tile, unsigned short *lut, const FuseOp *__restrict__ op,
                                                 uint64_t base, unsigned tsize)
{
    const uint64_t mext = op->mask;
    if ((base & mext) != mext) return;                                   // control outside the tile is 0: identity
    const unsigned M = op->a & 0xffu;
    const int ctl_local = (int)((op->a >> 8) & 0xffu) - 1;
    const FuseCamExtra X = *reinterpret_cast<const FuseCamExtra *>(&op->c);
    const unsigned blk = 1u << M, blkmask = blk - 1u;
    const bool perm = (X.d == 1);
    if (perm) {
        for (unsigned f = threadIdx.x; f < blk; f += BLOCK) lut[f] = (unsigned short)(f < X.C ? (f * X.inv) % X.C : f);
        __syncthreads();
    }
    amp_t acc[EPTC];
    bool wr[EPTC];
#pragma unroll
    for (unsigned k = 0; k < EPTC; k++) {
        wr[k] = false;
        const unsigned e = k * BLOCK + threadIdx.x;
        if (e < tsize) {
            const unsigned f = e & blkmask;
            const bool on = ctl_local < 0 || ((e >> ctl_local) & 1u);
            if (on && f < X.C) {
                amp_t s2; s2.x = 0.0; s2.y = 0.0;
                if (perm) {
                    const amp_t sv = tile[(e - f) + lut[f]];
                    s2.x += sv.x; s2.y += sv.y;
                } else if (f % X.d == 0) {
                    unsigned src = ((f / X.d) * X.inv) % X.Cd;
                    const amp_t *b = tile + (e - f);
                    for (unsigned q = 0; q < X.d; q++, src += X.Cd) { s2.x += b[src].x; s2.y += b[src].y; }
                }
                acc[k] = s2; wr[k] = true;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (unsigned k = 0; k < EPTC; k++)
        if (wr[k]) tile[k * BLOCK + threadIdx.x] = acc[k];
    __syncthreads();
}

// the gate list applied to one LDS-resident tile (shared by the plain and the pipelined pass kernels)
template <int BLOCK, int TT, unsigned EPT>
__device__ __forceinline__ void fuse_apply_ops(amp_t *tile, unsigned short *lut, const FusePass &P, const FuseOp *__restrict__ ops,
                                               uint64_t base, uint64_t off_t, const uint64_t (&off_k)[EPT],
                                               unsigned tsize, unsigned ept)
{
    unsigned i = 0;
    while (i < P.nops) {
        const uint32_t type = ops[i].type;
        if (type == FUSE_H) {
            const unsigned j = ops[i].a;
#pragma unroll
            for (unsigned k = 0; k < (EPT + 1) / 2; k++) {
                const unsigned p = k * BLOCK + threadIdx.x;
                if (p < tsize / 2) {
                    const unsigned i0 = (unsigned)insert_zero(p, j), i1 = i0 | (1u << j);
                    amp_t a = tile[i0], b = tile[i1];
                    h_butterfly(a, b);
                    tile[i0] = a; tile[i1] = b;
                }
            }
            __syncthreads();
            i++;
        } else if (type == FUSE_PHASE) {
            // a run of consecutive diagonal gates: the thread's amplitudes stay in registers for the whole run.
            // mask (global bits outside the tile) is a per-tile constant -> scalar skip; a = mask of tile-local bits
            unsigned gend = i + 1;
            while (gend < P.nops && ops[gend].type == FUSE_PHASE) gend++;
            amp_t v[EPT];
#pragma unroll
            for (unsigned k = 0; k < EPT; k++) {
                const unsigned e = k * BLOCK + threadIdx.x;
                if (k < ept && e < tsize) v[k] = tile[e];
            }
            for (unsigned o = i; o < gend; o++) {
                const uint64_t mext = ops[o].mask;                      // wave-uniform: scalar loads
                if ((base & mext) != mext) continue;
                const uint32_t mloc = ops[o].a;
                const double cc = ops[o].c, ss = ops[o].s;
#pragma unroll
                for (unsigned k = 0; k < EPT; k++) {
                    const unsigned e = k * BLOCK + threadIdx.x;
                    if (k < ept && e < tsize && (e & mloc) == mloc) {
                        amp_t w;
                        w.x = ((cc * v[k].x) - (ss * v[k].y)) + 0.0;
                        w.y = ((cc * v[k].y) + (ss * v[k].x)) + 0.0;
                        v[k] = w;
                    }
                }
            }
#pragma unroll
            for (unsigned k = 0; k < EPT; k++) {
                const unsigned e = k * BLOCK + threadIdx.x;
                if (k < ept && e < tsize) tile[e] = v[k];
            }
            __syncthreads();
            i = gend;
        } else {    // FUSE_CAMODC
            fuse_camodc_step<BLOCK, EPT>(tile, lut, ops + i, base, tsize);
            i++;
        }
    }

}

// ROUNDS form of the gate list (tiles with exactly 4 amplitudes per thread).  The host cuts the list into
// rounds; in a round every thread keeps the 4 amplitudes that differ in the round's two REGISTER BITS
// (rb0 < rb1, tile-local) in registers and applies the whole round to them:
//   H on a register bit   = butterflies between registers (two H's per LDS round trip: radix 4);
//   controlled phase      = rotate the registers the host selected (rsel: which of the 4 have the mask's
//                           register bits set), after ONE test of the remaining local bits per thread and one
//                           test of the bits outside the tile per tile.
// Same arithmetic, same order per amplitude as the per-gate kernels.  Records of a round (32 B each):
//   FUSE_ROUND   a = rb0 | rb1 << 8 | (round contains an H) << 16,  mask = records that follow (items + gates)
//   item H       type = FUSE_H    | (32 | which register bit) << 8
//   item run     type = FUSE_PRUN | (rsel | (canonicalise at the end of the run) << 4) << 8 | gates << 16  (1..64 gates)
//   gate of run  type = FUSE_PHASE | rsel << 8,  a = tile-local mask without the register bits,
//                mask = index bits outside the tile,  c, s = cos, sin
// Between rounds: FUSE_CAMRUN (folded modular multiplies) and FUSE_CAMODC (a single one) with their own layouts.
enum : uint32_t { FUSE_ROUND = 3, FUSE_CAMRUN = 5 };

__device__ __forceinline__ void rotate_amp(amp_t &v, double cc, double ss)
{
    amp_t w;
    w.x = ((cc * v.x) - (ss * v.y)) + 0.0;
    w.y = ((cc * v.y) + (ss * v.x)) + 0.0;
    v = w;
}

// A run of consecutive controlled phases that rotate the same registers (rsel).
//  * Which gates of the run act on this TILE (their controls outside the tile are all 1): lane l tests gate l
//    against the tile's base index, one ballot gives the live set, and only live gates are walked.  A skipped
//    gate costs nothing.
//  * The walk is hand-scheduled: measured, the compiler's version of this loop is bound by the scalar-ALU issue
//    port (about 20 s_* instructions per gate for mask iteration, record address, select and EXEC bookkeeping),
//    not by the FP64 work.  Here a gate costs ~9 scalar instructions: find/clear its bit, three scalar loads of
//    its record (the next gate's record is in flight while the current one is applied), a v_cmpx that masks the
//    lanes whose tile-local control bits are not all set, and the EXEC restore.
//  * A rotation is the same four products and two sums as rotate_amp, in place, without FMA.
//  * The reference's result of every rotation is canonical (its "0 + ..." turns -0 into +0).  The sign of a zero
//    never changes a later non-zero result and every zero result is canonicalised anyway, so the "+ 0.0" is
//    applied ONCE: at the end of a round that contains an H (an H touches every amplitude), otherwise at the end of
//    the run to the lanes that were rotated at least once (tch collects their EXEC masks); untouched lanes keep
//    their bits.  Same values, bit for bit, as rotate_amp per gate.
enum : uint32_t { FUSE_PRUN = 4 };

// one record = one s_load_dwordx8 into a fixed block of scalar registers (A: s[72:79], B: s[80:87]; listed as clobbers):
// dword 1 = tile-local mask, dwords 4-5 = cos, 6-7 = sin.  (Three separate loads of m, c, s cost two more instructions of the
// scalar port per gate, and this walk is bound by that port as much as by the vector one: DESIGN.md s4.)
#define QCX_RECA_M "s73"
#define QCX_RECA_C "s[76:77]"
#define QCX_RECA_S "s[78:79]"
#define QCX_RECB_M "s81"
#define QCX_RECB_C "s[84:85]"
#define QCX_RECB_S "s[86:87]"
#define QCX_LOADREC(BLK)                                             \
    "s_ff1_i32_b64 %[b], %[live]\n\t"                                \
    "s_bitset0_b64 %[live], %[b]\n\t"                                \
    "s_lshl_b32 %[b], %[b], 5\n\t"                                   \
    "s_load_dwordx8 " BLK ", %[base], %[b] offset:0x20\n\t"
#define QCX_ROT1(C, S, X, Y)                                         \
    "v_mul_f64 %[t0], " C ", %[" X "]\n\t"                          \
    "v_mul_f64 %[t1], " S ", %[" Y "]\n\t"                          \
    "v_mul_f64 %[t2], " C ", %[" Y "]\n\t"                          \
    "v_mul_f64 %[t3], " S ", %[" X "]\n\t"                          \
    "v_add_f64 %[" X "], %[t0], -%[t1]\n\t"                          \
    "v_add_f64 %[" Y "], %[t2], %[t3]\n\t"
#define QCX_ROT2(C, S, X0, Y0, X1, Y1)                               \
    "v_mul_f64 %[t0], " C ", %[" X0 "]\n\t"                         \
    "v_mul_f64 %[t1], " S ", %[" Y0 "]\n\t"                         \
    "v_mul_f64 %[t2], " C ", %[" Y0 "]\n\t"                         \
    "v_mul_f64 %[t3], " S ", %[" X0 "]\n\t"                         \
    "v_add_f64 %[" X0 "], %[t0], -%[t1]\n\t"                         \
    "v_mul_f64 %[t0], " C ", %[" X1 "]\n\t"                         \
    "v_mul_f64 %[t1], " S ", %[" Y1 "]\n\t"                         \
    "v_add_f64 %[" Y0 "], %[t2], %[t3]\n\t"                          \
    "v_mul_f64 %[t2], " C ", %[" Y1 "]\n\t"                         \
    "v_mul_f64 %[t3], " S ", %[" X1 "]\n\t"                         \
    "v_add_f64 %[" X1 "], %[t0], -%[t1]\n\t"                         \
    "v_add_f64 %[" Y1 "], %[t2], %[t3]\n\t"
#define QCX_R1(C, S, i) QCX_ROT1(C, S, "x" #i, "y" #i)
#define QCX_R2(C, S, i, j) QCX_ROT2(C, S, "x" #i, "y" #i, "x" #j, "y" #j)
// the rotations of one gate for every register selection RSEL (bit q set: amplitude register q is rotated)
#define QCX_ROTS_1(C, S)  QCX_R1(C, S, 0)
#define QCX_ROTS_2(C, S)  QCX_R1(C, S, 1)
#define QCX_ROTS_3(C, S)  QCX_R2(C, S, 0, 1)
#define QCX_ROTS_4(C, S)  QCX_R1(C, S, 2)
#define QCX_ROTS_5(C, S)  QCX_R2(C, S, 0, 2)
#define QCX_ROTS_6(C, S)  QCX_R2(C, S, 1, 2)
#define QCX_ROTS_7(C, S)  QCX_R2(C, S, 0, 1) QCX_R1(C, S, 2)
#define QCX_ROTS_8(C, S)  QCX_R1(C, S, 3)
#define QCX_ROTS_9(C, S)  QCX_R2(C, S, 0, 3)
#define QCX_ROTS_10(C, S) QCX_R2(C, S, 1, 3)
#define QCX_ROTS_11(C, S) QCX_R2(C, S, 0, 1) QCX_R1(C, S, 3)
#define QCX_ROTS_12(C, S) QCX_R2(C, S, 2, 3)
#define QCX_ROTS_13(C, S) QCX_R2(C, S, 0, 2) QCX_R1(C, S, 3)
#define QCX_ROTS_14(C, S) QCX_R2(C, S, 1, 2) QCX_R1(C, S, 3)
#define QCX_ROTS_15(C, S) QCX_R2(C, S, 0, 1) QCX_R2(C, S, 2, 3)
#define QCX_Z1(i) "v_add_f64 %[x" #i "], %[x" #i "], 0\n\t" "v_add_f64 %[y" #i "], %[y" #i "], 0\n\t"
#define QCX_ZERO_1  QCX_Z1(0)
#define QCX_ZERO_2  QCX_Z1(1)
#define QCX_ZERO_3  QCX_Z1(0) QCX_Z1(1)
#define QCX_ZERO_4  QCX_Z1(2)
#define QCX_ZERO_5  QCX_Z1(0) QCX_Z1(2)
#define QCX_ZERO_6  QCX_Z1(1) QCX_Z1(2)
#define QCX_ZERO_7  QCX_Z1(0) QCX_Z1(1) QCX_Z1(2)
#define QCX_ZERO_8  QCX_Z1(3)
#define QCX_ZERO_9  QCX_Z1(0) QCX_Z1(3)
#define QCX_ZERO_10 QCX_Z1(1) QCX_Z1(3)
#define QCX_ZERO_11 QCX_Z1(0) QCX_Z1(1) QCX_Z1(3)
#define QCX_ZERO_12 QCX_Z1(2) QCX_Z1(3)
#define QCX_ZERO_13 QCX_Z1(0) QCX_Z1(2) QCX_Z1(3)
#define QCX_ZERO_14 QCX_Z1(1) QCX_Z1(2) QCX_Z1(3)
#define QCX_ZERO_15 QCX_Z1(0) QCX_Z1(1) QCX_Z1(2) QCX_Z1(3)
// one gate: mask the lanes, rotate, restore EXEC.  A gate without tile-local conditions (its mask word is 0: the usual
// case, a phase whose target lies outside the tile) skips the two vector instructions of the lane mask and the EXEC
// round trip: the pass is bound by vector issue, the scalar compare + branch ride the other port.
#define QCX_GATE(M, ROTS)                                            \
    "s_cmp_eq_u32 " M ", 0\n\t"                                      \
    "s_cbranch_scc1 6f\n\t"                                         \
    "v_and_b32 %[t], " M ", %[p]\n\t"                                \
    "v_cmpx_eq_u32_e32 vcc, " M ", %[t]\n\t"                         \
    "s_cbranch_execz 1f\n\t"                                        \
    ROTS                                                             \
    "1:\n\t"                                                        \
    "s_mov_b64 exec, %[ex]\n\t"                                     \
    "s_branch 7f\n\t"                                               \
    "6:\n\t"                                                        \
    ROTS                                                             \
    "7:\n\t"
// H on one of the round's two register bits, in place and without the final "+ 0.0" (the round canonicalises once at
// its end): the same four products and four sums as h_butterfly
#define QCX_HBF(AX, AY, BX, BY)                                      \
    "v_mul_f64 %[t0], %[hs], %[" AX "]\n\t"                          \
    "v_mul_f64 %[t1], %[hs], %[" AY "]\n\t"                          \
    "v_mul_f64 %[t2], %[hs], %[" BX "]\n\t"                          \
    "v_mul_f64 %[t3], %[hs], %[" BY "]\n\t"                          \
    "v_add_f64 %[" AX "], %[t0], %[t2]\n\t"                          \
    "v_add_f64 %[" AY "], %[t1], %[t3]\n\t"                          \
    "v_add_f64 %[" BX "], %[t0], -%[t2]\n\t"                         \
    "v_add_f64 %[" BY "], %[t1], -%[t3]\n\t"
// all live gates of one run for register selection R (live != 0 on entry, 0 on exit); records ping-pong between
// blocks A and B; then the canonical zeros if bit 4 of rsel asks for them -- on ALL lanes of the run's registers: a gate
// kernel never meets a -0 it has not made itself (the state is canonical whenever gates run: a state written by the caller
// is canonicalised first, qcx_api.hip), so "+ 0.0" on an amplitude no gate of the run rotated changes nothing
#define QCX_RUN_BODY(R)                                              \
    QCX_LOADREC("s[72:79]")                                          \
    "2:\n\t"                                                         \
    "s_waitcnt lgkmcnt(0)\n\t"                                       \
    "s_cmp_eq_u64 %[live], 0\n\t"                                    \
    "s_cbranch_scc1 3f\n\t"                                          \
    QCX_LOADREC("s[80:87]")                                          \
    QCX_GATE(QCX_RECA_M, QCX_ROTS_##R(QCX_RECA_C, QCX_RECA_S))       \
    "s_waitcnt lgkmcnt(0)\n\t"                                       \
    "s_cmp_eq_u64 %[live], 0\n\t"                                    \
    "s_cbranch_scc1 4f\n\t"                                          \
    QCX_LOADREC("s[72:79]")                                          \
    QCX_GATE(QCX_RECB_M, QCX_ROTS_##R(QCX_RECB_C, QCX_RECB_S))       \
    "s_branch 2b\n\t"                                                \
    "3:\n\t"                                                         \
    QCX_GATE(QCX_RECA_M, QCX_ROTS_##R(QCX_RECA_C, QCX_RECA_S))       \
    "s_branch 5f\n\t"                                                \
    "4:\n\t"                                                         \
    QCX_GATE(QCX_RECB_M, QCX_ROTS_##R(QCX_RECB_C, QCX_RECB_S))       \
    "5:\n\t"                                                         \
    "s_bitcmp1_b32 %[rsel], 12\n\t"                                  \
    "s_cbranch_scc0 99f\n\t"                                         \
    QCX_ZERO_##R                                                     \
    "s_branch 99f\n\t"
// ONE asm statement for every register selection (entry 20 + R, binary dispatch on rsel & 15): with one statement
// per selection behind a C++ switch the compiler copied all four amplitudes into fresh VGPRs at every run
#define QCX_ITEM_ALL                                                 \
    /* prefetch for the NEXT item (1 + gates records further on): its header dword and its lanes' outside-tile masks */ \
    "s_lshr_b32 %[b], %[rsel], 16\n\t"                               \
    "s_add_u32 %[b], %[b], 1\n\t"                                    \
    "s_lshl_b32 %[tn], %[b], 5\n\t"                                  \
    "s_load_dword %[tn], %[base], %[tn]\n\t"                         \
    "v_lshl_add_u32 %[xa], %[b], 3, %[xa]\n\t"                       \
    "ds_read_b64 %[mn], %[xa]\n\t"                                   \
    "s_bitcmp1_b32 %[rsel], 13\n\t"                                  \
    "s_cbranch_scc1 50f\n\t"                                         \
    "s_mov_b64 %[ex], exec\n\t"                                      \
    "s_cmp_eq_u64 %[live], 0\n\t"                                    \
    "s_cbranch_scc1 99f\n\t"                                         \
    "s_bfe_u32 %[b], %[rsel], 0x40008\n\t"                           \
    "s_cmp_lt_u32 %[b], 8\n\t s_cbranch_scc1 40f\n\t"                \
    "s_cmp_lt_u32 %[b], 12\n\t s_cbranch_scc1 41f\n\t"               \
    "s_cmp_lt_u32 %[b], 14\n\t s_cbranch_scc1 42f\n\t"               \
    "s_cmp_eq_u32 %[b], 14\n\t s_cbranch_scc1 34f\n\t s_branch 35f\n\t" \
    "42:\n\t s_cmp_eq_u32 %[b], 12\n\t s_cbranch_scc1 32f\n\t s_branch 33f\n\t" \
    "41:\n\t s_cmp_lt_u32 %[b], 10\n\t s_cbranch_scc1 43f\n\t s_cmp_eq_u32 %[b], 10\n\t s_cbranch_scc1 30f\n\t s_branch 31f\n\t" \
    "43:\n\t s_cmp_eq_u32 %[b], 8\n\t s_cbranch_scc1 28f\n\t s_branch 29f\n\t" \
    "40:\n\t s_cmp_lt_u32 %[b], 4\n\t s_cbranch_scc1 44f\n\t"        \
    "s_cmp_lt_u32 %[b], 6\n\t s_cbranch_scc1 45f\n\t s_cmp_eq_u32 %[b], 6\n\t s_cbranch_scc1 26f\n\t s_branch 27f\n\t" \
    "45:\n\t s_cmp_eq_u32 %[b], 4\n\t s_cbranch_scc1 24f\n\t s_branch 25f\n\t" \
    "44:\n\t s_cmp_lt_u32 %[b], 2\n\t s_cbranch_scc1 21f\n\t s_cmp_eq_u32 %[b], 2\n\t s_cbranch_scc1 22f\n\t s_branch 23f\n\t" \
    "21:\n\t" QCX_RUN_BODY(1)  "22:\n\t" QCX_RUN_BODY(2)  "23:\n\t" QCX_RUN_BODY(3)  "24:\n\t" QCX_RUN_BODY(4)  \
    "25:\n\t" QCX_RUN_BODY(5)  "26:\n\t" QCX_RUN_BODY(6)  "27:\n\t" QCX_RUN_BODY(7)  "28:\n\t" QCX_RUN_BODY(8)  \
    "29:\n\t" QCX_RUN_BODY(9)  "30:\n\t" QCX_RUN_BODY(10) "31:\n\t" QCX_RUN_BODY(11) "32:\n\t" QCX_RUN_BODY(12) \
    "33:\n\t" QCX_RUN_BODY(13) "34:\n\t" QCX_RUN_BODY(14) "35:\n\t" QCX_RUN_BODY(15) \
    "50:\n\t"                                                        \
    "s_bitcmp1_b32 %[rsel], 8\n\t"                                   \
    "s_cbranch_scc1 51f\n\t"                                         \
    QCX_HBF("x0", "y0", "x1", "y1") QCX_HBF("x2", "y2", "x3", "y3")  \
    "s_branch 99f\n\t"                                               \
    "51:\n\t"                                                        \
    QCX_HBF("x0", "y0", "x2", "y2") QCX_HBF("x1", "y1", "x3", "y3")  \
    "99:\n\t"                                                        \
    "s_waitcnt lgkmcnt(0)\n\t"

// the four amplitudes a thread holds during a round, as eight separate doubles
struct Quad { double x0, y0, x1, y1, x2, y2, x3, y3; };

// One item of a round: an H on one of the round's register bits, or a phase run (at most 64 gates: the host cuts
// longer ones).  hdr = the item's header dword: type | code << 8 | gates << 16 with code bit 5 = H (bit 0: which
// register bit); otherwise code bits 0-3 = which amplitude registers the run rotates, bit 4 = apply the canonical
// zeros (the reference's "+ 0") to the rotated lanes at its end -- clear when the round canonicalises everything at
// its end anyway (it contains an H).  item = the item's header record, the gates of a run follow it; live = the run's
// gates whose outside-tile controls are all 1 for this tile.
// The statement also fetches what the NEXT item needs first -- its header dword (hdr_next) and this lane's entry of
// its outside-tile masks (mask_next; xaddr = LDS byte address of the current item's entry, advanced in place) -- so
// that those two latencies overlap this item's work instead of starting the next one.
// ONE asm statement for everything that touches the amplitudes inside the round's item loop: with separate
// statements (or C++ butterflies) the compiler keeps two sets of VGPRs for the four amplitudes and copies between
// them at every item, which costs more than a gate.
// The record blocks s[72:79] / s[80:87] are named registers on the clobber list.  The kernels that hold this statement are
// built for at most 7 waves per SIMD (round 5), where the blocks lie inside the allocator's budget: it works around the
// clobbers, the kernel descriptor counts them, and hipcc has nothing to warn about (tests/test_abi_and_build.py checks
// the build log and the metadata).  (Rounds 2-4 also had 8-wave builds, with the blocks beyond the allocator's budget --
// "reserved registers" -- which were no faster: 96 SGPRs admit 7 blocks of 256 threads per CU, not 8.)
__device__ __forceinline__ void fuse_round_item(Quad &q, const FuseOp *item, uint64_t live, unsigned p, uint32_t hdr,
                                                uint32_t &xaddr, uint32_t &hdr_next, uint64_t &mask_next)
{
    uint64_t ex; uint32_t bidx, tv;
    double t0, t1, t2, t3;
    const double hs = QCX_SQRT1_2;
    asm(QCX_ITEM_ALL
        : [live] "+s"(live), [ex] "=&s"(ex), [b] "=&s"(bidx), [t] "=&v"(tv),
          [x0] "+v"(q.x0), [y0] "+v"(q.y0), [x1] "+v"(q.x1), [y1] "+v"(q.y1), [x2] "+v"(q.x2), [y2] "+v"(q.y2), [x3] "+v"(q.x3), [y3] "+v"(q.y3),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
          [tn] "=&s"(hdr_next), [mn] "=&v"(mask_next), [xa] "+v"(xaddr)
        : [base] "s"(item), [p] "v"(p), [rsel] "s"(hdr), [hs] "s"(hs)
        : "vcc", "scc", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87");
}


// a run of consecutive permutation-type modular multiplies (gcd(A, C) = 1, same C) folded into ONE
// gather: inside a 2^M block all amplitudes share the L-register bits, so the run moves the amplitude
// of value g from  f0 = (prod of the inverses whose control is set) * g  mod C.  The product is
// walked through per-gate tables x -> x * inv_i mod C kept in LDS (camtab); pure data movement, the
// same bits as applying the gates one by one.  Returns the number of gate records behind the header.
template <int BLOCK, int TT>
__device__ __forceinline__ unsigned fuse_camrun_step(amp_t *tile, unsigned short *lut, const unsigned char *camtab,
                                                     const FuseOp *__restrict__ ops, unsigned i, uint64_t base)
{
    const unsigned cnt = ops[i].a & 0xffffu, cpad = ops[i].a >> 16;
    const unsigned char *tabs = camtab + (unsigned)ops[i].mask;
    const FuseOp *rec = ops + i + 1;
    const unsigned M = rec[0].a & 0xffu, blkmask = (1u << M) - 1u;
    const unsigned Cn = reinterpret_cast<const FuseCamExtra *>(&rec[0].c)->C;
    // The factor depends on the index only through the control bits, i.e. through the 2^M-BLOCK an amplitude
    // sits in (controls are L-register qubits, at or above M): it is walked once per block by the first
    // 2^(T-M) threads and left in LDS (the scratch behind the tile), not once per amplitude by everybody.
    unsigned char *xblk = reinterpret_cast<unsigned char *>(lut);
    for (unsigned b = threadIdx.x; b < ((1u << TT) >> M); b += BLOCK) {
        const unsigned e = b << M;
        unsigned x = 1;
        for (unsigned g = 0; g < cnt; g++) {
            const uint64_t mext = rec[g].mask;
            if ((base & mext) != mext) continue;                  // outside control is 0 for this tile
            const int cl = (int)((rec[g].a >> 8) & 0xffu) - 1;
            if (cl < 0 || ((e >> cl) & 1u)) x = tabs[g * cpad + x];
        }
        xblk[b] = (unsigned char)x;
    }
    __syncthreads();
    const float rc = 1.0f / (float)Cn;
    amp_t acc[4];
    bool wr[4];
#pragma unroll
    for (unsigned k = 0; k < 4; k++) {
        const unsigned e = k * BLOCK + threadIdx.x, f = e & blkmask;
        const unsigned x = xblk[e >> M];
        wr[k] = (x != 1u) && (f < Cn);
        if (wr[k]) {
            // (x * f) mod C for x, f < C <= 256: quotient estimate from the float reciprocal, off by at most one
            const unsigned v = x * f;
            unsigned q = (unsigned)((float)v * rc);
            int rem = (int)v - (int)(q * Cn);
            if (rem < 0) rem += (int)Cn; else if (rem >= (int)Cn) rem -= (int)Cn;
            const amp_t sv = tile[(e - f) + (unsigned)rem];
            acc[k].x = 0.0 + sv.x; acc[k].y = 0.0 + sv.y;
        }
    }
    __syncthreads();
#pragma unroll
    for (unsigned k = 0; k < 4; k++)
        if (wr[k]) tile[k * BLOCK + threadIdx.x] = acc[k];
    __syncthreads();
    return cnt;
}

// ---------------------------------------------------------------------------
// K6t  TOLERANCE MODE of the rounds form (qcx_set_fusion(reg, 2); opt-in, NOT bit-exact: amplitudes agree with the
// reference to ~1e-15, north_star allows 1e-10).  The planner merges every run of consecutive controlled phases that
// share a control qubit l into ONE diagonal (Q:682-689 applies l - M of them after each Hadamard):
//     amp[i] *= prod over the targets k with bit k of i set of (c_k + i s_k),     for every i with bit l set.
// The product splits over where the target bits live:
//     E_out(tile)        targets outside the tile: a per-tile constant.  One thread per diagonal builds it at tile start
//                        (while the LDS-DMA fill is in flight) from <= 5 tables of 256 entries indexed by the bytes of the
//                        tile's base index (global memory, L2-resident), and leaves it in LDS;
//     G0 G1 G2 (thread)  targets on tile-local bits, 4 local bits per table, 16 entries each, in LDS (768 B per diagonal);
//                        the entry for a single set bit is that bit's own factor, so the factors of the round's two
//                        register bits are read from the same tables.
// A thread then multiplies the two amplitudes whose control bit is set by F and F w(other register bit): one complex
// multiply per amplitude instead of one rotation per gate per amplitude, FMA allowed.
// Only rounds of the shape the schedule of Q:682-689 produces carry merged diagonals (FUSE_QROUND below); everywhere
// else the planner puts the original phases back, and those -- like every Hadamard outside a fast round and the modular
// multiplies -- run through the exact interpreter, same bits as in modes 0 / 1.
// ---------------------------------------------------------------------------
// The schedule of Q:682-689 gives rounds of the shape  H(x) D(x) H(y) D(y)  (x, y = the round's register bits, D(b) = a
// merged diagonal whose control is b).  The host recognises them (FUSE_QROUND: header + ONE record with two step words)
// and the kernel runs them as straight-line code: every table read of the round is issued up front, nothing is fetched
// between the butterflies.  Step word: bit 0 = the step's bit is rb1, bit 1 = a diagonal follows the H, bits 8-15 its
// slot, bits 16-18 its groups, bit 19 = the round's OTHER register bit is one of its targets; 0xffffffff = no step.
enum : uint32_t { FUSE_DIAG = 6 /* planner-side only: a merged diagonal before the rounds are formed; never reaches a kernel */, FUSE_QROUND = 7 };
struct DiagInfo {               // 48 B = 3 units of 16 B; offsets in 16-B units from the table area
    uint32_t field_off[5]; uint32_t present; uint32_t pad[2];
    double   kc, ks;            // constant factor of the diagonal: the product of the run's phases whose other qubit is a constant 1
};                              // of the view (a shard-id or slice bit): a sharded register hands such phases down with one mask bit

__device__ __forceinline__ void cmul_tol(amp_t &v, const amp_t w)        // v *= w, 2 mul + 2 fma
{
    const double nx = __builtin_fma(w.x, v.x, -(w.y * v.y));
    const double ny = __builtin_fma(w.x, v.y, w.y * v.x);
    v.x = nx; v.y = ny;
}
__device__ __forceinline__ void h_butterfly_tol(amp_t &a, amp_t &b)
{
    const double s = QCX_SQRT1_2;
    const double t0r = s * a.x, t0i = s * a.y, t1r = s * b.x, t1i = s * b.y;
    a.x = t0r + t1r;  a.y = t0i + t1i;
    b.x = t0r - t1r;  b.y = t0i - t1i;
}

// factors of one step of a fast round: F for the register whose other register bit is clear, F3 for the one where it is set
__device__ __forceinline__ void qround_factors(uint32_t s, unsigned p, unsigned rb0, unsigned rb1, const amp_t *dg, const amp_t *gtab,
                                               amp_t &F, amp_t &F3)
{
    const unsigned slot = (s >> 8) & 0xffu;
    const amp_t *G = gtab + slot * 48u;
    F = dg[slot];
    if (s & (1u << 16)) cmul_tol(F, G[p & 15u]);
    if (s & (1u << 17)) cmul_tol(F, G[16u + ((p >> 4) & 15u)]);
    if (s & (1u << 18)) cmul_tol(F, G[32u + (p >> 8)]);
    F3 = F;
    if (s & (1u << 19)) {
        const unsigned ob = (s & 1u) ? rb0 : rb1;
        cmul_tol(F3, G[16u * (ob >> 2) + (1u << (ob & 3u))]);
    }
}

// one step of a fast round on fixed registers: H between (lo0, hi0) and (lo1, hi1), then (dgl) the diagonal on the two
// amplitudes that have the step's bit set
__device__ __forceinline__ void qround_step(bool dgl, amp_t &lo0, amp_t &hi0, amp_t &lo1, amp_t &hi1, const amp_t F, const amp_t F3)
{
    h_butterfly_tol(lo0, hi0); h_butterfly_tol(lo1, hi1);
    if (dgl) { cmul_tol(hi0, F); cmul_tol(hi1, F3); }
}

// a whole fast round with the bit of each step fixed at compile time (HA / HB: the step works on rb1), from the LDS reads
// to the LDS writes: written once per combination so that which register a step touches is never a run-time choice (a
// run-time choice makes the compiler keep the four amplitudes in scratch memory and index them)
template <bool HA, bool HB>
__device__ __forceinline__ void qround_run(amp_t *tile, unsigned p, unsigned e1, unsigned e2, unsigned e3, unsigned rb0, unsigned rb1,
                                           uint32_t sA, uint32_t sB, const amp_t *dg, const amp_t *gtab)
{
    amp_t F, F3;
    F.x = F3.x = 1.0; F.y = F3.y = 0.0;
    const bool dA = (sA & 2u) != 0, two = sB != 0xffffffffu, dB = two && (sB & 2u);
    amp_t v0 = tile[p], v1 = tile[e1], v2 = tile[e2], v3 = tile[e3];
    if (dA) qround_factors(sA, p, rb0, rb1, dg, gtab, F, F3);
    if (HA) qround_step(dA, v0, v2, v1, v3, F, F3); else qround_step(dA, v0, v1, v2, v3, F, F3);
    __builtin_amdgcn_sched_barrier(0);          // (keeps the second step's table reads out of the first step's registers)
    if (two) {
        if (dB) qround_factors(sB, p, rb0, rb1, dg, gtab, F, F3);
        if (HB) qround_step(dB, v0, v2, v1, v3, F, F3); else qround_step(dB, v0, v1, v2, v3, F, F3);
    }
    tile[p] = v0; tile[e1] = v1; tile[e2] = v2; tile[e3] = v3;
}

// the same fast round on a COLUMN of k_gen_cols: the four amplitudes sit at l0 .. l3 of the column, p is their logical
// tile-local index (the table look-ups want its bits)
template <bool HA, bool HB>
__device__ __forceinline__ void qround_run_col(amp_t *col, unsigned l0, unsigned l1, unsigned l2, unsigned l3, unsigned p, unsigned rb0, unsigned rb1,
                                               uint32_t sA, uint32_t sB, const amp_t *dg, const amp_t *gtab)
{
    amp_t F, F3;
    F.x = F3.x = 1.0; F.y = F3.y = 0.0;
    const bool dA = (sA & 2u) != 0, two = sB != 0xffffffffu, dB = two && (sB & 2u);
    amp_t v0 = col[l0], v1 = col[l1], v2 = col[l2], v3 = col[l3];
    if (dA) qround_factors(sA, p, rb0, rb1, dg, gtab, F, F3);
    if (HA) qround_step(dA, v0, v2, v1, v3, F, F3); else qround_step(dA, v0, v1, v2, v3, F, F3);
    __builtin_amdgcn_sched_barrier(0);
    if (two) {
        if (dB) qround_factors(sB, p, rb0, rb1, dg, gtab, F, F3);
        if (HB) qround_step(dB, v0, v2, v1, v3, F, F3); else qround_step(dB, v0, v1, v2, v3, F, F3);
    }
    col[l0] = v0; col[l1] = v1; col[l2] = v2; col[l3] = v3;
}

// zero-wave skipping (FusePass::zskip): the element index of a thread when the wave number rides on the tile-local bits zb[]
// and the lane number on the remaining positions, ascending.  `list` = the positions to leave out of the lane number (the zb[]
// and, inside a round, its register bits), ascending, one per byte (the host sorts them: FusePass::zlist for the whole tile,
// the c field of a round's header record for a round); `wavepart` = the wave number's bits already at their zb[] positions.
__device__ __forceinline__ unsigned zskip_index(unsigned lanebits, unsigned wavepart, uint64_t list, unsigned cnt)
{
    unsigned x = lanebits;
    for (unsigned k = 0; k < cnt; k++) x = (unsigned)insert_zero(x, (unsigned)(list >> (8u * k)) & 0xffu);
    return x | wavepart;
}
// does this wave's share of the tile (the elements whose zb[] bits spell its number) hold nothing but +0 ?
template <int TT>
__device__ __forceinline__ bool zskip_wave_is_zero(const amp_t *tile, const FusePass &P, unsigned wavepart)
{
    const unsigned lane = threadIdx.x & 63u;
    constexpr unsigned per = (1u << TT) >> 6;                   // elements per lane over the whole tile
    const unsigned share = per >> P.zskip;                        // ... of this wave's share
    uint64_t any = 0;
    for (unsigned k = 0; k < share; k++) {
        const amp_t v = tile[zskip_index(lane | (k << 6), wavepart, P.zlist, P.zskip)];
        any |= (uint64_t)__double_as_longlong(v.x) | (uint64_t)__double_as_longlong(v.y);
    }
    return __ballot(any != 0) == 0ULL;
}

template <int BLOCK, int TT, bool CAM = true, int TOL = 0>     // CAM = false: the pass holds no modular multiply (smaller kernel); TOL: 1 tolerance mode, 2 fast rounds only
__device__ __forceinline__ void fuse_apply_rounds(amp_t *tile, unsigned short *lut, const unsigned char *camtab,
                                                  const uint64_t *xm, const FusePass &P, const FuseOp *__restrict__ ops,
                                                  const FuseOp *ops_asm, uint64_t base, const amp_t *dg = nullptr, unsigned zrot = 0)
{
    static_assert((1u << TT) == 4u * BLOCK, "rounds form needs 4 amplitudes per thread");
    unsigned i = 0;
    // zskip: which share of the tile this wave takes rotates with the tile number (zrot) -- the populated shares are the same in
    // every tile, and a wave sits on one SIMD for good: without the rotation one SIMD would do all the work
    const unsigned zwave = ((threadIdx.x >> 6) + zrot) & ((BLOCK >> 6) - 1u);
    unsigned zpart = 0;
    for (unsigned j = 0; j < P.zskip; j++) zpart |= ((zwave >> j) & 1u) << P.zb[j];
    const bool idle = !CAM && P.zskip && zskip_wave_is_zero<TT>(tile, P, zpart);      // this wave holds nothing: it only keeps the barriers
    while (i < P.nops) {
        const uint32_t type = ops[i].type & 0xffu;
        if (TOL && type == FUSE_QROUND) {
            // tolerance mode, fast round  H(x) [D(x)] [H(y) [D(y)]]: straight-line, every table read issued up front
            const amp_t *gtab = dg + P.dg_cnt;                      // behind the per-tile E_out slots
            const unsigned rb0 = ops[i].a & 0xffu, rb1 = (ops[i].a >> 8) & 0xffu;
            const uint32_t sA = ops[i + 1].type, sB = ops[i + 1].a;
            if (!idle) {
            uint64_t zl; memcpy(&zl, &ops[i].c, sizeof zl);           // (zskip: the positions the lane number leaves out, sorted by the host)
            const unsigned p = P.zskip ? zskip_index(threadIdx.x & 63u, zpart, zl, P.zskip + 2u)
                                       : (unsigned)insert_zero(insert_zero(threadIdx.x, rb0), rb1);
            const unsigned e1 = p | (1u << rb0), e2 = p | (1u << rb1), e3 = e1 | (1u << rb1);
            if (sA & 1u) { if (sB & 1u) qround_run<true, true>(tile, p, e1, e2, e3, rb0, rb1, sA, sB, dg, gtab);
                           else         qround_run<true, false>(tile, p, e1, e2, e3, rb0, rb1, sA, sB, dg, gtab); }
            else         { if (sB & 1u) qround_run<false, true>(tile, p, e1, e2, e3, rb0, rb1, sA, sB, dg, gtab);
                           else         qround_run<false, false>(tile, p, e1, e2, e3, rb0, rb1, sA, sB, dg, gtab); }
            }
            __syncthreads();
            i += 2;
        } else if (TOL != 2 && type == FUSE_ROUND) {
            const unsigned rb0 = ops[i].a & 0xffu, rb1 = (ops[i].a >> 8) & 0xffu;
            const unsigned cnt = (unsigned)ops[i].mask;
            if (!idle) {
            uint64_t zl; memcpy(&zl, &ops[i].c, sizeof zl);           // (zskip: the positions the lane number leaves out, sorted by the host)
            const unsigned p = P.zskip ? zskip_index(threadIdx.x & 63u, zpart, zl, P.zskip + 2u)
                                       : (unsigned)insert_zero(insert_zero(threadIdx.x, rb0), rb1);
            const unsigned e1 = p | (1u << rb0), e2 = p | (1u << rb1), e3 = e1 | (1u << rb1);
            Quad q;
            { const amp_t v0 = tile[p], v1 = tile[e1], v2 = tile[e2], v3 = tile[e3];
              q.x0 = v0.x; q.y0 = v0.y; q.x1 = v1.x; q.y1 = v1.y; q.x2 = v2.x; q.y2 = v2.y; q.x3 = v3.x; q.y3 = v3.y; }
            // bit 16 of the header: the round contains an H.  It then canonicalises all amplitudes once at its end and
            // its phase runs skip their own canonical zeros (canon_bit clear).
            const bool has_h = (ops[i].a >> 16) & 1u;
            // items: the header's first dword says everything (one scalar load per item): type | code << 8 | gates << 16,
            // code as in fuse_round_item.  The records of a run's gates follow its header.
            // The walk reads the records through ops_asm, a second (non-restrict) kernel argument with the same value:
            // handing `ops` itself to an asm statement would count as a capture, after which the memory-clobbering
            // s_waitcnt statements of the pipeline could "modify" the records and every ops[] access in this function
            // would turn from a scalar load into a vector load.
            unsigned o = i + 1;
            const unsigned oend = i + cnt;
            const unsigned lane = threadIdx.x & 63u;
            // xm: the records' outside-tile masks staged in LDS (padded: a lane may look up to 64 entries past a run)
            const __attribute__((address_space(3))) uint64_t *xl = (const __attribute__((address_space(3))) uint64_t *)xm + o + 1 + lane;
            uint32_t xaddr = (uint32_t)(uintptr_t)xl;
            uint32_t t = ops[o].type;
            uint64_t m = *xl;
            do {        // a round has at least one item
                const unsigned rc = t >> 16;                             // 0 for an H: no live gates
                const uint64_t live = __builtin_amdgcn_ballot_w64((base & m) == m) & (rc >= 64 ? ~(uint64_t)0 : ((uint64_t)1 << rc) - 1);
                uint32_t tn; uint64_t mn;
                fuse_round_item(q, ops_asm + o, live, p, t, xaddr, tn, mn);
                o += 1 + rc; t = tn; m = mn;
            } while (o <= oend);
            if (has_h) {
                q.x0 += 0.0; q.y0 += 0.0; q.x1 += 0.0; q.y1 += 0.0; q.x2 += 0.0; q.y2 += 0.0; q.x3 += 0.0; q.y3 += 0.0;
            }
            { amp_t v0, v1, v2, v3;
              v0.x = q.x0; v0.y = q.y0; v1.x = q.x1; v1.y = q.y1; v2.x = q.x2; v2.y = q.y2; v3.x = q.x3; v3.y = q.y3;
              tile[p] = v0; tile[e1] = v1; tile[e2] = v2; tile[e3] = v3; }
            }
            __syncthreads();
            i += 1 + cnt;
        } else if (CAM && type == FUSE_CAMRUN) {
            i += 1 + fuse_camrun_step<BLOCK, TT>(tile, lut, camtab, ops, i, base);
        } else if (CAM) {    // FUSE_CAMODC between rounds
            fuse_camodc_step<BLOCK, 4>(tile, lut, ops + i, base, 1u << TT);
            i++;
        } else {
            __builtin_unreachable();
        }
    }
}


template <int BLOCK, int TT, bool LDSDMA>   // TT = tile bits when known at compile time (loops unroll, loads batch); 0 = generic
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(6))) void k_fused(amp_t *__restrict__ amp, unsigned n, FusePass P,
                                                   const FuseOp *__restrict__ ops, uint64_t ntiles, const FuseOp *ops_asm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char qcx_lds_raw[];
    amp_t *tile = reinterpret_cast<amp_t *>(qcx_lds_raw);
    const unsigned T = TT ? (unsigned)TT : P.T, c = P.c, nh = P.nh;
    const unsigned tsize = 1u << T;
    unsigned short *lut = reinterpret_cast<unsigned short *>(tile + tsize);  // behind the tile (host sizes the LDS)
    unsigned char *camtab = reinterpret_cast<unsigned char *>(lut) + P.cam_ctl_local[3];   // tables of folded multiply runs
    for (unsigned b = threadIdx.x; b < (unsigned)P.cam_ctl_local[1]; b += BLOCK)
        camtab[b] = reinterpret_cast<const unsigned char *>(ops + P.cam_ctl_local[2])[b];
    uint64_t *xm = P.xm_cnt ? reinterpret_cast<uint64_t *>(reinterpret_cast<unsigned char *>(lut) + P.xm_off) : nullptr;
    for (unsigned b = threadIdx.x; b < P.xm_cnt + 66u && P.xm_cnt; b += BLOCK) xm[b] = b < P.xm_cnt ? ops[b].mask : 0;    // + padding
    __syncthreads();
    constexpr unsigned EPT = TT ? ((1u << TT) + BLOCK - 1) / BLOCK : 16;      // elements per thread (<= 16)
    const unsigned ept = TT ? EPT : (tsize + BLOCK - 1) / BLOCK;
    (void)c; (void)nh;
    // global offset of a tile-local element index: linear in its bits, so split thread part / k part (this kernel works in
    // place on the identity layout: the output tables of FusePass equal the input ones)
    auto scatter = [&](unsigned e) -> uint64_t { return fuse_spread(e, P.in_pos, T); };
    const uint64_t off_t = scatter(threadIdx.x);
    uint64_t off_k[EPT];
#pragma unroll
    for (unsigned k = 0; k < EPT; k++) off_k[k] = scatter(k * BLOCK);

    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t base = fuse_deposit(t, P.seg_in, P.nseg_in);
        amp_t *g = amp + (base | off_t);

        if constexpr (TT != 0 && LDSDMA) {
            // LDS-DMA: every wave instruction moves 64 x 16 B straight from HBM into 1 KiB of the tile
            // (lane l lands at base + 16 l); no staging registers, all loads of the tile in flight at once
            const unsigned wbase = (threadIdx.x >> 6) * 64;
#pragma unroll
            for (unsigned k = 0; k < EPT; k++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + off_k[k]),
                                                 (__attribute__((address_space(3))) void *)(tile + k * BLOCK + wbase), 16, 0, 2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {   // all loads of the tile in flight before the first LDS write
            amp_t v[EPT];
#pragma unroll
            for (unsigned k = 0; k < EPT; k++)
                if (k < ept && k * BLOCK + threadIdx.x < tsize) v[k] = __builtin_nontemporal_load(g + off_k[k]);
#pragma unroll
            for (unsigned k = 0; k < EPT; k++)
                if (k < ept && k * BLOCK + threadIdx.x < tsize) tile[k * BLOCK + threadIdx.x] = v[k];
        }
        __syncthreads();

        if constexpr (TT != 0 && (1u << TT) == 4u * BLOCK) {
            if (P.cam_ctl_local[0]) fuse_apply_rounds<BLOCK, TT>(tile, lut, camtab, xm, P, ops, ops_asm, base);
            else fuse_apply_ops<BLOCK, TT, EPT>(tile, lut, P, ops, base, off_t, off_k, tsize, ept);
        } else {
            fuse_apply_ops<BLOCK, TT, EPT>(tile, lut, P, ops, base, off_t, off_k, tsize, ept);
        }

        {
            amp_t v[EPT];
#pragma unroll
            for (unsigned k = 0; k < EPT; k++)
                if (k < ept && k * BLOCK + threadIdx.x < tsize) v[k] = tile[k * BLOCK + threadIdx.x];
#pragma unroll
            for (unsigned k = 0; k < EPT; k++)
                if (k < ept && k * BLOCK + threadIdx.x < tsize) __builtin_nontemporal_store(v[k], g + off_k[k]);
        }
        __syncthreads();
    }
}

// The ROUNDS form on its own: the kernel the scheduler launches for every pass in rounds form (4 amplitudes per
// thread: BLOCK = 2^TT / 4).  One tile per workgroup at a time, LDS-DMA fill, a grid of a few workgroups per CU
// slot that walk the tiles (the table fill below is paid once per workgroup, not once per tile).  Load / process /
// store phases of the resident workgroups overlap each other; measured, that beats the double-buffered pipeline
// inside one workgroup (the round-1 pipelined form, removed) as soon as enough workgroups are resident, so this kernel carries nothing but
// the rounds interpreter and is held to OCC waves per SIMD.
template <int BLOCK, int TT, int OCC, bool CAM, int TOL = 0, bool GEN = false>      // GEN: the tiles are generated, not read (GenFront)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(OCC))) void k_fused_rounds(
    const amp_t *amp, amp_t *amp_out, unsigned n, FusePass P, const FuseOp *__restrict__ ops, uint64_t ntiles, const FuseOp *ops_asm)
{
    static_assert((1u << TT) == 4u * BLOCK, "rounds form: 4 amplitudes per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char qcx_lds_raw[];
    amp_t *tile = reinterpret_cast<amp_t *>(qcx_lds_raw);
    constexpr unsigned tsize = 1u << TT;
    unsigned short *lut = reinterpret_cast<unsigned short *>(tile + tsize);
    unsigned char *camtab = reinterpret_cast<unsigned char *>(lut) + P.cam_ctl_local[3];
    for (unsigned b = threadIdx.x; b < (unsigned)P.cam_ctl_local[1]; b += BLOCK)
        camtab[b] = reinterpret_cast<const unsigned char *>(ops + P.cam_ctl_local[2])[b];
    uint64_t *xm = reinterpret_cast<uint64_t *>(reinterpret_cast<unsigned char *>(lut) + P.xm_off);
    if constexpr (TOL != 2)
        for (unsigned b = threadIdx.x; b < P.xm_cnt + 66u; b += BLOCK) xm[b] = b < P.xm_cnt ? ops[b].mask : 0;    // + padding
    // tolerance mode: [E_out slot per diagonal][G tables, 48 entries per diagonal] in LDS; the tables are staged once
    amp_t *dg = reinterpret_cast<amp_t *>(reinterpret_cast<unsigned char *>(lut) + P.dg_lds_off);
    const amp_t *dg_area = reinterpret_cast<const amp_t *>(ops + P.dg_rec_off);             // global: DiagInfo[], G tables, field tables
    if constexpr (TOL)
        for (unsigned b = threadIdx.x; b < P.dg_cnt * 48u; b += BLOCK) dg[P.dg_cnt + b] = dg_area[3u * P.dg_cnt + b];
    __syncthreads();
    const uint64_t off_t = fuse_spread(threadIdx.x, P.in_pos, TT);
    uint64_t off_k[4], st_k[4];
    unsigned ld_k[4];
#pragma unroll
    for (unsigned k = 0; k < 4; k++) {
        off_k[k] = fuse_spread(k * BLOCK, P.in_pos, TT);
        st_k[k] = fuse_spread(k * BLOCK, P.st_pos, TT);
        ld_k[k] = (unsigned)fuse_spread(k * BLOCK, P.st_loc, TT);
    }
    const uint64_t st_t = fuse_spread(threadIdx.x, P.st_pos, TT);
    const unsigned ld_t = (unsigned)fuse_spread(threadIdx.x, P.st_loc, TT);
    const unsigned wbase = (threadIdx.x >> 6) * 64;
    // gen: the tiles are generated (the circuit front on a basis state that was never written), not read
    const GenFront *GF = reinterpret_cast<const GenFront *>(ops + P.gen_rec_off);
    unsigned short *gen_phot = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(lut) + P.gen_lds_off);
    unsigned short *gen_res = gen_phot + 1024;
    uint32_t packT = 0, packK[4] = {0, 0, 0, 0};
    if constexpr (GEN) {
        gen_setup<BLOCK>(GF, gen_phot);
        packT = gen_pack(threadIdx.x, GF, TT);
#pragma unroll
        for (unsigned k = 0; k < 4; k++) packK[k] = gen_pack(k * BLOCK, GF, TT);
        __syncthreads();
    }
    const unsigned slog = (P.dbg >> 12) & 15u;
    for (uint64_t b0 = blockIdx.x; b0 < ntiles; b0 += gridDim.x) {
        const uint64_t t = fuse_stream_tile(b0, n - TT, slog, (P.dbg >> 16) & 31u);       // (round 3's fuse_swz, the same idea for this kernel alone, is gone)
        const uint64_t base_in = fuse_deposit(t, P.seg_in, P.nseg_in);
        uint64_t base = base_in, base_out = base_in;               // logical base (gate records), output base
        if (P.chained) { base = fuse_deposit(t, P.seg_lg, P.nseg_lg); base_out = fuse_deposit(t, P.seg_out, P.nseg_out); }
        const amp_t *g = amp + (base_in | off_t);
        amp_t *go = amp_out + (base_out | st_t);
        if constexpr (GEN) gen_tile<BLOCK, 4>(GF, tile, gen_phot, gen_res, base, packT, packK);
        else if (!(P.dbg & 4u)) {
#pragma unroll
            for (unsigned k = 0; k < 4; k++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + off_k[k]),
                                                 (__attribute__((address_space(3))) void *)(tile + k * BLOCK + wbase), 16, 0, 2);
        }
        if constexpr (TOL) {
            // E_out of this tile for every diagonal of the pass: one thread each, while the tile fill is in flight
            if (threadIdx.x < P.dg_cnt) {
                const DiagInfo *info = reinterpret_cast<const DiagInfo *>(dg_area) + threadIdx.x;
                const uint32_t present = info->present;
                amp_t E; E.x = info->kc; E.y = info->ks;
#pragma unroll
                for (unsigned f = 0; f < 5; f++)
                    if ((present >> f) & 1u) cmul_tol(E, dg_area[info->field_off[f] + (unsigned)((base >> (8u * f)) & 255u)]);
                dg[threadIdx.x] = E;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!(P.dbg & 1u)) fuse_apply_rounds<BLOCK, TT, CAM, TOL>(tile, lut, camtab, xm, P, ops, ops_asm, base, dg, (unsigned)t);
        amp_t v[4];
#pragma unroll
        for (unsigned k = 0; k < 4; k++) v[k] = tile[ld_k[k] | ld_t];          // (store order: ascending OUTPUT positions)
        if (!(P.dbg & 2u)) {
#pragma unroll
            for (unsigned k = 0; k < 4; k++) __builtin_nontemporal_store(v[k], go + st_k[k]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K6g  the generated first pass of a circuit by COLUMNS (FusePass::gen = 2; bit-exact).  Behind a circuit front only a few of
// the 2^M low index values are populated -- the residues of the multiply ladder's orbit (C = 21, a = 2: six of 32) -- and no
// gate of an inverse QFT touches the M register, so an amplitude's low bits never change: the tile of 2^12 amplitudes =
// 16 values of the four lowest M-register bits x 2^8 combinations of eight hot bits falls into 16 independent COLUMNS of
// which at most a handful hold anything but +0.  The workgroup (256 threads = the 2^8 hot combinations) generates the tile
// as GenFront says, keeps ONLY the populated columns in LDS (maxcols of them, host-computed bound from the orbit; 257
// elements apart so that the store's column-major reads spread over the banks), and each wave takes whole columns through
// all rounds of the pass on its own: a column is 64 lanes x 4 registers, so there is no barrier between rounds and no wave
// without work -- the k_fused_rounds form of this pass spends half of its waves on zeros (zskip finds a quarter or three
// quarters of a 2^10 tile empty) or, on 2^12 tiles, leaves 13 of 16 waves of a 64-KiB workgroup idle.  Empty columns are
// stored as +0 straight from registers.  Same records, same walk (fuse_round_item), same bits.
// ---------------------------------------------------------------------------
#define QCX_COL_STRIDE 257u
// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL stores
// (s_waitcnt vmcnt(0) in front of s_barrier): inside k_gen_cols' tile loop that made every wave sit out the round trip of the
// tile it had just stored before it could generate the next one -- and since all workgroups of a CU run in step, the chip
// alternated between computing and storing.  The loop has no vector load from global memory (records: scalar loads, masks and
// tables: LDS), so nothing in it needs the stores to have landed; they drain while the next tile is generated and walked.
#define QCX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// x mod C with the host's Cinv = floor(2^32 / C): the quotient estimate is at most one short
__device__ __forceinline__ unsigned gen_mod(unsigned x, unsigned C, unsigned Cinv)
{
    unsigned r = x - __umulhi(x, Cinv) * C;
    return r >= C ? r - C : r;
}
// gen = 2: on the register itself -- tile = its four lowest M-register bits x 8 hot bits, columns = the values of those four
//          bits, the populated ones found per tile (a tile whose other M-register bits, outside the tile, do not match a
//          residue holds nothing there);
// gen = 3: on the VIRTUAL register of a compact chain (qcx_fuse.inc.h) -- index = [L-register bits][column], cb column bits,
//          column j = the amplitudes whose M register reads GenFront::orbit[j]: every column is kept, the store writes the
//          compact layout (the real index of a tile's base is its L part shifted up by M).
// Launched with 64 * W threads, W = 4 ... 8 (fuse_cols_waves; 4 by default): the first 256 threads generate and store the tile,
// further waves only walk columns.
template <int OCC, bool TOL = false>     // TOL: the pass holds merged diagonals (tolerance mode): fast rounds run as in k_fused_rounds<.., TOL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(OCC))) void k_gen_cols(
    amp_t *__restrict__ amp_out, unsigned n, FusePass P, const FuseOp *__restrict__ ops, uint64_t ntiles, const FuseOp *ops_asm)
{
    constexpr unsigned BLOCK = 256;
    const unsigned nthreads = blockDim.x, nwaves = blockDim.x >> 6;
    const bool worker = threadIdx.x < BLOCK;                                      // takes part in generating and storing
    extern __shared__ __attribute__((aligned(16))) unsigned char qcx_lds_raw[];
    const unsigned maxcols = P.zpad;
    amp_t *cols = reinterpret_cast<amp_t *>(qcx_lds_raw);                         // [maxcols][QCX_COL_STRIDE]
    unsigned char *behind = reinterpret_cast<unsigned char *>(cols + maxcols * QCX_COL_STRIDE);
    uint64_t *xm = reinterpret_cast<uint64_t *>(behind + P.xm_off);
    for (unsigned b = threadIdx.x; b < P.xm_cnt + 66u; b += nthreads) xm[b] = b < P.xm_cnt ? ops[b].mask : 0;    // + padding
    __shared__ unsigned s_mask;
    // tolerance mode: [E_out slot per diagonal][G tables, 48 entries per diagonal] in LDS; the tables are staged once
    amp_t *dg = reinterpret_cast<amp_t *>(behind + P.dg_lds_off);
    const amp_t *dg_area = reinterpret_cast<const amp_t *>(ops + P.dg_rec_off);
    if constexpr (TOL)
        for (unsigned b = threadIdx.x; b < P.dg_cnt * 48u; b += nthreads) dg[P.dg_cnt + b] = dg_area[3u * P.dg_cnt + b];
    const GenFront *GF = reinterpret_cast<const GenFront *>(ops + P.gen_rec_off);
    const bool compact = P.gen == 3;
    const unsigned cb = compact ? GF->cb : 4u, TT = cb + 8u, cmask = (1u << cb) - 1u;
    // this thread IS hot combination h = threadIdx.x (tile-local bits cb .. cb + 7; slot h of the generated fill)
    const unsigned h = threadIdx.x & 255u, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned phot = GF->f0;                                                       // residue factor of the tile-local controls, times f0
    if (!worker || (h & GF->sfm) != GF->sbv) phot = 0xffffu;
    else if (GF->C)
        for (unsigned g = 0; g < GF->ncam; g++)
            if (GF->camloc[g] != 0xff && ((h >> GF->camloc[g]) & 1u)) phot = gen_mod(phot * GF->camA[g], GF->C, GF->Cinv);
    const uint32_t parH = compact ? ((uint32_t)__builtin_popcount(h & GF->sgn_slots) & 1u) : ((gen_pack(h << 4, GF, TT) >> 24) & 1u);
    const uint64_t st_t = fuse_spread(threadIdx.x, P.st_pos, TT);
    const unsigned ld_t = (unsigned)fuse_spread(threadIdx.x, P.st_loc, TT);
    // store order: element k * 256 + thread of the OUTPUT order; the bits of k spread linearly (uniform values)
    unsigned bl[4]; uint64_t bp[4];
#pragma unroll
    for (unsigned b = 0; b < 4; b++) { bl[b] = (unsigned)fuse_spread(256u << b, P.st_loc, TT); bp[b] = fuse_spread(256u << b, P.st_pos, TT); }
    if (threadIdx.x == 0) s_mask = 0;
    // the front's constants, read ONCE: amp_out may alias the record buffer as far as the compiler knows, so every GF-> access
    // inside the tile loop was a fresh scalar load -- some twenty per tile, half of them in a dependent chain (the orbit search, the
    // factor tables): 0.73 ms of the pass at n = 30 was this latency (generation alone, gates and stores skipped)
    const uint64_t gf_fixed_out = GF->fixed_out, gf_fixed_val = GF->basis & GF->fixed_out, gf_sign_out = GF->sign_out;
    const double gf_v = GF->v;
    const unsigned gf_C = GF->C, gf_Cinv = GF->Cinv, gf_present = GF->present, gf_M = GF->M, gf_lowout = GF->lowout_mask, gf_ncols = GF->ncols;
    // compact form: column of a residue, tabulated once per workgroup (residues < C <= 4096; 0xff: not on the orbit)
    unsigned char *col_of = reinterpret_cast<unsigned char *>(behind + P.gen_lds_off);
    if (compact) {
        for (unsigned f = threadIdx.x; f < (gf_C ? gf_C : (1u << gf_M)); f += nthreads) {
            unsigned char c = 0xff;
            for (unsigned jj = 0; jj < gf_ncols; jj++) if (GF->orbit[jj] == f) c = (unsigned char)jj;
            col_of[f] = c;
        }
    }
    __syncthreads();
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t base_in = fuse_deposit(t, P.seg_in, P.nseg_in);
        uint64_t base = base_in, base_out = base_in;
        if (P.chained) { base = fuse_deposit(t, P.seg_lg, P.nseg_lg); base_out = fuse_deposit(t, P.seg_out, P.nseg_out); }
        amp_t *go = amp_out + (base_out | st_t);
        const uint64_t rbase = compact ? (base >> cb) << gf_M : base;            // the tile's base as an index of the REAL register
        // ---- generate: this thread's one candidate amplitude (hot combination h): its residue, column and sign ----------
        unsigned f = 0xffffu;
        if ((rbase & gf_fixed_out) == gf_fixed_val && phot != 0xffffu) {
            f = phot;
            if (gf_C) {
                unsigned pt = 1;                                                  // controls outside the tile: one factor per tile
#pragma unroll
                for (unsigned k = 0; k < 5; k++)
                    if ((gf_present >> k) & 1u) pt = gen_mod(pt * (unsigned)GF->tabP[k][(unsigned)((rbase >> (8u * k)) & 255u)], gf_C, gf_Cinv);
                f = gen_mod(f * pt, gf_C, gf_Cinv);
            }
            if (!compact && (f & gf_lowout) != ((uint32_t)base & gf_lowout)) f = 0xffffu;   // its low bits outside the tile belong to another tile
        }
        if constexpr (TOL) {
            // E_out of this tile for every diagonal of the pass: one thread each (read by the rounds, behind the barriers below)
            if (threadIdx.x < P.dg_cnt) {
                const DiagInfo *info = reinterpret_cast<const DiagInfo *>(dg_area) + threadIdx.x;
                const uint32_t present = info->present;
                amp_t E; E.x = info->kc; E.y = info->ks;
#pragma unroll
                for (unsigned k = 0; k < 5; k++)
                    if ((present >> k) & 1u) cmul_tol(E, dg_area[info->field_off[k] + (unsigned)((base >> (8u * k)) & 255u)]);
                dg[threadIdx.x] = E;
            }
        }
        unsigned mycol = f & 15u;
        if (compact) {                                                            // column = the residue's place in the orbit
            mycol = f != 0xffffu ? (unsigned)col_of[f] : 0xffu;
            if (mycol == 0xffu) f = 0xffffu;                                      // (cannot happen: the orbit holds every reachable residue)
        } else if (f != 0xffffu) atomicOr(&s_mask, 1u << mycol);
        QCX_LDS_BARRIER();
        const unsigned mask = compact ? (1u << gf_ncols) - 1u : s_mask;
        const unsigned ncol = (unsigned)__builtin_popcount(mask);
        if (worker) for (unsigned s = 0; s < ncol; s++) { amp_t z; z.x = 0.0; z.y = 0.0; cols[s * QCX_COL_STRIDE + h] = z; }
        if (f != 0xffffu) {
            const uint32_t par = (parH ^ (uint32_t)__builtin_popcountll(rbase & gf_sign_out)) & 1u;
            amp_t a; a.x = par ? -gf_v : gf_v; a.y = 0.0;
            cols[(unsigned)__builtin_popcount(mask & ((1u << mycol) - 1u)) * QCX_COL_STRIDE + h] = a;
        }
        QCX_LDS_BARRIER();
        // ---- rounds: every wave takes whole columns through the pass on its own --------------------------------------
        // (which columns a wave takes rotates with the tile number: six columns over four waves are 2 + 2 + 1 + 1, a wave sits on one
        //  SIMD for good, and with a fixed assignment SIMDs 0 and 1 of every CU would carry twice the work of SIMDs 2 and 3)
        if (!(P.dbg & 1u))
        for (unsigned s = (wave + (unsigned)t + (unsigned)(t >> 2)) % nwaves; s < ncol; s += nwaves) {
            unsigned cpat = 0;                                                    // the column's low-bit value: the s-th set bit of mask
            { unsigned m = mask; for (unsigned k = 0; k < s; k++) m &= m - 1u; cpat = (unsigned)__builtin_ctz(m); }
            amp_t *col = cols + s * QCX_COL_STRIDE;
            unsigned i = 0;
            while (i < P.nops) {
                const unsigned rb0 = (ops[i].a & 0xffu) - cb, rb1 = ((ops[i].a >> 8) & 0xffu) - cb;     // (hot bits; the host checks)
                const unsigned cnt = (unsigned)ops[i].mask;
                const unsigned hh = (unsigned)insert_zero(insert_zero(lane, rb0), rb1);
                const unsigned p = (hh << cb) | cpat;                                     // logical tile-local index (the walk tests its bits)
                const unsigned l0 = hh, l1 = hh | (1u << rb0), l2 = hh | (1u << rb1), l3 = l1 | l2;
                if (TOL && (ops[i].type & 0xffu) == FUSE_QROUND) {
                    const amp_t *gtab = dg + P.dg_cnt;
                    const uint32_t sA = ops[i + 1].type, sB = ops[i + 1].a;
                    if (sA & 1u) { if (sB & 1u) qround_run_col<true, true>(col, l0, l1, l2, l3, p, rb0 + cb, rb1 + cb, sA, sB, dg, gtab);
                                   else         qround_run_col<true, false>(col, l0, l1, l2, l3, p, rb0 + cb, rb1 + cb, sA, sB, dg, gtab); }
                    else         { if (sB & 1u) qround_run_col<false, true>(col, l0, l1, l2, l3, p, rb0 + cb, rb1 + cb, sA, sB, dg, gtab);
                                   else         qround_run_col<false, false>(col, l0, l1, l2, l3, p, rb0 + cb, rb1 + cb, sA, sB, dg, gtab); }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    i += 2;
                    continue;
                }
                Quad q;
                { const amp_t v0 = col[l0], v1 = col[l1], v2 = col[l2], v3 = col[l3];
                  q.x0 = v0.x; q.y0 = v0.y; q.x1 = v1.x; q.y1 = v1.y; q.x2 = v2.x; q.y2 = v2.y; q.x3 = v3.x; q.y3 = v3.y; }
                const bool has_h = (ops[i].a >> 16) & 1u;
                unsigned o = i + 1;
                const unsigned oend = i + cnt;
                const __attribute__((address_space(3))) uint64_t *xl = (const __attribute__((address_space(3))) uint64_t *)xm + o + 1 + lane;
                uint32_t xaddr = (uint32_t)(uintptr_t)xl;
                uint32_t ty = ops[o].type;
                uint64_t m = *xl;
                do {
                    const unsigned rc = ty >> 16;
                    const uint64_t live = __builtin_amdgcn_ballot_w64((base & m) == m) & (rc >= 64 ? ~(uint64_t)0 : ((uint64_t)1 << rc) - 1);
                    uint32_t tn; uint64_t mn;
                    fuse_round_item(q, ops_asm + o, live, p, ty, xaddr, tn, mn);
                    o += 1 + rc; ty = tn; m = mn;
                } while (o <= oend);
                if (has_h) {
                    q.x0 += 0.0; q.y0 += 0.0; q.x1 += 0.0; q.y1 += 0.0; q.x2 += 0.0; q.y2 += 0.0; q.x3 += 0.0; q.y3 += 0.0;
                }
                { amp_t v0, v1, v2, v3;
                  v0.x = q.x0; v0.y = q.y0; v1.x = q.x1; v1.y = q.y1; v2.x = q.x2; v2.y = q.y2; v3.x = q.x3; v3.y = q.y3;
                  col[l0] = v0; col[l1] = v1; col[l2] = v2; col[l3] = v3; }
                // (the column is this wave's own: its LDS accesses are served in order, no workgroup barrier between rounds)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                i += 1 + cnt;
            }
        }
        QCX_LDS_BARRIER();
        if (threadIdx.x == 0) s_mask = 0;                                         // (read by everyone before the barrier above)
        // ---- store the whole tile: populated columns from LDS, the others as +0 ------------------------------------------
        if (worker && !(P.dbg & 2u)) {
            const unsigned nk = 1u << cb;
#pragma unroll
            for (unsigned k = 0; k < 16; k++) {
                if (k >= nk) break;
                unsigned elk = 0; uint64_t offk = 0;
#pragma unroll
                for (unsigned b = 0; b < 4; b++) if ((k >> b) & 1u) { elk |= bl[b]; offk |= bp[b]; }
                const unsigned el = elk | ld_t;                                   // tile-local element, in store order
                const unsigned c = el & cmask, eh = el >> cb;
                amp_t v; v.x = 0.0; v.y = 0.0;
                if ((mask >> c) & 1u) v = cols[(unsigned)__builtin_popcount(mask & ((1u << c) - 1u)) * QCX_COL_STRIDE + eh];
                __builtin_nontemporal_store(v, go + offk);
            }
        }
        QCX_LDS_BARRIER();
    }
}

// the end of a compact chain: the real register from its compact form.  real[(l << M) | f] = compact[(l << cb) | j] where
// f = orbit[j], and +0 for every other f; the whole register is written (16 * 2^n bytes).  A workgroup takes 64 blocks at a
// time: their compact rows (64 * 2^cb amplitudes, contiguous) are staged in LDS with coalesced loads, then the 64 * 2^M real
// amplitudes leave as coalesced nontemporal stores (a per-amplitude gather of the compact form kept a dependent load in front
// of every store: 8.0 ms at n = 30 against 3 ms for the bytes).  direct: no stage -- a thread loads the (few) compact sources of
// its next 8 real amplitudes first, then stores the 8: 3.76 against 4.02 ms (the default).
struct ExpandParams { unsigned M, cb, ncols; uint16_t orbit[16]; };
__global__ __launch_bounds__(256) void k_expand_compact(const amp_t *__restrict__ compact, amp_t *__restrict__ real, uint64_t nchunks, ExpandParams E, int direct)
{
    __shared__ __attribute__((aligned(16))) amp_t stage[64 << 4];
    __shared__ unsigned char lut[1 << 12];
    const unsigned M = E.M, cb = E.cb, lowmask = (1u << M) - 1u;
    for (unsigned f = threadIdx.x; f <= lowmask; f += 256) {
        unsigned char j = 0xff;
        for (unsigned k = 0; k < E.ncols; k++) if (E.orbit[k] == f) j = (unsigned char)k;
        lut[f] = j;
    }
    __syncthreads();
    if (direct) {
        // variant without the LDS stage: 8 real amplitudes per thread, their (few) compact sources loaded first, then 8 stores
        for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
            const amp_t *from = compact + ((chunk * 64) << cb);
            amp_t *to = real + ((chunk * 64) << M);
            for (unsigned e0 = threadIdx.x; e0 < (64u << M); e0 += 256u * 8u) {
                amp_t v[8];
#pragma unroll
                for (unsigned k = 0; k < 8; k++) {
                    const unsigned e = e0 + k * 256u;
                    const unsigned j = e < (64u << M) ? lut[e & lowmask] : 0xffu;
                    v[k].x = 0.0; v[k].y = 0.0;
                    if (j != 0xffu) v[k] = from[((e >> M) << cb) | j];
                }
#pragma unroll
                for (unsigned k = 0; k < 8; k++) if (e0 + k * 256u < (64u << M)) __builtin_nontemporal_store(v[k], to + e0 + k * 256u);
            }
        }
        return;
    }
    for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const amp_t *from = compact + ((chunk * 64) << cb);
        for (unsigned e = threadIdx.x; e < (64u << cb); e += 256) stage[e] = __builtin_nontemporal_load(from + e);
        __syncthreads();
        amp_t *to = real + ((chunk * 64) << M);
        for (unsigned e = threadIdx.x; e < (64u << M); e += 256) {
            const unsigned j = lut[e & lowmask];
            amp_t v; v.x = 0.0; v.y = 0.0;
            if (j != 0xffu) v = stage[((e >> M) << cb) | j];
            __builtin_nontemporal_store(v, to + e);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// K6t-8  tolerance mode, RADIX-8 fast rounds on 2^12-amplitude tiles: a thread keeps the 8 amplitudes that differ in THREE
// register bits, a round is up to three steps  H(x) [D(x)]  -- three Hadamards and their merged diagonals per LDS round trip
// and per barrier instead of two, with 512-thread workgroups (8 waves) on a tile that holds 8 hot bits: the n = 28 inverse
// QFT in 3 passes instead of 4.  Passes whose rounds are ALL of that shape and that hold no modular multiply take this
// kernel; everything else takes k_fused_rounds.
// Records: FUSE_QROUND3  a = rb0 | rb1 << 8 | rb2 << 16 | steps << 24, mask = 1 (one record follows), then one record with
// the step words in type, a and the low half of mask.  Step word: bits 0-1 = which register bit (0..2), bit 2 = a diagonal
// follows the H, bits 8-15 its slot, bits 16-18 its groups, bit 19 / 20 = the lower / higher of the OTHER two register
// bits is one of its targets.
// ---------------------------------------------------------------------------
enum : uint32_t { FUSE_QROUND3 = 8 };

// Hadamard butterfly WITHOUT its 1/sqrt(2): the radix-8 passes multiply every amplitude by (1/sqrt 2)^(Hadamards of the pass)
// once, when the tile is stored (every amplitude of a tile takes part in every Hadamard of the pass)
__device__ __forceinline__ void h_butterfly_unscaled(amp_t &a, amp_t &b)
{
    const double ar = a.x, ai = a.y;
    a.x = ar + b.x;  a.y = ai + b.y;
    b.x = ar - b.x;  b.y = ai - b.y;
}

// the exact Hadamard butterfly without its trailing "+ 0.0" (the pass canonicalises once, when the tile is stored)
__device__ __forceinline__ void h_butterfly_noz(amp_t &a, amp_t &b)
{
    const double s = QCX_SQRT1_2;
    const double t0r = s * a.x, t0i = s * a.y, t1r = s * b.x, t1i = s * b.y;
    a.x = t0r + t1r;  a.y = t0i + t1i;
    b.x = t0r - t1r;  b.y = t0i - t1i;
}

template <int R, bool EXACT>        // step on register bit R of an 8-amplitude register file (index bit j of v[] = register bit j)
__device__ __forceinline__ void q3_step(uint32_t s, unsigned p, unsigned rbo1, unsigned rbo2, amp_t (&v)[8], const amp_t *dg, const amp_t *gtab)
{
    constexpr int O1 = (R == 0) ? 1 : 0, O2 = (R == 2) ? 1 : 2;          // the other two register bits, ascending
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            if constexpr (EXACT) h_butterfly_noz(v[(a << O1) | (b << O2)], v[(1 << R) | (a << O1) | (b << O2)]);
            else h_butterfly_unscaled(v[(a << O1) | (b << O2)], v[(1 << R) | (a << O1) | (b << O2)]);
        }
    if constexpr (EXACT) return;
    if (s & 4u) {
        const unsigned slot = (s >> 8) & 0xffu;
        const amp_t *G = gtab + slot * 48u;
        amp_t F = dg[slot];
        if (s & (1u << 16)) cmul_tol(F, G[p & 15u]);
        if (s & (1u << 17)) cmul_tol(F, G[16u + ((p >> 4) & 15u)]);
        if (s & (1u << 18)) cmul_tol(F, G[32u + (p >> 8)]);
        amp_t F1 = F, F2 = F;
        if (s & (1u << 19)) cmul_tol(F1, G[16u * (rbo1 >> 2) + (1u << (rbo1 & 3u))]);
        if (s & (1u << 20)) { const amp_t w2 = G[16u * (rbo2 >> 2) + (1u << (rbo2 & 3u))]; cmul_tol(F2, w2); amp_t F3 = F1; cmul_tol(F3, w2);
                              cmul_tol(v[(1 << R) | (1 << O1) | (1 << O2)], F3); }
        else cmul_tol(v[(1 << R) | (1 << O1) | (1 << O2)], F1);
        cmul_tol(v[1 << R], F);
        cmul_tol(v[(1 << R) | (1 << O1)], F1);
        cmul_tol(v[(1 << R) | (1 << O2)], F2);
    }
}

template <int RA, int RB, int RC, bool EXACT>
__device__ __forceinline__ void q3_run(amp_t *tile, unsigned p, const unsigned (&rb)[3], unsigned nsteps, uint32_t sA, uint32_t sB, uint32_t sC,
                                       const amp_t *dg, const amp_t *gtab)
{
    amp_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = tile[p | ((j & 1u) << rb[0]) | (((j >> 1) & 1u) << rb[1]) | ((unsigned)(j >> 2) << rb[2])];
    q3_step<RA, EXACT>(sA, p, rb[RA == 0 ? 1 : 0], rb[RA == 2 ? 1 : 2], v, dg, gtab);
    __builtin_amdgcn_sched_barrier(0);
    if (nsteps > 1) q3_step<RB, EXACT>(sB, p, rb[RB == 0 ? 1 : 0], rb[RB == 2 ? 1 : 2], v, dg, gtab);
    __builtin_amdgcn_sched_barrier(0);
    if (nsteps > 2) q3_step<RC, EXACT>(sC, p, rb[RC == 0 ? 1 : 0], rb[RC == 2 ? 1 : 2], v, dg, gtab);
#pragma unroll
    for (int j = 0; j < 8; j++) tile[p | ((j & 1u) << rb[0]) | (((j >> 1) & 1u) << rb[1]) | ((unsigned)(j >> 2) << rb[2])] = v[j];
}

// EXACT = true: the same radix-8 structure for passes of nothing but Hadamards in the BIT-EXACT modes (the fused Hadamard
// sweep): exact butterflies (separate roundings, no FMA), canonical zeros once at the store; no tables, no diagonals.
template <int BLOCK, int TT, int OCC, bool EXACT = false, bool GEN = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(OCC))) void k_fused_q3(
    const amp_t *amp, amp_t *amp_out, unsigned n, FusePass P, const FuseOp *__restrict__ ops, uint64_t ntiles)
{
    static_assert((1u << TT) == 8u * BLOCK, "radix-8 rounds: 8 amplitudes per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char qcx_lds_raw[];
    amp_t *tile = reinterpret_cast<amp_t *>(qcx_lds_raw);
    constexpr unsigned tsize = 1u << TT;
    amp_t *dg = reinterpret_cast<amp_t *>(reinterpret_cast<unsigned char *>(tile + tsize) + P.dg_lds_off);
    const amp_t *dg_area = reinterpret_cast<const amp_t *>(ops + P.dg_rec_off);
    bool staged = EXACT;                 // (the diagonals' G tables are staged once per workgroup, behind the first tile's fill: see below)
    const amp_t *gtab = dg + P.dg_cnt;
    const uint64_t off_t = fuse_spread(threadIdx.x, P.in_pos, TT);
    uint64_t off_k[8], st_k[8];
    unsigned ld_k[8];
#pragma unroll
    for (unsigned k = 0; k < 8; k++) {
        off_k[k] = fuse_spread(k * BLOCK, P.in_pos, TT);
        st_k[k] = fuse_spread(k * BLOCK, P.st_pos, TT);
        ld_k[k] = (unsigned)fuse_spread(k * BLOCK, P.st_loc, TT);
    }
    const uint64_t st_t = fuse_spread(threadIdx.x, P.st_pos, TT);
    const unsigned ld_t = (unsigned)fuse_spread(threadIdx.x, P.st_loc, TT);
    const unsigned wbase = (threadIdx.x >> 6) * 64;
    // gen: the tiles are generated (the circuit front on a basis state that was never written), not read
    const GenFront *GF = reinterpret_cast<const GenFront *>(ops + P.gen_rec_off);
    unsigned short *gen_phot = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(tile + tsize) + P.gen_lds_off);
    unsigned short *gen_res = gen_phot + 512;
    uint32_t packT = 0, packK[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (GEN) {
        gen_setup<BLOCK>(GF, gen_phot);
        packT = gen_pack(threadIdx.x, GF, TT);
#pragma unroll
        for (unsigned k = 0; k < 8; k++) packK[k] = gen_pack(k * BLOCK, GF, TT);
        __syncthreads();
    }
    const unsigned slog = (P.dbg >> 12) & 15u;
    for (uint64_t b0 = blockIdx.x; b0 < ntiles; b0 += gridDim.x) {
        const uint64_t t = fuse_stream_tile(b0, n - TT, slog, (P.dbg >> 16) & 31u);
        const uint64_t base_in = fuse_deposit(t, P.seg_in, P.nseg_in);
        uint64_t base = base_in, base_out = base_in;               // logical base (gate records), output base
        if (P.chained) { base = fuse_deposit(t, P.seg_lg, P.nseg_lg); base_out = fuse_deposit(t, P.seg_out, P.nseg_out); }
        const amp_t *g = amp + (base_in | off_t);
        amp_t *go = amp_out + (base_out | st_t);
        if constexpr (GEN) gen_tile<BLOCK, 8>(GF, tile, gen_phot, gen_res, base, packT, packK);
        else if (!(P.dbg & 4u)) {
#pragma unroll
            for (unsigned k = 0; k < 8; k++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + off_k[k]),
                                                 (__attribute__((address_space(3))) void *)(tile + k * BLOCK + wbase), 16, 0, 2);
        }
        if (!staged) {          // round 5: a workgroup usually takes ONE tile -- the table staging used to sit in front of its fill, a
            staged = true;      // memory latency per tile with nothing else in flight (the tables are read after the barrier below)
            for (unsigned b = threadIdx.x; b < P.dg_cnt * 48u; b += BLOCK) dg[P.dg_cnt + b] = dg_area[3u * P.dg_cnt + b];
        }
        if (!EXACT && threadIdx.x < P.dg_cnt) {           // E_out of this tile for every diagonal of the pass, while the fill is in flight
            const DiagInfo *info = reinterpret_cast<const DiagInfo *>(dg_area) + threadIdx.x;
            const uint32_t present = info->present;
            amp_t E; E.x = info->kc; E.y = info->ks;
#pragma unroll
            for (unsigned f = 0; f < 5; f++)
                if ((present >> f) & 1u) cmul_tol(E, dg_area[info->field_off[f] + (unsigned)((base >> (8u * f)) & 255u)]);
            dg[threadIdx.x] = E;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!(P.dbg & 1u)) {
            for (unsigned i = 0; i < P.nops; i += 2) {
                const uint32_t a = ops[i].a;
                const unsigned rb[3] = {a & 0xffu, (a >> 8) & 0xffu, (a >> 16) & 0xffu};
                const unsigned nsteps = (a >> 24) & 3u;
                const uint32_t sA = ops[i + 1].type, sB = ops[i + 1].a, sC = (uint32_t)ops[i + 1].mask;
                const unsigned p = (unsigned)insert_zero(insert_zero(insert_zero(threadIdx.x, rb[0]), rb[1]), rb[2]);
                // (which register bit each step works on is a compile-time fact of the variant: see qround_run)
                const unsigned perm = (sA & 3u) | ((sB & 3u) << 2) | ((sC & 3u) << 4);
                switch (perm) {
                case 0x06: q3_run<2, 1, 0, EXACT>(tile, p, rb, nsteps, sA, sB, sC, dg, gtab); break;     // sA=2, sB=1, sC=0: descending (Q:682-689)
                case 0x12: q3_run<2, 0, 1, EXACT>(tile, p, rb, nsteps, sA, sB, sC, dg, gtab); break;
                case 0x09: q3_run<1, 2, 0, EXACT>(tile, p, rb, nsteps, sA, sB, sC, dg, gtab); break;
                case 0x21: q3_run<1, 0, 2, EXACT>(tile, p, rb, nsteps, sA, sB, sC, dg, gtab); break;
                case 0x18: q3_run<0, 2, 1, EXACT>(tile, p, rb, nsteps, sA, sB, sC, dg, gtab); break;
                default:   q3_run<0, 1, 2, EXACT>(tile, p, rb, nsteps, sA, sB, sC, dg, gtab); break;     // 0x24: ascending
                }
                __syncthreads();
            }
        }
        amp_t v[8];
        const double sc = (P.dbg & 1u) ? 1.0 : P.tol_scale;
#pragma unroll
        for (unsigned k = 0; k < 8; k++) {
            v[k] = tile[ld_k[k] | ld_t];                                    // (store order: ascending OUTPUT positions)
            if constexpr (EXACT) { v[k].x += 0.0; v[k].y += 0.0; }          // the reference's canonical zeros, once per pass
            else { v[k].x *= sc; v[k].y *= sc; }
        }
        if (!(P.dbg & 2u)) {
#pragma unroll
            for (unsigned k = 0; k < 8; k++) __builtin_nontemporal_store(v[k], go + st_k[k]);
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------
// K6x  the EXACT walk on 8 amplitudes per thread (round 5; bit-exact).  Same records, same per-amplitude arithmetic in the same
// order as the radix-4 rounds of k_fused_rounds -- a phase is the four products and two sums of rotate_amp without FMA, an H the
// butterfly of h_butterfly, the canonical "+ 0.0" once per round / run -- but a round carries THREE register bits: a thread keeps
// the 8 amplitudes that differ in them, three Hadamards and their phase runs cost one LDS round trip and one barrier, and a tile
// of 2^12 amplitudes needs 512 threads instead of 1024 (8 hot bits per pass next to c = 4: the n = 28 inverse QFT in 3 passes
// instead of 4).  Round 3 tried this in C++ and lost to the compiler's register copies (21 ms against 12); here the WHOLE round
// -- the eight LDS reads, the item loop, every rotation, the LDS writes -- is ONE asm statement on FIXED registers:
//     v[32:63]   the 8 amplitudes: register r (bit j of r = the round's register bit j) is x = v[32+4r : 33+4r], y = v[34+4r : 35+4r]
//     v[24:31]   four temporaries; v[20:21] this lane's outside-tile mask of the current item (the next item's is fetched into it
//                as soon as the current one has been tested); v18, v19 scratch
//     s[72:79] / s[80:87]  the two record blocks (one s_load_dwordx8 per gate, the next gate's in flight); s[88:89] the item
//     pointer; s[90:91] live gates of the run; s[92:93] saved EXEC; s94 scratch; s95 / s96 header of this / the next item;
//     s97 records left in the round; s98 gates of the item
// all named on the clobber list (the kernel is built for 4 waves per SIMD -- what its LDS tile allows anyway -- so they lie
// inside the allocator's budget).  Records of a round:
//   FUSE_ROUND8  a = rb0 | rb1 << 8 | rb2 << 16 | (round contains an H) << 24 | (no barrier needed in front of it) << 25,  mask = records that follow,
//                c (as 64 bits) = the tile-local bit that thread bit i rides on, 4 bits each (the lane and wave numbers are laid
//                over the non-register bits in the order the host likes best)
//   item H       type = FUSE_H | (32 | j) << 8                        H on register bit j
//   item run     type = FUSE_PRUN | (pat | canon << 4) << 8 | gates << 16   1..63 gates that rotate the same registers:
//                pat 0 = all eight; 1 / 2 / 3 = those with register bit 0 / 1 / 2 set; 4 / 5 / 6 = with bits {0,1} / {0,2} / {1,2} set
//   gate of run  as in the radix-4 form: a = tile-local mask without the register bits, mask = outside-tile bits, c, s; conditions on
//                the tile bits the round's wave number rides on sit in mask bits 48 .. (bit 48 + k: thread bit 6 + k)
// ---------------------------------------------------------------------------
enum : uint32_t { FUSE_ROUND8 = 9 };

#define X8X0 "v[32:33]"
#define X8Y0 "v[34:35]"
#define X8X1 "v[36:37]"
#define X8Y1 "v[38:39]"
#define X8X2 "v[40:41]"
#define X8Y2 "v[42:43]"
#define X8X3 "v[44:45]"
#define X8Y3 "v[46:47]"
#define X8X4 "v[48:49]"
#define X8Y4 "v[50:51]"
#define X8X5 "v[52:53]"
#define X8Y5 "v[54:55]"
#define X8X6 "v[56:57]"
#define X8Y6 "v[58:59]"
#define X8X7 "v[60:61]"
#define X8Y7 "v[62:63]"
#define X8T0 "v[24:25]"
#define X8T1 "v[26:27]"
#define X8T2 "v[28:29]"
#define X8T3 "v[30:31]"
// two rotations interleaved on the four temporaries (the same sequence as QCX_ROT2)
#define X8_ROT2(C, S, XA, YA, XB, YB)                                \
    "v_mul_f64 " X8T0 ", " C ", " XA "\n\t"                          \
    "v_mul_f64 " X8T1 ", " S ", " YA "\n\t"                          \
    "v_mul_f64 " X8T2 ", " C ", " YA "\n\t"                          \
    "v_mul_f64 " X8T3 ", " S ", " XA "\n\t"                          \
    "v_add_f64 " XA ", " X8T0 ", -" X8T1 "\n\t"                      \
    "v_mul_f64 " X8T0 ", " C ", " XB "\n\t"                          \
    "v_mul_f64 " X8T1 ", " S ", " YB "\n\t"                          \
    "v_add_f64 " YA ", " X8T2 ", " X8T3 "\n\t"                       \
    "v_mul_f64 " X8T2 ", " C ", " YB "\n\t"                          \
    "v_mul_f64 " X8T3 ", " S ", " XB "\n\t"                          \
    "v_add_f64 " XB ", " X8T0 ", -" X8T1 "\n\t"                      \
    "v_add_f64 " YB ", " X8T2 ", " X8T3 "\n\t"
#define X8_R2(C, S, i, j) X8_ROT2(C, S, X8X##i, X8Y##i, X8X##j, X8Y##j)
#define X8_ROTS_0(C, S) X8_R2(C, S, 0, 1) X8_R2(C, S, 2, 3) X8_R2(C, S, 4, 5) X8_R2(C, S, 6, 7)
#define X8_ROTS_1(C, S) X8_R2(C, S, 1, 3) X8_R2(C, S, 5, 7)
#define X8_ROTS_2(C, S) X8_R2(C, S, 2, 3) X8_R2(C, S, 6, 7)
#define X8_ROTS_3(C, S) X8_R2(C, S, 4, 5) X8_R2(C, S, 6, 7)
#define X8_ROTS_4(C, S) X8_R2(C, S, 3, 7)
#define X8_ROTS_5(C, S) X8_R2(C, S, 5, 7)
#define X8_ROTS_6(C, S) X8_R2(C, S, 6, 7)
#define X8_Z1(i) "v_add_f64 " X8X##i ", " X8X##i ", 0\n\t" "v_add_f64 " X8Y##i ", " X8Y##i ", 0\n\t"
#define X8_ZERO_0 X8_Z1(0) X8_Z1(1) X8_Z1(2) X8_Z1(3) X8_Z1(4) X8_Z1(5) X8_Z1(6) X8_Z1(7)
#define X8_ZERO_1 X8_Z1(1) X8_Z1(3) X8_Z1(5) X8_Z1(7)
#define X8_ZERO_2 X8_Z1(2) X8_Z1(3) X8_Z1(6) X8_Z1(7)
#define X8_ZERO_3 X8_Z1(4) X8_Z1(5) X8_Z1(6) X8_Z1(7)
#define X8_ZERO_4 X8_Z1(3) X8_Z1(7)
#define X8_ZERO_5 X8_Z1(5) X8_Z1(7)
#define X8_ZERO_6 X8_Z1(6) X8_Z1(7)
// H between registers A and B (A: the bit clear), without the final "+ 0.0" (the round canonicalises once at its end)
#define X8_HBF(A, B)                                                 \
    "v_mul_f64 " X8T0 ", %[hs], " X8X##A "\n\t"                      \
    "v_mul_f64 " X8T1 ", %[hs], " X8Y##A "\n\t"                      \
    "v_mul_f64 " X8T2 ", %[hs], " X8X##B "\n\t"                      \
    "v_mul_f64 " X8T3 ", %[hs], " X8Y##B "\n\t"                      \
    "v_add_f64 " X8X##A ", " X8T0 ", " X8T2 "\n\t"                   \
    "v_add_f64 " X8Y##A ", " X8T1 ", " X8T3 "\n\t"                   \
    "v_add_f64 " X8X##B ", " X8T0 ", -" X8T2 "\n\t"                  \
    "v_add_f64 " X8Y##B ", " X8T1 ", -" X8T3 "\n\t"
#define X8_LOADREC(BLK)                                              \
    "s_ff1_i32_b64 s94, s[90:91]\n\t"                                \
    "s_bitset0_b64 s[90:91], s94\n\t"                                \
    "s_lshl_b32 s94, s94, 5\n\t"                                     \
    "s_load_dwordx8 " BLK ", s[88:89], s94 offset:0x20\n\t"
// one gate: a tile-local mask word of 0 (the usual case: the target lies outside the tile or on a register bit) skips the lane
// mask and the EXEC round trip
#define X8_GATE(M, ROTS)                                             \
    "s_cmp_eq_u32 " M ", 0\n\t"                                      \
    "s_cbranch_scc1 6f\n\t"                                          \
    "v_and_b32 v18, " M ", %[p]\n\t"                                 \
    "v_cmpx_eq_u32_e32 vcc, " M ", v18\n\t"                          \
    "s_cbranch_execz 1f\n\t"                                         \
    ROTS                                                             \
    "1:\n\t"                                                         \
    "s_mov_b64 exec, s[92:93]\n\t"                                   \
    "s_branch 7f\n\t"                                                \
    "6:\n\t"                                                         \
    ROTS                                                             \
    "7:\n\t"
#define X8_RUN_BODY(R)                                               \
    X8_LOADREC("s[72:79]")                                           \
    "2:\n\t"                                                         \
    "s_waitcnt lgkmcnt(0)\n\t"                                       \
    "s_cmp_eq_u64 s[90:91], 0\n\t"                                   \
    "s_cbranch_scc1 3f\n\t"                                          \
    X8_LOADREC("s[80:87]")                                           \
    X8_GATE("s73", X8_ROTS_##R("s[76:77]", "s[78:79]"))              \
    "s_waitcnt lgkmcnt(0)\n\t"                                       \
    "s_cmp_eq_u64 s[90:91], 0\n\t"                                   \
    "s_cbranch_scc1 4f\n\t"                                          \
    X8_LOADREC("s[72:79]")                                           \
    X8_GATE("s81", X8_ROTS_##R("s[84:85]", "s[86:87]"))              \
    "s_branch 2b\n\t"                                                \
    "3:\n\t"                                                         \
    X8_GATE("s73", X8_ROTS_##R("s[76:77]", "s[78:79]"))              \
    "s_branch 5f\n\t"                                                \
    "4:\n\t"                                                         \
    X8_GATE("s81", X8_ROTS_##R("s[84:85]", "s[86:87]"))              \
    "5:\n\t"                                                         \
    "s_bitcmp1_b32 s95, 12\n\t"                                      \
    "s_cbranch_scc0 99f\n\t"                                         \
    X8_ZERO_##R                                                      \
    "s_branch 99f\n\t"
// address of register r's amplitude: a0 XOR the deltas of its set register bits (v19) -- XOR, not +: the tile sits in LDS under the
// swizzle x8_swz, which is linear over XOR
#define X8_ADDR1(D)        "v_xor_b32 v19, " D ", %[a0]\n\t"
#define X8_ADDR2(D, E)     "v_xor_b32 v19, " D ", %[a0]\n\t" "v_xor_b32 v19, " E ", v19\n\t"
#define X8_ADDR3(D, E, F)  "v_xor_b32 v19, " D ", %[a0]\n\t" "v_xor_b32 v19, " E ", v19\n\t" "v_xor_b32 v19, " F ", v19\n\t"
#define X8_ROUND_ALL                                                 \
    "s_mov_b64 s[88:89], %[item]\n\t"                                \
    "s_mov_b32 s97, %[cnt]\n\t"                                      \
    "s_load_dword s95, s[88:89], 0x0\n\t"                            \
    "ds_read_b64 v[20:21], %[xa]\n\t"                                \
    "ds_read_b128 v[32:35], %[a0]\n\t"                               \
    X8_ADDR1("%[d0]") "ds_read_b128 v[36:39], v19\n\t"               \
    X8_ADDR1("%[d1]") "ds_read_b128 v[40:43], v19\n\t"               \
    X8_ADDR2("%[d0]", "%[d1]") "ds_read_b128 v[44:47], v19\n\t"      \
    X8_ADDR1("%[d2]") "ds_read_b128 v[48:51], v19\n\t"               \
    X8_ADDR2("%[d0]", "%[d2]") "ds_read_b128 v[52:55], v19\n\t"      \
    X8_ADDR2("%[d1]", "%[d2]") "ds_read_b128 v[56:59], v19\n\t"      \
    X8_ADDR3("%[d0]", "%[d1]", "%[d2]") "ds_read_b128 v[60:63], v19\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"                                       \
    /* ---- one item ---- */                                         \
    "90:\n\t"                                                        \
    "s_lshr_b32 s98, s95, 16\n\t"                                    \
    /* what the NEXT item needs first: its header dword and this lane's entry of its outside-tile masks */ \
    "s_add_u32 s94, s98, 1\n\t"                                      \
    "s_lshl_b32 s96, s94, 5\n\t"                                     \
    "s_load_dword s96, s[88:89], s96\n\t"                            \
    "v_lshl_add_u32 %[xa], s94, 3, %[xa]\n\t"                        \
    "s_bitcmp1_b32 s95, 13\n\t"                                      \
    "s_cbranch_scc1 50f\n\t"                                         \
    /* the run's live gates: lane l tests gate l's outside-tile controls against the tile's base */ \
    "v_and_b32 v18, %[blo], v20\n\t"                                 \
    "v_and_b32 v19, %[bhi], v21\n\t"                                 \
    "v_cmp_eq_u64_e64 s[90:91], v[18:19], v[20:21]\n\t"              \
    "ds_read_b64 v[20:21], %[xa]\n\t"                                \
    "s_bfm_b64 s[92:93], s98, 0\n\t"                                 \
    "s_and_b64 s[90:91], s[90:91], s[92:93]\n\t"                     \
    "s_mov_b64 s[92:93], exec\n\t"                                   \
    "s_cmp_eq_u64 s[90:91], 0\n\t"                                   \
    "s_cbranch_scc1 99f\n\t"                                         \
    "s_bfe_u32 s94, s95, 0x30008\n\t"                                \
    "s_cmp_lt_u32 s94, 4\n\t s_cbranch_scc1 40f\n\t"                 \
    "s_cmp_lt_u32 s94, 6\n\t s_cbranch_scc1 41f\n\t s_branch 26f\n\t" \
    "41:\n\t s_cmp_eq_u32 s94, 4\n\t s_cbranch_scc1 24f\n\t s_branch 25f\n\t" \
    "40:\n\t s_cmp_lt_u32 s94, 2\n\t s_cbranch_scc1 42f\n\t s_cmp_eq_u32 s94, 2\n\t s_cbranch_scc1 22f\n\t s_branch 23f\n\t" \
    "42:\n\t s_cmp_eq_u32 s94, 0\n\t s_cbranch_scc1 20f\n\t s_branch 21f\n\t" \
    "20:\n\t" X8_RUN_BODY(0) "21:\n\t" X8_RUN_BODY(1) "22:\n\t" X8_RUN_BODY(2) "23:\n\t" X8_RUN_BODY(3) \
    "24:\n\t" X8_RUN_BODY(4) "25:\n\t" X8_RUN_BODY(5) "26:\n\t" X8_RUN_BODY(6) \
    "50:\n\t"                                                        \
    "ds_read_b64 v[20:21], %[xa]\n\t"                                \
    "s_bfe_u32 s94, s95, 0x20008\n\t"                                \
    "s_cmp_eq_u32 s94, 0\n\t s_cbranch_scc1 51f\n\t"                 \
    "s_cmp_eq_u32 s94, 1\n\t s_cbranch_scc1 52f\n\t"                 \
    X8_HBF(0, 4) X8_HBF(1, 5) X8_HBF(2, 6) X8_HBF(3, 7)              \
    "s_branch 99f\n\t"                                               \
    "51:\n\t"                                                        \
    X8_HBF(0, 1) X8_HBF(2, 3) X8_HBF(4, 5) X8_HBF(6, 7)              \
    "s_branch 99f\n\t"                                               \
    "52:\n\t"                                                        \
    X8_HBF(0, 2) X8_HBF(1, 3) X8_HBF(4, 6) X8_HBF(5, 7)              \
    "99:\n\t"                                                        \
    "s_waitcnt lgkmcnt(0)\n\t"                                       \
    /* on to the next item */                                        \
    "s_add_u32 s94, s98, 1\n\t"                                      \
    "s_sub_u32 s97, s97, s94\n\t"                                    \
    "s_lshl_b32 s94, s94, 5\n\t"                                     \
    "s_add_u32 s88, s88, s94\n\t"                                    \
    "s_addc_u32 s89, s89, 0\n\t"                                     \
    "s_mov_b32 s95, s96\n\t"                                         \
    "s_cmp_lg_u32 s97, 0\n\t"                                        \
    "s_cbranch_scc1 90b\n\t"                                         \
    /* ---- end of the round: canonical zeros if it held an H, then the eight LDS writes ---- */ \
    "s_bitcmp1_b32 %[fl], 0\n\t"                                     \
    "s_cbranch_scc0 98f\n\t"                                         \
    X8_ZERO_0                                                        \
    "98:\n\t"                                                        \
    "ds_write_b128 %[a0], v[32:35]\n\t"                              \
    X8_ADDR1("%[d0]") "ds_write_b128 v19, v[36:39]\n\t"              \
    X8_ADDR1("%[d1]") "ds_write_b128 v19, v[40:43]\n\t"              \
    X8_ADDR2("%[d0]", "%[d1]") "ds_write_b128 v19, v[44:47]\n\t"     \
    X8_ADDR1("%[d2]") "ds_write_b128 v19, v[48:51]\n\t"              \
    X8_ADDR2("%[d0]", "%[d2]") "ds_write_b128 v19, v[52:55]\n\t"     \
    X8_ADDR2("%[d1]", "%[d2]") "ds_write_b128 v19, v[56:59]\n\t"     \
    X8_ADDR3("%[d0]", "%[d1]", "%[d2]") "ds_write_b128 v19, v[60:63]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"

// The tile's LDS layout: element e lives in slot e ^ (bits 4-7 of e) ^ (bits 8-11 of e) -- the two upper nibbles folded onto the
// lowest.  A wave's 64 lanes ride on whatever six tile bits the host picked for the round (x8_assign_maps: bits that few gates of
// the round test), seldom the lowest six; unswizzled, lanes that differ only in high bits would share their 16-byte bank group
// (measured: 74 % of the LDS cycles of the first build were bank conflicts).  The map is an involution and linear over XOR, so
// the fill (LDS-DMA: slot s receives element x8_swz(s)), the rounds and the store all address through XOR of per-bit terms.
__device__ __forceinline__ uint32_t x8_swz(uint32_t e) { return e ^ ((e >> 4) & 15u) ^ ((e >> 8) & 15u); }

// one whole round of the exact walk on this thread's 8 amplitudes (see K6x above).  a0 = LDS byte address of the slot of element p;
// d0..d2 = what a set register bit XORs onto it; xaddr = LDS byte address of this lane's entry of the first item's outside-tile masks;
// item = the first item's record; cnt = records of the round; base = the tile's LOGICAL base index
__device__ __forceinline__ void fuse_round8(uint32_t a0, unsigned p, uint32_t xaddr, uint64_t base, const FuseOp *item, uint32_t cnt,
                                            uint32_t d0, uint32_t d1, uint32_t d2, uint32_t has_h)
{
    const double hs = QCX_SQRT1_2;
    const uint32_t blo = (uint32_t)base, bhi = (uint32_t)(base >> 32);
    asm volatile(X8_ROUND_ALL
        : [xa] "+v"(xaddr)
        : [a0] "v"(a0), [p] "v"(p), [blo] "s"(blo), [bhi] "s"(bhi), [item] "s"(item), [cnt] "s"(cnt),
          [d0] "s"(d0), [d1] "s"(d1), [d2] "s"(d2), [fl] "s"(has_h), [hs] "s"(hs)
        : "memory", "vcc", "scc",
          "v18", "v19", "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",
          "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
          "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
          "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87",
          "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98");
}


// ---------------------------------------------------------------------------
// K6x-t  TOLERANCE MODE on the same shell (round 5; opt-in, NOT bit-exact -- the arithmetic of K6t-8 / k_fused_q3, whose records it
// reads: FUSE_QROUND3 rounds of up to three steps H(x) [D(x)]).  k_fused_q3 is C++ and the compiler's version of a round copies the
// eight amplitudes between register sets (44 % of its vector instructions were moves: the pass was bound by them, 4.3 ms of
// arithmetic next to 4.6 ms of memory for the n = 28 inverse QFT).  Here a round is one asm statement on fixed registers:
//     v[32:63]   the 8 amplitudes; the unscaled butterfly of a pair works in place as a' = a + b, b' = a' - 2 b (an add and an FMA)
//     v[64:67]   F = E_out(tile) x the G-table entries of this thread's tile bits;  v[68:79] the entries, then F w1, F w2, F w1 w2
//     v[24:31]   w1, w2: the factors of the step's other two register bits ((1, 0) where the diagonal does not target them)
//     v[20:23]   temporaries, v19 an address
// The header of a round carries what FUSE_ROUND8 carries: the thread map in c (x8_assign_maps) and -- bit 28 of a -- "no barrier
// needed in front of this round".
// ---------------------------------------------------------------------------
// (generated by a small script: the butterflies of a step per register bit -- in place: a' = a + b, b' = a' - 2 b, one rounding
//  each, no second register set --, the diagonal of a step per register bit, F times a table entry)
#define X8T_H_0 \
    "v_add_f64 v[32:33], v[32:33], v[36:37]\n\t" \
    "v_fma_f64 v[36:37], -2.0, v[36:37], v[32:33]\n\t" \
    "v_add_f64 v[34:35], v[34:35], v[38:39]\n\t" \
    "v_fma_f64 v[38:39], -2.0, v[38:39], v[34:35]\n\t" \
    "v_add_f64 v[40:41], v[40:41], v[44:45]\n\t" \
    "v_fma_f64 v[44:45], -2.0, v[44:45], v[40:41]\n\t" \
    "v_add_f64 v[42:43], v[42:43], v[46:47]\n\t" \
    "v_fma_f64 v[46:47], -2.0, v[46:47], v[42:43]\n\t" \
    "v_add_f64 v[48:49], v[48:49], v[52:53]\n\t" \
    "v_fma_f64 v[52:53], -2.0, v[52:53], v[48:49]\n\t" \
    "v_add_f64 v[50:51], v[50:51], v[54:55]\n\t" \
    "v_fma_f64 v[54:55], -2.0, v[54:55], v[50:51]\n\t" \
    "v_add_f64 v[56:57], v[56:57], v[60:61]\n\t" \
    "v_fma_f64 v[60:61], -2.0, v[60:61], v[56:57]\n\t" \
    "v_add_f64 v[58:59], v[58:59], v[62:63]\n\t" \
    "v_fma_f64 v[62:63], -2.0, v[62:63], v[58:59]\n\t"
#define X8T_H_1 \
    "v_add_f64 v[32:33], v[32:33], v[40:41]\n\t" \
    "v_fma_f64 v[40:41], -2.0, v[40:41], v[32:33]\n\t" \
    "v_add_f64 v[34:35], v[34:35], v[42:43]\n\t" \
    "v_fma_f64 v[42:43], -2.0, v[42:43], v[34:35]\n\t" \
    "v_add_f64 v[36:37], v[36:37], v[44:45]\n\t" \
    "v_fma_f64 v[44:45], -2.0, v[44:45], v[36:37]\n\t" \
    "v_add_f64 v[38:39], v[38:39], v[46:47]\n\t" \
    "v_fma_f64 v[46:47], -2.0, v[46:47], v[38:39]\n\t" \
    "v_add_f64 v[48:49], v[48:49], v[56:57]\n\t" \
    "v_fma_f64 v[56:57], -2.0, v[56:57], v[48:49]\n\t" \
    "v_add_f64 v[50:51], v[50:51], v[58:59]\n\t" \
    "v_fma_f64 v[58:59], -2.0, v[58:59], v[50:51]\n\t" \
    "v_add_f64 v[52:53], v[52:53], v[60:61]\n\t" \
    "v_fma_f64 v[60:61], -2.0, v[60:61], v[52:53]\n\t" \
    "v_add_f64 v[54:55], v[54:55], v[62:63]\n\t" \
    "v_fma_f64 v[62:63], -2.0, v[62:63], v[54:55]\n\t"
#define X8T_H_2 \
    "v_add_f64 v[32:33], v[32:33], v[48:49]\n\t" \
    "v_fma_f64 v[48:49], -2.0, v[48:49], v[32:33]\n\t" \
    "v_add_f64 v[34:35], v[34:35], v[50:51]\n\t" \
    "v_fma_f64 v[50:51], -2.0, v[50:51], v[34:35]\n\t" \
    "v_add_f64 v[36:37], v[36:37], v[52:53]\n\t" \
    "v_fma_f64 v[52:53], -2.0, v[52:53], v[36:37]\n\t" \
    "v_add_f64 v[38:39], v[38:39], v[54:55]\n\t" \
    "v_fma_f64 v[54:55], -2.0, v[54:55], v[38:39]\n\t" \
    "v_add_f64 v[40:41], v[40:41], v[56:57]\n\t" \
    "v_fma_f64 v[56:57], -2.0, v[56:57], v[40:41]\n\t" \
    "v_add_f64 v[42:43], v[42:43], v[58:59]\n\t" \
    "v_fma_f64 v[58:59], -2.0, v[58:59], v[42:43]\n\t" \
    "v_add_f64 v[44:45], v[44:45], v[60:61]\n\t" \
    "v_fma_f64 v[60:61], -2.0, v[60:61], v[44:45]\n\t" \
    "v_add_f64 v[46:47], v[46:47], v[62:63]\n\t" \
    "v_fma_f64 v[62:63], -2.0, v[62:63], v[46:47]\n\t"
#define X8T_D_0 \
    "v_mul_f64 v[20:21], v[26:27], v[66:67]\n\t" \
    "v_mul_f64 v[70:71], v[26:27], v[64:65]\n\t" \
    "v_fma_f64 v[68:69], v[24:25], v[64:65], -v[20:21]\n\t" \
    "v_fma_f64 v[70:71], v[24:25], v[66:67], v[70:71]\n\t" \
    "v_mul_f64 v[20:21], v[30:31], v[66:67]\n\t" \
    "v_mul_f64 v[74:75], v[30:31], v[64:65]\n\t" \
    "v_fma_f64 v[72:73], v[28:29], v[64:65], -v[20:21]\n\t" \
    "v_fma_f64 v[74:75], v[28:29], v[66:67], v[74:75]\n\t" \
    "v_mul_f64 v[20:21], v[30:31], v[70:71]\n\t" \
    "v_mul_f64 v[78:79], v[30:31], v[68:69]\n\t" \
    "v_fma_f64 v[76:77], v[28:29], v[68:69], -v[20:21]\n\t" \
    "v_fma_f64 v[78:79], v[28:29], v[70:71], v[78:79]\n\t" \
    "v_mul_f64 v[20:21], v[66:67], v[38:39]\n\t" \
    "v_mul_f64 v[22:23], v[66:67], v[36:37]\n\t" \
    "v_fma_f64 v[36:37], v[64:65], v[36:37], -v[20:21]\n\t" \
    "v_fma_f64 v[38:39], v[64:65], v[38:39], v[22:23]\n\t" \
    "v_mul_f64 v[20:21], v[70:71], v[46:47]\n\t" \
    "v_mul_f64 v[22:23], v[70:71], v[44:45]\n\t" \
    "v_fma_f64 v[44:45], v[68:69], v[44:45], -v[20:21]\n\t" \
    "v_fma_f64 v[46:47], v[68:69], v[46:47], v[22:23]\n\t" \
    "v_mul_f64 v[20:21], v[74:75], v[54:55]\n\t" \
    "v_mul_f64 v[22:23], v[74:75], v[52:53]\n\t" \
    "v_fma_f64 v[52:53], v[72:73], v[52:53], -v[20:21]\n\t" \
    "v_fma_f64 v[54:55], v[72:73], v[54:55], v[22:23]\n\t" \
    "v_mul_f64 v[20:21], v[78:79], v[62:63]\n\t" \
    "v_mul_f64 v[22:23], v[78:79], v[60:61]\n\t" \
    "v_fma_f64 v[60:61], v[76:77], v[60:61], -v[20:21]\n\t" \
    "v_fma_f64 v[62:63], v[76:77], v[62:63], v[22:23]\n\t"
#define X8T_D_1 \
    "v_mul_f64 v[20:21], v[26:27], v[66:67]\n\t" \
    "v_mul_f64 v[70:71], v[26:27], v[64:65]\n\t" \
    "v_fma_f64 v[68:69], v[24:25], v[64:65], -v[20:21]\n\t" \
    "v_fma_f64 v[70:71], v[24:25], v[66:67], v[70:71]\n\t" \
    "v_mul_f64 v[20:21], v[30:31], v[66:67]\n\t" \
    "v_mul_f64 v[74:75], v[30:31], v[64:65]\n\t" \
    "v_fma_f64 v[72:73], v[28:29], v[64:65], -v[20:21]\n\t" \
    "v_fma_f64 v[74:75], v[28:29], v[66:67], v[74:75]\n\t" \
    "v_mul_f64 v[20:21], v[30:31], v[70:71]\n\t" \
    "v_mul_f64 v[78:79], v[30:31], v[68:69]\n\t" \
    "v_fma_f64 v[76:77], v[28:29], v[68:69], -v[20:21]\n\t" \
    "v_fma_f64 v[78:79], v[28:29], v[70:71], v[78:79]\n\t" \
    "v_mul_f64 v[20:21], v[66:67], v[42:43]\n\t" \
    "v_mul_f64 v[22:23], v[66:67], v[40:41]\n\t" \
    "v_fma_f64 v[40:41], v[64:65], v[40:41], -v[20:21]\n\t" \
    "v_fma_f64 v[42:43], v[64:65], v[42:43], v[22:23]\n\t" \
    "v_mul_f64 v[20:21], v[70:71], v[46:47]\n\t" \
    "v_mul_f64 v[22:23], v[70:71], v[44:45]\n\t" \
    "v_fma_f64 v[44:45], v[68:69], v[44:45], -v[20:21]\n\t" \
    "v_fma_f64 v[46:47], v[68:69], v[46:47], v[22:23]\n\t" \
    "v_mul_f64 v[20:21], v[74:75], v[58:59]\n\t" \
    "v_mul_f64 v[22:23], v[74:75], v[56:57]\n\t" \
    "v_fma_f64 v[56:57], v[72:73], v[56:57], -v[20:21]\n\t" \
    "v_fma_f64 v[58:59], v[72:73], v[58:59], v[22:23]\n\t" \
    "v_mul_f64 v[20:21], v[78:79], v[62:63]\n\t" \
    "v_mul_f64 v[22:23], v[78:79], v[60:61]\n\t" \
    "v_fma_f64 v[60:61], v[76:77], v[60:61], -v[20:21]\n\t" \
    "v_fma_f64 v[62:63], v[76:77], v[62:63], v[22:23]\n\t"
#define X8T_D_2 \
    "v_mul_f64 v[20:21], v[26:27], v[66:67]\n\t" \
    "v_mul_f64 v[70:71], v[26:27], v[64:65]\n\t" \
    "v_fma_f64 v[68:69], v[24:25], v[64:65], -v[20:21]\n\t" \
    "v_fma_f64 v[70:71], v[24:25], v[66:67], v[70:71]\n\t" \
    "v_mul_f64 v[20:21], v[30:31], v[66:67]\n\t" \
    "v_mul_f64 v[74:75], v[30:31], v[64:65]\n\t" \
    "v_fma_f64 v[72:73], v[28:29], v[64:65], -v[20:21]\n\t" \
    "v_fma_f64 v[74:75], v[28:29], v[66:67], v[74:75]\n\t" \
    "v_mul_f64 v[20:21], v[30:31], v[70:71]\n\t" \
    "v_mul_f64 v[78:79], v[30:31], v[68:69]\n\t" \
    "v_fma_f64 v[76:77], v[28:29], v[68:69], -v[20:21]\n\t" \
    "v_fma_f64 v[78:79], v[28:29], v[70:71], v[78:79]\n\t" \
    "v_mul_f64 v[20:21], v[66:67], v[50:51]\n\t" \
    "v_mul_f64 v[22:23], v[66:67], v[48:49]\n\t" \
    "v_fma_f64 v[48:49], v[64:65], v[48:49], -v[20:21]\n\t" \
    "v_fma_f64 v[50:51], v[64:65], v[50:51], v[22:23]\n\t" \
    "v_mul_f64 v[20:21], v[70:71], v[54:55]\n\t" \
    "v_mul_f64 v[22:23], v[70:71], v[52:53]\n\t" \
    "v_fma_f64 v[52:53], v[68:69], v[52:53], -v[20:21]\n\t" \
    "v_fma_f64 v[54:55], v[68:69], v[54:55], v[22:23]\n\t" \
    "v_mul_f64 v[20:21], v[74:75], v[58:59]\n\t" \
    "v_mul_f64 v[22:23], v[74:75], v[56:57]\n\t" \
    "v_fma_f64 v[56:57], v[72:73], v[56:57], -v[20:21]\n\t" \
    "v_fma_f64 v[58:59], v[72:73], v[58:59], v[22:23]\n\t" \
    "v_mul_f64 v[20:21], v[78:79], v[62:63]\n\t" \
    "v_mul_f64 v[22:23], v[78:79], v[60:61]\n\t" \
    "v_fma_f64 v[60:61], v[76:77], v[60:61], -v[20:21]\n\t" \
    "v_fma_f64 v[62:63], v[76:77], v[62:63], v[22:23]\n\t"
#define X8T_FMUL_0 \
    "v_mul_f64 v[20:21], v[70:71], v[66:67]\n\t" \
    "v_mul_f64 v[22:23], v[70:71], v[64:65]\n\t" \
    "v_fma_f64 v[64:65], v[68:69], v[64:65], -v[20:21]\n\t" \
    "v_fma_f64 v[66:67], v[68:69], v[66:67], v[22:23]\n\t"
#define X8T_FMUL_1 \
    "v_mul_f64 v[20:21], v[74:75], v[66:67]\n\t" \
    "v_mul_f64 v[22:23], v[74:75], v[64:65]\n\t" \
    "v_fma_f64 v[64:65], v[72:73], v[64:65], -v[20:21]\n\t" \
    "v_fma_f64 v[66:67], v[72:73], v[66:67], v[22:23]\n\t"
#define X8T_FMUL_2 \
    "v_mul_f64 v[20:21], v[78:79], v[66:67]\n\t" \
    "v_mul_f64 v[22:23], v[78:79], v[64:65]\n\t" \
    "v_fma_f64 v[64:65], v[76:77], v[64:65], -v[20:21]\n\t" \
    "v_fma_f64 v[66:67], v[76:77], v[66:67], v[22:23]\n\t"
#define X8T_STEP(K1, SW, NEXT)                                  \
    "s_cmp_lt_u32 %[ns], " #K1 "\n\t"                               \
    "s_cbranch_scc1 " NEXT "\n\t"                                    \
    "s_and_b32 s94, " SW ", 3\n\t"                                   \
    "s_bitcmp1_b32 " SW ", 2\n\t"                                    \
    "s_cbranch_scc0 60f\n\t"                                         \
    /* the diagonal's table reads, issued in front of the butterflies */ \
    "s_bfe_u32 s95, " SW ", 0x80008\n\t"                             \
    "s_lshl_b32 s96, s95, 4\n\t"                                     \
    "s_add_u32 s96, s96, %[dg]\n\t"                                  \
    "v_mov_b32 v19, s96\n\t"                                         \
    "ds_read_b128 v[64:67], v19\n\t"                                 \
    "s_mulk_i32 s95, 0x300\n\t"                                      \
    "s_add_u32 s95, s95, %[gt]\n\t"                                  \
    "s_bitcmp1_b32 " SW ", 16\n\t"                                   \
    "s_cbranch_scc0 61f\n\t"                                         \
    "v_add_u32 v19, s95, %[n0]\n\t"                                  \
    "ds_read_b128 v[68:71], v19\n\t"                               \
    "61:\n\t"                                                        \
    "s_bitcmp1_b32 " SW ", 17\n\t"                                   \
    "s_cbranch_scc0 62f\n\t"                                         \
    "v_add_u32 v19, s95, %[n1]\n\t"                                  \
    "ds_read_b128 v[72:75], v19\n\t"                               \
    "62:\n\t"                                                        \
    "s_bitcmp1_b32 " SW ", 18\n\t"                                   \
    "s_cbranch_scc0 63f\n\t"                                         \
    "v_add_u32 v19, s95, %[n2]\n\t"                                  \
    "ds_read_b128 v[76:79], v19\n\t"                               \
    "63:\n\t"                                                        \
    /* the other two register bits of the step: O1 = (R == 0 ? bit 1 : bit 0), O2 = (R == 2 ? bit 1 : bit 2) */ \
    "s_cmp_eq_u32 s94, 0\n\t"                                        \
    "s_cselect_b32 s97, %[o1], %[o0]\n\t"                            \
    "s_cmp_eq_u32 s94, 2\n\t"                                        \
    "s_cselect_b32 s98, %[o1], %[o2]\n\t"                            \
    "v_mov_b64 v[24:25], 1.0\n\t"                                    \
    "v_mov_b64 v[26:27], 0\n\t"                                      \
    "s_bitcmp1_b32 " SW ", 19\n\t"                                   \
    "s_cbranch_scc0 64f\n\t"                                         \
    "s_add_u32 s97, s97, s95\n\t"                                    \
    "v_mov_b32 v19, s97\n\t"                                         \
    "ds_read_b128 v[24:27], v19\n\t"                                 \
    "64:\n\t"                                                        \
    "v_mov_b64 v[28:29], 1.0\n\t"                                    \
    "v_mov_b64 v[30:31], 0\n\t"                                      \
    "s_bitcmp1_b32 " SW ", 20\n\t"                                   \
    "s_cbranch_scc0 60f\n\t"                                         \
    "s_add_u32 s98, s98, s95\n\t"                                    \
    "v_mov_b32 v19, s98\n\t"                                         \
    "ds_read_b128 v[28:31], v19\n\t"                                 \
    "60:\n\t"                                                        \
    /* the butterflies (unscaled: the pass multiplies by 1/sqrt 2 ^ Hadamards once, at the store) */ \
    "s_cmp_eq_u32 s94, 0\n\t s_cbranch_scc1 70f\n\t"                 \
    "s_cmp_eq_u32 s94, 1\n\t s_cbranch_scc1 71f\n\t"                 \
    X8T_H_2 "s_branch 72f\n\t"                                       \
    "70:\n\t" X8T_H_0 "s_branch 72f\n\t"                             \
    "71:\n\t" X8T_H_1                                                \
    "72:\n\t"                                                        \
    "s_bitcmp1_b32 " SW ", 2\n\t"                                    \
    "s_cbranch_scc0 " NEXT "\n\t"                                    \
    "s_waitcnt lgkmcnt(0)\n\t"                                       \
    "s_bitcmp1_b32 " SW ", 16\n\t s_cbranch_scc0 65f\n\t" X8T_FMUL_0 "65:\n\t" \
    "s_bitcmp1_b32 " SW ", 17\n\t s_cbranch_scc0 66f\n\t" X8T_FMUL_1 "66:\n\t" \
    "s_bitcmp1_b32 " SW ", 18\n\t s_cbranch_scc0 67f\n\t" X8T_FMUL_2 "67:\n\t" \
    "s_cmp_eq_u32 s94, 0\n\t s_cbranch_scc1 73f\n\t"                 \
    "s_cmp_eq_u32 s94, 1\n\t s_cbranch_scc1 74f\n\t"                 \
    X8T_D_2 "s_branch " NEXT "\n\t"                                  \
    "73:\n\t" X8T_D_0 "s_branch " NEXT "\n\t"                        \
    "74:\n\t" X8T_D_1
#define X8T_ROUND_ALL                                                \
    "ds_read_b128 v[32:35], %[a0]\n\t"                               \
    X8_ADDR1("%[d0]") "ds_read_b128 v[36:39], v19\n\t"               \
    X8_ADDR1("%[d1]") "ds_read_b128 v[40:43], v19\n\t"               \
    X8_ADDR2("%[d0]", "%[d1]") "ds_read_b128 v[44:47], v19\n\t"      \
    X8_ADDR1("%[d2]") "ds_read_b128 v[48:51], v19\n\t"               \
    X8_ADDR2("%[d0]", "%[d2]") "ds_read_b128 v[52:55], v19\n\t"      \
    X8_ADDR2("%[d1]", "%[d2]") "ds_read_b128 v[56:59], v19\n\t"      \
    X8_ADDR3("%[d0]", "%[d1]", "%[d2]") "ds_read_b128 v[60:63], v19\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"                                       \
    X8T_STEP(1, "%[sA]", "81f") "81:\n\t"                            \
    X8T_STEP(2, "%[sB]", "82f") "82:\n\t"                            \
    X8T_STEP(3, "%[sC]", "83f") "83:\n\t"                            \
    "ds_write_b128 %[a0], v[32:35]\n\t"                              \
    X8_ADDR1("%[d0]") "ds_write_b128 v19, v[36:39]\n\t"              \
    X8_ADDR1("%[d1]") "ds_write_b128 v19, v[40:43]\n\t"              \
    X8_ADDR2("%[d0]", "%[d1]") "ds_write_b128 v19, v[44:47]\n\t"     \
    X8_ADDR1("%[d2]") "ds_write_b128 v19, v[48:51]\n\t"              \
    X8_ADDR2("%[d0]", "%[d2]") "ds_write_b128 v19, v[52:55]\n\t"     \
    X8_ADDR2("%[d1]", "%[d2]") "ds_write_b128 v19, v[56:59]\n\t"     \
    X8_ADDR3("%[d0]", "%[d1]", "%[d2]") "ds_write_b128 v19, v[60:63]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t"

// one fast round of the tolerance mode on this thread's 8 amplitudes.  n0..n2: byte offsets of this thread's entries in the three
// 16-entry groups of a diagonal's G tables (from its tile-local index p); o0..o2: byte offsets of the single-bit entries of the
// round's three register bits; dg / gt: LDS byte addresses of the E_out slots / the G tables
__device__ __forceinline__ void fuse_round8t(uint32_t a0, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t n0, uint32_t n1, uint32_t n2,
                                             uint32_t o0, uint32_t o1, uint32_t o2, uint32_t sA, uint32_t sB, uint32_t sC, uint32_t ns,
                                             uint32_t dg, uint32_t gt)
{
    asm volatile(X8T_ROUND_ALL
        :
        : [a0] "v"(a0), [d0] "s"(d0), [d1] "s"(d1), [d2] "s"(d2), [n0] "v"(n0), [n1] "v"(n1), [n2] "v"(n2),
          [o0] "s"(o0), [o1] "s"(o1), [o2] "s"(o2), [sA] "s"(sA), [sB] "s"(sB), [sC] "s"(sC), [ns] "s"(ns), [dg] "s"(dg), [gt] "s"(gt)
        : "memory", "vcc", "scc", "v19", "v20", "v21", "v22", "v23",
          "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",
          "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
          "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
          "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
          "s94", "s95", "s96", "s97", "s98");
}

template <int BLOCK, int TT, bool GEN = false, bool TOL = false>     // TOL: the pass holds FUSE_QROUND3 rounds (tolerance mode, K6x-t)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_fused_x8(
    const amp_t *amp, amp_t *amp_out, unsigned n, FusePass P, const FuseOp *__restrict__ ops, uint64_t ntiles, const FuseOp *ops_asm)
{
    static_assert((1u << TT) == 8u * BLOCK, "the walk on 8 amplitudes per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char qcx_lds_raw[];
    amp_t *tile = reinterpret_cast<amp_t *>(qcx_lds_raw);
    constexpr unsigned tsize = 1u << TT;
    uint64_t *xm = reinterpret_cast<uint64_t *>(reinterpret_cast<unsigned char *>(tile + tsize) + P.xm_off);
    // tolerance mode: [E_out slot per diagonal][G tables, 48 entries per diagonal] behind the tile
    amp_t *dg = reinterpret_cast<amp_t *>(reinterpret_cast<unsigned char *>(tile + tsize) + P.dg_lds_off);
    const amp_t *dg_area = reinterpret_cast<const amp_t *>(ops + P.dg_rec_off);             // global: DiagInfo[], G tables, field tables
    bool staged = false;
    // fill: slot s = k * BLOCK + thread (what the LDS-DMA writes linearly) receives element x8_swz(s); spread and swizzle are both
    // linear over XOR, so the thread part and the k part are computed once and combined with XOR
    const uint64_t off_t = fuse_spread(x8_swz(threadIdx.x), P.in_pos, TT);
    uint64_t off_k[8], st_k[8];
    unsigned ld_k[8];
#pragma unroll
    for (unsigned k = 0; k < 8; k++) {
        off_k[k] = fuse_spread(x8_swz(k * BLOCK), P.in_pos, TT);
        st_k[k] = fuse_spread(k * BLOCK, P.st_pos, TT);
        ld_k[k] = x8_swz((unsigned)fuse_spread(k * BLOCK, P.st_loc, TT));
    }
    const uint64_t st_t = fuse_spread(threadIdx.x, P.st_pos, TT);
    const unsigned ld_t = x8_swz((unsigned)fuse_spread(threadIdx.x, P.st_loc, TT));
    const unsigned wbase = (threadIdx.x >> 6) * 64, lane = threadIdx.x & 63u;
    const unsigned wave_id = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t tile_lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) amp_t *)tile;
    const uint32_t xm_lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint64_t *)xm;
    const uint32_t dg_lds = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) amp_t *)dg;
    // the expanding store of a compact chain's last pass (FusePass::xp_on): this thread's share of an expanded store index --
    // its M-register value f (BLOCK >= 2^M: the same in every store), hence its column, and the thread bits above f
    uint64_t xp_off_t = 0;
    unsigned xp_slot_t = 0;
    bool xp_live = false;
    if (!GEN && P.xp_on) {
        const unsigned M = P.xp_M, f = threadIdx.x & ((1u << M) - 1u);
        unsigned col = 0xffu;
        for (unsigned j = 0; j < P.xp_ncols; j++) if (P.xp_orbit[j] == f) col = j;
        xp_live = col != 0xffu;
        unsigned loc = 0;
        if (xp_live) for (unsigned b = 0; b < P.xp_cb; b++) loc |= ((col >> b) & 1u) << P.xp_colloc[b];
        xp_off_t = f;
        for (unsigned i = M; (1u << i) < (unsigned)BLOCK; i++)
            if ((threadIdx.x >> i) & 1u) { xp_off_t |= (uint64_t)1 << P.xp_pos[i]; loc |= 1u << P.xp_loc[i]; }
        xp_slot_t = x8_swz(loc);
    }
    // GEN: the tiles are generated (the circuit front on a basis state that was never written, GenFront), not read: slot s holds
    // element x8_swz(s), and the per-element words of the generated fill are linear over XOR like everything else here
    const GenFront *GF = reinterpret_cast<const GenFront *>(ops + P.gen_rec_off);
    unsigned short *gen_phot = reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(tile + tsize) + P.gen_lds_off);
    unsigned short *gen_res = gen_phot + 1024;
    uint32_t packT = 0, packK[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (GEN) {
        gen_setup<BLOCK>(GF, gen_phot);
        packT = gen_pack(x8_swz(threadIdx.x), GF, TT);
#pragma unroll
        for (unsigned k = 0; k < 8; k++) packK[k] = gen_pack(x8_swz(k * BLOCK), GF, TT);
        __syncthreads();
    }
    const unsigned slog = (P.dbg >> 12) & 15u;
    for (uint64_t b0 = blockIdx.x; b0 < ntiles; b0 += gridDim.x) {
        const uint64_t t = fuse_stream_tile(b0, n - TT, slog, (P.dbg >> 16) & 31u);
        const uint64_t base_in = fuse_deposit(t, P.seg_in, P.nseg_in);
        uint64_t base = base_in, base_out = base_in;               // logical base (gate records), output base
        if (P.chained) { base = fuse_deposit(t, P.seg_lg, P.nseg_lg); base_out = fuse_deposit(t, P.seg_out, P.nseg_out); }
        const amp_t *g = amp + base_in;
        amp_t *go = amp_out + (base_out | st_t);
        if constexpr (GEN) gen_tile<BLOCK, 8>(GF, tile, gen_phot, gen_res, base, packT, packK);
        else if (!(P.dbg & 4u)) {
#pragma unroll
            for (unsigned k = 0; k < 8; k++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (off_t ^ off_k[k])),
                                                 (__attribute__((address_space(3))) void *)(tile + k * BLOCK + wbase), 16, 0, 2);
        }
        if (!staged) {          // once per workgroup -- behind the first tile's fill, not in front of it (a workgroup usually takes ONE
            staged = true;      // tile: this latency used to be paid per tile, with nothing in flight)
            if constexpr (TOL) {
                for (unsigned b = threadIdx.x; b < P.dg_cnt * 48u; b += BLOCK) dg[P.dg_cnt + b] = dg_area[3u * P.dg_cnt + b];
            } else {            // the records' outside-tile masks (+ padding: a lane looks up to 64 entries past a run)
                for (unsigned b = threadIdx.x; b < P.xm_cnt + 66u; b += BLOCK) xm[b] = b < P.xm_cnt ? ops[b].mask : 0;
            }
        }
        if constexpr (TOL) {
            if (threadIdx.x < P.dg_cnt) {           // E_out of this tile for every diagonal of the pass, while the fill is in flight
                const DiagInfo *info = reinterpret_cast<const DiagInfo *>(dg_area) + threadIdx.x;
                const uint32_t present = info->present;
                amp_t E; E.x = info->kc; E.y = info->ks;
#pragma unroll
                for (unsigned f = 0; f < 5; f++)
                    if ((present >> f) & 1u) cmul_tol(E, dg_area[info->field_off[f] + (unsigned)((base >> (8u * f)) & 255u)]);
                dg[threadIdx.x] = E;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // Which part of the tile a wave takes through the rounds rotates with the tile number: gates whose condition sits on a
        // wave bit are skipped by the waves whose bit is 0, so the parts differ in work (part 7 of a pass with three filler
        // targets under the wave number runs 24 more gates than part 0), a wave sits on one SIMD for good, and a workgroup
        // lasts as long as its slowest wave -- with a fixed assignment one SIMD of every CU would carry the heavy parts of
        // BOTH resident workgroups.  (Fill and store go by the real thread number.)
        const unsigned wave_eff = TOL ? wave_id : (wave_id ^ (unsigned)t ^ (unsigned)(t >> 3)) & ((BLOCK >> 6) - 1u);
        const unsigned tid_eff = lane | (wave_eff << 6);
        if (!(P.dbg & 1u)) {
            for (unsigned i = 0; i < P.nops;) {
                const uint32_t a = ops[i].a;
                const unsigned rb0 = a & 0xffu, rb1 = (a >> 8) & 0xffu, rb2 = (a >> 16) & 0xffu;
                const uint32_t cnt = (uint32_t)ops[i].mask;
                uint64_t map; memcpy(&map, &ops[i].c, sizeof map);
                unsigned p = 0;
#pragma unroll
                for (unsigned k = 0; k < (unsigned)TT - 3u; k++) p |= ((tid_eff >> k) & 1u) << ((unsigned)(map >> (4u * k)) & 15u);
                // "no barrier in front of this round" (bit 25 of an exact round's header, bit 28 of a tolerance round's): its waves sit
                // on the same tile bits as the previous round's -- every wave finds its own amplitudes where it left them (its LDS
                // accesses are served in order) and nobody else's
                if (i != 0 && !((a >> (TOL ? 28 : 25)) & 1u)) __syncthreads();
                if constexpr (TOL) {
                    const uint32_t sA = ops[i + 1].type, sB = ops[i + 1].a, sC = (uint32_t)ops[i + 1].mask;
                    auto single = [](unsigned rb) { return 16u * (16u * (rb >> 2) + (1u << (rb & 3u))); };     // G-table entry of one tile bit alone
                    fuse_round8t(tile_lds + 16u * x8_swz(p), 16u * x8_swz(1u << rb0), 16u * x8_swz(1u << rb1), 16u * x8_swz(1u << rb2),
                                 16u * (p & 15u), 16u * (16u + ((p >> 4) & 15u)), 16u * (32u + (p >> 8)), single(rb0), single(rb1), single(rb2),
                                 sA, sB, sC, (a >> 24) & 3u, dg_lds, dg_lds + 16u * P.dg_cnt);
                } else {
                    // the walk reads the records through ops_asm (see fuse_apply_rounds: `ops` itself must not be captured by an asm)
                    // (bits 48 .. of a gate's outside mask: conditions on the tile bits this round's WAVE number rides on -- the host moved
                    //  them there from the lane mask, x8_assign_maps -- so that the per-item ballot already drops the gates this wave skips)
                    fuse_round8(tile_lds + 16u * x8_swz(p), p, xm_lds + 8u * (i + 2u + lane), base | ((uint64_t)wave_eff << 48), ops_asm + i + 1, cnt,
                                16u * x8_swz(1u << rb0), 16u * x8_swz(1u << rb1), 16u * x8_swz(1u << rb2), (a >> 24) & 1u);
                }
                i += 1 + cnt;
            }
            __syncthreads();
        }
        amp_t v[8];
        const double sc = (TOL && !(P.dbg & 1u)) ? P.tol_scale : 1.0;        // tolerance mode: the Hadamards' 1/sqrt 2, once per pass
        if (!GEN && P.xp_on) {
            // the real register from the compact tile: 2^(T - cb + M) amplitudes, store index e = it * BLOCK + thread; consecutive
            // threads write consecutive amplitudes (whole 2^M-blocks, runs of 2^(M + c - cb) of them), the threads whose f is on
            // the orbit fetch theirs from the tile, all others write +0
            constexpr unsigned LB = TT - 3u;                                 // log2 BLOCK
            const unsigned ebits = (unsigned)TT - P.xp_cb + P.xp_M, nit = 1u << (ebits - LB);
            amp_t *gx = amp_out + (((base_out >> P.xp_cb) << P.xp_M) | xp_off_t);
            for (unsigned it0 = 0; it0 < nit; it0 += 8) {
                uint64_t oi[8];
#pragma unroll
                for (unsigned k = 0; k < 8; k++) {
                    const unsigned it = it0 + k;
                    uint64_t off = 0; unsigned loc = 0;
                    for (unsigned b = 0; b < ebits - LB; b++)
                        if ((it >> b) & 1u) { off |= (uint64_t)1 << P.xp_pos[LB + b]; loc |= 1u << P.xp_loc[LB + b]; }
                    oi[k] = off;
                    v[k].x = 0.0; v[k].y = 0.0;
                    if (xp_live) { v[k] = tile[xp_slot_t ^ x8_swz(loc)]; if constexpr (TOL) { v[k].x *= sc; v[k].y *= sc; } }
                }
                if (!(P.dbg & 2u)) {
#pragma unroll
                    for (unsigned k = 0; k < 8; k++) if (it0 + k < nit) __builtin_nontemporal_store(v[k], gx + oi[k]);
                }
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (unsigned k = 0; k < 8; k++) {
            v[k] = tile[ld_k[k] ^ ld_t];          // (store order: ascending OUTPUT positions; slots under the swizzle)
            if constexpr (TOL) { v[k].x *= sc; v[k].y *= sc; }
        }
        if (!(P.dbg & 2u)) {
#pragma unroll
            for (unsigned k = 0; k < 8; k++) __builtin_nontemporal_store(v[k], go + st_k[k]);
        }
        __syncthreads();
    }
}

// (A persistent, double-buffered form of the pass -- the LDS-DMA fill of tile i+1 in flight while tile i is processed and
// stored, the wait before use a COUNTED s_waitcnt vmcnt(stores issued after the fill) -- existed in round 1 and is gone:
// measured slower than this kernel, and the counted wait is not safe on gfx950.  A wave's loads and stores do not
// complete in one common order, so "all but the N youngest" may leave some of the older LOADS outstanding while the
// younger stores have already been acknowledged; round 2 saw exactly that as wrong amplitudes at n = 28 in a
// register-prefetch variant of an all-Hadamard pass, and the compiler itself always waits vmcnt(0) when loads and
// stores of one wave are pending together.  A wave that has stores in flight can only wait for ALL of its
// vector-memory operations; DESIGN.md s4 has the numbers of that experiment.)

// ---------------------------------------------------------------------------
// K0b  the circuit front on a BASIS state, in one write pass.  After reset_register (Q:318-324) -- or a measurement's
// collapse (Q:302-303) -- the register holds one amplitude (1, 0) at index b.  A prefix of the queued gates of the shape
//     H on distinct qubits (the set Hm)  then  controlled modular multiplies          (Q:720-731 is exactly that)
// has a closed form with the reference's own roundings:
//   * each H multiplies the populated amplitudes by s = M_SQRT1_2 ONCE MORE (the pair partner is an exact zero, so the
//     reference's (s*a +/- s*0) + 0 is fl(s*a) with sign -1 on the "1" side when the basis bit was 1): after k of them
//     every populated amplitude is +/- v_k, v_k = fl(s * v_(k-1)), v_0 = 1, sign = parity of (i & b & Hm);
//   * a modular multiply moves the one populated residue f of every 2^M block to (A f) mod C when its control bit is 1
//     and f < C (all other sources of that destination are exact zeros: 0 + ... + v = v), so after the ladder block x
//     holds its amplitude at f(x), the chain over the gates in issue order.
// The kernel writes EVERY amplitude of the shard once (the zeros too: it replaces the 16 * 2^n-byte memset of the reset),
// 16 * 2^n bytes instead of one read + write per Hadamard pass.  A wave owns 64 blocks: lane l walks the chain of block l,
// then the wave writes the 64 * 2^M amplitudes with coalesced 1-KiB stores, fetching each store's residues by shuffle.
// ---------------------------------------------------------------------------
struct BasisFront {
    uint64_t first;                         // global index of the shard's amplitude 0 (0 for an unsharded register)
    uint64_t basis, hmask, fixed_mask;      // populated blocks: (i & fixed_mask) == (basis & fixed_mask)
    uint64_t sign_mask;                     // sign = parity of popcount(i & sign_mask)
    double   v;                             // magnitude after the Hadamards
    unsigned M, ncam;                       // M = 0 and ncam = 0: no block structure (every amplitude is its own block)
    uint32_t C[64], A[64];
    uint8_t  ctl[64];
};

// the same rule, one thread per amplitude, for registers too small for a wave tile (n < M + 6: the reference's own sizes,
// e.g. L = 3, M = 4): every thread walks the residue chain of its block itself -- a few thousand amplitudes at most
__global__ __launch_bounds__(256) void k_basis_front_small(amp_t *__restrict__ amp, unsigned n, BasisFront B)
{
    const unsigned M = B.M;
    const uint64_t lowmask = ((uint64_t)1 << M) - 1;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < ((uint64_t)1 << n); i += (uint64_t)gridDim.x * 256) {
        const uint64_t gi = B.first + i, bi = gi & ~lowmask;
        unsigned f = (unsigned)(B.basis & lowmask);
        for (unsigned g = 0; g < B.ncam; g++)
            if (((bi >> B.ctl[g]) & 1u) && f < B.C[g]) f = (B.A[g] * f) % B.C[g];
        const unsigned low = (unsigned)(gi & lowmask), free_low = (unsigned)(B.hmask & lowmask);
        amp_t v; v.x = 0.0; v.y = 0.0;
        if ((bi & B.fixed_mask) == (B.basis & B.fixed_mask) && ((low ^ f) & ~free_low) == 0)
            v.x = (__builtin_popcountll(gi & B.sign_mask) & 1) ? -B.v : B.v;
        amp[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_basis_front(amp_t *__restrict__ amp, unsigned n, BasisFront B)
{
    const unsigned lane = threadIdx.x & 63u;
    const unsigned M = B.M;
    const uint64_t ntiles = ((uint64_t)1 << n) >> (6 + M);          // host guarantees n >= M + 6
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * 256) >> 6;
    const unsigned lowmask = (1u << M) - 1u;
    for (uint64_t t = wave; t < ntiles; t += nwaves) {
        const uint64_t tile0 = t << (6 + M);
        // lane l: block l of the tile (bi = its GLOBAL index: controls, signs and the populated test may involve shard-id bits)
        const uint64_t bi = B.first + tile0 + ((uint64_t)lane << M);
        unsigned f = (unsigned)(B.basis & lowmask);
        bool pop = (bi & B.fixed_mask) == (B.basis & B.fixed_mask);
        for (unsigned g = 0; g < B.ncam; g++)
            if (((bi >> B.ctl[g]) & 1u) && f < B.C[g]) f = (B.A[g] * f) % B.C[g];
        // one packed word per block for the shuffles: residue | sign << 16, or -1 for an empty block
        const unsigned sgn_blk = (unsigned)__builtin_popcountll(bi & B.sign_mask) & 1u;
        const int code = pop ? (int)(f | (sgn_blk << 16)) : -1;
        const unsigned free_low = (unsigned)(B.hmask & lowmask), sign_low = (unsigned)(B.sign_mask & lowmask);
        const double vp = B.v, vm = -B.v;
        for (unsigned j = 0; j < (1u << M); j++) {
            const unsigned e = j * 64u + lane;                      // element of the tile this lane stores
            const int bc = __shfl(code, (int)(e >> M), 64);        // its block's word
            const unsigned low = e & lowmask;
            amp_t v; v.x = 0.0; v.y = 0.0;
            // with a block structure (ncam > 0, or M bits outside the Hadamard set) the low bits must equal the residue;
            // low bits inside the Hadamard set are free (then ncam == 0 and fixed_mask covers the rest)
            if (bc >= 0 && ((low ^ (unsigned)bc) & ~free_low & lowmask) == 0) {
                const unsigned sg = ((unsigned)bc >> 16) ^ ((unsigned)__builtin_popcount(low & sign_low) & 1u);
                v.x = sg ? vm : vp;
            }
            __builtin_nontemporal_store(v, amp + tile0 + e);
        }
    }
}

// the same rule for M registers beyond the wave tile (M > 12): a workgroup per 2^M-block -- its first wave walks the block's
// residue chain (uniform over the lanes), the whole workgroup writes the block's 2^M amplitudes with coalesced stores
__global__ __launch_bounds__(256) void k_basis_front_big(amp_t *__restrict__ amp, unsigned n, BasisFront B)
{
    __shared__ unsigned s_code[2];
    const unsigned M = B.M;
    const uint64_t nblocks = ((uint64_t)1 << n) >> M;              // host guarantees n >= M
    const unsigned lowmask = (1u << M) - 1u;
    const unsigned free_low = (unsigned)(B.hmask & lowmask), sign_low = (unsigned)(B.sign_mask & lowmask);
    const double vp = B.v, vm = -B.v;
    for (uint64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const uint64_t bi = B.first + (blk << M);                  // GLOBAL index of the block (controls, signs, populated test)
        if (threadIdx.x < 64) {
            unsigned f = (unsigned)(B.basis & lowmask);
            for (unsigned g = 0; g < B.ncam; g++)
                if (((bi >> B.ctl[g]) & 1u) && f < B.C[g]) f = (B.A[g] * f) % B.C[g];
            if (threadIdx.x == 0) {
                s_code[0] = f;
                s_code[1] = ((bi & B.fixed_mask) == (B.basis & B.fixed_mask) ? 2u : 0u) | ((unsigned)__builtin_popcountll(bi & B.sign_mask) & 1u);
            }
        }
        __syncthreads();
        const unsigned f = s_code[0], fl = s_code[1];
        amp_t *to = amp + (blk << M);
        for (unsigned e = threadIdx.x; e <= lowmask; e += 256) {
            amp_t v; v.x = 0.0; v.y = 0.0;
            if ((fl & 2u) && ((e ^ f) & ~free_low & lowmask) == 0) {
                const unsigned sg = (fl & 1u) ^ ((unsigned)__builtin_popcount(e & sign_low) & 1u);
                v.x = sg ? vm : vp;
            }
            __builtin_nontemporal_store(v, to + e);
        }
        __syncthreads();
    }
}

// the circuit front written straight in the COMPACT form of a compact chain ([L register][orbit column], cb column bits): one
// thread per 2^M-block of the real register walks the block's residue chain and writes the block's 2^cb compact amplitudes
// (the one at its residue's place in the orbit = +/- v, the others +0).  B.first = REAL global index of the shard's amplitude 0.
__global__ __launch_bounds__(256) void k_basis_front_compact(amp_t *__restrict__ compact, unsigned n_local_compact, BasisFront B, ExpandParams E)
{
    const unsigned M = B.M, cb = E.cb;
    const uint64_t nblocks = ((uint64_t)1 << n_local_compact) >> cb;
    const unsigned lowmask = (1u << M) - 1u;
    for (uint64_t blk = (uint64_t)blockIdx.x * 256 + threadIdx.x; blk < nblocks; blk += (uint64_t)gridDim.x * 256) {
        const uint64_t bi = B.first + (blk << M);
        unsigned f = (unsigned)(B.basis & lowmask);
        for (unsigned g = 0; g < B.ncam; g++)
            if (((bi >> B.ctl[g]) & 1u) && f < B.C[g]) f = (B.A[g] * f) % B.C[g];
        unsigned col = 0xffu;
        if ((bi & B.fixed_mask) == (B.basis & B.fixed_mask))
            for (unsigned j = 0; j < E.ncols; j++) if (E.orbit[j] == f) col = j;
        const double v = (__builtin_popcountll(bi & B.sign_mask) & 1) ? -B.v : B.v;
        amp_t *to = compact + (blk << cb);
        for (unsigned j = 0; j < (1u << cb); j++) {
            amp_t a; a.x = (j == col) ? v : 0.0; a.y = 0.0;
            to[j] = a;
        }
    }
}

// ---------------------------------------------------------------------------
// X1  local index-bit permutation (pack pass of the sharded qubit remap): dst[j] = src[j with the bit
// pairs (a_m, b_m) exchanged].  Out of place, coalesced stores, gathered loads (runs of 2^min(a, b)).
// ---------------------------------------------------------------------------
struct SwapBits { unsigned npairs; unsigned a[8]; unsigned b[8]; };

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_swap_bits(const amp_t *__restrict__ src, amp_t *__restrict__ dst,
                                                       uint64_t count, SwapBits S)
{
    for (uint64_t j = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; j < count; j += (uint64_t)gridDim.x * BLOCK) {
        uint64_t i = j;
        for (unsigned m = 0; m < S.npairs; m++) {
            const uint64_t x = ((i >> S.a[m]) ^ (i >> S.b[m])) & 1u;      // exchange bits a and b
            i ^= (x << S.a[m]) | (x << S.b[m]);
        }
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + j);
    }
}

// ---------------------------------------------------------------------------
// X1b  pack + trade in ONE pass, for a sharded register whose shards live in one process (qcx_sharded.inc.h): shard
// `me` of W = 2^k reads its own buffer and writes every amplitude straight into the buffer of the shard that owns it
// after the trade -- peer stores over xGMI when that shard is on another GPU, so the pack pass IS the transfer:
//     j = chunk << zone_lo | low   (chunk = the k trade-zone bits = destination shard)
//     dst[chunk][me << zone_lo | low] = src[j with the bit pairs of S exchanged]
// Blocks are dealt round-robin over the destination shards (block b -> chunk b mod W), so that all W - 1 links of a
// GPU carry traffic at the same time instead of one after the other.  Stores are 1 KiB per wave instruction,
// contiguous in the destination.
// ---------------------------------------------------------------------------
struct PushDst { amp_t *dst[16]; };
// Multi-path striping: with fewer shards than GPUs on the node, the chunk for another shard is cut into
// a direct stripe (blocks [0, nb_direct) of the chunk, stored into the owner's buffer as above) and one stripe per RELAY
// GPU, stored into that GPU's staging area instead (slot = me * W + chunk, `stage_amps` amplitudes each); the host then
// forwards the staged stripes to their owners.  Each stripe rides a different xGMI link out of this GPU.
struct PushRelay { amp_t *stage[8]; unsigned nrelays; uint32_t nb_direct, nb_relay; uint64_t stage_amps; };

template <int BLOCK, bool DEAL>     // DEAL needs BLOCK | 2^zone_lo; small shards take the plain element order
__global__ __launch_bounds__(BLOCK) void k_pack_push(const amp_t *__restrict__ src, PushDst D, PushRelay Rl, uint64_t count, SwapBits S,
                                                       unsigned zone_lo, unsigned klog, unsigned me)
{
    const unsigned W = 1u << klog;
    for (uint64_t t = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; t < count; t += (uint64_t)gridDim.x * BLOCK) {
        unsigned chunk; uint64_t low;
        if (DEAL) { const uint64_t b = t / BLOCK; chunk = (unsigned)(b & (W - 1)); low = (b >> klog) * BLOCK + threadIdx.x; }
        else { chunk = (unsigned)(t >> zone_lo); low = t & (((uint64_t)1 << zone_lo) - 1); }
        uint64_t i = ((uint64_t)chunk << zone_lo) | low;
        for (unsigned m = 0; m < S.npairs; m++) {
            const uint64_t x = ((i >> S.a[m]) ^ (i >> S.b[m])) & 1u;
            i ^= (x << S.a[m]) | (x << S.b[m]);
        }
        amp_t *to = D.dst[chunk] + (((uint64_t)me << zone_lo) | low);
        if (DEAL && Rl.nrelays && chunk != me) {
            const uint64_t lb = low / BLOCK;                            // block of the chunk: uniform over the workgroup
            if (lb >= Rl.nb_direct) {
                unsigned path = (unsigned)((lb - Rl.nb_direct) / Rl.nb_relay);
                if (path >= Rl.nrelays) path = Rl.nrelays - 1;            // (the last relay takes the remainder)
                const uint64_t first = ((uint64_t)Rl.nb_direct + (uint64_t)path * Rl.nb_relay) * BLOCK;
                to = Rl.stage[path] + (uint64_t)(me * W + chunk) * Rl.stage_amps + (low - first);
            }
        }
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), to);
    }
}

// ---------------------------------------------------------------------------
// K3b  controlled modular multiply for M registers too large for an LDS tile (M > 12): IN PLACE through a bounded staging
// buffer, a batch of 2^M-blocks at a time.  k_cam_big_gather computes the new rows [0, R) of every control-set block of the
// batch into the staging buffer (R = C in the closed form: rows >= C keep their values; R = 2^M with a source table), with the
// reference's order of summation (ascending source row, starting from 0: Q:393, Q:409-412); k_cam_big_scatter copies them
// back over the block.  No second state buffer and no buffer swap: the path works on a shard view and on a register that fills
// the card.  Traffic: 64 B per rewritten amplitude (gathered read + staged write + staged read + write).
// The 16-B source reads of one block scatter over its 2^M * 16 bytes, so all workgroups that work on a block sit on ONE XCD
// (blockIdx mod 8 under the round-robin placement) and consecutive workgroups of that XCD take consecutive chunks of the same
// block: the block's lines are fetched into that XCD's L2 once (256 KiB at M = 14 ... 2 MiB at M = 17 of the 4 MiB).
// TABLE = false: closed form (control outside the M register, C <= 2^M, no 32-bit wrap);  true: CSR source table.
// ---------------------------------------------------------------------------
struct CamBig {
    unsigned M, R;           // rows staged per block
    int      ctl;            // control bit (>= M), or -1: every block of the view takes part
    unsigned C, d, Cd, inv;
    uint64_t first, nblk;    // the batch: control-set blocks [first, first + nblk) of the view
    unsigned chunks;         // ceil(R / 1024)
};

__device__ __forceinline__ uint64_t cam_big_block_base(const CamBig &P, uint64_t tb)
{
    const uint64_t blockno = (P.ctl >= (int)P.M) ? (insert_zero(tb, (unsigned)P.ctl - P.M) | ((uint64_t)1 << ((unsigned)P.ctl - P.M))) : tb;
    return blockno << P.M;
}

// (block of the batch, chunk of 1024 rows) of workgroup-iteration `it` under the XCD-aware order; false: past the end
__device__ __forceinline__ bool cam_big_item(const CamBig &P, uint64_t it, uint64_t *j, unsigned *chunk)
{
    const unsigned xcd = blockIdx.x & 7u;
    const uint64_t mine = (P.nblk + 7u - xcd) >> 3;               // blocks j of the batch with j mod 8 == xcd
    if (it >= mine * P.chunks) return false;
    *j = (it / P.chunks) * 8u + xcd;
    *chunk = (unsigned)(it % P.chunks);
    return true;
}

template <bool TABLE>
__global__ __launch_bounds__(256) void k_cam_big_gather(const amp_t *__restrict__ amp, amp_t *__restrict__ stage, CamBig P,
                                                        const uint32_t *__restrict__ off, const uint32_t *__restrict__ srcs)
{
    uint64_t j; unsigned chunk;
    for (uint64_t it = blockIdx.x >> 3; cam_big_item(P, it, &j, &chunk); it += gridDim.x >> 3) {
        const amp_t *blk = amp + cam_big_block_base(P, P.first + j);
        amp_t *to = stage + j * P.R;
#pragma unroll
        for (unsigned k = 0; k < 4; k++) {
            const unsigned g = chunk * 1024u + k * 256u + threadIdx.x;
            if (g >= P.R) continue;
            amp_t acc; acc.x = 0.0; acc.y = 0.0;
            if (TABLE) {
                for (uint32_t s = off[g]; s < off[g + 1]; s++) { const amp_t v = blk[srcs[s]]; acc.x += v.x; acc.y += v.y; }
            } else if (g % P.d == 0) {
                uint32_t s0 = (uint32_t)(((uint64_t)(g / P.d) * P.inv) % P.Cd);
                for (uint32_t q = 0; q < P.d; q++, s0 += P.Cd) { const amp_t v = blk[s0]; acc.x += v.x; acc.y += v.y; }
            }
            __builtin_nontemporal_store(acc, to + g);
        }
    }
}

__global__ __launch_bounds__(256) void k_cam_big_scatter(amp_t *__restrict__ amp, const amp_t *__restrict__ stage, CamBig P)
{
    uint64_t j; unsigned chunk;
    for (uint64_t it = blockIdx.x >> 3; cam_big_item(P, it, &j, &chunk); it += gridDim.x >> 3) {
        amp_t *blk = amp + cam_big_block_base(P, P.first + j);
        const amp_t *from = stage + j * P.R;
        amp_t v[4];
#pragma unroll
        for (unsigned k = 0; k < 4; k++) {
            const unsigned g = chunk * 1024u + k * 256u + threadIdx.x;
            if (g < P.R) v[k] = __builtin_nontemporal_load(from + g);
        }
#pragma unroll
        for (unsigned k = 0; k < 4; k++) {
            const unsigned g = chunk * 1024u + k * 256u + threadIdx.x;
            if (g < P.R) blk[g] = v[k];
        }
    }
}

}  // namespace qcx
