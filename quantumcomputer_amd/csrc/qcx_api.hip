// qcx_api.hip -- the C ABI of libqcx.so (include/qcx.h) on top of the gfx950
// kernels in qcx_kernels.h.  Host side of the drop-in boundary: argument checks,
// launch geometry, the two gate schedules (Q:678-690, Q:712-737), MT19937 and
// measurement bookkeeping.  HIP only -- nothing here computes on the CPU.
#include "../../include/qcx.h"
#include "qcx_kernels.h"

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

using namespace qcx;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_last_error[256] = "";

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            snprintf(g_last_error, sizeof g_last_error, "%s: %s", #expr, hipGetErrorString(e_)); \
            return (e_ == hipErrorOutOfMemory) ? QCX_INSUFFICIENT_MEMORY : QCX_HIP_ERROR;      \
        }                                                                                      \
    } while (0)

#define set_error(...) snprintf(g_last_error, sizeof g_last_error, __VA_ARGS__)

#define QCX_TRY(expr)                     \
    do {                                  \
        int s_ = (expr);                  \
        if (s_ != QCX_NO_ERROR) return s_; \
    } while (0)

extern "C" const char *qcx_last_error(void) { return g_last_error; }

extern "C" const char *qcx_version(void) { return "qcx 0.1.0 (gfx950)"; }

extern "C" const char *qcx_status_string(int s)
{
    switch (s) {
    case QCX_NO_ERROR: return "NO_ERROR";
    case QCX_INSUFFICIENT_MEMORY: return "INSUFFICIENT_MEMORY";
    case QCX_BAD_ARGUMENTS: return "BAD_ARGUMENTS";
    case QCX_PERIOD_NOT_FOUND: return "PERIOD_NOT_FOUND";
    case QCX_UNKNOWN_ERROR: return "UNKNOWN_ERROR";
    case QCX_HIP_ERROR: return "HIP_ERROR";
    case QCX_BAD_QUBIT: return "BAD_QUBIT";
    case QCX_UNSUPPORTED: return "UNSUPPORTED";
    default: return "?";
    }
}

extern "C" int qcx_device_count(int *count)
{
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *count = 0; snprintf(g_last_error, sizeof g_last_error, "hipGetDeviceCount: %s", hipGetErrorString(e)); return QCX_HIP_ERROR; }
    *count = c;
    return QCX_NO_ERROR;
}

extern "C" int qcx_set_device(int device) { HIP_TRY(hipSetDevice(device)); return QCX_NO_ERROR; }

// ---------------------------------------------------------------------------
// launch configuration (tunable at run time so one build can be swept on the GPU)
// ---------------------------------------------------------------------------
struct Tune {
    long h_variant   = 0;      // 0: auto, 1: always pair form, 2: wave-tile form where it applies
    long h_ppt       = 1;      // pairs per thread, pair form
    long h_nt        = 3;      // nontemporal: bit 0 loads, bit 1 stores (pair form, q >= 3)
    long h_wave_nt   = 3;      // same for the wave-tile form in auto mode
    long h_wc        = 0;      // pair form: per-wave-contiguous load order
    long h_block     = 64;     // pair form: threads per block (64/128/256/512)
    long h_streams_log2 = 0;   // pair form: deal tiles as 2^k interleaved streams (3 = one per XCD)
    long h_skew      = 0;      // pair form: stream j is rotated by j * skew tiles inside its segment
    long h_grid_cap  = 0;      // 0: one tile per block (no cap)
    long h_wave_r    = 4;      // registers per lane, wave-tile form (2, 4 or 8)
    long h_wave_block = 256;   // wave-tile form: threads per block (64 or 256)
    long h_wave_maxq = 7;      // auto: use the wave-tile form for q <= this
    long ph_apt      = 1;      // measured (tools/experiments/tune_phase.py): one amplitude per lane, one wave per block,
    long ph_grid_cap = 0;      // nontemporal, 2 or 4 interleaved streams: 6.4-6.9 TB/s on the touched quarter
    long ph_block    = 64;
    long ph_nt       = 1;
    long ph_streams_log2 = -1; // -1: auto (1 when the lowest mask bit >= 8, else 2)
    long ph_lines    = 1;      // masks with a bit below 3: the whole-line kernel k_phase_lines (0: k_phase for every mask)
    long cam_grid_cap = 0;     // (round 3: one tile per workgroup, 2.80-2.89 ms per gate at n = 30 against 2.85-3.0 with 4096 workgroups)
                               // (measured n = 30, C = 21, M = 5: whole tiles 2.9-3.4 ms = 17.2 GB at 5.0-5.9 TB/s, partial 2.8-2.9 ms = 14.5 GB at 5.0-5.2)
    long cam_logT    = 8;      // modular multiply: tile = 2^this amplitudes, at least the 2^M block (n = 30, C = 21, M = 5: 2^11 2.65 ms, 2^10 2.50, 2^9 2.3-2.6, 2^8 2.3; 2^12 4.2)
    long cam_skip    = 1;      // modular multiply: the 128-B lines of a 2^M block above row C are neither sources nor destinations: not read
    long cam_nt_lines = 0;     // modular multiply: completely rewritten lines leave as nontemporal stores (the partly rewritten one through L2)
    long fuse_T      = 11;     // fused passes: tile = 2^T amplitudes in LDS (8..12)
    long fuse_c      = 4;      // fused passes: contiguous low bits of a tile (runs of 16 * 2^c bytes)
    long fuse_grid_cap = 65536; // (round 4, chained passes: 65536 beats 24576 by 1-2 %, one per tile loses 15 % on the n = 30 exact Shor circuit)
                               // workgroups of the one-tile-per-workgroup form (each walks several tiles: the table fill at kernel start is amortised)
    long fuse_qround = 1;      // tolerance mode: rounds of the shape H D H D run as straight-line code (FUSE_QROUND)
    long fuse_q3_cap = 65536;  // workgroups of k_fused_q3 with tables (round 3, in place: 3072 best; round 4, chained: n = 28 inverse QFT 6.12 ms with 3072, 5.85 with 65536;
                               // n = 30 Shor circuit 18.7 with 65536, 20.5 with one per tile)
    long fuse_q3_cap_exact = 0;  // all-Hadamard radix-8 passes (no tables to stage): one workgroup per tile (round 4, chained passes: n = 30 sweep 20.2 ms with 8192, 19.2 with one per tile)
    long fuse_q3 = 1;          // tolerance mode: radix-8 fast rounds on 2^12 tiles when they save a pass (k_fused_q3)
    long fuse_x8     = 1;      // bit-exact phase-dominated passes: the walk on 8 amplitudes per thread (k_fused_x8, round 5)
    long fuse_x8_T   = 12;     // ... on tiles of 2^this amplitudes (10 .. 12) with
    long fuse_x8_c   = 4;      // ... this many contiguous low bits
    long fuse_streams_log2 = -1;   // which tile a workgroup of k_fused_q3 / k_fused_x8 / k_fused_rounds takes (fuse_stream_tile): the 2^this low bits of its slot number ...
    long fuse_streams_pos  = -1;   // ... go to this bit of the tile number (+ 1; 0 = on top).  -1: by the kind of pass (launch_pass)
    long fuse_x8t    = 1;      // tolerance mode: radix-8 fast rounds run on the k_fused_x8 shell (hand-written round, K6x-t) instead of k_fused_q3
    long fuse_x8_min_tiles_log2 = 2;   // ... on registers of at least 2^this tiles
    long fuse_x8_ratio = 1;    // ... for passes without multiplies that hold at least this many phases per Hadamard
    long fuse_x8_map = 1;      // ... the wave number rides on the tile bits most gates of a round test (0: plain ascending order)
    long fuse_x8_cap = 65536;  // ... workgroups
    long fuse_gen    = 1;      // a pending reset / collapse + circuit front is GENERATED inside the first fused pass behind it (no write pass, no read)
    long fuse_chain_dir = -1;  // chained passes: 0 = a pass stores gathered so that the NEXT tile is contiguous (reads whole tiles); 1 = stores its own tile contiguously, the next pass gathers; -1 = by the kind of chain
    long cam_block   = 0;      // threads per workgroup of k_camodc (0: 256; 1024 for tiles of 2^12 amplitudes)
    long cam_stage_mb = 1024;  // M > 12: staging buffer of the in-place modular multiply (MiB; at least one 2^M-block)
    long fuse_compact = 1;     // behind a circuit front whose M register stays on a small orbit: the flush runs on a compact copy of the state (compact_chain)
    long fuse_compact_lazy = 1; // ... and stays compact behind a whole-circuit entry point until something other than measure_state looks at the state
    long fuse_plan_cache = 1;   // a flush whose inputs (shape, mode, knobs, front, gate list) are bit for bit those of the last one reuses its plan and, when nothing was uploaded since, its records on the device
    long fuse_expand_fused = 1;  // the last pass of a compact chain stores the real register itself (k_fused_x8's expanding store) instead of k_expand_compact
    long fuse_cols_tol = 1;    // tolerance mode: the by-columns pass of a compact chain keeps its merged diagonals (fast rounds inside k_gen_cols)
    long fuse_cols_cap = 12288; // workgroups of a k_gen_cols launch (a workgroup's prologue is long; n = 30 Shor circuit 11.06-11.09 ms with 12288-24576, 11.14-11.3 with 65536, 12.07 with 1536; tolerance 7.4-7.8 against 8.1)
    long fuse_cols_waves = 0;  // waves per workgroup of k_gen_cols (4 ... 8; the first four generate and store the tile); 0: 8 for launches of at most 1024 tiles, else 4
    long fuse_expand_direct = 1; // k_expand_compact: 1 = gather the compact sources straight from memory (8 per thread in flight) instead of staging 64 blocks in LDS
    long fuse_gen_cols = 1;    // the generated first pass by COLUMNS of the four lowest M-register bits (K6g, k_gen_cols)
    long fuse_zskip  = 1;      // passes behind a circuit front: waves whose share of the tile is all +0 skip the rounds (FusePass::zskip)
    long fuse_zskip_maxw = 2;  // ... on tiles of at most 2^(8 + this) amplitudes
    long fuse_lowtile = 1;     // a pass whose hot bits lie below bit 12 takes the whole low end of the index as its (contiguous) tile
    long fuse_q3_c3 = 1;       // tolerance mode: radix-8 passes with 2^3-amplitude runs (9 hot bits) when that saves a pass
    long fuse_front = 1;       // a pending reset / collapse is written together with the closed-form front of the queue (K0b)
    long fuse_tol_T = 10;      // tolerance mode: tile bits of diagonal passes when that costs no extra pass (0: same as the rest)
    long fuse_tol_occ = 6;     // tolerance-mode passes (merged diagonals): waves per SIMD the kernel is built for (6 or 8)
    long fuse_rounds_occ = 7;  // rounds-form passes: k_fused_rounds built for this many waves per SIMD (6 or 7; 0 = the general kernel)
    long fuse_T_phase = 10;    // tile bits of phase-dominated passes (one tile per workgroup, not pipelined); 0 = same as the rest
    long fuse_c_phase = 4;
    long fuse_phase_ratio = 6; // a pass is phase-dominated when it holds at least this many phases per H (and nothing else)
    long fuse_camruns = 1;     // rounds form: fold runs of permutation-type modular multiplies into one gather
    long fuse_hsweep_T = 12;   // tile geometry of an all-Hadamard tail when it saves passes (0: never)
    long fuse_hsweep_c = 3;
    long fuse_dbg    = 0;      // diagnostics (tools/experiments/probe_pass.py): bit 0 skip the gates, bit 1 skip the stores, bit 2 skip the fill of a rounds pass
    long fuse_rounds = 1;      // fused passes: rounds form (4 amplitudes per thread in registers, radix-4 H steps)
    long fuse_ldsdma = 1;      // fused passes: fill the tile with global_load_lds (LDS-DMA)
    long fuse_max_queue = 4096;
    long fuse_chain  = 1;      // runs of consecutive rounds-form passes go through the register's second buffer: every pass but the first reads
                               // contiguous tiles, only its stores are gathered, the last one stores the identity layout again (FusePass)
    long fuse_chain_min_n = 20; // ... for registers of at least 2^this amplitudes
    long meas_block_log = 0;   // parallel measurement: 2^this amplitudes per block (8..13); 0 = from the shard size
    long meas_parallel = 1;    // 0: always the single-wave sequential scan
    long meas_min_log2 = 12;   // shards below 2^this amplitudes use the single-wave scan (tools/experiments/probe_shots.py)
    long meas_dbg = 0;         // K4c diagnostics: bit 0 = no look-back (every binade guess from cum_in alone: times the pass without it); bit 1 = k_meas_fast hands over to the walk at its third candidate
    long meas_host_out = 1;    // the scan's result is written straight into pinned host memory (no copy back on the stream)
    long meas_fast = 1;        // K4c: the scan's events by k_meas_fast (list of candidate records, 8 waves) with k_meas_walk as the fallback; 0: the walk alone
    long meas_spin_limit = 4000000;   // K4c: polls a look-back may spend on one window before it gives up (the block is then scanned exactly)
};
static Tune g_tune;
static std::mutex g_tune_mutex;
// every entry point works on ONE consistent snapshot of the tunables (qcx_tune_set may run on another thread)
static Tune tune_now() { std::lock_guard<std::mutex> lock(g_tune_mutex); return g_tune; }

extern "C" int qcx_tune_set(const char *key, long value)
{
#define K(name) if (!strcmp(key, #name)) { std::lock_guard<std::mutex> lock(g_tune_mutex); g_tune.name = value; return QCX_NO_ERROR; }
    K(h_variant) K(h_ppt) K(h_nt) K(h_wave_nt) K(h_wc) K(h_block) K(h_streams_log2) K(h_skew) K(h_grid_cap) K(h_wave_r) K(h_wave_block) K(h_wave_maxq) K(ph_apt) K(ph_grid_cap) K(ph_block) K(ph_nt) K(ph_streams_log2) K(ph_lines) K(cam_grid_cap) K(cam_skip) K(cam_nt_lines) K(cam_logT) K(cam_block) K(cam_stage_mb) K(meas_parallel) K(meas_min_log2) K(meas_block_log) K(meas_spin_limit) K(meas_dbg) K(meas_fast) K(meas_host_out) K(fuse_T) K(fuse_c) K(fuse_grid_cap) K(fuse_max_queue) K(fuse_ldsdma) K(fuse_rounds) K(fuse_dbg) K(fuse_hsweep_T) K(fuse_hsweep_c) K(fuse_camruns) K(fuse_T_phase) K(fuse_c_phase) K(fuse_phase_ratio) K(fuse_rounds_occ) K(fuse_tol_occ) K(fuse_qround) K(fuse_tol_T) K(fuse_front) K(fuse_q3) K(fuse_q3_cap) K(fuse_q3_cap_exact) K(fuse_chain) K(fuse_chain_dir) K(fuse_chain_min_n) K(fuse_q3_c3) K(fuse_lowtile) K(fuse_gen) K(fuse_gen_cols) K(fuse_cols_waves) K(fuse_cols_cap) K(fuse_cols_tol) K(fuse_compact) K(fuse_expand_direct) K(fuse_compact_lazy) K(fuse_expand_fused) K(fuse_plan_cache) K(fuse_zskip) K(fuse_zskip_maxw) K(fuse_x8) K(fuse_x8_T) K(fuse_x8_c) K(fuse_x8_map) K(fuse_x8_cap) K(fuse_x8_ratio) K(fuse_x8_min_tiles_log2) K(fuse_x8t) K(fuse_streams_log2) K(fuse_streams_pos)
#undef K
    return QCX_BAD_ARGUMENTS;
}

extern "C" long qcx_tune_get(const char *key)
{
#define K(name) if (!strcmp(key, #name)) { std::lock_guard<std::mutex> lock(g_tune_mutex); return g_tune.name; }
    K(h_variant) K(h_ppt) K(h_nt) K(h_wave_nt) K(h_wc) K(h_block) K(h_streams_log2) K(h_skew) K(h_grid_cap) K(h_wave_r) K(h_wave_block) K(h_wave_maxq) K(ph_apt) K(ph_grid_cap) K(ph_block) K(ph_nt) K(ph_streams_log2) K(ph_lines) K(cam_grid_cap) K(cam_skip) K(cam_nt_lines) K(cam_logT) K(cam_block) K(cam_stage_mb) K(meas_parallel) K(meas_min_log2) K(meas_block_log) K(meas_spin_limit) K(meas_dbg) K(meas_fast) K(meas_host_out) K(fuse_T) K(fuse_c) K(fuse_grid_cap) K(fuse_max_queue) K(fuse_ldsdma) K(fuse_rounds) K(fuse_dbg) K(fuse_hsweep_T) K(fuse_hsweep_c) K(fuse_camruns) K(fuse_T_phase) K(fuse_c_phase) K(fuse_phase_ratio) K(fuse_rounds_occ) K(fuse_tol_occ) K(fuse_qround) K(fuse_tol_T) K(fuse_front) K(fuse_q3) K(fuse_q3_cap) K(fuse_q3_cap_exact) K(fuse_chain) K(fuse_chain_dir) K(fuse_chain_min_n) K(fuse_q3_c3) K(fuse_lowtile) K(fuse_gen) K(fuse_gen_cols) K(fuse_cols_waves) K(fuse_cols_cap) K(fuse_cols_tol) K(fuse_compact) K(fuse_expand_direct) K(fuse_compact_lazy) K(fuse_expand_fused) K(fuse_plan_cache) K(fuse_zskip) K(fuse_zskip_maxw) K(fuse_x8) K(fuse_x8_T) K(fuse_x8_c) K(fuse_x8_map) K(fuse_x8_cap) K(fuse_x8_ratio) K(fuse_x8_min_tiles_log2) K(fuse_x8t) K(fuse_streams_log2) K(fuse_streams_pos)
#undef K
    return -1;
}

static inline unsigned grid_for(uint64_t work_items, uint64_t per_block, long cap, unsigned block_threads = 1024)
{
    uint64_t g = (work_items + per_block - 1) / per_block;
    if (g == 0) g = 1;
    if (cap > 0 && g > (uint64_t)cap) g = (uint64_t)cap;
    // kernels grid-stride, so any cap is valid; HIP wants grid * block < 2^32 threads (n = 33, 34 registers get there)
    const uint64_t hard = std::min<uint64_t>(0x7fffffffULL, 0xffffffffULL / block_threads);
    if (g > hard) g = hard;
    return (unsigned)g;
}

// ---------------------------------------------------------------------------
// per-device scratch for the shard-level entry points
// ---------------------------------------------------------------------------
struct Workspace {
    double     *partials = nullptr;     // NORM_BLOCKS doubles + 1
    MeasureOut *mout = nullptr;
    MeasureOut *h_mout = nullptr;       // pinned
    size_t      meas_slots_pending = 0;
    size_t      meas_clean_slots = 0;   // ... and this many look-back slots from the start of meas_look are known to read "not published"
    bool        meas_clean = false;     // the last parallel scan completed: its walk kernel left ticket and candidate count at zero
    double     *h_scalar = nullptr;     // pinned
    uint32_t   *tab = nullptr;          // camodc CSR table (off + srcs)
    size_t      tab_cap = 0;
    amp_t      *cam_stage = nullptr;    // staging buffer of the M > 12 modular multiply (K3b)
    size_t      cam_stage_cap = 0;
    MeasBlock  *meas_blocks = nullptr;
    unsigned    meas_cap = 0;
    unsigned   *meas_stats = nullptr, *h_meas_stats = nullptr; // [slow-path blocks, blocks] of the last scan
    MeasCands  *meas_cands = nullptr;                           // k_meas_fast: the candidate records of the scan in flight ...
    MeasResume *meas_resume = nullptr;                          // ... and what it leaves for k_meas_walk
    meas_slot_t *meas_look = nullptr;   // K4c: look-back area (one allocation): agg, incl per workgroup, gincl per group (filled with ones), then
                                        // gsum, gcount per group and the ticket (zeroed)
    MeasBlock  *meas_up = nullptr;      // K4c: the levels above the records (sums of 64, 64^2, ... records), back to back
};
static const unsigned NORM_BLOCKS = 2048;
static std::mutex g_ws_mutex;
static std::mutex g_ws_use[64];       // one user of a device's scratch (norm / measurement) at a time
static Workspace g_ws[64];

static int workspace(Workspace **out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return QCX_HIP_ERROR;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Workspace &w = g_ws[dev];
    if (!w.partials) {
        HIP_TRY(hipMalloc(&w.partials, (NORM_BLOCKS + 1) * sizeof(double)));
        HIP_TRY(hipMalloc(&w.mout, sizeof(MeasureOut)));
        HIP_TRY(hipHostMalloc(&w.h_mout, sizeof(MeasureOut)));
        HIP_TRY(hipHostMalloc(&w.h_scalar, sizeof(double)));
        HIP_TRY(hipMalloc(&w.meas_stats, 2 * sizeof(unsigned)));
        HIP_TRY(hipMalloc(&w.meas_cands, sizeof(MeasCands)));
        HIP_TRY(hipMalloc(&w.meas_resume, sizeof(MeasResume)));
        HIP_TRY(hipHostMalloc(&w.h_meas_stats, 2 * sizeof(unsigned)));
        w.h_meas_stats[0] = w.h_meas_stats[1] = 0;
    }
    *out = &w;
    return QCX_NO_ERROR;
}

// ---------------------------------------------------------------------------
// shard-level launches
// ---------------------------------------------------------------------------
extern "C" int qcx_shard_reset(void *amp, unsigned n_local, int holds_index_one, void *stream)
{
    if (!amp || n_local > 40) return QCX_BAD_ARGUMENTS;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(amp, 0, (size_t)16 << n_local, st));
    if (holds_index_one) {
        if (n_local == 0) return QCX_BAD_ARGUMENTS;          // a 1-amplitude shard has no index 1
        hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, st, (amp_t *)amp, (uint64_t)1);
        HIP_TRY(hipGetLastError());
    }
    return QCX_NO_ERROR;
}

extern "C" int qcx_shard_collapse(void *amp, unsigned n_local, int64_t local_index, void *stream)
{
    if (!amp || n_local > 40) return QCX_BAD_ARGUMENTS;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(amp, 0, (size_t)16 << n_local, st));
    if (local_index >= 0) {
        if ((uint64_t)local_index >= ((uint64_t)1 << n_local)) return QCX_BAD_ARGUMENTS;
        hipLaunchKernelGGL(k_set_one, dim3(1), dim3(64), 0, st, (amp_t *)amp, (uint64_t)local_index);
        HIP_TRY(hipGetLastError());
    }
    return QCX_NO_ERROR;
}

// -0 components become +0 (what the reference's mat-vec does to every amplitude at every gate, Q:393-413): for a state the
// caller wrote, before the first gate runs on it (k_canon_zeros)
extern "C" int qcx_shard_canon_zeros(void *amp, unsigned n_local, void *stream)
{
    if (!amp || n_local > 40) return QCX_BAD_ARGUMENTS;
    const uint64_t count = (uint64_t)1 << n_local;
    hipLaunchKernelGGL(k_canon_zeros, dim3(grid_for(count, 256 * 4, 0, 256)), dim3(256), 0, (hipStream_t)stream, (amp_t *)amp, count);
    HIP_TRY(hipGetLastError());
    return QCX_NO_ERROR;
}

extern "C" int qcx_shard_fill_random(void *amp, unsigned n_local, uint64_t first_global, uint64_t seed, double scale, void *stream)
{
    if (!amp || n_local > 40) return QCX_BAD_ARGUMENTS;
    const uint64_t count = (uint64_t)1 << n_local;
    hipLaunchKernelGGL(k_fill_random, dim3(grid_for(count, 256 * 4, 0, 256)), dim3(256), 0, (hipStream_t)stream,
                       (amp_t *)amp, count, first_global, seed, scale);
    HIP_TRY(hipGetLastError());
    return QCX_NO_ERROR;
}

template <int PPT, bool NTL, bool NTS, bool WC, int BLOCK>
static void launch_h_pair(const Tune &t, amp_t *a, unsigned q, uint64_t npairs, hipStream_t st)
{
    const unsigned grid = grid_for(npairs, BLOCK * PPT, t.h_grid_cap, BLOCK);
    // stream-interleaved tile order only for an uncapped power-of-two grid that divides evenly
    unsigned glog = 0, slog = 0;
    if ((grid & (grid - 1)) == 0 && (uint64_t)grid * (BLOCK * PPT) == npairs) {
        glog = 31u - (unsigned)__builtin_clz(grid);
        if (t.h_streams_log2 > 0 && (unsigned)t.h_streams_log2 <= glog) slog = (unsigned)t.h_streams_log2;
    }
    if (slog && t.h_skew > 0) slog |= (unsigned)t.h_skew << 8;
    hipLaunchKernelGGL((k_h_pair<PPT, NTL, NTS, WC, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, a, q, npairs, glog, slog);
}

template <int PPT, int BLOCK>
static void launch_h_pair_flags(const Tune &t, amp_t *a, unsigned q, uint64_t npairs, hipStream_t st, long nt, bool wc)
{
    const int f = (int)(nt & 3) | (wc ? 4 : 0);
    switch (f) {
    case 0: launch_h_pair<PPT, false, false, false, BLOCK>(t, a, q, npairs, st); break;
    case 3: launch_h_pair<PPT, true, true, false, BLOCK>(t, a, q, npairs, st); break;
    case 4: launch_h_pair<PPT, false, false, true, BLOCK>(t, a, q, npairs, st); break;
    case 7: launch_h_pair<PPT, true, true, true, BLOCK>(t, a, q, npairs, st); break;
    case 1: case 5: launch_h_pair<PPT, true, false, false, BLOCK>(t, a, q, npairs, st); break;
    default: launch_h_pair<PPT, false, true, false, BLOCK>(t, a, q, npairs, st); break;
    }
}

template <int PPT>
static void launch_h_pair_block(const Tune &t, amp_t *a, unsigned q, uint64_t npairs, hipStream_t st, long nt, bool wc)
{
    switch (t.h_block) {
    case 64:  launch_h_pair_flags<PPT, 64>(t, a, q, npairs, st, nt, wc); break;
    case 128: launch_h_pair_flags<PPT, 128>(t, a, q, npairs, st, nt, wc); break;
    case 512: launch_h_pair_flags<PPT, 512>(t, a, q, npairs, st, nt, wc); break;
    default:  launch_h_pair_flags<PPT, 256>(t, a, q, npairs, st, nt, wc); break;
    }
}

static void stream_map(unsigned grid, uint64_t covered, uint64_t total, long want_slog, unsigned *glog, unsigned *slog)
{
    *glog = 0; *slog = 0;
    if ((grid & (grid - 1)) == 0 && covered == total) {       // uncapped power-of-two grid that divides evenly
        *glog = 31u - (unsigned)__builtin_clz(grid);
        if (want_slog > 0 && (unsigned)want_slog <= *glog) *slog = (unsigned)want_slog;
    }
}

template <int Q, int R, bool NTL, bool NTS>
static void launch_h_wave_q(const Tune &t, amp_t *a, uint64_t namps, hipStream_t st)
{
    const uint64_t ntiles = namps / (64 * R);
    unsigned glog, slog;
    if (t.h_wave_block == 64) {
        const unsigned grid = grid_for(ntiles, 1, t.h_grid_cap, 64);
        stream_map(grid, grid, ntiles, t.h_streams_log2, &glog, &slog);
        hipLaunchKernelGGL((k_h_wave<Q, R, NTL, NTS, 64>), dim3(grid), dim3(64), 0, st, a, ntiles, glog, slog);
    } else {
        const unsigned grid = grid_for(ntiles, 4 /* waves per 256-thread block */, t.h_grid_cap, 256);
        stream_map(grid, (uint64_t)grid * 4, ntiles, t.h_streams_log2, &glog, &slog);
        hipLaunchKernelGGL((k_h_wave<Q, R, NTL, NTS, 256>), dim3(grid), dim3(256), 0, st, a, ntiles, glog, slog);
    }
}

template <int R, bool NTL, bool NTS>
static bool launch_h_wave(const Tune &t, amp_t *a, unsigned q, uint64_t namps, hipStream_t st)
{
    switch (q) {
    case 0: launch_h_wave_q<0, R, NTL, NTS>(t, a, namps, st); return true;
    case 1: launch_h_wave_q<1, R, NTL, NTS>(t, a, namps, st); return true;
    case 2: launch_h_wave_q<2, R, NTL, NTS>(t, a, namps, st); return true;
    case 3: launch_h_wave_q<3, R, NTL, NTS>(t, a, namps, st); return true;
    case 4: launch_h_wave_q<4, R, NTL, NTS>(t, a, namps, st); return true;
    case 5: launch_h_wave_q<5, R, NTL, NTS>(t, a, namps, st); return true;
    case 6: launch_h_wave_q<6, R, NTL, NTS>(t, a, namps, st); return true;
    case 7: if constexpr (R >= 4) { launch_h_wave_q<7, R, NTL, NTS>(t, a, namps, st); return true; } return false;
    case 8: if constexpr (R >= 8) { launch_h_wave_q<8, R, NTL, NTS>(t, a, namps, st); return true; } return false;
    default: return false;
    }
}

template <int R>
static bool launch_h_wave_flags(const Tune &t, amp_t *a, unsigned q, uint64_t namps, hipStream_t st, long nt)
{
    if ((nt & 3) == 3) return launch_h_wave<R, true, true>(t, a, q, namps, st);
    return launch_h_wave<R, false, false>(t, a, q, namps, st);
}

// Launch plan per target qubit, measured on MI355X (tools/experiments/tune_h.py, profiles/r01_tune_h_*.json).
// What matters is the ABSOLUTE pair distance 2^q * 16 B (the same q behaves the same at n = 26, 28
// and 30), i.e. how the two streams of a wave fall onto HBM channels and banks:
//   - the smallest work item wins everywhere: one wave, 2 x 16 B per lane (2 KiB in flight per wave);
//   - below q = 3 a pair shares a 128-B line, so the wave-tile (shuffle) form with whole-line
//     nontemporal accesses is used; from q = 3 the pair form with nontemporal loads and stores;
//   - dealing the tiles as 2^s interleaved streams (s = 1..3) moves the concurrently active windows
//     apart; the best s depends on q (q = 20..23 are the hard distances, s = 3 and 2 pairs/thread).
struct HPlan { int wave_form; int ppt; int block; int slog; };
static HPlan h_plan(unsigned q)
{
    if (q <= 2)  return {1, 0, 256, 2};
    if (q <= 6)  return {0, 1, 64, 2};
    if (q <= 17) return {0, 1, 64, 1};
    if (q <= 19) return {0, 1, 64, 0};
    if (q == 20) return {0, 1, 64, 1};
    if (q <= 23) return {0, 2, 64, 3};
    if (q == 24 || q == 26 || q == 27) return {0, 1, 64, 1};
    return {0, 1, 64, 0};
}

extern "C" int qcx_shard_hadamard(void *amp, unsigned n_local, unsigned q, void *stream)
{
    if (!amp || n_local == 0 || n_local > 40) return QCX_BAD_ARGUMENTS;
    if (q >= n_local) return QCX_BAD_QUBIT;
    hipStream_t st = (hipStream_t)stream;
    amp_t *a = (amp_t *)amp;
    const uint64_t namps = (uint64_t)1 << n_local, npairs = namps >> 1;

    Tune t = tune_now();                             // local copy: the entry point is re-entrant
    if (t.h_variant == 0) {                          // auto: the measured plan
        const HPlan pl = h_plan(q);
        t.h_variant = pl.wave_form ? 2 : 1;
        t.h_wave_r = 2; t.h_wave_block = pl.block;
        t.h_ppt = pl.ppt; t.h_block = pl.block;
        t.h_streams_log2 = pl.slog; t.h_nt = 3; t.h_wc = 0; t.h_grid_cap = 0;
    }
    int status = QCX_NO_ERROR;
    bool launched = false;
    // wave-tile form: needs whole 64*R tiles and the partner inside the tile
    const int R = (t.h_wave_r >= 8) ? 8 : (t.h_wave_r >= 4 ? 4 : 2);
    const unsigned tile_bits = (R == 8) ? 9 : (R == 4 ? 8 : 7);
    if (t.h_variant == 2 && q < tile_bits && n_local >= tile_bits + 2) {
        launched = (R == 8) ? launch_h_wave_flags<8>(t, a, q, namps, st, t.h_nt)
                 : (R == 4) ? launch_h_wave_flags<4>(t, a, q, namps, st, t.h_nt)
                            : launch_h_wave_flags<2>(t, a, q, namps, st, t.h_nt);
    }
    if (!launched) {
        long ppt = t.h_ppt;
        while (ppt > 1 && npairs < (uint64_t)512 * (uint64_t)ppt) ppt >>= 1;
        // nontemporal accesses only pay when a wave-instruction covers whole 128-B lines (q >= 3)
        const long nt = (q >= 3) ? t.h_nt : 0;
        const bool wc = t.h_wc != 0;
        switch (ppt) {
        case 8: launch_h_pair_block<8>(t, a, q, npairs, st, nt, wc); break;
        case 4: launch_h_pair_block<4>(t, a, q, npairs, st, nt, wc); break;
        case 2: launch_h_pair_block<2>(t, a, q, npairs, st, nt, wc); break;
        default: launch_h_pair_block<1>(t, a, q, npairs, st, nt, wc); break;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_last_error, sizeof g_last_error, "hadamard launch: %s", hipGetErrorString(e)); status = QCX_HIP_ERROR; }
    return status;
}

template <int NB, int APT, bool NT, int BLOCK>
static void launch_phase_cfg(const Tune &t, amp_t *a, unsigned b0, unsigned b1, double c, double s, uint64_t count, hipStream_t st)
{
    const unsigned grid = grid_for(count, (uint64_t)BLOCK * APT, t.ph_grid_cap, BLOCK);
    unsigned glog, slog;
    long want = t.ph_streams_log2;
    if (want < 0) want = (NB == 0 || b0 >= 8) ? 1 : 2;
    stream_map(grid, (uint64_t)grid * BLOCK * APT, count, want, &glog, &slog);
    hipLaunchKernelGGL((k_phase<NB, APT, NT, BLOCK>), dim3(grid), dim3(BLOCK), 0, st, a, b0, b1, c, s, count, glog, slog);
}

template <int NB, int APT, bool NT>
static void launch_phase_blk(const Tune &t, amp_t *a, unsigned b0, unsigned b1, double c, double s, uint64_t count, hipStream_t st)
{
    if (t.ph_block == 64) launch_phase_cfg<NB, APT, NT, 64>(t, a, b0, b1, c, s, count, st);
    else launch_phase_cfg<NB, APT, NT, 256>(t, a, b0, b1, c, s, count, st);
}

template <int NB>
static void launch_phase(const Tune &t, amp_t *a, unsigned b0, unsigned b1, double c, double s, uint64_t count, hipStream_t st)
{
    long apt = t.ph_apt;
    while (apt > 1 && count < (uint64_t)256 * (uint64_t)apt) apt >>= 1;
    // nontemporal only when the touched runs are whole 128-B lines (lowest mask bit >= 3, or no mask)
    const bool nt = t.ph_nt != 0 && (NB == 0 || b0 >= 3);
    if (apt >= 4)      { if (nt) launch_phase_blk<NB, 4, true>(t, a, b0, b1, c, s, count, st); else launch_phase_blk<NB, 4, false>(t, a, b0, b1, c, s, count, st); }
    else if (apt == 2) { if (nt) launch_phase_blk<NB, 2, true>(t, a, b0, b1, c, s, count, st); else launch_phase_blk<NB, 2, false>(t, a, b0, b1, c, s, count, st); }
    else               { if (nt) launch_phase_blk<NB, 1, true>(t, a, b0, b1, c, s, count, st); else launch_phase_blk<NB, 1, false>(t, a, b0, b1, c, s, count, st); }
}

extern "C" int qcx_shard_phase(void *amp, unsigned n_local, uint64_t mask, double cos_t, double sin_t, void *stream)
{
    if (!amp || n_local > 40) return QCX_BAD_ARGUMENTS;
    if (n_local < 64 && (mask >> n_local) != 0) return QCX_BAD_QUBIT;
    const int nb = __builtin_popcountll(mask);
    if (nb > 2) return QCX_BAD_ARGUMENTS;
    hipStream_t st = (hipStream_t)stream;
    amp_t *a = (amp_t *)amp;
    unsigned b0 = 0, b1 = 0;
    if (nb >= 1) b0 = (unsigned)__builtin_ctzll(mask);
    if (nb == 2) b1 = 63u - (unsigned)__builtin_clzll(mask);
    const uint64_t count = ((uint64_t)1 << n_local) >> nb;
    const Tune t = tune_now();
    if (nb >= 1 && b0 < 3 && t.ph_lines && n_local >= 9) {
        // a mask bit inside a 128-B line: one lane per amplitude of every touched line (k_phase_lines)
        unsigned lowmask = 0, hb[2] = {0, 0}, nh = 0;
        const unsigned mbits[2] = {b0, b1};                          // ascending
        for (int k = 0; k < nb; k++) { if (mbits[k] < 3) lowmask |= 1u << mbits[k]; else hb[nh++] = mbits[k]; }
        const uint64_t lines_amps = ((uint64_t)1 << n_local) >> nh;
        const int store_all = (lowmask & 3u) ? 1 : 0;               // 16- or 32-B granularity: write whole lines back (measured: masked 32-B stores 4.0 ms vs 2.6)
        const unsigned grid = grid_for(lines_amps, 64, 0, 64);
        unsigned glog, slog;
        stream_map(grid, (uint64_t)grid * 64, lines_amps, 2, &glog, &slog);
        if (nh == 0) hipLaunchKernelGGL((k_phase_lines<0, 64>), dim3(grid), dim3(64), 0, st, a, 0u, 0u, lowmask, store_all, cos_t, sin_t, lines_amps, glog, slog);
        else if (nh == 1) hipLaunchKernelGGL((k_phase_lines<1, 64>), dim3(grid), dim3(64), 0, st, a, hb[0], 0u, lowmask, store_all, cos_t, sin_t, lines_amps, glog, slog);
        else hipLaunchKernelGGL((k_phase_lines<2, 64>), dim3(grid), dim3(64), 0, st, a, hb[0], hb[1], lowmask, store_all, cos_t, sin_t, lines_amps, glog, slog);
        HIP_TRY(hipGetLastError());
        return QCX_NO_ERROR;
    }
    if (nb == 0) launch_phase<0>(t, a, 0, 0, cos_t, sin_t, count, st);
    else if (nb == 1) launch_phase<1>(t, a, b0, 0, cos_t, sin_t, count, st);
    else launch_phase<2>(t, a, b0, b1, cos_t, sin_t, count, st);
    HIP_TRY(hipGetLastError());
    return QCX_NO_ERROR;
}

static unsigned gcd_u32(unsigned a, unsigned b) { while (b) { unsigned t = a % b; a = b; b = t; } return a; }

// modular inverse of a mod m (gcd(a, m) == 1, m >= 1); 0 when m == 1
static unsigned modinv_u32(unsigned a, unsigned m)
{
    if (m == 1) return 0;
    long long t = 0, nt = 1, r = m, nr = a % m;
    while (nr != 0) {
        long long qd = r / nr;
        long long tmp = t - qd * nt; t = nt; nt = tmp;
        tmp = r - qd * nr; r = nr; nr = tmp;
    }
    if (t < 0) t += m;
    return (unsigned)t;
}

// M > 12: the 2^M-block does not fit an LDS tile -> in place through the device's staging buffer, a batch of blocks at a time
// (K3b: k_cam_big_gather / k_cam_big_scatter).  Works on any view (a whole register, a shard, a slice of at least one block).
static int shard_camodc_large(amp_t *a, unsigned n_local, unsigned M, unsigned C, unsigned A, int ctl, hipStream_t st, const Tune &tn)
{
    if (M > 26) return QCX_UNSUPPORTED;
    const uint64_t blk = (uint64_t)1 << M;
    const uint64_t fmax = (C < blk ? C : blk) - 1;
    const bool wraps = (uint64_t)(C - 1) * fmax > 0xffffffffULL;
    const bool closed = (ctl < 0 || ctl >= (int)M) && C <= blk && !wraps;        // as in qcx_shard_camodc
    CamBig P;
    memset(&P, 0, sizeof P);
    P.M = M; P.C = C;
    P.ctl = (ctl >= (int)M) ? ctl : -1;               // a control inside the M register is encoded in the table
    P.R = closed ? C : (unsigned)blk;
    P.chunks = (P.R + 1023u) / 1024u;
    const uint64_t all_blocks = (uint64_t)1 << (n_local - M);
    const uint64_t touched = (P.ctl >= 0) ? all_blocks >> 1 : all_blocks;
    if (touched == 0) return QCX_NO_ERROR;

    Workspace *w;
    QCX_TRY(workspace(&w));
    std::lock_guard<std::mutex> use(g_ws_use[w - g_ws]);     // table and staging buffer are per-device scratch: one user at a time
    const size_t row_bytes = (size_t)P.R * sizeof(amp_t);
    const size_t want = std::max<size_t>(row_bytes, std::min<size_t>((size_t)std::max<long>(tn.cam_stage_mb, 1) << 20, row_bytes * touched));
    {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        if (w->cam_stage_cap < want) {
            HIP_TRY(hipStreamSynchronize(st));
            if (w->cam_stage) HIP_TRY(hipFree(w->cam_stage));
            w->cam_stage = nullptr; w->cam_stage_cap = 0;
            if (hipMalloc(&w->cam_stage, want) != hipSuccess) {
                (void)hipGetLastError();
                set_error("hipMalloc(%zu bytes) for the staging buffer of a modular multiply with M = %u", want, M);
                return QCX_INSUFFICIENT_MEMORY;
            }
            w->cam_stage_cap = want;
        }
    }
    const uint32_t *d_off = nullptr, *d_srcs = nullptr;
    if (closed) {
        P.d = gcd_u32(A, C); P.Cd = C / P.d; P.inv = modinv_u32(A / P.d, P.Cd);
    } else {
        // CSR of sources per destination built as Q:611-654 maps them
        std::vector<uint32_t> cnt(blk + 1, 0), dst(blk);
        for (uint64_t f = 0; f < blk; f++) {
            uint32_t d = (uint32_t)f;
            const bool on = (ctl >= 0 && ctl < (int)M) ? ((f >> ctl) & 1u) : true;
            if (on && f < C) d = (uint32_t)(((uint32_t)(A * (uint32_t)f)) % C) & (uint32_t)(blk - 1);
            dst[f] = d; cnt[d + 1]++;
        }
        for (uint64_t g = 0; g < blk; g++) cnt[g + 1] += cnt[g];
        std::vector<uint32_t> tab(2 * blk + 1);
        memcpy(tab.data(), cnt.data(), (blk + 1) * sizeof(uint32_t));
        std::vector<uint32_t> fill(cnt.begin(), cnt.end() - 1);
        for (uint64_t f = 0; f < blk; f++) tab[blk + 1 + fill[dst[f]]++] = (uint32_t)f;       // ascending f per destination
        const size_t need = tab.size() * sizeof(uint32_t);
        {
            std::lock_guard<std::mutex> lock(g_ws_mutex);
            if (w->tab_cap < need) {
                if (w->tab) HIP_TRY(hipFree(w->tab));
                w->tab = nullptr; w->tab_cap = 0;
                HIP_TRY(hipMalloc(&w->tab, need));
                w->tab_cap = need;
            }
        }
        HIP_TRY(hipStreamSynchronize(st));                   // the previous table may still be in use
        HIP_TRY(hipMemcpy(w->tab, tab.data(), need, hipMemcpyHostToDevice));
        d_off = w->tab; d_srcs = w->tab + blk + 1;
    }
    const uint64_t per_batch = std::max<uint64_t>(1, w->cam_stage_cap / row_bytes);
    for (uint64_t first = 0; first < touched; first += per_batch) {
        P.first = first; P.nblk = std::min<uint64_t>(per_batch, touched - first);
        const uint64_t items = ((P.nblk + 7) / 8) * P.chunks;                    // per XCD, at most
        const unsigned grid = 8u * (unsigned)std::min<uint64_t>(items, 1024);
        if (closed) hipLaunchKernelGGL((k_cam_big_gather<false>), dim3(grid), dim3(256), 0, st, (const amp_t *)a, w->cam_stage, P, d_off, d_srcs);
        else hipLaunchKernelGGL((k_cam_big_gather<true>), dim3(grid), dim3(256), 0, st, (const amp_t *)a, w->cam_stage, P, d_off, d_srcs);
        hipLaunchKernelGGL(k_cam_big_scatter, dim3(grid), dim3(256), 0, st, a, (const amp_t *)w->cam_stage, P);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));                       // rare path: table and staging buffer are free again when the lock is
    return QCX_NO_ERROR;
}

extern "C" int qcx_shard_camodc(void *amp, unsigned n_local, unsigned M, unsigned C, unsigned A, int ctl, void *stream)
{
    if (!amp || n_local > 40 || C == 0 || M > n_local) return QCX_BAD_ARGUMENTS;
    if (ctl >= (int)n_local) return QCX_BAD_QUBIT;
    A %= C;
    hipStream_t st = (hipStream_t)stream;
    amp_t *a = (amp_t *)amp;
    const Tune tn = tune_now();
    if (M > 12) return shard_camodc_large(a, n_local, M, C, A, ctl, st, tn);     // 2^M-block no longer fits the LDS tile

    CamodcParams P;
    memset(&P, 0, sizeof P);
    P.M = M;
    { const unsigned want = (unsigned)std::min<long>(12, std::max<long>(8, tn.cam_logT)); P.logT = (M < want) ? want : M; }     // tile: >= the 2^M block
    if (P.logT > n_local) P.logT = n_local;
    // a control at or above the M register is squeezed out of the tile numbering (tiles hold control-set amplitudes
    // only: half the vector is never read); that needs at least two tiles' worth of index space
    if (ctl >= (int)M && P.logT == n_local && P.logT > M) P.logT--;
    if (P.logT < M) return QCX_BAD_ARGUMENTS;
    P.ctl = ctl;
    P.C = C;
    P.skip = tn.cam_skip != 0; P.ntl = tn.cam_nt_lines != 0;
    const uint64_t blk = (uint64_t)1 << M;
    const uint64_t all_tiles = (uint64_t)1 << (n_local - P.logT);
    const uint64_t ctl_tiles = (ctl >= (int)M) ? all_tiles >> 1 : all_tiles;
    const size_t lds = ((size_t)16 << P.logT) + (((size_t)2 << M) + 15) / 16 * 16;       // tile + the source table of the permutation case

    // closed form is valid when the control is outside the M register, every residue fits the
    // register (C <= 2^M) and the reference's 32-bit product A*f cannot wrap (Q:598-600, Q:639)
    const uint64_t fmax = (C < blk ? C : blk) - 1;
    const bool wraps = (uint64_t)(C - 1) * fmax > 0xffffffffULL;
    const bool closed = (ctl < 0 || ctl >= (int)M) && C <= blk && !wraps;

    if (closed) {
        P.d = gcd_u32(A, C);            // gcd(0, C) = C
        P.Cd = C / P.d;
        P.inv = modinv_u32(A / P.d, P.Cd);
        P.ntiles = ctl_tiles;
        const unsigned grid = grid_for(P.ntiles, 1, tn.cam_grid_cap);
        // tiles of 2^12 amplitudes (M = 12) get 1024 threads: with 256 the two workgroups a CU's LDS holds are 8 waves, too few to
        // overlap their load -> barrier -> store chains (n = 30: 4.55 -> 3.25 ms; 2^11 tiles: 512 threads 3.27, 256 threads 3.14)
        const unsigned blockT = (tn.cam_block > 0) ? (unsigned)tn.cam_block : (P.logT >= 12 ? 1024u : 256u);
#define QCX_CAM_LAUNCH(B) hipLaunchKernelGGL((k_camodc<B>), dim3(grid), dim3(B), lds, st, a, P)
        if (blockT >= 1024) QCX_CAM_LAUNCH(1024); else if (blockT >= 512) QCX_CAM_LAUNCH(512); else QCX_CAM_LAUNCH(256);
#undef QCX_CAM_LAUNCH
        HIP_TRY(hipGetLastError());
        return QCX_NO_ERROR;
    }

    // generic path: CSR of sources per destination built as Q:611-654 maps them
    std::vector<uint32_t> cnt(blk + 1, 0), dst(blk);
    for (uint64_t f = 0; f < blk; f++) {
        uint32_t d = (uint32_t)f;
        const bool on = (ctl >= 0 && ctl < (int)M) ? ((f >> ctl) & 1u) : true;
        if (on && f < C) d = (uint32_t)(((uint32_t)(A * (uint32_t)f)) % C) & (uint32_t)(blk - 1);
        dst[f] = d;
        cnt[d + 1]++;
    }
    for (uint64_t g = 0; g < blk; g++) cnt[g + 1] += cnt[g];
    std::vector<uint32_t> tab(blk + 1 + blk);
    memcpy(tab.data(), cnt.data(), (blk + 1) * sizeof(uint32_t));
    std::vector<uint32_t> fill(cnt.begin(), cnt.end() - 1);
    for (uint64_t f = 0; f < blk; f++) tab[blk + 1 + fill[dst[f]]++] = (uint32_t)f;   // ascending f per destination

    Workspace *w;
    QCX_TRY(workspace(&w));
    std::lock_guard<std::mutex> use(g_ws_use[w - g_ws]);     // the table is per-device scratch: one user at a time
    const size_t need = tab.size() * sizeof(uint32_t);
    {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        if (w->tab_cap < need) {
            if (w->tab) HIP_TRY(hipFree(w->tab));
            w->tab = nullptr; w->tab_cap = 0;
            HIP_TRY(hipMalloc(&w->tab, need));
            w->tab_cap = need;
        }
    }
    HIP_TRY(hipStreamSynchronize(st));                       // the previous table may still be in use
    HIP_TRY(hipMemcpy(w->tab, tab.data(), need, hipMemcpyHostToDevice));
    CamodcParams Pt = P;
    Pt.d = 1; Pt.Cd = C; Pt.inv = 0;
    if (ctl >= 0 && ctl < (int)M) Pt.ctl = -1;               // the table already encodes the control
    Pt.ntiles = (Pt.ctl >= (int)M) ? all_tiles >> 1 : all_tiles;
    const unsigned grid = grid_for(Pt.ntiles, 1, tn.cam_grid_cap);
    hipLaunchKernelGGL((k_camodc_table<256>), dim3(grid), dim3(256), lds, st, a, Pt, w->tab, w->tab + blk + 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));                       // rare path: finish before the table can be replaced
    return QCX_NO_ERROR;
}

extern "C" int qcx_shard_swap_bits(const void *src, void *dst, unsigned n_local, unsigned npairs,
                                   const unsigned *pos_a, const unsigned *pos_b, void *stream)
{
    if (!src || !dst || src == dst || n_local > 40 || npairs > 8 || (npairs && (!pos_a || !pos_b))) return QCX_BAD_ARGUMENTS;
    SwapBits S;
    memset(&S, 0, sizeof S);
    S.npairs = npairs;
    for (unsigned m = 0; m < npairs; m++) {
        if (pos_a[m] >= n_local || pos_b[m] >= n_local || pos_a[m] == pos_b[m]) return QCX_BAD_QUBIT;
        S.a[m] = pos_a[m]; S.b[m] = pos_b[m];
    }
    const uint64_t count = (uint64_t)1 << n_local;
    hipLaunchKernelGGL((k_swap_bits<256>), dim3(grid_for(count, 256, 0, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const amp_t *)src, (amp_t *)dst, count, S);
    HIP_TRY(hipGetLastError());
    return QCX_NO_ERROR;
}

extern "C" int qcx_shard_norm2(const void *amp, unsigned n_local, double *out, void *stream)
{
    if (!amp || !out || n_local > 40) return QCX_BAD_ARGUMENTS;
    Workspace *w;
    QCX_TRY(workspace(&w));
    std::lock_guard<std::mutex> use(g_ws_use[w - g_ws]);
    hipStream_t st = (hipStream_t)stream;
    const uint64_t count = (uint64_t)1 << n_local;
    const unsigned grid = grid_for(count, 256 * 8, NORM_BLOCKS);
    hipLaunchKernelGGL((k_norm_partial<256>), dim3(grid), dim3(256), 0, st, (const amp_t *)amp, count, w->partials);
    hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(64), 0, st, w->partials, grid, w->partials + NORM_BLOCKS);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(w->h_scalar, w->partials + NORM_BLOCKS, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *out = *w->h_scalar;
    return QCX_NO_ERROR;
}

extern "C" int qcx_shard_measure_scan(const void *amp, unsigned n_local, uint64_t first_global,
                                      uint64_t last_excluded, double cum_in, double r,
                                      int *found, uint64_t *index, double *cum_out, void *stream)
{
    if (!amp || !found || !index || !cum_out || n_local > 40) return QCX_BAD_ARGUMENTS;
    Workspace *w;
    QCX_TRY(workspace(&w));
    std::lock_guard<std::mutex> use(g_ws_use[w - g_ws]);
    hipStream_t st = (hipStream_t)stream;
    uint64_t count = (uint64_t)1 << n_local;
    if (first_global >= last_excluded) count = 0;
    else if (last_excluded - first_global < count) count = last_excluded - first_global;
    if (count == 0) { *found = 0; *index = 0; *cum_out = cum_in; return QCX_NO_ERROR; }
    const Tune tn = tune_now();
    const bool parallel = tn.meas_parallel != 0 && count >= ((uint64_t)1 << tn.meas_min_log2);
    // The scan's last kernel writes its result (a MeasureOut, 32 bytes) STRAIGHT into pinned host memory (meas_host_out, round 5): the
    // copy back used to be one more operation on the stream -- a blit kernel, 6-10 us of an attempt that takes 27 us at n = 7 and
    // 100 us at n = 20.  The kernels only write it (what the fast scan leaves for the walk travels in MeasResume, device memory).
    MeasureOut *const res = tn.meas_host_out ? w->h_mout : w->mout;
    if (!parallel) {
        // small shards: the strictly sequential single-wave scan
        hipLaunchKernelGGL(k_measure_scan, dim3(1), dim3(64), 0, st, (const amp_t *)amp, count, cum_in, r, res);
    } else {
        // exact parallel form (qcx_kernels.h, K4c): one read of the state (look-back for the binade guesses), a few tiny group
        // launches, the tree walk.  A "block" of 2^blog amplitudes is one RECORD (one wave); a workgroup takes four of them.
        unsigned blog = (unsigned)tn.meas_block_log;
        if (blog == 0) {
            unsigned bits = 0;
            while (bits < 63 && ((uint64_t)1 << bits) < count) bits++;
            blog = bits ? (bits - 1) / 2 : 8;
        }
        blog = std::min<unsigned>(std::max<unsigned>(blog, 8u), 11u);     // a record is one wave's 2^8 .. 2^11 amplitudes
        // k_meas_onepass takes four records per 256-thread workgroup and a launch holds fewer than 2^32 threads: a record size
        // forced too small for the register (meas_block_log = 8 at n = 34) is raised, not refused
        while (blog < 11u && ((((count + (((uint64_t)1 << blog) - 1)) >> blog) + 3u) / 4u) * 256u > 0xffffffffULL) blog++;
        const uint64_t nb64 = (count + (((uint64_t)1 << blog) - 1)) >> blog;
        if (nb64 > 0x7fffffffULL) return QCX_UNSUPPORTED;
        const unsigned nblocks = (unsigned)nb64;
        {
            std::lock_guard<std::mutex> lock(g_ws_mutex);
            if (w->meas_cap < nblocks) {
                if (w->meas_blocks) { HIP_TRY(hipFree(w->meas_blocks)); HIP_TRY(hipFree(w->meas_look)); HIP_TRY(hipFree(w->meas_up)); }
                w->meas_blocks = nullptr; w->meas_cap = 0; w->meas_clean = false; w->meas_clean_slots = 0;
                w->meas_look = nullptr; w->meas_up = nullptr;
                HIP_TRY(hipMalloc(&w->meas_blocks, ((size_t)nblocks + 4) * sizeof(MeasBlock)));
                HIP_TRY(hipMalloc(&w->meas_look, (2 * (size_t)nblocks + 4 * ((size_t)nblocks / 64 + 2) + 8) * sizeof(meas_slot_t)));
                HIP_TRY(hipMalloc(&w->meas_up, ((size_t)nblocks / 32 + 16) * sizeof(MeasBlock)));
                w->meas_cap = nblocks;
            }
        }
        {
            const unsigned nwg = (nblocks + 3u) / 4u, ngrp = (nwg + 63u) / 64u;
            MeasLookback LB;
            LB.agg = w->meas_look; LB.incl = LB.agg + nwg; LB.gsum = LB.incl + nwg; LB.gincl = LB.gsum + ngrp;
            LB.ticket = w->meas_cands->ticket;                       // (a fixed address: the scan's last kernel leaves it at zero for the next one)
            const bool clean = w->meas_clean;
            const size_t nslots = 2 * (size_t)nwg + 2 * (size_t)ngrp;
            if (!clean || nslots > w->meas_clean_slots) HIP_TRY(hipMemsetAsync(w->meas_look, 0xff, nslots * sizeof(meas_slot_t), st));
            const size_t slots_after = clean ? std::max(w->meas_clean_slots, nslots) : nslots;
            w->meas_clean = false;                                  // (true again when this call has completed)
            if (!clean) HIP_TRY(hipMemsetAsync(w->meas_cands, 0, 6 * sizeof(unsigned), st));       // candidate count + ticket
            const unsigned spin = (unsigned)std::max<long>(1000, tn.meas_spin_limit);
#define QCX_ONEPASS(B) hipLaunchKernelGGL((k_meas_onepass<B>), dim3(nwg), dim3(256), 0, st, (const amp_t *)amp, count, cum_in, LB, w->meas_blocks, spin, (unsigned)tn.meas_dbg, r)
            switch (blog) {
            case 8: QCX_ONEPASS(8); break;   case 9: QCX_ONEPASS(9); break;   case 10: QCX_ONEPASS(10); break;
            default: QCX_ONEPASS(11); break;
            }
#undef QCX_ONEPASS
            MeasLevels T;
            memset(&T, 0, sizeof T);
            T.lv[0] = w->meas_blocks; T.n[0] = nblocks; T.top = 0;
            MeasBlock *up = w->meas_up;
            // the events of the scan go to k_meas_fast (the launch over the records lists them), the walk stands behind it
            const bool fast = tn.meas_fast != 0 && T.n[0] > 64u;
            while (T.n[T.top] > 64u && T.top < 4) {
                const unsigned nin = T.n[T.top], nout = (nin + 63u) / 64u;
                hipLaunchKernelGGL(k_meas_groups, dim3(nout), dim3(64), 0, st, T.lv[T.top], nin, up, (fast && T.top == 0) ? w->meas_cands : (MeasCands *)nullptr,
                                   T.top == 0 ? w->meas_look : (meas_slot_t *)nullptr, (unsigned)nslots);
                T.top++;
                T.lv[T.top] = up; T.n[T.top] = nout;
                up += nout;
            }
            if (fast)
                hipLaunchKernelGGL(k_meas_fast, dim3(1), dim3(512), 0, st, (const amp_t *)amp, count, T, cum_in, r, res, res->stats, blog,
                                   (const MeasCands *)w->meas_cands, w->meas_resume, (unsigned)tn.meas_dbg);
            hipLaunchKernelGGL(k_meas_walk, dim3(1), dim3(64), 0, st, (const amp_t *)amp, count, T, cum_in, r, res, res->stats, blog,
                               fast ? (const MeasResume *)w->meas_resume : (const MeasResume *)nullptr, w->meas_cands,
                               T.top == 0 ? w->meas_look : (meas_slot_t *)nullptr, (unsigned)nslots);
            w->meas_slots_pending = slots_after;
        }
    }
    HIP_TRY(hipGetLastError());
    if (res != w->h_mout) HIP_TRY(hipMemcpyAsync(w->h_mout, w->mout, sizeof(MeasureOut), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (parallel) { w->h_meas_stats[0] = w->h_mout->stats[0]; w->h_meas_stats[1] = w->h_mout->stats[1]; w->meas_clean = true; w->meas_clean_slots = w->meas_slots_pending; }
    *found = w->h_mout->found;
    *index = first_global + w->h_mout->index;
    *cum_out = w->h_mout->cum;
    return QCX_NO_ERROR;
}

// diagnostics of the last parallel measurement scan on the current device: blocks redone sequentially / blocks
extern "C" int qcx_measure_last_stats(unsigned *slow_blocks, unsigned *blocks)
{
    Workspace *w;
    QCX_TRY(workspace(&w));
    if (slow_blocks) *slow_blocks = w->h_meas_stats[0];
    if (blocks) *blocks = w->h_meas_stats[1];
    return QCX_NO_ERROR;
}

// ---------------------------------------------------------------------------
// register (single-GPU handle): one in-place amplitude buffer in HBM
// ---------------------------------------------------------------------------
struct qcx_register {
    int        L, M;
    unsigned   n;
    uint64_t   dim;
    amp_t     *amp;
    hipStream_t own_stream, stream;
    hipEvent_t ev0, ev1;
    hipEvent_t *events;
    unsigned   n_events;
    amp_t     *scratch;         // second buffer, allocated on first use (chained passes)
    int        no_chain;        // the buffer pointer was handed out (qcx_device_pointer) or no second buffer fits: passes work in place
    int        nonfinite;       // the caller wrote a component that is not finite (or >= 2^500): every gate runs as a STRICT pass (K9: the
                                // mat-vec's own products, identity rows included) until a reset, a fill or a measurement replaces the state
    int        zeros_dirty;     // the caller wrote amplitudes (qcx_state_write / _load): they may hold -0, which the reference's gates would
                                // canonicalise (Q:393-413); one k_canon_zeros pass runs before the next gate
    int        fusion;          // 1: every gate call is queued (fused passes, qcx_fuse.inc.h); 0: only the whole-circuit entry points; -1: nothing
    int        composite;       // > 0 while a whole-circuit entry point is queueing its gates
    struct GateQueue *queue;
    int        basis_pending;   // the state IS the basis state basis_index (reset / collapse) but has not been written yet:
    uint64_t   basis_index;     // the next flush writes it, together with a closed-form gate prefix if the queue has one
    unsigned long fronts;       // basis-state fronts executed as one write pass (K0b)
    // a compact chain left the state in its compact form (qcx_fuse.inc.h, compact_chain): r->amp is stale until expand_pending()
    // runs -- at the next flush, unless measure_state gets there first (it scans the compact form and collapses lazily)
    int        compact_pending;
    unsigned long compact_measures;            // measurements that scanned the compact form
    amp_t     *compact_amp;     // inside r->scratch
    unsigned   compact_cb, compact_ncols;
    uint16_t   compact_orbit[16];
    struct ShardSet *sh;     // non-null: the register is sharded over several GPUs by this process (qcx_sharded.inc.h)
};

static int reg_camodc(qcx_register *r, unsigned C, unsigned A, unsigned ctl);

// a state the caller wrote may hold -0 components; the reference's next gate would turn them into +0 wherever they sit
// (Q:393-413), the gate kernels only where they act: one pass over the state before the first gate after such a write
static int canon_if_dirty(qcx_register *r)
{
    if (!r->zeros_dirty) return QCX_NO_ERROR;
    r->zeros_dirty = 0;
    return qcx_shard_canon_zeros(r->amp, r->n, r->stream);
}

#include "qcx_fuse.inc.h"
#include "qcx_sharded.inc.h"

#define FLUSH(r) QCX_TRY(fuse_flush(r))

static int reg_camodc(qcx_register *r, unsigned C, unsigned A, unsigned ctl)
{
    return qcx_shard_camodc(r->amp, r->n, (unsigned)r->M, C, A, (int)ctl, r->stream);
}

extern "C" int qcx_register_create_sharded(int L, int M, unsigned nshards, const int *devices, qcx_register **out)
{
    if (!out) return QCX_BAD_ARGUMENTS;
    *out = nullptr;
    if (L < 0 || M < 0 || L + M < 1 || L + M > 40) return QCX_BAD_ARGUMENTS;
    if (nshards == 1 && !(devices && devices[0] < 0)) {
        if (devices) QCX_TRY(qcx_set_device(devices[0]));
        return qcx_register_create(L, M, out);
    }
    qcx_register *r = (qcx_register *)calloc(1, sizeof(qcx_register));
    if (!r) return QCX_INSUFFICIENT_MEMORY;
    r->L = L; r->M = M; r->n = (unsigned)(L + M); r->dim = (uint64_t)1 << r->n;
    const int s = sh_create(L, M, nshards, devices, &r->sh);
    if (s != QCX_NO_ERROR) { free(r); return s; }
    if (!r->sh->dry && (hipEventCreate(&r->ev0) != hipSuccess || hipEventCreate(&r->ev1) != hipSuccess)) { sh_free(r->sh); free(r); return QCX_HIP_ERROR; }
    if (const char *e = getenv("QCX_SHARD_RELAYS")) {             // "4,5,6,7": GPUs without a shard that relay stripes of every trade
        int devs[8], nr = 0;
        for (const char *p = e; *p && nr < 8; nr++) { devs[nr] = atoi(p); while (*p && *p != ',') p++; if (*p == ',') p++; }
        const int rs = sh_set_relays(r->sh, (unsigned)nr, devs);
        if (rs != QCX_NO_ERROR) { sh_free(r->sh); (void)hipEventDestroy(r->ev0); (void)hipEventDestroy(r->ev1); free(r); return rs; }
    }
    *out = r;
    return QCX_NO_ERROR;
}

extern "C" int qcx_sharded_set_relays(qcx_register *r, unsigned nrelays, const int *devices)
{
    if (!r || !r->sh) return QCX_BAD_ARGUMENTS;
    return sh_set_relays(r->sh, nrelays, devices);
}

extern "C" int qcx_sharded_overlap_stats(qcx_register *r, unsigned *slices_log2, unsigned long *gates_in_windows)     // diagnostics
{
    if (!r || !r->sh) return QCX_BAD_ARGUMENTS;
    if (slices_log2) *slices_log2 = r->sh->sigma;
    if (gates_in_windows) *gates_in_windows = r->sh->overlapped_gates;
    return QCX_NO_ERROR;
}

extern "C" int qcx_sharded_relay_stats(qcx_register *r, unsigned *nrelays, unsigned long *relayed_bytes)     // diagnostics
{
    if (!r || !r->sh) return QCX_BAD_ARGUMENTS;
    if (nrelays) *nrelays = (unsigned)r->sh->relay_dev.size();
    if (relayed_bytes) *relayed_bytes = r->sh->relayed_bytes;
    return QCX_NO_ERROR;
}

// where qcx_register_create_sharded(devices = NULL) puts the shards when `visible` devices can be seen (visible <= 0: ask HIP)
extern "C" int qcx_spread_devices(unsigned nshards, int visible, int *devices_out)
{
    if (!devices_out || nshards < 1 || nshards > 16 || (nshards & (nshards - 1))) return QCX_BAD_ARGUMENTS;
    if (visible <= 0) { QCX_TRY(qcx_device_count(&visible)); if (visible < 1) { set_error("no HIP device"); return QCX_HIP_ERROR; } }
    sh_spread(nshards, visible, devices_out);
    return QCX_NO_ERROR;
}

// the pre-flight exchange check on demand (it runs by itself at creation when the shards sit on different devices)
extern "C" int qcx_sharded_selfcheck(qcx_register *r)
{
    if (!r || !r->sh) return QCX_BAD_ARGUMENTS;
    if (r->sh->dry) return QCX_UNSUPPORTED;
    std::vector<int> rel(r->sh->relay_dev);
    int prev = 0;
    (void)hipGetDevice(&prev);
    const int s = sh_selfcheck(r->sh, (unsigned)rel.size(), rel.data());
    (void)hipSetDevice(prev);
    return s;
}

extern "C" unsigned long qcx_sharded_selfchecks(const qcx_register *r) { return (r && r->sh) ? r->sh->selfchecks : 0ul; }

extern "C" unsigned qcx_register_shards(const qcx_register *r) { return r ? (r->sh ? r->sh->W : 1u) : 0u; }

extern "C" int qcx_sharded_stats(qcx_register *r, unsigned long *exchanges, unsigned long *pack_passes)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (exchanges) *exchanges = r->sh ? r->sh->exchanges : 0;
    if (pack_passes) *pack_passes = r->sh ? r->sh->pack_passes : 0;
    return QCX_NO_ERROR;
}

// diagnostics (not in the public header): the step list a dry-run register has scheduled since the last call, and the
// restoration of the identity layout on demand (measurement and read-back do it themselves)
extern "C" int qcx_sharded_trace(qcx_register *r, char *buf, size_t cap, size_t *need)
{
    if (!r || !r->sh || !need) return QCX_BAD_ARGUMENTS;
    *need = r->sh->trace.size() + 1;
    if (!buf || cap < *need) return QCX_INSUFFICIENT_MEMORY;
    memcpy(buf, r->sh->trace.c_str(), *need);
    r->sh->trace.clear();
    return QCX_NO_ERROR;
}
extern "C" int qcx_sharded_restore_identity(qcx_register *r) { return (r && r->sh) ? sh_identity(r->sh) : QCX_BAD_ARGUMENTS; }
extern "C" int qcx_sharded_layout(qcx_register *r, unsigned *perm, unsigned cap)
{
    if (!r || !r->sh || !perm || cap < r->n) return QCX_BAD_ARGUMENTS;
    for (unsigned q = 0; q < r->n; q++) perm[q] = r->sh->perm[q];
    return QCX_NO_ERROR;
}

extern "C" int qcx_register_create(int L, int M, qcx_register **out)
{
    if (!out) return QCX_BAD_ARGUMENTS;
    *out = nullptr;
    // QCX_SHARDS=N (N = 2, 4, 8, 16): every register this process creates is sharded over N devices -- how a program
    // written against the reference's interface (qcx_compat.h, host/qcx_shor) gets onto the 8 GPUs of a node without
    // an edit.  QCX_SHARD_DEVICES="0,0,1,1" places the shards (default: shard r on device r).
    if (const char *e = getenv("QCX_SHARDS")) {
        const int ns = atoi(e);
        if (ns > 1) {
            int devs[16];
            const char *d = getenv("QCX_SHARD_DEVICES");
            if (d && *d) {
                int i = 0;
                for (const char *p = d; *p && i < 16; i++) { devs[i] = atoi(p); while (*p && *p != ',') p++; if (*p == ',') p++; }
                for (int j = i; j < 16; j++) devs[j] = devs[i ? i - 1 : 0];
            }
            return qcx_register_create_sharded(L, M, (unsigned)ns, (d && *d) ? devs : nullptr, out);    // no list: spread over the visible devices
        }
    }
    if (L < 0 || M < 0 || L + M < 1 || L + M > 36) return QCX_BAD_ARGUMENTS;
    int ndev = 0;
    QCX_TRY(qcx_device_count(&ndev));
    if (ndev < 1) { snprintf(g_last_error, sizeof g_last_error, "no HIP device"); return QCX_HIP_ERROR; }
    qcx_register *r = (qcx_register *)calloc(1, sizeof(qcx_register));
    if (!r) return QCX_INSUFFICIENT_MEMORY;
    r->L = L; r->M = M; r->n = (unsigned)(L + M); r->dim = (uint64_t)1 << r->n;
    hipError_t e = hipMalloc(&r->amp, r->dim * sizeof(amp_t));
    if (e != hipSuccess) { free(r); snprintf(g_last_error, sizeof g_last_error, "hipMalloc: %s", hipGetErrorString(e)); return QCX_INSUFFICIENT_MEMORY; }
    // (each step is undone on a later failure: nothing of a half-built register stays behind)
    bool ok = hipStreamCreate(&r->own_stream) == hipSuccess;
    const bool have_stream = ok;
    const bool have_ev0 = ok && hipEventCreate(&r->ev0) == hipSuccess;
    const bool have_ev1 = have_ev0 && hipEventCreate(&r->ev1) == hipSuccess;
    ok = have_ev1;
    r->stream = r->own_stream;
    // the reference's buffers start zeroed by calloc-like GSL allocs only after reset; be deterministic
    if (ok && hipMemsetAsync(r->amp, 0, r->dim * sizeof(amp_t), r->stream) != hipSuccess) ok = false;
    if (!ok) {
        snprintf(g_last_error, sizeof g_last_error, "qcx_register_create: %s", hipGetErrorString(hipGetLastError()));
        if (have_ev1) (void)hipEventDestroy(r->ev1);
        if (have_ev0) (void)hipEventDestroy(r->ev0);
        if (have_stream) (void)hipStreamDestroy(r->own_stream);
        (void)hipFree(r->amp); free(r);
        return QCX_HIP_ERROR;
    }
    *out = r;
    return QCX_NO_ERROR;
}

extern "C" int qcx_register_destroy(qcx_register *r)
{
    if (!r) return QCX_NO_ERROR;
    if (r->sh) {
        const bool dry = r->sh->dry;
        sh_free(r->sh);
        if (!dry) { (void)hipEventDestroy(r->ev0); (void)hipEventDestroy(r->ev1); }
        free(r);
        return QCX_NO_ERROR;
    }
    (void)fuse_flush(r);
    (void)hipStreamSynchronize(r->stream);
    queue_free(r->queue);
    (void)hipEventDestroy(r->ev0); (void)hipEventDestroy(r->ev1);
    for (unsigned i = 0; i < r->n_events; i++) (void)hipEventDestroy(r->events[i]);
    free(r->events);
    (void)hipStreamDestroy(r->own_stream);
    (void)hipFree(r->amp);
    if (r->scratch) (void)hipFree(r->scratch);
    free(r);
    return QCX_NO_ERROR;
}

extern "C" unsigned qcx_num_qubits(const qcx_register *r) { return r ? r->n : 0; }
extern "C" unsigned long qcx_num_states(const qcx_register *r) { return r ? (unsigned long)r->dim : 0; }
extern "C" int qcx_L_size(const qcx_register *r) { return r ? r->L : 0; }
extern "C" int qcx_M_size(const qcx_register *r) { return r ? r->M : 0; }
extern "C" void *qcx_device_pointer(qcx_register *r)
{
    if (!r || r->sh) return nullptr;
    (void)fuse_flush(r);
    r->no_chain = 1;             // the caller may keep the pointer: from now on the state stays in THIS buffer (no chained passes, which alternate between two)
    return (void *)r->amp;
}

// shard-level gate list through the fusion scheduler: one queue (record buffers + the event that guards them) per
// (device, stream), one user at a time.  A queue's event is only ever recorded on its own stream; the owner of a
// stream drops the queue before destroying the stream (qcx_shard_release_stream) -- an event whose last record was on
// a destroyed stream must not be synchronised any more.
struct ShardQueue { GateQueue q; std::mutex use; };
static std::mutex g_shard_queue_mutex;
static std::map<std::pair<int, hipStream_t>, ShardQueue *> g_shard_queues;

extern "C" int qcx_shard_release_stream(void *stream)
{
    std::vector<ShardQueue *> gone;
    {
        std::lock_guard<std::mutex> lock(g_shard_queue_mutex);
        for (auto it = g_shard_queues.begin(); it != g_shard_queues.end();)
            if (it->first.second == (hipStream_t)stream) { gone.push_back(it->second); it = g_shard_queues.erase(it); } else ++it;
    }
    for (ShardQueue *sq : gone) {
        std::lock_guard<std::mutex> use(sq->use);
        if (sq->q.ev_valid) { (void)hipEventSynchronize(sq->q.ev); (void)hipEventDestroy(sq->q.ev); sq->q.ev_valid = false; }
        if (sq->q.d_ops) (void)hipFree(sq->q.d_ops);
        if (sq->q.h_ops) (void)hipHostFree(sq->q.h_ops);
        sq->q.d_ops = nullptr; sq->q.h_ops = nullptr;
    }
    for (ShardQueue *sq : gone) delete sq;
    return QCX_NO_ERROR;
}

static int descs_to_gates(unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates, std::vector<QGate> &out)
{
    for (unsigned k = 0; k < count; k++) {
        const qcx_gate_desc &d = gates[k];
        QGate g; memset(&g, 0, sizeof g);
        if (d.type == 0) {
            if (d.q >= n_local) return QCX_BAD_QUBIT;
            g.type = FUSE_H; g.q = d.q;
        } else if (d.type == 1) {
            if (n_local < 64 && (d.mask >> n_local) != 0) return QCX_BAD_QUBIT;
            if (__builtin_popcountll(d.mask) > 2) return QCX_BAD_ARGUMENTS;
            g.type = FUSE_PHASE; g.mask = d.mask; g.c = d.c; g.s = d.s;
        } else if (d.type == 2) {
            if (d.C == 0 || (d.q != 0xffffffffu && d.q >= n_local)) return QCX_BAD_ARGUMENTS;
            g.q = d.q; g.C = d.C; g.A = d.A % d.C;
            const bool closed = (d.q == 0xffffffffu) ? camodc_closed_form(n_local, M, d.C, g.A, M)   // "control" outside M
                                                     : camodc_closed_form(n_local, M, d.C, g.A, d.q);
            g.type = closed ? (uint32_t)FUSE_CAMODC : 99u;          // (M > 12: the planner keeps it out of the tile passes -- K3b, stand-alone)
        } else return QCX_BAD_ARGUMENTS;
        out.push_back(g);
    }
    return QCX_NO_ERROR;
}

// ---- compact circuits for a one-process-per-GPU host (quantumcomputer_amd/sharded.py; the C host: sh_compact) ----------------
// qcx_compact_plan: pure host.  The closed-form front of `gates` on basis state `basis` (as qcx_shard_basis_front consumes it)
// and, when the M register stays on a small orbit behind it (compact_orbit), the compact form's column bits and the orbit.
// *ncols = 0: no compact form.  Every rank calls it with the same arguments and gets the same answer.
extern "C" int qcx_compact_plan(unsigned n, unsigned M, uint64_t basis, unsigned count, const qcx_gate_desc *gates,
                                unsigned *used, unsigned *cb, unsigned *ncols, uint16_t *orbit16)
{
    if (!used || !cb || !ncols || !orbit16 || n == 0 || n > 40 || M > n || (count && !gates) || (basis >> n)) return QCX_BAD_ARGUMENTS;
    *used = *cb = *ncols = 0;
    std::vector<QGate> q;
    for (unsigned k = 0; k < count; k++) {
        std::vector<QGate> one;
        if (gates[k].type == 1 || descs_to_gates(n, M, 1, gates + k, one) != QCX_NO_ERROR) break;
        q.push_back(one[0]);
    }
    BasisFront B;
    const Tune tn = tune_now();
    const size_t k = (M <= 26) ? front_plan(n, M, basis, tn, q, &B) : 0;
    *used = (unsigned)k;
    std::vector<uint16_t> orbit;
    unsigned c = 0;
    if (!k || !tn.fuse_compact || !compact_orbit(B, M, orbit, &c)) return QCX_NO_ERROR;
    *cb = c; *ncols = (unsigned)orbit.size();
    for (size_t j = 0; j < orbit.size(); j++) orbit16[j] = orbit[j];
    return QCX_NO_ERROR;
}

// this rank's part of the front, written in the compact form ([L register][orbit column]); count = the front's gates (*used of
// qcx_compact_plan), first_global = REAL global index of the rank's amplitude 0, n_local_compact = n_local - M + cb
extern "C" int qcx_shard_compact_front(void *compact, unsigned n_local_compact, uint64_t first_global, unsigned n, unsigned M, uint64_t basis,
                                       unsigned count, const qcx_gate_desc *gates, unsigned cb, unsigned ncols, const uint16_t *orbit16, void *stream)
{
    if (!compact || !orbit16 || !count || !gates || n == 0 || n > 40 || M > n || ncols == 0 || ncols > 16 || cb < 2 || cb > 4 || (1u << cb) < ncols ||
        n_local_compact <= cb || n_local_compact > 40) return QCX_BAD_ARGUMENTS;
    std::vector<QGate> q;
    QCX_TRY(descs_to_gates(n, M, count, gates, q));
    BasisFront B;
    if (front_plan(n, M, basis, tune_now(), q, &B) != count) return QCX_BAD_ARGUMENTS;
    B.first = first_global;
    ExpandParams E;
    memset(&E, 0, sizeof E);
    E.M = M; E.cb = cb; E.ncols = ncols;
    for (unsigned j = 0; j < ncols; j++) E.orbit[j] = orbit16[j];
    const uint64_t nblocks = (uint64_t)1 << (n_local_compact - cb);
    hipLaunchKernelGGL(k_basis_front_compact, dim3(grid_for(nblocks, 256, 65536)), dim3(256), 0, (hipStream_t)stream, (amp_t *)compact, n_local_compact, B, E);
    HIP_TRY(hipGetLastError());
    return QCX_NO_ERROR;
}

// a rank's part of the real register from its compact form (n_local - M >= 6)
extern "C" int qcx_shard_expand_compact(const void *compact, void *real, unsigned n_local, unsigned M, unsigned cb, unsigned ncols,
                                        const uint16_t *orbit16, void *stream)
{
    if (!compact || !real || !orbit16 || n_local > 40 || M > 12 || n_local < M + 6 || ncols == 0 || ncols > 16 || cb < 2 || cb > 4 || (1u << cb) < ncols) return QCX_BAD_ARGUMENTS;
    ExpandParams E;
    memset(&E, 0, sizeof E);
    E.M = M; E.cb = cb; E.ncols = ncols;
    for (unsigned j = 0; j < ncols; j++) E.orbit[j] = orbit16[j];
    const uint64_t nchunks = ((uint64_t)1 << (n_local - M)) >> 6;
    hipLaunchKernelGGL(k_expand_compact, dim3(grid_for(nchunks, 1, 65536)), dim3(256), 0, (hipStream_t)stream, (const amp_t *)compact, (amp_t *)real, nchunks, E, (int)tune_now().fuse_expand_direct);
    HIP_TRY(hipGetLastError());
    return QCX_NO_ERROR;
}

extern "C" int qcx_shard_run_fused_mode(int mode, void *amp, unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates, void *stream);

extern "C" int qcx_shard_run_fused(void *amp, unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates, void *stream)
{
    return qcx_shard_run_fused_mode(1, amp, n_local, M, count, gates, stream);
}

// mode 1: bit-exact passes; 2: the tolerance mode's passes (merged diagonals; a phase with ONE mask bit -- its other qubit
// is a constant 1 of the shard -- joins the diagonal of its run as a constant factor)
extern "C" int qcx_shard_run_fused_mode(int mode, void *amp, unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates, void *stream)
{
    if (!amp || n_local == 0 || n_local > 40 || M > n_local || (count && !gates)) return QCX_BAD_ARGUMENTS;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return QCX_HIP_ERROR;
    ShardQueue *sq;
    {
        std::lock_guard<std::mutex> lock(g_shard_queue_mutex);
        ShardQueue *&slot = g_shard_queues[std::make_pair(dev, (hipStream_t)stream)];
        if (!slot) slot = new ShardQueue();
        sq = slot;
    }
    std::lock_guard<std::mutex> use(sq->use);
    qcx_register tmp;
    memset(&tmp, 0, sizeof tmp);
    tmp.L = (int)(n_local - M); tmp.M = (int)M; tmp.n = n_local; tmp.dim = (uint64_t)1 << n_local;
    tmp.amp = (amp_t *)amp; tmp.stream = tmp.own_stream = (hipStream_t)stream;
    tmp.fusion = mode == 2 ? 2 : 1; tmp.queue = &sq->q;
    tmp.no_chain = 1;                                    // a view of somebody else's memory: there is no second buffer behind it
    tmp.queue->gates.clear();
    QCX_TRY(descs_to_gates(n_local, M, count, gates, tmp.queue->gates));
    return fuse_flush(&tmp);
}

// The planner alone, on the host (no GPU needed): which passes / stand-alone gates a gate list becomes and the
// records the pass kernels would interpret.  actions[k] describes action k; records receives the raw 32-byte
// records of all passes back to back (action.rec_off / rec_cnt index into it, in records).
extern "C" int qcx_fusion_plan_mode(int mode, unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates,
                                    qcx_plan_action *actions, unsigned max_actions, unsigned *n_actions,
                                    qcx_fuse_record *records, size_t max_records, size_t *n_records);

extern "C" int qcx_fusion_plan(unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates,
                               qcx_plan_action *actions, unsigned max_actions, unsigned *n_actions,
                               qcx_fuse_record *records, size_t max_records, size_t *n_records)
{
    return qcx_fusion_plan_mode(1, n_local, M, count, gates, actions, max_actions, n_actions, records, max_records, n_records);
}

extern "C" int qcx_fusion_plan_mode(int mode, unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates,
                                    qcx_plan_action *actions, unsigned max_actions, unsigned *n_actions,
                                    qcx_fuse_record *records, size_t max_records, size_t *n_records)
{
    if (n_local == 0 || n_local > 40 || M > n_local || (count && !gates) || !n_actions || !n_records) return QCX_BAD_ARGUMENTS;
    if ((mode & 3) != 1 && (mode & 3) != 2) return QCX_BAD_ARGUMENTS;
    static_assert(sizeof(qcx_fuse_record) == sizeof(FuseOp), "record layout");
    qcx_register tmp;
    memset(&tmp, 0, sizeof tmp);
    tmp.L = (int)(n_local - M); tmp.M = (int)M; tmp.n = n_local; tmp.dim = (uint64_t)1 << n_local;
    std::vector<QGate> q;
    QCX_TRY(descs_to_gates(n_local, M, count, gates, q));
    std::vector<FuseAction> acts;
    std::vector<FuseOp> ops;
    // mode | 4: with chained passes (a register with a second buffer); | 8: as compact_chain plans the virtual register of a compact
    // chain (the first pass generated by columns: a tile of the M register's bits x 8 hot bits)
    fuse_plan(&tmp, tune_now(), q, acts, ops, (mode & 3) == 2, (mode & 4) != 0, (mode & 8) ? (((mode & 3) == 2 && tune_now().fuse_cols_tol) ? 2 : 1) : 0);
    *n_actions = (unsigned)acts.size();
    *n_records = ops.size();
    if (acts.size() > max_actions || ops.size() > max_records || (!actions && !acts.empty()) || (!records && !ops.empty()))
        return QCX_INSUFFICIENT_MEMORY;
    for (size_t k = 0; k < acts.size(); k++) {
        const FuseAction &a = acts[k];
        qcx_plan_action &o = actions[k];
        memset(&o, 0, sizeof o);
        o.fused = a.fused;
        if (!a.fused) { o.first_gate = (unsigned)a.gate; o.ngates = 1; continue; }
        o.first_gate = (unsigned)a.first_gate; o.ngates = (unsigned)a.ngates;
        o.T = a.P.T; o.c = a.P.c; o.nh = a.P.nh; o.nopipe = (unsigned)a.nopipe;
        memcpy(o.hbit, a.P.hbit, sizeof o.hbit);
        o.rounds_form = (unsigned)a.P.cam_ctl_local[0];
        o.rec_off = a.op_off; o.rec_cnt = a.op_cnt; o.nops = a.P.nops;
        o.table_bytes = (unsigned)a.P.cam_ctl_local[1]; o.table_rec_off = (unsigned)a.P.cam_ctl_local[2];
        o.diag_cnt = a.P.dg_cnt; o.diag_rec_off = a.P.dg_rec_off;
        o.chained = a.P.chained;
        memcpy(o.tl, a.tl, sizeof o.tl);
        memcpy(o.in_pos, a.P.in_pos, sizeof o.in_pos); memcpy(o.st_loc, a.P.st_loc, sizeof o.st_loc); memcpy(o.st_pos, a.P.st_pos, sizeof o.st_pos);
        o.nseg_in = a.P.nseg_in; o.nseg_out = a.P.nseg_out; o.nseg_lg = a.P.nseg_lg;
        static_assert(sizeof o.seg_in == sizeof a.P.seg_in, "segment layout");
        memcpy(o.seg_in, a.P.seg_in, sizeof o.seg_in); memcpy(o.seg_out, a.P.seg_out, sizeof o.seg_out); memcpy(o.seg_lg, a.P.seg_lg, sizeof o.seg_lg);
    }
    if (!ops.empty()) memcpy(records, ops.data(), ops.size() * sizeof(FuseOp));
    return QCX_NO_ERROR;
}

// One shard's part of the basis state |basis> of an (n, M) register -- amplitudes [first_global, first_global + 2^n_local) --
// written together with the longest prefix of `gates` (GLOBAL qubit numbers, identity layout) that has the closed form on
// a basis state: Hadamards on distinct qubits, then controlled modular multiplies (K0b).  Every rank of a multi-process
// host calls it with the same list and gets the same *used; no communication.  used = 0: the plain basis state.
extern "C" int qcx_shard_basis_front(void *amp, unsigned n_local, uint64_t first_global, unsigned n, unsigned M, uint64_t basis,
                                     unsigned count, const qcx_gate_desc *gates, unsigned *used, void *stream)
{
    if (!amp || !used || n_local == 0 || n_local > 40 || n < n_local || n > 40 || M > n_local || (count && !gates)) return QCX_BAD_ARGUMENTS;
    if (basis >> n) return QCX_BAD_ARGUMENTS;
    if (first_global & (((uint64_t)1 << n_local) - 1)) return QCX_BAD_ARGUMENTS;
    *used = 0;
    std::vector<QGate> q;
    for (unsigned k = 0; k < count; k++) {                 // the front ends at the first gate that cannot be part of it
        std::vector<QGate> one;
        if (gates[k].type == 1 || descs_to_gates(n, M, 1, gates + k, one) != QCX_NO_ERROR) break;
        q.push_back(one[0]);
    }
    BasisFront B;
    size_t k = 0;
    if (M <= 26) k = front_plan(n, M, basis, tune_now(), q, &B);
    hipStream_t st = (hipStream_t)stream;
    if (k == 0) {
        const uint64_t per = (uint64_t)1 << n_local;
        const bool mine = basis >= first_global && basis - first_global < per;
        return qcx_shard_collapse(amp, n_local, mine ? (int64_t)(basis - first_global) : -1, st);
    }
    B.first = first_global;
    QCX_TRY(launch_basis_front((amp_t *)amp, n_local, B, st));
    *used = (unsigned)k;
    return QCX_NO_ERROR;
}

// The host half of the basis-state circuit front alone (no GPU needed; test interface, not in the public header): how many
// leading gates of the list have the closed form on basis state `basis` of an (n_local, M) register, and the parameters
// k_basis_front would get (struct BasisFront, csrc/qcx_kernels.h).
extern "C" int qcx_front_plan(unsigned n_local, unsigned M, uint64_t basis, unsigned count, const qcx_gate_desc *gates,
                              unsigned *used, void *front_out, size_t front_bytes)
{
    if (n_local == 0 || n_local > 40 || M > n_local || (count && !gates) || !used || !front_out || front_bytes < sizeof(BasisFront)) return QCX_BAD_ARGUMENTS;
    std::vector<QGate> q;
    QCX_TRY(descs_to_gates(n_local, M, count, gates, q));
    BasisFront B;
    *used = (unsigned)front_plan(n_local, M, basis, tune_now(), q, &B);
    memcpy(front_out, &B, sizeof B);
    return QCX_NO_ERROR;
}

// gate fusion: 1 = queue gates and run them as fused LDS-tile passes; results are
// bit-identical to the per-gate kernels.  Observing calls (read, norm, measure, synchronize, timers) flush.
extern "C" int qcx_set_fusion(qcx_register *r, int enable)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (r->sh) { QCX_TRY(sh_flush(r->sh)); r->sh->fusion = enable >= 2 ? 2 : (enable >= 0 ? 1 : -1); return QCX_NO_ERROR; }   // (gates are always queued; -1 = one launch per gate, 2 = tolerance mode per shard)
    FLUSH(r);
    r->fusion = enable >= 2 ? 2 : (enable > 0 ? 1 : (enable < 0 ? -1 : 0));
    return QCX_NO_ERROR;
}

extern "C" int qcx_flush(qcx_register *r)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (r->sh) return sh_flush(r->sh);
    FLUSH(r);
    return QCX_NO_ERROR;
}

extern "C" int qcx_fusion_stats(qcx_register *r, unsigned long *passes, unsigned long *gates)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (r->sh) { if (passes) *passes = r->sh->fronts; if (gates) *gates = 0; return QCX_NO_ERROR; }     // (circuit fronts written in one pass; the per-device pass queues are not counted: see qcx_sharded_stats)
    if (passes) *passes = (r->queue ? r->queue->passes_launched : 0) + r->fronts;
    if (gates) *gates = r->queue ? r->queue->gates_fused : 0;
    return QCX_NO_ERROR;
}

// diagnostics (not in the public header): fused passes that went out of place through the second buffer; circuit fronts
// that were generated inside the first pass behind them instead of being written by a pass of their own
extern "C" int qcx_chain_stats(qcx_register *r, unsigned long *chained_passes)
{
    if (!r || !chained_passes) return QCX_BAD_ARGUMENTS;
    *chained_passes = (!r->sh && r->queue) ? r->queue->chained_passes : 0;
    return QCX_NO_ERROR;
}
extern "C" int qcx_gen_stats(qcx_register *r, unsigned long *generated_fronts)
{
    if (!r || !generated_fronts) return QCX_BAD_ARGUMENTS;
    *generated_fronts = (!r->sh && r->queue) ? r->queue->gen_fronts : 0;
    return QCX_NO_ERROR;
}
extern "C" int qcx_gen_cols_stats(qcx_register *r, unsigned long *by_columns)
{
    if (!r || !by_columns) return QCX_BAD_ARGUMENTS;
    *by_columns = (!r->sh && r->queue) ? r->queue->gen_cols : 0;
    return QCX_NO_ERROR;
}
extern "C" int qcx_compact_stats(qcx_register *r, unsigned long *compact_chains)
{
    if (!r || !compact_chains) return QCX_BAD_ARGUMENTS;
    *compact_chains = r->sh ? r->sh->compact_circuits : (r->queue ? r->queue->compact_chains : 0);
    return QCX_NO_ERROR;
}

// host logic only (no GPU): the expanding-store tables compact_chain would give the last pass of a compact chain -- T = 12, the
// pass's store order (st_pos / st_loc as qcx_fusion_plan reports them for the plan of the VIRTUAL register, mode | 8), an M
// register of M bits and cb column bits.  *ok = 0: that pass does not qualify (the separate expansion runs).
extern "C" int qcx_expand_store_plan(unsigned T, const unsigned char *st_pos, const unsigned char *st_loc, unsigned M, unsigned cb,
                                     unsigned char *xp_pos24, unsigned char *xp_loc24, unsigned char *xp_colloc4, int *ok)
{
    if (!st_pos || !st_loc || !xp_pos24 || !xp_loc24 || !xp_colloc4 || !ok || T > 16) return QCX_BAD_ARGUMENTS;
    FusePass P;
    memset(&P, 0, sizeof P);
    P.T = T;
    memcpy(P.st_pos, st_pos, 16); memcpy(P.st_loc, st_loc, 16);
    const std::vector<uint16_t> none;
    *ok = xp_setup(P, M, cb, none) ? 1 : 0;
    memcpy(xp_pos24, P.xp_pos, 24); memcpy(xp_loc24, P.xp_loc, 24); memcpy(xp_colloc4, P.xp_colloc, 4);
    return QCX_NO_ERROR;
}

// diagnostics: flushes that took their plan from the plan cache (GateQueue::pc)
extern "C" int qcx_plan_cache_stats(qcx_register *r, unsigned long *hits)
{
    if (!r || !hits) return QCX_BAD_ARGUMENTS;
    *hits = (!r->sh && r->queue) ? r->queue->plan_hits : 0;
    return QCX_NO_ERROR;
}

// diagnostics: compact chains whose LAST pass stored the real register itself (no k_expand_compact launch)
extern "C" int qcx_expanding_store_stats(qcx_register *r, unsigned long *expanding_stores)
{
    if (!r || !expanding_stores) return QCX_BAD_ARGUMENTS;
    *expanding_stores = (!r->sh && r->queue) ? r->queue->expanding_stores : 0;
    return QCX_NO_ERROR;
}

// diagnostics: measurements that scanned the compact form of a circuit's result instead of the register (no expansion written)
extern "C" int qcx_compact_measure_stats(qcx_register *r, unsigned long *compact_measures)
{
    if (!r || !compact_measures) return QCX_BAD_ARGUMENTS;
    *compact_measures = r->sh ? r->sh->compact_measures : r->compact_measures;
    return QCX_NO_ERROR;
}

extern "C" int qcx_register_set_stream(qcx_register *r, void *hip_stream)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (r->sh) return QCX_UNSUPPORTED;                 // one stream per shard, owned by the register
    FLUSH(r);
    HIP_TRY(hipStreamSynchronize(r->stream));
    r->stream = hip_stream ? (hipStream_t)hip_stream : r->own_stream;
    return QCX_NO_ERROR;
}

extern "C" int qcx_synchronize(qcx_register *r)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (r->sh) return sh_sync(r->sh);
    FLUSH(r);
    HIP_TRY(hipStreamSynchronize(r->stream));
    return QCX_NO_ERROR;
}

// ---- strict gates (K9): a register that was handed non-finite amplitudes -------------------------------------------------------
// did the caller just write something that is not finite (or >= 2^500) into [first, first + count)?  (scanned on the device)
static int note_nonfinite(qcx_register *r, uint64_t first, uint64_t count)
{
    if (!count || r->nonfinite) return QCX_NO_ERROR;
    Workspace *w;
    QCX_TRY(workspace(&w));
    std::lock_guard<std::mutex> use(g_ws_use[w - g_ws]);
    HIP_TRY(hipMemsetAsync(w->meas_stats, 0, sizeof(unsigned), r->stream));
    hipLaunchKernelGGL(k_scan_nonfinite, dim3(grid_for(count, 256 * 8, 65536, 256)), dim3(256), 0, r->stream, (const amp_t *)(r->amp + first), count, w->meas_stats);
    HIP_TRY(hipGetLastError());
    const unsigned keep = w->h_meas_stats[0];                // (the pinned word doubles as the last scan's statistics: put it back)
    HIP_TRY(hipMemcpyAsync(w->h_meas_stats, w->meas_stats, sizeof(unsigned), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (w->h_meas_stats[0]) r->nonfinite = 1;
    w->h_meas_stats[0] = keep;
    return QCX_NO_ERROR;
}

static int strict_gate(qcx_register *r, const QGate &g)
{
    FLUSH(r);                                       // (nothing is queued in this mode; a pending reset would have cleared it)
    r->zeros_dirty = 0;                             // every strict pass rewrites every amplitude as 0 + ...: canonical zeros
    const uint64_t dim = r->dim;
    if (g.type == FUSE_H) {
        if (r->n < 1) return QCX_BAD_ARGUMENTS;
        hipLaunchKernelGGL(k_strict_h, dim3(grid_for(dim >> 1, 256, 65536, 256)), dim3(256), 0, r->stream, r->amp, r->n, g.q, M_SQRT1_2, 0.0);
    } else if (g.type == FUSE_PHASE) {
        hipLaunchKernelGGL(k_strict_phase, dim3(grid_for(dim, 256, 65536, 256)), dim3(256), 0, r->stream, r->amp, dim, g.mask, g.c, g.s, 1.0, 0.0);
    } else {
        const unsigned M = (unsigned)r->M;
        if (M > 12) { set_error("c_amodc_gate on a state with non-finite amplitudes: M = %u > 12 is not supported", M); return QCX_UNSUPPORTED; }
        const size_t lds = ((size_t)16 << M) + ((size_t)2 << M);
        hipLaunchKernelGGL(k_strict_camodc, dim3(grid_for(dim >> M, 1, 65536, 256)), dim3(256), lds, r->stream, r->amp, r->n, M, g.C, g.A, (int)g.q, 1.0, 0.0);
    }
    HIP_TRY(hipGetLastError());
    return QCX_NO_ERROR;
}

extern "C" int qcx_reset_register(qcx_register *r)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (r->sh) return sh_reset(r->sh);
    if (r->queue) r->queue->gates.clear();         // pending gates act on a state that is being overwritten
    r->nonfinite = 0;
    r->zeros_dirty = 0;
    r->compact_pending = 0;
    if (r->fusion >= 0 && r->n >= 1) {             // lazily: the write happens at the next flush, fused with the circuit front (K0b)
        r->basis_pending = 1; r->basis_index = 1;
        return QCX_NO_ERROR;
    }
    r->basis_pending = 0;
    return qcx_shard_reset(r->amp, r->n, 1, r->stream);
}

extern "C" int qcx_hadamard_gate(unsigned q, qcx_register *r)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (q >= r->n) return QCX_BAD_QUBIT;
    if (r->sh) { SGate g; memset(&g, 0, sizeof g); g.type = FUSE_H; g.q = q; return sh_push(r->sh, g); }
    if (r->nonfinite) { QGate g; memset(&g, 0, sizeof g); g.type = FUSE_H; g.q = q; return strict_gate(r, g); }
    if (r->fusion > 0 || r->composite) { QGate g; memset(&g, 0, sizeof g); g.type = FUSE_H; g.q = q; return fuse_push(r, g); }
    FLUSH(r);                                         // (a lazily pending reset / collapse is written first)
    QCX_TRY(canon_if_dirty(r));
    return qcx_shard_hadamard(r->amp, r->n, q, r->stream);
}

// e^{i theta} the way the reference obtains it: gsl_complex_polar(1.0, theta) = (1*cos, 1*sin) (Q:526), which
// gcc -O2 compiles to ONE glibc sincos() call (cos and sin of the same argument are merged).  glibc's sincos
// and its stand-alone sin/cos can differ in the last bit (e.g. sin(0.20966817126512538)), so sincos is called
// explicitly here and in the oracle instead of leaving the choice to each compiler.
extern "C" void qcx_polar(double theta, double *cos_out, double *sin_out)
{
    double sn, cs;
    sincos(theta, &sn, &cs);
    *cos_out = 1.0 * cs;
    *sin_out = 1.0 * sn;
}

extern "C" int qcx_c_phase_shift_gate(unsigned c, unsigned t, double theta, qcx_register *r)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (c >= r->n || t >= r->n || c == t) return QCX_BAD_QUBIT;
    double er, ei;
    qcx_polar(theta, &er, &ei);
    if (r->sh) { SGate g; memset(&g, 0, sizeof g); g.type = FUSE_PHASE; g.q = c; g.q2 = t; g.c = er; g.s = ei; return sh_push(r->sh, g); }
    if (r->nonfinite) {
        QGate g; memset(&g, 0, sizeof g);
        g.type = FUSE_PHASE; g.mask = ((uint64_t)1 << c) | ((uint64_t)1 << t); g.c = er; g.s = ei;
        return strict_gate(r, g);
    }
    if (r->fusion > 0 || r->composite) {
        QGate g; memset(&g, 0, sizeof g);
        g.type = FUSE_PHASE; g.mask = ((uint64_t)1 << c) | ((uint64_t)1 << t); g.c = er; g.s = ei;
        return fuse_push(r, g);
    }
    FLUSH(r);
    QCX_TRY(canon_if_dirty(r));
    return qcx_shard_phase(r->amp, r->n, ((uint64_t)1 << c) | ((uint64_t)1 << t), er, ei, r->stream);
}

extern "C" int qcx_c_amodc_gate(unsigned C, unsigned long long atox, unsigned c, qcx_register *r)
{
    if (!r || C == 0) return QCX_BAD_ARGUMENTS;
    if (c >= r->n) return QCX_BAD_QUBIT;
    if (r->sh) { SGate g; memset(&g, 0, sizeof g); g.type = FUSE_CAMODC; g.q = c; g.C = C; g.A = (unsigned)(atox % C); return sh_push(r->sh, g); }
    if (r->nonfinite) { QGate g; memset(&g, 0, sizeof g); g.type = FUSE_CAMODC; g.q = c; g.C = C; g.A = (unsigned)(atox % C); return strict_gate(r, g); }
    if (r->fusion > 0 || r->composite) {
        QGate g; memset(&g, 0, sizeof g);
        g.q = c; g.C = C; g.A = (unsigned)(atox % C);
        g.type = camodc_closed_form(r->n, (unsigned)r->M, C, g.A, c) ? (uint32_t)FUSE_CAMODC : 99u;
        return fuse_push(r, g);
    }
    FLUSH(r);
    QCX_TRY(canon_if_dirty(r));
    return reg_camodc(r, C, (unsigned)(atox % C), c);
}

extern "C" int qcx_swap_states(qcx_register *r) { return r ? QCX_NO_ERROR : QCX_BAD_ARGUMENTS; }

// The whole-circuit entry points know their complete gate list, so unless the register is in strict per-gate mode
// (qcx_set_fusion(reg, -1)) they hand it to the pass scheduler in one piece: same bits, one HBM round trip per pass
// instead of one per gate.  Nested use (quantum_computation -> inverse_QFT) flushes once, at the outermost exit.
struct CircuitScope {
    qcx_register *r;
    bool mine;
    explicit CircuitScope(qcx_register *r_) : r(r_), mine(!r_->sh && r_->fusion == 0) { if (mine) r->composite++; }
    int done(int status)
    {
        if (!mine) return status;
        mine = false;
        if (--r->composite == 0) { const int f = fuse_flush(r, true); if (status == QCX_NO_ERROR) status = f; }     // (a compact chain's result may stay compact)
        return status;
    }
    ~CircuitScope() { (void)done(QCX_NO_ERROR); }
};

static int inverse_qft_body(qcx_register *r)
{
    for (int l = r->L + r->M - 1; l >= r->M; l--) {
        QCX_TRY(qcx_hadamard_gate((unsigned)l, r));
        for (int k = l - 1; k >= r->M; k--) {
            const double theta = M_PI / (double)((uint64_t)1 << (unsigned)(l - k));   // Q:686
            QCX_TRY(qcx_c_phase_shift_gate((unsigned)l, (unsigned)k, theta, r));
        }
    }
    return QCX_NO_ERROR;
}

extern "C" int qcx_inverse_QFT(qcx_register *r)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    CircuitScope scope(r);
    return scope.done(inverse_qft_body(r));
}

// the reference's INT_POW (Q:158-159) as x86-64 gcc evaluates it: pow, +0.5, truncation to a
// signed 64-bit integer (out of range gives 0x8000000000000000), low 32 bits kept
extern "C" unsigned qcx_ref_int_pow(double base, double power)
{
    const double d = pow(base, power) + 0.5;
    if (!(d > -9223372036854775808.0 && d < 9223372036854775808.0)) return 0u;
    return (unsigned)(unsigned long long)(long long)d;
}

static int quantum_computation_body(unsigned C, unsigned a, int intpow_mode, qcx_register *r);

extern "C" int qcx_quantum_computation(unsigned C, unsigned a, int intpow_mode, qcx_register *r)
{
    if (!r || C == 0) return QCX_BAD_ARGUMENTS;
    CircuitScope scope(r);
    return scope.done(quantum_computation_body(C, a, intpow_mode, r));
}

static int quantum_computation_body(unsigned C, unsigned a, int intpow_mode, qcx_register *r)
{
    const unsigned lo = r->n - (unsigned)r->L;
    for (unsigned l = lo; l < r->n; l++) QCX_TRY(qcx_hadamard_gate(l, r));
    unsigned x = 1;                                   // Q:714
    unsigned long long exact = a % C;                 // a^(2^k) mod C by repeated squaring
    for (unsigned l = lo; l < r->n; l++) {
        const unsigned long long atox = intpow_mode ? (unsigned long long)qcx_ref_int_pow((double)a, (double)x) : exact;
        QCX_TRY(qcx_c_amodc_gate(C, atox, l, r));
        x *= 2;                                       // Q:730
        exact = (exact * exact) % C;
    }
    return inverse_qft_body(r);
}

extern "C" int qcx_measure_state_r(qcx_register *r, double rnd, unsigned long *state_num)
{
    if (!r || !state_num) return QCX_BAD_ARGUMENTS;
    if (r->sh) return sh_measure(r->sh, rnd, state_num);
    QCX_TRY(fuse_flush(r, true));                                           // (a compact chain's result may stay compact)
    int found = 0; uint64_t idx = 0; double cum = 0.0;
    if (r->compact_pending == 2) QCX_TRY(compact_finish_last(r, false));   // (the chain's deferred last pass, as an ordinary pass: the scan wants the compact form)
    if (r->compact_pending) {
        // The scan of Q:283-292 on the compact form: the amplitudes it leaves out are +0 and add exactly nothing to the running
        // sum, and the compact order IS the index order (orbit ascending), so the first compact element with cum >= r is the
        // first real one -- except for r <= 0, where the reference stops at index 0 whatever it holds.
        const unsigned M = (unsigned)r->M, cb = r->compact_cb, nv = r->n - M + cb;
        if (rnd <= 0.0) { found = 1; idx = 0; }
        else {
            uint64_t last_excl = (uint64_t)1 << nv;                         // compact elements whose real index is below dim - 1
            if (r->compact_orbit[r->compact_ncols - 1] == (1u << M) - 1u) last_excl = ((((uint64_t)1 << (r->n - M)) - 1) << cb) | (r->compact_ncols - 1);
            uint64_t cidx = 0;
            QCX_TRY(qcx_shard_measure_scan(r->compact_amp, nv, 0, last_excl, 0.0, rnd, &found, &cidx, &cum, r->stream));
            const unsigned col = (unsigned)(cidx & ((1u << cb) - 1u));
            if (found && col >= r->compact_ncols) {
                // a hit in a PADDING column: those hold +0 through every gate and add nothing to the sum, so the scan cannot
                // stop there -- unless the premise broke.  Do not map it through stale orbit slots: expand and scan the register.
                QCX_TRY(expand_pending(r));
                found = 0;
                QCX_TRY(qcx_shard_measure_scan(r->amp, r->n, 0, r->dim - 1, 0.0, rnd, &found, &idx, &cum, r->stream));
            }
            else if (found) idx = ((cidx >> cb) << M) | r->compact_orbit[col];
        }
        r->compact_pending = 0;                                             // the collapse below replaces the whole state
        r->compact_measures++;
    } else
    QCX_TRY(qcx_shard_measure_scan(r->amp, r->n, 0, r->dim - 1, 0.0, rnd, &found, &idx, &cum, r->stream));
    if (!found) idx = r->dim - 1;                                           // Q:283 fall-through
    r->zeros_dirty = 0;                                                     // (the collapse replaces the whole state)
    r->nonfinite = 0;
    if (r->fusion >= 0) { r->basis_pending = 1; r->basis_index = idx; }     // Q:302-303, written at the next flush (or never: a reset may follow)
    else QCX_TRY(qcx_shard_collapse(r->amp, r->n, (int64_t)idx, r->stream));
    *state_num = (unsigned long)idx;
    return QCX_NO_ERROR;
}

extern "C" int qcx_measure_state(qcx_register *r, qcx_rng *rng, unsigned long *state_num)
{
    if (!rng) return QCX_BAD_ARGUMENTS;
    return qcx_measure_state_r(r, qcx_rng_uniform(rng), state_num);         // Q:281
}

extern "C" int qcx_state_read(qcx_register *r, unsigned long first, unsigned long count, double *out)
{
    if (!r || (!out && count)) return QCX_BAD_ARGUMENTS;
    if ((uint64_t)first > r->dim || (uint64_t)count > r->dim - first) return QCX_BAD_ARGUMENTS;
    if (r->sh) return sh_copy(r->sh, first, count, out, true);
    FLUSH(r);
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (count) HIP_TRY(hipMemcpy(out, r->amp + first, (size_t)count * sizeof(amp_t), hipMemcpyDeviceToHost));
    return QCX_NO_ERROR;
}

extern "C" int qcx_state_write(qcx_register *r, unsigned long first, unsigned long count, const double *in)
{
    if (!r || (!in && count)) return QCX_BAD_ARGUMENTS;
    if ((uint64_t)first > r->dim || (uint64_t)count > r->dim - first) return QCX_BAD_ARGUMENTS;
    if (r->sh) return sh_copy(r->sh, first, count, const_cast<double *>(in), false);
    FLUSH(r);
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (count) {
        HIP_TRY(hipMemcpy(r->amp + first, in, (size_t)count * sizeof(amp_t), hipMemcpyHostToDevice));
        r->zeros_dirty = 1;
        QCX_TRY(note_nonfinite(r, first, count));
    }
    return QCX_NO_ERROR;
}

// ---- state files (golden-vector I/O, debugging, checkpoint) -------------------------------
// 64-byte header, then the 2^n amplitudes as interleaved little-endian binary64 (re, im), index order.  Streamed
// through a pinned staging buffer, so a 16 GiB state needs 64 MiB of host memory.  The checksum is FNV-1a 64 over the
// payload bytes.  The payload goes straight to the device chunk by chunk, so a load that fails half-way (truncated
// file, checksum mismatch) leaves a partial copy in the register and says so (QCX_UNKNOWN_ERROR + qcx_last_error).
struct StateFileHeader {
    char     magic[8];          // "QCXSTATE"
    uint32_t version;           // 1
    uint32_t bytes_per_amp;     // 16
    int32_t  L, M;
    uint64_t dim;               // 2^(L+M)
    uint64_t checksum;          // FNV-1a 64 of the payload
    uint8_t  pad[24];
};
static_assert(sizeof(StateFileHeader) == 64, "state file header");

static uint64_t fnv1a64(const unsigned char *p, size_t len, uint64_t h)
{
    for (size_t k = 0; k < len; k++) { h ^= p[k]; h *= 0x100000001b3ull; }
    return h;
}

extern "C" int qcx_state_save(qcx_register *r, const char *path)
{
    if (!r || !path) return QCX_BAD_ARGUMENTS;
    if (r->sh) QCX_TRY(sh_identity(r->sh)); else { FLUSH(r); HIP_TRY(hipStreamSynchronize(r->stream)); }
    FILE *f = fopen(path, "wb");
    if (!f) { set_error("qcx_state_save: cannot open %s", path); return QCX_BAD_ARGUMENTS; }
    StateFileHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "QCXSTATE", 8); h.version = 1; h.bytes_per_amp = (uint32_t)sizeof(amp_t);
    h.L = r->L; h.M = r->M; h.dim = r->dim;
    int st = QCX_NO_ERROR;
    void *stage = nullptr;
    const size_t chunk = (size_t)std::min<uint64_t>(r->dim, (uint64_t)1 << 22);          // 4 Mi amplitudes = 64 MiB
    if (hipHostMalloc(&stage, chunk * sizeof(amp_t)) != hipSuccess) { fclose(f); return QCX_INSUFFICIENT_MEMORY; }
    uint64_t sum = 0xcbf29ce484222325ull;
    if (fwrite(&h, sizeof h, 1, f) != 1) st = QCX_UNKNOWN_ERROR;
    for (uint64_t at = 0; st == QCX_NO_ERROR && at < r->dim; at += chunk) {
        const size_t cnt = (size_t)std::min<uint64_t>(chunk, r->dim - at);
        if (r->sh ? sh_copy(r->sh, at, cnt, (double *)stage, true) != QCX_NO_ERROR
                  : hipMemcpy(stage, r->amp + at, cnt * sizeof(amp_t), hipMemcpyDeviceToHost) != hipSuccess) { st = QCX_HIP_ERROR; break; }
        sum = fnv1a64((const unsigned char *)stage, cnt * sizeof(amp_t), sum);
        if (fwrite(stage, sizeof(amp_t), cnt, f) != cnt) st = QCX_UNKNOWN_ERROR;
    }
    if (st == QCX_NO_ERROR) {
        h.checksum = sum;
        if (fseek(f, 0, SEEK_SET) != 0 || fwrite(&h, sizeof h, 1, f) != 1) st = QCX_UNKNOWN_ERROR;
    }
    (void)hipHostFree(stage);
    if (fclose(f) != 0 && st == QCX_NO_ERROR) st = QCX_UNKNOWN_ERROR;
    if (st != QCX_NO_ERROR) set_error("qcx_state_save: writing %s failed", path);
    return st;
}

extern "C" int qcx_state_load(qcx_register *r, const char *path)
{
    if (!r || !path) return QCX_BAD_ARGUMENTS;
    if (r->sh) QCX_TRY(sh_identity(r->sh)); else { FLUSH(r); HIP_TRY(hipStreamSynchronize(r->stream)); }
    r->zeros_dirty = 1;                              // (a sharded register: sh_copy notes it per set)
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("qcx_state_load: cannot open %s", path); return QCX_BAD_ARGUMENTS; }
    StateFileHeader h;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "QCXSTATE", 8) != 0 || h.version != 1 || h.bytes_per_amp != sizeof(amp_t)) {
        fclose(f); set_error("qcx_state_load: %s is not a state file of this version", path); return QCX_BAD_ARGUMENTS;
    }
    if (h.L != r->L || h.M != r->M || h.dim != r->dim) {
        fclose(f); set_error("qcx_state_load: %s holds L=%d M=%d, the register is L=%d M=%d", path, h.L, h.M, r->L, r->M);
        return QCX_BAD_ARGUMENTS;
    }
    void *stage = nullptr;
    const size_t chunk = (size_t)std::min<uint64_t>(r->dim, (uint64_t)1 << 22);
    if (hipHostMalloc(&stage, chunk * sizeof(amp_t)) != hipSuccess) { fclose(f); return QCX_INSUFFICIENT_MEMORY; }
    int st = QCX_NO_ERROR;
    uint64_t sum = 0xcbf29ce484222325ull;
    for (uint64_t at = 0; st == QCX_NO_ERROR && at < r->dim; at += chunk) {
        const size_t cnt = (size_t)std::min<uint64_t>(chunk, r->dim - at);
        if (fread(stage, sizeof(amp_t), cnt, f) != cnt) { st = QCX_UNKNOWN_ERROR; break; }
        sum = fnv1a64((const unsigned char *)stage, cnt * sizeof(amp_t), sum);
        if (r->sh ? sh_copy(r->sh, at, cnt, (double *)stage, false) != QCX_NO_ERROR
                  : hipMemcpy(r->amp + at, stage, cnt * sizeof(amp_t), hipMemcpyHostToDevice) != hipSuccess) st = QCX_HIP_ERROR;
    }
    (void)hipHostFree(stage);
    fclose(f);
    if (st == QCX_NO_ERROR && sum != h.checksum) st = QCX_UNKNOWN_ERROR;
    if (st != QCX_NO_ERROR) set_error("qcx_state_load: %s is truncated or corrupt (the register now holds a partial copy)", path);
    if (!r->sh) { const int nf = note_nonfinite(r, 0, r->dim); if (st == QCX_NO_ERROR) st = nf; }
    return st;
}

extern "C" int qcx_norm2(qcx_register *r, double *out)
{
    if (!r || !out) return QCX_BAD_ARGUMENTS;
    if (r->sh) return sh_norm2(r->sh, out);
    FLUSH(r);
    return qcx_shard_norm2(r->amp, r->n, out, r->stream);
}

// T:28-37 exactly: the reference's check_normalisation adds |amp|^2 one by one in index order.  The exact scan of the
// measurement (run to the end: r = +inf is never reached) returns that very sum, bit for bit; qcx_norm2 above is the
// faster tree sum.
extern "C" int qcx_total_probability(qcx_register *r, double *out)
{
    if (!r || !out) return QCX_BAD_ARGUMENTS;
    int found = 0; uint64_t idx = 0; double cum = 0.0;
    if (r->sh) {
        ShardSet *sh = r->sh;
        QCX_TRY(sh_identity(sh));
        if (sh->dry) return QCX_UNSUPPORTED;
        QCX_TRY(sh_sync(sh));
        for (unsigned s = 0; s < sh->W; s++) {
            HIP_TRY(hipSetDevice(sh->dev[s]));
            double c2 = cum;
            QCX_TRY(qcx_shard_measure_scan(sh->buf[sh->cur][s], sh->n_local, (uint64_t)s << sh->n_local, r->dim, cum, INFINITY,
                                           &found, &idx, &c2, sh->st[s]));
            cum = c2;
        }
        *out = cum;
        return QCX_NO_ERROR;
    }
    FLUSH(r);
    QCX_TRY(qcx_shard_measure_scan(r->amp, r->n, 0, r->dim, 0.0, INFINITY, &found, &idx, &cum, r->stream));
    *out = cum;
    return QCX_NO_ERROR;
}

extern "C" int qcx_state_fill_random(qcx_register *r, uint64_t seed)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (r->sh) return sh_fill_random(r->sh, seed);
    if (r->queue) r->queue->gates.clear();
    r->basis_pending = 0;                           // everything is overwritten
    r->compact_pending = 0;
    r->nonfinite = 0;
    r->zeros_dirty = 0;
    // U(-0.5, 0.5) components have variance 1/12: this scale makes the expected norm 1
    return qcx_shard_fill_random(r->amp, r->n, 0, seed, sqrt(6.0 / (double)r->dim), r->stream);
}

// a pool of HIP events on the register's stream: record between gates inside a timed region,
// read the differences afterwards (per-kernel durations for the roofline, bench.py)
extern "C" int qcx_events_create(qcx_register *r, unsigned count)
{
    if (!r || count > 65536) return QCX_BAD_ARGUMENTS;
    if (r->sh) return QCX_UNSUPPORTED;               // per-gate event pools are a single-GPU bench tool; use qcx_timer_start/stop
    for (unsigned i = 0; i < r->n_events; i++) (void)hipEventDestroy(r->events[i]);
    free(r->events);
    r->events = nullptr; r->n_events = 0;
    if (count == 0) return QCX_NO_ERROR;
    r->events = (hipEvent_t *)calloc(count, sizeof(hipEvent_t));
    if (!r->events) return QCX_INSUFFICIENT_MEMORY;
    for (unsigned i = 0; i < count; i++) { HIP_TRY(hipEventCreate(&r->events[i])); r->n_events = i + 1; }
    return QCX_NO_ERROR;
}

extern "C" int qcx_event_record(qcx_register *r, unsigned slot)
{
    if (!r || slot >= r->n_events) return QCX_BAD_ARGUMENTS;
    FLUSH(r);
    HIP_TRY(hipEventRecord(r->events[slot], r->stream));
    return QCX_NO_ERROR;
}

extern "C" int qcx_event_elapsed(qcx_register *r, unsigned from_slot, unsigned to_slot, double *ms)
{
    if (!r || !ms || from_slot >= r->n_events || to_slot >= r->n_events) return QCX_BAD_ARGUMENTS;
    HIP_TRY(hipEventSynchronize(r->events[to_slot]));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, r->events[from_slot], r->events[to_slot]));
    *ms = (double)f;
    return QCX_NO_ERROR;
}

extern "C" int qcx_timer_start(qcx_register *r)
{
    if (!r) return QCX_BAD_ARGUMENTS;
    if (r->sh) {                                     // all shards idle, then the start event on shard 0's stream
        if (r->sh->dry) return QCX_UNSUPPORTED;
        QCX_TRY(sh_sync(r->sh));
        HIP_TRY(hipSetDevice(r->sh->dev[0]));
        HIP_TRY(hipEventRecord(r->ev0, r->sh->st[0]));
        return QCX_NO_ERROR;
    }
    FLUSH(r);
    HIP_TRY(hipEventRecord(r->ev0, r->stream));
    return QCX_NO_ERROR;
}

extern "C" int qcx_timer_stop(qcx_register *r, double *ms)
{
    if (!r || !ms) return QCX_BAD_ARGUMENTS;
    if (r->sh) {                                     // everything queued has run on every shard, then the stop event
        if (r->sh->dry) return QCX_UNSUPPORTED;
        QCX_TRY(sh_sync(r->sh));
        HIP_TRY(hipSetDevice(r->sh->dev[0]));
        HIP_TRY(hipEventRecord(r->ev1, r->sh->st[0]));
        HIP_TRY(hipEventSynchronize(r->ev1));
        float f = 0.f;
        HIP_TRY(hipEventElapsedTime(&f, r->ev0, r->ev1));
        *ms = (double)f;
        return QCX_NO_ERROR;
    }
    FLUSH(r);
    HIP_TRY(hipEventRecord(r->ev1, r->stream));
    HIP_TRY(hipEventSynchronize(r->ev1));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, r->ev0, r->ev1));
    *ms = (double)f;
    return QCX_NO_ERROR;
}

// ---------------------------------------------------------------------------
// MT19937 with gsl_rng_mt19937 semantics (GSL 2.6 rng/mt.c: 2002 seeding, seed 0 -> 4357,
// uniform = u32 / 2^32).  Restated from the published generator definition.
// ---------------------------------------------------------------------------
struct qcx_rng {
    uint32_t s[624];
    int pos;
};

extern "C" qcx_rng *qcx_rng_alloc(void)
{
    qcx_rng *g = (qcx_rng *)malloc(sizeof(qcx_rng));
    if (g) qcx_rng_set(g, 0);
    return g;
}

extern "C" void qcx_rng_free(qcx_rng *g) { free(g); }

extern "C" void qcx_rng_set(qcx_rng *g, unsigned long seed)
{
    uint32_t v = (uint32_t)(seed & 0xffffffffUL);
    if (v == 0) v = 4357u;
    for (int i = 0; i < 624; i++) {
        g->s[i] = v;
        v = 1812433253u * (v ^ (v >> 30)) + (uint32_t)(i + 1);
    }
    g->pos = 624;
}

extern "C" unsigned long qcx_rng_get(qcx_rng *g)
{
    if (g->pos == 624) {
        uint32_t *s = g->s;
        auto twist = [](uint32_t u, uint32_t v) { uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu); return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u); };
        int k = 0;
        for (; k < 624 - 397; k++) s[k] = s[k + 397] ^ twist(s[k], s[k + 1]);
        for (; k < 623; k++)       s[k] = s[k + 397 - 624] ^ twist(s[k], s[k + 1]);
        s[623] = s[396] ^ twist(s[623], s[0]);
        g->pos = 0;
    }
    uint32_t y = g->s[g->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return (unsigned long)y;
}

extern "C" double qcx_rng_uniform(qcx_rng *g) { return (double)qcx_rng_get(g) / 4294967296.0; }
