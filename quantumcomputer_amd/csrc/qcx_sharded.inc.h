// qcx_sharded.inc.h -- a register sharded over the GPUs of one node by ONE host process.
// Included into qcx_api.hip after struct qcx_register and the fusion queue.
//
// north_star: "the state vector shards across the 8 GPUs of one node by the top 3 qubit indices ... host code stays
// in C".  This is the C side of that: qcx_register_create_sharded() returns an ordinary qcx_register*, so every entry
// point of include/qcx.h -- and with it the reference's own circuit builders through qcx_compat.h, Q:678-737, and its
// main, Q:1284-1347 -- drives W = 2^k shards without knowing.  (quantumcomputer_amd/sharded.py is the other host for
// the same kernels: one PROCESS per GPU with the exchange as an RCCL all-to-all, which is what bench.py --gpus N runs.)
//
// Layout: shard r holds the 2^(n-k) amplitudes whose top k PHYSICAL index bits are r; a logical -> physical qubit
// permutation lives on the host.
//   * H on a physically local qubit, every controlled phase (diagonal) and the controlled modular multiply (permutes
//     only the low M bits, which never move) touch no other shard: a control that sits in the shard id is a per-shard
//     constant -- shards where it is 0 skip the gate.
//   * H on a physically global qubit is the one exchange step: all k shard-id bits are traded for k local bits at
//     once.  Every GPU runs ONE kernel (k_pack_push) that brings the bits to give up to the top of the local index
//     AND writes each amplitude straight into the buffer of the GPU that owns it afterwards -- peer stores over xGMI,
//     all 7 links of a GPU busy together, no staging buffer, no second pass.  Which local qubits to give up is
//     decided with look-ahead over the queued gates (Belady: those whose next H lies furthest in the future).
//     The trade is issued slice by slice (the top sigma local bits are spectators that are never traded) on a second
//     stream per shard, and the gates next to it in the queue run on the slices that are not in flight: exchange and
//     compute overlap.  With fewer shards than GPUs, idle GPUs can relay stripes of every chunk (multi-path striping).
//   * measurement hands the exact running sum from shard to shard in index order (the index the unsharded scan
//     would pick); norm and read-back flush the queue; measurement and read-back restore the identity layout.
// Shards may share a device ("virtual" shards: several entries of `devices` equal) -- that is how the path is tested
// on a one-GPU box; the code is the same, peer stores become local stores.
// With devices[0] = -1 the register is a DRY RUN: no device memory, no launches; the schedule it would execute is
// recorded as text (qcx_sharded_trace) and replayed against the oracle by the CPU-only tests.

struct SGate {
    uint32_t type;            // FUSE_H: H(q); FUSE_PHASE: phase on (q, q2) with (c, s); FUSE_CAMODC: modular multiply, control q
    unsigned q, q2;           // LOGICAL qubits
    double   c, s;
    unsigned C, A;
};

struct ShardSet {
    unsigned k = 0, W = 1, n = 0, n_local = 0, M = 0;
    int      L = 0;
    bool     dry = false;
    std::vector<int>         dev;
    std::vector<hipStream_t> st;
    std::vector<amp_t *>     buf[2];
    int      cur = 0;
    std::vector<unsigned>    perm, inv;          // logical -> physical, physical -> logical
    std::vector<SGate>       queue;
    unsigned min_evict = 0, zone_lo = 0;
    // exchange/compute overlap: the top sigma local bits are SPECTATORS (never traded); a slice = one
    // value of them = 2^slice_bits consecutive amplitudes of a shard.  A trade is done slice by slice on a second stream
    // per shard, with the gates before / after it running on the slices that are not in flight.
    unsigned sigma = 0, slice_bits = 0;
    bool     overlap = true;
    unsigned long overlapped_gates = 0;
    std::vector<hipStream_t> xs;                 // per shard: the stream the pack+push kernels run on
    std::vector<hipEvent_t>  ev_pre, ev_push;    // [shard * 8 + slice]: pre-window gates done / slice pushed
    unsigned long exchanges = 0, pack_passes = 0;
    std::vector<hipEvent_t>  ev_a;               // per shard: its earlier work is done (recorded at the start of a trade)
    // multi-path striping: GPUs that hold no shard relay a share of every chunk (qcx_sharded_set_relays)
    std::vector<int>         relay_dev;
    std::vector<hipStream_t> relay_st;
    std::vector<hipEvent_t>  relay_ev;
    std::vector<amp_t *>     relay_stage;
    uint32_t nb_direct = 0, nb_relay = 0;         // blocks of 256 amplitudes of a chunk: direct stripe, each relay's stripe
    uint64_t stage_amps = 0;                      // amplitudes per staging slot (the last relay's stripe, the longest)
    unsigned long relayed_bytes = 0;
    bool     basis_pending = false;               // reset_register is lazy: the basis state |1> is written at the next flush, together with
                                                  // the closed-form circuit front if the queue starts with one (K0b, zero communication)
    unsigned long fronts = 0;
    bool     in_selfcheck = false;                // this set IS the small register of a self-check (no nested checks)
    unsigned long selfchecks = 0;                 // pre-flight checks this register has passed (qcx_sharded_selfchecks)
    bool     staged = false;                      // no peer access between some of the devices (or QCX_SHARD_FORCE_STAGED=1): a trade packs into the
                                                  // shard's OWN spare buffer and the chunks travel by hipMemcpyPeerAsync (no peer stores, no relays)
    bool     zeros_dirty = false;                 // the caller wrote amplitudes: one k_canon_zeros pass per shard before the next gate (Q:393-413)
    ShardSet *comp = nullptr;                     // the COMPANION register of compact circuits (sh_compact): L + cb qubits on the same devices, created on first use
    unsigned long compact_circuits = 0, compact_measures = 0;
    // round 5: the result of a compact circuit STAYS on the companion until something other than measure_state looks at the state
    // (sh_flush expands it then); measure_state scans the companion's shards -- the left-out amplitudes are +0 and add nothing to the
    // reference's running sum, the compact order is the index order -- so an attempt never writes the 16 * 2^n bytes at all
    bool comp_pending = false;
    ExpandParams comp_E;
    int      fusion = 1;                          // 1: each shard's gate list goes through the fused-pass scheduler; -1/0: one launch per gate
    size_t   max_queue = 8192;
    std::string trace;
};

#define SH_DEV(sh, r) do { if (!(sh)->dry) HIP_TRY(hipSetDevice((sh)->dev[r])); } while (0)

// spectator bits: QCX_SHARD_SLICES_LOG2 (default 3 -> 8 slices)
static unsigned sh_default_sigma()
{
    unsigned sigma = 3;
    if (const char *e = getenv("QCX_SHARD_SLICES_LOG2")) sigma = (unsigned)std::min(3, std::max(0, atoi(e)));
    return sigma;
}

// fix the slice geometry (fewer spectator bits than asked when the register is too small for them).  The trade zone
// moves with it, so this is only legal in the identity layout with nothing queued.
static void sh_set_slices(ShardSet *sh, unsigned sigma)
{
    const unsigned k = sh->k;
    for (;; sigma--) {
        sh->min_evict = std::max<unsigned>((sh->n_local >= sigma + 2 * k + 6) ? 6u : 0u, sh->M);   // never trade away the M register or runs < 1 KiB
        if (sigma == 0 || sh->n_local >= sigma + sh->min_evict + 2 * k) break;
    }
    sh->sigma = sigma;
    sh->slice_bits = sh->n_local - sigma;
    sh->zone_lo = sh->slice_bits - k;               // the trade zone: the top k bits of a slice
}

static void sh_identity_perm(ShardSet *sh)
{
    for (unsigned q = 0; q < sh->n; q++) { sh->perm[q] = q; sh->inv[q] = q; }
}

static void sh_drop_relays(ShardSet *sh)
{
    for (size_t i = 0; i < sh->relay_dev.size(); i++) {
        (void)hipSetDevice(sh->relay_dev[i]);
        if (i < sh->relay_st.size() && sh->relay_st[i]) { (void)hipStreamSynchronize(sh->relay_st[i]); (void)hipStreamDestroy(sh->relay_st[i]); }
        if (i < sh->relay_ev.size() && sh->relay_ev[i]) (void)hipEventDestroy(sh->relay_ev[i]);
        if (i < sh->relay_stage.size() && sh->relay_stage[i]) (void)hipFree(sh->relay_stage[i]);
    }
    sh->relay_dev.clear(); sh->relay_st.clear(); sh->relay_ev.clear(); sh->relay_stage.clear();
    sh->nb_direct = sh->nb_relay = 0; sh->stage_amps = 0;
}

static void sh_free(ShardSet *sh)
{
    if (!sh) return;
    if (sh->comp) { sh_free(sh->comp); sh->comp = nullptr; }
    if (!sh->dry) {
        sh_drop_relays(sh);
        for (unsigned r = 0; r < sh->W; r++) {
            if (r < sh->dev.size()) (void)hipSetDevice(sh->dev[r]);
            if (r < sh->st.size() && sh->st[r]) { (void)hipStreamSynchronize(sh->st[r]); (void)qcx_shard_release_stream(sh->st[r]); }
            for (int b = 0; b < 2; b++) if (r < sh->buf[b].size() && sh->buf[b][r]) (void)hipFree(sh->buf[b][r]);
            if (r < sh->ev_a.size() && sh->ev_a[r]) (void)hipEventDestroy(sh->ev_a[r]);
            for (unsigned e = 8 * r; e < 8 * r + 8; e++) {
                if (e < sh->ev_pre.size() && sh->ev_pre[e]) (void)hipEventDestroy(sh->ev_pre[e]);
                if (e < sh->ev_push.size() && sh->ev_push[e]) (void)hipEventDestroy(sh->ev_push[e]);
            }
            if (r < sh->xs.size() && sh->xs[r]) { (void)hipStreamSynchronize(sh->xs[r]); (void)hipStreamDestroy(sh->xs[r]); }
            if (r < sh->st.size() && sh->st[r]) (void)hipStreamDestroy(sh->st[r]);
        }
    }
    delete sh;
}

// Default placement of W shards on the visible devices: over the largest power-of-two number of devices that is at most
// min(W, visible), neighbouring shards together (W = 8 on 8 GPUs: shard r on device r; on 4: two shards per GPU; on one
// GPU: everything on device 0 -- the "virtual shards" of the one-GPU tests).  Pure arithmetic: qcx_spread_devices
// exposes it so that hosts and tests place shards the same way the library does.
static void sh_spread(unsigned nshards, int ndev, int *out)
{
    unsigned m = 1;
    while (2 * m <= nshards && (int)(2 * m) <= ndev) m *= 2;
    for (unsigned r = 0; r < nshards; r++) out[r] = (int)(r / (nshards / m));
}

static int sh_selfcheck(const ShardSet *big, unsigned nrelays, const int *relay_devices);

static int sh_create(int L, int M, unsigned nshards, const int *devices, ShardSet **out, bool selfcheck = true)
{
    *out = nullptr;
    if (nshards < 2 || nshards > 16 || (nshards & (nshards - 1))) { set_error("sharded register: 2, 4, 8 or 16 shards"); return QCX_BAD_ARGUMENTS; }
    ShardSet *sh = new ShardSet();
    sh->W = nshards;
    while ((1u << sh->k) < nshards) sh->k++;
    sh->L = L; sh->M = (unsigned)M; sh->n = (unsigned)(L + M);
    if (sh->n <= sh->k) { delete sh; return QCX_BAD_ARGUMENTS; }
    sh->n_local = sh->n - sh->k;
    const unsigned k = sh->k;
    sh_set_slices(sh, sh_default_sigma());
    if (sh->n_local < sh->min_evict + 2 * k) {
        set_error("register too small for %u shards (need n_local - max(M, 6) >= 2 log2(shards))", nshards);
        delete sh; return QCX_BAD_ARGUMENTS;
    }
    sh->perm.resize(sh->n); sh->inv.resize(sh->n);
    sh_identity_perm(sh);
    sh->dry = devices && devices[0] < 0;
    if (sh->dry) {
        if (const char *e = getenv("QCX_SHARD_OVERLAP")) sh->overlap = atoi(e) != 0;
        if (!sh->overlap) sh_set_slices(sh, 0);
        *out = sh; return QCX_NO_ERROR;
    }
    int ndev = 0;
    { const int s = qcx_device_count(&ndev); if (s != QCX_NO_ERROR) { delete sh; return s; } }
    sh->dev.resize(nshards);
    if (!devices) {                                  // no placement given: spread over the visible devices
        if (ndev < 1) { set_error("sharded register: no HIP device"); delete sh; return QCX_HIP_ERROR; }
        sh_spread(nshards, ndev, sh->dev.data());
    }
    for (unsigned r = 0; r < nshards; r++) {
        if (devices) sh->dev[r] = devices[r];
        if (sh->dev[r] < 0 || sh->dev[r] >= ndev) {
            set_error("sharded register: shard %u wants device %d, %d visible", r, sh->dev[r], ndev);
            delete sh; return QCX_HIP_ERROR;
        }
    }
    // exchange windows pay when a trade crosses xGMI (tens of ms at n_local = 30); with every shard on ONE device a trade
    // is a local pass as fast as a gate and the windows only fragment the gate lists (measured at n = 30, 8 shards on one
    // GPU: fused sweep 31.8 -> 39.6 ms, Shor circuit 86 -> 97 ms).  QCX_SHARD_OVERLAP=0|1 overrides.
    if (const char *e = getenv("QCX_SHARD_FORCE_STAGED")) sh->staged = atoi(e) != 0;
    sh->overlap = false;
    for (unsigned r = 1; r < nshards; r++) sh->overlap |= sh->dev[r] != sh->dev[0];
    if (const char *e = getenv("QCX_SHARD_OVERLAP")) sh->overlap = atoi(e) != 0;
    if (!sh->overlap) sh_set_slices(sh, 0);            // slices only serve the windows
    sh->st.assign(nshards, nullptr); sh->ev_a.assign(nshards, nullptr);
    sh->xs.assign(nshards, nullptr); sh->ev_pre.assign(8 * nshards, nullptr); sh->ev_push.assign(8 * nshards, nullptr);
    sh->buf[0].assign(nshards, nullptr); sh->buf[1].assign(nshards, nullptr);
    int prev = 0;
    (void)hipGetDevice(&prev);
    const size_t bytes = (size_t)16 << sh->n_local;
    int status = QCX_NO_ERROR;
    for (unsigned r = 0; r < nshards && status == QCX_NO_ERROR; r++) {
        hipError_t e = hipSetDevice(sh->dev[r]);
        for (unsigned c = 0; c < nshards && e == hipSuccess; c++) {         // peer stores need the mapping both ways
            if (sh->dev[c] == sh->dev[r]) continue;
            int can = 0;
            (void)hipDeviceCanAccessPeer(&can, sh->dev[r], sh->dev[c]);
            if (!can) { sh->staged = true; continue; }               // no peer stores between these two: the staged exchange (sh_exchange)
            const hipError_t pe = hipDeviceEnablePeerAccess(sh->dev[c], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) sh->staged = true;
            (void)hipGetLastError();
        }
        if (status != QCX_NO_ERROR) break;
        if (e == hipSuccess) e = hipStreamCreate(&sh->st[r]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_a[r], hipEventDisableTiming);
        if (e == hipSuccess) e = hipStreamCreate(&sh->xs[r]);
        for (unsigned q = 8 * r; q < 8 * r + 8 && e == hipSuccess; q++) {
            e = hipEventCreateWithFlags(&sh->ev_pre[q], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->ev_push[q], hipEventDisableTiming);
        }
        for (int b = 0; b < 2 && e == hipSuccess; b++) {
            e = hipMalloc(&sh->buf[b][r], bytes);
            if (e == hipSuccess) e = hipMemsetAsync(sh->buf[b][r], 0, bytes, sh->st[r]);
        }
        if (e != hipSuccess) {
            set_error("sharded register, shard %u on device %d: %s", r, sh->dev[r], hipGetErrorString(e));
            status = (e == hipErrorOutOfMemory) ? QCX_INSUFFICIENT_MEMORY : QCX_HIP_ERROR;
        }
    }
    (void)hipSetDevice(prev);
    if (status != QCX_NO_ERROR) { sh_free(sh); return status; }
    // Pre-flight check of the exchange machinery on THESE devices (peer stores into hipMalloc'ed memory of another GPU,
    // cross-device stream/event ordering): a wrong trade would otherwise still produce plausible-looking numbers.  Runs
    // when the shards sit on more than one device (QCX_SHARD_SELFCHECK=1 forces it, =0 skips it); creation fails on a
    // mismatch.
    if (selfcheck) {
        bool distinct = false;
        for (unsigned r = 1; r < nshards; r++) distinct |= sh->dev[r] != sh->dev[0];
        bool run = distinct;
        if (const char *e = getenv("QCX_SHARD_SELFCHECK")) run = atoi(e) != 0;
        if (run) {
            const int cs = sh_selfcheck(sh, 0, nullptr);
            (void)hipSetDevice(prev);
            if (cs != QCX_NO_ERROR) { sh_free(sh); return cs; }
        }
    }
    *out = sh;
    return QCX_NO_ERROR;
}

// ---- one gate list on every shard ------------------------------------------------------------------------------------
// A queued gate under a given layout, in PHYSICAL index bits (the layout may change before the gate runs: the gates
// of an exchange's pre-window are resolved before the trade is booked).
struct POp { uint32_t type; unsigned p1, p2; double c, s; unsigned C, A; };

static POp sh_phys(const ShardSet *sh, const SGate &g)
{
    POp o; memset(&o, 0, sizeof o);
    o.type = g.type; o.p1 = sh->perm[g.q]; o.p2 = (g.type == FUSE_PHASE) ? sh->perm[g.q2] : 0;
    o.c = g.c; o.s = g.s; o.C = g.C; o.A = g.A;
    return o;
}

// the operation on a VIEW of 2^nbits consecutive amplitudes of shard r (the whole shard, or slice sidx of it): index
// bits at or above nbits are constants of the view -- spectator bits come from the slice number, the rest from the
// shard id.  false = the gate is the identity on this view
static bool sh_desc(const ShardSet *sh, const POp &o, unsigned r, unsigned nbits, unsigned sidx, qcx_gate_desc *d)
{
    auto outside = [&](unsigned pq) -> unsigned {
        return pq >= sh->n_local ? (r >> (pq - sh->n_local)) & 1u : (sidx >> (pq - nbits)) & 1u;
    };
    memset(d, 0, sizeof *d);
    if (o.type == FUSE_H) { d->type = 0; d->q = o.p1; return true; }              // (the scheduler keeps p1 < nbits)
    if (o.type == FUSE_PHASE) {
        d->type = 1; d->c = o.c; d->s = o.s;
        for (unsigned pq : {o.p1, o.p2}) {
            if (pq >= nbits) { if (!outside(pq)) return false; }
            else d->mask |= (uint64_t)1 << pq;
        }
        return true;
    }
    d->type = 2; d->C = o.C; d->A = o.A;
    if (o.p1 >= nbits) { if (!outside(o.p1)) return false; d->q = 0xffffffffu; }
    else d->q = o.p1;
    return true;
}

static void sh_trace_ops(ShardSet *sh, const std::vector<POp> &ops)
{
    char line[160];
    snprintf(line, sizeof line, "ops %zu\n", ops.size()); sh->trace += line;
    for (const POp &o : ops) {
        if (o.type == FUSE_H) snprintf(line, sizeof line, "h %u\n", o.p1);
        else if (o.type == FUSE_PHASE) snprintf(line, sizeof line, "p %u %u %a %a\n", o.p1, o.p2, o.c, o.s);
        else snprintf(line, sizeof line, "c %u %u %u\n", o.C, o.A, o.p1);
        sh->trace += line;
    }
}

// the operations on one view of shard r, on that shard's compute stream
static int sh_run_view(ShardSet *sh, const std::vector<POp> &ops, unsigned r, amp_t *a, unsigned nbits, unsigned sidx)
{
    std::vector<qcx_gate_desc> descs;
    for (const POp &o : ops) { qcx_gate_desc d; if (sh_desc(sh, o, r, nbits, sidx, &d)) descs.push_back(d); }
    if (descs.empty()) return QCX_NO_ERROR;
    if (sh->fusion > 0 && descs.size() > 1)
        return qcx_shard_run_fused_mode(sh->fusion, a, nbits, sh->M, (unsigned)descs.size(), descs.data(), sh->st[r]);
    for (const qcx_gate_desc &d : descs) {
        if (d.type == 0) QCX_TRY(qcx_shard_hadamard(a, nbits, d.q, sh->st[r]));
        else if (d.type == 1) QCX_TRY(qcx_shard_phase(a, nbits, d.mask, d.c, d.s, sh->st[r]));
        else QCX_TRY(qcx_shard_camodc(a, nbits, sh->M, d.C, d.A, d.q == 0xffffffffu ? -1 : (int)d.q, sh->st[r]));
    }
    return QCX_NO_ERROR;
}

static std::vector<POp> sh_phys_list(const ShardSet *sh, const SGate *g, size_t cnt)
{
    std::vector<POp> ops;
    for (size_t i = 0; i < cnt; i++) ops.push_back(sh_phys(sh, g[i]));
    return ops;
}

// a gate list on every whole shard
static int sh_run_ops(ShardSet *sh, const SGate *g, size_t cnt)
{
    if (!cnt) return QCX_NO_ERROR;
    const std::vector<POp> ops = sh_phys_list(sh, g, cnt);
    if (sh->dry) { sh_trace_ops(sh, ops); return QCX_NO_ERROR; }
    for (unsigned r = 0; r < sh->W; r++) {
        SH_DEV(sh, r);
        QCX_TRY(sh_run_view(sh, ops, r, sh->buf[sh->cur][r], sh->n_local, 0));
    }
    return QCX_NO_ERROR;
}

// ---- layout changes --------------------------------------------------------------------------------------------------
typedef std::vector<std::pair<unsigned, unsigned>> SwapList;

// transpositions (application order) that bring local position give[j] to trade-zone slot j
static SwapList sh_plan_give(const ShardSet *sh, std::vector<unsigned> pos)
{
    SwapList swaps;
    for (unsigned j = 0; j < sh->k; j++) {
        const unsigned t = sh->zone_lo + j, p = pos[j];
        if (p != t) {
            swaps.push_back({p, t});
            for (unsigned jj = j + 1; jj < sh->k; jj++) if (pos[jj] == t) pos[jj] = p;
        }
    }
    return swaps;
}

static void sh_book_vec(std::vector<unsigned> &perm, std::vector<unsigned> &inv, const SwapList &swaps, bool trade,
                        unsigned k, unsigned zone_lo, unsigned n_local)
{
    auto set_phys = [&](unsigned logical, unsigned pos) { perm[logical] = pos; inv[pos] = logical; };
    for (const auto &ab : swaps) {
        const unsigned la = inv[ab.first], lb = inv[ab.second];
        set_phys(la, ab.second); set_phys(lb, ab.first);
    }
    if (trade)
        for (unsigned j = 0; j < k; j++) {
            const unsigned lt = inv[zone_lo + j], lr = inv[n_local + j];
            set_phys(lt, n_local + j); set_phys(lr, zone_lo + j);
        }
}

static void sh_book(ShardSet *sh, const SwapList &swaps, bool trade)
{
    sh_book_vec(sh->perm, sh->inv, swaps, trade, sh->k, sh->zone_lo, sh->n_local);
}

static void sh_swapbits_arg(const SwapList &swaps, size_t lo, size_t hi, SwapBits *S)
{
    // the kernels compute the SOURCE index of destination j by applying their list in array order; data moved by
    // s_1, ..., s_m in that order has source s_1(s_2(...s_m(j))): the list goes in reversed
    memset(S, 0, sizeof *S);
    S->npairs = (unsigned)(hi - lo);
    for (size_t m = 0; m < hi - lo; m++) { S->a[m] = swaps[hi - 1 - m].first; S->b[m] = swaps[hi - 1 - m].second; }
}

static void sh_trace_swaps(ShardSet *sh, const char *what, const SwapList &swaps, size_t lo, size_t hi)
{
    sh->trace += what;
    char t[32];
    for (size_t m = lo; m < hi; m++) { snprintf(t, sizeof t, " %u:%u", swaps[m].first, swaps[m].second); sh->trace += t; }
    sh->trace += "\n";
}

// pack (the transpositions) + trade of all k shard-id bits with the trade zone: one k_pack_push per shard and SLICE, on
// the shard's exchange stream, with `pre` (resolved under the layout before the trade) run on every slice before it
// leaves and `post` (resolved under the layout after it) on every slice after it has arrived -- both on the compute
// stream, so the transfer of one slice hides behind the gates of the others.  Slices are independent: every gate of a
// window acts inside a slice (the scheduler only admits such gates), the trade maps slice s of every shard onto slice
// s of every shard.  Relay striping (set_relays) moves whole shards: no slices then.
static int sh_exchange(ShardSet *sh, const SwapList &swaps, const std::vector<POp> &pre, const std::vector<POp> &post)
{
    if (sh->dry) {
        if (!pre.empty()) sh_trace_ops(sh, pre);
        sh_trace_swaps(sh, "pack", swaps, 0, swaps.size());
        { char t[32]; snprintf(t, sizeof t, "trade %u\n", sh->zone_lo); sh->trace += t; }      // the trade zone: zone_lo .. zone_lo + k
        if (!post.empty()) sh_trace_ops(sh, post);
    } else {
        const unsigned W = sh->W;
        const bool staged = sh->staged;
        const unsigned R = (sh->zone_lo >= 8 && !staged) ? (unsigned)sh->relay_dev.size() : 0u;
        const unsigned S = 1u << sh->sigma;                 // (relays are only set up with sigma = 0: sh_set_relays)
        const unsigned vbits = sh->slice_bits;              // bits of one view (a slice, or the whole shard when sigma = 0)
        const unsigned zlo = sh->zone_lo;                   // its trade zone = its top k bits
        // the buffer a slice arrives in: the OTHER buffer of its new owner (peer stores), or -- staged -- the current one again:
        // every shard packs into its own spare buffer first, then the chunks are copied over the data that has been packed
        const int land = staged ? sh->cur : (sh->cur ^ 1);
        // (1) every shard's earlier work is done before anyone writes into its spare buffer
        for (unsigned r = 0; r < W; r++) { SH_DEV(sh, r); HIP_TRY(hipEventRecord(sh->ev_a[r], sh->st[r])); }
        for (unsigned r = 0; r < W; r++) {
            SH_DEV(sh, r);
            for (unsigned c = 0; c < W; c++) HIP_TRY(hipStreamWaitEvent(sh->xs[r], sh->ev_a[c], 0));
        }
        SwapBits Sb;
        sh_swapbits_arg(swaps, 0, swaps.size(), &Sb);
        PushRelay Rl;
        memset(&Rl, 0, sizeof Rl);
        if (R) {
            Rl.nrelays = R; Rl.nb_direct = sh->nb_direct; Rl.nb_relay = sh->nb_relay; Rl.stage_amps = sh->stage_amps;
            for (unsigned i = 0; i < R; i++) Rl.stage[i] = sh->relay_stage[i];
        }
        const uint64_t count = (uint64_t)1 << vbits;
        auto post_on = [&](unsigned sidx) -> int {          // slice sidx has arrived everywhere: run the post-window on it
            for (unsigned r = 0; r < W; r++) {
                SH_DEV(sh, r);
                for (unsigned c = 0; c < W; c++) HIP_TRY(hipStreamWaitEvent(sh->st[r], sh->ev_push[8 * c + sidx], 0));
                for (unsigned i = 0; i < R; i++) HIP_TRY(hipStreamWaitEvent(sh->st[r], sh->relay_ev[i], 0));
                if (!post.empty()) QCX_TRY(sh_run_view(sh, post, r, sh->buf[land][r] + ((uint64_t)sidx << vbits), vbits, sidx));
            }
            return QCX_NO_ERROR;
        };
        for (unsigned sidx = 0; sidx < S; sidx++) {
            PushDst D;
            memset(&D, 0, sizeof D);
            for (unsigned c = 0; c < W; c++) D.dst[c] = sh->buf[sh->cur ^ 1][c] + ((uint64_t)sidx << vbits);
            for (unsigned r = 0; r < W; r++) {
                SH_DEV(sh, r);
                amp_t *src = sh->buf[sh->cur][r] + ((uint64_t)sidx << vbits);
                if (!pre.empty()) QCX_TRY(sh_run_view(sh, pre, r, src, vbits, sidx));
                HIP_TRY(hipEventRecord(sh->ev_pre[8 * r + sidx], sh->st[r]));
                HIP_TRY(hipStreamWaitEvent(sh->xs[r], sh->ev_pre[8 * r + sidx], 0));
                PushDst Dr = D;
                if (staged) {
                    // pack only: chunk c of this shard's slice lands in its OWN spare buffer at [c << zlo, (c + 1) << zlo)
                    // (k_pack_push stores chunk c at dst[c] + (me << zlo) + low)
                    amp_t *spare = sh->buf[sh->cur ^ 1][r] + ((uint64_t)sidx << vbits);
                    for (unsigned c = 0; c < W; c++) Dr.dst[c] = spare + ((uint64_t)c << zlo) - ((uint64_t)r << zlo);
                }
                if (zlo >= 8)
                    hipLaunchKernelGGL((k_pack_push<256, true>), dim3(grid_for(count, 256, 0, 256)), dim3(256), 0, sh->xs[r],
                                       (const amp_t *)src, Dr, Rl, count, Sb, zlo, sh->k, r);
                else
                    hipLaunchKernelGGL((k_pack_push<64, false>), dim3(grid_for(count, 64, 0, 64)), dim3(64), 0, sh->xs[r],
                                       (const amp_t *)src, Dr, Rl, count, Sb, zlo, sh->k, r);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(sh->ev_push[8 * r + sidx], sh->xs[r]));
            }
            if (staged) {
                // (1a) every shard has packed this slice (its current buffer is free to be overwritten): the chunks travel as
                // copies, shard r's chunk c -> slot r of shard c's slice; the arrival event replaces the pack event
                for (unsigned r = 0; r < W; r++) {
                    SH_DEV(sh, r);
                    for (unsigned c = 0; c < W; c++) HIP_TRY(hipStreamWaitEvent(sh->xs[r], sh->ev_push[8 * c + sidx], 0));
                }
                for (unsigned r = 0; r < W; r++) {
                    SH_DEV(sh, r);
                    const amp_t *spare = sh->buf[sh->cur ^ 1][r] + ((uint64_t)sidx << vbits);
                    for (unsigned c = 0; c < W; c++) {
                        amp_t *to = sh->buf[sh->cur][c] + ((uint64_t)sidx << vbits) + ((uint64_t)r << zlo);
                        const size_t bytes = ((size_t)1 << zlo) * sizeof(amp_t);
                        if (sh->dev[c] == sh->dev[r]) HIP_TRY(hipMemcpyAsync(to, spare + ((uint64_t)c << zlo), bytes, hipMemcpyDeviceToDevice, sh->xs[r]));
                        else HIP_TRY(hipMemcpyPeerAsync(to, sh->dev[c], spare + ((uint64_t)c << zlo), sh->dev[r], bytes, sh->xs[r]));
                    }
                }
                for (unsigned r = 0; r < W; r++) { SH_DEV(sh, r); HIP_TRY(hipEventRecord(sh->ev_push[8 * r + sidx], sh->xs[r])); }
            }
            // (1b) the relays forward their staged stripes to the owners once every shard has pushed (unsliced only)
            if (R) {
                const uint64_t per_chunk_blocks = ((uint64_t)1 << zlo) / 256;
                for (unsigned i = 0; i < R; i++) {
                    HIP_TRY(hipSetDevice(sh->relay_dev[i]));
                    for (unsigned r = 0; r < W; r++) HIP_TRY(hipStreamWaitEvent(sh->relay_st[i], sh->ev_push[8 * r], 0));
                    const uint64_t b0 = (uint64_t)sh->nb_direct + (uint64_t)i * sh->nb_relay;
                    const uint64_t b1 = (i + 1 == R) ? per_chunk_blocks : b0 + sh->nb_relay;
                    const size_t bytes = (size_t)(b1 - b0) * 256 * sizeof(amp_t);
                    for (unsigned r = 0; r < W && bytes; r++)
                        for (unsigned c = 0; c < W; c++) {
                            if (c == r) continue;
                            amp_t *to = sh->buf[sh->cur ^ 1][c] + (((uint64_t)r << zlo) | (b0 * 256));
                            const amp_t *from = sh->relay_stage[i] + (uint64_t)(r * W + c) * sh->stage_amps;
                            HIP_TRY(hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, sh->relay_st[i]));
                            sh->relayed_bytes += bytes;
                        }
                    HIP_TRY(hipEventRecord(sh->relay_ev[i], sh->relay_st[i]));
                }
            }
            if (sidx >= 1) QCX_TRY(post_on(sidx - 1));       // ... while slice sidx is in flight
        }
        QCX_TRY(post_on(S - 1));
        if (!staged) sh->cur ^= 1;
    }
    sh_book(sh, swaps, true);
    sh->exchanges++;
    if (!swaps.empty()) sh->pack_passes++;
    sh->overlapped_gates += pre.size() + post.size();
    return QCX_NO_ERROR;
}

static int sh_exchange(ShardSet *sh, const SwapList &swaps)
{
    static const std::vector<POp> none;
    return sh_exchange(sh, swaps, none, none);
}

// transpositions among local positions on every shard (8 per out-of-place pass)
static int sh_local_permute(ShardSet *sh, const SwapList &swaps)
{
    for (size_t lo = 0; lo < swaps.size(); lo += 8) {
        const size_t hi = std::min(swaps.size(), lo + 8);
        if (sh->dry) sh_trace_swaps(sh, "permute", swaps, lo, hi);
        else {
            SwapBits S;
            sh_swapbits_arg(swaps, lo, hi, &S);
            for (unsigned r = 0; r < sh->W; r++) {
                SH_DEV(sh, r);
                QCX_TRY(qcx_shard_swap_bits(sh->buf[sh->cur][r], sh->buf[sh->cur ^ 1][r], sh->n_local, S.npairs, S.a, S.b, sh->st[r]));
            }
            sh->cur ^= 1;
        }
        sh_book(sh, SwapList(swaps.begin() + lo, swaps.begin() + hi), false);
        sh->pack_passes++;
    }
    return QCX_NO_ERROR;
}

static size_t sh_next_use(const std::vector<SGate> &q, unsigned logical, size_t start)
{
    for (size_t i = start; i < q.size(); i++) if (q[i].type == FUSE_H && q[i].q == logical) return i;
    return q.size() + 1;
}

// the k local positions whose qubits are H targets latest (Belady)
static std::vector<unsigned> sh_choose_give(const ShardSet *sh, const std::vector<SGate> &q, size_t at)
{
    std::vector<std::pair<size_t, unsigned>> cand;                 // (next use, position)
    for (unsigned p = sh->min_evict; p < sh->slice_bits; p++) cand.push_back({sh_next_use(q, sh->inv[p], at), p});     // (spectator bits are never traded)
    std::sort(cand.begin(), cand.end(), [](const std::pair<size_t, unsigned> &x, const std::pair<size_t, unsigned> &y) {
        return x.first != y.first ? x.first > y.first : x.second > y.second; });
    std::vector<unsigned> give;
    for (unsigned j = 0; j < sh->k; j++) give.push_back(cand[j].second);
    std::sort(give.begin(), give.end());
    return give;
}

// may this queued gate run slice by slice under the current layout?  (an H must not target a spectator bit, nor a
// qubit of the shard id)
static bool sh_sliceable(const ShardSet *sh, const SGate &g) { return g.type != FUSE_H || sh->perm[g.q] < sh->slice_bits; }

// a lazily pending reset: every shard writes its part of the basis state |1> -- fused with the circuit front (Hadamards on
// distinct qubits, then controlled modular multiplies, Q:720-731) when the queue starts with one.  The front's closed form
// only looks at GLOBAL index bits, so a Hadamard or a control on a shard-id qubit costs nothing here: no exchange.
static int sh_materialize_basis(ShardSet *sh)
{
    std::vector<QGate> qg;                          // (the layout is the identity: sh_reset set it)
    for (const SGate &g : sh->queue) {
        QGate q; memset(&q, 0, sizeof q);
        if (g.type == FUSE_H) { q.type = FUSE_H; q.q = g.q; }
        else if (g.type == FUSE_CAMODC && camodc_closed_form(sh->n, sh->M, g.C, g.A % g.C, g.q)) { q.type = FUSE_CAMODC; q.q = g.q; q.C = g.C; q.A = g.A; }
        else break;
        qg.push_back(q);
    }
    BasisFront B;
    size_t used = 0;
    used = front_plan(sh->n, sh->M, 1, tune_now(), qg, &B);
    for (unsigned r = 0; r < sh->W; r++) {
        SH_DEV(sh, r);
        if (!used) { QCX_TRY(qcx_shard_reset(sh->buf[sh->cur][r], sh->n_local, r == 0, sh->st[r])); continue; }
        B.first = (uint64_t)r << sh->n_local;
        QCX_TRY(launch_basis_front(sh->buf[sh->cur][r], sh->n_local, B, sh->st[r]));
    }
    sh->basis_pending = false;                      // only now: a failed launch above leaves the reset pending (and the queue whole)
    if (used) { sh->queue.erase(sh->queue.begin(), sh->queue.begin() + used); sh->fronts++; }
    return QCX_NO_ERROR;
}

static int sh_flush(ShardSet *sh, bool keep_compact = false);
static int sh_identity(ShardSet *sh);
static int sh_set_relays(ShardSet *sh, unsigned nrelays, const int *devices);

// Compact circuits on a sharded register (the sharded form of compact_chain, qcx_fuse.inc.h).  Behind the circuit front the M
// register reads one of the residues of the multiply ladder's orbit, and when nothing else in the queue touches it the whole
// queue runs on a COMPANION register of L + cb qubits -- [L register][orbit column], the same shards on the same devices,
// 2^(M - cb) times smaller -- with the unchanged machinery (fused passes per shard, trades of the shard-id qubits: 2^(M - cb)
// times fewer bytes over the links); every shard then expands its part into the real register.  The front is written in the
// compact form directly (k_basis_front_compact).  *done = false: not applicable, nothing happened.
static int sh_compact(ShardSet *sh, bool *done)
{
    *done = false;
    const Tune tn = tune_now();
    const unsigned M = sh->M, L = (unsigned)sh->L;
    if (sh->dry || !tn.fuse_compact || !tn.fuse_front || sh->fusion <= 0 || sh->in_selfcheck || M < 4 || M > 12 || sh->n_local < M + 6) return QCX_NO_ERROR;
    std::vector<QGate> qg;                          // (the layout is the identity: sh_reset set it)
    for (const SGate &g : sh->queue) {
        QGate q; memset(&q, 0, sizeof q);
        if (g.type == FUSE_H) { q.type = FUSE_H; q.q = g.q; }
        else if (g.type == FUSE_CAMODC && camodc_closed_form(sh->n, sh->M, g.C, g.A % g.C, g.q)) { q.type = FUSE_CAMODC; q.q = g.q; q.C = g.C; q.A = g.A; }
        else break;
        qg.push_back(q);
    }
    BasisFront B;
    const size_t used = front_plan(sh->n, M, 1, tn, qg, &B);
    if (!used || used >= sh->queue.size()) return QCX_NO_ERROR;
    const uint32_t lowmask = (1u << M) - 1u;
    if ((B.hmask & lowmask) != 0 || B.ncam > 64) return QCX_NO_ERROR;
    for (size_t i = used; i < sh->queue.size(); i++) {
        const SGate &g = sh->queue[i];
        if (g.type == FUSE_H) { if (g.q < M) return QCX_NO_ERROR; }
        else if (g.type == FUSE_PHASE) { if (g.q < M || g.q2 < M) return QCX_NO_ERROR; }
        else return QCX_NO_ERROR;
    }
    std::vector<uint16_t> orbit;
    unsigned cb = 0;
    if (!compact_orbit(B, M, orbit, &cb)) return QCX_NO_ERROR;
    // the companion register
    if (sh->comp && (sh->comp->M != cb || sh->comp->L != sh->L)) { sh_free(sh->comp); sh->comp = nullptr; }
    if (!sh->comp) {
        ShardSet *c = nullptr;
        if (sh_create((int)L, (int)cb, sh->W, sh->dev.data(), &c, false) != QCX_NO_ERROR) { g_last_error[0] = 0; return QCX_NO_ERROR; }   // (too small for that many shards, no memory ...)
        sh->comp = c;
        sh->comp->in_selfcheck = true;                 // (never a compact circuit or a pre-flight check of its own)
        if (!sh->relay_dev.empty() && sh_set_relays(sh->comp, (unsigned)sh->relay_dev.size(), sh->relay_dev.data()) != QCX_NO_ERROR) { g_last_error[0] = 0; sh_free(sh->comp); sh->comp = nullptr; return QCX_NO_ERROR; }
    }
    ShardSet *c = sh->comp;
    c->fusion = sh->fusion;
    c->queue.clear(); c->basis_pending = false; c->zeros_dirty = false;
    sh_identity_perm(c);
    ExpandParams E;
    memset(&E, 0, sizeof E);
    E.M = M; E.cb = cb; E.ncols = (unsigned)orbit.size();
    for (size_t j = 0; j < orbit.size(); j++) E.orbit[j] = orbit[j];
    const unsigned long ex0 = c->exchanges, pp0 = c->pack_passes, rb0 = c->relayed_bytes, og0 = c->overlapped_gates;
    for (unsigned r = 0; r < sh->W; r++) {
        SH_DEV(c, r);
        B.first = (uint64_t)r << sh->n_local;
        const uint64_t nblocks = (uint64_t)1 << (c->n_local - cb);
        hipLaunchKernelGGL(k_basis_front_compact, dim3(grid_for(nblocks, 256, 65536)), dim3(256), 0, c->st[r], c->buf[c->cur][r], c->n_local, B, E);
        HIP_TRY(hipGetLastError());
    }
    for (size_t i = used; i < sh->queue.size(); i++) {
        SGate g = sh->queue[i];
        g.q -= M - cb;
        if (g.type == FUSE_PHASE) g.q2 -= M - cb;
        c->queue.push_back(g);
    }
    QCX_TRY(sh_flush(c));
    QCX_TRY(sh_identity(c));
    for (unsigned r = 0; r < sh->W; r++) {                  // every companion shard is complete (peers push into it; its streams are not the register's)
        SH_DEV(c, r);
        HIP_TRY(hipStreamSynchronize(c->st[r]));
        HIP_TRY(hipStreamSynchronize(c->xs[r]));
    }
    for (size_t i = 0; i < c->relay_st.size(); i++) { HIP_TRY(hipSetDevice(c->relay_dev[i])); HIP_TRY(hipStreamSynchronize(c->relay_st[i])); }
    sh->comp_pending = true;                                // the real register is written by sh_expand_pending -- or never
    sh->comp_E = E;
    sh->exchanges += c->exchanges - ex0; sh->pack_passes += c->pack_passes - pp0;
    sh->relayed_bytes += c->relayed_bytes - rb0; sh->overlapped_gates += c->overlapped_gates - og0;
    sh->basis_pending = false;
    sh->queue.clear();
    sh->fronts++;
    sh->compact_circuits++;
    *done = true;
    return QCX_NO_ERROR;
}

// the real register from the companion's compact form (every shard expands its own part)
static int sh_expand_pending(ShardSet *sh)
{
    if (!sh->comp_pending) return QCX_NO_ERROR;
    sh->comp_pending = false;
    ShardSet *c = sh->comp;
    if (!c) { set_error("sharded register: a compact result without its companion"); return QCX_UNKNOWN_ERROR; }
    const Tune tn = tune_now();
    const uint64_t nchunks = ((uint64_t)1 << (sh->n_local - sh->M)) >> 6;
    for (unsigned r = 0; r < sh->W; r++) {
        SH_DEV(sh, r);
        hipLaunchKernelGGL(k_expand_compact, dim3(grid_for(nchunks, 1, 65536)), dim3(256), 0, sh->st[r], (const amp_t *)c->buf[c->cur][r], sh->buf[sh->cur][r], nchunks, sh->comp_E, (int)tn.fuse_expand_direct);
        HIP_TRY(hipGetLastError());
        // the expansion READS the companion's buffer on the register's stream: whatever the companion's own streams do next
        // (the front of the next compact circuit overwrites that buffer) has to wait for it
        if (r < c->ev_a.size() && c->ev_a[r]) {
            HIP_TRY(hipEventRecord(c->ev_a[r], sh->st[r]));
            HIP_TRY(hipStreamWaitEvent(c->st[r], c->ev_a[r], 0));
            HIP_TRY(hipStreamWaitEvent(c->xs[r], c->ev_a[r], 0));
        } else HIP_TRY(hipStreamSynchronize(sh->st[r]));
    }
    return QCX_NO_ERROR;
}

static int sh_flush(ShardSet *sh, bool keep_compact)
{
    if (sh->comp_pending) {                       // an earlier flush left the state on the companion register
        if (keep_compact && sh->queue.empty() && !sh->basis_pending) return QCX_NO_ERROR;
        if (sh->basis_pending) sh->comp_pending = false;        // (a reset came after it: the compact form is history)
        else QCX_TRY(sh_expand_pending(sh));
    }
    if (sh->basis_pending && !sh->queue.empty()) {
        bool done = false;
        QCX_TRY(sh_compact(sh, &done));
        if (done) return keep_compact ? QCX_NO_ERROR : sh_expand_pending(sh);      // (only measure_state leaves the result on the companion)
    }
    if (sh->basis_pending) QCX_TRY(sh_materialize_basis(sh));
    if (sh->queue.empty()) return QCX_NO_ERROR;
    if (sh->zeros_dirty) {
        sh->zeros_dirty = false;
        if (!sh->dry) for (unsigned r = 0; r < sh->W; r++) { SH_DEV(sh, r); QCX_TRY(qcx_shard_canon_zeros(sh->buf[sh->cur][r], sh->n_local, sh->st[r])); }
    }
    std::vector<SGate> q;
    q.swap(sh->queue);
    const bool windows = sh->overlap && sh->sigma > 0 && sh->relay_dev.empty();
    auto global_h = [&](const SGate &g) { return g.type == FUSE_H && sh->perm[g.q] >= sh->n_local; };
    size_t i = 0;
    while (i < q.size()) {
        size_t x = i;
        while (x < q.size() && !global_h(q[x])) x++;
        if (x == q.size()) { QCX_TRY(sh_run_ops(sh, q.data() + i, x - i)); break; }
        // gates that can share the pipeline with the exchange at x: a run before it ...
        if (!windows) {                                // no overlap: everything up to the gate that needs the trade, the trade, go on
            QCX_TRY(sh_run_ops(sh, q.data() + i, x - i));
            QCX_TRY(sh_exchange(sh, sh_plan_give(sh, sh_choose_give(sh, q, x))));
            i = x;                                     // the H at x is local now and joins what follows
            continue;
        }
        size_t a = x;
        while (a > i && sh_sliceable(sh, q[a - 1])) a--;
        QCX_TRY(sh_run_ops(sh, q.data() + i, a - i));
        const std::vector<POp> pre = sh_phys_list(sh, q.data() + a, x - a);        // under the layout BEFORE the trade
        const SwapList swaps = sh_plan_give(sh, sh_choose_give(sh, q, x));
        // ... and a run after it, under the layout the trade leaves behind (peek: book on copies of the two tables)
        std::vector<unsigned> perm_after = sh->perm, inv_after = sh->inv;
        sh_book_vec(perm_after, inv_after, swaps, true, sh->k, sh->zone_lo, sh->n_local);
        auto global_h_after = [&](const SGate &g) { return g.type == FUSE_H && perm_after[g.q] >= sh->n_local; };
        auto sliceable_after = [&](const SGate &g) { return g.type != FUSE_H || perm_after[g.q] < sh->slice_bits; };
        size_t b = x;
        while (b < q.size() && sliceable_after(q[b]) && !global_h_after(q[b])) b++;
        if (b < q.size() && global_h_after(q[b]))
            b = x + std::max<size_t>(1, (b - x + 1) / 2);       // the run ends at the NEXT trade: leave it half for its pre-window
        if (b == x) {                                  // (the H at x targets a spectator bit after the trade: cannot happen --
            QCX_TRY(sh_exchange(sh, swaps, pre, std::vector<POp>()));       //  give positions lie below the spectators -- but stay safe)
            i = x;
            continue;
        }
        std::vector<POp> post;
        {
            std::vector<unsigned> keep = sh->perm;         // resolve the post-window under the layout after the trade
            sh->perm.swap(perm_after);
            post = sh_phys_list(sh, q.data() + x, b - x);
            sh->perm.swap(perm_after);
            (void)keep;
        }
        QCX_TRY(sh_exchange(sh, swaps, pre, post));
        i = b;
    }
    return QCX_NO_ERROR;
}

static int sh_push(ShardSet *sh, const SGate &g)
{
    sh->queue.push_back(g);
    if (sh->queue.size() >= sh->max_queue) return sh_flush(sh);
    return QCX_NO_ERROR;
}

static bool sh_is_identity(const ShardSet *sh)
{
    for (unsigned q = 0; q < sh->n; q++) if (sh->perm[q] != q) return false;
    return true;
}

// restore logical == physical (the order measurement and read-back need)
static int sh_identity(ShardSet *sh)
{
    QCX_TRY(sh_flush(sh));
    if (sh_is_identity(sh)) return QCX_NO_ERROR;
    const unsigned n = sh->n, nl = sh->n_local, k = sh->k;
    bool moved = false, some_in_id = false;
    for (unsigned g = nl; g < n; g++) { moved |= sh->perm[g] != g; some_in_id |= sh->perm[g] >= nl; }
    if (moved) {
        if (some_in_id) {
            // some rightful shard-id qubits sit in the shard id but in the wrong slot / beside strangers: one trade
            // brings the whole shard id local (giving up positions that hold none of them)
            std::vector<unsigned> cand;
            for (unsigned p = sh->slice_bits; p-- > sh->min_evict && cand.size() < k;) if (sh->inv[p] < nl) cand.push_back(p);
            QCX_TRY(sh_exchange(sh, sh_plan_give(sh, cand)));
        }
        std::vector<unsigned> give;
        for (unsigned j = 0; j < k; j++) give.push_back(sh->perm[nl + j]);      // shard-id bit j <- logical qubit nl + j
        QCX_TRY(sh_exchange(sh, sh_plan_give(sh, give)));
    }
    SwapList swaps;
    std::vector<unsigned> perm = sh->perm, inv = sh->inv;
    for (unsigned q = 0; q < nl; q++) {
        const unsigned a = perm[q];
        if (a != q) {
            swaps.push_back({a, q});
            const unsigned other = inv[q];
            perm[q] = q; perm[other] = a; inv[q] = q; inv[a] = other;
        }
    }
    QCX_TRY(sh_local_permute(sh, swaps));
    if (!sh_is_identity(sh)) { set_error("sharded register: identity layout not restored"); return QCX_UNKNOWN_ERROR; }
    return QCX_NO_ERROR;
}

static int sh_sync(ShardSet *sh)
{
    QCX_TRY(sh_flush(sh));
    if (sh->dry) return QCX_NO_ERROR;
    for (unsigned r = 0; r < sh->W; r++) { SH_DEV(sh, r); HIP_TRY(hipStreamSynchronize(sh->st[r])); }
    return QCX_NO_ERROR;
}

// Relay GPUs for multi-path striping.  Shares: with R relays and W shards the direct link of a pair carries the
// fraction a = (W-1)/(R+W-1) of its chunk and every relay 1/(R+W-1) of EVERY chunk, which loads all links out of a GPU
// equally (a relay link carries the stripes of the W-1 destinations).  Both hops of a relayed stripe cost link time,
// so the model gain is  1 / (2 a)  for W = 2 (3.5x with 6 relays) and shrinks as W grows (tools/model_sharded.py).
static int sh_set_relays(ShardSet *sh, unsigned nrelays, const int *devices)
{
    if (sh->dry) return QCX_NO_ERROR;
    if (sh->comp) { sh_free(sh->comp); sh->comp = nullptr; }       // (the companion of compact circuits is rebuilt with the new relays on its next use)
    QCX_TRY(sh_identity(sh));                      // the trade zone moves with the slice geometry: identity layout, nothing queued
    QCX_TRY(sh_sync(sh));
    sh_drop_relays(sh);
    sh_set_slices(sh, (nrelays || !sh->overlap) ? 0u : sh_default_sigma());     // relay stripes are cut from whole shards: no slices with relays
    if (!nrelays) return QCX_NO_ERROR;
    if (nrelays > 8 || !devices) return QCX_BAD_ARGUMENTS;
    if (sh->zone_lo < 8) return QCX_NO_ERROR;                      // tiny shards: nothing worth striping
    const uint64_t blocks = ((uint64_t)1 << sh->zone_lo) / 256;
    if (blocks < 2 * (uint64_t)(nrelays + 1)) return QCX_NO_ERROR;
    const unsigned R = nrelays, W = sh->W;
    uint64_t nb_relay = blocks / (R + W - 1);
    if (!nb_relay) nb_relay = 1;
    uint64_t nb_direct = blocks - nb_relay * R;                   // direct takes what the relays leave (>= its share)
    const uint64_t last = blocks - nb_direct - nb_relay * (R - 1);
    sh->nb_direct = (uint32_t)nb_direct; sh->nb_relay = (uint32_t)nb_relay;
    sh->stage_amps = std::max(nb_relay, last) * 256;
    int ndev = 0;
    QCX_TRY(qcx_device_count(&ndev));
    for (unsigned i = 0; i < R; i++) {
        if (devices[i] < 0 || devices[i] >= ndev) { sh_drop_relays(sh); set_error("relay device %d: %d visible", devices[i], ndev); return QCX_HIP_ERROR; }
        sh->relay_dev.push_back(devices[i]);
        sh->relay_st.push_back(nullptr); sh->relay_ev.push_back(nullptr); sh->relay_stage.push_back(nullptr);
        hipError_t e = hipSetDevice(devices[i]);
        for (unsigned c = 0; c < W && e == hipSuccess; c++) {
            if (sh->dev[c] == devices[i]) continue;
            hipError_t pe = hipDeviceEnablePeerAccess(sh->dev[c], 0);            // relay -> owners (forwarding)
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) e = pe;
            (void)hipGetLastError();
            if (e == hipSuccess) e = hipSetDevice(sh->dev[c]);
            if (e == hipSuccess) { pe = hipDeviceEnablePeerAccess(devices[i], 0);  // shards -> relay (staging stores)
                                   if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) e = pe; (void)hipGetLastError(); }
            if (e == hipSuccess) e = hipSetDevice(devices[i]);
        }
        if (e == hipSuccess) e = hipStreamCreate(&sh->relay_st[i]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->relay_ev[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc(&sh->relay_stage[i], (size_t)W * W * sh->stage_amps * sizeof(amp_t));
        if (e != hipSuccess) {
            set_error("relay on device %d: %s", devices[i], hipGetErrorString(e));
            sh_drop_relays(sh);
            return e == hipErrorOutOfMemory ? QCX_INSUFFICIENT_MEMORY : QCX_HIP_ERROR;
        }
    }
    // the relayed path (staging stores into the relays, forwarding copies) gets the same pre-flight check as the direct one
    bool run = false;
    for (unsigned i = 0; i < R; i++) run |= devices[i] != sh->dev[0];
    for (unsigned r = 1; r < W; r++) run |= sh->dev[r] != sh->dev[0];
    if (const char *e = getenv("QCX_SHARD_SELFCHECK")) run = atoi(e) != 0;
    if (run && !sh->in_selfcheck) {
        const int cs = sh_selfcheck(sh, R, devices);
        if (cs != QCX_NO_ERROR) { sh_drop_relays(sh); sh_set_slices(sh, sh->overlap ? sh_default_sigma() : 0u); return cs; }
    }
    return QCX_NO_ERROR;
}

static int sh_reset(ShardSet *sh)
{
    sh->queue.clear();                              // pending gates act on a state that is being overwritten
    sh->zeros_dirty = false;
    sh_identity_perm(sh);
    if (sh->dry) { sh->trace += "reset\n"; return QCX_NO_ERROR; }
    if (sh->fusion > 0) { sh->basis_pending = true; return QCX_NO_ERROR; }      // lazily: see sh_materialize_basis
    sh->basis_pending = false;
    for (unsigned r = 0; r < sh->W; r++) { SH_DEV(sh, r); QCX_TRY(qcx_shard_reset(sh->buf[sh->cur][r], sh->n_local, r == 0, sh->st[r])); }
    return QCX_NO_ERROR;
}

static int sh_fill_random(ShardSet *sh, uint64_t seed)
{
    sh->queue.clear();
    sh->basis_pending = false;
    sh->zeros_dirty = false;
    sh_identity_perm(sh);
    if (sh->dry) return QCX_UNSUPPORTED;
    const double scale = sqrt(6.0 / (double)((uint64_t)1 << sh->n));
    for (unsigned r = 0; r < sh->W; r++) {
        SH_DEV(sh, r);
        QCX_TRY(qcx_shard_fill_random(sh->buf[sh->cur][r], sh->n_local, (uint64_t)r << sh->n_local, seed, scale, sh->st[r]));
    }
    return QCX_NO_ERROR;
}

static int sh_norm2(ShardSet *sh, double *out)
{
    QCX_TRY(sh_flush(sh));
    if (sh->dry) return QCX_UNSUPPORTED;
    double t = 0.0;
    for (unsigned r = 0; r < sh->W; r++) {
        SH_DEV(sh, r);
        double v = 0.0;
        QCX_TRY(qcx_shard_norm2(sh->buf[sh->cur][r], sh->n_local, &v, sh->st[r]));
        t += v;
    }
    *out = t;
    return QCX_NO_ERROR;
}

// Q:272-306 over the shards: the sequential cumulative sum is handed from shard to shard in index order
static int sh_measure(ShardSet *sh, double rnd, unsigned long *state_num)
{
    QCX_TRY(sh_flush(sh, true));                                                // (a compact circuit's result may stay on the companion)
    if (sh->comp_pending && !sh->dry) {
        // the scan of Q:283-292 on the compact form, shard by shard (qcx_measure_state_r does the same on one GPU): the amplitudes it
        // leaves out are +0, the compact order is the index order; r <= 0 stops at index 0 whatever it holds; the register's last
        // index stays unexamined also when it lies on the orbit
        ShardSet *c = sh->comp;
        const ExpandParams &E = sh->comp_E;
        const uint64_t dim = (uint64_t)1 << sh->n;
        uint64_t idx = dim - 1;
        bool have = false, rescan = false;
        if (rnd <= 0.0) { idx = 0; have = true; }
        else {
            uint64_t last_excl = (uint64_t)1 << c->n;
            if (E.orbit[E.ncols - 1] == (1u << E.M) - 1u) last_excl = ((((uint64_t)1 << (sh->n - E.M)) - 1) << E.cb) | (E.ncols - 1);
            double cum = 0.0;
            for (unsigned r = 0; r < c->W && !have && !rescan; r++) {
                SH_DEV(c, r);
                int found = 0; uint64_t ci = 0; double c2 = cum;
                QCX_TRY(qcx_shard_measure_scan(c->buf[c->cur][r], c->n_local, (uint64_t)r << c->n_local, last_excl, cum, rnd, &found, &ci, &c2, c->st[r]));
                cum = c2;
                if (found) {
                    const unsigned col = (unsigned)(ci & ((1u << E.cb) - 1u));
                    if (col >= E.ncols) rescan = true;                        // a hit in a padding column: the premise broke -- expand and scan the register
                    else { idx = ((ci >> E.cb) << E.M) | E.orbit[col]; have = true; }
                }
            }
            if (!rescan) have = true;                                           // (nothing found: Q:283 fall-through to the last index)
        }
        if (have && !rescan) {
            sh->comp_pending = false;                                           // the collapse below replaces the whole state
            sh->compact_measures++;
            const unsigned owner = (unsigned)(idx >> sh->n_local);
            for (unsigned r = 0; r < sh->W; r++) {
                SH_DEV(sh, r);
                QCX_TRY(qcx_shard_collapse(sh->buf[sh->cur][r], sh->n_local, r == owner ? (int64_t)(idx & ((((uint64_t)1) << sh->n_local) - 1)) : -1, sh->st[r]));
            }
            *state_num = (unsigned long)idx;
            return QCX_NO_ERROR;
        }
        QCX_TRY(sh_expand_pending(sh));
    }
    QCX_TRY(sh_identity(sh));
    if (sh->dry) return QCX_UNSUPPORTED;
    QCX_TRY(sh_sync(sh));
    const uint64_t dim = (uint64_t)1 << sh->n, last_excluded = dim - 1;
    double cum = 0.0;
    uint64_t idx = last_excluded;                                               // Q:283 fall-through
    for (unsigned r = 0; r < sh->W; r++) {
        SH_DEV(sh, r);
        int found = 0; uint64_t i = 0; double c2 = cum;
        QCX_TRY(qcx_shard_measure_scan(sh->buf[sh->cur][r], sh->n_local, (uint64_t)r << sh->n_local, last_excluded, cum, rnd,
                                       &found, &i, &c2, sh->st[r]));
        cum = c2;
        if (found) { idx = i; break; }
    }
    const unsigned owner = (unsigned)(idx >> sh->n_local);
    for (unsigned r = 0; r < sh->W; r++) {
        SH_DEV(sh, r);
        QCX_TRY(qcx_shard_collapse(sh->buf[sh->cur][r], sh->n_local, r == owner ? (int64_t)(idx & ((((uint64_t)1) << sh->n_local) - 1)) : -1, sh->st[r]));
    }
    *state_num = (unsigned long)idx;
    return QCX_NO_ERROR;
}

// host <-> shards, identity layout: amplitudes [first, first + count)
static int sh_copy(ShardSet *sh, uint64_t first, uint64_t count, double *host, bool to_host)
{
    QCX_TRY(sh_identity(sh));
    if (sh->dry) return QCX_UNSUPPORTED;
    QCX_TRY(sh_sync(sh));
    const uint64_t per = (uint64_t)1 << sh->n_local;
    uint64_t at = first, left = count;
    while (left) {
        const unsigned r = (unsigned)(at >> sh->n_local);
        const uint64_t off = at & (per - 1), cnt = std::min(left, per - off);
        SH_DEV(sh, r);
        amp_t *d = sh->buf[sh->cur][r] + off;
        double *h = host + 2 * (at - first);
        if (to_host) HIP_TRY(hipMemcpy(h, d, (size_t)cnt * sizeof(amp_t), hipMemcpyDeviceToHost));
        else { HIP_TRY(hipMemcpy(d, h, (size_t)cnt * sizeof(amp_t), hipMemcpyHostToDevice)); sh->zeros_dirty = true; }
        at += cnt; left -= cnt;
    }
    return QCX_NO_ERROR;
}

// ---- pre-flight self-check -------------------------------------------------------------------------------------------
// A small register (M = 0, at most 2^(2k+12) amplitudes per shard) on the SAME devices, with the same slice geometry
// rules and, if asked, the same relays:
//   (1) fill it with the counter-based generator (its CPU twin gives every expected amplitude, bit for bit);
//   (2) one trade (pack + push of all k shard-id bits, exactly as a Hadamard on a global qubit would trigger it);
//       every amplitude must now sit where the booked layout says it does -- checked on the raw shard buffers;
//   (3) restore the identity layout (more trades and a local permutation) and compare with the generator again.
// No arithmetic touches the amplitudes, so the comparison is exact.  QCX_SHARD_SELFCHECK_INJECT=1 corrupts one amplitude
// after step (2): the tests use it to prove that a wrong trade is caught.
static int sh_selfcheck(const ShardSet *big, unsigned nrelays, const int *relay_devices)
{
    const unsigned k = big->k;
    const unsigned nl = std::min<unsigned>(big->n_local, 2 * k + 12);
    ShardSet *t = nullptr;
    int st = sh_create((int)(nl + k), 0, big->W, big->dev.data(), &t, false);
    if (st != QCX_NO_ERROR) {
        const std::string why = g_last_error;
        set_error("sharded register self-check: cannot build the %u-qubit test register (%s)", nl + k, why.c_str());
        return st;
    }
    t->in_selfcheck = true;
    auto fail = [&](int code) { sh_free(t); return code; };
    if (nrelays) { st = sh_set_relays(t, nrelays, relay_devices); if (st != QCX_NO_ERROR) return fail(st); }
    const uint64_t seed = 0x5e1fc4ecULL, per = (uint64_t)1 << nl, dim = per << k;
    const double scale = sqrt(6.0 / (double)dim);
    auto expect = [&](uint64_t logical, double *re, double *im) {
        const uint64_t c = 2 * logical;
        *re = ((double)(splitmix64(seed + c) >> 11) * 0x1p-53 - 0.5) * scale;
        *im = ((double)(splitmix64(seed + c + 1) >> 11) * 0x1p-53 - 0.5) * scale;
    };
    if ((st = sh_fill_random(t, seed)) != QCX_NO_ERROR) return fail(st);
    // (2) one trade; give positions as the scheduler would choose them with nothing queued
    const std::vector<SGate> none;
    if ((st = sh_exchange(t, sh_plan_give(t, sh_choose_give(t, none, 0)))) != QCX_NO_ERROR) return fail(st);
    if ((st = sh_sync(t)) != QCX_NO_ERROR) return fail(st);
    if (const char *e = getenv("QCX_SHARD_SELFCHECK_INJECT"))
        if (atoi(e)) { (void)hipSetDevice(t->dev[t->W - 1]); (void)hipMemset(t->buf[t->cur][t->W - 1] + (per >> 1), 0x3c, sizeof(amp_t)); }
    std::vector<double> host(2 * per);
    uint64_t bad = 0, first_bad = 0; unsigned bad_shard = 0;
    for (unsigned r = 0; r < t->W; r++) {
        if (hipSetDevice(t->dev[r]) != hipSuccess ||
            hipMemcpy(host.data(), t->buf[t->cur][r], per * sizeof(amp_t), hipMemcpyDeviceToHost) != hipSuccess) {
            set_error("sharded register self-check: read-back of shard %u failed: %s", r, hipGetErrorString(hipGetLastError()));
            return fail(QCX_HIP_ERROR);
        }
        for (uint64_t j = 0; j < per; j++) {
            const uint64_t phys = ((uint64_t)r << nl) | j;
            uint64_t logical = 0;
            for (unsigned q = 0; q < t->n; q++) logical |= ((phys >> t->perm[q]) & 1u) << q;
            double re, im;
            expect(logical, &re, &im);
            if (memcmp(&re, &host[2 * j], 8) || memcmp(&im, &host[2 * j + 1], 8)) { if (!bad++) { first_bad = j; bad_shard = r; } }
        }
    }
    if (bad) {
        set_error("sharded register self-check FAILED after one trade: %llu amplitudes wrong (first: shard %u on device %d, local index %llu)",
                  (unsigned long long)bad, bad_shard, t->dev[bad_shard], (unsigned long long)first_bad);
        return fail(QCX_HIP_ERROR);
    }
    // (3) back to the identity layout
    if ((st = sh_identity(t)) != QCX_NO_ERROR || (st = sh_sync(t)) != QCX_NO_ERROR) return fail(st);
    for (unsigned r = 0; r < t->W; r++) {
        if (hipSetDevice(t->dev[r]) != hipSuccess ||
            hipMemcpy(host.data(), t->buf[t->cur][r], per * sizeof(amp_t), hipMemcpyDeviceToHost) != hipSuccess) return fail(QCX_HIP_ERROR);
        for (uint64_t j = 0; j < per; j++) {
            double re, im;
            expect(((uint64_t)r << nl) | j, &re, &im);
            if (memcmp(&re, &host[2 * j], 8) || memcmp(&im, &host[2 * j + 1], 8)) { if (!bad++) { first_bad = j; bad_shard = r; } }
        }
    }
    if (bad) {
        set_error("sharded register self-check FAILED after restoring the identity layout: %llu amplitudes wrong (first: shard %u on device %d, local index %llu)",
                  (unsigned long long)bad, bad_shard, t->dev[bad_shard], (unsigned long long)first_bad);
        return fail(QCX_HIP_ERROR);
    }
    sh_free(t);
    const_cast<ShardSet *>(big)->selfchecks++;
    return QCX_NO_ERROR;
}
