// qcx_fuse.inc.h -- lazy gate queue and pass scheduler for the fused tile kernel (k_fused).
// Included into qcx_api.hip (single translation unit) after struct qcx_register.
//
// With fusion enabled (qcx_set_fusion) the gate entry points only append to a queue.  Anything that
// observes the state (read, norm, measure, synchronize, timers) flushes it: the queue is cut greedily
// into PASSES, each one launch of k_fused over tiles of 2^T amplitudes = the c lowest index bits plus
// up to T - c "hot" higher bits.  A gate joins the current pass if the bits it needs inside the tile
// (the target of an H; the M register of a modular multiply) still fit; controlled phases are
// diagonal and always join.  Gates keep their issue order inside a pass, and each performs the same
// arithmetic as its stand-alone kernel, so fused and unfused results are bit-identical.

struct QGate {
    uint32_t type;            // FUSE_H / FUSE_PHASE / FUSE_CAMODC, or 99 = C_AMODC that must run stand-alone
    unsigned q;               // H: target.  CAMODC: control qubit
    uint64_t mask;            // PHASE: control|target mask
    double   c, s;            // PHASE
    unsigned C, A;            // CAMODC
};

// TOLERANCE MODE (qcx_set_fusion(reg, 2)): a run of consecutive controlled phases that share a qubit (their
// "control": the gates are symmetric in their two qubits) becomes ONE diagonal -- amp[i] *= prod over the other qubits k of
// the run with bit k of i set of (c_k + i s_k), for every i with the shared bit set (Q:682-689 issues
// exactly such runs: all controlled phases after H(l) share l).  The factors of a qubit that occurs twice are multiplied
// on the host (long double).  Not bit-exact: see K6t in qcx_kernels.h.
struct DiagSpec {
    unsigned ctl;                 // the shared qubit
    uint64_t tmask;               // the other qubits of the run
    double   wc[40], ws[40];      // their factors (cos, sin), (1, 0) where not a target
    double   kc, ks;              // constant factor: the run's phases on the shared qubit ALONE (their other qubit is a constant 1 of
                                  // the view -- a shard-id or slice bit: sharded registers hand such gates down with one mask bit)
    size_t   first, count;        // the run in the ORIGINAL gate list
};

static void merge_diagonals(const std::vector<QGate> &in, std::vector<QGate> &out, std::vector<DiagSpec> &specs,
                            std::vector<size_t> &ofirst, std::vector<size_t> &ocnt)
{
    auto phase_bits = [&](size_t i) { return in[i].type == FUSE_PHASE ? __builtin_popcountll(in[i].mask) : -1; };
    // a run member on shared qubit b: a two-qubit phase containing b, or a phase on b alone
    auto member = [&](size_t i, unsigned b) { const int pb = phase_bits(i); return (pb == 2 && ((in[i].mask >> b) & 1)) || (pb == 1 && in[i].mask == ((uint64_t)1 << b)); };
    size_t i = 0;
    unsigned last_h = 64;                  // target of the Hadamard just before (a single phase takes it as its control)
    while (i < in.size()) {
        size_t best_len = 0; unsigned best_ctl = 0;
        const int pb0 = phase_bits(i);
        if (pb0 == 1 || pb0 == 2) {
            for (unsigned b = 0; b < 64; b++) {
                if (!((in[i].mask >> b) & 1)) continue;
                size_t j = i, two = 0;
                while (j < in.size() && member(j, b)) { two += phase_bits(j) == 2; j++; }
                if (!two) continue;                                        // nothing but constants: not worth a diagonal
                if (j - i > best_len || (j - i == best_len && b == last_h)) { best_len = j - i; best_ctl = b; }
            }
        }
        if (best_len < 1) {
            if (in[i].type == FUSE_H) last_h = in[i].q; else last_h = 64;
            out.push_back(in[i]); ofirst.push_back(i); ocnt.push_back(1); i++; continue;
        }
        last_h = 64;
        DiagSpec d; memset(&d, 0, sizeof d);
        d.ctl = best_ctl; d.first = i; d.count = best_len;
        long double wc[40], ws[40], kc = 1.0L, ks = 0.0L;
        for (unsigned k = 0; k < 40; k++) { wc[k] = 1.0L; ws[k] = 0.0L; }
        for (size_t j = i; j < i + best_len; j++) {
            const long double c = in[j].c, s2 = in[j].s;
            if (phase_bits(j) == 1) { const long double x = kc, y = ks; kc = x * c - y * s2; ks = x * s2 + y * c; continue; }
            const unsigned k = (unsigned)__builtin_ctzll(in[j].mask & ~((uint64_t)1 << best_ctl));
            d.tmask |= (uint64_t)1 << k;
            const long double x = wc[k], y = ws[k];
            wc[k] = x * c - y * s2; ws[k] = x * s2 + y * c;
        }
        for (unsigned k = 0; k < 40; k++) { d.wc[k] = (double)wc[k]; d.ws[k] = (double)ws[k]; }
        d.kc = (double)kc; d.ks = (double)ks;
        QGate g; memset(&g, 0, sizeof g);
        g.type = FUSE_DIAG; g.q = best_ctl; g.mask = d.tmask; g.C = (unsigned)specs.size();
        specs.push_back(d);
        out.push_back(g); ofirst.push_back(i); ocnt.push_back(best_len);
        i += best_len;
    }
}

struct FuseAction {
    int      fused;          // 0: stand-alone gate `gate`, 1: fused pass
    size_t   gate;
    FusePass P;
    size_t   op_off, op_cnt, ngates, first_gate;
    int      nopipe;         // phase-dominated pass (planned on the smaller tile of fuse_T_phase); reported by qcx_fusion_plan
    uint8_t  tl[16];         // the qubit that is tile-local bit j (the records' numbering)
};

struct GateQueue {
    std::vector<QGate> gates;
    FuseOp  *d_ops = nullptr;       // device copy of the ops of the passes in flight
    size_t   d_cap = 0;
    FuseOp  *h_ops = nullptr;       // pinned staging
    size_t   h_cap = 0;
    unsigned long passes_launched = 0, gates_fused = 0, chained_passes = 0, gen_fronts = 0, gen_cols = 0, compact_chains = 0, expanding_stores = 0;
    hipEvent_t ev;                  // recorded after the last kernel of a flush: guards the record buffers
    bool     ev_valid = false;
    // the LAST pass of a compact chain that was asked to stay compact (qcx_register::compact_pending == 2): not launched yet.
    // It runs when somebody looks -- as an ordinary pass when that is measure_state (which scans the compact form), with the
    // expanding store (FusePass::xp_on: straight into the register, no k_expand_compact) for everybody else -- or never, when
    // a reset comes first.  Its records stay in d_ops: nothing uploads before the pending state is resolved (fuse_flush).
    struct {
        FusePass P, Pxp;            // as planned / with the expanding store
        size_t   op_off = 0;
        bool     nopipe = false;
        amp_t   *in = nullptr, *out = nullptr;      // compact buffers: the pass reads in, an ordinary launch writes out (== in: in place)
        unsigned nv = 0, cb = 0, ngates = 0;
        Tune     tn;                // the knobs the plan was made under
    } last;
    // The plan of the last flush, kept: a period-finding run issues the same circuit attempt after attempt, and planning it
    // (cutting the list into passes, the records, thread maps, merged diagonals: 0.2-0.4 ms for the 406 gates of an n = 28 inverse
    // QFT, on the host, with the GPU idle) plus uploading the records is pure repetition.  A flush whose inputs are bit for
    // bit those of the cached one -- register shape, fusion mode, every knob, the circuit front, the gate list -- reuses
    // actions and records; when nothing has been uploaded since, the records are still on the device as well.
    // kind 1: a flush without a front (fuse_flush's plain path); 2: a compact chain.
    struct {
        bool     valid = false;
        int      kind = 0;
        unsigned n = 0, M = 0;
        int      fusion = 0;
        bool     chain = false;
        bool     front_flush = false, gen_try = false;      // kind 1: the flush stood behind a lazily pending basis state / generated its front
        bool     gen_built = false, gen2 = false;           // ... and how that went (what a hit replays)
        Tune     tn;
        BasisFront Bf;
        size_t   kfront = 0;
        std::vector<QGate> gates;
        std::vector<FuseAction> acts;
        std::vector<FuseOp> all_ops;
        std::vector<uint16_t> orbit;    // kind 2
        unsigned cb = 0;                // kind 2
        unsigned long stamp = 0;        // upload_stamp right after these records were uploaded
    } pc;
    unsigned long upload_stamp = 0, plan_hits = 0;
};

static bool same_gate_list(const std::vector<QGate> &a, const std::vector<QGate> &b)
{
    return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(QGate)) == 0);
}

static void queue_free(GateQueue *gq)
{
    if (!gq) return;
    if (gq->d_ops) (void)hipFree(gq->d_ops);
    if (gq->h_ops) (void)hipHostFree(gq->h_ops);
    if (gq->ev_valid) (void)hipEventDestroy(gq->ev);
    delete gq;
}

// bytes of the scratch area behind a pass's tile: the source table of a single modular multiply (2 bytes per residue of
// the M register) or, for a folded run of them, one byte per 2^M-block of the largest tile (the run's per-block factor)
static size_t cam_lut_bytes(unsigned M)
{
    const unsigned m = std::min(12u, M);
    return std::max((size_t)2 << m, (size_t)1 << (12u - m)) + 16;
}

static bool camodc_closed_form(unsigned n, unsigned M, unsigned C, unsigned A, unsigned ctl)
{
    (void)A;
    if (M > n || ctl < M || C == 0) return false;
    const uint64_t blk = (uint64_t)1 << M;
    if (C > blk) return false;
    return (uint64_t)(C - 1) * (uint64_t)(C - 1) <= 0xffffffffULL;
}



static int launch_standalone(qcx_register *r, const QGate &g)
{
    if (g.type == FUSE_H) return qcx_shard_hadamard(r->amp, r->n, g.q, r->stream);
    if (g.type == FUSE_PHASE) return qcx_shard_phase(r->amp, r->n, g.mask, g.c, g.s, r->stream);
    if (g.q == 0xffffffffu)          // shard mode: the control is a rank bit that is 1
        return qcx_shard_camodc(r->amp, r->n, (unsigned)r->M, g.C, g.A, -1, r->stream);
    return reg_camodc(r, g.C, g.A, g.q);
}

// tile-local form of the gates [first, last) for a tile with hot bits `hbits` (one record per gate)
// `specs` / `orig` / `expand`: tolerance mode.  A merged diagonal becomes one FUSE_DIAG record (its slot = position in
// `pass_diags`, whose tables diag_tables() builds), or -- expand -- the plain phases it was merged from (passes that do not
// run in the rounds form have no diagonal interpreter).
static void build_pass_ops(const qcx_register *r, const std::vector<QGate> &gates, size_t first, size_t last,
                           const std::vector<unsigned> &tl, std::vector<FuseOp> &out,
                           const std::vector<DiagSpec> *specs = nullptr, const std::vector<QGate> *orig = nullptr,
                           bool expand = false, std::vector<unsigned> *pass_diags = nullptr)
{
    // tl[j] = the qubit that is tile-local bit j (in place on the identity layout: the c low bits, then the hot bits ascending;
    // a chained pass orders them by where its input layout puts them)
    const unsigned n = r->n;
    auto local_of = [&](unsigned b, bool *inside) -> unsigned {
        auto it = std::find(tl.begin(), tl.end(), b);
        if (it != tl.end()) { *inside = true; return (unsigned)(it - tl.begin()); }
        *inside = false; return 0;
    };
    for (size_t k = first; k < last; k++) {
        const QGate &g = gates[k];
        FuseOp o;
        memset(&o, 0, sizeof o);
        o.type = g.type;
        bool in;
        if (g.type == FUSE_DIAG) {
            const DiagSpec &d = (*specs)[g.C];
            if (expand) {
                build_pass_ops(r, *orig, d.first, d.first + d.count, tl, out);
                continue;
            }
            const unsigned lc = local_of(d.ctl, &in);
            uint64_t tloc = 0; unsigned groups = 0;
            for (unsigned b = 0; b < n; b++) {
                if (!((d.tmask >> b) & 1)) continue;
                bool tin;
                const unsigned lb = local_of(b, &tin);
                if (tin) { tloc |= (uint64_t)1 << lb; groups |= 1u << (lb >> 2); }
            }
            o.a = (in ? lc + 1 : 0u) | ((unsigned)pass_diags->size() << 8) | (groups << 16);
            o.mask = in ? 0 : (uint64_t)1 << d.ctl;
            memcpy(&o.c, &tloc, sizeof tloc);
            pass_diags->push_back(g.C);
            // the phases it was merged from follow as ALTERNATES (their number rides in the s field): to_rounds keeps
            // either the diagonal (fast round) or the phases
            const uint64_t nalt = d.count;
            memcpy(&o.s, &nalt, sizeof nalt);
            out.push_back(o);
            build_pass_ops(r, *orig, d.first, d.first + d.count, tl, out);
            continue;
        }
        if (g.type == FUSE_H) {
            o.a = local_of(g.q, &in);
        } else if (g.type == FUSE_PHASE) {
            // split the control|target mask into tile-local bits (vector test) and outside bits (scalar test)
            uint32_t mloc = 0; uint64_t mext = 0;
            for (unsigned b = 0; b < n; b++) {
                if (!((g.mask >> b) & 1)) continue;
                const unsigned lb = local_of(b, &in);
                if (in) mloc |= 1u << lb; else mext |= (uint64_t)1 << b;
            }
            o.a = mloc; o.mask = mext; o.c = g.c; o.s = g.s;
        } else {
            // control: tile-local position (+1) in bits 8.., or an outside bit tested against the tile base
            unsigned lb = 0;
            in = false;
            if (g.q != 0xffffffffu) lb = local_of(g.q, &in);
            o.a = (unsigned)r->M | ((in ? lb + 1 : 0u) << 8);
            o.mask = (in || g.q == 0xffffffffu) ? 0 : (uint64_t)1 << g.q;     // 0xffffffff: always on
            FuseCamExtra X;
            X.C = g.C; X.d = gcd_u32(g.A, g.C); X.Cd = g.C / X.d; X.inv = modinv_u32(g.A / X.d, X.Cd);
            memcpy(&o.c, &X, sizeof X);
        }
        out.push_back(o);
    }
}

// tolerance mode: the table area of a pass's merged diagonals, in 16-byte units (one complex double each):
//   [DiagInfo x nd (3 units each)] [G tables: nd x 48] [field tables: 256 entries per (diagonal, byte of the base index)
//   that holds targets outside the tile]
static void diag_tables(unsigned n, const std::vector<unsigned> &tl, const std::vector<DiagSpec> &specs,
                        const std::vector<unsigned> &pass_diags, std::vector<double> &area)
{
    const size_t nd = pass_diags.size();
    area.assign(2 * (3 * nd + 48 * nd), 0.0);
    std::vector<int> local_of_bit(64, -1);
    for (size_t j = 0; j < tl.size(); j++) local_of_bit[tl[j]] = (int)j;
    for (size_t s = 0; s < nd; s++) {
        const DiagSpec &d = specs[pass_diags[s]];
        // G tables: group g covers tile-local bits 4g .. 4g+3; entry j = product of the factors of the targets among them
        for (unsigned g = 0; g < 3; g++)
            for (unsigned j = 0; j < 16; j++) {
                long double x = 1.0L, y = 0.0L;
                for (unsigned t = 0; t < 4; t++) {
                    if (!((j >> t) & 1u)) continue;
                    const unsigned lb = 4 * g + t;
                    unsigned gb = 64;
                    for (unsigned b = 0; b < n; b++) if (local_of_bit[b] == (int)lb) gb = b;
                    if (gb >= 64 || !((d.tmask >> gb) & 1)) continue;
                    const long double cc = d.wc[gb], ss = d.ws[gb], nx = x * cc - y * ss, ny = x * ss + y * cc;
                    x = nx; y = ny;
                }
                double *e = &area[2 * (3 * nd + 48 * s + 16 * g + j)];
                e[0] = (double)x; e[1] = (double)y;
            }
        // field tables: byte f of the tile's base index (global bits 8f .. 8f+7), targets outside the tile only
        DiagInfo info; memset(&info, 0, sizeof info);
        for (unsigned f = 0; f < 5; f++) {
            std::vector<unsigned> bits(8, 64);
            bool any = false;
            for (unsigned t = 0; t < 8; t++) {
                const unsigned gb = 8 * f + t;
                if (gb < n && ((d.tmask >> gb) & 1) && local_of_bit[gb] < 0) { bits[t] = gb; any = true; }
            }
            if (!any) continue;
            info.present |= 1u << f;
            info.field_off[f] = (uint32_t)(area.size() / 2);
            const size_t at = area.size();
            area.resize(at + 2 * 256);
            for (unsigned j = 0; j < 256; j++) {
                long double x = 1.0L, y = 0.0L;
                for (unsigned t = 0; t < 8; t++)
                    if (((j >> t) & 1u) && bits[t] < 64) {
                        const long double cc = d.wc[bits[t]], ss = d.ws[bits[t]], nx = x * cc - y * ss, ny = x * ss + y * cc;
                        x = nx; y = ny;
                    }
                area[at + 2 * j] = (double)x; area[at + 2 * j + 1] = (double)y;
            }
        }
        info.kc = d.kc; info.ks = d.ks;
        memcpy(&area[2 * (3 * s)], &info, sizeof info);
    }
    if (area.size() % 4) area.resize(area.size() + 2, 0.0);       // whole 32-byte records
}

// ---- k_fused_x8: the thread map of every round of a pass ------------------------------------------------------------------
// LDS slot of tile-local element e in k_fused_x8 (the kernel's x8_swz): the two upper nibbles folded onto the lowest
static inline unsigned x8_swz_host(unsigned e) { return e ^ ((e >> 4) & 15u) ^ ((e >> 8) & 15u); }

// bank-conflict cost of a lane order (lane bit k rides on tile bit lanes[k]) for the 16-byte LDS accesses of a round: a wave's
// ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32), each in one cycle
// when the 16 slots fall on different 16-byte bank groups (slot mod 16); ds_write_b128 in eight groups of 8 consecutive lanes over
// half the banks (slot mod 8).  Cost = sum over the groups of the worst multiplicity (12 = conflict-free).
static unsigned x8_lane_cost(const unsigned *lanes)
{
    unsigned slot[64];
    for (unsigned l = 0; l < 64; l++) {
        unsigned e = 0;
        for (unsigned k = 0; k < 6; k++) e |= ((l >> k) & 1u) << lanes[k];
        slot[l] = x8_swz_host(e);
    }
    static const unsigned char rgroup[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    unsigned cost = 0;
    for (unsigned half = 0; half < 2; half++)
        for (unsigned g = 0; g < 2; g++) {
            unsigned cnt[16] = {0}, worst = 0;
            for (unsigned k = 0; k < 16; k++) worst = std::max(worst, ++cnt[slot[32 * half + rgroup[g][k]] & 15u]);
            cost += worst;
        }
    for (unsigned g = 0; g < 8; g++) {
        unsigned cnt[8] = {0}, worst = 0;
        for (unsigned k = 0; k < 8; k++) worst = std::max(worst, ++cnt[slot[8 * g + k] & 7u]);
        cost += worst;
    }
    return cost;
}

// Who rides where, for every FUSE_ROUND8 round of the pass whose records start at out[first] (T - 3 thread bits: 6 lane bits, then
// T - 9 wave bits):
//  * the WAVE number rides on tile bits that no Hadamard of the coming rounds targets, and stays on them as long as that holds --
//    a wave then owns its part of the tile through all those rounds: header bit 25 tells the kernel to skip the workgroup
//    barrier in front of the round (the fillers of a pass with outside hot bits are passive for the whole pass: its waves never
//    meet between fill and store).  Among equally long-lived bits the ones most gates of the round test come first: a gate whose
//    tile-local condition sits on a wave bit is skipped by half of the waves with two instructions, on a lane bit it leaves half
//    of every wave idle through its rotations;
//  * the LANE number takes the other six non-register bits, in the order that costs the fewest LDS bank conflicts.
// fuse_x8_map = 0: ascending order, a barrier in front of every round.
static void x8_assign_maps(std::vector<FuseOp> &out, size_t first, unsigned T, const Tune &tn)
{
    struct Rd { size_t at; unsigned rb[3]; std::vector<unsigned> votes; };
    std::vector<Rd> rounds;
    // (FUSE_QROUND3: the tolerance mode's radix-8 fast rounds, run by the same kernel shell -- same maps, the barrier bit is bit 28)
    for (size_t o = first; o < out.size() && ((out[o].type & 0xffu) == FUSE_ROUND8 || (out[o].type & 0xffu) == FUSE_QROUND3); o += 1 + (size_t)out[o].mask) {
        Rd r; r.at = o; r.votes.assign(T, 0);
        r.rb[0] = out[o].a & 0xffu; r.rb[1] = (out[o].a >> 8) & 0xffu; r.rb[2] = (out[o].a >> 16) & 0xffu;
        for (size_t k = 1; k <= (size_t)out[o].mask; k++)
            if ((out[o + k].type & 0xffu) == FUSE_PHASE) for (unsigned b = 0; b < T; b++) r.votes[b] += (out[o + k].a >> b) & 1u;
        rounds.push_back(r);
    }
    const unsigned nw = T - 9;
    auto is_reg = [&](size_t r, unsigned b) { return b == rounds[r].rb[0] || b == rounds[r].rb[1] || b == rounds[r].rb[2]; };
    std::vector<unsigned> W;
    for (size_t r = 0; r < rounds.size(); r++) {
        bool keep = tn.fuse_x8_map && r > 0 && W.size() == nw;
        for (unsigned b : W) keep = keep && !is_reg(r, b);
        std::vector<unsigned> free_bits;
        for (unsigned b = 0; b < T; b++) if (!is_reg(r, b)) free_bits.push_back(b);
        if (!keep) {
            if (!tn.fuse_x8_map) W.assign(free_bits.end() - nw, free_bits.end());        // plain ascending order: the top free bits
            else {
                auto life = [&](unsigned b) { size_t q = r; while (q < rounds.size() && !is_reg(q, b)) q++; return q - r; };
                std::vector<unsigned> cand(free_bits);
                std::stable_sort(cand.begin(), cand.end(), [&](unsigned x, unsigned y) {
                    const size_t lx = life(x), ly = life(y);
                    if (lx != ly) return lx > ly;
                    if (rounds[r].votes[x] != rounds[r].votes[y]) return rounds[r].votes[x] > rounds[r].votes[y];
                    return x > y;
                });
                W.assign(cand.begin(), cand.begin() + nw);
                std::sort(W.begin(), W.end());
            }
        }
        unsigned lanes[6], best[6];
        { unsigned k = 0; for (unsigned b : free_bits) if (std::find(W.begin(), W.end(), b) == W.end()) lanes[k++] = b; }
        memcpy(best, lanes, sizeof best);
        if (tn.fuse_x8_map) {
            unsigned perm[6] = {0, 1, 2, 3, 4, 5}, best_cost = ~0u;
            do {
                unsigned cand[6];
                for (unsigned k = 0; k < 6; k++) cand[k] = lanes[perm[k]];
                const unsigned cst = x8_lane_cost(cand);
                if (cst < best_cost) { best_cost = cst; memcpy(best, cand, sizeof best); }
            } while (best_cost > 12u && std::next_permutation(perm, perm + 6));
        }
        uint64_t map = 0;
        for (unsigned k = 0; k < 6; k++) map |= (uint64_t)best[k] << (4 * k);
        for (unsigned k = 0; k < nw; k++) map |= (uint64_t)W[k] << (4 * (6 + k));
        FuseOp &h = out[rounds[r].at];
        memcpy(&h.c, &map, sizeof map);
        const unsigned nobar = (h.type & 0xffu) == FUSE_QROUND3 ? 28u : 25u;
        h.a &= ~(1u << nobar);
        if (keep) h.a |= 1u << nobar;
        if ((h.type & 0xffu) == FUSE_QROUND3) continue;
        // a gate's conditions on the wave bits are wave-uniform: they move from the lane mask into bits 48 .. of the outside mask,
        // which the kernel tests against base | wave number << 48 in the per-item ballot -- a gate this wave skips costs nothing
        for (size_t k = 1; k <= (size_t)h.mask; k++) {
            FuseOp &g = out[rounds[r].at + k];
            if ((g.type & 0xffu) != FUSE_PHASE) continue;
            for (unsigned w = 0; w < nw; w++)
                if ((g.a >> W[w]) & 1u) { g.a &= ~(1u << W[w]); g.mask |= (uint64_t)1 << (48 + w); }
        }
    }
}

// ROUNDS form: group a pass's records into rounds of at most two distinct H bits (the round's register bits);
// inside a round consecutive phases that rotate the same registers form runs (FUSE_PRUN)
static void to_rounds(const Tune &tn, const std::vector<FuseOp> &legacy, unsigned T, std::vector<FuseOp> &out, std::vector<unsigned char> &blob,
                      std::vector<unsigned> *kept_slots = nullptr, unsigned *generic_rounds = nullptr, unsigned maxrb = 2, bool x8 = false)
{
    // x8 (with maxrb = 3): the EXACT walk on 8 amplitudes per thread (k_fused_x8, FUSE_ROUND8): rounds of up to three distinct
    // Hadamard bits in the general form -- H items and phase runs -- with a run's registers named by a pattern instead of rsel
    // maxrb = 3: radix-8 fast rounds (k_fused_q3): up to three steps H(x) [D(x)] per round; a round of any other shape
    // counts as generic (the caller then plans the pass again in the radix-4 form)
    // tolerance mode (kept_slots != nullptr): a merged diagonal arrives as a FUSE_DIAG record followed by the phases it
    // stands for; a round of the shape H(x) [D(x)] [H(y) [D(y)]] keeps its diagonals (FUSE_QROUND, slots renumbered in the
    // order kept), every other round gets the phases back and is emitted exactly as in the bit-exact modes
    struct Item { FuseOp o; std::vector<FuseOp> alts; };
    const size_t out_first = out.size();
    std::vector<Item> cur;
    std::vector<unsigned> rb;
    auto close_round = [&]() {
        if (cur.empty()) { rb.clear(); return; }
        for (unsigned b = T; !x8 && rb.size() < maxrb && b-- > 0;)
            if (std::find(rb.begin(), rb.end(), b) == rb.end()) rb.push_back(b);
        if (x8) {
            // a round with fewer than three Hadamard bits: the free register bits go to tile-local TARGETS of the round's phases
            // (a phase whose mask holds a register bit rotates whole waves; on a lane bit it leaves half of every wave idle) --
            // the targets that most gates of the round name first
            std::vector<unsigned> votes(T, 0);
            for (const Item &it : cur) if (it.o.type == FUSE_PHASE) for (unsigned b = 0; b < T; b++) votes[b] += (it.o.a >> b) & 1u;
            while (rb.size() < 3) {
                int best = -1;
                for (unsigned b = 0; b < T; b++) {
                    if (std::find(rb.begin(), rb.end(), b) != rb.end()) continue;
                    if (best < 0 || votes[b] >= votes[(unsigned)best]) best = (int)b;          // (ties: the higher bit)
                }
                rb.push_back((unsigned)best);
            }
            std::sort(rb.begin(), rb.end());
            bool has_h = false;
            for (const Item &it : cur) has_h |= (it.o.type == FUSE_H);
            FuseOp hdr; memset(&hdr, 0, sizeof hdr);
            hdr.type = FUSE_ROUND8; hdr.a = rb[0] | (rb[1] << 8) | (rb[2] << 16) | ((has_h ? 1u : 0u) << 24);
            out.push_back(hdr);                                  // (the thread map and the barrier bit: x8_assign_maps, once the pass's rounds are all known)
            const size_t hdr_at = out.size() - 1;
            const uint32_t regmask = (1u << rb[0]) | (1u << rb[1]) | (1u << rb[2]);
            size_t run_hdr = (size_t)-1; uint32_t run_pat = 0;
            for (const Item &it : cur) {
                FuseOp o = it.o;
                if (o.type == FUSE_H) {
                    const uint32_t j = o.a == rb[0] ? 0u : o.a == rb[1] ? 1u : 2u;
                    o.a = j; o.type = FUSE_H | ((32u | j) << 8);
                    run_hdr = (size_t)-1; out.push_back(o); continue;
                }
                const uint32_t S = ((o.a >> rb[0]) & 1u) | (((o.a >> rb[1]) & 1u) << 1) | (((o.a >> rb[2]) & 1u) << 2);
                static const uint32_t pat_of[8] = {0, 1, 2, 4, 3, 5, 6, 7};          // 7: all three register bits -- a phase names at most two qubits
                const uint32_t pat = pat_of[S];
                o.a &= ~regmask;
                o.type = FUSE_PHASE | (pat << 8);
                if (run_hdr == (size_t)-1 || pat != run_pat || out[run_hdr].mask >= 63u) {
                    FuseOp rh; memset(&rh, 0, sizeof rh);
                    rh.type = FUSE_PRUN | ((pat | (has_h ? 0u : 16u)) << 8); rh.a = pat;     // bit 4: the run canonicalises its zeros
                    run_hdr = out.size(); run_pat = pat;
                    out.push_back(rh);
                }
                out[run_hdr].mask++;
                out[run_hdr].type += 1u << 16;
                out.push_back(o);
            }
            out[hdr_at].mask = out.size() - 1 - hdr_at;
            cur.clear(); rb.clear();
            return;
        }
        std::sort(rb.begin(), rb.end());
        if (maxrb == 3) {
            uint32_t step[3] = {0, 0, 0};
            unsigned old_slot[3] = {0, 0, 0}, which[3] = {0, 0, 0};
            unsigned ns = 0, nd = 0; bool ok = true;
            auto index_of = [&](unsigned lb) { return lb == rb[0] ? 0u : lb == rb[1] ? 1u : 2u; };
            for (size_t k = 0; k < cur.size() && ok; k++) {
                const FuseOp &o = cur[k].o;
                if (o.type == FUSE_H && ns < 3 && (o.a == rb[0] || o.a == rb[1] || o.a == rb[2])) { which[ns] = index_of(o.a); step[ns] = which[ns]; ns++; }
                else if (o.type == FUSE_DIAG && ns >= 1 && !(step[ns - 1] & 4u) && o.mask == 0) {
                    const unsigned R = which[ns - 1], o1 = (R == 0) ? 1u : 0u, o2 = (R == 2) ? 1u : 2u;
                    uint64_t tloc; memcpy(&tloc, &o.c, sizeof tloc);
                    if ((o.a & 0xffu) != rb[R] + 1) { ok = false; break; }
                    old_slot[ns - 1] = (o.a >> 8) & 0xffu;
                    step[ns - 1] |= 4u | (((o.a >> 16) & 7u) << 16) | ((uint32_t)((tloc >> rb[o1]) & 1u) << 19) | ((uint32_t)((tloc >> rb[o2]) & 1u) << 20);
                    nd++;
                } else ok = false;
            }
            for (unsigned q = 0; q < ns && ok; q++) for (unsigned q2 = q + 1; q2 < ns; q2++) if (which[q] == which[q2]) ok = false;
            if (ok && ns >= 1 && (nd == 0 || (kept_slots && kept_slots->size() + nd <= 16))) {
                bool used[3] = {false, false, false};
                for (unsigned q = 0; q < ns; q++) used[which[q]] = true;
                for (unsigned q = ns; q < 3; q++) for (unsigned w = 0; w < 3; w++) if (!used[w]) { step[q] = w; used[w] = true; break; }   // absent steps: the remaining bits
                for (unsigned q = 0; q < ns; q++)
                    if (step[q] & 4u) { step[q] |= (uint32_t)kept_slots->size() << 8; kept_slots->push_back(old_slot[q]); }
                FuseOp hdr; memset(&hdr, 0, sizeof hdr);
                hdr.type = FUSE_QROUND3; hdr.a = rb[0] | (rb[1] << 8) | (rb[2] << 16) | (ns << 24); hdr.mask = 1;
                out.push_back(hdr);
                FuseOp st; memset(&st, 0, sizeof st);
                st.type = step[0]; st.a = step[1]; st.mask = step[2];
                out.push_back(st);
                cur.clear(); rb.clear();
                return;
            }
            if (generic_rounds) ++*generic_rounds;       // not a radix-8 fast round: the caller re-plans
            cur.clear(); rb.clear();
            return;
        }
        if (kept_slots && tn.fuse_qround) {
            uint32_t step[2] = {0xffffffffu, 0xffffffffu};
            unsigned old_slot[2] = {0, 0};
            unsigned ns = 0, nd = 0; bool ok = true;
            for (size_t k = 0; k < cur.size() && ok; k++) {
                const FuseOp &o = cur[k].o;
                if (o.type == FUSE_H && ns < 2 && (o.a == rb[0] || o.a == rb[1])) step[ns++] = (o.a == rb[1]) ? 1u : 0u;
                else if (o.type == FUSE_DIAG && ns >= 1 && !(step[ns - 1] & 2u) && o.mask == 0) {
                    const unsigned hb = (step[ns - 1] & 1u) ? rb[1] : rb[0], ob = (step[ns - 1] & 1u) ? rb[0] : rb[1];
                    uint64_t tloc; memcpy(&tloc, &o.c, sizeof tloc);
                    if ((o.a & 0xffu) != hb + 1) { ok = false; break; }
                    old_slot[ns - 1] = (o.a >> 8) & 0xffu;
                    step[ns - 1] |= 2u | (((o.a >> 16) & 7u) << 16) | ((uint32_t)((tloc >> ob) & 1u) << 19);
                    nd++;
                } else ok = false;
            }
            if (ok && ns >= 1 && nd >= 1 && kept_slots->size() + nd <= 16) {
                for (unsigned q = 0; q < ns; q++)
                    if (step[q] & 2u) { step[q] |= (uint32_t)kept_slots->size() << 8; kept_slots->push_back(old_slot[q]); }
                FuseOp hdr; memset(&hdr, 0, sizeof hdr);
                hdr.type = FUSE_QROUND; hdr.a = rb[0] | (rb[1] << 8) | (ns << 16); hdr.mask = 1;
                out.push_back(hdr);
                FuseOp st; memset(&st, 0, sizeof st);
                st.type = step[0]; st.a = step[1];
                out.push_back(st);
                cur.clear(); rb.clear();
                return;
            }
        }
        // the general form: diagonals (if any) give way to their phases
        if (generic_rounds) ++*generic_rounds;
        std::vector<FuseOp> flat;
        for (const Item &it : cur) {
            if (it.o.type == FUSE_DIAG) flat.insert(flat.end(), it.alts.begin(), it.alts.end());
            else flat.push_back(it.o);
        }
        FuseOp hdr; memset(&hdr, 0, sizeof hdr);
        bool has_h = false;
        for (const FuseOp &o : flat) has_h |= (o.type == FUSE_H);
        hdr.type = FUSE_ROUND; hdr.a = rb[0] | (rb[1] << 8) | ((has_h ? 1u : 0u) << 16);
        out.push_back(hdr);
        const size_t hdr_at = out.size() - 1;
        const uint32_t regmask = (1u << rb[0]) | (1u << rb[1]);
        size_t run_hdr = (size_t)-1; uint32_t run_rsel = 0;
        for (FuseOp o : flat) {
            if (o.type == FUSE_H) {          // item headers carry everything in their first dword (one scalar load per item)
                o.a = (o.a == rb[0]) ? 0u : 1u; o.type = FUSE_H | ((32u | o.a) << 8);
                run_hdr = (size_t)-1; out.push_back(o); continue;
            }
            const uint32_t mr = o.a & regmask;
            uint32_t rsel = 0;
            for (unsigned q = 0; q < 4; q++) {
                const uint32_t bits = ((q & 1u) << rb[0]) | ((q >> 1) << rb[1]);
                if ((bits & mr) == mr) rsel |= 1u << q;
            }
            o.a &= ~regmask;
            o.type = FUSE_PHASE | (rsel << 8);
            {
                if (run_hdr == (size_t)-1 || rsel != run_rsel || out[run_hdr].mask >= 64u) {
                    FuseOp rh; memset(&rh, 0, sizeof rh);
                    rh.type = FUSE_PRUN | ((rsel | (has_h ? 0u : 16u)) << 8); rh.a = rsel;     // bit 4: the run canonicalises its zeros
                    run_hdr = out.size(); run_rsel = rsel;
                    out.push_back(rh);
                }
                out[run_hdr].mask++;
                out[run_hdr].type += 1u << 16;           // gates of the run, next to the kind and rsel
            }
            out.push_back(o);
        }
        out[hdr_at].mask = out.size() - 1 - hdr_at;          // the round spans everything emitted after its header
        cur.clear(); rb.clear();
    };
    for (size_t li = 0; li < legacy.size(); li++) {
        const FuseOp &o = legacy[li];
        if (o.type == FUSE_CAMODC) {
            close_round();
            // fold a run of >= 2 consecutive permutation-type multiplies with the same modulus into one gather
            FuseCamExtra X0; memcpy(&X0, &o.c, sizeof X0);
            size_t lj = li;
            while (tn.fuse_camruns && lj < legacy.size() && legacy[lj].type == FUSE_CAMODC) {
                FuseCamExtra X; memcpy(&X, &legacy[lj].c, sizeof X);
                if (X.d != 1 || X.C != X0.C || X.C > 256 || (legacy[lj].a & 0xffu) != (o.a & 0xffu)) break;
                lj++;
            }
            const size_t cnt = lj - li;
            if (cnt >= 2 && cnt < 65536) {
                const unsigned cpad = (X0.C + 31u) & ~31u;
                FuseOp hdr; memset(&hdr, 0, sizeof hdr);
                hdr.type = FUSE_CAMRUN; hdr.a = (uint32_t)cnt | (cpad << 16); hdr.mask = blob.size();
                out.push_back(hdr);
                for (size_t q = li; q < lj; q++) {
                    FuseCamExtra X; memcpy(&X, &legacy[q].c, sizeof X);
                    const size_t at = blob.size();
                    blob.resize(at + cpad, 0);
                    for (unsigned x = 0; x < X.C; x++) blob[at + x] = (unsigned char)(((uint64_t)x * X.inv) % X.C);
                    out.push_back(legacy[q]);
                }
                li = lj - 1;
            } else out.push_back(o);
        }
        else if (o.type == FUSE_H) {
            if (std::find(rb.begin(), rb.end(), o.a) == rb.end()) {
                if (rb.size() == maxrb) close_round();
                rb.push_back(o.a);
            }
            cur.push_back(Item{o, {}});
        } else if (o.type == FUSE_DIAG) {
            uint64_t nalt; memcpy(&nalt, &o.s, sizeof nalt);
            Item it{o, std::vector<FuseOp>(legacy.begin() + li + 1, legacy.begin() + li + 1 + nalt)};
            cur.push_back(it);
            li += nalt;
        } else cur.push_back(Item{o, {}});
    }
    close_round();
    if (x8 || (maxrb == 3 && T == 12)) x8_assign_maps(out, out_first, T, tn);
}

// the rounds-only kernel, built for 6, 7 or 8 waves per SIMD; false when the geometry has no rounds form
template <int B, int TT>
static bool launch_rounds_kernel(int occ, int tol_occ, unsigned grid, size_t lds, hipStream_t st, const amp_t *amp, amp_t *amp_out, unsigned n, const FusePass &P,
                                 const FuseOp *d_ops, uint64_t ntiles)
{
    if constexpr ((1u << TT) == 4u * B) {
        const bool cam = P.has_cam != 0;
        if (P.gen) {          // the tiles are generated (circuit front on an unwritten basis state): passes without multiplies only
#define QCX_GEN_LAUNCH(O, S) hipLaunchKernelGGL((k_fused_rounds<B, TT, O, false, S, true>), dim3(grid), dim3(B), lds, st, amp, amp_out, n, P, d_ops, ntiles, d_ops)
            if (P.dg_cnt) { if (P.dg_slim) { if (tol_occ >= 8) QCX_GEN_LAUNCH(8, 2); else QCX_GEN_LAUNCH(6, 2); } else QCX_GEN_LAUNCH(6, 1); }
            else if (occ >= 7) QCX_GEN_LAUNCH(7, 0);
            else QCX_GEN_LAUNCH(6, 0);
#undef QCX_GEN_LAUNCH
            return true;
        }
        if (P.dg_cnt) {       // tolerance mode: the pass holds merged diagonals (K6t); 2 = every round is a fast round (slim kernel)
#define QCX_TOL_LAUNCH(O, C, S) hipLaunchKernelGGL((k_fused_rounds<B, TT, O, C, S>), dim3(grid), dim3(B), lds, st, amp, amp_out, n, P, d_ops, ntiles, d_ops)
            if (P.dg_slim) {
                if (tol_occ >= 8) { if (cam) QCX_TOL_LAUNCH(8, true, 2); else QCX_TOL_LAUNCH(8, false, 2); }
                else { if (cam) QCX_TOL_LAUNCH(6, true, 2); else QCX_TOL_LAUNCH(6, false, 2); }
            } else { if (cam) QCX_TOL_LAUNCH(6, true, 1); else QCX_TOL_LAUNCH(6, false, 1); }
#undef QCX_TOL_LAUNCH
            return true;
        }
#define QCX_ROUNDS_LAUNCH(O, C) hipLaunchKernelGGL((k_fused_rounds<B, TT, O, C>), dim3(grid), dim3(B), lds, st, amp, amp_out, n, P, d_ops, ntiles, d_ops)
        // (round 5: no 8-wave build of the kernels that hold the walk any more.  Its 16 record SGPRs lay beyond the allocator's
        //  budget there -- hipcc: "reserved registers" on the clobber list -- and with 96 SGPRs a CU admits 7 blocks of 256 threads,
        //  not 8, anyway; measured 11.81 (8) / 11.95 (7) / 11.89 (6) ms on the n = 28 inverse QFT, 12.71 / 12.53 on the n = 30
        //  Shor circuit: no difference beyond the noise between boxes.  profiles/r05_occupancy.txt)
        if (occ >= 7) { if (cam) QCX_ROUNDS_LAUNCH(7, true); else QCX_ROUNDS_LAUNCH(7, false); }
        else { if (cam) QCX_ROUNDS_LAUNCH(6, true); else QCX_ROUNDS_LAUNCH(6, false); }
#undef QCX_ROUNDS_LAUNCH
        return true;
    } else {
        (void)occ; (void)tol_occ; (void)grid; (void)lds; (void)st; (void)amp; (void)amp_out; (void)n; (void)P; (void)d_ops; (void)ntiles;
        return false;
    }
}

static bool pass_tables_cover(const FusePass &P, unsigned n);

// does launch_pass hand this pass to k_fused_x8<512, 12> (exact walk on 8 amplitudes, or its tolerance round)?  The only kernel
// with the expanding store of a compact chain's last pass.
static bool pass_is_x8(const FusePass &P, const Tune &tn)
{
    if (P.T != 12 || P.gen || P.has_cam || P.zskip || !tn.fuse_ldsdma) return false;
    return P.dg_slim == 3 || (P.dg_slim == 2 && P.dg_cnt && tn.fuse_x8t);
}

// amp_in / amp_out: the buffer the pass reads / writes (the same for a pass that works in place; a chained pass goes from one of
// the register's two buffers to the other)
static int launch_pass(qcx_register *r, const Tune &tn, const FusePass &P_in, const FuseOp *d_ops, bool nopipe, amp_t *amp_in, amp_t *amp_out)
{
    (void)nopipe;
    FusePass P = P_in;
    if (!P.xp_on && (P.chained != 0) != (amp_in != amp_out)) { set_error("fused pass: chained flag and buffers disagree"); return QCX_UNKNOWN_ERROR; }
    if (P.xp_on && !pass_is_x8(P, tn)) { set_error("expanding store on a pass that is not a k_fused_x8 pass"); return QCX_UNKNOWN_ERROR; }
    const unsigned n = r->n;
    if (!pass_tables_cover(P, n)) { set_error("fused pass: the addressing tables do not cover the %u tile-number bits", n - P.T); return QCX_UNKNOWN_ERROR; }
    const uint64_t ntiles = (uint64_t)1 << (n - P.T);
    const unsigned grid = grid_for(ntiles, 1, tn.fuse_grid_cap);
    const size_t lut_only = P.has_cam ? cam_lut_bytes((unsigned)r->M) : 16;        // scratch of the modular-multiply steps
    size_t lut_bytes = lut_only + (size_t)P.cam_ctl_local[1];                      // + tables of folded multiply runs
    P.xm_off = 0;
    P.dbg = (uint32_t)tn.fuse_dbg & 0xffu;
    {   // which tile a workgroup takes (fuse_stream_tile): memory-bound radix-8 passes (Hadamard sweeps, the tolerance mode's fast
        // rounds) put the XCD number on tile-number bits 3-5 -- one XCD then stores 2^3 neighbouring runs back to back (n = 30
        // sweep -6 %, n = 28 tolerance inverse QFT -1.6 %); the exact phase walk (FP64-bound) measures the same either way and
        // keeps t = b (profiles/r05_streams.txt)
        const bool membound = P.dg_slim == 2 && !P.gen;
        long sl = tn.fuse_streams_log2 >= 0 ? tn.fuse_streams_log2 : (membound ? 3 : 0);
        long pos1 = tn.fuse_streams_pos >= 0 ? tn.fuse_streams_pos : (membound ? 4 : 0);
        sl = std::min<long>(std::min<long>(sl, 15), (long)(n - P.T));
        P.dbg = (P.dbg & 0xfffu) | ((uint32_t)sl << 12) | (((uint32_t)pos1 & 31u) << 16);
    }
    if ((P.dg_cnt || P.dg_slim == 2) && !P.has_cam) { lut_bytes = 16; P.cam_ctl_local[3] = 16; }       // tolerance-mode pass without multiplies: no scratch
    if (P.dg_slim == 3) {                 // the exact walk on 8 amplitudes per thread (k_fused_x8): the tile, then the records' outside-tile masks
        if (P.T < 10 || P.T > 12 || P.has_cam || P.gen > 1 || P.zskip || !tn.fuse_ldsdma) { set_error("radix-8 exact pass: unsupported shape (T = %u)", P.T); return QCX_UNKNOWN_ERROR; }
        P.xm_off = 0;
        size_t lds8 = ((size_t)16 << P.T) + 8 * ((size_t)P.xm_cnt + 66);
        if (P.gen) { P.gen_lds_off = (uint32_t)((8 * ((size_t)P.xm_cnt + 66) + 15) & ~(size_t)15); lds8 = ((size_t)16 << P.T) + P.gen_lds_off + 2 * 1024 * sizeof(unsigned short); }
        const unsigned grid8 = grid_for(ntiles, 1, tn.fuse_x8_cap);
#define QCX_X8_LAUNCH(B, TTv) do { \
        if (P.gen) hipLaunchKernelGGL((k_fused_x8<B, TTv, true>), dim3(grid8), dim3(B), lds8, r->stream, amp_in, amp_out, n, P, d_ops, ntiles, d_ops); \
        else hipLaunchKernelGGL((k_fused_x8<B, TTv, false>), dim3(grid8), dim3(B), lds8, r->stream, amp_in, amp_out, n, P, d_ops, ntiles, d_ops); } while (0)
        switch (P.T) {
        case 12: QCX_X8_LAUNCH(512, 12); break;
        case 11: QCX_X8_LAUNCH(256, 11); break;
        default: QCX_X8_LAUNCH(128, 10); break;
        }
#undef QCX_X8_LAUNCH
        HIP_TRY(hipGetLastError());
        return QCX_NO_ERROR;
    }
    if (P.xm_cnt && !P.dg_slim) {                                                  // + the records' outside-tile masks (phase runs)
        P.xm_off = (uint32_t)((lut_bytes + 7) & ~(size_t)7);
        lut_bytes = P.xm_off + 8 * ((size_t)P.xm_cnt + 66);          // padded: lanes look up to 64 entries past a run
    }
    P.dg_lds_off = 0;
    if (P.dg_cnt) {                                                                // + tolerance mode: E_out slots and G tables of the merged diagonals
        P.dg_lds_off = (uint32_t)((lut_bytes + 15) & ~(size_t)15);
        lut_bytes = P.dg_lds_off + 16 * (size_t)P.dg_cnt * 49;
    }
    P.gen_lds_off = 0;
    if (P.gen) {                                                                   // + the generated fill's slot tables (2 x 1024 entries of 2 B; 2 x 512 for the radix-8 kernel, whose two workgroups per CU have no room for more)
        P.gen_lds_off = (uint32_t)((lut_bytes + 15) & ~(size_t)15);
        lut_bytes = P.gen_lds_off + 2 * (P.dg_slim == 2 ? 512 : 1024) * sizeof(unsigned short);
    }
    if (P.gen >= 2) {                     // the generated first pass by columns (K6g; 3: of a compact chain): populated columns + the records' masks in LDS
        P.xm_off = 0;
        size_t lds_cols = (size_t)P.zpad * QCX_COL_STRIDE * sizeof(amp_t) + 8 * ((size_t)P.xm_cnt + 66);
        if (P.dg_cnt) {                                            // tolerance mode: E_out slots and G tables of the merged diagonals
            P.dg_lds_off = (uint32_t)((8 * ((size_t)P.xm_cnt + 66) + 15) & ~(size_t)15);
            lds_cols = (size_t)P.zpad * QCX_COL_STRIDE * sizeof(amp_t) + P.dg_lds_off + 16 * (size_t)P.dg_cnt * 49;
        }
        P.gen_lds_off = (uint32_t)((lds_cols - (size_t)P.zpad * QCX_COL_STRIDE * sizeof(amp_t) + 15) & ~(size_t)15);   // the residue -> column table of a compact chain
        lds_cols = (size_t)P.zpad * QCX_COL_STRIDE * sizeof(amp_t) + P.gen_lds_off + (P.gen == 3 ? P.gen_tab_bytes : 0);
        // (four waves; one wave per column -- six for the orbit of N = 21 -- is slower: 5.3 against 3.8 ms at n = 30, the extra waves idle through generation and store)
        // ... while a launch that does not fill the chip (<= 1024 tiles: n <= 22 behind a six-residue front) is a latency chain per tile,
        // which more waves shorten: attempts at n = 16 .. 22 take 88-127 us with eight waves against 97-138 with four
        // (profiles/r05_attempts_cols_waves.txt).  fuse_cols_waves = 0: chosen here.
        const unsigned waves = tn.fuse_cols_waves <= 0 ? (ntiles <= 1024 ? 8u : 4u) : (unsigned)std::min<long>(8, std::max<long>(4, tn.fuse_cols_waves));
        const unsigned gridc = grid_for(ntiles, 1, tn.fuse_cols_cap);
        if (P.dg_cnt) hipLaunchKernelGGL((k_gen_cols<6, true>), dim3(gridc), dim3(64 * waves), lds_cols, r->stream, amp_out, n, P, d_ops, ntiles, d_ops);
        else hipLaunchKernelGGL((k_gen_cols<6, false>), dim3(gridc), dim3(64 * waves), lds_cols, r->stream, amp_out, n, P, d_ops, ntiles, d_ops);
        HIP_TRY(hipGetLastError());
        return QCX_NO_ERROR;
    }
    const size_t lds = ((size_t)16 << P.T) + lut_bytes;
    // 4 amplitudes per thread (all loads of a tile in flight at once, few registers): block = 2^T / 4
#define QCX_FUSE_LAUNCH(B, TTv) do { \
        if (P.cam_ctl_local[0] && tn.fuse_ldsdma && tn.fuse_rounds_occ >= 6 && launch_rounds_kernel<B, TTv>((int)tn.fuse_rounds_occ, (int)tn.fuse_tol_occ, grid, lds, r->stream, amp_in, amp_out, n, P, d_ops, ntiles)) { \
        } else if (P.chained) { set_error("chained pass without a rounds kernel"); return QCX_UNKNOWN_ERROR; \
        } else if (tn.fuse_ldsdma) hipLaunchKernelGGL((k_fused<B, TTv, true>), dim3(grid), dim3(B), lds, r->stream, amp_in, n, P, d_ops, ntiles, d_ops); \
        else hipLaunchKernelGGL((k_fused<B, TTv, false>), dim3(grid), dim3(B), lds, r->stream, amp_in, n, P, d_ops, ntiles, d_ops); } while (0)
    if (P.dg_slim == 2) {                 // tolerance mode, radix-8 fast rounds only
        if (P.T != 12) { set_error("radix-8 pass on a tile of 2^%u amplitudes", P.T); return QCX_UNKNOWN_ERROR; }
        if (P.dg_cnt && !P.gen && tn.fuse_x8t) {              // round 5: the hand-written round on the k_fused_x8 shell (K6x-t)
            P.dg_lds_off = 0;
            const size_t lds8 = ((size_t)16 << P.T) + 16 * (size_t)P.dg_cnt * 49;
            hipLaunchKernelGGL((k_fused_x8<512, 12, false, true>), dim3(grid_for(ntiles, 1, tn.fuse_x8_cap)), dim3(512), lds8, r->stream, amp_in, amp_out, n, P, d_ops, ntiles, d_ops);
            HIP_TRY(hipGetLastError());
            return QCX_NO_ERROR;
        }
        // (a few workgroups per CU that walk the tiles: 2048 workgroups 6.3 ms, 24576 7.0 ms, one per tile 8.3 ms at n = 28)
        // (the Hadamard-only exact form likes more of them: n = 30 sweep 23.5 ms with 8192, 23.9 with 3072)
        const unsigned grid3 = grid_for(ntiles, 1, P.dg_cnt ? tn.fuse_q3_cap : tn.fuse_q3_cap_exact);
        if (P.gen) {
            if (P.dg_cnt) hipLaunchKernelGGL((k_fused_q3<512, 12, 4, false, true>), dim3(grid3), dim3(512), lds, r->stream, amp_in, amp_out, n, P, d_ops, ntiles);
            else hipLaunchKernelGGL((k_fused_q3<512, 12, 4, true, true>), dim3(grid3), dim3(512), lds, r->stream, amp_in, amp_out, n, P, d_ops, ntiles);
        }
        else if (P.dg_cnt) hipLaunchKernelGGL((k_fused_q3<512, 12, 4, false>), dim3(grid3), dim3(512), lds, r->stream, amp_in, amp_out, n, P, d_ops, ntiles);
        else hipLaunchKernelGGL((k_fused_q3<512, 12, 4, true>), dim3(grid3), dim3(512), lds, r->stream, amp_in, amp_out, n, P, d_ops, ntiles);
        HIP_TRY(hipGetLastError());
        return QCX_NO_ERROR;
    }
    switch (P.T) {
    case 12: QCX_FUSE_LAUNCH(1024, 12); break;
    case 11: QCX_FUSE_LAUNCH(512, 11); break;
    case 10: QCX_FUSE_LAUNCH(256, 10); break;
    case 9:  QCX_FUSE_LAUNCH(256, 9); break;
    // registers of the reference's own sizes (n = 5 ... 8: the whole state is one tile): one wave, everything known at compile time
    // (the generic kernel below took 60 us for the 13 gates of the C = 15, L = 3, M = 4 circuit; these take ~10)
    case 8:  hipLaunchKernelGGL((k_fused<64, 8, false>), dim3(grid), dim3(64), lds, r->stream, amp_in, n, P, d_ops, ntiles, d_ops); break;
    case 7:  hipLaunchKernelGGL((k_fused<64, 7, false>), dim3(grid), dim3(64), lds, r->stream, amp_in, n, P, d_ops, ntiles, d_ops); break;
    case 6:  hipLaunchKernelGGL((k_fused<64, 6, false>), dim3(grid), dim3(64), lds, r->stream, amp_in, n, P, d_ops, ntiles, d_ops); break;
    case 5:  hipLaunchKernelGGL((k_fused<64, 5, false>), dim3(grid), dim3(64), lds, r->stream, amp_in, n, P, d_ops, ntiles, d_ops); break;
    default: hipLaunchKernelGGL((k_fused<256, 0, false>), dim3(grid), dim3(256), lds, r->stream, amp_in, n, P, d_ops, ntiles, d_ops); break;
    }
#undef QCX_FUSE_LAUNCH
    HIP_TRY(hipGetLastError());
    return QCX_NO_ERROR;
}

// ---- general tile addressing of a pass (FusePass::in_pos ... seg_lg; round 4) -----------------------------------------------
// run-length segments that deposit the bits of a tile number: bit k of the number goes to index bit pos[k]
static bool make_segments(const std::vector<unsigned> &pos, FuseSeg *seg, uint8_t *nseg)
{
    unsigned cnt = 0;
    for (size_t k = 0; k < pos.size();) {
        size_t e = k + 1;
        while (e < pos.size() && pos[e] == pos[e - 1] + 1) e++;
        if (cnt == QCX_MAX_SEG) return false;
        seg[cnt].src = (uint8_t)k; seg[cnt].dst = (uint8_t)pos[k]; seg[cnt].len = (uint8_t)(e - k); seg[cnt].pad = 0;
        cnt++;
        k = e;
    }
    *nseg = (uint8_t)cnt;
    return true;
}

// the tables of a pass that works IN PLACE on the identity layout: tile-local bit j = index bit tl[j] (ascending)
static bool pass_tables_inplace(FusePass &P, unsigned n, const std::vector<unsigned> &tl)
{
    P.chained = 0;
    std::vector<bool> in_tile(n, false);
    for (size_t j = 0; j < tl.size(); j++) { P.in_pos[j] = (uint8_t)tl[j]; P.st_loc[j] = (uint8_t)j; P.st_pos[j] = (uint8_t)tl[j]; in_tile[tl[j]] = true; }
    std::vector<unsigned> fr;
    for (unsigned b = 0; b < n; b++) if (!in_tile[b]) fr.push_back(b);
    if (!make_segments(fr, P.seg_in, &P.nseg_in)) return false;
    P.nseg_out = P.nseg_lg = 0;
    return true;
}

// the tables of a CHAINED pass: reads the layout lin (lin[q] = physical index bit of qubit q in the input buffer), writes the
// layout lout into the other buffer.  tl = the tile's qubits in local order: ascending input position.
static bool pass_tables_chained(FusePass &P, unsigned n, const std::vector<unsigned> &tl, const std::vector<unsigned> &lin, const std::vector<unsigned> &lout, bool by_out)
{
    P.chained = 1;
    const size_t T = tl.size();
    std::vector<bool> in_tile(n, false);
    for (size_t j = 0; j < T; j++) { P.in_pos[j] = (uint8_t)lin[tl[j]]; in_tile[tl[j]] = true; if (j && lin[tl[j]] <= lin[tl[j - 1]]) return false; }
    // store order: the tile's bits by ascending OUTPUT position
    std::vector<unsigned> ord(T);
    for (size_t j = 0; j < T; j++) ord[j] = (unsigned)j;
    std::sort(ord.begin(), ord.end(), [&](unsigned x, unsigned y) { return lout[tl[x]] < lout[tl[y]]; });
    for (size_t j = 0; j < T; j++) { P.st_loc[j] = (uint8_t)ord[j]; P.st_pos[j] = (uint8_t)lout[tl[ord[j]]]; }
    // the tile number's bits = the qubits outside the tile.  by_out: by ascending OUTPUT position -- tiles that run at the same
    // time (consecutive numbers) then store NEIGHBOURING runs: 2^(T - c) consecutive tiles fill the same 2^(T - c) blocks of
    // the output completely (a tile is read as one contiguous block wherever it lies).  Otherwise by ascending INPUT position
    // (neighbouring tiles read neighbouring blocks).  Measured at n = 28 / 30 (tools/experiments/probe_chain.py): memory-bound passes
    // (Hadamard sweeps, tolerance mode) gain 1-4 % from the output order, the FP64-bound exact phase passes lose 5 % with it.
    std::vector<unsigned> fq;
    for (unsigned q = 0; q < n; q++) if (!in_tile[q]) fq.push_back(q);
    std::sort(fq.begin(), fq.end(), [&](unsigned x, unsigned y) { return by_out ? lout[x] < lout[y] : lin[x] < lin[y]; });
    std::vector<unsigned> pin, pout, plg;
    for (unsigned q : fq) { pin.push_back(lin[q]); pout.push_back(lout[q]); plg.push_back(q); }
    return make_segments(pin, P.seg_in, &P.nseg_in) && make_segments(pout, P.seg_out, &P.nseg_out) && make_segments(plg, P.seg_lg, &P.nseg_lg);
}

// do the segment lists of a pass deposit ALL n - T bits of the tile number?  (a list left short -- or stale from another
// plan -- would send every tile to base 0: launch_pass refuses such a pass)
static bool pass_tables_cover(const FusePass &P, unsigned n)
{
    auto total = [](const FuseSeg *seg, unsigned cnt) { unsigned t = 0; for (unsigned k = 0; k < cnt && k < QCX_MAX_SEG; k++) t += seg[k].len; return t; };
    if (P.T > n || P.nseg_in > QCX_MAX_SEG || P.nseg_out > QCX_MAX_SEG || P.nseg_lg > QCX_MAX_SEG) return false;
    if (total(P.seg_in, P.nseg_in) != n - P.T) return false;
    if (!P.chained) return true;
    return total(P.seg_out, P.nseg_out) == n - P.T && total(P.seg_lg, P.nseg_lg) == n - P.T;
}

// The planner: cut a gate list into actions (fused passes and stand-alone gates) and emit every pass's records.
// Pure host code (no HIP call): qcx_fusion_plan exposes it so that the CPU-only tests can check the records against
// the oracle with an emulator of the pass kernels (tests/fuse_emulator.py).
// tol: tolerance mode (qcx_set_fusion(reg, 2)).  Actions always refer to the ORIGINAL gate list (first_gate / ngates /
// gate), whatever was merged.
// chain: runs of consecutive rounds-form passes may be CHAINED through the register's second buffer (see FusePass): every
// pass but the first then reads whole contiguous tiles, only its stores are gathered, and the last one stores the identity
// layout again -- the same records, the same arithmetic, other addresses.
struct PassShape {
    size_t first, last;                 // gates [first, last) of the (merged) list
    unsigned c;
    std::vector<unsigned> hbits;        // the tile's qubits above the low c, ascending
    std::vector<unsigned> tl;           // the tile's qubits in local order
    bool want_q3;
    bool want_x8;                       // the exact walk on 8 amplitudes per thread (k_fused_x8): phase-dominated bit-exact passes
    bool cols;                          // the generated first pass by columns (K6g): radix-4 rounds; merged diagonals expanded unless cols_tol
    bool cols_tol;
    size_t n_h, n_ph, n_other;
    int nopipe;
};

static void fuse_plan(const qcx_register *r, const Tune &tn, const std::vector<QGate> &gates_arg, std::vector<FuseAction> &acts, std::vector<FuseOp> &all_ops,
                      bool tol = false, bool chain = false, int first_cols = 0)
{
    // first_cols (1; 2: it may keep merged diagonals): the list sits behind a circuit front that the first pass generates by columns (K6g, k_gen_cols): that pass takes a
    // tile of 2^12 amplitudes = the four lowest M-register bits x 8 hot bits
    // an M register beyond the LDS tile (M > 12): its modular multiplies never join a tile pass -- stand-alone (K3b), like the table form
    std::vector<QGate> big;
    if ((unsigned)r->M > 12) { big = gates_arg; for (QGate &g : big) if (g.type == FUSE_CAMODC) g.type = 99; }
    const std::vector<QGate> &gates_in = (unsigned)r->M > 12 ? big : gates_arg;
    std::vector<QGate> merged;
    std::vector<DiagSpec> specs;
    std::vector<size_t> ofirst, ocnt;                  // per (merged) gate: its run in the original list
    if (tol) merge_diagonals(gates_in, merged, specs, ofirst, ocnt);
    const std::vector<QGate> &gates = tol ? merged : gates_in;
    if (!tol) { ofirst.resize(gates.size()); ocnt.assign(gates.size(), 1); for (size_t k = 0; k < gates.size(); k++) ofirst[k] = k; }
    std::vector<int> shape_of;                         // per action: index into shapes, or -1 (stand-alone gate)
    std::vector<PassShape> shapes;
    auto standalone = [&](size_t k) {                  // gate k (merged numbering) as stand-alone kernel launches
        for (size_t j = 0; j < ocnt[k]; j++) { FuseAction a; memset(&a, 0, sizeof a); a.gate = ofirst[k] + j; acts.push_back(a); shape_of.push_back(-1); }
    };
    const unsigned n = r->n;
    unsigned T = (unsigned)tn.fuse_T, c_def = (unsigned)tn.fuse_c;
    if (T > 12) T = 12;
    if (T < 1) T = 1;
    if (T > n) T = n;
    if (c_def > T) c_def = T;

    // records and FusePass fields of one pass; 0 = done, 1 = the radix-8 plan did not work out (plan the pass again, radix 4)
    auto emit = [&](const PassShape &sh, FuseAction &act) -> int {
        const size_t first = sh.first, i = sh.last;
        const std::vector<unsigned> &tl = sh.tl;
        act.fused = 1;
        act.nopipe = sh.nopipe;
        for (size_t j = 0; j < tl.size() && j < 16; j++) act.tl[j] = (uint8_t)tl[j];
        act.first_gate = ofirst[first]; act.ngates = ofirst[i - 1] + ocnt[i - 1] - ofirst[first];
        act.P.has_cam = sh.n_other ? 1u : 0u;
        act.P.c = sh.c; act.P.nh = (uint32_t)sh.hbits.size(); act.P.T = sh.c + act.P.nh;
        // (the table area sits behind the modular-multiply scratch; a pass without multiplies carries none: with M >= 12 its 8 KiB
        //  pushed the 2^10-tile phase passes over the LDS budget of the rounds form and onto the general kernel: 24 ms against 7)
        act.P.cam_ctl_local[3] = (int32_t)(sh.n_other ? cam_lut_bytes((unsigned)r->M) : 16);
        for (unsigned j = 0; j < act.P.nh; j++) act.P.hbit[j] = (uint8_t)sh.hbits[j];
        std::vector<FuseOp> legacy;
        bool rounds = tn.fuse_rounds && act.P.T >= 10 && (act.P.T <= 12 || (act.P.T == 13 && sh.want_q3 && !tol && sh.n_ph == 0));
        if (sh.want_q3 && !(rounds && (act.P.T == 12 || act.P.T == 13) && sh.n_other == 0)) return 1;
        if (sh.want_x8 && !(rounds && sh.n_other == 0 && !tol)) return 1;
        // (tolerance mode: merged diagonals exist in the rounds form only, at most 16 per pass -- their tables live in LDS;
        // otherwise the pass gets the plain phases they were merged from)
        std::vector<unsigned> pass_diags;
        size_t n_diag = 0;
        for (size_t k = first; k < i; k++) n_diag += gates[k].type == FUSE_DIAG;
        bool keep_diags = tol && rounds && n_diag > 0 && n_diag <= 255 && (!sh.cols || sh.cols_tol);
        build_pass_ops(r, gates, first, i, tl, legacy, &specs, &gates_in, !keep_diags, &pass_diags);
        act.op_off = all_ops.size();
        if (rounds) {
            // ROUNDS form: phases run as "phase runs", which need the records' outside-tile masks in LDS next to the
            // tiles (8 B per record); when that would cost a resident workgroup the pass uses the plain gate list
            std::vector<unsigned char> blob;
            std::vector<unsigned> kept;                  // old slot numbers of the diagonals that stayed merged, in new-slot order
            unsigned generic_rounds = 0;
            to_rounds(tn, legacy, act.P.T, all_ops, blob, keep_diags ? &kept : nullptr, &generic_rounds, (sh.want_q3 || sh.want_x8) ? 3u : 2u, sh.want_x8);
            if (sh.want_q3 && (generic_rounds || (n_diag > 0 && (!keep_diags || kept.empty())) || (n_diag == 0 && sh.n_ph + sh.n_other > 0))) {       // not all radix-8 fast rounds: plan this pass again, radix 4
                all_ops.resize(act.op_off); return 1;
            }
            act.P.dg_slim = sh.want_x8 ? 3u : sh.want_q3 ? 2u : (keep_diags && !kept.empty() && generic_rounds == 0) ? 1u : 0u;     // every round is a fast round (2: radix 8); 3: the exact walk on 8 amplitudes per thread
            act.P.tol_scale = 1.0;
            if (sh.want_q3) for (size_t k = 0; k < sh.n_h; k++) act.P.tol_scale *= QCX_SQRT1_2;
            if (keep_diags) {
                std::vector<unsigned> kd;
                for (unsigned os : kept) kd.push_back(pass_diags[os]);
                pass_diags.swap(kd);
                keep_diags = !pass_diags.empty();
            }
            const size_t nrec = all_ops.size() - act.op_off;
            size_t lds = 2 * ((size_t)16 << act.P.T) + (size_t)act.P.cam_ctl_local[3] + blob.size() + 64 + 8 * (nrec + 66);
            size_t limit = (size_t)160 * 1024 / (act.P.T >= 12 ? 1 : act.P.T == 11 ? 2 : 4);
            if (act.P.T == 13) lds = ((size_t)16 << 13) + 64 + 8 * (nrec + 66);          // one tile per CU
            if (keep_diags) {            // tolerance mode: what the workgroup really needs; two of them must fit a CU
                lds = ((size_t)16 << act.P.T) + (sh.n_other ? (size_t)act.P.cam_ctl_local[3] : 16) + blob.size() + 64 + 8 * (nrec + 66) + 16 * 49 * pass_diags.size();
                limit = (size_t)80 * 1024;
            }
            if (sh.want_x8) {            // one tile + the records' masks; two workgroups of 2^12 (four of 2^11, eight of 2^10) per CU
                lds = ((size_t)16 << act.P.T) + 8 * (nrec + 66);
                limit = (size_t)160 * 1024 / (act.P.T == 12 ? 2 : act.P.T == 11 ? 4 : 8);
            }
            if (lds > limit) {
                all_ops.resize(act.op_off); rounds = false;
                if (sh.want_q3 || sh.want_x8) return 1;                                       // radix-8 records mean nothing to the plain gate list: plan again
                act.P.dg_slim = 0; act.P.tol_scale = 1.0;                       // (launch_pass looks at dg_slim first)
                if (keep_diags) {        // the plain gate list has no diagonal interpreter: back to the phases
                    legacy.clear(); pass_diags.clear(); keep_diags = false;
                    build_pass_ops(r, gates, first, i, tl, legacy, &specs, &gates_in, true, &pass_diags);
                }
            }
            else {
                act.P.xm_cnt = (uint32_t)nrec;
                act.P.cam_ctl_local[0] = 1;
                if (!blob.empty()) {             // the tables ride behind the pass's records, padded to whole records
                    blob.resize((blob.size() + sizeof(FuseOp) - 1) / sizeof(FuseOp) * sizeof(FuseOp), 0);
                    act.P.cam_ctl_local[1] = (int32_t)blob.size();
                    act.P.cam_ctl_local[2] = (int32_t)(all_ops.size() - act.op_off);
                    const size_t at = all_ops.size();
                    all_ops.resize(at + blob.size() / sizeof(FuseOp));
                    memcpy(&all_ops[at], blob.data(), blob.size());
                }
                if (keep_diags) {                // the diagonals' table area rides behind everything else of the pass
                    std::vector<double> area;
                    diag_tables(n, tl, specs, pass_diags, area);
                    act.P.dg_cnt = (uint32_t)pass_diags.size();
                    act.P.dg_rec_off = (uint32_t)(all_ops.size() - act.op_off);
                    const size_t at = all_ops.size();
                    all_ops.resize(at + area.size() * sizeof(double) / sizeof(FuseOp));
                    memcpy(&all_ops[at], area.data(), area.size() * sizeof(double));
                }
                act.P.nops = (uint32_t)nrec;
            }
        }
        if (!rounds) { all_ops.insert(all_ops.end(), legacy.begin(), legacy.end()); act.P.nops = (uint32_t)legacy.size(); }
        act.op_cnt = all_ops.size() - act.op_off;
        static const bool dump = getenv("QCX_FUSE_DUMP") != nullptr;      // planner diagnostics (tools/experiments/probe_fuse3.py)
        if (dump) {
            unsigned nround = 0, nh = 0, nrun = 0, nsingle = 0, ncam = 0, run_gates[16] = {0}, ext = 0, loc = 0;
            for (size_t q = act.op_off; q < act.op_off + act.P.nops; q++) {
                const FuseOp &o = all_ops[q];
                switch (o.type & 0xffu) {
                case FUSE_ROUND: nround++; break;
                case FUSE_H: nh++; break;
                case FUSE_PRUN: nrun++; run_gates[o.a & 15u] += (unsigned)o.mask; break;
                case FUSE_PHASE: nsingle++; if (o.mask) ext++; if (o.a) loc++; break;
                default: ncam++; break;
                }
            }
            fprintf(stderr, "[qcx fuse] pass T=%u c=%u tile=", act.P.T, act.P.c);
            for (unsigned q : tl) fprintf(stderr, "%u,", q);
            fprintf(stderr, " gates=%zu records=%u rounds=%u H=%u runs=%u phases=%u (ext-ctl %u, lane-ctl %u) other=%u run gates by rsel:",
                    act.ngates, act.P.nops, nround, nh, nrun, nsingle, ext, loc, ncam);
            for (unsigned q = 1; q < 16; q++) if (run_gates[q]) fprintf(stderr, " %x:%u", q, run_gates[q]);
            fprintf(stderr, "\n");
        }
        return 0;
    };

    std::vector<unsigned> need;
    size_t i = 0;
    bool q3_refused = false;               // the radix-8 plan of the pass starting at i did not work out: plan it the usual way
    while (i < gates.size()) {
        FuseAction act;
        memset(&act, 0, sizeof act);
        const bool q3_allowed = !q3_refused;
        q3_refused = false;
        if (gates[i].type == 99) { standalone(i++); continue; }   // table-form modular multiply
        // ---- grow one pass: a gate joins while the bits it needs inside the tile still fit ----------------
        std::vector<unsigned> hbits;
        const size_t first = i;
        size_t n_h = 0, n_ph = 0, n_other = 0;
        auto grow = [&](unsigned cc, unsigned bud) {
            hbits.clear(); i = first; n_h = n_ph = n_other = 0;
            while (i < gates.size() && gates[i].type != 99) {
                const QGate &g = gates[i];
                need.clear();
                if (g.type == FUSE_H) { if (g.q >= cc) need.push_back(g.q); }
                else if (g.type == FUSE_CAMODC) { for (unsigned b = cc; b < (unsigned)r->M; b++) need.push_back(b); }
                std::vector<unsigned> merged = hbits;
                for (unsigned b : need) if (std::find(merged.begin(), merged.end(), b) == merged.end()) merged.push_back(b);
                if (merged.size() > bud) break;
                hbits.swap(merged);
                if (g.type == FUSE_H) n_h++; else if (g.type == FUSE_PHASE || g.type == FUSE_DIAG) n_ph++; else n_other++;
                i++;
            }
        };
        // A tail of nothing but Hadamards (the H sweep) is bound by the tile passes' memory pipeline: what counts is the
        // NUMBER of passes.  Tiles of 2^12 amplitudes with 2^3-amplitude runs hold 9 hot bits instead of 7 -- a 30-qubit
        // sweep in 3 passes instead of 4 -- at 1.23x the time per pass (128-B runs, 64 KiB tiles: measured 8.2 vs 6.7 ms at
        // n = 30), so that geometry is taken when it saves enough passes.  Passes with phases or multiplies keep theirs.
        unsigned Tcur = T, ccur = c_def;
        bool want_q3 = false;
        const bool cols_pass = first_cols && first == 0 && n >= 14;
        {
            const unsigned Ta = (unsigned)tn.fuse_hsweep_T, ca = (unsigned)tn.fuse_hsweep_c;
            bool tail_h = Ta >= 10 && Ta <= 12 && Ta <= n && ca <= Ta && tn.fuse_rounds;       // (2^13-amplitude tiles, one 128-KiB workgroup per CU: built in round 5, slower, removed -- DESIGN.md s4.6)
            for (size_t k = first; tail_h && k < gates.size(); k++) tail_h = gates[k].type == FUSE_H;
            if (tail_h) {
                auto passes = [&](unsigned TT, unsigned cc) {
                    unsigned np = 0;
                    for (size_t k = first; k < gates.size(); np++) {
                        std::vector<unsigned> hb;
                        const size_t k0 = k;
                        for (; k < gates.size(); k++) {
                            const unsigned q = gates[k].q;
                            if (q >= cc && std::find(hb.begin(), hb.end(), q) == hb.end()) { if (hb.size() == TT - cc) break; hb.push_back(q); }
                        }
                        if (k == k0) k++;                                    // (cannot happen: one Hadamard always fits)
                    }
                    return np;
                };
                if (passes(Ta, ca) * 123u < passes(T, c_def) * 100u) {
                    Tcur = Ta; ccur = ca;
                    if (Ta == 12 && q3_allowed && tn.fuse_q3) want_q3 = true;      // radix-8 rounds, exact butterflies (k_fused_q3<.., EXACT>)
                }
            }
        }
        // Tolerance mode: passes with merged diagonals overlap their arithmetic better on the smaller tile (2^10 amplitudes,
        // 256-thread workgroups: n = 28 inverse QFT 7.4 against 8.0 ms) -- taken when it does not cost a pass, estimated from
        // the distinct Hadamard targets above the tile's low bits in the rest of the queue.
        if (tol && !cols_pass && tn.fuse_tol_T >= 10 && (unsigned)tn.fuse_tol_T < Tcur && (unsigned)tn.fuse_tol_T <= n && tn.fuse_rounds) {
            bool pure = true; uint64_t hot = 0;
            for (size_t k = first; k < gates.size(); k++) {
                if (gates[k].type == FUSE_H) { if (gates[k].q >= ccur) hot |= (uint64_t)1 << gates[k].q; }
                else if (gates[k].type != FUSE_DIAG && gates[k].type != FUSE_PHASE) { pure = false; break; }
            }
            const unsigned hb = (unsigned)__builtin_popcountll(hot), Ts = (unsigned)tn.fuse_tol_T;
            // (a tile needs at least one hot bit: with fuse_c == the tile bits the estimate would divide by zero)
            auto passes_at = [&](unsigned TT) { return TT > ccur ? (hb + (TT - ccur) - 1) / (TT - ccur) : 0xffffffffu; };
            // radix-8 rounds on 2^12 tiles (k_fused_q3) when they save a pass against both radix-4 geometries -- with 2^3-amplitude
            // runs (9 hot bits per pass) when that saves yet another one: the 25 Hadamards of the n = 30 Shor circuit's inverse
            // QFT take 3 passes instead of 4 (runs of 128 B cost a pass ~10 %, a pass less saves 25 %)
            unsigned hb3 = 0;
            for (unsigned q = 3; q < ccur && pure; q++) {           // Hadamard targets in [3, c) would become hot bits with c = 3
                bool tgt = false;
                for (size_t k = first; k < gates.size() && !tgt; k++) tgt = gates[k].type == FUSE_H && gates[k].q == q;
                hb3 += tgt;
            }
            const unsigned p12 = passes_at(12), p12c3 = ccur > 3 ? (hb + hb3 + 8) / 9 : p12;
            const unsigned others = std::min(passes_at(Ts), passes_at(Tcur));
            if (pure && hb && q3_allowed && tn.fuse_q3 && n >= 12 && ccur <= 4 && std::min(p12, p12c3) < others) {
                if (p12c3 < p12 && tn.fuse_q3_c3) ccur = 3;
                Tcur = 12; want_q3 = true;
            }
            else if (pure && hb && passes_at(Ts) == passes_at(Tcur)) Tcur = Ts;
        }
        unsigned c = ccur, budget = Tcur - ccur;
        grow(c, budget);
        // A pass dominated by controlled phases is bound by FP64 issue and latency, not by HBM: it runs better on
        // smaller tiles (256-thread workgroups: smaller barrier domains, more of them resident), at the price of
        // fewer hot bits per pass; and never on the pipelined kernel.
        const unsigned colb = std::min(4u, (unsigned)r->M);       // column bits of the by-columns pass: the lowest M-register bits (a compact chain's virtual register: all of them)
        const unsigned Tp = cols_pass ? colb + 8u : (unsigned)tn.fuse_T_phase;
        bool want_x8 = false;
        // (the walk on 8 amplitudes takes a pass as soon as it holds fuse_x8_ratio phases per Hadamard -- far fewer than what makes
        //  a radix-4 pass "phase-dominated": the tail of an inverse QFT, 9 Hadamards with 36 phases, is then ONE pass of 8 hot bits
        //  instead of two of 7)
        // (registers of a tile or two keep the radix-4 kernels: a lone 512-thread workgroup is slower there -- the n = 12 attempt 43 -> 50 us)
        const bool x8_ok = !tol && !cols_pass && q3_allowed && tn.fuse_x8 && tn.fuse_x8_T >= 10 && tn.fuse_x8_T <= 12 && (unsigned)tn.fuse_x8_T + (unsigned)tn.fuse_x8_min_tiles_log2 <= n &&
                           tn.fuse_ldsdma && tn.fuse_rounds_occ >= 6 && n_ph >= 1 && n_ph >= (size_t)tn.fuse_x8_ratio * std::max<size_t>(n_h, 1);
        if (!want_q3 && Tp >= 9 && Tp <= 12 && Tp <= n && tn.fuse_rounds && n_other == 0 &&
            (cols_pass || x8_ok || n_ph >= (size_t)tn.fuse_phase_ratio * std::max<size_t>(n_h, 1))) {
            // bit-exact phase passes (round 5): the walk on 8 amplitudes per thread (k_fused_x8) -- a tile of 2^12 amplitudes on 512
            // threads holds 8 hot bits next to c = 4 instead of 6: a pass less for the n = 28 inverse QFT, half the rounds
            const unsigned Tx = (unsigned)tn.fuse_x8_T;
            if (x8_ok) {
                c = std::min((unsigned)tn.fuse_x8_c, Tx - 1); budget = Tx - c;
                want_x8 = true;
            } else {
                c = cols_pass ? colb : std::min((unsigned)tn.fuse_c_phase, Tp); budget = Tp - c;
            }
            grow(c, budget);
            act.nopipe = 1;
        }
        if (i == first) { standalone(i++); continue; }             // does not fit a tile at all
        if (i - first == 1) { standalone(first); continue; }       // alone: its tuned kernel
        // A pass whose hot bits all lie below bit 12 takes the WHOLE low end of the index as its tile (up to 2^12 amplitudes): the
        // tile is then one contiguous block -- e.g. the last pass of the n = 30 Shor circuit's inverse QFT, Hadamards on qubits
        // 5 .. 11 with c = 4: bit 4 (an M-register bit no gate of the pass needs) would otherwise stay outside and cut the tile
        // into 256-B runs (7.4 -> 5.9 ms for that pass)
        if (!want_q3 && tn.fuse_rounds && tn.fuse_lowtile && !hbits.empty()) {
            const unsigned top = *std::max_element(hbits.begin(), hbits.end()) + 1;
            // (bit-exact passes only up to 2^11: the 1024-thread form of the exact rounds kernel loses more than the contiguous
            // tile gains -- n = 30 Shor circuit 36.9 -> 40.2 ms when its last pass went to 2^12; the tolerance mode's passes gain: 24.6 -> 23.7)
            if (top <= ((tol || want_x8) ? 12u : 11u) && top <= n && top >= 10 && top > c + hbits.size()) {
                hbits.clear();
                for (unsigned b = c; b < top; b++) hbits.push_back(b);
                budget = top - c;
            }
        }
        // pad the tile with the lowest free bits (longer contiguous runs) up to T bits
        for (unsigned b = c; hbits.size() < budget && b < n; b++)
            if (std::find(hbits.begin(), hbits.end(), b) == hbits.end()) hbits.push_back(b);
        std::sort(hbits.begin(), hbits.end());

        PassShape sh;
        sh.first = first; sh.last = i; sh.c = c; sh.hbits = hbits; sh.want_q3 = want_q3; sh.want_x8 = want_x8; sh.cols = cols_pass; sh.cols_tol = cols_pass && first_cols == 2;
        sh.n_h = n_h; sh.n_ph = n_ph; sh.n_other = n_other; sh.nopipe = act.nopipe;
        for (unsigned b = 0; b < c; b++) sh.tl.push_back(b);
        for (unsigned b : hbits) sh.tl.push_back(b);
        if (emit(sh, act)) { q3_refused = true; i = first; continue; }
        if (!pass_tables_inplace(act.P, n, sh.tl)) {                // (cannot happen: at most T - c + 1 segments)
            all_ops.resize(act.op_off);
            for (size_t k = first; k < i; k++) standalone(k);
            continue;
        }
        acts.push_back(act);
        shape_of.push_back((int)shapes.size());
        shapes.push_back(sh);
    }

    // ---- chains ---------------------------------------------------------------------------------------------------------
    if (!chain || !tn.fuse_chain || n < (unsigned)std::max<long>(tn.fuse_chain_min_n, 13)) return;
    auto chainable = [&](size_t a) {
        if (!acts[a].fused) return false;
        const FusePass &P = acts[a].P;
        if (!tn.fuse_ldsdma || tn.fuse_rounds_occ < 6) return false;               // (launch_pass would take the general kernel, which works in place only)
        return P.cam_ctl_local[0] == 1 && !P.has_cam && P.T >= 10 && (P.T <= 12 || (P.T == 13 && P.dg_slim == 2)) && P.T < n;
    };
    bool any = false;
    for (size_t a0 = 0; a0 < acts.size();) {
        if (!chainable(a0)) { a0++; continue; }
        size_t a1 = a0;
        while (a1 < acts.size() && chainable(a1)) a1++;
        const size_t m = a1 - a0;
        if (m >= 2) {
            // layouts: lay[0] = identity (what the first pass reads); pass k of the chain writes lay[k + 1]; the last one the identity
            std::vector<std::vector<unsigned>> lay(m + 1, std::vector<unsigned>(n));
            for (unsigned q = 0; q < n; q++) lay[0][q] = lay[m][q] = q;
            // which side of a pass is the gathered one (fuse_chain_dir; -1: chosen here).  Memory-bound chains (radix-8 rounds:
            // Hadamard sweeps, tolerance mode) run better when a pass READS whole tiles and stores gathered (n = 30 sweep 19.7 vs
            // 21.6 ms, tolerance Shor 18.9 vs 20.3); chains of the exact phase walk when a pass STORES whole tiles and the next one
            // gathers (n = 30 Shor circuit 30.9 -> 29.7 ms: its third pass 7.6 -> 6.7)
            bool slim_any = false;
            for (size_t k = 0; k < m; k++) slim_any |= acts[a0 + k].P.dg_slim == 1 || acts[a0 + k].P.dg_slim == 2;       // (3 = the exact walk on 8 amplitudes: FP64-bound like the radix-4 walk)
            const bool store_whole = tn.fuse_chain_dir < 0 ? !slim_any : tn.fuse_chain_dir != 0;
            for (size_t k = 0; k + 1 < m; k++) {
                const PassShape &cur = shapes[shape_of[a0 + k]], &nxt = shapes[shape_of[a0 + k + 1]];
                std::vector<bool> in_cur(n, false), in_nxt(n, false), placed(n, false);
                for (unsigned q : cur.tl) in_cur[q] = true;
                for (unsigned q : nxt.tl) in_nxt[q] = true;
                std::vector<unsigned> order;                       // qubits by ascending position in the layout this pass writes
                // the next tile first (it will be read as one contiguous block): the qubits both tiles hold lowest -- they are
                // what this pass's stores are contiguous in --, then the rest of the next tile
                for (unsigned q = 0; q < n; q++) if (in_cur[q] && in_nxt[q]) { order.push_back(q); placed[q] = true; }
                if (!store_whole)
                    for (unsigned q = 0; q < n; q++) if (in_nxt[q] && !placed[q]) { order.push_back(q); placed[q] = true; }
                // then the rest of this pass's tile (short strides for its stores), then everything else
                // (store_whole, the other way round: THIS pass's tile is the contiguous one -- whole-tile stores, and the next
                //  pass gathers its tile in runs of the shared qubits)
                for (unsigned q = 0; q < n; q++) if (in_cur[q] && !placed[q]) { order.push_back(q); placed[q] = true; }
                for (unsigned q = 0; q < n; q++) if (!placed[q]) order.push_back(q);
                for (unsigned pos = 0; pos < n; pos++) lay[k + 1][order[pos]] = pos;
            }
            // the passes again, each with its tile ordered by where its input layout puts the qubits
            // (the tables are built on COPIES, with the tile order the launch will use, and taken over only when every pass
            //  of the chain has all three of its segment lists: a tile order whose free bits need more than QCX_MAX_SEG runs
            //  -- by output position at n >= 30 on scattered Hadamard lists -- falls back to the input order, and a chain that
            //  does not fit either way stays in place.  Round 4 validated one order and launched the other.)
            std::vector<PassShape> ns;
            std::vector<FusePass> nt;
            bool ok = true;
            for (size_t k = 0; k < m && ok; k++) {
                PassShape sh = shapes[shape_of[a0 + k]];
                std::sort(sh.tl.begin(), sh.tl.end(), [&](unsigned x, unsigned y) { return lay[k][x] < lay[k][y]; });
                FusePass Pt = acts[a0 + k].P;
                const bool by_out = (Pt.dg_slim == 1 || Pt.dg_slim == 2) && !store_whole;
                ok = pass_tables_chained(Pt, n, sh.tl, lay[k], lay[k + 1], by_out);
                if (!ok && by_out) { Pt = acts[a0 + k].P; ok = pass_tables_chained(Pt, n, sh.tl, lay[k], lay[k + 1], false); }
                ok = ok && pass_tables_cover(Pt, n);
                ns.push_back(sh);
                nt.push_back(Pt);
            }
            if (ok) {
                for (size_t k = 0; k < m; k++) {
                    shapes[shape_of[a0 + k]] = ns[k];
                    acts[a0 + k].P = nt[k];
                }
                any = true;
            }
        }
        a0 = a1;
    }
    if (!any) return;
    // re-emit every pass's records: a chained pass numbers its tile-local bits differently
    std::vector<FuseAction> old;
    old.swap(acts);
    all_ops.clear();
    for (size_t a = 0; a < old.size(); a++) {
        if (!old[a].fused) { acts.push_back(old[a]); continue; }
        FuseAction act;
        memset(&act, 0, sizeof act);
        if (emit(shapes[shape_of[a]], act)) {                      // (cannot happen: the same gates were emitted once already)
            acts.clear(); all_ops.clear();
            fuse_plan(r, tn, gates_in, acts, all_ops, tol, false, first_cols);
            return;
        }
        const FusePass &O = old[a].P;                             // the addressing tables were computed above
        act.P.chained = O.chained;
        act.P.nseg_in = O.nseg_in; act.P.nseg_out = O.nseg_out; act.P.nseg_lg = O.nseg_lg;
        memcpy(act.P.in_pos, O.in_pos, sizeof act.P.in_pos);
        memcpy(act.P.st_loc, O.st_loc, sizeof act.P.st_loc);
        memcpy(act.P.st_pos, O.st_pos, sizeof act.P.st_pos);
        memcpy(act.P.seg_in, O.seg_in, sizeof act.P.seg_in);
        memcpy(act.P.seg_out, O.seg_out, sizeof act.P.seg_out);
        memcpy(act.P.seg_lg, O.seg_lg, sizeof act.P.seg_lg);
        acts.push_back(act);
    }
}

// the host half of the front: which prefix of `gates` has the closed form on basis state `basis`, and the kernel's
// parameters for it.  Pure (no HIP call): qcx_front_plan exposes it to the CPU-only tests, which emulate k_basis_front in
// numpy and compare with the oracle.  Returns the number of gates consumed (0: nothing to fuse).
static size_t front_plan(unsigned n, unsigned M, uint64_t basis, const Tune &tn, const std::vector<QGate> &gates, BasisFront *Bout)
{
    size_t k = 0;
    uint64_t hmask = 0;
    unsigned nh = 0;
    const uint64_t lowmask = ((uint64_t)1 << M) - 1;
    BasisFront B; memset(&B, 0, sizeof B);
    if (tn.fuse_front && M <= 26) {
        while (k < gates.size() && gates[k].type == FUSE_H && !((hmask >> gates[k].q) & 1)) { hmask |= (uint64_t)1 << gates[k].q; nh++; k++; }
        if (!(hmask & lowmask))
            while (k < gates.size() && gates[k].type == FUSE_CAMODC && gates[k].q != 0xffffffffu && B.ncam < 64) {
                B.C[B.ncam] = gates[k].C; B.A[B.ncam] = gates[k].A % gates[k].C; B.ctl[B.ncam] = (uint8_t)gates[k].q; B.ncam++; k++;
            }
    }
    const uint64_t nmask = (n >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << n) - 1);
    B.first = 0; B.basis = basis; B.hmask = hmask; B.M = M;
    B.fixed_mask = ~hmask & ~lowmask & nmask;
    B.sign_mask = basis & hmask;
    double v = 1.0;
    for (unsigned q = 0; q < nh; q++) v = QCX_SQRT1_2 * v;              // fl(s * v), one rounding per Hadamard like the kernels
    B.v = v;
    *Bout = B;
    return k;
}

// one shard's (or the whole register's) front: the wave-tile kernel, or one thread per amplitude for tiny registers
static int launch_basis_front(amp_t *amp, unsigned n_local, const BasisFront &B, hipStream_t st)
{
    if (B.M > 12) {                       // a workgroup per 2^M-block (the wave tile would be 64 blocks = 2^(M + 6) amplitudes)
        const uint64_t nblocks = ((uint64_t)1 << n_local) >> B.M;
        hipLaunchKernelGGL(k_basis_front_big, dim3(grid_for(nblocks, 1, 16384)), dim3(256), 0, st, amp, n_local, B);
    } else if (n_local >= B.M + 6) {
        const uint64_t nwaves = ((uint64_t)1 << n_local) >> (6 + B.M);
        hipLaunchKernelGGL(k_basis_front, dim3(grid_for(nwaves, 4, 65536, 256)), dim3(256), 0, st, amp, n_local, B);
    } else
        hipLaunchKernelGGL(k_basis_front_small, dim3(grid_for((uint64_t)1 << n_local, 256, 1024, 256)), dim3(256), 0, st, amp, n_local, B);
    HIP_TRY(hipGetLastError());
    return QCX_NO_ERROR;
}

// The register is the basis state r->basis_index but nothing has been written yet (lazy reset / collapse).  Write it now --
// together with the longest prefix of the queue that has a closed form on a basis state: Hadamards on distinct qubits, then
// controlled modular multiplies (K0b, k_basis_front; the front of Q:712-737 is exactly that).  *used = gates consumed.
// The register is the basis state r->basis_index but nothing has been written yet (lazy reset / collapse).  Write it now --
// together with the longest prefix of the queue that has a closed form on a basis state: Hadamards on distinct qubits, then
// controlled modular multiplies (K0b, k_basis_front; the front of Q:712-737 is exactly that).  *used = gates consumed.
// the separate write of the basis state + the closed-form front B (k gates of the queue)
static int launch_front(qcx_register *r, const BasisFront &B, size_t k)
{
    const unsigned n = r->n;
    // (basis_pending is the only record of the logical state: it is cleared once the write has been launched, not before)
    if (k == 0) {                                                       // nothing to fuse: the plain write
        if (r->basis_index == 1) QCX_TRY(qcx_shard_reset(r->amp, n, 1, r->stream));
        else QCX_TRY(qcx_shard_collapse(r->amp, n, (int64_t)r->basis_index, r->stream));
        r->basis_pending = 0;
        return QCX_NO_ERROR;
    }
    QCX_TRY(launch_basis_front(r->amp, n, B, r->stream));
    r->basis_pending = 0;
    r->fronts++;
    return QCX_NO_ERROR;
}

static int basis_front(qcx_register *r, const std::vector<QGate> &gates, size_t *used)
{
    BasisFront B;
    const size_t k = front_plan(r->n, (unsigned)r->M, r->basis_index, tune_now(), gates, &B);
    QCX_TRY(launch_front(r, B, k));
    *used = k;
    return QCX_NO_ERROR;
}

// Can the front B be generated tile by tile inside the pass `act` -- the first pass behind it, which reads the identity
// layout: its tile-local bit j IS qubit in_pos[j] -- instead of being written by a pass of its own?  (GenFront, qcx_kernels.h:
// one modulus for the whole ladder, the basis residue below it; a rounds-form pass without multiplies.)  Fills G.
static bool gen_front_build(unsigned n, unsigned M, const BasisFront &B, const FuseAction &act, GenFront *G)
{
    const FusePass &P = act.P;
    if (!act.fused || P.has_cam || P.cam_ctl_local[0] != 1 || P.T < 10 || P.T > 12 || M > 12 || B.first != 0 || n > 40) return false;
    memset(G, 0, sizeof *G);
    const uint32_t lowmask = (1u << M) - 1u;
    const uint32_t f0 = (uint32_t)(B.basis & lowmask);
    uint32_t C = 0;
    for (unsigned g = 0; g < B.ncam; g++) {
        if (g == 0) C = B.C[0]; else if (B.C[g] != C) return false;
    }
    if (B.ncam && (C == 0 || f0 >= C || C > 4096u || B.ncam > 64)) return false;
    uint64_t tilemask = 0;
    unsigned h = 0;
    int slot_of[64];
    for (unsigned q = 0; q < 64; q++) slot_of[q] = -1;
    for (unsigned j = 0; j < P.T; j++) {
        const unsigned q = P.in_pos[j];
        tilemask |= (uint64_t)1 << q;
        G->slotbit[j] = G->lowbit[j] = 0xff;
        G->signbit[j] = (uint8_t)((B.sign_mask >> q) & 1u);
        if (q < M) G->lowbit[j] = (uint8_t)q;
        else { slot_of[q] = (int)h; G->slotbit[j] = (uint8_t)h; h++; }
    }
    if (h > (P.dg_slim == 2 ? 9u : 10u)) return false;              // the slot tables' room in LDS (k_fused_x8, dg_slim 3: 1024 slots like the radix-4 kernel)
    G->h = h;
    for (unsigned q = M; q < n; q++)
        if (slot_of[q] >= 0 && ((B.fixed_mask >> q) & 1u)) { G->sfm |= 1u << slot_of[q]; G->sbv |= (uint32_t)((B.basis >> q) & 1u) << slot_of[q]; }
    G->basis = B.basis;
    G->fixed_out = B.fixed_mask & ~tilemask;
    G->sign_out = B.sign_mask & ~tilemask;
    G->v = B.v;
    G->M = M; G->ncam = B.ncam; G->C = B.ncam ? C : 0u; G->f0 = f0;
    G->Cinv = G->C ? (uint32_t)(((uint64_t)1 << 32) / G->C) : 0u;
    G->cmpmask = ~(uint32_t)(B.hmask & lowmask) & lowmask;
    G->lowout_mask = lowmask & ~(uint32_t)tilemask;
    for (unsigned f = 0; f < 5; f++) for (unsigned b = 0; b < 256; b++) G->tabP[f][b] = 1;
    for (unsigned g = 0; g < B.ncam; g++) {
        const unsigned ctl = B.ctl[g];
        G->camA[g] = B.A[g] % C;
        if (ctl >= 40) return false;
        if (slot_of[ctl] >= 0) { G->camloc[g] = (uint8_t)slot_of[ctl]; continue; }
        if ((tilemask >> ctl) & 1u) return false;                   // (a control below M inside the tile: front_plan never produces it)
        G->camloc[g] = 0xff;
        const unsigned f = ctl >> 3, bit = ctl & 7u;
        G->present |= 1u << f;
        for (unsigned b = 0; b < 256; b++)
            if ((b >> bit) & 1u) G->tabP[f][b] = (uint16_t)(((uint32_t)G->tabP[f][b] * G->camA[g]) % C);
    }
    return true;
}

// zero-wave skipping (FusePass::zskip) for the passes of a flush whose M register no gate touches: the wave number rides on
// M-register bits of the tile.  skip_first: the first pass is the generated one by columns (no empty waves there).
static void zskip_setup(std::vector<FuseAction> &acts, std::vector<FuseOp> &all_ops, unsigned M, const Tune &tn, bool skip_first)
{
    for (FuseAction &a : acts) {
        if (skip_first && &a == &acts[0]) continue;
        if (!a.fused || a.P.cam_ctl_local[0] != 1 || a.P.has_cam || a.P.dg_slim >= 2 || a.P.T < 10 || a.P.T > 12) continue;
        const unsigned W = a.P.T - 8;                          // waves per workgroup = 2^W (4 amplitudes per thread)
        std::vector<unsigned> passive;                         // tile-local positions of M-register bits, by ascending qubit
        for (unsigned q = 0; q < M; q++)
            for (unsigned j = 0; j < a.P.T; j++) if (a.tl[j] == q) passive.push_back(j);
        if (passive.size() < W || W > (unsigned)tn.fuse_zskip_maxw) continue;          // (W = 3: the lanes' LDS stride costs 8-way bank conflicts -- the 2^11 pass of the n = 30 Shor circuit 7.4 -> 8.7 ms)
        std::vector<unsigned> z(passive.end() - W, passive.end());
        std::sort(z.begin(), z.end());
        a.P.zskip = (uint8_t)W;
        a.P.zlist = 0;
        for (unsigned j = 0; j < W; j++) { a.P.zb[j] = (uint8_t)z[j]; a.P.zlist |= (uint64_t)z[j] << (8 * j); }
        // every round's header gets the sorted list of the positions its threads' lane numbers leave out: zb[] + its register bits
        bool ok = true;
        for (size_t o = a.op_off; o < a.op_off + a.P.nops && ok; ) {
            FuseOp &h = all_ops[o];
            const uint32_t ty = h.type & 0xffu;
            if (ty != FUSE_ROUND && ty != FUSE_QROUND) { ok = false; break; }
            std::vector<unsigned> e(z);
            e.push_back(h.a & 0xffu); e.push_back((h.a >> 8) & 0xffu);
            std::sort(e.begin(), e.end());
            for (size_t k = 1; k < e.size(); k++) ok &= e[k] != e[k - 1];          // (a register bit among the zb[]: cannot happen -- no Hadamard on the M register)
            uint64_t list = 0;
            for (size_t k = 0; k < e.size(); k++) list |= (uint64_t)e[k] << (8 * k);
            memcpy(&h.c, &list, sizeof list);
            o += 1 + (size_t)h.mask;
        }
        if (!ok) a.P.zskip = 0;
    }
}

// the records of a flush: one pinned copy, one upload
static int upload_ops(qcx_register *r, GateQueue *gq, const std::vector<FuseOp> &all_ops)
{
    if (all_ops.empty()) return QCX_NO_ERROR;
    if (gq->ev_valid) HIP_TRY(hipEventSynchronize(gq->ev));       // the previous flush may still read the buffers
    const size_t need_ops = all_ops.size() + 1;                  // + 1: the walk prefetches one header past the last item
    if (gq->h_cap < need_ops) {
        if (gq->h_ops) HIP_TRY(hipHostFree(gq->h_ops));
        gq->h_ops = nullptr; gq->h_cap = 0;
        HIP_TRY(hipHostMalloc(&gq->h_ops, need_ops * sizeof(FuseOp)));
        gq->h_cap = need_ops;
    }
    if (gq->d_cap < need_ops) {
        if (gq->d_ops) HIP_TRY(hipFree(gq->d_ops));
        gq->d_ops = nullptr; gq->d_cap = 0;
        HIP_TRY(hipMalloc(&gq->d_ops, need_ops * sizeof(FuseOp)));
        gq->d_cap = need_ops;
    }
    memcpy(gq->h_ops, all_ops.data(), all_ops.size() * sizeof(FuseOp));
    memset(gq->h_ops + all_ops.size(), 0, sizeof(FuseOp));
    HIP_TRY(hipMemcpyAsync(gq->d_ops, gq->h_ops, need_ops * sizeof(FuseOp), hipMemcpyHostToDevice, r->stream));
    gq->upload_stamp++;
    return QCX_NO_ERROR;
}

// ---- compact chains (round 4) ----------------------------------------------------------------------------------------------
// Behind a circuit front the M register reads one of the R residues of the multiply ladder's orbit (C = 21, a = 2: six of 32
// values) and nothing that follows in the queue touches it (an inverse QFT works on the L register): all other amplitudes
// are +0 and stay +0.  Such a flush runs on a COMPACT copy of the state -- index [L-register bits][column], 2^cb >= R columns,
// column j = the amplitudes whose M register reads orbit[j] -- which is a register of L + cb qubits in its own right (the
// "virtual" register: M' = cb, qubit q of the real register is qubit q - M + cb): the first pass generates it by columns
// (K6g, gen = 3), the ordinary planner and pass kernels take the rest of the gate list through it (chained between two
// compact buffers carved out of the register's second buffer), and k_expand_compact writes the real register once at the
// end: 2^(M - cb) times less memory traffic in every pass but the last write.  Same arithmetic on the same amplitudes in the
// same order: same bits.  *done = false: not applicable, nothing was launched.
// The orbit a circuit front leaves the M register on, and whether a compact form pays: one modulus, f0 < C <= 4096, every residue
// inside the register, at most 16 of them, and 2^cb columns (cb >= 2) at least four times fewer than 2^M.  The orbit is the
// closure of f0 under every multiplier of the ladder -- a superset of the subset products the front can reach.
static bool compact_orbit(const BasisFront &B, unsigned M, std::vector<uint16_t> &orbit, unsigned *cb_out)
{
    orbit.clear();
    if (M < 4 || M > 12) return false;
    const uint32_t lowmask = (1u << M) - 1u;
    if ((B.hmask & lowmask) != 0 || B.ncam > 64) return false;
    const uint32_t Cn = B.ncam ? B.C[0] : 0u, f0 = (uint32_t)(B.basis & lowmask);
    for (unsigned g = 0; g < B.ncam; g++) if (B.C[g] != Cn) return false;
    if (!B.ncam) orbit.push_back((uint16_t)f0);
    else {
        if (Cn == 0 || Cn > 4096u || f0 >= Cn) return false;
        std::vector<char> seen(Cn, 0);
        std::vector<uint32_t> todo(1, f0);
        seen[f0] = 1;
        while (!todo.empty()) {
            const uint32_t x = todo.back(); todo.pop_back();
            for (unsigned g = 0; g < B.ncam; g++) { const uint32_t y = (uint32_t)(((uint64_t)x * (B.A[g] % Cn)) % Cn); if (!seen[y]) { seen[y] = 1; todo.push_back(y); } }
        }
        for (uint32_t x = 0; x < Cn; x++) if (seen[x]) { if (x > lowmask) { orbit.clear(); return false; } orbit.push_back((uint16_t)x); }
    }
    if (orbit.size() > 16) { orbit.clear(); return false; }
    unsigned cb = 2;
    while ((1u << cb) < orbit.size()) cb++;
    if (cb + 2 > M) { orbit.clear(); return false; }
    *cb_out = cb;
    return true;
}

// the deferred last pass of a compact chain (compact_pending == 2, GateQueue::last): into_register: with the expanding store,
// the state is then in r->amp (compact_pending = 0); otherwise as planned, the compact form is complete (compact_pending = 1)
static int launch_pass(qcx_register *r, const Tune &tn, const FusePass &P_in, const FuseOp *d_ops, bool nopipe, amp_t *amp_in, amp_t *amp_out);
static int compact_finish_last(qcx_register *r, bool into_register)
{
    if (r->compact_pending != 2) return QCX_NO_ERROR;
    GateQueue *gq = r->queue;
    if (!gq) return QCX_UNKNOWN_ERROR;
    qcx_register v;
    memset(&v, 0, sizeof v);
    v.L = (int)(gq->last.nv - gq->last.cb); v.M = (int)gq->last.cb; v.n = gq->last.nv; v.dim = (uint64_t)1 << gq->last.nv;
    v.amp = gq->last.in; v.scratch = gq->last.out;
    v.own_stream = r->own_stream; v.stream = r->stream; v.fusion = r->fusion;
    if (into_register) {
        QCX_TRY(launch_pass(&v, gq->last.tn, gq->last.Pxp, gq->d_ops + gq->last.op_off, gq->last.nopipe, gq->last.in, r->amp));
        r->compact_pending = 0;
        gq->expanding_stores++;
    } else {
        QCX_TRY(launch_pass(&v, gq->last.tn, gq->last.P, gq->d_ops + gq->last.op_off, gq->last.nopipe, gq->last.in, gq->last.out));
        r->compact_amp = gq->last.out;
        r->compact_pending = 1;
    }
    if (gq->last.P.chained) gq->chained_passes++;
    gq->passes_launched++;
    gq->gates_fused += gq->last.ngates;
    if (!gq->ev_valid) { HIP_TRY(hipEventCreateWithFlags(&gq->ev, hipEventDisableTiming)); gq->ev_valid = true; }
    HIP_TRY(hipEventRecord(gq->ev, r->stream));            // (the record buffers are in use until this pass is through)
    return QCX_NO_ERROR;
}

// the real register from a pending compact form
static int expand_pending(qcx_register *r)
{
    if (r->compact_pending == 2) QCX_TRY(compact_finish_last(r, true));
    if (!r->compact_pending) return QCX_NO_ERROR;
    ExpandParams E;
    memset(&E, 0, sizeof E);
    E.M = (unsigned)r->M; E.cb = r->compact_cb; E.ncols = r->compact_ncols;
    for (unsigned j = 0; j < r->compact_ncols; j++) E.orbit[j] = r->compact_orbit[j];
    const uint64_t nchunks = ((uint64_t)1 << (r->n - (unsigned)r->M)) >> 6;  // 64 blocks per workgroup iteration (L >= 8)
    hipLaunchKernelGGL(k_expand_compact, dim3(grid_for(nchunks, 1, 65536)), dim3(256), 0, r->stream, (const amp_t *)r->compact_amp, r->amp, nchunks, E, (int)tune_now().fuse_expand_direct);
    HIP_TRY(hipGetLastError());
    r->compact_pending = 0;
    return QCX_NO_ERROR;
}

// the expanding store of a compact chain's last pass (FusePass::xp_*; k_fused_x8): possible when the pass's output order starts
// with the column bits (they are the cb lowest bits of the virtual register and all inside the tile).  An expanded store
// index is [the tile's store-order bits above the column bits][f: M bits]; bit i >= M of it lands on REAL index bit
// st_pos - cb + M and selects tile-local bit st_loc of the store-order bit it stands for.
static bool xp_setup(FusePass &PL, unsigned M, unsigned cb, const std::vector<uint16_t> &orbit)
{
    if (PL.T != 12 || M > 9 || cb > 4 || cb > PL.T || orbit.size() > 16 || M + PL.T - cb > 24) return false;
    for (unsigned j = 0; j < cb; j++) if (PL.st_pos[j] != j) return false;
    PL.xp_on = 1; PL.xp_M = (uint8_t)M; PL.xp_cb = (uint8_t)cb; PL.xp_ncols = (uint8_t)orbit.size();
    for (size_t j = 0; j < orbit.size(); j++) PL.xp_orbit[j] = orbit[j];
    for (unsigned j = 0; j < cb; j++) PL.xp_colloc[j] = PL.st_loc[j];
    for (unsigned i = M; i < M + PL.T - cb; i++) { PL.xp_pos[i] = (uint8_t)(PL.st_pos[cb + i - M] - cb + M); PL.xp_loc[i] = PL.st_loc[cb + i - M]; }
    return true;
}

// keep: leave the result in its compact form (r->compact_pending; measure_state reads it there, everything else expands it first)
static int compact_chain(qcx_register *r, GateQueue *gq, const Tune &tn, const BasisFront &Bf, size_t kfront, const std::vector<QGate> &gates, bool keep, bool *done)
{
    *done = false;
    const unsigned M = (unsigned)r->M, n = r->n, L = n - M;
    if (!tn.fuse_compact || !tn.fuse_gen_cols || !tn.fuse_chain || !tn.fuse_ldsdma || tn.fuse_rounds_occ < 6 || !tn.fuse_rounds) return QCX_NO_ERROR;
    if (M < 4 || M > 12 || r->no_chain || !r->own_stream || Bf.first != 0 || gates.empty()) return QCX_NO_ERROR;
    const uint32_t lowmask = (1u << M) - 1u;
    if ((Bf.hmask & lowmask) != 0 || Bf.ncam > 64) return QCX_NO_ERROR;
    for (const QGate &g : gates) {
        if (g.type == FUSE_H) { if (g.q < M) return QCX_NO_ERROR; }
        else if (g.type == FUSE_PHASE) { if (g.mask & lowmask) return QCX_NO_ERROR; }
        else return QCX_NO_ERROR;
    }
    std::vector<uint16_t> orbit;
    unsigned cb = 0;
    // (the plan cache, GateQueue::pc: the same front and the same gate list under the same knobs as the last chain -- the next
    //  attempt of a period-finding run -- take orbit, actions and records from there)
    const bool hit = tn.fuse_plan_cache && gq->pc.valid && gq->pc.kind == 2 && gq->pc.n == n && gq->pc.M == M && gq->pc.fusion == r->fusion
                     && gq->pc.kfront == kfront && memcmp(&gq->pc.tn, &tn, sizeof tn) == 0 && memcmp(&gq->pc.Bf, &Bf, sizeof Bf) == 0
                     && same_gate_list(gq->pc.gates, gates);
    if (hit) { orbit = gq->pc.orbit; cb = gq->pc.cb; }
    else if (!compact_orbit(Bf, M, orbit, &cb)) return QCX_NO_ERROR;
    const uint32_t Cn = Bf.ncam ? Bf.C[0] : 0u, f0 = (uint32_t)(Bf.basis & lowmask);
    const unsigned nv = L + cb;
    if (nv < 14 || L < 8) return QCX_NO_ERROR;
    // the virtual register's gate list
    std::vector<QGate> vg(gates);
    for (QGate &g : vg) { if (g.type == FUSE_H) g.q -= M - cb; else g.mask >>= (M - cb); }
    if (!r->scratch) {
        if (hipMalloc(&r->scratch, r->dim * sizeof(amp_t)) != hipSuccess) { (void)hipGetLastError(); r->scratch = nullptr; r->no_chain = 1; return QCX_NO_ERROR; }
    }
    qcx_register v;
    memset(&v, 0, sizeof v);
    v.L = (int)L; v.M = (int)cb; v.n = nv; v.dim = (uint64_t)1 << nv;
    v.amp = r->scratch; v.scratch = r->scratch + v.dim;             // (2 * 2^nv <= 2^n amplitudes: cb + 1 <= M)
    v.own_stream = r->own_stream; v.stream = r->stream; v.fusion = r->fusion;
    std::vector<FuseAction> acts;
    std::vector<FuseOp> all_ops;
    if (hit) { acts = gq->pc.acts; gq->plan_hits++; }
    else {
    fuse_plan(&v, tn, vg, acts, all_ops, r->fusion == 2, true, (r->fusion == 2 && tn.fuse_cols_tol) ? 2 : 1);
    if (acts.empty() || !acts[0].fused) return QCX_NO_ERROR;
    {
        const FusePass &P0 = acts[0].P;
        if (P0.T != cb + 8 || P0.c != cb || P0.cam_ctl_local[0] != 1 || P0.has_cam || P0.dg_slim == 2 || P0.dg_cnt > 64) return QCX_NO_ERROR;
        for (unsigned j = 0; j < cb; j++) if (acts[0].tl[j] != j) return QCX_NO_ERROR;
        for (unsigned j = cb; j < cb + 8; j++) if (acts[0].tl[j] < cb) return QCX_NO_ERROR;
        for (size_t o = acts[0].op_off; o < acts[0].op_off + P0.nops; ) {
            const FuseOp &hdr = all_ops[o];
            const uint32_t ty = hdr.type & 0xffu;
            if ((ty != FUSE_ROUND && !(ty == FUSE_QROUND && P0.dg_cnt)) || (hdr.a & 0xffu) < cb || ((hdr.a >> 8) & 0xffu) < cb) return QCX_NO_ERROR;
            o += 1 + (size_t)hdr.mask;
        }
    }
    // (columns beyond the orbit hold nothing -- 2 of 8 for the six residues of N = 21 -- but letting the waves that sit on them
    //  skip the rounds of the later passes, zskip_setup, does not pay: 2.50 / 2.09 ms against 2.28 / 1.96 for the two 2^10-tile
    //  passes at n = 30: the busy waves are a tile's critical path either way, and the lanes' stride costs bank conflicts)
    // the generated fill of the first pass, in REAL qubit numbers (its controls, fixed bits and signs), by hot slot
    GenFront G;
    memset(&G, 0, sizeof G);
    {
        uint64_t tilemask = lowmask;
        int slot_of[64];
        for (unsigned q = 0; q < 64; q++) slot_of[q] = -1;
        for (unsigned j = cb; j < cb + 8; j++) {
            const unsigned rq = acts[0].tl[j] - cb + M, slot = j - cb;
            slot_of[rq] = (int)slot; tilemask |= (uint64_t)1 << rq;
            if ((Bf.sign_mask >> rq) & 1u) G.sgn_slots |= 1u << slot;
            if ((Bf.fixed_mask >> rq) & 1u) { G.sfm |= 1u << slot; G.sbv |= (uint32_t)((Bf.basis >> rq) & 1u) << slot; }
        }
        G.basis = Bf.basis; G.fixed_out = Bf.fixed_mask & ~tilemask; G.sign_out = Bf.sign_mask & ~tilemask;
        G.v = Bf.v; G.M = M; G.ncam = Bf.ncam; G.C = Bf.ncam ? Cn : 0u; G.f0 = f0;
        G.Cinv = G.C ? (uint32_t)(((uint64_t)1 << 32) / G.C) : 0u;
        G.cmpmask = lowmask; G.lowout_mask = 0; G.h = 8;
        for (unsigned f = 0; f < 5; f++) for (unsigned b = 0; b < 256; b++) G.tabP[f][b] = 1;
        for (unsigned g = 0; g < Bf.ncam; g++) {
            const unsigned ctl = Bf.ctl[g];
            G.camA[g] = Bf.A[g] % Cn;
            if (ctl >= 40 || ctl < M) return QCX_NO_ERROR;
            if (slot_of[ctl] >= 0) { G.camloc[g] = (uint8_t)slot_of[ctl]; continue; }
            G.camloc[g] = 0xff;
            const unsigned f = ctl >> 3, bit = ctl & 7u;
            G.present |= 1u << f;
            for (unsigned b = 0; b < 256; b++)
                if ((b >> bit) & 1u) G.tabP[f][b] = (uint16_t)(((uint32_t)G.tabP[f][b] * G.camA[g]) % Cn);
        }
        G.cb = cb; G.ncols = (uint32_t)orbit.size();
        for (size_t j = 0; j < orbit.size(); j++) G.orbit[j] = orbit[j];
    }
    {
        const size_t at = all_ops.size(), nrec = (sizeof(GenFront) + sizeof(FuseOp) - 1) / sizeof(FuseOp);
        all_ops.resize(at + nrec);
        memset(&all_ops[at], 0, nrec * sizeof(FuseOp));
        memcpy(&all_ops[at], &G, sizeof G);
        acts[0].P.gen = 3;
        acts[0].P.gen_tab_bytes = (uint32_t)(((G.C ? G.C : (1u << M)) + 15u) & ~15u);
        acts[0].P.zpad = (uint16_t)orbit.size();
        acts[0].P.zskip = 0;
        acts[0].P.gen_rec_off = (uint32_t)(at - acts[0].op_off);
    }
    gq->pc.valid = false;
    if (tn.fuse_plan_cache) {               // (valid once the records are uploaded, below)
        gq->pc.kind = 2; gq->pc.n = n; gq->pc.M = M; gq->pc.fusion = r->fusion; gq->pc.chain = true; gq->pc.tn = tn; gq->pc.Bf = Bf; gq->pc.kfront = kfront;
        gq->pc.gates = gates; gq->pc.acts = acts; gq->pc.all_ops = all_ops; gq->pc.orbit = orbit; gq->pc.cb = cb;
    }
    }
    // Round 5: when the chain's last pass is a k_fused_x8 pass whose tile holds the column bits, that pass can store the REAL
    // register itself (FusePass::xp_on) -- k_expand_compact's extra read and write of the compact form (8.6 of its 21.5 GB at
    // n = 30, M = 5) disappear.  keep = false: it does.  keep = true (a whole-circuit entry point: measure_state may come next and
    // wants the compact form): the last pass is DEFERRED (GateQueue::last, compact_pending = 2) until somebody looks.
    // (The last pass of a plan always stores the identity layout.)
    bool expanded = false, deferred = false;
    if (tn.fuse_expand_fused && acts.size() > 1 && acts.back().fused && pass_is_x8(acts.back().P, tn) && M <= 9 && cb <= 4) {
        FusePass PL = acts.back().P;
        if (xp_setup(PL, M, cb, orbit)) {
            if (keep) {
                deferred = true;
                gq->last.P = acts.back().P; gq->last.Pxp = PL;
                gq->last.op_off = acts.back().op_off; gq->last.nopipe = acts.back().nopipe != 0; gq->last.ngates = acts.back().ngates;
                gq->last.nv = nv; gq->last.cb = cb; gq->last.tn = tn;
            } else { acts.back().P = PL; expanded = true; }
        }
    }
    if (hit) {
        if (gq->pc.stamp != gq->upload_stamp) { QCX_TRY(upload_ops(r, gq, gq->pc.all_ops)); gq->pc.stamp = gq->upload_stamp; }
    } else {
        QCX_TRY(upload_ops(r, gq, all_ops));
        if (tn.fuse_plan_cache) { gq->pc.stamp = gq->upload_stamp; gq->pc.valid = true; }
    }
    for (size_t ai = 0; ai < acts.size(); ai++) {
        const FuseAction &act = acts[ai];
        if (!act.fused) { QCX_TRY(launch_standalone(&v, vg[act.gate])); continue; }
        if (deferred && ai + 1 == acts.size()) {
            gq->last.in = v.amp; gq->last.out = act.P.chained ? v.scratch : v.amp;
            break;
        }
        if (act.P.xp_on) {
            QCX_TRY(launch_pass(&v, tn, act.P, gq->d_ops + act.op_off, act.nopipe != 0, v.amp, r->amp));
            if (act.P.chained) gq->chained_passes++;
        } else if (act.P.chained) {
            QCX_TRY(launch_pass(&v, tn, act.P, gq->d_ops + act.op_off, act.nopipe != 0, v.amp, v.scratch));
            std::swap(v.amp, v.scratch);
            gq->chained_passes++;
        } else
            QCX_TRY(launch_pass(&v, tn, act.P, gq->d_ops + act.op_off, act.nopipe != 0, v.amp, v.amp));
        gq->passes_launched++;
        gq->gates_fused += act.ngates;
    }
    if (expanded) { r->compact_pending = 0; gq->expanding_stores++; }
    else {
        r->compact_pending = deferred ? 2 : 1;
        r->compact_amp = v.amp; r->compact_cb = cb; r->compact_ncols = (unsigned)orbit.size();
        for (size_t j = 0; j < orbit.size(); j++) r->compact_orbit[j] = orbit[j];
        if (!keep) QCX_TRY(expand_pending(r));
    }
    r->basis_pending = 0;
    r->zeros_dirty = 0;
    r->fronts++;
    gq->gen_fronts++; gq->gen_cols++; gq->compact_chains++;
    gq->gates_fused += kfront;
    if (!gq->ev_valid) { HIP_TRY(hipEventCreateWithFlags(&gq->ev, hipEventDisableTiming)); gq->ev_valid = true; }
    HIP_TRY(hipEventRecord(gq->ev, r->stream));
    *done = true;
    return QCX_NO_ERROR;
}

// plan -> upload every pass's records in one copy -> launch in order.  No host synchronisation except waiting for
// the PREVIOUS flush's kernels before its record buffers are reused.
static int fuse_flush(qcx_register *r, bool keep_compact = false)
{
    GateQueue *gq = r->queue;
    const Tune tn = tune_now();
    if (r->compact_pending) {                    // an earlier flush left the state compact
        if (keep_compact && (!gq || gq->gates.empty()) && !r->basis_pending) return QCX_NO_ERROR;
        if (r->basis_pending) r->compact_pending = 0;       // (a reset / collapse came after it: the compact form is history)
        else QCX_TRY(expand_pending(r));
    }
    // A lazily pending reset / collapse: the register IS a basis state that has not been written.  The closed-form front of
    // the queue (Hadamards, then the multiply ladder) is either written by a pass of its own (K0b) or -- when a fused pass
    // follows it -- generated tile by tile inside that pass (GenFront): no write pass, and that pass reads nothing.
    bool gen_try = false;
    const bool front_flush = r->basis_pending != 0;
    BasisFront Bf;
    size_t kfront = 0;
    if (r->basis_pending) {
        if (gq && !gq->gates.empty() && tn.fuse_gen && r->own_stream && r->n >= 12) {
            kfront = front_plan(r->n, (unsigned)r->M, r->basis_index, tn, gq->gates, &Bf);
            gen_try = gq->gates.size() > kfront;
        }
        if (!gen_try) {
            static const std::vector<QGate> none;
            size_t used = 0;
            QCX_TRY(basis_front(r, gq ? gq->gates : none, &used));
            if (used) { gq->gates.erase(gq->gates.begin(), gq->gates.begin() + used); gq->gates_fused += used; }
        }
    }
    if (!gq || gq->gates.empty()) return QCX_NO_ERROR;
    if (r->own_stream && !gen_try) QCX_TRY(canon_if_dirty(r));      // gates are about to run on a state the caller wrote (a shard view: its host's business)
    std::vector<QGate> gates;
    gates.swap(gq->gates);                       // the queue is empty from here on (re-entrancy safe)
    if (gen_try) gates.erase(gates.begin(), gates.begin() + kfront);
    if (gen_try) {                                // the whole flush on a compact copy of the state, when the front allows it
        bool done = false;
        QCX_TRY(compact_chain(r, gq, tn, Bf, kfront, gates, keep_compact && tn.fuse_compact_lazy, &done));
        if (done) return QCX_NO_ERROR;
    }
    std::vector<FuseAction> acts;
    std::vector<FuseOp> all_ops;
    // chains of passes go through the register's second buffer (allocated on first use; a register whose buffer pointer has
    // been handed out, a shard view and a register too large for a second buffer work in place)
    bool chain = tn.fuse_chain && r->own_stream && !r->no_chain && r->n >= (unsigned)std::max<long>(tn.fuse_chain_min_n, 13);
    // The generated first pass by columns (K6g): possible when nothing of the list touches the M register (>= 4 qubits), the front
    // left it on the orbit of ONE multiplier ladder, and that orbit populates at most 8 of the 16 values of the four lowest
    // M-register bits per value of the others (the columns a workgroup keeps in LDS).  maxcols = that bound.
    unsigned maxcols = 0;
    if (gen_try && tn.fuse_gen_cols && r->M >= 4 && r->M <= 12 && r->n >= 14 && tn.fuse_ldsdma && tn.fuse_rounds_occ >= 6) {
        const uint32_t lowmask = (1u << r->M) - 1u;
        bool ok = (Bf.hmask & lowmask) == 0 && Bf.ncam <= 64;
        for (const QGate &g : gates) ok &= !((g.type == FUSE_H && g.q < (unsigned)r->M) || g.type == FUSE_CAMODC || g.type == 99);
        const uint32_t Cn = Bf.ncam ? Bf.C[0] : 0u, f0 = (uint32_t)(Bf.basis & lowmask);
        for (unsigned g = 0; g < Bf.ncam && ok; g++) ok &= Bf.C[g] == Cn;
        if (Bf.ncam) ok &= Cn > 0 && Cn <= 4096u && f0 < Cn;
        if (ok && !Bf.ncam) maxcols = 1;                            // no multiplies: every populated block holds f0
        else if (ok) {
            std::vector<char> seen(Cn, 0);
            std::vector<uint32_t> todo(1, f0);
            seen[f0] = 1;
            while (!todo.empty()) {                                 // closure under every multiplier: a superset of the subset products
                const uint32_t x = todo.back(); todo.pop_back();
                for (unsigned g = 0; g < Bf.ncam; g++) { const uint32_t y = (uint32_t)(((uint64_t)x * (Bf.A[g] % Cn)) % Cn); if (!seen[y]) { seen[y] = 1; todo.push_back(y); } }
            }
            std::vector<uint16_t> colsets(((size_t)lowmask >> 4) + 1, 0);
            for (uint32_t x = 0; x < Cn; x++) if (seen[x] && x <= lowmask) colsets[x >> 4] |= (uint16_t)(1u << (x & 15u));
            for (uint16_t s : colsets) maxcols = std::max(maxcols, (unsigned)__builtin_popcount(s));
            if (maxcols > 8) maxcols = 0;
        }
    }
    auto cols_shape_ok = [&]() {
        if (acts.empty() || !acts[0].fused) return false;
        const FusePass &P0 = acts[0].P;
        if (P0.T != 12 || P0.c != 4 || P0.cam_ctl_local[0] != 1 || P0.has_cam || P0.dg_cnt || P0.dg_slim) return false;
        for (unsigned j = 0; j < 4; j++) if (acts[0].tl[j] != j) return false;
        for (unsigned j = 4; j < 12; j++) if (acts[0].tl[j] < (unsigned)r->M) return false;
        for (size_t o = acts[0].op_off; o < acts[0].op_off + P0.nops; ) {
            const FuseOp &hdr = all_ops[o];
            if ((hdr.type & 0xffu) != FUSE_ROUND || (hdr.a & 0xffu) < 4 || ((hdr.a >> 8) & 0xffu) < 4) return false;
            o += 1 + (size_t)hdr.mask;
        }
        return true;
    };
    // (the plan cache, GateQueue::pc: a flush whose inputs are those of the last one -- the front it stands behind included --
    //  takes its plan from there)
    const bool cacheable = tn.fuse_plan_cache != 0;
    bool hit = false;
    if (cacheable && gq->pc.valid && gq->pc.kind == 1 && gq->pc.n == r->n && gq->pc.M == (unsigned)r->M && gq->pc.fusion == r->fusion && gq->pc.chain == chain
        && gq->pc.front_flush == front_flush && gq->pc.gen_try == gen_try
        && (!gen_try || (gq->pc.kfront == kfront && memcmp(&gq->pc.Bf, &Bf, sizeof Bf) == 0))
        && memcmp(&gq->pc.tn, &tn, sizeof tn) == 0 && same_gate_list(gq->pc.gates, gates)) {
        bool needs_scratch = false;
        for (const FuseAction &a : gq->pc.acts) needs_scratch |= a.fused && a.P.chained;
        hit = !needs_scratch || r->scratch;
    }
    if (hit) { acts = gq->pc.acts; all_ops.clear(); gq->plan_hits++; }
    else {
    fuse_plan(r, tn, gates, acts, all_ops, r->fusion == 2, chain, maxcols != 0);
    if (maxcols && !cols_shape_ok()) { maxcols = 0; acts.clear(); all_ops.clear(); fuse_plan(r, tn, gates, acts, all_ops, r->fusion == 2, chain); }
    }
    bool chained_any = false;
    for (const FuseAction &a : acts) chained_any |= a.fused && a.P.chained;
    if (chained_any && !r->scratch) {
        if (hipMalloc(&r->scratch, r->dim * sizeof(amp_t)) != hipSuccess) {
            (void)hipGetLastError();
            r->scratch = nullptr; r->no_chain = 1;                 // no room (n = 34 fills the card): in place from now on
            acts.clear(); all_ops.clear();
            fuse_plan(r, tn, gates, acts, all_ops, r->fusion == 2, false, maxcols != 0);
            if (maxcols && !cols_shape_ok()) { maxcols = 0; acts.clear(); all_ops.clear(); fuse_plan(r, tn, gates, acts, all_ops, r->fusion == 2, false); }
        }
    }
    // Behind a circuit front only a few of the 2^M low index values are populated (the multiply ladder's orbit), and no gate of
    // an inverse QFT touches the M register: whole waves of a tile hold nothing but +0.  Passes of such a flush map their wave
    // number onto M-register bits of the tile and let all-zero waves skip the rounds (FusePass::zskip; found at run time, so
    // any state is handled correctly)
    if (!hit && front_flush && tn.fuse_zskip && r->M >= 2) {
        bool h_on_m = false;
        for (const QGate &g : gates) h_on_m |= (g.type == FUSE_H && g.q < (unsigned)r->M) || g.type == FUSE_CAMODC || g.type == 99;
        if (!h_on_m) zskip_setup(acts, all_ops, (unsigned)r->M, tn, maxcols != 0);
    }
    bool gen_built = false, gen2 = false;
    if (gen_try && hit) {                            // what the cached flush did with its front, again
        if (gq->pc.gen_built) {
            r->basis_pending = 0; r->fronts++; gq->gen_fronts++; r->zeros_dirty = 0;
            if (gq->pc.gen2) gq->gen_cols++;
        } else {
            QCX_TRY(launch_front(r, Bf, kfront));
            r->zeros_dirty = 0;
        }
        gq->gates_fused += kfront;
    } else if (gen_try) {
        GenFront G;
        if (!acts.empty() && gen_front_build(r->n, (unsigned)r->M, Bf, acts[0], &G)) {
            gen_built = true;
            const size_t at = all_ops.size(), nrec = (sizeof(GenFront) + sizeof(FuseOp) - 1) / sizeof(FuseOp);
            all_ops.resize(at + nrec);
            memset(&all_ops[at], 0, nrec * sizeof(FuseOp));
            memcpy(&all_ops[at], &G, sizeof G);
            acts[0].P.gen = 1;
            if (maxcols && G.cmpmask == (1u << r->M) - 1u && G.h == 8) { acts[0].P.gen = 2; acts[0].P.zpad = (uint16_t)maxcols; gq->gen_cols++; gen2 = true; }
            acts[0].P.gen_rec_off = (uint32_t)(at - acts[0].op_off);
            r->basis_pending = 0;                                 // (the pass that generates it is launched below; a failed launch returns its error)
            r->fronts++;
            gq->gen_fronts++;
            r->zeros_dirty = 0;
        } else {
            QCX_TRY(launch_front(r, Bf, kfront));                 // the separate write pass after all
            r->zeros_dirty = 0;
        }
        gq->gates_fused += kfront;
    }
    if (hit) {
        // the cached records: still on the device unless something was uploaded in between
        if (gq->pc.stamp != gq->upload_stamp) { QCX_TRY(upload_ops(r, gq, gq->pc.all_ops)); gq->pc.stamp = gq->upload_stamp; }
    } else {
        QCX_TRY(upload_ops(r, gq, all_ops));
        gq->pc.valid = false;
        if (cacheable) {
            gq->pc.kind = 1; gq->pc.n = r->n; gq->pc.M = (unsigned)r->M; gq->pc.fusion = r->fusion; gq->pc.chain = chain; gq->pc.tn = tn;
            gq->pc.front_flush = front_flush; gq->pc.gen_try = gen_try; gq->pc.gen_built = gen_built; gq->pc.gen2 = gen2;
            if (gen_try) { gq->pc.Bf = Bf; gq->pc.kfront = kfront; }
            gq->pc.gates = gates; gq->pc.acts = acts; gq->pc.all_ops = all_ops; gq->pc.stamp = gq->upload_stamp;
            gq->pc.valid = true;
        }
    }
    const bool any_records = hit ? !gq->pc.all_ops.empty() : !all_ops.empty();
    for (const FuseAction &act : acts) {
        if (!act.fused) { QCX_TRY(launch_standalone(r, gates[act.gate])); continue; }
        if (act.P.chained) {
            // out of place into the other buffer, which then IS the register's state (a chain ends on the identity layout, and
            // nothing between its passes looks at the buffers)
            QCX_TRY(launch_pass(r, tn, act.P, gq->d_ops + act.op_off, act.nopipe != 0, r->amp, r->scratch));
            std::swap(r->amp, r->scratch);
            gq->chained_passes++;
        } else
            QCX_TRY(launch_pass(r, tn, act.P, gq->d_ops + act.op_off, act.nopipe != 0, r->amp, r->amp));
        gq->passes_launched++;
        gq->gates_fused += act.ngates;
    }
    if (any_records) {
        if (!gq->ev_valid) { HIP_TRY(hipEventCreateWithFlags(&gq->ev, hipEventDisableTiming)); gq->ev_valid = true; }
        HIP_TRY(hipEventRecord(gq->ev, r->stream));
    }
    return QCX_NO_ERROR;
}

static int fuse_push(qcx_register *r, const QGate &g)
{
    if (!r->queue) r->queue = new GateQueue();
    r->queue->gates.push_back(g);
    if (r->queue->gates.size() >= (size_t)tune_now().fuse_max_queue) return fuse_flush(r);
    return QCX_NO_ERROR;
}
