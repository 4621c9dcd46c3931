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

struct GateQueue {
    std::vector<QGate> gates;
    FuseOp  *d_ops = nullptr;       // device copy of the ops of the passes in flight
    size_t   d_cap = 0;
    FuseOp  *h_ops = nullptr;       // pinned staging
    size_t   h_cap = 0;
    unsigned long passes_launched = 0, gates_fused = 0;
};

static void queue_free(GateQueue *gq)
{
    if (!gq) return;
    if (gq->d_ops) (void)hipFree(gq->d_ops);
    if (gq->h_ops) (void)hipHostFree(gq->h_ops);
    delete gq;
}

static bool camodc_closed_form(unsigned n, unsigned M, unsigned C, unsigned A, unsigned ctl)
{
    (void)A;
    if (M > n || ctl < M || C == 0) return false;
    const uint64_t blk = (uint64_t)1 << M;
    if (C > blk) return false;
    return (uint64_t)(C - 1) * (uint64_t)(C - 1) <= 0xffffffffULL;
}

struct PassPlan {
    std::vector<unsigned> hbits;    // hot bits >= c, kept sorted
    size_t first, last;             // gate range [first, last)
};

static int fuse_flush(qcx_register *r)
{
    GateQueue *gq = r->queue;
    if (!gq || gq->gates.empty()) return QCX_NO_ERROR;
    const unsigned n = r->n;
    unsigned T = (unsigned)g_tune.fuse_T, c = (unsigned)g_tune.fuse_c;
    if (T > 12) T = 12;
    if (T < 1) T = 1;
    if (T > n) T = n;
    if (c > T) c = T;
    const unsigned budget = T - c;

    std::vector<QGate> gates;
    gates.swap(gq->gates);                       // the queue is empty from here on (re-entrancy safe)

    size_t i = 0;
    while (i < gates.size()) {
        if (gates[i].type == 99) {
            // stand-alone launch (table-form modular multiply)
            const QGate &g = gates[i];
            int s = QCX_NO_ERROR;
            if (g.type == FUSE_H) s = qcx_shard_hadamard(r->amp, n, g.q, r->stream);
            else if (g.type == FUSE_PHASE) s = qcx_shard_phase(r->amp, n, g.mask, g.c, g.s, r->stream);
            else s = reg_camodc(r, g.C, g.A, g.q);
            if (s != QCX_NO_ERROR) return s;
            i++;
            continue;
        }
        // ---- grow one pass ------------------------------------------------------------------
        PassPlan pl;
        pl.first = i;
        auto need_of = [&](const QGate &g, std::vector<unsigned> &need) {
            need.clear();
            if (g.type == FUSE_H) { if (g.q >= c) need.push_back(g.q); }
            else if (g.type == FUSE_CAMODC) { for (unsigned b = c; b < (unsigned)r->M; b++) need.push_back(b); }
        };
        std::vector<unsigned> need;
        while (i < gates.size() && gates[i].type != 99) {
            need_of(gates[i], need);
            std::vector<unsigned> merged = pl.hbits;
            for (unsigned b : need) if (std::find(merged.begin(), merged.end(), b) == merged.end()) merged.push_back(b);
            if (merged.size() > budget) break;
            pl.hbits.swap(merged);
            i++;
        }
        pl.last = i;
        if (pl.last == pl.first) {               // a single gate that does not fit the tile budget: stand-alone
            const QGate &g = gates[i];
            int s = (g.type == FUSE_H) ? qcx_shard_hadamard(r->amp, n, g.q, r->stream)
                                       : reg_camodc(r, g.C, g.A, g.q);
            if (s != QCX_NO_ERROR) return s;
            i++;
            continue;
        }
        // one gate alone gains nothing from staging: use its tuned stand-alone kernel
        if (pl.last - pl.first == 1) {
            const QGate &g = gates[pl.first];
            int s;
            if (g.type == FUSE_H) s = qcx_shard_hadamard(r->amp, n, g.q, r->stream);
            else if (g.type == FUSE_PHASE) s = qcx_shard_phase(r->amp, n, g.mask, g.c, g.s, r->stream);
            else s = reg_camodc(r, g.C, g.A, g.q);
            if (s != QCX_NO_ERROR) return s;
            continue;
        }
        // pad the tile with the lowest free bits (longer contiguous runs) up to T bits
        std::sort(pl.hbits.begin(), pl.hbits.end());
        for (unsigned b = c; pl.hbits.size() < budget && b < n; b++)
            if (std::find(pl.hbits.begin(), pl.hbits.end(), b) == pl.hbits.end()) pl.hbits.push_back(b);
        std::sort(pl.hbits.begin(), pl.hbits.end());

        FusePass P;
        memset(&P, 0, sizeof P);
        P.c = c; P.nh = (uint32_t)pl.hbits.size(); P.T = c + P.nh;
        for (unsigned j = 0; j < P.nh; j++) P.hbit[j] = (uint8_t)pl.hbits[j];
        P.nops = (uint32_t)(pl.last - pl.first);

        const size_t nops = pl.last - pl.first;
        if (gq->h_cap < nops) {
            if (gq->h_ops) HIP_TRY(hipHostFree(gq->h_ops));
            gq->h_ops = nullptr; gq->h_cap = 0;
            HIP_TRY(hipHostMalloc(&gq->h_ops, nops * sizeof(FuseOp)));
            gq->h_cap = nops;
        }
        if (gq->d_cap < nops) {
            HIP_TRY(hipStreamSynchronize(r->stream));
            if (gq->d_ops) HIP_TRY(hipFree(gq->d_ops));
            gq->d_ops = nullptr; gq->d_cap = 0;
            HIP_TRY(hipMalloc(&gq->d_ops, nops * sizeof(FuseOp)));
            gq->d_cap = nops;
        }
        // the previous pass may still be reading d_ops/h_ops: passes are rare and long, a stream sync is cheap
        HIP_TRY(hipStreamSynchronize(r->stream));
        for (size_t k = 0; k < nops; k++) {
            const QGate &g = gates[pl.first + k];
            FuseOp &o = gq->h_ops[k];
            memset(&o, 0, sizeof o);
            o.type = g.type;
            if (g.type == FUSE_H) {
                o.a = g.q;
                if (g.q >= c) o.a = c + (unsigned)(std::find(pl.hbits.begin(), pl.hbits.end(), g.q) - pl.hbits.begin());
            } else if (g.type == FUSE_PHASE) {
                // split the control|target mask into tile-local bits (vector test) and outside bits (scalar test)
                uint32_t mloc = 0; uint64_t mext = 0;
                for (unsigned b = 0; b < n; b++) {
                    if (!((g.mask >> b) & 1)) continue;
                    if (b < c) mloc |= 1u << b;
                    else {
                        auto it = std::find(pl.hbits.begin(), pl.hbits.end(), b);
                        if (it != pl.hbits.end()) mloc |= 1u << (c + (unsigned)(it - pl.hbits.begin()));
                        else mext |= (uint64_t)1 << b;
                    }
                }
                o.a = mloc; o.mask = mext; o.c = g.c; o.s = g.s;
            } else {
                // control: tile-local position (+1) in bits 8.., or an outside bit tested against the tile base
                unsigned ctl_local_p1 = 0; uint64_t mext = 0;
                if (g.q < c) ctl_local_p1 = g.q + 1;
                else {
                    auto it = std::find(pl.hbits.begin(), pl.hbits.end(), g.q);
                    if (it != pl.hbits.end()) ctl_local_p1 = c + (unsigned)(it - pl.hbits.begin()) + 1;
                    else mext = (uint64_t)1 << g.q;
                }
                o.a = (unsigned)r->M | (ctl_local_p1 << 8);
                o.mask = mext;
                FuseCamExtra X;
                X.C = g.C; X.d = gcd_u32(g.A, g.C); X.Cd = g.C / X.d; X.inv = modinv_u32(g.A / X.d, X.Cd);
                memcpy(&o.c, &X, sizeof X);
            }
        }
        size_t nsend = nops;
        if (g_tune.fuse_rounds && P.T >= 10 && P.T <= 12) {
            // ROUNDS form: group the list into rounds of at most two distinct H bits (the round's register bits)
            std::vector<FuseOp> out, cur;
            std::vector<unsigned> rb;
            auto close_round = [&]() {
                if (cur.empty()) { rb.clear(); return; }
                for (unsigned b = P.T; rb.size() < 2 && b-- > 0;)
                    if (std::find(rb.begin(), rb.end(), b) == rb.end()) rb.push_back(b);
                std::sort(rb.begin(), rb.end());
                FuseOp hdr; memset(&hdr, 0, sizeof hdr);
                hdr.type = FUSE_ROUND; hdr.a = rb[0] | (rb[1] << 8); hdr.mask = cur.size();
                out.push_back(hdr);
                const uint32_t regmask = (1u << rb[0]) | (1u << rb[1]);
                const size_t hdr_at = out.size() - 1;
                size_t run_hdr = (size_t)-1; uint32_t run_rsel = 0;
                for (FuseOp o : cur) {
                    if (o.type == FUSE_H) { o.a = (o.a == rb[0]) ? 0u : 1u; run_hdr = (size_t)-1; out.push_back(o); continue; }
                    const uint32_t mr = o.a & regmask;
                    uint32_t rsel = 0;
                    for (unsigned q = 0; q < 4; q++) {
                        const uint32_t bits = ((q & 1u) << rb[0]) | ((q >> 1) << rb[1]);
                        if ((bits & mr) == mr) rsel |= 1u << q;
                    }
                    o.a &= ~regmask;
                    o.type = FUSE_PHASE | (rsel << 8);
                    if (g_tune.fuse_pruns) {
                        // consecutive phases that rotate the same registers form a run (branch-free kernel loop)
                        if (run_hdr == (size_t)-1 || rsel != run_rsel) {
                            FuseOp rh; memset(&rh, 0, sizeof rh);
                            rh.type = FUSE_PRUN; rh.a = rsel; rh.mask = 0;
                            run_hdr = out.size(); run_rsel = rsel;
                            out.push_back(rh);
                        }
                        out[run_hdr].mask++;
                    }
                    out.push_back(o);
                }
                out[hdr_at].mask = out.size() - 1 - hdr_at;          // the round spans everything emitted after its header
                cur.clear(); rb.clear();
            };
            for (size_t k2 = 0; k2 < nops; k2++) {
                const FuseOp &o = gq->h_ops[k2];
                if (o.type == FUSE_CAMODC) { close_round(); out.push_back(o); }
                else if (o.type == FUSE_H) {
                    if (std::find(rb.begin(), rb.end(), o.a) == rb.end()) {
                        if (rb.size() == 2) close_round();
                        rb.push_back(o.a);
                    }
                    cur.push_back(o);
                } else cur.push_back(o);
            }
            close_round();
            nsend = out.size();
            if (gq->h_cap < nsend) {
                HIP_TRY(hipHostFree(gq->h_ops)); gq->h_ops = nullptr; gq->h_cap = 0;
                HIP_TRY(hipHostMalloc(&gq->h_ops, nsend * sizeof(FuseOp))); gq->h_cap = nsend;
            }
            if (gq->d_cap < nsend) {
                HIP_TRY(hipFree(gq->d_ops)); gq->d_ops = nullptr; gq->d_cap = 0;
                HIP_TRY(hipMalloc(&gq->d_ops, nsend * sizeof(FuseOp))); gq->d_cap = nsend;
            }
            memcpy(gq->h_ops, out.data(), nsend * sizeof(FuseOp));
            P.nops = (uint32_t)nsend;
            P.cam_ctl_local[0] = 1;
        }
        HIP_TRY(hipMemcpyAsync(gq->d_ops, gq->h_ops, nsend * sizeof(FuseOp), hipMemcpyHostToDevice, r->stream));
        const uint64_t ntiles = (uint64_t)1 << (n - P.T);
        const unsigned grid = grid_for(ntiles, 1, g_tune.fuse_grid_cap);
        const size_t lut_bytes = ((size_t)2 << std::min(12u, (unsigned)r->M)) + 16;   // source table of a modular-multiply step
        const size_t lds = ((size_t)16 << P.T) + lut_bytes;
        // 4 amplitudes per thread (all loads of a tile in flight at once, few registers): block = 2^T / 4
#define QCX_FUSE_LAUNCH(B, TTv) do { \
            if (g_tune.fuse_pipe && ntiles >= 4096) { \
                const unsigned pg = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)g_tune.fuse_pipe_grid); \
                hipLaunchKernelGGL((k_fused_pipe<B, TTv>), dim3(pg), dim3(B), 2 * ((size_t)16 << P.T) + lut_bytes, r->stream, r->amp, n, P, gq->d_ops, ntiles); \
            } else if (g_tune.fuse_ldsdma) hipLaunchKernelGGL((k_fused<B, TTv, true>), dim3(grid), dim3(B), lds, r->stream, r->amp, n, P, gq->d_ops, ntiles); \
            else hipLaunchKernelGGL((k_fused<B, TTv, false>), dim3(grid), dim3(B), lds, r->stream, r->amp, n, P, gq->d_ops, ntiles); } while (0)
        switch (P.T) {
        case 12: QCX_FUSE_LAUNCH(1024, 12); break;
        case 11: QCX_FUSE_LAUNCH(512, 11); break;
        case 10: QCX_FUSE_LAUNCH(256, 10); break;
        case 9:  QCX_FUSE_LAUNCH(256, 9); break;
        default: hipLaunchKernelGGL((k_fused<256, 0, false>), dim3(grid), dim3(256), lds, r->stream, r->amp, n, P, gq->d_ops, ntiles); break;
        }
#undef QCX_FUSE_LAUNCH
        HIP_TRY(hipGetLastError());
        gq->passes_launched++;
        gq->gates_fused += nops;
    }
    return QCX_NO_ERROR;
}

static int fuse_push(qcx_register *r, const QGate &g)
{
    if (!r->queue) r->queue = new GateQueue();
    r->queue->gates.push_back(g);
    if (r->queue->gates.size() >= (size_t)g_tune.fuse_max_queue) return fuse_flush(r);
    return QCX_NO_ERROR;
}
