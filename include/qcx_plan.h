/*
 * qcx_plan.h -- the fused-pass PLANNER of libqcx.so on its own, host code only (no GPU needed).  A test and tooling
 * interface: the CPU-only suite plans gate lists, interprets the returned records with a numpy restatement of the pass
 * kernels (tests/fuse_emulator.py) and compares with the oracle; tools/plan_cost.py prices plans.  Nothing a user of the
 * gate engine needs.
 */
#ifndef QCX_PLAN_H
#define QCX_PLAN_H

#include <stddef.h>
#include <stdint.h>
#include "qcx_shard.h"          /* qcx_gate_desc */

#ifdef __cplusplus
extern "C" {
#endif

/* The pass planner alone, on the host (no GPU needed; test and tooling interface).  Cuts a gate list into actions --
 * fused passes over LDS tiles, or single gates that run as their stand-alone kernel -- and returns the records the
 * pass kernels interpret, 32 bytes each, all passes back to back.  Record formats: csrc/qcx_kernels.h (FuseOp and
 * the ROUNDS form); tests/fuse_emulator.py interprets them on the CPU and compares with the oracle.
 * Returns QCX_INSUFFICIENT_MEMORY (with the needed counts in n_actions / n_records) when an array is too small. */
typedef struct {
    uint32_t type, a;
    uint64_t mask;
    double   c, s;
} qcx_fuse_record;
typedef struct {
    int      fused;                 /* 0: the single gate first_gate, stand-alone kernel; 1: one fused pass */
    unsigned first_gate, ngates;    /* the gates of the list this action covers (in order, no gaps between actions) */
    unsigned T, c, nh;              /* tile = 2^T amplitudes: the c lowest index bits + nh higher bits hbit[0..nh) */
    unsigned char hbit[16];
    unsigned nopipe;                /* 1: phase-dominated pass, planned on the smaller tile (fuse_T_phase) */
    unsigned rounds_form;           /* 1: records in ROUNDS form (rounds / items / runs), 0: plain gate list */
    size_t   rec_off, rec_cnt;      /* this pass's records (tables included) inside `records` */
    unsigned nops;                  /* records the kernel walks (the rest, if any, are tables) */
    unsigned table_bytes, table_rec_off;    /* folded modular-multiply tables: size, and offset in records from rec_off */
    unsigned diag_cnt, diag_rec_off;        /* tolerance mode: merged diagonals of the pass, record offset of their table area */
    /* tile addressing (round 4).  tl[j] = the qubit that is tile-local bit j in the records.  A CHAINED pass (chained = 1) reads
     * the register's current buffer under one logical -> physical layout and writes the other buffer under another one: tile-local
     * bit j is input index bit in_pos[j]; the j-th lowest output position of the tile's bits is output index bit st_pos[j] and
     * belongs to tile-local bit st_loc[j]; bits [src, src + len) of the tile number go to index bits [dst, dst + len) of the
     * input / output / logical index (seg_in / seg_out / seg_lg).  In place (chained = 0): out = in = logical, seg_in only. */
    unsigned chained;
    unsigned char tl[16], in_pos[16], st_loc[16], st_pos[16];
    unsigned char nseg_in, nseg_out, nseg_lg, pad_;
    struct { unsigned char src, dst, len, pad; } seg_in[16], seg_out[16], seg_lg[16];
} qcx_plan_action;
int  qcx_fusion_plan(unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates,
                     qcx_plan_action *actions, unsigned max_actions, unsigned *n_actions,
                     qcx_fuse_record *records, size_t max_records, size_t *n_records);
/* the same for a fusion mode: 1 = the bit-exact plan (what qcx_fusion_plan returns), 2 = the tolerance mode's plan;
 * | 4: as a register with a second buffer plans (runs of passes chained through it, see qcx_plan_action) */
int  qcx_fusion_plan_mode(int mode, unsigned n_local, unsigned M, unsigned count, const qcx_gate_desc *gates,
                          qcx_plan_action *actions, unsigned max_actions, unsigned *n_actions,
                          qcx_fuse_record *records, size_t max_records, size_t *n_records);
#ifdef __cplusplus
}
#endif
#endif /* QCX_PLAN_H */
