/*
 * qcx_shor.c -- C host driver for Shor's algorithm on the MI355X gate engine (libqcx.so).
 *
 * Same command line as the reference program (Q:1173-1264):  -C num -L L_reg_size -M M_reg_size
 * [-a trial_int] [-v] [-V]; additionally -f is accepted as a synonym of -a (the reference documents
 * -f but parses -a), -s seeds the MT19937 stream (the reference uses time(NULL)),
 * -Q turns on the reference's 32-bit INT_POW behaviour for differential runs, -j prints a one-line
 * JSON performance summary.  The circuit (qcx_quantum_computation) runs as fused passes by default; -G forces one kernel
 * launch per gate (qcx_set_fusion(reg, -1)), -F queues every gate call (qcx_set_fusion(reg, 1)); the results are the same
 * bits in all three modes; -T selects the opt-in tolerance mode (qcx_set_fusion(reg, 2): runs of controlled phases merged into
 * one diagonal, rounding-level differences in the amplitudes).  -o file writes the register's state after the LAST period-finding attempt (post-measurement,
 * i.e. collapsed) and -O file the state right after the last circuit, before measuring (qcx_state_save).  Exit code = the reference's ErrorCode (Q:164-170, Q:1340-1347).
 *
 * The quantum part (reset, circuit, measurement) runs on the GPU through include/qcx.h; everything
 * here is host-side control flow written from scratch after the behaviour of find_period
 * (Q:912-964) and shors_algorithm (Q:1003-1134).
 */
#define _POSIX_C_SOURCE 200809L
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../include/qcx.h"
#include "qcx_classical.h"

static bool verbose = false, very_verbose = false;

typedef struct {
    unsigned C, forced_a;
    int L, M;
    unsigned long seed;
    bool seed_given, ref_quirks, json, fusion, per_gate, tolerance;
    const char *dump_final, *dump_circuit;
    int gpus;                   /* -g N: shard the register over N GPUs (2, 4, 8, 16) from this one process */
    const char *gpu_list;       /* -d "0,0,1,1": HIP device of each shard (default: spread over the visible GPUs) */
} Options;

typedef struct {
    unsigned long gates;        /* gate kernels launched */
    unsigned long attempts;     /* period-finding attempts (circuit + measurement) */
} Stats;

static const char *USAGE =
    "Usage: qcx_shor -C num -L L_reg_size -M M_reg_size [-a trial_int | -f trial_int] [-v] [-V] [-s seed] [-Q] [-j] [-F | -G | -T] [-g gpus [-d dev,dev,...]] [-o state_file] [-O state_file]\n";

static double now_seconds(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + (double)t.tv_nsec * 1e-9;
}

static int parse_args(int argc, char **argv, Options *o)
{
    bool haveC = false, haveL = false, haveM = false;
    int ch;
    memset(o, 0, sizeof *o);
    while ((ch = getopt(argc, argv, "C:L:M:a:f:s:o:O:g:d:vVQjFGT")) != -1) {
        switch (ch) {
        case 'C': o->C = (unsigned)atoi(optarg); haveC = true; break;
        case 'L': o->L = atoi(optarg); haveL = true; break;
        case 'M': o->M = atoi(optarg); haveM = true; break;
        case 'a': case 'f': o->forced_a = (unsigned)atoi(optarg); break;
        case 's': o->seed = strtoul(optarg, NULL, 10); o->seed_given = true; break;
        case 'v': verbose = true; break;
        case 'V': verbose = very_verbose = true; break;
        case 'Q': o->ref_quirks = true; break;
        case 'j': o->json = true; break;
        case 'F': o->fusion = true; break;
        case 'G': o->per_gate = true; break;
        case 'T': o->tolerance = true; break;
        case 'g': o->gpus = atoi(optarg); break;
        case 'd': o->gpu_list = optarg; break;
        case 'o': o->dump_final = optarg; break;
        case 'O': o->dump_circuit = optarg; break;
        default: fputs(USAGE, stdout); return QCX_BAD_ARGUMENTS;
        }
    }
    if (!haveC) { fprintf(stderr, "Error: Number to be factorised 'C' not given.\n"); fputs(USAGE, stdout); return QCX_BAD_ARGUMENTS; }
    if (!haveL) { fprintf(stderr, "Error: Size of L register not given.\n"); fputs(USAGE, stdout); return QCX_BAD_ARGUMENTS; }
    if (!haveM) { fprintf(stderr, "Error: Size of M register not given.\n"); fputs(USAGE, stdout); return QCX_BAD_ARGUMENTS; }
    /* the reference only prints for these (Q:1240-1253); a register cannot be built from them, so stop */
    if (o->C < 2 || o->L <= 0 || o->M <= 0) {
        fprintf(stderr, "Error: C, L and M must be positive (C >= 2).\n");
        fputs(USAGE, stdout);
        return QCX_BAD_ARGUMENTS;
    }
    return QCX_NO_ERROR;
}

/* Q:340-351 */
static void issue_warnings(unsigned C, int L, int M)
{
    if ((M < 32 ? (1ULL << M) : ~0ULL) < C) {
        int need = 0;
        while ((1ULL << need) < C) need++;
        printf(" --- *WARNING* The M register is not large enough for reliable results. Ensure 2^M >= C. Minimum: M = %d.\n", need);
    }
    if ((L < 63 ? (1ULL << L) : ~0ULL) < (unsigned long long)C * C) {
        int need = 0;
        while ((1ULL << need) < (unsigned long long)C * C) need++;
        printf(" --- *WARNING* The L register is not large enough for full confidence in finding the period. "
               "Ensure 2^L >= C^2 for confidence. Suggested: L = %d.\n", need);
    }
}

/* one period-finding attempt: reset, circuit, measure on the GPU; continued fractions on the host */
static int find_period(unsigned *period, unsigned C, unsigned a, qcx_register *reg, qcx_rng *rng,
                       const Options *o, Stats *st)
{
    unsigned long state = 0;
    int s;
    if (very_verbose) printf("      - Performing quantum computation...\n");
    if ((s = qcx_reset_register(reg)) != QCX_NO_ERROR) return s;
    if (very_verbose) printf("         - Applying Hadamard matrices, a^x mod (C) gates, inverse quantum Fourier transform.\n");
    if ((s = qcx_quantum_computation(C, a, o->ref_quirks ? 1 : 0, reg)) != QCX_NO_ERROR) return s;
    st->gates += 3UL * (unsigned long)o->L + (unsigned long)o->L * (unsigned long)(o->L - 1) / 2;
    st->attempts++;
    if (o->dump_circuit && (s = qcx_state_save(reg, o->dump_circuit)) != QCX_NO_ERROR) return s;
    if (very_verbose) printf("      - Measuring state...\n");
    if ((s = qcx_measure_state(reg, rng, &state)) != QCX_NO_ERROR) return s;
    const double omega = qcx_read_omega(state, o->L, o->M);
    if (very_verbose) printf("      - Measured state %lu, x~ = %u, omega = %.10f; using continued fractions to guess period...\n",
                             state, qcx_read_x_tilde(state, o->L, o->M), omega);
    *period = qcx_period_from_omega(omega, a, C, o->ref_quirks ? 1 : 0);
    return *period ? QCX_NO_ERROR : QCX_PERIOD_NOT_FOUND;
}

static int try_trial_integer(unsigned a, unsigned factors[2], const Options *o, qcx_register *reg, qcx_rng *rng,
                             Stats *st, bool forced)
{
    unsigned period = 0;
    const char *tail = forced ? "\n" : "\n\n";
    int s = find_period(&period, o->C, a, reg, rng, o, st);
    if (s == QCX_PERIOD_NOT_FOUND) {
        if (verbose && !forced) printf(" --- A valid period could not be found for a = %u.\n\n", a);
        return QCX_PERIOD_NOT_FOUND;
    }
    if (s != QCX_NO_ERROR) return s;
    if (qcx_factors_from_period(a, period, o->C, o->ref_quirks ? 1 : 0, factors) != 0) {
        if (verbose) printf(" --- Period was found to be %u, but it did not pass the validity requirements.%s", period, tail);
        return QCX_PERIOD_NOT_FOUND;
    }
    if (verbose)
        printf(" --- A valid period = %u has been found so the factors of C = %u have been found quantum mechanically.\n\n", period, o->C);
    return QCX_NO_ERROR;
}

/* Q:1003-1134: one forced trial integer, or a = 2 .. C-2 until non-trivial factors appear */
static int shors_algorithm(unsigned factors[2], const Options *o, qcx_register *reg, qcx_rng *rng, Stats *st)
{
    printf("\n --- Finding factors...\n\n");
    if (o->forced_a != 0) {
        if (verbose) printf(" --- Forced trial integer a = %u, finding period ...\n", o->forced_a);
        int s = try_trial_integer(o->forced_a, factors, o, reg, rng, st, true);
        if (s == QCX_PERIOD_NOT_FOUND) {
            printf(" --- A valid period was not found and hence C = %u could not be factorised.\n", o->C);
            return s;
        }
        if (s != QCX_NO_ERROR) return s;
        if (factors[0] == 1 || factors[1] == 1)
            printf(" --- The factors found are trivial, consider trying a different trial integer.\n");
        return QCX_NO_ERROR;
    }
    for (unsigned a = 2; a + 1 < o->C; a++) {
        if (verbose) printf(" --- Trial integer a = %u, finding period ...\n", a);
        int s = try_trial_integer(a, factors, o, reg, rng, st, false);
        if (s == QCX_PERIOD_NOT_FOUND) continue;
        if (s != QCX_NO_ERROR) return s;
        if (factors[0] == 1 || factors[1] == 1) {
            printf(" --- Factors found are trivial. Continuing to find non-trivial factors.\n");
            continue;
        }
        return QCX_NO_ERROR;
    }
    printf(" --- A valid period was not found and hence C = %u could not be factorised.\n", o->C);
    return QCX_PERIOD_NOT_FOUND;
}

int main(int argc, char **argv)
{
    Options o;
    Stats st = {0, 0};
    unsigned factors[2] = {0, 0};
    int s = parse_args(argc, argv, &o);
    if (s != QCX_NO_ERROR) return s;

    qcx_rng *rng = qcx_rng_alloc();
    if (!rng) { fprintf(stderr, "Error: Insufficient memory.\n"); return QCX_INSUFFICIENT_MEMORY; }
    qcx_rng_set(rng, o.seed_given ? o.seed : (unsigned long)time(NULL));       /* Q:1299 */

    issue_warnings(o.C, o.L, o.M);

    qcx_register *reg = NULL;                                                    /* Q:1316-1324 */
    if (o.gpus > 1) {                       /* one process, N shards: the top log2 N qubits select the GPU */
        int devs[16], nd = 0;
        for (const char *p = o.gpu_list; p && *p && nd < 16; nd++) { devs[nd] = atoi(p); while (*p && *p != ',') p++; if (*p == ',') p++; }
        for (int i = nd; nd > 0 && i < 16; i++) devs[i] = devs[nd - 1];
        /* no -d: the library spreads the shards over the visible GPUs (and checks the exchange between them first) */
        s = qcx_register_create_sharded(o.L, o.M, (unsigned)o.gpus, nd ? devs : NULL, &reg);
    } else
        s = qcx_register_create(o.L, o.M, &reg);
    if (s != QCX_NO_ERROR) {
        fprintf(stderr, "Error: could not create the %d-qubit register on the GPU: %s (%s).\n", o.L + o.M, qcx_status_string(s), qcx_last_error());
        qcx_rng_free(rng);
        return s == QCX_INSUFFICIENT_MEMORY ? QCX_INSUFFICIENT_MEMORY : QCX_UNKNOWN_ERROR;
    }

    if (o.fusion) qcx_set_fusion(reg, 1);        /* -F: every gate call is queued and run as fused passes (same bits) */
    if (o.per_gate) qcx_set_fusion(reg, -1);     /* -G: one kernel launch per gate, also inside the circuit call */
    if (o.tolerance) qcx_set_fusion(reg, 2);     /* -T: opt-in tolerance mode (merged diagonals; amplitudes to ~1e-15, not bit-exact) */
    const double t0 = now_seconds();
    s = shors_algorithm(factors, &o, reg, rng, &st);
    qcx_synchronize(reg);
    const double dt = now_seconds() - t0;
    if (verbose) printf(" --- Time to run Shor's Algorithm: %.6fs.\n", dt);

    if (o.dump_final) {
        const int sd = qcx_state_save(reg, o.dump_final);
        if (sd != QCX_NO_ERROR) fprintf(stderr, "Error: could not write %s: %s.\n", o.dump_final, qcx_last_error());
    }
    if (o.json) {
        const double dim = (double)qcx_num_states(reg);
        unsigned long exchanges = 0;
        qcx_sharded_stats(reg, &exchanges, NULL);
        printf("{\"C\": %u, \"L\": %d, \"M\": %d, \"qubits\": %d, \"attempts\": %lu, \"gates\": %lu, \"seconds\": %.6f, "
               "\"amplitude_updates_per_s\": %.6e, \"shards\": %u, \"exchanges\": %lu, \"status\": %d}\n",
               o.C, o.L, o.M, o.L + o.M, st.attempts, st.gates, dt, dt > 0 ? (double)st.gates * dim / dt : 0.0,
               qcx_register_shards(reg), exchanges, s);
    }
    qcx_register_destroy(reg);                                                   /* Q:1330-1333 */
    qcx_rng_free(rng);

    if (s == QCX_NO_ERROR) {
        printf(" --- Factors of %u found: (%u, %u).\n", o.C, factors[0], factors[1]);
        if (factors[0] == 0 || o.C / factors[0] != factors[1])
            printf(" --- These factors are incorrect. Consider increasing register sizes as per the warnings.\n");
        return QCX_NO_ERROR;
    }
    if (s == QCX_PERIOD_NOT_FOUND) return QCX_PERIOD_NOT_FOUND;
    fprintf(stderr, "Error: %s.\n", qcx_status_string(s));
    return QCX_UNKNOWN_ERROR;
}
