/*
 * qcx_oracle.h -- CPU ORACLE for the gate-application hot path of qc_shor.c.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (adamalderton/QuantumComputer, /root/reference/qc_shor.c, cited
 * below as Q:line).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (libqcx.so, quantumcomputer_amd/,
 * host/) never links, imports or executes anything under oracle/.
 *
 * PARITY UNPINNED.  The reference cannot be compiled in this image (it needs
 * GSL 2.6, which is absent; a stand-in for a missing library is not allowed),
 * so there is no oracle/_ref, and the reference holds no golden vectors or
 * tests of its own for this path.  What the oracle IS checked against
 * (tests/test_oracle_pinning.py, tests/test_independent_derivation.py):
 *   - the reference's published known answers (Q:25-29, Q:78-79, report
 *     Table I and SIV.A) -- scenarios, not vectors;
 *   - an independent second derivation: dense Kronecker-product operators
 *     built from the base matrices of Q:210-225 (agreement to 1e-13), the
 *     DFT-matrix identity of the inverse-QFT schedule, numpy's running sum
 *     for the measurement rule; a Python-float restatement of Q:393-413 that the
 *     oracle equals bit for bit (rounding order independent of any C compiler);
 *   - tests/golden/survey_appendix_c.json: values the survey session read off
 *     a build of qc_shor.c against a GSL stand-in; no generator is committed,
 *     so by this project's rules they pin nothing -- kept as regression values.
 * See DESIGN.md "Oracle and pinning".
 *
 * Two forms of every gate are provided:
 *   orc_lit_*   the LITERAL algorithm: scan index pairs, build a COO sparse
 *               matrix, COO mat-vec into the second buffer, swap  (Q:370-660).
 *               O(4^n) per gate for H / CPHASE, so only usable for n <= ~12.
 *   orc_pair_*  the same arithmetic, same rounding, applied in place on
 *               amplitude pairs / quarters; usable at any n and the form the
 *               HIP kernels are compared with.  tests/ prove lit == pair
 *               bit for bit on every size the literal form can reach.
 */
#ifndef QCX_ORACLE_H
#define QCX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- MT19937, GSL 2.6 flavour (gsl_rng_mt19937; Q:1296-1299, Q:281) ---- */
typedef struct {
    uint32_t mt[624];
    int      idx;
} orc_rng;

void     orc_rng_set(orc_rng *g, uint32_t seed);   /* gsl_rng_set: seed 0 -> 4357 */
uint32_t orc_rng_get(orc_rng *g);                  /* tempered 32-bit output */
double   orc_rng_uniform(orc_rng *g);              /* get / 4294967296.0 */

/* ---- COO container with the GSL triplet semantics the reference relies on
 * (Q:385-390 reads ->i rows, ->p cols, ->data interleaved, ->nz). ---------- */
typedef struct {
    int32_t *row;
    int32_t *col;
    double  *val;      /* interleaved re, im */
    size_t   nz;
    size_t   cap;
    int      keep_zeros;   /* 1: explicit zeros are stored (numerically neutral) */
} orc_coo;

int  orc_coo_init(orc_coo *m, size_t cap, int keep_zeros);
void orc_coo_free(orc_coo *m);

/* ---- register: double-buffered state like Q:194-203 ---------------------- */
typedef struct {
    int       L, M;
    unsigned  n;
    uint64_t  dim;
    double   *buf[2];   /* interleaved re, im; 2*dim doubles each */
    int       cur;      /* which buffer is "current_state" */
} orc_reg;

int     orc_reg_init(orc_reg *r, int L, int M);       /* allocs both buffers */
void    orc_reg_free(orc_reg *r);
double *orc_reg_state(orc_reg *r);                    /* current buffer */

/* ---- literal gate layer (Q:318-324, Q:370-420, Q:442-484, Q:513-565, Q:595-660) */
void orc_lit_reset(orc_reg *r);
void orc_lit_hadamard(unsigned q, orc_reg *r, orc_coo *m);
void orc_lit_cphase(unsigned c, unsigned t, double theta, orc_reg *r, orc_coo *m);
void orc_lit_camodc(unsigned C, unsigned long long atox, unsigned ctl, orc_reg *r, orc_coo *m);
void orc_lit_iqft(orc_reg *r, orc_coo *m);                                 /* Q:678-690 */
void orc_lit_quantum_computation(unsigned C, unsigned a, int ref_intpow,
                                 orc_reg *r, orc_coo *m);                  /* Q:712-737 */

/* "T2" tier of BASELINE.md: the reference's mat-vec loop only (Q:396-413),
 * with the Hadamard COO built in O(2^n) in the same row-major order. */
void orc_spmv_hadamard(unsigned q, orc_reg *r, orc_coo *m);

/* ---- pairwise in-place layer (amp = interleaved re,im, 2^n amplitudes) ---- */
void orc_pair_reset(double *amp, unsigned n);
void orc_pair_hadamard(double *amp, unsigned n, unsigned q, int threads);
void orc_pair_cphase(double *amp, unsigned n, unsigned c, unsigned t, double theta, int threads);
/* scratch: 2*2^M doubles (caller) or NULL to malloc */
void orc_pair_camodc(double *amp, unsigned n, unsigned M, unsigned C,
                     unsigned long long atox, unsigned ctl, int threads);
void orc_pair_iqft(double *amp, unsigned n, unsigned M, int threads);
void orc_pair_quantum_computation(double *amp, unsigned n, unsigned M, unsigned C,
                                  unsigned a, int ref_intpow, int threads);

/* measure (Q:272-306): sequential cumulative sum, first index with cum >= r,
 * fall through to dim-1; collapses the state.  Returns the index. */
uint64_t orc_measure(double *amp, unsigned n, double r);
/* same decision without the collapse, starting from a carried-in cumulative
 * value over [first, first+count) of a larger vector (sharded form).
 * returns 1 and sets *idx if crossed inside the range; *cum_out = running sum */
int orc_measure_range(const double *amp, uint64_t first, uint64_t count,
                      uint64_t last_excluded, double cum_in, double r,
                      uint64_t *idx, double *cum_out);
double orc_norm2(const double *amp, unsigned n);        /* T:28-37, sequential */

/* CPU twin of the product's device-side synthetic-state generator */
void orc_fill_random(double *amp, uint64_t first, uint64_t count, uint64_t seed, double scale);

/* whole circuits on a basis-state input, evaluated per output index (no 2^n array): tests at n = 28 / 30 */
void orc_basis_iqft_window(uint64_t x, unsigned n, unsigned M, uint64_t first, uint64_t count, double *out);
void orc_shor_front_window(unsigned n, unsigned M, unsigned C, unsigned a, int ref_intpow,
                           uint64_t first, uint64_t count, double *out);
void orc_polar(double theta, double *re, double *im);   /* gsl_complex_polar(1, theta) as gcc -O2 + glibc evaluate it */

/* ---- host-side scalar helpers restated from the reference ---------------- */
unsigned orc_ref_intpow(double base, double power);     /* Q:158-159 incl. x86-64 wrap */
unsigned long long orc_modpow(unsigned long long a, unsigned long long e, unsigned long long m);
unsigned orc_gcd(unsigned a, unsigned b);                               /* Q:756-779 */
void     orc_cf_denominators(double omega, unsigned count, unsigned *den); /* Q:806-846 */
double   orc_read_omega(uint64_t state, int L, int M);                  /* Q:868-883 */

#ifdef __cplusplus
}
#endif
#endif
