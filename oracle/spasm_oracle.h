/*
 * oracle/spasm_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded restatement of the reference (cbouilla/spasm)
 * algorithms that sit on the echelonization hot path.  It exists so that the
 * HIP path can be checked bit-for-bit on the same inputs.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (spasm_amd/csrc) never links or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function here is cross-checked against
 *   (1) the real reference compiled from /root/reference/src by
 *       oracle/Makefile into oracle/_ref/libspasm_ref.so (all files except
 *       spasm_echelonize.c and spasm_ffpack.cpp, which need FFLAS-FFPACK /
 *       Givaro and are unbuildable in this image), and
 *   (2) the reference's own test matrices (tests/golden/Matrix, *.sms) under
 *       the reference's moduli {3, 257, 65537, 67108859, 189812507,
 *       4294967291} and the reference's test properties (tests/schur.c,
 *       tests/schur_dense.c, tests/echelonize.c, tests/dense_rref_ffpack.c,
 *       tests/GFp.c, tests/Expected/prng).
 * The dense RREF (FFLAS-FFPACK pReducedRowEchelonForm, third-party, version
 * unpinned by the reference: "Debian package fflas-ffpack") is restated from
 * its published contract (reduced row echelon form, column rank profile
 * pivots) and pinned through the reference's tests/dense_rref_ffpack.c
 * property plus uniqueness of the RREF.
 */
#ifndef SPASM_ORACLE_H
#define SPASM_ORACLE_H

#include <stdint.h>

typedef int32_t orc_zp;        /* balanced representative, as spasm_ZZp (spasm.h:27) */

typedef struct {               /* same field meaning as struct spasm_csr (spasm.h:37-50) */
	int64_t nzmax;
	int n;                 /* rows */
	int m;                 /* columns */
	int64_t *p;            /* row pointers, n+1 */
	int *j;                /* column indices */
	orc_zp *x;             /* values */
	int64_t prime;
} orc_csr;

typedef struct {               /* the part of struct spasm_lu (spasm.h:63-71) the path uses */
	orc_csr *U;
	int *qinv;             /* size m: row of U holding the pivot of column j, or -1 */
	int r;
	/* optional L as triplets (i, j, x): row i of input, column = pivot index */
	int64_t lnz, lnzmax;
	int *Li, *Lj;
	orc_zp *Lx;
	int *Lp;               /* pivot j of L sits on (original) row Lp[j] */
	int want_L;
} orc_lu;

typedef struct {               /* struct echelonize_opts (spasm.h:84-108) */
	int enable_greedy_pivot_search;
	int enable_tall_and_skinny;
	int enable_dense;
	int enable_GPLU;
	int L;
	int complete;
	double min_pivot_proportion;
	int max_round;
	double sparsity_threshold;
	int dense_block_size;
	double low_rank_ratio;
	double tall_and_skinny_ratio;
	double low_rank_start_weight;
} orc_opts;

/* --- field arithmetic (spasm_ZZp.c) --- */
orc_zp orc_zp_init(int64_t p, int64_t x);
orc_zp orc_zp_add(int64_t p, orc_zp a, orc_zp b);
orc_zp orc_zp_sub(int64_t p, orc_zp a, orc_zp b);
orc_zp orc_zp_mul(int64_t p, orc_zp a, orc_zp b);
orc_zp orc_zp_axpy(int64_t p, orc_zp a, orc_zp x, orc_zp y);
orc_zp orc_zp_inverse(int64_t p, orc_zp a);

/* --- containers --- */
orc_csr *orc_csr_alloc(int n, int m, int64_t nzmax, int64_t prime);
void orc_csr_free(orc_csr *A);
orc_csr *orc_compress(int64_t prime, int n, int m, int64_t nz,
                      const int *Ti, const int *Tj, const int64_t *Tx);
orc_csr *orc_transpose(const orc_csr *A);
orc_lu *orc_lu_alloc(int n, int m, int64_t nzmax, int64_t prime, int want_L);
void orc_lu_free(orc_lu *F);

/* --- sparse triangular solve (spasm_scatter.c, spasm_reach.c, spasm_triangular.c) --- */
void orc_scatter(const orc_csr *A, int i, orc_zp beta, orc_zp *x);
int orc_reach(const orc_csr *U, const orc_csr *B, int k, int *xj, const int *qinv);
int orc_sparse_triangular_solve(const orc_csr *U, const orc_csr *B, int k,
                                int *xj, orc_zp *x, const int *qinv);

/* --- pivot search (spasm_pivots.c), single-thread semantics --- */
int orc_pivots_extract_structural(const orc_csr *A, const int *p_in, orc_lu *F,
                                  int *p, int enable_greedy);

/* --- Schur complement (spasm_schur.c) --- */
double orc_schur_estimate_density(const orc_csr *A, const int *p, int n,
                                  const orc_csr *U, const int *qinv, int R, unsigned seed);
orc_csr *orc_schur(const orc_csr *A, const int *p, int n, orc_lu *F,
                   const int *p_in, int *p_out);
void orc_schur_dense(const orc_csr *A, const int *p, int n, const int *p_in,
                     orc_lu *F, int64_t *S, int *q, int *p_out);

/* --- dense tail (spasm_ffpack.cpp contract) --- */
int orc_dense_rref(int64_t prime, int n, int m, int64_t *A, int ldA, int64_t *qinv);

/* --- driver (spasm_echelonize.c) --- */
void orc_opts_init(orc_opts *o);
orc_lu *orc_echelonize(const orc_csr *A, const orc_opts *opts);
orc_csr *orc_rref(const orc_lu *F, int *Rqinv);

#endif
