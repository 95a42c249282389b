/*
 * oracle/spasm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see spasm_oracle.h).
 *
 * Single-threaded CPU restatement of the cbouilla/spasm echelonization path.
 * Each function cites the reference file:line whose behaviour it restates.
 * Arithmetic is done with exact 64-bit integers (the reference goes through
 * a double-precision quotient estimate); results are the same balanced
 * representatives, which tests/test_oracle.py pins against the compiled
 * reference (oracle/_ref/libspasm_ref.so).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <assert.h>
#include "spasm_oracle.h"

static void *xmalloc(int64_t sz)
{
	void *q = malloc(sz > 0 ? (size_t) sz : 1);
	if (q == NULL) {
		fprintf(stderr, "[oracle] out of memory (%lld bytes)\n", (long long) sz);
		abort();
	}
	return q;
}

static void *xrealloc(void *old, int64_t sz)
{
	void *q = realloc(old, sz > 0 ? (size_t) sz : 1);
	if (q == NULL) {
		fprintf(stderr, "[oracle] out of memory (%lld bytes)\n", (long long) sz);
		abort();
	}
	return q;
}

/* ------------------------------------------------------------------ */
/* GF(p), balanced representatives.  spasm_ZZp.c:5-24 (field init +   */
/* NORMALISE): representatives live in [p/2 - p + 1, p/2].            */
/* ------------------------------------------------------------------ */
static inline orc_zp balance(int64_t p, int64_t r)
{
	int64_t hi = p / 2;
	int64_t lo = p / 2 - p + 1;
	if (r < lo)
		r += p;
	else if (r > hi)
		r -= p;
	return (orc_zp) r;
}

orc_zp orc_zp_init(int64_t p, int64_t x)          /* spasm_ZZp.c:26-30 */
{
	return balance(p, x % p);
}

orc_zp orc_zp_add(int64_t p, orc_zp a, orc_zp b)  /* spasm_ZZp.c:32-35 */
{
	return balance(p, (int64_t) a + (int64_t) b);
}

orc_zp orc_zp_sub(int64_t p, orc_zp a, orc_zp b)  /* spasm_ZZp.c:37-40 */
{
	return balance(p, (int64_t) a - (int64_t) b);
}

orc_zp orc_zp_mul(int64_t p, orc_zp a, orc_zp b)  /* spasm_ZZp.c:42-46 */
{
	return balance(p, ((int64_t) a * (int64_t) b) % p);
}

orc_zp orc_zp_axpy(int64_t p, orc_zp a, orc_zp x, orc_zp y)   /* spasm_ZZp.c:76-83: a*x + y */
{
	return balance(p, ((int64_t) a * (int64_t) x + (int64_t) y) % p);
}

orc_zp orc_zp_inverse(int64_t p, orc_zp a)        /* spasm_ZZp.c:48-73 (extended Euclid) */
{
	int64_t v = a;
	if (v < 0)
		v += p;
	/* invariants: r0 == t0 * v (mod p), r1 == t1 * v (mod p) */
	int64_t r0 = p, r1 = v, t0 = 0, t1 = 1;
	while (r1 != 0) {
		int64_t q = r0 / r1;
		int64_t r2 = r0 - q * r1;
		int64_t t2 = t0 - q * t1;
		r0 = r1; r1 = r2;
		t0 = t1; t1 = t2;
	}
	return balance(p, t0);
}

/* ------------------------------------------------------------------ */
/* containers (spasm_util.c:85-99, 177-191; spasm_triplet.c)           */
/* ------------------------------------------------------------------ */
orc_csr *orc_csr_alloc(int n, int m, int64_t nzmax, int64_t prime)
{
	orc_csr *A = xmalloc(sizeof(*A));
	A->n = n;
	A->m = m;
	A->nzmax = nzmax;
	A->prime = prime;
	A->p = xmalloc(((int64_t) n + 1) * sizeof(int64_t));
	A->j = xmalloc(nzmax * sizeof(int));
	A->x = xmalloc(nzmax * sizeof(orc_zp));
	A->p[0] = 0;
	return A;
}

void orc_csr_free(orc_csr *A)
{
	if (A == NULL)
		return;
	free(A->p);
	free(A->j);
	free(A->x);
	free(A);
}

static void csr_reserve(orc_csr *A, int64_t nzmax)
{
	if (nzmax <= A->nzmax)
		return;
	A->j = xrealloc(A->j, nzmax * sizeof(int));
	A->x = xrealloc(A->x, nzmax * sizeof(orc_zp));
	A->nzmax = nzmax;
}

/*
 * Triplets -> CSR.  Restates spasm_add_entry (spasm_triplet.c:7-23: values
 * reduced on entry, zero values dropped, dimensions grow to fit),
 * spasm_compress (:108-165: stable bucket by row), deduplicate (:59-94:
 * later duplicates are summed into the first occurrence) and
 * remove_explicit_zeroes (:34-56).
 */
orc_csr *orc_compress(int64_t prime, int n, int m, int64_t nz,
                      const int *Ti, const int *Tj, const int64_t *Tx)
{
	int64_t kept = 0;
	int *ri = xmalloc(nz * sizeof(int));
	int *rj = xmalloc(nz * sizeof(int));
	orc_zp *rx = xmalloc(nz * sizeof(orc_zp));
	for (int64_t k = 0; k < nz; k++) {
		orc_zp v = orc_zp_init(prime, Tx[k]);
		if (v == 0)
			continue;
		ri[kept] = Ti[k];
		rj[kept] = Tj[k];
		rx[kept] = v;
		if (Ti[k] + 1 > n) n = Ti[k] + 1;
		if (Tj[k] + 1 > m) m = Tj[k] + 1;
		kept += 1;
	}
	orc_csr *C = orc_csr_alloc(n, m, kept, prime);
	int64_t *cnt = xmalloc(((int64_t) n + 1) * sizeof(int64_t));
	for (int i = 0; i <= n; i++)
		cnt[i] = 0;
	for (int64_t k = 0; k < kept; k++)
		cnt[ri[k] + 1] += 1;
	for (int i = 0; i < n; i++)
		cnt[i + 1] += cnt[i];
	for (int i = 0; i <= n; i++)
		C->p[i] = cnt[i];
	for (int64_t k = 0; k < kept; k++) {
		int64_t dst = cnt[ri[k]]++;
		C->j[dst] = rj[k];
		C->x[dst] = rx[k];
	}
	free(cnt);
	free(ri);
	free(rj);
	free(rx);

	/* merge duplicates inside each row, then drop the zeros this creates */
	int64_t *where = xmalloc((int64_t) m * sizeof(int64_t));
	for (int j = 0; j < m; j++)
		where[j] = -1;
	int64_t out = 0;
	for (int i = 0; i < n; i++) {
		int64_t row_start = out;
		int64_t lo = C->p[i], hi = C->p[i + 1];
		for (int64_t px = lo; px < hi; px++) {
			int j = C->j[px];
			if (where[j] < row_start) {
				where[j] = out;
				C->j[out] = j;
				C->x[out] = C->x[px];
				out += 1;
			} else {
				int64_t q = where[j];
				C->x[q] = orc_zp_add(prime, C->x[q], C->x[px]);
			}
		}
		C->p[i] = row_start;
	}
	C->p[n] = out;
	free(where);
	int64_t nnz = 0;
	for (int i = 0; i < n; i++) {
		int64_t lo = C->p[i], hi = C->p[i + 1];
		C->p[i] = nnz;
		for (int64_t px = lo; px < hi; px++) {
			if (C->x[px] == 0)
				continue;
			C->j[nnz] = C->j[px];
			C->x[nnz] = C->x[px];
			nnz += 1;
		}
	}
	C->p[n] = nnz;
	return C;
}

orc_csr *orc_transpose(const orc_csr *A)          /* spasm_transpose.c:5-52 */
{
	int n = A->n, m = A->m;
	int64_t nnz = A->p[n];
	orc_csr *T = orc_csr_alloc(m, n, nnz, A->prime);
	int64_t *w = xmalloc(((int64_t) m + 1) * sizeof(int64_t));
	for (int j = 0; j <= m; j++)
		w[j] = 0;
	for (int64_t px = 0; px < nnz; px++)
		w[A->j[px] + 1] += 1;
	for (int j = 0; j < m; j++)
		w[j + 1] += w[j];
	for (int j = 0; j <= m; j++)
		T->p[j] = w[j];
	for (int i = 0; i < n; i++)
		for (int64_t px = A->p[i]; px < A->p[i + 1]; px++) {
			int64_t dst = w[A->j[px]]++;
			T->j[dst] = i;
			T->x[dst] = A->x[px];
		}
	free(w);
	return T;
}

orc_lu *orc_lu_alloc(int n, int m, int64_t nzmax, int64_t prime, int want_L)
{
	orc_lu *F = xmalloc(sizeof(*F));
	F->U = orc_csr_alloc(n, m, nzmax, prime);
	F->U->n = 0;
	F->qinv = xmalloc((int64_t) m * sizeof(int));
	for (int j = 0; j < m; j++)
		F->qinv[j] = -1;
	F->r = 0;
	F->want_L = want_L;
	F->lnz = 0;
	F->lnzmax = want_L ? nzmax : 0;
	F->Li = want_L ? xmalloc(F->lnzmax * sizeof(int)) : NULL;
	F->Lj = want_L ? xmalloc(F->lnzmax * sizeof(int)) : NULL;
	F->Lx = want_L ? xmalloc(F->lnzmax * sizeof(orc_zp)) : NULL;
	F->Lp = want_L ? xmalloc((int64_t) n * sizeof(int)) : NULL;
	if (want_L)
		for (int i = 0; i < n; i++)
			F->Lp[i] = -1;
	return F;
}

void orc_lu_free(orc_lu *F)
{
	if (F == NULL)
		return;
	orc_csr_free(F->U);
	free(F->qinv);
	free(F->Li);
	free(F->Lj);
	free(F->Lx);
	free(F->Lp);
	free(F);
}

static void L_push(orc_lu *F, int i, int j, orc_zp x)
{
	if (F->lnz == F->lnzmax) {
		F->lnzmax = 2 * F->lnzmax + 16;
		F->Li = xrealloc(F->Li, F->lnzmax * sizeof(int));
		F->Lj = xrealloc(F->Lj, F->lnzmax * sizeof(int));
		F->Lx = xrealloc(F->Lx, F->lnzmax * sizeof(orc_zp));
	}
	F->Li[F->lnz] = i;
	F->Lj[F->lnz] = j;
	F->Lx[F->lnz] = x;
	F->lnz += 1;
}

/* ------------------------------------------------------------------ */
/* x += beta * A[i]   (spasm_scatter.c:7-16)                          */
/* ------------------------------------------------------------------ */
void orc_scatter(const orc_csr *A, int i, orc_zp beta, orc_zp *x)
{
	int64_t p = A->prime;
	for (int64_t px = A->p[i]; px < A->p[i + 1]; px++) {
		int j = A->j[px];
		x[j] = orc_zp_axpy(p, beta, A->x[px], x[j]);
	}
}

/* ------------------------------------------------------------------ */
/* Gilbert-Peierls reach.  spasm_reach.c:21-85 (spasm_dfs) and         */
/* :100-135 (spasm_reach).  The visiting order (hence the order of     */
/* xj[top:m]) is the reference's: a column is emitted after everything */
/* reachable from it, neighbours are explored in row-storage order.    */
/* Workspace: xj has 3*m ints, all zero on entry and on exit.          */
/* ------------------------------------------------------------------ */
static int dfs_from(int jstart, const orc_csr *G, int top, int *xj, int *resume, int *seen, const int *qinv)
{
	int depth = 0;
	xj[0] = jstart;
	while (depth >= 0) {
		int j = xj[depth];
		int i = qinv[j];
		if (!seen[j]) {
			seen[j] = 1;
			resume[depth] = 0;
		}
		int descended = 0;
		if (i >= 0) {
			int64_t base = G->p[i];
			int w = (int) (G->p[i + 1] - base);
			for (int k = resume[depth]; k < w; k++) {
				int jj = G->j[base + k];
				if (seen[jj])
					continue;
				resume[depth] = k + 1;
				xj[++depth] = jj;
				descended = 1;
				break;
			}
		}
		if (!descended) {       /* non-pivotal column, or pivot row exhausted */
			xj[--top] = j;
			depth -= 1;
		}
	}
	return top;
}

int orc_reach(const orc_csr *U, const orc_csr *B, int k, int *xj, const int *qinv)
{
	int m = U->m;
	int top = m;
	int *resume = xj + m;
	int *seen = xj + 2 * m;
	for (int64_t px = B->p[k]; px < B->p[k + 1]; px++) {
		int j = B->j[px];
		if (!seen[j])
			top = dfs_from(j, U, top, xj, resume, seen, qinv);
	}
	for (int px = top; px < m; px++)
		seen[xj[px]] = 0;
	return top;
}

/*
 * x * U = B[k]   (spasm_triangular.c:109-146).  On exit x_b*U + x_a == B[k]
 * with x_a = x on non-pivotal columns and x_b = x on pivotal ones.
 */
int orc_sparse_triangular_solve(const orc_csr *U, const orc_csr *B, int k,
                                int *xj, orc_zp *x, const int *qinv)
{
	int m = U->m;
	int top = orc_reach(U, B, k, xj, qinv);
	for (int px = top; px < m; px++)
		x[xj[px]] = 0;
	orc_scatter(B, k, 1, x);
	for (int px = top; px < m; px++) {
		int j = xj[px];
		int i = qinv[j];
		if (i < 0)
			continue;
		orc_zp keep = x[j];
		orc_scatter(U, i, (orc_zp) (-(int64_t) keep), x);   /* pivot of U[i] is 1: kills x[j] */
		x[j] = keep;
	}
	return top;
}

/* ------------------------------------------------------------------ */
/* structural pivots (spasm_pivots.c), one thread                     */
/* ------------------------------------------------------------------ */
static int claim_pivot(int i, int j, int *pinv, int *qinv)     /* spasm_pivots.c:11-32 */
{
	int fresh = 1;
	int old_col = pinv[i];
	int old_row = qinv[j];
	if (old_col != -1) {
		qinv[old_col] = -1;
		fresh = 0;
	}
	if (old_row != -1) {
		pinv[old_row] = -1;
		fresh = 0;
	}
	pinv[i] = j;
	qinv[j] = i;
	return fresh;
}

static inline int row_weight(const orc_csr *A, int i)
{
	return (int) (A->p[i + 1] - A->p[i]);
}

static int pivots_FL(const orc_csr *A, int *pinv, int *qinv)   /* spasm_pivots.c:42-68 */
{
	int found = 0;
	for (int i = 0; i < A->n; i++) {
		int left = A->m + 1;
		for (int64_t px = A->p[i]; px < A->p[i + 1]; px++)
			if (A->j[px] < left)
				left = A->j[px];
		if (left == A->m + 1)
			continue;
		if (qinv[left] == -1 || row_weight(A, i) < row_weight(A, qinv[left]))
			found += claim_pivot(i, left, pinv, qinv);
	}
	return found;
}

static int pivots_FL_columns(const orc_csr *A, int *pinv, int *qinv)   /* spasm_pivots.c:78-125 */
{
	int n = A->n, m = A->m, found = 0;
	char *open = xmalloc(m);
	memset(open, 1, m);
	for (int i = 0; i < n; i++) {
		if (pinv[i] < 0)
			continue;
		for (int64_t px = A->p[i]; px < A->p[i + 1]; px++)
			open[A->j[px]] = 0;
	}
	for (int i = 0; i < n; i++) {
		if (pinv[i] >= 0)
			continue;
		for (int64_t px = A->p[i]; px < A->p[i + 1]; px++) {
			int j = A->j[px];
			if (!open[j] || qinv[j] >= 0)
				continue;
			found += claim_pivot(i, j, pinv, qinv);
			for (int64_t py = A->p[i]; py < A->p[i + 1]; py++)
				open[A->j[py]] = 0;
			break;
		}
	}
	free(open);
	return found;
}

/*
 * greedy alternating-cycle-free search, spasm_pivots.c:147-305 executed by a
 * single thread (every transaction commits on the first attempt).
 */
static int pivots_greedy(const orc_csr *A, int *pinv, int *qinv)
{
	int n = A->n, m = A->m, found = 0;
	signed char *w = xmalloc(m);
	int *queue = xmalloc((int64_t) m * sizeof(int));
	memset(w, 0, m);
	for (int i = 0; i < n; i++) {
		if (pinv[i] >= 0)
			continue;
		int head = 0, tail = 0, alive = 0;
		for (int64_t px = A->p[i]; px < A->p[i + 1]; px++) {
			int j = A->j[px];
			if (qinv[j] < 0) {
				w[j] = 1;
				alive += 1;
			} else {
				queue[tail++] = j;
				alive -= w[j];
				w[j] = -1;
			}
		}
		while (head < tail && alive > 0) {
			int j = queue[head++];
			int I = qinv[j];
			if (I == -1)
				continue;
			for (int64_t px = A->p[I]; px < A->p[I + 1]; px++) {
				int jj = A->j[px];
				if (w[jj] >= 0) {
					queue[tail++] = jj;
					alive -= w[jj];
					w[jj] = -1;
				}
			}
		}
		if (alive > 0) {
			int pick = -1;
			for (int64_t px = A->p[i]; px < A->p[i + 1]; px++) {
				pick = A->j[px];
				if (w[pick] == 1)
					break;
			}
			found += claim_pivot(i, pick, pinv, qinv);
		}
		for (int64_t px = A->p[i]; px < A->p[i + 1]; px++)
			w[A->j[px]] = 0;
		for (int px = 0; px < tail; px++)
			w[queue[px]] = 0;
	}
	free(w);
	free(queue);
	return found;
}

/*
 * spasm_pivots.c:316-366 (find + topological reorder) and :374-451 (copy the
 * pivotal rows, made unitary, pivot first, into U; update Uqinv / L).
 */
int orc_pivots_extract_structural(const orc_csr *A, const int *p_in, orc_lu *F,
                                  int *p, int enable_greedy)
{
	int n = A->n, m = A->m;
	int64_t prime = A->prime;
	int *qinv = xmalloc((int64_t) m * sizeof(int));
	int *pinv = xmalloc((int64_t) n * sizeof(int));
	for (int j = 0; j < m; j++)
		qinv[j] = -1;
	for (int i = 0; i < n; i++)
		pinv[i] = -1;
	int npiv = pivots_FL(A, pinv, qinv);
	npiv += pivots_FL_columns(A, pinv, qinv);
	if (enable_greedy)
		npiv += pivots_greedy(A, pinv, qinv);

	/* pivotal rows first, in topological order, then the others */
	int *xj = xmalloc((int64_t) m * sizeof(int));
	int *seen = xmalloc((int64_t) m * sizeof(int));
	int *resume = xmalloc((int64_t) m * sizeof(int));
	for (int j = 0; j < m; j++)
		seen[j] = 0;
	int top = m;
	for (int j = 0; j < m; j++)
		if (qinv[j] != -1 && !seen[j])
			top = dfs_from(j, A, top, xj, resume, seen, qinv);
	int k = 0;
	for (int px = top; px < m; px++) {
		int i = qinv[xj[px]];
		if (i != -1)
			p[k++] = i;
	}
	assert(k == npiv);
	for (int i = 0; i < n; i++)
		if (pinv[i] == -1)
			p[k++] = i;
	assert(k == n);
	free(xj);
	free(seen);
	free(resume);

	orc_csr *U = F->U;
	int64_t extra = 0;
	for (int t = 0; t < npiv; t++)
		extra += row_weight(A, p[t]);
	int64_t unz = U->p[U->n];
	csr_reserve(U, unz + extra);
	for (int t = 0; t < npiv; t++) {
		int i = p[t];
		int j = pinv[i];
		F->qinv[j] = U->n;
		orc_zp piv = 0;
		for (int64_t px = A->p[i]; px < A->p[i + 1]; px++)
			if (A->j[px] == j && A->x[px] != 0) {
				piv = A->x[px];
				break;
			}
		assert(piv != 0);
		if (F->want_L) {
			int i_out = (p_in != NULL) ? p_in[i] : i;
			L_push(F, i_out, U->n, piv);
			F->Lp[U->n] = i_out;
		}
		orc_zp scale = orc_zp_inverse(prime, piv);
		U->j[unz] = j;
		U->x[unz] = 1;
		unz += 1;
		for (int64_t px = A->p[i]; px < A->p[i + 1]; px++) {
			if (A->j[px] == j)
				continue;
			U->j[unz] = A->j[px];
			U->x[unz] = orc_zp_mul(prime, scale, A->x[px]);
			unz += 1;
		}
		U->n += 1;
		U->p[U->n] = unz;
	}
	F->r = U->n;
	free(pinv);
	free(qinv);
	return npiv;
}

/* ------------------------------------------------------------------ */
/* Schur complement (spasm_schur.c)                                   */
/* ------------------------------------------------------------------ */
static int *solve_workspace(int m)
{
	int *xj = xmalloc(3 * (int64_t) m * sizeof(int));
	for (int64_t t = 0; t < 3 * (int64_t) m; t++)
		xj[t] = 0;
	return xj;
}

/* spasm_schur.c:11-48 with a private LCG instead of rand() */
double orc_schur_estimate_density(const orc_csr *A, const int *p, int n,
                                  const orc_csr *U, const int *qinv, int R, unsigned seed)
{
	if (n == 0)
		return 0;
	int m = A->m;
	orc_zp *x = xmalloc((int64_t) m * sizeof(orc_zp));
	int *xj = solve_workspace(m);
	int64_t nnz = 0;
	uint64_t state = seed * 2862933555777941757ULL + 3037000493ULL;
	for (int t = 0; t < R; t++) {
		state = state * 6364136223846793005ULL + 1442695040888963407ULL;
		int inew = p[(state >> 33) % (uint64_t) n];
		int top = orc_sparse_triangular_solve(U, A, inew, xj, x, qinv);
		for (int px = top; px < m; px++) {
			int j = xj[px];
			if (qinv[j] < 0 && x[j] != 0)
				nnz += 1;
		}
	}
	free(x);
	free(xj);
	return ((double) nnz) / (m - U->n) / R;
}

/*
 * spasm_schur.c:61-193 run by one thread: row k of S is the reduction of row
 * p[k] of A; entries come out in reach order.
 */
orc_csr *orc_schur(const orc_csr *A, const int *p, int n, orc_lu *F,
                   const int *p_in, int *p_out)
{
	int m = A->m;
	const orc_csr *U = F->U;
	const int *qinv = F->qinv;
	orc_csr *S = orc_csr_alloc(n, m, 16 + 2 * (A->p[A->n] / (A->n > 0 ? A->n : 1)) * (int64_t) n, A->prime);
	orc_zp *x = xmalloc((int64_t) m * sizeof(orc_zp));
	int *xj = solve_workspace(m);
	int64_t snz = 0;
	for (int k = 0; k < n; k++) {
		int inew = p[k];
		int i_orig = (p_in != NULL) ? p_in[inew] : inew;
		int top = orc_sparse_triangular_solve(U, A, inew, xj, x, qinv);
		if (snz + m > S->nzmax)
			csr_reserve(S, 2 * S->nzmax + m);
		if (p_out != NULL)
			p_out[k] = i_orig;
		for (int px = top; px < m; px++) {
			int j = xj[px];
			if (x[j] == 0)
				continue;
			if (qinv[j] < 0) {
				S->j[snz] = j;
				S->x[snz] = x[j];
				snz += 1;
			} else if (F->want_L) {
				L_push(F, i_orig, qinv[j], x[j]);
			}
		}
		S->p[k + 1] = snz;
	}
	free(x);
	free(xj);
	return S;
}

/*
 * spasm_schur.c:257-343: dense rows of the Schur complement, columns q[0..Sm)
 * = the non-pivotal columns in increasing order (prepare_q, :195-203).
 * S is n x Sm row-major, values as int64 (the SPASM_I64 datatype).
 */
void orc_schur_dense(const orc_csr *A, const int *p, int n, const int *p_in,
                     orc_lu *F, int64_t *S, int *q, int *p_out)
{
	int m = A->m;
	const orc_csr *U = F->U;
	const int *qinv = F->qinv;
	int Sm = 0;
	for (int j = 0; j < m; j++)
		if (qinv[j] < 0)
			q[Sm++] = j;
	assert(Sm == m - U->n);
	orc_zp *x = xmalloc((int64_t) m * sizeof(orc_zp));
	int *xj = solve_workspace(m);
	for (int k = 0; k < n; k++) {
		int i = p[k];
		int i_orig = (p_in != NULL) ? p_in[i] : i;
		p_out[k] = i_orig;
		for (int j = 0; j < m; j++)
			x[j] = 0;
		int top = orc_sparse_triangular_solve(U, A, i, xj, x, qinv);
		for (int l = 0; l < Sm; l++)
			S[(int64_t) k * Sm + l] = x[q[l]];
		if (F->want_L)
			for (int px = top; px < m; px++) {
				int j = xj[px];
				if (qinv[j] < 0 || x[j] == 0)
					continue;
				L_push(F, i_orig, qinv[j], x[j]);
			}
	}
	free(x);
	free(xj);
}

/* ------------------------------------------------------------------ */
/* Dense reduced row echelon form mod p.                              */
/* Contract of spasm_ffpack_rref (spasm_ffpack.cpp:23-49, 88-96) as    */
/* consumed by update_U_after_rref (spasm_echelonize.c:189-222):       */
/*   returns r = rank; qinv[0..r) = pivot columns (row i's pivot is    */
/*   column qinv[i]); qinv[r..m) = the other columns; for i < r and    */
/*   k >= r, A[i*ldA + k] is the coefficient of row i of the RREF on   */
/*   column qinv[k] (the pivots are implicit ones).                    */
/* The RREF of a row space is unique and its pivots are the column     */
/* rank profile; FFPACK's ReducedRowEchelonForm computes exactly that. */
/* Here: qinv[r..m) increasing; A[i*ldA+k] for k < r is the identity.  */
/* ------------------------------------------------------------------ */
int orc_dense_rref(int64_t prime, int n, int m, int64_t *A, int ldA, int64_t *qinv)
{
	for (int i = 0; i < n; i++)
		for (int j = 0; j < m; j++)
			A[(int64_t) i * ldA + j] = orc_zp_init(prime, A[(int64_t) i * ldA + j]);
	int r = 0;
	int *pivcol = xmalloc((int64_t) (m > 0 ? m : 1) * sizeof(int));
	for (int c = 0; c < m && r < n; c++) {
		int src = -1;
		for (int i = r; i < n; i++)
			if (A[(int64_t) i * ldA + c] != 0) {
				src = i;
				break;
			}
		if (src < 0)
			continue;
		if (src != r)
			for (int j = 0; j < m; j++) {
				int64_t t = A[(int64_t) r * ldA + j];
				A[(int64_t) r * ldA + j] = A[(int64_t) src * ldA + j];
				A[(int64_t) src * ldA + j] = t;
			}
		orc_zp inv = orc_zp_inverse(prime, (orc_zp) A[(int64_t) r * ldA + c]);
		for (int j = 0; j < m; j++)
			A[(int64_t) r * ldA + j] = orc_zp_mul(prime, inv, (orc_zp) A[(int64_t) r * ldA + j]);
		for (int i = 0; i < n; i++) {
			if (i == r)
				continue;
			orc_zp f = (orc_zp) A[(int64_t) i * ldA + c];
			if (f == 0)
				continue;
			orc_zp mf = orc_zp_sub(prime, 0, f);
			for (int j = 0; j < m; j++)
				A[(int64_t) i * ldA + j] = orc_zp_axpy(prime, mf, (orc_zp) A[(int64_t) r * ldA + j],
				                                       (orc_zp) A[(int64_t) i * ldA + j]);
		}
		pivcol[r] = c;
		r += 1;
	}
	/* column permutation: pivots first (row order), then the rest (increasing) */
	char *is_piv = xmalloc(m > 0 ? m : 1);
	memset(is_piv, 0, m > 0 ? m : 1);
	for (int i = 0; i < r; i++) {
		qinv[i] = pivcol[i];
		is_piv[pivcol[i]] = 1;
	}
	int k = r;
	for (int j = 0; j < m; j++)
		if (!is_piv[j])
			qinv[k++] = j;
	/* permute the columns of each row in place */
	int64_t *tmp = xmalloc((int64_t) (m > 0 ? m : 1) * sizeof(int64_t));
	for (int i = 0; i < n; i++) {
		for (int t = 0; t < m; t++)
			tmp[t] = A[(int64_t) i * ldA + qinv[t]];
		for (int t = 0; t < m; t++)
			A[(int64_t) i * ldA + t] = tmp[t];
	}
	free(tmp);
	free(is_piv);
	free(pivcol);
	return r;
}

/* ------------------------------------------------------------------ */
/* driver (spasm_echelonize.c)                                        */
/* ------------------------------------------------------------------ */
void orc_opts_init(orc_opts *o)                   /* spasm_echelonize.c:9-28 */
{
	o->enable_greedy_pivot_search = 1;
	o->enable_tall_and_skinny = 1;
	o->enable_dense = 1;
	o->enable_GPLU = 1;
	o->L = 0;
	o->complete = 0;
	o->min_pivot_proportion = 0.1;
	o->max_round = 3;
	o->sparsity_threshold = 0.05;
	o->tall_and_skinny_ratio = 5;
	o->dense_block_size = 1000;
	o->low_rank_ratio = 0.5;
	o->low_rank_start_weight = -1;
}

/*
 * spasm_echelonize.c:55-183 (echelonize_GPLU) without the probabilistic
 * early-abort test (it only shortens the loop once no pivot is left).
 */
static void finish_GPLU(const orc_csr *A, const int *p, int n, const int *p_in, orc_lu *F)
{
	int m = A->m;
	int64_t prime = A->prime;
	orc_csr *U = F->U;
	int rmax = (A->n < m) ? A->n : m;
	orc_zp *x = xmalloc((int64_t) m * sizeof(orc_zp));
	int *xj = solve_workspace(m);
	int64_t unz = U->p[U->n];
	for (int t = 0; t < n; t++) {
		if (!F->want_L && U->n == rmax)
			break;
		csr_reserve(U, unz + m + 1 > U->nzmax ? 2 * U->nzmax + m + 1 : U->nzmax);
		int inew = p[t];
		int i_orig = (p_in != NULL) ? p_in[inew] : inew;
		int top = orc_sparse_triangular_solve(U, A, inew, xj, x, F->qinv);
		int jpiv = m;
		for (int px = top; px < m; px++) {
			int j = xj[px];
			if (x[j] == 0)
				continue;
			if (F->qinv[j] < 0) {
				if (j < jpiv)
					jpiv = j;
			} else if (F->want_L) {
				L_push(F, i_orig, F->qinv[j], x[j]);
			}
		}
		if (jpiv == m)
			continue;
		if (F->want_L) {
			F->Lp[U->n] = i_orig;
			L_push(F, i_orig, U->n, x[jpiv]);
		}
		F->qinv[jpiv] = U->n;
		U->j[unz] = jpiv;
		U->x[unz] = 1;
		unz += 1;
		orc_zp beta = orc_zp_inverse(prime, x[jpiv]);
		for (int px = top; px < m; px++) {
			int j = xj[px];
			if (x[j] != 0 && F->qinv[j] < 0) {
				U->j[unz] = j;
				U->x[unz] = orc_zp_mul(prime, beta, x[j]);
				unz += 1;
			}
		}
		U->n += 1;
		U->p[U->n] = unz;
	}
	free(x);
	free(xj);
}

/*
 * spasm_echelonize.c:379-467 (echelonize_dense, no L) with
 * update_U_after_rref (:189-222).  The low-rank switch is not taken: the
 * oracle always walks every block, which yields the same row space.
 */
static void finish_dense(const orc_csr *A, const int *p, int n, const int *p_in, orc_lu *F, int block)
{
	int m = A->m;
	int64_t prime = A->prime;
	orc_csr *U = F->U;
	int done = 0;
	while (done < n) {
		int Sn = (n - done < block) ? n - done : block;
		int Sm = m - U->n;
		if (Sm <= 0)
			break;
		int64_t *S = xmalloc((int64_t) Sn * Sm * sizeof(int64_t));
		int *q = xmalloc((int64_t) Sm * sizeof(int));
		int *p_out = xmalloc((int64_t) Sn * sizeof(int));
		int64_t *Sqinv = xmalloc((int64_t) Sm * sizeof(int64_t));
		int keepL = F->want_L;
		F->want_L = 0;
		orc_schur_dense(A, p + done, Sn, p_in, F, S, q, p_out);
		F->want_L = keepL;
		int rr = orc_dense_rref(prime, Sn, Sm, S, Sm, Sqinv);
		int64_t unz = U->p[U->n];
		csr_reserve(U, unz + (int64_t) (1 + Sm - rr) * rr);
		for (int i = 0; i < rr; i++) {
			int jp = q[Sqinv[i]];
			U->j[unz] = jp;
			U->x[unz] = 1;
			unz += 1;
			F->qinv[jp] = U->n;
			for (int k = rr; k < Sm; k++) {
				orc_zp v = (orc_zp) S[(int64_t) i * Sm + k];
				if (v == 0)
					continue;
				U->j[unz] = q[Sqinv[k]];
				U->x[unz] = v;
				unz += 1;
			}
			U->n += 1;
			U->p[U->n] = unz;
		}
		free(S);
		free(q);
		free(p_out);
		free(Sqinv);
		done += Sn;
	}
}

/* spasm_echelonize.c:473-616 */
orc_lu *orc_echelonize(const orc_csr *A0, const orc_opts *opts_in)
{
	orc_opts o;
	if (opts_in == NULL)
		orc_opts_init(&o);
	else
		o = *opts_in;
	if (o.complete)
		o.L = 1;
	if (o.L)
		o.enable_tall_and_skinny = 0;
	const orc_csr *A = A0;
	int n = A->n, m = A->m;
	orc_lu *F = orc_lu_alloc(n, m, A->p[A->n] + 16, A->prime, o.L);
	int *p = xmalloc((int64_t) (n > 0 ? n : 1) * sizeof(int));
	int *p_in = NULL;
	double density = (n > 0 && m > 0) ? (double) A->p[A->n] / n / m : 0;
	int npiv = 0, status = 0, round;
	for (round = 0; round < o.max_round; round++) {
		if (A->p[A->n] == 0) {
			status = 1;
			break;
		}
		npiv = orc_pivots_extract_structural(A, p_in, F, p, o.enable_greedy_pivot_search);
		int bound = (n < m - F->U->n) ? n : m - F->U->n;
		if (npiv < o.min_pivot_proportion * bound) {
			status = 2;
			break;
		}
		density = orc_schur_estimate_density(A, p + npiv, n - npiv, F->U, F->qinv, 100, 42u + round);
		if (density > o.sparsity_threshold) {
			status = 2;
			break;
		}
		int *p_out = xmalloc((int64_t) (n - npiv > 0 ? n - npiv : 1) * sizeof(int));
		orc_csr *S = orc_schur(A, p + npiv, n - npiv, F, p_in, p_out);
		if (A != A0)
			orc_csr_free((orc_csr *) A);
		A = S;
		n = n - npiv;
		free(p_in);
		p_in = p_out;
	}
	if (status == 0) {
		npiv = 0;
		for (int i = 0; i < n; i++)
			p[i] = i;
	}
	if (status != 1) {
		if (o.enable_dense && density > o.sparsity_threshold && !o.L)
			finish_dense(A, p + npiv, n - npiv, p_in, F, o.dense_block_size);
		else if (o.enable_GPLU)
			finish_GPLU(A, p + npiv, n - npiv, p_in, F);
	}
	free(p);
	free(p_in);
	if (A != A0)
		orc_csr_free((orc_csr *) A);
	F->r = F->U->n;
	return F;
}

/*
 * spasm_rref.c:25-146, one thread: RREF of A*Q from an echelonized U.
 * Row i of R is U[i] reduced by every other pivotal row; pivot first.
 */
orc_csr *orc_rref(const orc_lu *F, int *Rqinv)
{
	const orc_csr *U = F->U;
	int n = U->n, m = U->m;
	orc_csr *R = orc_csr_alloc(n, m, U->p[n] + m, U->prime);
	int *ql = xmalloc((int64_t) m * sizeof(int));
	memcpy(ql, F->qinv, (int64_t) m * sizeof(int));
	orc_zp *x = xmalloc((int64_t) m * sizeof(orc_zp));
	int *xj = solve_workspace(m);
	int64_t nnz = 0;
	for (int i = 0; i < n; i++) {
		int piv = U->j[U->p[i]];
		ql[piv] = -1;
		int top = orc_sparse_triangular_solve(U, U, i, xj, x, ql);
		for (int px = top + 1; px < m; px++)
			if (xj[px] == piv) {
				xj[px] = xj[top];
				xj[top] = piv;
				break;
			}
		if (nnz + m > R->nzmax)
			csr_reserve(R, 2 * R->nzmax + m);
		for (int px = top; px < m; px++) {
			int j = xj[px];
			if (ql[j] < 0 && x[j] != 0) {
				R->j[nnz] = j;
				R->x[nnz] = x[j];
				nnz += 1;
			}
		}
		R->p[i + 1] = nnz;
		ql[piv] = i;
	}
	for (int j = 0; j < m; j++)
		Rqinv[j] = -1;
	for (int i = 0; i < n; i++)
		Rqinv[R->j[R->p[i]]] = i;
	free(ql);
	free(x);
	free(xj);
	return R;
}
