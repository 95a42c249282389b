// Host-only parts of the multi-GPU layer (include/spasm_hip.h, section (M)): no HIP, no RCCL -- what they compute can be
// checked on a CPU box for any world size, which the collectives themselves cannot.
//   * the exchange plan of the all-gatherv of Schur complements (who sends what to whom, in which order, at which
//     offsets): dist_api.hip executes exactly this list;
//   * the column-slab problem of one rank: the split that fits the back-substituted path (backsolve.hip), whose columns
//     never meet.
#include <algorithm>
#include <vector>

#include "common.h"

using namespace sh;

extern "C" {

void spasm_hip_shard(int n, int rank, int world, int *lo, int *hi);

// Steps of spasm_hip_dschur_allgatherv as rank `me` issues them, in order.  sizes[2r], sizes[2r + 1] = rows and entries
// of rank r's slice.  RCCL matches the k-th send of a to b with the k-th receive b posts for a (one group), so for every
// ordered pair the sequence of (array, count) sent must be the sequence received: tests/test_dist_cpu.py checks that
// for worlds of 2 to 8, and that the receives and the local copy tile the gathered arrays.  row_base / nz_base
// (world + 1 entries each, may be NULL) receive the offsets of the slices.  Returns the number of steps (written up to
// cap; call with cap = 0 to size the list).
int spasm_hip_allgatherv_plan(int world, int me, const i64 *sizes, spasm_hip_xfer *out, int cap, i64 *row_base, i64 *nz_base)
{
	if (world < 1 || me < 0 || me >= world)
		die("spasm_hip_allgatherv_plan: rank %d of %d", me, world);
	std::vector<i64> rb((size_t) world + 1, 0), zb((size_t) world + 1, 0);
	for (int r = 0; r < world; r++) {
		if (sizes[2 * r] < 0 || sizes[2 * r + 1] < 0)
			die("spasm_hip_allgatherv_plan: negative size for rank %d", r);
		rb[r + 1] = rb[r] + sizes[2 * r];
		zb[r + 1] = zb[r] + sizes[2 * r + 1];
	}
	if (row_base != nullptr)
		std::copy(rb.begin(), rb.end(), row_base);
	if (nz_base != nullptr)
		std::copy(zb.begin(), zb.end(), nz_base);
	int count = 0;
	auto emit = [&](int kind, int peer, int array, i64 src, i64 dst, i64 n) {
		if (n <= 0)
			return;                  // empty transfers are not posted at all (by either side: both know every size)
		if (count < cap)
			out[count] = spasm_hip_xfer{kind, peer, array, src, dst, n};
		count += 1;
	};
	const i64 my_rows = sizes[2 * me], my_nz = sizes[2 * me + 1];
	for (int r = 0; r < world; r++) {
		if (r == me)
			continue;
		// to peer r: my slice; from peer r: its slice.  Per peer and direction the order is: row pointers, columns, values.
		emit(SPASM_HIP_XFER_SEND, r, 0, 0, 0, my_rows);
		emit(SPASM_HIP_XFER_RECV, r, 0, 0, rb[r], sizes[2 * r]);
		emit(SPASM_HIP_XFER_SEND, r, 1, 0, 0, my_nz);
		emit(SPASM_HIP_XFER_SEND, r, 2, 0, 0, my_nz);
		emit(SPASM_HIP_XFER_RECV, r, 1, 0, zb[r], sizes[2 * r + 1]);
		emit(SPASM_HIP_XFER_RECV, r, 2, 0, zb[r], sizes[2 * r + 1]);
	}
	emit(SPASM_HIP_XFER_COPY, me, 0, 0, rb[me], my_rows);
	emit(SPASM_HIP_XFER_COPY, me, 1, 0, zb[me], my_nz);
	emit(SPASM_HIP_XFER_COPY, me, 2, 0, zb[me], my_nz);
	return count;
}

// The column-slab problem `part` of `parts`.  S = A_n - A_p (U_pp^-1 U_pn): the non-pivotal columns never meet, so the
// Schur complement restricted to a set C of non-pivotal columns is the Schur complement of (A, U) with every non-pivotal
// column outside C deleted.  The non-pivotal columns of the factor, in increasing order, are cut into `parts` contiguous
// ranges (sizes differ by at most one); the problem of range `part` keeps the pivotal columns and that range, renumbered
// in increasing order (cols[c'] = original column of column c' of the slab problem, m' entries; returned value = m').
// *A_slab: same rows as A; *fact_slab: U (every row keeps its pivot first) and qinv in the new numbering.  Rows of the
// slab's Schur complement mapped back through cols[] and concatenated over the parts in order are the rows of the full
// one, sorted by column.  Nothing is replicated between ranks except the pivotal part of U (a few entries per row); the
// image R of a rank is r x |C|.  Free the results with spasm_hip_csr_free / spasm_hip_lu_free.
int spasm_hip_column_slab(const struct spasm_csr *A, const struct spasm_lu *fact, int part, int parts, struct spasm_csr **A_slab,
                          struct spasm_lu **fact_slab, int *cols)
{
	const struct spasm_csr *U = fact->U;
	const int m = A->m, r = U->n;
	const i64 prime = A->field->p;
	if (U->m != m)
		die("spasm_hip_column_slab: A has %d columns, the factor %d", m, U->m);
	if (parts < 1 || part < 0 || part >= parts)
		die("spasm_hip_column_slab: part %d of %d", part, parts);
	int lo = 0, hi = 0;
	spasm_hip_shard(m - r, part, parts, &lo, &hi);
	std::vector<int> newcol((size_t) (m > 0 ? m : 1), -1);
	int mm = 0, np = 0;
	for (int j = 0; j < m; j++) {
		bool keep = fact->qinv[j] >= 0;
		if (!keep) {
			keep = np >= lo && np < hi;
			np += 1;
		}
		if (keep) {
			cols[mm] = j;
			newcol[j] = mm;
			mm += 1;
		}
	}
	auto restrict_to = [&](const struct spasm_csr *M, int nrows) {
		i64 nz = 0;
		for (i64 px = 0; px < M->p[nrows]; px++)
			nz += newcol[M->j[px]] >= 0;
		struct spasm_csr *R = spasm_hip_csr_alloc(nrows, mm, nz, prime, true);
		i64 w = 0;
		for (int i = 0; i < nrows; i++) {
			for (i64 px = M->p[i]; px < M->p[i + 1]; px++) {
				const int c = newcol[M->j[px]];
				if (c < 0)
					continue;
				R->j[w] = c;
				R->x[w] = M->x[px];
				w += 1;
			}
			R->p[i + 1] = w;
		}
		return R;
	};
	*A_slab = restrict_to(A, A->n);
	struct spasm_lu *F = (struct spasm_lu *) xmalloc(sizeof(*F));
	F->r = r;
	F->complete = false;
	F->L = nullptr;
	F->p = nullptr;
	F->Ltmp = nullptr;
	F->U = restrict_to(U, r);          // (pivotal columns are all kept: every row still starts with its pivot)
	F->qinv = (int *) xmalloc((i64) (mm > 0 ? mm : 1) * sizeof(int));
	for (int c = 0; c < mm; c++)
		F->qinv[c] = fact->qinv[cols[c]];
	*fact_slab = F;
	return mm;
}

}  // extern "C"
