// C ABI of the dense tail: dense rows of the Schur complement and dense RREF mod p.
#include <algorithm>
#include <cinttypes>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "device_types.h"

namespace sh {
int device_rref(int64_t prime, int n, int m, uint32_t *dA, int64_t ld, int *d_pivcol, hipStream_t stream, int use_mfma,
                float *ms_update);
int device_lu(int64_t prime, int n, int m, uint32_t *dA, int64_t ld, int *dP, int *dQ, hipStream_t stream);
int device_echelon_extend(int64_t prime, int m, uint32_t *dM, int64_t ld, int k, int Sn, int *d_piv, hipStream_t stream);
spasm_hip_dfact *cached_dfact(const struct spasm_csr *U, const int *qinv, hipStream_t stream);
void launch_split_rows(const int64_t *Sp, int N, int pieces, int64_t *out, hipStream_t stream);
void launch_sum_pieces(const uint32_t *parts, int64_t ldp, int N, int pieces, int m, uint32_t p, uint32_t *out, int64_t ldo, hipStream_t stream);
void launch_combine(const int64_t *Ap, const int *Aj, const int *Ax, const int *rows, int nrows, int N, int w, int m,
                    uint64_t salt, unsigned long long *Y, const Mont &M, hipStream_t stream, const uint32_t *colmap, uint32_t base, int64_t annz);
void launch_dense_reduce_rows(const unsigned long long *Y, int N, int m, const Mont &M, uint32_t *S, int64_t ldS, hipStream_t stream);
bool rows_are_nonpivotal(const int64_t *Ap, const int *Aj, const int *rows, int nrows, const uint32_t *lab, uint32_t base, hipStream_t stream);
void launch_dense_count(const unsigned long long *Y, int N, int m, uint32_t p, int *row_len, hipStream_t stream);
void launch_dense_pack(const unsigned long long *Y, int N, int m, uint32_t p, const int64_t *Sp, int *Sj, int *Sx,
                       hipStream_t stream, const int *unmap = nullptr);
void launch_row_scan(const int *row_len, int n, int64_t *blocksum, int64_t *Sp, hipStream_t stream);
void launch_echelon_count(const uint32_t *M, int64_t ld, int m, int k, int *row_len, hipStream_t stream);
void launch_echelon_pack(const uint32_t *M, int64_t ld, int m, int k, const int *piv, const int *q, uint32_t p, const int64_t *Sp, int *Uj,
                         int *Ux, hipStream_t stream);
void wave_dense_geometry(int rpad, int Sm, bool wide, int64_t *slot_bytes, int64_t *off_bm, int64_t *off_xn);
void group_geometry(int rpad, int Sm, bool wide, int64_t *slot_bytes, int64_t *off_bm);
void launch_schur_group(const SchurArgs &a, unsigned char *scratch, int64_t slot_bytes, int64_t off_bm, bool wide,
                        uint32_t *dense_out, int64_t ldS, int blocks, hipStream_t stream, int watch, float min_eff,
                        long long min_w, int waves);
int64_t regroup_scratch_ints(int nrows, int r);
void launch_regroup_rows(const SchurArgs &a, int *sortbuf, int *order, hipStream_t stream);
void launch_schur_wave_dense(const SchurArgs &a, unsigned char *scratch, int64_t slot_bytes, int64_t off_bm, int64_t off_xn,
                             bool wide, uint32_t *dense_out, int64_t ldS, int blocks, hipStream_t stream);
void backsolve_build(const spasm_hip_dfact *F, hipStream_t stream);
void launch_backsolve_apply(const SchurArgs &a, const spasm_hip_dfact *F, uint32_t *dense_out, int64_t ldS, hipStream_t stream,
                            BsDirectOut *direct);
bool backsolve_wanted(const spasm_hip_dfact *F, bool other_path_forced, int nrows);
}  // namespace sh

using namespace sh;

namespace {

int env_int(const char *name, int dflt)
{
	const char *e = sh::env_get(name);
	return (e == nullptr || *e == 0) ? dflt : std::atoi(e);
}

template <typename T> T *dalloc(int64_t count)
{
	return static_cast<T *>(sh::big_alloc((size_t) (count > 0 ? count : 1) * sizeof(T)));
}

}  // namespace

extern "C" {

int spasm_hip_drref(i64 prime, int n, int m, u32 *d_A, i64 ld, int *d_pivcol, void *stream)
{
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_drref: no HIP device (this library has no CPU path)");
	return device_rref(prime, n, m, d_A, ld, d_pivcol, (hipStream_t) stream, env_int("SPASM_HIP_RREF_MFMA", 1), nullptr);
}

// Rows [0, k) of d_M are reduced echelon rows with pivot columns d_piv[0..k) (identity on them); the Sn rows below are
// reduced by them and by each other (row panels, any pivots: dense_kernels.hip) and the non-zero ones appended: on return
// rows [0, k') are reduced echelon rows, d_piv[0..k') their pivot columns.  Returns k'.  p <= 65279.
int spasm_hip_dechelon_extend(i64 prime, int m, u32 *d_M, i64 ld, int k, int Sn, int *d_piv, void *stream)
{
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_dechelon_extend: no HIP device (this library has no CPU path)");
	return device_echelon_extend(prime, m, d_M, ld, k, Sn, d_piv, (hipStream_t) stream);
}

// timing variant used by bench/profiles: ms of the trailing-update kernels only
int spasm_hip_drref_timed(i64 prime, int n, int m, u32 *d_A, i64 ld, int *d_pivcol, void *stream, int use_mfma,
                          float *ms_update)
{
	return device_rref(prime, n, m, d_A, ld, d_pivcol, (hipStream_t) stream, use_mfma, ms_update);
}

}  // extern "C"

// dense rows; Lout != nullptr: also record the elimination coefficients.  Returns the status bits of
// the kernels (2 = L pool exhausted).
static int dschur_dense_impl(const spasm_hip_dcsr *A, const int *d_rows, int nrows, const spasm_hip_dfact *F,
                             spasm_hip_dwork *W, u32 *d_S, i64 ldS, void *stream_, LOut *Lout)
{
	hipStream_t stream = (hipStream_t) stream_;
	if (nrows > W->max_rows)
		die("spasm_hip_dschur_dense: %d rows but the workspace was sized for %d", nrows, W->max_rows);
	if (A->m != F->m)
		die("spasm_hip_dschur_dense: column count mismatch (A %d, factor %d)", A->m, F->m);
	if (ldS < F->Sm)
		die("spasm_hip_dschur_dense: leading dimension %" PRId64 " below the %d non-pivotal columns", ldS, F->Sm);
	if (nrows == 0)
		return 0;
	if (Lout == nullptr && backsolve_wanted(F, env_int("SPASM_HIP_FORCE_TIER", 0) != 0 || env_int("SPASM_HIP_GROUP", -1) >= 0, nrows)) {
		// dense rows straight from the back-substituted image (backsolve.hip)
		if (!F->bs.valid)
			backsolve_build(F, stream);
		HIP_CHECK(hipMemsetAsync(W->d_ctr, 0, CTR_COUNT * sizeof(int), stream));
		HIP_CHECK(hipMemsetAsync(W->d_ctr64, 0, C64_COUNT * sizeof(unsigned long long), stream));
		SchurArgs a{};
		a.Ap = A->p;
		a.Aj = A->j;
		a.Ax = A->x;
		a.rows = d_rows;
		a.nrows = nrows;
		a.q = F->d_q;
		a.r = F->rpad;
		a.Sm = F->Sm;
		a.m = F->m;
		a.F = to_dev(F->mont);
		a.row_len = W->d_row_len;
		a.ctr = W->d_ctr;
		a.ctr64 = W->d_ctr64;
		a.done_ctr = CTR_DONE2;
		a.avg_row_entries = (A->nnz > 0 && A->n > 0) ? (int) std::min<i64>(A->nnz / A->n, 1 << 30) : 0;
		launch_backsolve_apply(a, F, d_S, ldS, stream, nullptr);
		HIP_CHECK(hipStreamSynchronize(stream));
		return 0;
	}
	ensure_row_tables(F, stream);
	// a column receives at most maxdeg + 1 terms, each below 2p (the row-group kernel adds unreduced products)
	const bool wide = (2.0 * (double) F->prime * ((double) F->maxdeg + 3.0) >= 4294967296.0);
	int dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	hipDeviceProp_t prop;
	HIP_CHECK(hipGetDeviceProperties(&prop, dev));
	const int cus = prop.multiProcessorCount;
	i64 slot_bytes, off_bm, off_xn;
	wave_dense_geometry(F->rpad, F->Sm, wide, &slot_bytes, &off_bm, &off_xn);
	int slots = (cus * 32);
	const i64 budget = (i64) env_int("SPASM_HIP_SCRATCH_GB", 48) << 30;
	slots = (int) std::max<i64>(cus, std::min<i64>(slots, budget / slot_bytes));
	slots = std::max(1, std::min(slots, nrows));
	i64 need = slot_bytes * slots;
	// dense blocks are small batches: per-row kernel by default; the row-group kernel (64 consecutive rows
	// per wave) on request or for blocks of at least 4096 rows
	int group_mode = env_int("SPASM_HIP_GROUP", -1);
	if (group_mode < 0)
		group_mode = nrows >= 4096 ? 1 : 0;
	i64 gslot = 0, goff = 0;
	int gslots = 0, gwaves = 1;
	if (group_mode) {
		group_geometry(F->rpad, F->Sm, wide, &gslot, &goff);
		// (very wide matrices: too few slices fit the budget to fill the chip -- per-row kernel instead, as in schur_api.hip)
		if (budget / gslot < 1 || (budget / gslot < std::min<i64>((nrows + 63) / 64, cus / 2) && env_int("SPASM_HIP_GROUP", -1) < 0))
			group_mode = 0;
	}
	if (group_mode) {
		const i64 in_flight = std::min<i64>((nrows + 63) / 64, budget / gslot);          // (groups, or what the budget holds: schur_api.hip)
		gwaves = env_int("SPASM_HIP_GROUP_WAVES", in_flight <= cus * 3 ? 4 : in_flight <= cus * 12 ? 2 : 1);
		gslots = (int) std::max<i64>(1, std::min<i64>((nrows + 63) / 64, std::min<i64>(gwaves >= 4 ? cus * 2 : gwaves >= 2 ? cus * 4 : cus * 8, budget / gslot)));
		need = gslot * gslots;
	}
	if (need > W->scratch_bytes) {
		if (W->d_scratch != nullptr)
			sh::big_free(W->d_scratch);
		HIP_CHECK(sh::malloc_or_trim((void **) &W->d_scratch, (size_t) need));
		W->scratch_bytes = need;
		HIP_CHECK(hipMemsetAsync(W->d_scratch, 0, (size_t) need, stream));
	}
	HIP_CHECK(hipMemsetAsync(W->d_ctr, 0, CTR_COUNT * sizeof(int), stream));
	HIP_CHECK(hipMemsetAsync(W->d_ctr64, 0, C64_COUNT * sizeof(unsigned long long), stream));
	SchurArgs a{};
	a.Ap = A->p;
	a.Aj = A->j;
	a.Ax = A->x;
	a.rows = d_rows;
	a.nrows = nrows;
	a.lab = F->d_lab;
	a.q = F->d_q;
	a.rp = F->d_rp;
	a.ent = F->d_ent;
	a.head = F->d_head;
	a.lvl_end = F->d_lvl_end;
	a.lvl_end_w = F->d_lvl_end_w;
	a.r = F->rpad;
	a.Sm = F->Sm;
	a.m = F->m;
	a.F = to_dev(F->mont);
	a.pool_j = W->d_pool_j;
	a.pool_x = W->d_pool_x;
	a.pool_cap = W->pool_cap;
	a.row_off = W->d_row_off;
	a.row_len = W->d_row_len;
	a.ctr = W->d_ctr;
	a.ctr64 = W->d_ctr64;
	a.list = nullptr;
	a.list_count = nullptr;
	a.ovf_list = nullptr;
	a.next_ctr = CTR_ROW_NEXT3;
	a.ovf_ctr = CTR_OVF2;
	a.done_ctr = CTR_DONE2;
	if (Lout != nullptr) {
		a.L_i = Lout->Li;
		a.L_j = Lout->Lj;
		a.L_x = Lout->Lx;
		a.L_cap = Lout->cap;
		a.kof = F->d_kof;
		a.row_orig = Lout->row_orig;
	}
	if (group_mode) {
		a.next_ctr = CTR_ROW_NEXT_G;
		// rows grouped by connected component of the pivot graph when there are several (schur_api.hip)
		if (Lout == nullptr && F->ncomp > 1 && (i64) F->comp_largest * 10 < (i64) F->r * 9 && env_int("SPASM_HIP_GROUP_REGROUP", 1)) {
			a.comp = F->d_comp;
			const int64_t need = regroup_scratch_ints(nrows, F->rpad);
			if (W->sortbuf_ints < need) {
				if (W->d_sortbuf != nullptr)
					sh::big_free(W->d_sortbuf);
				HIP_CHECK(sh::malloc_or_trim((void **) &W->d_sortbuf, (size_t) need * sizeof(int)));
				W->sortbuf_ints = need;
			}
			if (W->d_order == nullptr)
				HIP_CHECK(sh::malloc_or_trim((void **) &W->d_order, (size_t) W->max_rows * sizeof(int)));
			launch_regroup_rows(a, W->d_sortbuf, W->d_order, stream);
			a.order = W->d_order;
		}
		launch_schur_group(a, W->d_scratch, gslot, goff, wide, d_S, ldS, gslots, stream, 0, 0.0f, 0, gwaves);
	} else {
		launch_schur_wave_dense(a, W->d_scratch, slot_bytes, off_bm, off_xn, wide, d_S, ldS, slots, stream);
	}
	int ctr[CTR_COUNT];
	unsigned long long ctr64[C64_COUNT];
	HIP_CHECK(hipMemcpyAsync(ctr, W->d_ctr, sizeof(ctr), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipMemcpyAsync(ctr64, W->d_ctr64, sizeof(ctr64), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	if (Lout != nullptr)
		Lout->used = (i64) ctr64[C64_LPOOL];
	return ctr[CTR_STATUS] & 2;
}


namespace sh {
// N random combinations of the rows d_rows[0..n) of A (w > 0: of w random rows each, first coefficient 1; w <= 0: of all
// the rows), reduced by F, as dense rows on the non-pivotal columns: d_S is N x ldS, values in [0, p).  Everything on
// `stream`, which is synchronised before returning.  W needs max_rows >= N.
// `compact`: the rows only hold non-pivotal columns of F (rows of a Schur complement by this factor): the accumulators then
// span those Sm columns instead of all m -- mk15.b4: 4,096 combinations x 675,675 columns x 8 bytes = 22 GB of which the
// Schur complement occupies 71,000 columns, memory that every call had to get, clear and scan.
void device_random_dense_rows(const spasm_hip_dcsr &dA, const int *d_rows, int n, const spasm_hip_dfact *F, int N, int w,
                              uint64_t salt, u32 *d_S, i64 ldS, spasm_hip_dwork *W, hipStream_t stream, bool compact)
{
	const int m_all = dA.m;
	const int m = compact ? F->Sm : m_all;          // width of the accumulators
	const uint32_t *colmap = compact ? F->d_lab : nullptr;
	const uint32_t base = compact ? (uint32_t) F->rpad : 0u;
	const double t0 = wtime();
	// Y = C * A[p, :], dense 64-bit accumulators, then CSR
	unsigned long long *dY = (unsigned long long *) big_alloc((size_t) N * (size_t) m * sizeof(unsigned long long));
	HIP_CHECK(hipMemsetAsync(dY, 0, (size_t) N * m * sizeof(unsigned long long), stream));
	launch_combine(dA.p, dA.j, dA.x, d_rows, n, N, w, m, salt, dY, F->mont, stream, colmap, base, dA.nnz);
	if (compact && (1) != 0) {
		// the rows combined hold non-pivotal columns only: their combinations, reduced mod p, ARE the dense rows on those columns
		// (mk15.b4: 20-26 ms per 4,096 combinations through count / scan / pack / the elimination kernels, of a finish of 100)
		launch_dense_reduce_rows(dY, N, m, F->mont, d_S, ldS, stream);
		HIP_CHECK(hipStreamSynchronize(stream));
		big_free(dY);
		if (env_int("SPASM_HIP_COMPACT_CHECK", 0) != 0) {
			u32 *d_ref = (u32 *) big_alloc((size_t) N * (size_t) ldS * sizeof(u32));
			device_random_dense_rows(dA, d_rows, n, F, N, w, salt, d_ref, ldS, W, stream, false);
			std::vector<u32> x((size_t) N * (size_t) ldS), y((size_t) N * (size_t) ldS);
			HIP_CHECK(hipMemcpy(x.data(), d_S, x.size() * sizeof(u32), hipMemcpyDeviceToHost));
			HIP_CHECK(hipMemcpy(y.data(), d_ref, y.size() * sizeof(u32), hipMemcpyDeviceToHost));
			size_t differ = 0;
			for (size_t k = 0; k < (size_t) N; k++)
				for (size_t j = 0; j < (size_t) F->Sm; j++)
					differ += x[k * (size_t) ldS + j] != y[k * (size_t) ldS + j];
			logmsg("[dense rows/check] combinations on the non-pivotal columns, reduced directly, against the full path: %zu of %zu entries differ\n", differ, (size_t) N * (size_t) F->Sm);
			big_free(d_ref);
			if (differ != 0)
				die("combinations formed on the non-pivotal columns only differ from those formed on all columns (%zu entries)", differ);
		}
		if (verbose() >= 2)
			logmsg("[dense rows] %d combinations on the %d non-pivotal columns: %.3fs\n", N, m, wtime() - t0);
		return;
	}
	launch_dense_count(dY, N, m, (uint32_t) F->prime, W->d_row_len, stream);
	launch_row_scan(W->d_row_len, N, W->d_blocksum, W->d_Sp, stream);
	i64 ynnz = 0;
	HIP_CHECK(hipMemcpyAsync(&ynnz, W->d_Sp + N, sizeof(i64), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	int *dYj = (int *) big_alloc((size_t) (ynnz > 0 ? ynnz : 1) * sizeof(int));
	int *dYx = (int *) big_alloc((size_t) (ynnz > 0 ? ynnz : 1) * sizeof(int));
	launch_dense_pack(dY, N, m, (uint32_t) F->prime, W->d_Sp, dYj, dYx, stream, compact ? F->d_q : nullptr);
	i64 *dYp = dalloc<i64>((i64) N + 1);
	HIP_CHECK(hipMemcpyAsync(dYp, W->d_Sp, ((size_t) N + 1) * sizeof(i64), hipMemcpyDeviceToDevice, stream));
	std::vector<int> ident((size_t) N);
	for (int k = 0; k < N; k++)
		ident[k] = k;
	int *dident = dalloc<int>(N);
	HIP_CHECK(hipMemcpyAsync(dident, ident.data(), (size_t) N * sizeof(int), hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	const double t1 = wtime();
	big_free(dY);
	// a few very long rows: cut into pieces that are reduced side by side and added up (launch_split_rows)
	const int Sm = F->Sm;
	int pieces = 1;
	if (N <= 64 && ynnz / N >= 8192 && (1) != 0)
		pieces = (int) std::min<i64>(std::min<i64>(64, W->max_rows / N), (ynnz / N + 2047) / 2048);
	if (pieces > 1) {
		const int NP = N * pieces;
		i64 *dYp2 = dalloc<i64>((i64) NP + 1);
		launch_split_rows(dYp, N, pieces, dYp2, stream);
		std::vector<int> ident2((size_t) NP);
		for (int k = 0; k < NP; k++)
			ident2[k] = k;
		int *dident2 = dalloc<int>(NP);
		HIP_CHECK(hipMemcpyAsync(dident2, ident2.data(), (size_t) NP * sizeof(int), hipMemcpyHostToDevice, stream));
		u32 *parts = (u32 *) big_alloc((size_t) NP * (size_t) Sm * sizeof(u32));
		spasm_hip_dcsr dYcsr{NP, m_all, ynnz, dYp2, dYj, dYx};
		dschur_dense_impl(&dYcsr, dident2, NP, F, W, parts, Sm, stream, nullptr);
		launch_sum_pieces(parts, Sm, N, pieces, Sm, (uint32_t) F->prime, d_S, ldS, stream);
		HIP_CHECK(hipStreamSynchronize(stream));
		big_free(parts);
		sh::big_free(dident2);
		sh::big_free(dYp2);
	} else {
		spasm_hip_dcsr dYcsr{N, m_all, ynnz, dYp, dYj, dYx};
		dschur_dense_impl(&dYcsr, dident, N, F, W, d_S, ldS, stream, nullptr);
	}
	if (compact && env_int("SPASM_HIP_COMPACT_CHECK", 0) != 0) {
		// (tests) the same combinations with accumulators over all the columns: the dense rows must be the same
		u32 *d_ref = (u32 *) big_alloc((size_t) N * (size_t) ldS * sizeof(u32));
		device_random_dense_rows(dA, d_rows, n, F, N, w, salt, d_ref, ldS, W, stream, false);
		std::vector<u32> x((size_t) N * (size_t) ldS), y((size_t) N * (size_t) ldS);
		HIP_CHECK(hipMemcpy(x.data(), d_S, x.size() * sizeof(u32), hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(y.data(), d_ref, y.size() * sizeof(u32), hipMemcpyDeviceToHost));
		size_t differ = 0, nzx = 0, nzy = 0, first = (size_t) -1;
		for (size_t k = 0; k < (size_t) N; k++)
			for (size_t j = 0; j < (size_t) F->Sm; j++) {
				const size_t t = k * (size_t) ldS + j;
				nzx += x[t] != 0;
				nzy += y[t] != 0;
				if (x[t] != y[t]) {
					differ += 1;
					if (first == (size_t) -1)
						first = t;
				}
			}
		logmsg("[dense rows/check] compact against full accumulators: %zu of %zu entries differ (non-zero: %zu / %zu; first at row %zu column %zu; %" PRId64 " packed entries)\n",
		       differ, (size_t) N * (size_t) F->Sm, nzx, nzy, first == (size_t) -1 ? 0 : first / (size_t) ldS, first == (size_t) -1 ? 0 : first % (size_t) ldS, ynnz);
		big_free(d_ref);
		if (differ != 0)
			die("combinations formed on the non-pivotal columns only differ from those formed on all columns (%zu entries)", differ);
	}
	if (verbose() >= 2)
		logmsg("[dense rows] %d combinations: combine + pack %.3fs (%" PRId64 " entries), reduction %.3fs\n", N, t1 - t0, ynnz, wtime() - t1);
	sh::big_free(dident);
	sh::big_free(dYp);
	big_free(dYj);
	big_free(dYx);
}
}  // namespace sh


// --------------------------------------------------------------------------
// Device-resident dense finish (echelonize_dense / echelonize_dense_lowrank, spasm_echelonize.c:299-467).
//
// The reference loops: dense rows of the Schur complement (a block of rows, or random combinations of all of them)
// -> FFPACK RREF of the block -> the new pivotal rows are appended to U -> the next block is reduced by the larger U.
// Every trip goes through the host and, here, would rebuild the factor image of a U that has grown.  With the
// back-substituted image R of the factor at hand the whole loop lives in the space of its Sm non-pivotal columns:
// blocks are reduced by the ORIGINAL factor only (a_n - a_p R), stacked under the echelon rows E found so far, and one
// dense RREF of [E; Y] on the matrix cores both reduces Y by E and extends E.  A, R, E and the blocks never leave HBM;
// the host gets the final E once and appends it to U.  Same row space, same rank; the control flow (block size, weights,
// low-rank switch, completion test) is the reference's.
// --------------------------------------------------------------------------
namespace sh {

bool finish_on_device(const struct spasm_csr *A, const int *p, int n, struct spasm_lu *fact, struct echelonize_opts *opts,
                      bool lowrank_first)
{
	if (opts->L || env_int("SPASM_HIP_DEVICE_FINISH", 1) == 0 || n <= 0)
		return false;
	hipStream_t stream = nullptr;
	spasm_hip_dfact *F = cached_dfact(fact->U, fact->qinv, stream);
	const int Sm0 = F->Sm, m = A->m;
	F->bs.density_hint = 1.0;            // (this is the finish of a Schur complement that was found dense)
	if (Sm0 <= 0)
		return false;
	// With the back-substituted image R the blocks are a_n - a_p R; without it (Sm too wide for R, or another path forced)
	// they come from the row-by-row kernels.  Either way they are reduced by the ORIGINAL factor only: what the host loops
	// would do instead -- append every block's dense echelon rows to U, re-plan and re-upload the grown factor, reduce the
	// next block by it -- costs seconds per block on a wide remainder (ch8-8.b5: 0.4 -> 2.7 s per 1000 combinations).
	// SPASM_HIP_DEVICE_FINISH=2 keeps the round-2 rule (device finish only with R).
	// (the finish reduces all n rows in the end -- as blocks, or inside random combinations: R is judged on that)
	const bool have_R = backsolve_wanted(F, env_int("SPASM_HIP_FORCE_TIER", 0) != 0 || env_int("SPASM_HIP_GROUP", -1) >= 0, std::max(n, 1024));
	if (!have_R && env_int("SPASM_HIP_DEVICE_FINISH", 1) == 2)
		return false;
	if (have_R && !F->bs.valid)
		backsolve_build(F, stream);          // (the blocks below are small batches: they use R when it is there, they do not ask for it)
	const i64 prime = A->field->p;
	// The reference's block size (1000 rows by default) is sized for FFPACK on a CPU; here every block costs one dense RREF of
	// [E; Y] on top of the echelon rows found so far, so blocks of at least 4096 rows are taken (SPASM_HIP_DENSE_BLOCK=0:
	// exactly opts->dense_block_size, > 0: that many).  Rank and row space do not depend on it.
	const int block_env = env_int("SPASM_HIP_DENSE_BLOCK", -1);
	const int block = std::max(1, block_env > 0 ? block_env : block_env == 0 ? opts->dense_block_size : std::max(opts->dense_block_size, 4096));
	const int Sn_test = (int) std::ceil(128.0 / std::log2((double) prime));
	const double start = wtime();
	// A Schur complement that is resident as column slabs (round 6: schur_api.hip) stays that way: dense rows and random
	// combinations are linear in the columns, so every rank forms ITS columns of a block from its slab and the block is summed over
	// the ranks (the slabs are disjoint ranges of columns: exact) -- Sn x Sm words per block instead of the whole of S gathered
	// first; the echelon form is then extended on identical data on every rank, as before.
	const i64 ld_of_stack = Sm0;
	spasm_hip_comm *slab_comm = resident_slab_comm(A);
	DeviceMatrix devA(A, stream, true);
	const i64 annz = devA.nnz;
	int *drows = dalloc<int>(n);
	HIP_CHECK(hipMemcpy(drows, p, (size_t) n * sizeof(int), hipMemcpyHostToDevice));
	spasm_hip_dcsr dA{A->n, m, annz, devA.p, devA.j, devA.x};
	auto whole_block = [&](u32 *blockptr, int rows_of_block) {
		if (slab_comm != nullptr)
			comm_allreduce_sum_u32(slab_comm, blockptr, (i64) rows_of_block * ld_of_stack, stream);
	};
	const int maxblock = std::max(block, Sn_test);
	spasm_hip_dwork *W = spasm_hip_dwork_create(maxblock, m, 64);
	const i64 ld = Sm0;
	// the stack [E; Y]: room for the echelon rows found so far plus one block, grown on demand (the rank is only
	// bounded by min(n, Sm0), and a worst-case allocation would be Sm0^2 words: 45 GB on ch8-8.b5 for 3,900 rows used)
	i64 cap_rows = std::min<i64>((i64) std::min(n, Sm0) + maxblock, (i64) 3 * maxblock);
	{
		size_t free_b = 0, total_b = 0;
		sh::mem_info(&free_b, &total_b);
		if ((size_t) cap_rows * (size_t) ld * sizeof(u32) > free_b / 2) {
			spasm_hip_dwork_destroy(W);
			sh::big_free(drows);
			return false;                    // the blocked host loops work in dense_block_size x Sm pieces
		}
	}
	u32 *dM = (u32 *) big_alloc((size_t) cap_rows * (size_t) ld * sizeof(u32));
	int *dpiv = dalloc<int>(Sm0);
	bool out_of_memory = false;
	auto room_for = [&](int rows_added, int k_now) {
		if ((i64) k_now + rows_added <= cap_rows)
			return true;
		const i64 want = std::min<i64>((i64) std::min(n, Sm0) + maxblock, std::max<i64>(2 * cap_rows, (i64) k_now + rows_added));
		size_t free_b = 0, total_b = 0;
		sh::mem_info(&free_b, &total_b);
		if ((size_t) want * (size_t) ld * sizeof(u32) > free_b - free_b / 8) {
			out_of_memory = true;
			return false;
		}
		u32 *bigger = (u32 *) big_alloc((size_t) want * (size_t) ld * sizeof(u32));
		HIP_CHECK(hipMemcpyAsync(bigger, dM, (size_t) k_now * ld * sizeof(u32), hipMemcpyDeviceToDevice, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		big_free(dM);
		dM = bigger;
		cap_rows = want;
		return true;
	};
	// the generator of the random combinations is keyed by the problem, not by the history of the process: ranks of a sharded
	// run that replicate this finish draw the same combinations whatever else they have computed before
	uint64_t salt = 0x5DEECE66DULL ^ ((uint64_t) (uint32_t) n << 32) ^ (uint64_t) (uint32_t) Sm0 ^ ((uint64_t) prime * 0x9E3779B97F4A7C15ULL) ^
	                ((uint64_t) A->p[A->n] << 17);
	int k = 0;                           // echelon rows found so far: rows [0, k) of dM, in reduced form
	int rank_ub = std::min(n, Sm0);
	int processed = 0, round = 0;
	bool lowrank = lowrank_first;
	double t_rows = 0.0, t_rref = 0.0;
	// Wide remainders (tens of thousands of columns, rank a few thousand): the rows are added by ROW panels
	// (device_echelon_extend: one pass over the stack per 64 pivots, E is not factored again); narrow ones go through the
	// column-panel RREF of the whole stack [E; Y], which is at its best there.  SPASM_HIP_ROW_PANELS=0/1 forces the choice.
	const int rp_env = (-1);
	const bool row_panels = prime <= 65279 && (rp_env > 0 || (rp_env < 0 && Sm0 >= (16384)));
	const bool row_panels_later = prime <= 65279 && rp_env != 0 && (1) != 0;
	auto stack_and_reduce = [&](int rows_added) {
		const double t0 = wtime();
		const int rk = (row_panels || (row_panels_later && k > 0)) ? device_echelon_extend(prime, Sm0, dM, ld, k, rows_added, dpiv, stream)
		                          : spasm_hip_drref(prime, k + rows_added, Sm0, dM, ld, dpiv, stream);
		t_rref += wtime() - t0;
		const int rr = rk - k;
		k = rk;
		return rr;
	};
	if (!lowrank) {
		// echelonize_dense: the rows themselves, block by block
		logmsg("[echelonize/dense/device] dense schur complement of dimension %d x %d; block size=%d\n", n, Sm0, block);
		for (;;) {
			const int Sn = std::min(block, n - processed);
			if (Sn <= 0 || k >= Sm0 || !room_for(Sn, k))
				break;
			const double tr0 = wtime();
			dschur_dense_impl(&dA, drows + processed, Sn, F, W, dM + (i64) k * ld, ld, stream, nullptr);
			whole_block(dM + (i64) k * ld, Sn);
			t_rows += wtime() - tr0;
			const int rr = stack_and_reduce(Sn);
			logmsg("[echelonize/dense/device] round %d: S[%d:%d], %d new pivots (%d in all)\n", round, processed, processed + Sn, rr, k);
			round += 1;
			processed += Sn;
			rank_ub = std::min(n - processed + 0, Sm0 - k);
			if (opts->enable_tall_and_skinny && rr < opts->low_rank_ratio * Sn) {
				lowrank = true;
				break;
			}
		}
		if (!(lowrank && rank_ub > 0 && n - processed > 0))
			lowrank = false;
		else
			logmsg("[echelonize/dense/device] too few pivots; switching to low-rank mode\n");
	}
	if (lowrank) {
		// echelonize_dense_lowrank: random combinations of the remaining rows
		const int nleft = n - processed;
		const int *rows_left = drows + processed;
		// (rows of a Schur complement by this very factor hold non-pivotal columns only -- checked, once: the accumulators of the
		//  combinations then span Sm columns instead of m)
		const bool compact = m > 2 * Sm0 && (1) != 0 &&
		                     rows_are_nonpivotal(dA.p, dA.j, rows_left, nleft, F->d_lab, (uint32_t) F->rpad, stream);
		rank_ub = std::min(nleft, Sm0 - k);
		int w = (opts->low_rank_start_weight < 0) ? (int) std::ceil(-std::log(0.01) * nleft / (rank_ub > 0 ? rank_ub : 1))
		                                          : (int) opts->low_rank_start_weight;
		logmsg("[echelonize/dense/low-rank/device] dense schur complement of dimension %d x %d; block size=%d\n", nleft, Sm0 - k, block);
		for (;;) {
			const int Sn = std::min(rank_ub, block);
			if (Sn <= 0 || !room_for(std::max(Sn, Sn_test), k))
				break;
			salt += 0x9E3779B97F4A7C15ULL;
			const double tr0 = wtime();
			device_random_dense_rows(dA, rows_left, nleft, F, Sn, w, salt, dM + (i64) k * ld, ld, W, stream, compact);
			whole_block(dM + (i64) k * ld, Sn);
			t_rows += wtime() - tr0;
			int rr = stack_and_reduce(Sn);
			logmsg("[echelonize/dense/low-rank/device] round %d, weight %d, %d combinations: %d new pivots (%d in all)\n", round, w, Sn, rr, k);
			if (rr == 0) {
				// spasm_echelonize_test_completion (spasm_echelonize.c:30-52): a few combinations of ALL the rows
				salt += 0x9E3779B97F4A7C15ULL;
				device_random_dense_rows(dA, rows_left, nleft, F, Sn_test, 0, salt, dM + (i64) k * ld, ld, W, stream, compact);
				whole_block(dM + (i64) k * ld, Sn_test);
				rr = stack_and_reduce(Sn_test);
				if (rr == 0)
					break;
				logmsg("[echelonize/dense/low-rank/device] failed termination test; switching to full linear combinations\n");
				w = 0;
			}
			if (rr < 0.9 * Sn && w > 0)
				w *= 2;
			rank_ub -= rr;
			round += 1;
		}
	}
	// the echelon rows join U: pivot first (value 1), then the other entries, on the original columns.  They are packed
	// on the device (count, scan, pack: the stack is k x Sm words of which a tenth is non-zero -- 1.9 GB for 54 M entries on
	// ch8-8.b5, 0.44 s through a pageable copy and two host passes) and land in U's own arrays.
	const double t_tail0 = wtime();
	struct spasm_csr *U = fact->U;
	const int old_un = U->n;
	std::vector<int> piv((size_t) std::max(1, k));
	if (k > 0) {
		int *d_len = dalloc<int>(k);
		i64 *d_Up = dalloc<i64>((i64) k + 1);
		i64 *d_bsum = dalloc<i64>((k + 1023) / 1024 + 2);
		launch_echelon_count(dM, ld, Sm0, k, d_len, stream);
		launch_row_scan(d_len, k, d_bsum, d_Up, stream);
		std::vector<i64> Up((size_t) k + 1);
		HIP_CHECK(hipMemcpyAsync(Up.data(), d_Up, ((size_t) k + 1) * sizeof(i64), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipMemcpyAsync(piv.data(), dpiv, (size_t) k * sizeof(int), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		const i64 add = Up[(size_t) k];
		int *d_Uj = (int *) big_alloc((size_t) std::max<i64>(add, 1) * sizeof(int));
		int *d_Ux = (int *) big_alloc((size_t) std::max<i64>(add, 1) * sizeof(int));
		launch_echelon_pack(dM, ld, Sm0, k, dpiv, F->d_q, (uint32_t) prime, d_Up, d_Uj, d_Ux, stream);
		const i64 unz0 = U->p[U->n];
		spasm_hip_csr_realloc(U, unz0 + add);
		HIP_CHECK(hipMemcpyAsync(U->j + unz0, d_Uj, (size_t) add * sizeof(int), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipMemcpyAsync(U->x + unz0, d_Ux, (size_t) add * sizeof(int), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		const std::vector<int> &q0 = F->h_q;
		for (int i = 0; i < k; i++) {
			U->p[old_un + i + 1] = unz0 + Up[(size_t) i + 1];
			if (U->j[unz0 + Up[(size_t) i]] != q0[piv[i]] || U->x[unz0 + Up[(size_t) i]] != 1)
				die("finish_on_device: echelon row %d does not start with a unit pivot on column %d", i, q0[piv[i]]);
			fact->qinv[q0[piv[i]]] = old_un + i;
		}
		U->n = old_un + k;
		big_free(d_Uj);
		big_free(d_Ux);
		sh::big_free(d_len);
		sh::big_free(d_Up);
		sh::big_free(d_bsum);
	}
	const double t_tail1 = wtime();
	big_free(dM);
	sh::big_free(dpiv);
	spasm_hip_dwork_destroy(W);
	sh::big_free(drows);
	logmsg("[echelonize/dense/device] completed in %.2fs (dense rows %.2fs, %s %.2fs; %s). %d new pivots found\n", wtime() - start,
	       t_rows, row_panels ? "echelon rows by row panels" : "RREF", t_rref,
	       have_R ? "blocks from the back-substituted image" : "blocks from the row-by-row kernels", U->n - old_un);
	if (verbose() >= 2)
		logmsg("[echelonize/dense/device] of that: %.2fs packing the %d echelon rows on the device and appending them to U\n", t_tail1 - t_tail0, k);
	if (out_of_memory)
		logmsg("[echelonize/dense/device] the stack of echelon rows no longer fits in HBM: the host loops take over\n");
	return !out_of_memory;          // (the echelon rows found so far are in U either way)
}

}  // namespace sh

extern "C" {

int spasm_hip_dschur_dense(const spasm_hip_dcsr *A, const int *d_rows, int nrows, const spasm_hip_dfact *F,
                           spasm_hip_dwork *W, u32 *d_S, i64 ldS, void *stream)
{
	return dschur_dense_impl(A, d_rows, nrows, F, W, d_S, ldS, stream, nullptr);
}

// --------------------------------------------------------------------------
// host-pointer drop-ins
// --------------------------------------------------------------------------
void spasm_hip_schur_dense(const struct spasm_csr *A, const int *p, int n, const int *p_in, struct spasm_lu *fact, void *S,
                           spasm_datatype datatype, int *q, int *p_out)
{
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_schur_dense: no HIP device (this library has no CPU path)");
	const int m = A->m;
	const i64 prime = A->field->p;
	const double t0 = wtime();
	hipStream_t stream = nullptr;
	spasm_hip_dfact *F = cached_dfact(fact->U, fact->qinv, stream);
	const int Sm = F->Sm;
	for (int l = 0; l < Sm; l++)
		q[l] = F->h_q[l];
	for (int k = 0; k < n; k++)
		p_out[k] = (p_in != nullptr) ? p_in[p[k]] : p[k];
	if (n > 0 && Sm > 0) {
		const i64 annz = A->p[A->n];
		DeviceMatrix devA(A, stream);
		int *drows = dalloc<int>(n);
		HIP_CHECK(hipMemcpy(drows, p, (size_t) n * sizeof(int), hipMemcpyHostToDevice));
		spasm_hip_dcsr dA{A->n, m, annz, devA.p, devA.j, devA.x};
		spasm_hip_dwork *W = spasm_hip_dwork_create(n, m, 64);
		u32 *dS = dalloc<u32>((i64) n * Sm);
		struct spasm_triplet *L = fact->Ltmp;
		if (L == nullptr) {
			dschur_dense_impl(&dA, drows, n, F, W, dS, Sm, stream, nullptr);
		} else {
			// elimination coefficients -> L, pool grown on demand
			int *d_row_orig = dalloc<int>(n);
			HIP_CHECK(hipMemcpy(d_row_orig, p_out, (size_t) n * sizeof(int), hipMemcpyHostToDevice));
			i64 lcap = std::max<i64>((i64) n * 1024, (i64) 1 << 22);
			for (;;) {
				LOut lout;
				lout.row_orig = d_row_orig;
				lout.cap = lcap;
				lout.Li = dalloc<int>(lcap);
				lout.Lj = dalloc<int>(lcap);
				lout.Lx = dalloc<int>(lcap);
				HIP_CHECK(hipMemset(lout.Li, 0xFF, (size_t) lcap * sizeof(int)));
				const int rc = dschur_dense_impl(&dA, drows, n, F, W, dS, Sm, stream, &lout);
				if (rc == 0) {
					const i64 used = std::min(lout.used, lcap);
					std::vector<int> hi((size_t) (used > 0 ? used : 1)), hj(hi.size()), hx(hi.size());
					if (used > 0) {
						HIP_CHECK(hipMemcpy(hi.data(), lout.Li, (size_t) used * sizeof(int), hipMemcpyDeviceToHost));
						HIP_CHECK(hipMemcpy(hj.data(), lout.Lj, (size_t) used * sizeof(int), hipMemcpyDeviceToHost));
						HIP_CHECK(hipMemcpy(hx.data(), lout.Lx, (size_t) used * sizeof(int), hipMemcpyDeviceToHost));
					}
					i64 extra = 0;
					for (i64 t = 0; t < used; t++)
						extra += hi[t] >= 0;
					if (L->nz + extra > L->nzmax)
						spasm_hip_triplet_realloc(L, 2 * L->nzmax + extra);
					for (i64 t = 0; t < used; t++) {
						if (hi[t] < 0)
							continue;
						L->i[L->nz] = hi[t];
						L->j[L->nz] = hj[t];
						L->x[L->nz] = hx[t];
						L->nz += 1;
					}
				}
				sh::big_free(lout.Li);
				sh::big_free(lout.Lj);
				sh::big_free(lout.Lx);
				if (rc == 0)
					break;
				lcap *= 4;
			}
			sh::big_free(d_row_orig);
		}
		std::vector<u32> h((size_t) n * Sm);
		HIP_CHECK(hipMemcpy(h.data(), dS, (size_t) n * Sm * sizeof(u32), hipMemcpyDeviceToHost));
		const u32 half = (u32) (prime / 2);
		for (i64 t = 0; t < (i64) n * Sm; t++) {
			const i64 v = (h[t] > half) ? (i64) h[t] - prime : (i64) h[t];
			switch (datatype) {
			case SPASM_DOUBLE: ((double *) S)[t] = (double) v; break;
			case SPASM_FLOAT: ((float *) S)[t] = (float) v; break;
			case SPASM_I64: ((i64 *) S)[t] = v; break;
			}
		}
		sh::big_free(dS);
		spasm_hip_dwork_destroy(W);
		sh::big_free(drows);
	}
	logmsg("[schur/dense/hip] %d x %d dense rows in %.1fs\n", n, Sm, wtime() - t0);
}

// replaces spasm_schur_dense_randomized (spasm_schur.c:346-425): N random combinations of the rows
// p[0..n) of A (w > 0: of w random rows each, first coefficient 1; w <= 0: of all the rows), reduced
// by U, as dense rows on the non-pivotal columns.  Everything runs on the device.
void spasm_hip_schur_dense_randomized(const struct spasm_csr *A, const int *p, int n, const struct spasm_csr *U,
                                      const int *qinv, void *S, spasm_datatype datatype, int *q, int N, int w)
{
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_schur_dense_randomized: no HIP device (this library has no CPU path)");
	if (p == nullptr || n <= 0)
		die("spasm_hip_schur_dense_randomized: empty row list");
	static uint64_t call_id = 0;       // every call draws fresh combinations
	call_id += 1;
	const int m = A->m;
	const i64 prime = A->field->p;
	const double t0 = wtime();
	hipStream_t stream = nullptr;
	spasm_hip_dfact *F = cached_dfact(U, qinv, stream);
	const int Sm = F->Sm;
	for (int l = 0; l < Sm; l++)
		q[l] = F->h_q[l];
	if (N > 0 && Sm > 0) {
		const i64 annz = A->p[A->n];
		DeviceMatrix devA(A, stream);
		int *drows = dalloc<int>(n);
		HIP_CHECK(hipMemcpy(drows, p, (size_t) n * sizeof(int), hipMemcpyHostToDevice));
		spasm_hip_dcsr dA{A->n, m, annz, devA.p, devA.j, devA.x};
		spasm_hip_dwork *W = spasm_hip_dwork_create(N, m, 64);
		u32 *dS = dalloc<u32>((i64) N * Sm);
		device_random_dense_rows(dA, drows, n, F, N, w, call_id * 0x9E3779B97F4A7C15ULL, dS, Sm, W, stream, false);
		std::vector<u32> h((size_t) N * Sm);
		HIP_CHECK(hipMemcpy(h.data(), dS, (size_t) N * Sm * sizeof(u32), hipMemcpyDeviceToHost));
		const u32 half = (u32) (prime / 2);
		for (i64 t = 0; t < (i64) N * Sm; t++) {
			const i64 v = (h[t] > half) ? (i64) h[t] - prime : (i64) h[t];
			switch (datatype) {
			case SPASM_DOUBLE: ((double *) S)[t] = (double) v; break;
			case SPASM_FLOAT: ((float *) S)[t] = (float) v; break;
			case SPASM_I64: ((i64 *) S)[t] = v; break;
			}
		}
		sh::big_free(dS);
		spasm_hip_dwork_destroy(W);
		sh::big_free(drows);
	}
	logmsg("[schur/dense/random/hip] %d combinations (weight %d) of %d rows, %d columns, %.1fs\n", N, w, n, Sm, wtime() - t0);
}

// Same contract as the reference wrapper (spasm_ffpack.cpp:23-49 as consumed by
// update_U_after_rref, spasm_echelonize.c:189-222): returns the rank r; qinv[0..r) are the pivot
// columns in echelon-row order, qinv[r..m) the other columns (increasing); for i < r and k >= r,
// A[i*ldA + k] is the coefficient of echelon row i on column qinv[k]; A[i*ldA + k] for k < r is
// the identity.
int spasm_hip_ffpack_rref(i64 prime, int n, int m, void *A, int ldA, spasm_datatype datatype, size_t *qinv)
{
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_ffpack_rref: no HIP device (this library has no CPU path)");
	const double t0 = wtime();
	for (int j = 0; j < m; j++)
		qinv[j] = (size_t) j;
	if (n == 0 || m == 0)
		return 0;
	std::vector<u32> h((size_t) n * m);
	// rows in parallel: the conversion of a tall block would otherwise cost more than its elimination
	auto rows_in_parallel = [&](int nrows, auto &&body) {
		int nt = (int) std::min<i64>(16, std::max<i64>(1, (i64) nrows * m / (1 << 20)));
		nt = std::min(nt, std::max(1, (int) std::thread::hardware_concurrency()));
		if (nt <= 1) {
			body(0, nrows);
			return;
		}
		std::vector<std::thread> pool;
		for (int t = 0; t < nt; t++)
			pool.emplace_back([&, t]() { body((int) ((i64) nrows * t / nt), (int) ((i64) nrows * (t + 1) / nt)); });
		for (auto &th : pool)
			th.join();
	};
	rows_in_parallel(n, [&](int lo, int hi) {
		for (int i = lo; i < hi; i++)
			for (int j = 0; j < m; j++) {
				const size_t src = (size_t) i * ldA + j;
				i64 v = 0;
				switch (datatype) {
				case SPASM_DOUBLE: v = (i64) ((double *) A)[src]; break;
				case SPASM_FLOAT: v = (i64) ((float *) A)[src]; break;
				case SPASM_I64: v = ((i64 *) A)[src]; break;
				}
				if (v >= prime || v <= -prime)
					v %= prime;
				if (v < 0)
					v += prime;
				h[(size_t) i * m + j] = (u32) v;
			}
	});
	u32 *dA = dalloc<u32>((i64) n * m);
	int *dpiv = dalloc<int>(m);
	HIP_CHECK(hipMemcpy(dA, h.data(), (size_t) n * m * sizeof(u32), hipMemcpyHostToDevice));
	const int r = spasm_hip_drref(prime, n, m, dA, m, dpiv, nullptr);
	if (r > 0)          // (rows r.. are zero)
		HIP_CHECK(hipMemcpy(h.data(), dA, (size_t) r * m * sizeof(u32), hipMemcpyDeviceToHost));
	std::vector<int> pivcol((size_t) (r > 0 ? r : 1));
	if (r > 0)
		HIP_CHECK(hipMemcpy(pivcol.data(), dpiv, (size_t) r * sizeof(int), hipMemcpyDeviceToHost));
	sh::big_free(dA);
	sh::big_free(dpiv);
	std::vector<char> is_piv((size_t) m, 0);
	for (int i = 0; i < r; i++) {
		qinv[i] = (size_t) pivcol[i];
		is_piv[pivcol[i]] = 1;
	}
	int k = r;
	for (int j = 0; j < m; j++)
		if (!is_piv[j])
			qinv[k++] = (size_t) j;
	const u32 half = (u32) (prime / 2);
	const size_t esize = (datatype == SPASM_FLOAT) ? sizeof(float) : 8;
	rows_in_parallel(n, [&](int lo, int hi) {
		for (int i = lo; i < hi; i++) {
			if (i >= r) {          // zero row: all-zero bits in every datatype
				std::memset((char *) A + (size_t) i * ldA * esize, 0, (size_t) m * esize);
				continue;
			}
			for (int kk = 0; kk < m; kk++) {
				const u32 raw = h[(size_t) i * m + qinv[kk]];
				const i64 v = (raw > half) ? (i64) raw - prime : (i64) raw;
				const size_t dst = (size_t) i * ldA + kk;
				switch (datatype) {
				case SPASM_DOUBLE: ((double *) A)[dst] = (double) v; break;
				case SPASM_FLOAT: ((float *) A)[dst] = (float) v; break;
				case SPASM_I64: ((i64 *) A)[dst] = v; break;
				}
			}
		}
	});
	logmsg("[rref/hip] %d x %d mod %" PRId64 ": rank %d [%.1fs]\n", n, m, prime, r, wtime() - t0);
	return r;
}

// replaces spasm_ffpack_LU (spasm_ffpack.cpp:57-86, 137-145).  On return (r = rank): p[i] = original row
// of packed row i, qinv[j] = original column of packed column j; A[i*ldA + j] holds L for
// j < min(i + 1, r) and, on the rows i < r, U for j > i (unit diagonal implied); A == L * U.
int spasm_hip_ffpack_LU(i64 prime, int n, int m, void *A, int ldA, spasm_datatype datatype, size_t *p, size_t *qinv)
{
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_ffpack_LU: no HIP device (this library has no CPU path)");
	const double t0 = wtime();
	for (int i = 0; i < n; i++)
		p[i] = (size_t) i;
	for (int j = 0; j < m; j++)
		qinv[j] = (size_t) j;
	if (n == 0 || m == 0)
		return 0;
	std::vector<u32> h((size_t) n * m);
	for (int i = 0; i < n; i++)
		for (int j = 0; j < m; j++) {
			const size_t src = (size_t) i * ldA + j;
			i64 v = 0;
			switch (datatype) {
			case SPASM_DOUBLE: v = (i64) ((double *) A)[src]; break;
			case SPASM_FLOAT: v = (i64) ((float *) A)[src]; break;
			case SPASM_I64: v = ((i64 *) A)[src]; break;
			}
			v %= prime;
			if (v < 0)
				v += prime;
			h[(size_t) i * m + j] = (u32) v;
		}
	std::vector<int> hp((size_t) n), hq((size_t) m);
	for (int i = 0; i < n; i++)
		hp[i] = i;
	for (int j = 0; j < m; j++)
		hq[j] = j;
	u32 *dA = dalloc<u32>((i64) n * m);
	int *dP = dalloc<int>(n), *dQ = dalloc<int>(m);
	HIP_CHECK(hipMemcpy(dA, h.data(), (size_t) n * m * sizeof(u32), hipMemcpyHostToDevice));
	HIP_CHECK(hipMemcpy(dP, hp.data(), (size_t) n * sizeof(int), hipMemcpyHostToDevice));
	HIP_CHECK(hipMemcpy(dQ, hq.data(), (size_t) m * sizeof(int), hipMemcpyHostToDevice));
	const int r = device_lu(prime, n, m, dA, m, dP, dQ, nullptr);
	HIP_CHECK(hipMemcpy(h.data(), dA, (size_t) n * m * sizeof(u32), hipMemcpyDeviceToHost));
	HIP_CHECK(hipMemcpy(hp.data(), dP, (size_t) n * sizeof(int), hipMemcpyDeviceToHost));
	HIP_CHECK(hipMemcpy(hq.data(), dQ, (size_t) m * sizeof(int), hipMemcpyDeviceToHost));
	sh::big_free(dA);
	sh::big_free(dP);
	sh::big_free(dQ);
	for (int i = 0; i < n; i++)
		p[i] = (size_t) hp[i];
	for (int j = 0; j < m; j++)
		qinv[j] = (size_t) hq[j];
	const u32 half = (u32) (prime / 2);
	for (int i = 0; i < n; i++)
		for (int j = 0; j < m; j++) {
			const u32 raw = h[(size_t) i * m + j];
			const i64 v = (raw > half) ? (i64) raw - prime : (i64) raw;
			const size_t dst = (size_t) i * ldA + j;
			switch (datatype) {
			case SPASM_DOUBLE: ((double *) A)[dst] = (double) v; break;
			case SPASM_FLOAT: ((float *) A)[dst] = (float) v; break;
			case SPASM_I64: ((i64 *) A)[dst] = v; break;
			}
		}
	logmsg("[LU/hip] %d x %d mod %" PRId64 ": rank %d [%.1fs]\n", n, m, prime, r, wtime() - t0);
	return r;
}


// Test hook: the N combinations launch_combine forms of the rows rows[0..nrows) of the host matrix A (w > 0: of w random rows
// each; w <= 0: of all the rows -- the kernels of the completion test of a low-rank finish), as N x m residues in [0, p) on the
// host.  The coefficients come from the counter-based generator of dense_kernels.hip (splitmix64 of (salt, k, t)), which
// tests/test_gpu_dense.py restates in numpy: the sums are checked against a product computed without this library.
void spasm_hip_debug_combine(const struct spasm_csr *A, const int *rows, int nrows, int N, int w, uint64_t salt, u32 *out)
{
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_debug_combine: no HIP device (this library has no CPU path)");
	hipStream_t stream = nullptr;
	const int m = A->m;
	DeviceMatrix dA(A, stream);
	int *d_rows = static_cast<int *>(sh::big_alloc((size_t) (nrows > 0 ? nrows : 1) * sizeof(int)));
	HIP_CHECK(hipMemcpyAsync(d_rows, rows, (size_t) nrows * sizeof(int), hipMemcpyHostToDevice, stream));
	const size_t count = (size_t) N * (size_t) m;
	unsigned long long *dY = static_cast<unsigned long long *>(sh::big_alloc(count * sizeof(unsigned long long)));
	HIP_CHECK(hipMemsetAsync(dY, 0, count * sizeof(unsigned long long), stream));
	launch_combine(dA.p, dA.j, dA.x, d_rows, nrows, N, w, m, salt, dY, mont_setup(A->field->p), stream, nullptr, 0, (int64_t) A->p[A->n]);
	std::vector<unsigned long long> h(count);
	HIP_CHECK(hipMemcpyAsync(h.data(), dY, count * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	const unsigned long long p = (unsigned long long) A->field->p;
	for (size_t t = 0; t < count; t++)
		out[t] = (u32) (h[t] % p);
	sh::big_free(dY);
	sh::big_free(d_rows);
}

}  // extern "C"
