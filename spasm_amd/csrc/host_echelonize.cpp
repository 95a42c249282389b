// The echelonization driver (replaces spasm_echelonize.c).  Control flow and
// options are the reference's; every arithmetic stage is a GPU call:
//   rounds of [structural pivots on the host -> sparse Schur complement on the
//   GPU], then a finishing stage: dense blocks (dense Schur rows + RREF on the
//   GPU), the low-rank variant (random combinations, then the same), or --
//   where the reference runs its sequential GPLU loop -- further structural
//   rounds on the GPU until the remainder is empty.
#include <cinttypes>
#include <cmath>
#include <mutex>
#include <vector>

#include "device_types.h"
#include "sha256.h"

namespace sh {
// device-resident dense / low-rank finish (dense_api.hip); false: not applicable, use the loops below
bool finish_on_device(const struct spasm_csr *A, const int *p, int n, struct spasm_lu *fact, struct echelonize_opts *opts,
                      bool lowrank_first);
}  // namespace sh

using namespace sh;

static int env_int_host(const char *name, int dflt)
{
	const char *e = sh::env_get(name);
	return (e == nullptr || *e == 0) ? dflt : std::atoi(e);
}

// --------------------------------------------------------------------------
// SHA-256 counter-mode generator (same stream as spasm_prng.c, so that seeded
// runs draw the same coefficients as the reference)
// --------------------------------------------------------------------------
namespace sh {

struct Prng {
	uint8_t block[44];
	uint8_t hash[32];
	uint32_t prime, mask;
	uint32_t counter;
	int pos;

	static void be32(uint8_t *dst, uint32_t v)
	{
		dst[0] = (uint8_t) (v >> 24);
		dst[1] = (uint8_t) (v >> 16);
		dst[2] = (uint8_t) (v >> 8);
		dst[3] = (uint8_t) v;
	}

	void rehash()
	{
		Sha256 h;
		h.reset();
		h.update(block, 44);
		h.finish(hash);
		counter += 1;
		be32(block + 36, counter);
		pos = 0;
	}

	void seed(i64 p, uint64_t s, uint32_t seq)
	{
		std::memset(block, 0, sizeof(block));
		be32(block + 0, (uint32_t) (s & 0xffffffffu));
		be32(block + 4, (uint32_t) (s >> 32));
		prime = (uint32_t) p;
		i64 m = 1;
		while (m < p)
			m <<= 1;
		mask = (uint32_t) (m - 1);
		be32(block + 32, (uint32_t) p);
		be32(block + 40, seq);
		counter = 0;
		rehash();
	}

	uint32_t next_u32()
	{
		if (pos == 8)
			rehash();
		const uint8_t *b = hash + 4 * pos;
		pos += 1;
		return ((uint32_t) b[0] << 24) | ((uint32_t) b[1] << 16) | ((uint32_t) b[2] << 8) | b[3];
	}

	spasm_ZZp next_zp()
	{
		for (;;) {
			uint32_t x = next_u32() & mask;
			if (x < prime)
				return zp_init(prime, x);
		}
	}
};

}  // namespace sh

extern "C" {

// exported for the tests: the first `count` values of the stream (prime, seed, seq)
void spasm_hip_debug_prng(i64 prime, uint64_t seed, uint32_t seq, int count, spasm_ZZp *out)
{
	Prng g;
	g.seed(prime, seed, seq);
	for (int i = 0; i < count; i++)
		out[i] = g.next_zp();
}

// replaces spasm_schur_estimate_density (spasm_schur.c:11-48): average density of R sampled rows
double spasm_hip_schur_estimate_density(const struct spasm_csr *A, const int *p, int n, const struct spasm_csr *U,
                                        const int *qinv, int R)
{
	if (n == 0)
		return 0;
	std::vector<int> sample((size_t) R);
	uint64_t state = 0x9E3779B97F4A7C15ULL ^ (uint64_t) n;
	for (int t = 0; t < R; t++) {
		state = state * 6364136223846793005ULL + 1442695040888963407ULL;
		sample[t] = p[(state >> 33) % (uint64_t) n];
	}
	struct spasm_lu tmp;
	tmp.r = U->n;
	tmp.complete = false;
	tmp.L = nullptr;
	tmp.U = (struct spasm_csr *) U;
	tmp.qinv = (int *) qinv;
	tmp.p = nullptr;
	tmp.Ltmp = nullptr;
	const int keep = verbose();
	struct spasm_csr *S = spasm_hip_schur(A, sample.data(), R, &tmp, 0.0, nullptr, nullptr, nullptr);
	(void) keep;
	const i64 nnz = S->p[S->n];
	spasm_hip_csr_free(S);
	const int Sm = A->m - U->n;
	return (Sm > 0) ? ((double) nnz) / Sm / R : 0.0;
}

}  // extern "C"

// --------------------------------------------------------------------------
// the driver
// --------------------------------------------------------------------------
namespace {

// append the echelon rows of a dense RREF to U (update_U_after_rref, spasm_echelonize.c:189-222)
void absorb_rref(int rr, int Sm, const void *S, spasm_datatype datatype, const size_t *Sqinv, const int *q,
                 struct spasm_lu *fact)
{
	struct spasm_csr *U = fact->U;
	i64 unz = U->p[U->n];
	spasm_hip_csr_realloc(U, unz + (i64) (1 + Sm - rr) * rr);
	for (i64 i = 0; i < rr; i++) {
		const int jp = q[Sqinv[i]];
		U->j[unz] = jp;
		U->x[unz] = 1;
		unz += 1;
		fact->qinv[jp] = U->n;
		for (i64 k = rr; k < Sm; k++) {
			const spasm_ZZp x = spasm_hip_datatype_read(S, (size_t) (i * Sm + k), datatype);
			if (x == 0)
				continue;
			U->j[unz] = q[Sqinv[k]];
			U->x[unz] = x;
			unz += 1;
		}
		U->n += 1;
		U->p[U->n] = unz;
	}
}

// transfer a dense PLUQ to the factorization (update_fact_after_LU, spasm_echelonize.c:228-297)
void absorb_LU(int n, int Sm, int r, const void *S, spasm_datatype datatype, const size_t *Sp, const size_t *Sqinv,
               const int *q, const int *p_in, i64 lnz_before, bool complete, std::vector<char> &pivotal,
               struct spasm_lu *fact)
{
	struct spasm_csr *U = fact->U;
	struct spasm_triplet *L = fact->Ltmp;
	i64 unz = U->p[U->n];
	spasm_hip_csr_realloc(U, unz + (i64) (1 + 2 * (i64) Sm - r) * r);
	if (!complete) {
		// keep, among the coefficients recorded by the dense Schur rows, those of the pivotal rows only
		for (i64 i = 0; i < r; i++)
			pivotal[p_in[Sp[i]]] = 1;
		i64 w = lnz_before;
		for (i64 px = lnz_before; px < L->nz; px++) {
			if (!pivotal[L->i[px]])
				continue;
			L->i[w] = L->i[px];
			L->j[w] = L->j[px];
			L->x[w] = L->x[px];
			w += 1;
		}
		L->nz = w;
	}
	const i64 rows_L = complete ? n : r;
	for (i64 i = 0; i < rows_L; i++) {
		const int iorig = p_in[Sp[i]];
		const i64 jmax = (i + 1 < r) ? i + 1 : r;
		for (i64 j = 0; j < jmax; j++) {
			const spasm_ZZp v = spasm_hip_datatype_read(S, (size_t) (i * Sm + j), datatype);
			if (v == 0)
				continue;
			if (L->nz == L->nzmax)
				spasm_hip_triplet_realloc(L, 2 * L->nzmax + Sm);
			L->i[L->nz] = iorig;
			L->j[L->nz] = U->n + (int) j;
			L->x[L->nz] = v;
			L->nz += 1;
		}
		if (i < r)
			fact->p[U->n + i] = iorig;
	}
	for (i64 i = 0; i < r; i++) {
		const int jp = q[Sqinv[i]];
		U->j[unz] = jp;
		U->x[unz] = 1;
		unz += 1;
		fact->qinv[jp] = U->n;
		for (i64 j = i + 1; j < Sm; j++) {
			const spasm_ZZp x = spasm_hip_datatype_read(S, (size_t) (i * Sm + j), datatype);
			if (x == 0)
				continue;
			U->j[unz] = q[Sqinv[j]];
			U->x[unz] = x;
			unz += 1;
		}
		U->n += 1;
		U->p[U->n] = unz;
	}
}

// spasm_echelonize_test_completion (spasm_echelonize.c:30-52)
bool remainder_is_zero(const struct spasm_csr *A, const int *p, int n, struct spasm_lu *fact)
{
	if (n == 0 || A->p[A->n] == 0)
		return true;
	const int m = A->m;
	const i64 Sm = m - fact->U->n;
	const i64 prime = A->field->p;
	if (Sm <= 0)
		return true;
	const spasm_datatype dt = spasm_hip_datatype_choose(prime);
	const int Sn = (int) std::ceil(128.0 / std::log2((double) prime));
	std::vector<unsigned char> S((size_t) Sn * Sm * spasm_hip_datatype_size(dt));
	std::vector<int> q((size_t) Sm);
	std::vector<size_t> Sp((size_t) Sm);
	logmsg("[echelonize/completion] testing completion with %d random linear combinations (rank %d)\n", Sn, fact->U->n);
	spasm_hip_schur_dense_randomized(A, p, n, fact->U, fact->qinv, S.data(), dt, q.data(), Sn, 0);
	const int rr = spasm_hip_ffpack_rref(prime, Sn, (int) Sm, S.data(), (int) Sm, dt, Sp.data());
	return rr == 0;
}

// echelonize_dense_lowrank (spasm_echelonize.c:299-372)
void finish_lowrank(const struct spasm_csr *A, const int *p, int n, struct spasm_lu *fact, struct echelonize_opts *opts)
{
	struct spasm_csr *U = fact->U;
	const int m = A->m;
	int Sm = m - U->n;
	const i64 prime = A->field->p;
	const spasm_datatype dt = spasm_hip_datatype_choose(prime);
	const int block = opts->dense_block_size;
	std::vector<unsigned char> S((size_t) block * Sm * spasm_hip_datatype_size(dt));
	std::vector<int> q((size_t) Sm);
	std::vector<size_t> Sp((size_t) Sm);
	const double start = wtime();
	const int old_un = U->n;
	int round = 0;
	int rank_ub = (n < Sm) ? n : Sm;
	logmsg("[echelonize/dense/low-rank] dense schur complement of dimension %d x %d; block size=%d\n", n, Sm, block);
	int w = (opts->low_rank_start_weight < 0) ? (int) std::ceil(-std::log(0.01) * n / (rank_ub > 0 ? rank_ub : 1))
	                                          : (int) opts->low_rank_start_weight;
	for (;;) {
		int Sn = (rank_ub < block) ? rank_ub : block;
		if (Sn <= 0)
			break;
		logmsg("[echelonize/dense/low-rank] round %d, weight %d, chunk %d x %d\n", round, w, Sn, Sm);
		spasm_hip_schur_dense_randomized(A, p, n, U, fact->qinv, S.data(), dt, q.data(), Sn, w);
		const int rr = spasm_hip_ffpack_rref(prime, Sn, Sm, S.data(), Sm, dt, Sp.data());
		if (rr == 0) {
			if (remainder_is_zero(A, p, n, fact))
				break;
			logmsg("[echelonize/dense/low-rank] failed termination test; switching to full linear combinations\n");
			w = 0;
		}
		if (rr < 0.9 * Sn && w > 0)
			w *= 2;
		absorb_rref(rr, Sm, S.data(), dt, Sp.data(), q.data(), fact);
		Sm -= rr;
		rank_ub -= rr;
		round += 1;
	}
	logmsg("[echelonize/dense/low-rank] completed in %.1fs. %d new pivots found\n", wtime() - start, U->n - old_un);
}

// echelonize_dense (spasm_echelonize.c:379-467), L not recorded
void finish_dense(const struct spasm_csr *A, const int *p, int n, const int *p_in, struct spasm_lu *fact,
                  struct echelonize_opts *opts)
{
	struct spasm_csr *U = fact->U;
	const int m = A->m;
	int Sm = m - U->n;
	const i64 prime = A->field->p;
	const spasm_datatype dt = spasm_hip_datatype_choose(prime);
	const int block = opts->dense_block_size;
	std::vector<unsigned char> S((size_t) block * Sm * spasm_hip_datatype_size(dt));
	std::vector<int> p_out((size_t) block), q((size_t) (Sm > 0 ? Sm : 1));
	std::vector<size_t> Sqinv((size_t) (Sm > 0 ? Sm : 1)), Sp((size_t) block);
	std::vector<char> pivotal((size_t) (opts->L ? fact->Ltmp->n + 1 : 1), 0);
	int processed = 0, round = 0;
	const double start = wtime();
	const int old_un = U->n;
	bool lowrank = false;
	int rank_ub = std::min(A->n - U->n, A->m - U->n);
	logmsg("[echelonize/dense] dense schur complement of dimension %d x %d; block size=%d\n", n, Sm, block);
	for (;;) {
		const int Sn = std::min(block, n - processed);
		if (Sn <= 0 || Sm <= 0)
			break;
		logmsg("[echelonize/dense] round %d. processing S[%d:%d] (%d x %d)\n", round, processed, processed + Sn, Sn, Sm);
		const i64 lnz_before = opts->L ? fact->Ltmp->nz : -1;
		spasm_hip_schur_dense(A, p, Sn, p_in, fact, S.data(), dt, q.data(), p_out.data());
		int rr;
		if (opts->L) {
			rr = spasm_hip_ffpack_LU(prime, Sn, Sm, S.data(), Sm, dt, Sp.data(), Sqinv.data());
			absorb_LU(Sn, Sm, rr, S.data(), dt, Sp.data(), Sqinv.data(), q.data(), p_out.data(), lnz_before, opts->complete,
			          pivotal, fact);
		} else {
			rr = spasm_hip_ffpack_rref(prime, Sn, Sm, S.data(), Sm, dt, Sqinv.data());
			absorb_rref(rr, Sm, S.data(), dt, Sqinv.data(), q.data(), fact);
		}
		round += 1;
		processed += Sn;
		p += Sn;
		Sm = m - U->n;
		rank_ub = std::min(A->n - U->n, A->m - U->n);
		if (opts->enable_tall_and_skinny && rr < opts->low_rank_ratio * Sn) {
			lowrank = true;
			break;
		}
	}
	if (rank_ub > 0 && n - processed > 0 && lowrank) {
		logmsg("[echelonize/dense] too few pivots; switching to low-rank mode\n");
		finish_lowrank(A, p, n - processed, fact, opts);
	} else {
		logmsg("[echelonize/dense] completed in %.1fs. %d new pivots found\n", wtime() - start, U->n - old_un);
	}
}

}  // namespace

// where the last spasm_hip_echelonize call spent its time (seconds): read by bench.py / tools through
// spasm_hip_echelonize_profile
static double g_prof[8];

extern "C" void spasm_hip_echelonize_profile(double *out)
{
	for (int k = 0; k < 8; k++)
		out[k] = g_prof[k];
}

namespace {
extern "C" int spasm_hip_echelonize_counters(long long *out, int count)
{
	for (int k = 0; k < count && k < CNT_COUNT; k++)
		out[k] = counters()[k];
	return CNT_COUNT;
}

struct Stopwatch {          // adds the time of its scope to a slot of g_prof
	double t0;
	int slot;
	explicit Stopwatch(int s) : t0(wtime()), slot(s) {}
	~Stopwatch() { g_prof[slot] += wtime() - t0; }
};
}  // namespace

// The driver keeps process-wide state between its stages (the table of device-resident matrices, the cached factor image,
// the profile above): calls from several threads are taken one at a time.  (The device-level entry points -- spasm_hip_d* --
// keep their state in the handles they are given and may run concurrently on different handles.)
static std::mutex g_driver_mutex;

extern "C" struct spasm_lu *spasm_hip_echelonize(const struct spasm_csr *A0, struct echelonize_opts *opts)
{
	std::lock_guard<std::mutex> one_at_a_time(g_driver_mutex);
	for (int k = 0; k < 8; k++)
		g_prof[k] = 0.0;
	for (int k = 0; k < CNT_COUNT; k++)
		counters()[k] = 0;
	resident_begin();          // A, and every Schur complement after it, stays in HBM between the calls below
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_echelonize: no HIP device (this library has no CPU path)");
	struct echelonize_opts dflt;
	if (opts == nullptr) {
		spasm_hip_echelonize_init_opts(&dflt);
		opts = &dflt;
	}
	struct echelonize_opts local = *opts;
	opts = &local;
	if (opts->complete)
		opts->L = 1;
	if (opts->L)
		opts->enable_tall_and_skinny = 0;      // as the reference: no L in the low-rank mode
	const struct spasm_csr *A = A0;
	int n = A->n;
	const int m = A->m;
	const i64 prime = A->field->p;
	logmsg("[echelonize] start on %d x %d matrix with %" PRId64 " nnz\n", n, m, A->p[A->n]);
	struct spasm_csr *U = spasm_hip_csr_alloc(n, m, A->p[A->n], prime, true);
	U->n = 0;
	int *Uqinv = (int *) xmalloc((i64) m * sizeof(int));
	for (int j = 0; j < m; j++)
		Uqinv[j] = -1;
	struct spasm_lu *fact = (struct spasm_lu *) xmalloc(sizeof(*fact));
	fact->L = nullptr;
	fact->p = nullptr;
	fact->U = U;
	fact->qinv = Uqinv;
	fact->Ltmp = nullptr;
	fact->complete = false;
	if (opts->L) {
		fact->Ltmp = spasm_hip_triplet_alloc(n, n, A->p[A->n] + 16, prime, true);
		fact->p = (int *) xmalloc((i64) (n > 0 ? n : 1) * sizeof(int));
		for (int j = 0; j < n; j++)
			fact->p[j] = -1;
	}

	int *p = (int *) xmalloc((i64) n * sizeof(int));
	int *p_in = nullptr;
	const double start = wtime();
	double density = (n > 0 && m > 0) ? (double) A->p[A->n] / n / m : 0.0;
	int npiv = 0, status = 0, round;
	// status 0: round limit reached / 1: nothing left / 2: pivots found, Schur complement not computed
	// 3: a Schur complement was computed and the census says no further round (its entries may still be on the device only)
	for (round = 0; round < opts->max_round; round++) {
		if (A->p[A->n] == 0) {
			status = 1;
			break;
		}
		logmsg("[echelonize] round %d\n", round);
		{
			Stopwatch sw(1);
			npiv = spasm_hip_pivots_extract_structural(A, p_in, fact, p, opts);
		}
		if (npiv < opts->min_pivot_proportion * std::min(n, m - U->n)) {
			logmsg("[echelonize] not enough pivots found; stopping\n");
			status = 2;
			break;
		}
		{
			Stopwatch sw(2);
			density = spasm_hip_schur_estimate_density(A, p + npiv, n - npiv, U, Uqinv, 100);
		}
		if (density > opts->sparsity_threshold) {
			logmsg("[echelonize] Schur complement is dense (estimated %.2f%%)\n", 100 * density);
			status = 2;
			break;
		}
		logmsg("[echelonize] Schur complement is %d x %d, estimated density : %.4f\n", n - npiv, m - U->n, density);
		int *p_out = (int *) xmalloc((i64) (n - npiv) * sizeof(int));
		struct spasm_csr *S;
		{
			Stopwatch sw(3);
			// the entries of S stay on the device until somebody needs them on the host (see below)
			resident_lazy_downloads(round + 1 < opts->max_round && (1) != 0);
			S = spasm_hip_schur(A, p + npiv, n - npiv, fact, density, fact->Ltmp, p_in, p_out);
			resident_lazy_downloads(false);
		}
		g_prof[5] += 1.0;
		if (A != A0) {
			resident_forget(A);
			spasm_hip_csr_free((struct spasm_csr *) A);
		}
		A = S;
		n = n - npiv;
		std::free(p_in);
		p_in = p_out;
		// Is another structural round worth bringing S to the host for (8 bytes per entry over PCIe, then a search whose
		// graph walks grow with the density of S)?  The first step of the search -- one pivot per distinct leftmost column,
		// spasm_pivots.c:35-80 -- is counted on the device, where S already is: in every search logged so far the later
		// steps (columns, greedy) added less than that again, so when even EIGHT times the count would fall short of
		// min_pivot_proportion the round would end in "not enough pivots" and is skipped (mk14.b4: 89 leftmost pivots on a
		// 673,000 x 42,000 complement with 1.06e9 entries; the search found 299 in 1.4 s after a 0.7 s download and stopped).
		// The remainder goes to the finishing code as it would have, with no pivots of its own: same row space.
		const int census = (round + 1 < opts->max_round) ? resident_fl_census(A) : -1;
		// (the factor 8 is the largest ratio (pivots of a whole search) / (its first step) seen on the generated families, where it is
		//  1.1-1.5; a matrix whose greedy step finds ten times what its leftmost entries give would be cut short here:
		//  SPASM_HIP_CENSUS_FACTOR raises it, 0 switches the short cut off -- the round then runs and decides by itself)
		const double census_factor = (double) (8);
		if (census >= 0 && census_factor > 0 && census_factor * census < opts->min_pivot_proportion * std::min(n, m - U->n)) {
			logmsg("[echelonize] %d leftmost-entry pivots in the Schur complement (counted on the device): not enough for another round\n", census);
			npiv = 0;
			for (int i = 0; i < n; i++)
				p[i] = i;
			status = 3;
			break;
		}
		{
			Stopwatch sw(3);
			resident_materialize(A);
		}
	}
	if (status == 0) {
		npiv = 0;
		for (int i = 0; i < n; i++)
			p[i] = i;
	}
	if (status != 1) {
		const double aspect = (m - U->n > 0) ? (double) (n - npiv) / (m - U->n) : 0.0;
		logmsg("[echelonize] finishing; density = %.3f; aspect ratio = %.1f\n", density, aspect);
		// (status 3: the entries of A may be on the device only; the device finish never reads them on the host, everything
		// else gets them first)
		// (a remainder that the device finish cannot hold -- its stack of echelon rows is rows x non-pivotal columns words -- goes
		//  through the host loops of the reference: minutes to hours on a remainder of a million columns.  That is what the WIDE
		//  orientation of a matrix leads to (the transpose of mk15.b4: 730 s where the tall orientation takes 0.3 s); tools/rank
		//  transposes such a matrix first, a caller of this function is told)
		auto leaving_the_device = [&]() {
			const int cols_left = m - U->n;
			if (cols_left > 1000000 || (A0->m > 2 * (i64) A0->n && cols_left > 200000))
				std::fprintf(stderr, "[spasm-hip] warning: the dense finish of this %d x %d matrix (%d columns left) does not fit the device and runs in the host loops: "
				                     "expect minutes to hours.%s\n", A0->n, A0->m, cols_left,
				             A0->m > A0->n ? "  The matrix has more columns than rows: echelonize its transpose (same rank) as tools/rank does." : "");
		};
		if (opts->enable_tall_and_skinny && aspect > opts->tall_and_skinny_ratio) {
			Stopwatch sw(4);
			if (!finish_on_device(A, p + npiv, n - npiv, fact, opts, true)) {
				leaving_the_device();
				resident_materialize(A);
				finish_lowrank(A, p + npiv, n - npiv, fact, opts);
			}
		} else if (opts->enable_dense && density > opts->sparsity_threshold) {
			Stopwatch sw(4);
			if (!finish_on_device(A, p + npiv, n - npiv, fact, opts, false)) {
				leaving_the_device();
				resident_materialize(A);
				finish_dense(A, p + npiv, n - npiv, p_in, fact, opts);
			}
		} else if (opts->enable_GPLU) {
			Stopwatch sw(6);
			resident_materialize(A);
			// The reference reduces the remaining rows one by one (GPLU, a sequential loop).  Here the
			// remainder keeps going through structural rounds on the GPU: each one finds at least one
			// pivot while the remainder is non-zero, so this terminates with the same row space.
			// A remainder that turns dense is handed to the dense code.
			int extra = 0;
			for (;;) {
				if (status == 2 && extra == 0) {
					// pivots of the last search are already in U: eliminate them first
				} else {
					if (A->p[A->n] == 0)
						break;
					npiv = spasm_hip_pivots_extract_structural(A, p_in, fact, p, opts);
					if (npiv == 0)
						die("structural pivot search found nothing on a non-zero matrix");
					// a round is only guaranteed one pivot, and every round re-plans the whole factor: once the search
					// brings in less than 1 % of what is left, the remainder goes to the dense code whatever its density
					// (the reference's one-pass GPLU would take it here)
					if (opts->enable_dense && n - npiv > 0 && (double) npiv < 0.01 * (double) std::min(n, m - (U->n - npiv))) {
						logmsg("[echelonize/rounds] only %d pivots in extra round %d: dense finish for the remaining %d rows\n", npiv, extra, n - npiv);
						if (!finish_on_device(A, p + npiv, n - npiv, fact, opts, false))
							finish_dense(A, p + npiv, n - npiv, p_in, fact, opts);
						break;
					}
				}
				if (n - npiv == 0)
					break;
				density = spasm_hip_schur_estimate_density(A, p + npiv, n - npiv, U, Uqinv, 100);
				if (opts->enable_dense && density > opts->sparsity_threshold) {
					if (!finish_on_device(A, p + npiv, n - npiv, fact, opts, false))
						finish_dense(A, p + npiv, n - npiv, p_in, fact, opts);
					break;
				}
				int *p_out = (int *) xmalloc((i64) (n - npiv) * sizeof(int));
				struct spasm_csr *S = spasm_hip_schur(A, p + npiv, n - npiv, fact, density, fact->Ltmp, p_in, p_out);
				if (A != A0) {
					resident_forget(A);
					spasm_hip_csr_free((struct spasm_csr *) A);
				}
				A = S;
				n = n - npiv;
				std::free(p_in);
				p_in = p_out;
				extra += 1;
				logmsg("[echelonize/rounds] extra round %d: rank >= %d, %d rows left\n", extra, U->n, n);
			}
		} else {
			logmsg("[echelonize] cannot finish (no valid method enabled); incomplete echelonization returned\n");
		}
	}
	std::free(p);
	std::free(p_in);
	{
		int64_t uploads = 0, hits = 0;
		resident_counters(&uploads, &hits);
		g_prof[7] = (double) uploads;
		logmsg("[echelonize] matrices uploaded to the device so far in this process: %lld (reused in place %lld times)\n",
		       (long long) uploads, (long long) hits);
	}
	resident_end();
	g_prof[0] = wtime() - start;
	logmsg("[echelonize] done in %.1fs. Rank %d, %" PRId64 " nz in basis\n", wtime() - start, U->n, U->p[U->n]);
	spasm_hip_csr_resize(U, U->n, m);
	spasm_hip_csr_realloc(U, -1);
	if (A != A0)
		spasm_hip_csr_free((struct spasm_csr *) A);
	if (opts->L) {
		fact->Ltmp->n = A0->n;
		fact->Ltmp->m = U->n;
		fact->p = (int *) xrealloc(fact->p, (i64) (U->n > 0 ? U->n : 1) * sizeof(int));
		fact->L = spasm_hip_compress(fact->Ltmp);
		spasm_hip_triplet_free(fact->Ltmp);
		fact->Ltmp = nullptr;
		fact->complete = opts->complete;
	}
	fact->r = U->n;
	return fact;
}

// replaces spasm_rref (spasm_rref.c:25-146).  Row i of R is U[i] reduced by every other pivotal
// row.  U[i] minus its pivot entry only meets pivots of later rows, whose rows never reach back
// to column pivot(i), so reducing it against the whole of U gives the same row: one GPU call.
extern "C" struct spasm_csr *spasm_hip_rref(const struct spasm_lu *fact, int *Rqinv)
{
	const struct spasm_csr *U = fact->U;
	const int n = U->n, m = U->m;
	const i64 prime = U->field->p;
	struct spasm_csr *T = spasm_hip_csr_alloc(n, m, U->p[n], prime, true);
	i64 w = 0;
	std::vector<int> rows((size_t) (n > 0 ? n : 1));
	for (int i = 0; i < n; i++) {
		rows[i] = i;
		for (i64 px = U->p[i] + 1; px < U->p[i + 1]; px++) {
			T->j[w] = U->j[px];
			T->x[w] = U->x[px];
			w += 1;
		}
		T->p[i + 1] = w;
	}
	struct spasm_csr *S = spasm_hip_schur(T, rows.data(), n, fact, 0.0, nullptr, nullptr, nullptr);
	spasm_hip_csr_free(T);
	struct spasm_csr *R = spasm_hip_csr_alloc(n, m, S->p[n] + n, prime, true);
	w = 0;
	for (int j = 0; j < m; j++)
		Rqinv[j] = -1;
	for (int i = 0; i < n; i++) {
		const int piv = U->j[U->p[i]];
		R->j[w] = piv;
		R->x[w] = 1;
		w += 1;
		Rqinv[piv] = i;
		for (i64 px = S->p[i]; px < S->p[i + 1]; px++) {
			R->j[w] = S->j[px];
			R->x[w] = S->x[px];
			w += 1;
		}
		R->p[i + 1] = w;
	}
	spasm_hip_csr_free(S);
	return R;
}

// replaces spasm_kernel (spasm_kernel.c:9-140) through the RREF (spasm_kernel_from_rref, :146-178):
// for every non-pivotal column j the row  -e_j + sum_i R[i, j] e_{pivot(i)}.
extern "C" struct spasm_csr *spasm_hip_kernel(const struct spasm_lu *fact)
{
	const int m = fact->U->m;
	const i64 prime = fact->U->field->p;
	std::vector<int> Rqinv((size_t) (m > 0 ? m : 1));
	struct spasm_csr *R = spasm_hip_rref(fact, Rqinv.data());
	struct spasm_csr *Rt = spasm_hip_transpose(R, 1);
	const int n = R->n;
	struct spasm_csr *K = spasm_hip_csr_alloc(m - n, m, R->p[n] - n + (m - n), prime, true);
	K->n = 0;
	i64 w = 0;
	for (int j = 0; j < m; j++) {
		if (Rqinv[j] >= 0)
			continue;
		K->j[w] = j;
		K->x[w] = zp_init(prime, prime - 1);
		w += 1;
		for (i64 px = Rt->p[j]; px < Rt->p[j + 1]; px++) {
			const int i = Rt->j[px];
			K->j[w] = R->j[R->p[i]];
			K->x[w] = Rt->x[px];
			w += 1;
		}
		K->n += 1;
		K->p[K->n] = w;
	}
	spasm_hip_csr_free(Rt);
	spasm_hip_csr_free(R);
	return K;
}
