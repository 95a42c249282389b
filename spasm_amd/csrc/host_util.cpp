// Host-side containers, field arithmetic, triplet->CSR, transpose.
// Mirrors the reference's C API for these (spasm_util.c, spasm_ZZp.c,
// spasm_triplet.c, spasm_transpose.c) behind spasm_hip_* names; every array is
// malloc-owned so the reference's own free functions can release them.
#include <sys/time.h>

#include <cstring>
#include <vector>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "common.h"

namespace sh {

double wtime()
{
	struct timeval tv;
	gettimeofday(&tv, nullptr);
	return (double) tv.tv_sec + 1e-6 * (double) tv.tv_usec;
}

long long *counters()
{
	static long long c[CNT_COUNT];
	return c;
}

// ---- environment switches --------------------------------------------------------------------
// ---- a small pool of worker threads for the host planning loops ---------------------------------------------------
// The plans of a factor image are a dozen loops of a few hundred microseconds each over 10^5 .. 10^6 rows; a std::thread per
// part per loop cost more than the loops (50-100 us to create and join a thread on the boxes of this pool: eight threads, two
// passes = 1.4 ms around 0.6 ms of work).  The workers are created once, sleep on a condition variable between jobs and are
// never joined (the pool is leaked on purpose: no destructor runs at exit while a worker waits).  One job at a time; a job
// started from inside a job (or while another thread's job runs) is run by its caller alone.
namespace {
struct WorkerPool {
	std::mutex m;
	std::condition_variable wake, done;
	std::function<void(int)> job;
	int ntasks = 0, next = 0, running = 0;
	uint64_t generation = 0;
	int nworkers = 0;
	std::mutex one_job;

	void worker()
	{
		uint64_t seen = 0;
		std::unique_lock<std::mutex> lk(m);
		for (;;) {
			wake.wait(lk, [&] { return generation != seen; });
			seen = generation;
			while (next < ntasks) {
				const int t = next++;
				running += 1;
				lk.unlock();
				job(t);
				lk.lock();
				running -= 1;
			}
			if (running == 0)
				done.notify_all();
		}
	}
};
WorkerPool *g_pool = nullptr;
thread_local bool t_in_pool_job = false;
}  // namespace

int usable_cpus();

void pool_run(int ntasks, const std::function<void(int)> &fn)
{
	if (ntasks <= 0)
		return;
	static std::once_flag once;
	std::call_once(once, [] {
		g_pool = new WorkerPool();
		g_pool->nworkers = std::max(0, std::min(15, usable_cpus() - 1));
		for (int t = 0; t < g_pool->nworkers; t++)
			std::thread([] { t_in_pool_job = true; g_pool->worker(); }).detach();
	});
	WorkerPool &P = *g_pool;
	if (ntasks == 1 || P.nworkers == 0 || t_in_pool_job || !P.one_job.try_lock()) {
		for (int t = 0; t < ntasks; t++)
			fn(t);
		return;
	}
	std::unique_lock<std::mutex> lk(P.m);
	P.job = fn;
	P.ntasks = ntasks;
	P.next = 0;
	P.generation += 1;
	P.wake.notify_all();
	t_in_pool_job = true;
	while (P.next < P.ntasks) {          // the caller works too
		const int t = P.next++;
		P.running += 1;
		lk.unlock();
		fn(t);
		lk.lock();
		P.running -= 1;
	}
	t_in_pool_job = false;
	P.done.wait(lk, [&] { return P.running == 0 && P.next >= P.ntasks; });
	P.job = nullptr;
	lk.unlock();
	P.one_job.unlock();
}

const char *env_get(const char *name)
{
	// (the list of include/spasm_hip.h, section "Environment")
	static const char *const supported[] = {"SPASM_HIP_VERBOSE", "SPASM_HIP_THREADS", "SPASM_HIP_PIVOT_SEARCH", "SPASM_HIP_PIVOT_LABELS", "SPASM_HIP_BACKSOLVE",
	                                        "SPASM_HIP_SPARSE_IMAGE", "SPASM_HIP_DEVICE_FINISH", "SPASM_HIP_KEEP_GB", "SPASM_HIP_SCRATCH_GB", "SPASM_HIP_STAGE_GB",
	                                        "SPASM_HIP_SPARSE_IMAGE_GB", "SPASM_HIP_EXPERIMENT"};
	const char *v = std::getenv(name);
	if (v == nullptr)
		return nullptr;
	for (const char *s : supported)
		if (std::strcmp(s, name) == 0)
			return v;
	// (read at every look-up: a caller that sets SPASM_HIP_EXPERIMENT around one call -- tools/workloads.py does -- must neither
	//  find it ignored because an earlier look-up cached "off", nor leave it on for the rest of the process)
	const char *e = std::getenv("SPASM_HIP_EXPERIMENT");
	const bool experiment = e != nullptr && *e != 0 && std::strcmp(e, "0") != 0;
	if (!experiment) {
		static bool warned = false;
		if (!warned) {
			warned = true;
			std::fprintf(stderr, "[spasm-hip] %s is set but SPASM_HIP_EXPERIMENT is not: switches outside the supported list (include/spasm_hip.h) are ignored\n", name);
		}
		return nullptr;
	}
	return v;
}

int verbose()
{
	static int v = -1;
	if (v < 0) {
		const char *e = env_get("SPASM_HIP_VERBOSE");
		v = (e == nullptr) ? 1 : std::atoi(e);
	}
	return v;
}

void logmsg(const char *fmt, ...)
{
	if (!verbose())
		return;
	va_list ap;
	va_start(ap, fmt);
	std::vfprintf(stderr, fmt, ap);
	va_end(ap);
	std::fflush(stderr);
}

spasm_ZZp zp_inverse(int64_t p, spasm_ZZp a)
{
	int64_t v = a < 0 ? (int64_t) a + p : (int64_t) a;
	int64_t r0 = p, r1 = v, t0 = 0, t1 = 1;
	while (r1 != 0) {
		int64_t q = r0 / r1;
		int64_t r2 = r0 - q * r1, t2 = t0 - q * t1;
		r0 = r1; r1 = r2;
		t0 = t1; t1 = t2;
	}
	return zp_balance(p, t0);
}

Mont mont_setup(int64_t prime)
{
	if (prime < 3 || prime > 0xfffffffbLL || (prime & 1) == 0)
		die("modulus %lld unsupported on the GPU path (need an odd prime below 2^32)", (long long) prime);
	Mont M;
	M.p = (uint32_t) prime;
	uint32_t inv = M.p;                 // Newton iteration: inv * p == 1 mod 2^32
	for (int it = 0; it < 5; it++)
		inv *= 2u - M.p * inv;
	M.pinv = inv;
	M.r1 = (uint32_t) ((1ULL << 32) % (uint64_t) prime);
	M.r2 = (uint32_t) (((uint64_t) M.r1 * M.r1) % (uint64_t) prime);
	M.half = M.p / 2;
	return M;
}

}  // namespace sh

using namespace sh;

extern "C" {

const char *spasm_hip_version(void) { return "spasm-hip 0.3 (gfx950)"; }

// spasm_malloc / spasm_calloc / spasm_realloc (spasm_util.c:65-83): die instead of returning NULL
void *spasm_hip_malloc(i64 size) { return xmalloc(size); }
void *spasm_hip_calloc(i64 count, i64 size)
{
	void *q = std::calloc(count > 0 ? (size_t) count : 1, size > 0 ? (size_t) size : 1);
	if (q == nullptr)
		die("calloc failed (%lld x %lld bytes)", (long long) count, (long long) size);
	return q;
}
void *spasm_hip_realloc(void *ptr, i64 size) { return xrealloc(ptr, size); }
i64 spasm_hip_nnz(const struct spasm_csr *A) { return A->p[A->n]; }          // spasm_util.c:16
double spasm_hip_wtime(void) { return wtime(); }                             // spasm_util.c:9

void spasm_hip_field_init(i64 p, spasm_field F)
{
	F->p = p;
	if (p < 0)
		return;
	F->halfp = p / 2;
	F->mhalfp = p / 2 - p + 1;
	F->dinvp = 1.0 / (double) p;
}

spasm_ZZp spasm_hip_ZZp_init(const spasm_field F, i64 x) { return zp_init(F->p, x); }
spasm_ZZp spasm_hip_ZZp_add(const spasm_field F, spasm_ZZp a, spasm_ZZp b) { return zp_balance(F->p, (i64) a + b); }
spasm_ZZp spasm_hip_ZZp_sub(const spasm_field F, spasm_ZZp a, spasm_ZZp b) { return zp_balance(F->p, (i64) a - b); }
spasm_ZZp spasm_hip_ZZp_mul(const spasm_field F, spasm_ZZp a, spasm_ZZp b) { return zp_mul(F->p, a, b); }
spasm_ZZp spasm_hip_ZZp_inverse(const spasm_field F, spasm_ZZp a) { return zp_inverse(F->p, a); }
spasm_ZZp spasm_hip_ZZp_axpy(const spasm_field F, spasm_ZZp a, spasm_ZZp x, spasm_ZZp y)
{
	return zp_axpy(F->p, a, x, y);
}

struct spasm_csr *spasm_hip_csr_alloc(int n, int m, i64 nzmax, i64 prime, bool with_values)
{
	struct spasm_csr *A = (struct spasm_csr *) xmalloc(sizeof(*A));
	spasm_hip_field_init(prime, A->field);
	A->n = n;
	A->m = m;
	A->nzmax = nzmax;
	A->p = (i64 *) xmalloc(((i64) n + 1) * sizeof(i64));
	A->j = (int *) xmalloc(nzmax * sizeof(int));
	A->x = with_values ? (spasm_ZZp *) xmalloc(nzmax * sizeof(spasm_ZZp)) : nullptr;
	A->p[0] = 0;
	return A;
}

void spasm_hip_csr_realloc(struct spasm_csr *A, i64 nzmax)
{
	if (nzmax < 0)
		nzmax = A->p[A->n];
	A->j = (int *) xrealloc(A->j, nzmax * sizeof(int));
	if (A->x != nullptr)
		A->x = (spasm_ZZp *) xrealloc(A->x, nzmax * sizeof(spasm_ZZp));
	A->nzmax = nzmax;
}

void spasm_hip_csr_resize(struct spasm_csr *A, int n, int m)
{
	A->m = m;
	A->p = (i64 *) xrealloc(A->p, ((i64) n + 1) * sizeof(i64));
	if (A->n < n)
		for (int i = A->n; i < n + 1; i++)
			A->p[i] = A->p[A->n];
	A->n = n;
}

void spasm_hip_csr_free(struct spasm_csr *A)
{
	if (A == nullptr)
		return;
	std::free(A->p);
	std::free(A->j);
	std::free(A->x);
	std::free(A);
}

struct spasm_triplet *spasm_hip_triplet_alloc(int n, int m, i64 nzmax, i64 prime, bool with_values)
{
	struct spasm_triplet *T = (struct spasm_triplet *) xmalloc(sizeof(*T));
	spasm_hip_field_init(prime, T->field);
	T->n = n;
	T->m = m;
	T->nzmax = nzmax;
	T->nz = 0;
	T->i = (int *) xmalloc(nzmax * sizeof(int));
	T->j = (int *) xmalloc(nzmax * sizeof(int));
	T->x = with_values ? (spasm_ZZp *) xmalloc(nzmax * sizeof(spasm_ZZp)) : nullptr;
	return T;
}

void spasm_hip_triplet_realloc(struct spasm_triplet *T, i64 nzmax)
{
	if (nzmax < 0)
		nzmax = T->nz;
	T->i = (int *) xrealloc(T->i, nzmax * sizeof(int));
	T->j = (int *) xrealloc(T->j, nzmax * sizeof(int));
	if (T->x != nullptr)
		T->x = (spasm_ZZp *) xrealloc(T->x, nzmax * sizeof(spasm_ZZp));
	T->nzmax = nzmax;
}

void spasm_hip_triplet_free(struct spasm_triplet *T)
{
	if (T == nullptr)
		return;
	std::free(T->i);
	std::free(T->j);
	std::free(T->x);
	std::free(T);
}

void spasm_hip_lu_free(struct spasm_lu *N)
{
	if (N == nullptr)
		return;
	std::free(N->qinv);
	std::free(N->p);
	spasm_hip_csr_free(N->U);
	spasm_hip_csr_free(N->L);
	std::free(N);
}

void spasm_hip_add_entry(struct spasm_triplet *T, int i, int j, i64 x)
{
	if (i < 0 || j < 0)
		die("spasm_hip_add_entry: negative index (%d, %d)", i, j);
	if (T->nz == T->nzmax)
		spasm_hip_triplet_realloc(T, 1 + 2 * T->nzmax);
	if (T->x != nullptr) {
		spasm_ZZp v = zp_init(T->field->p, x);
		if (v == 0)
			return;
		T->x[T->nz] = v;
	}
	T->i[T->nz] = i;
	T->j[T->nz] = j;
	T->nz += 1;
	if (i + 1 > T->n) T->n = i + 1;
	if (j + 1 > T->m) T->m = j + 1;
}

void spasm_hip_triplet_transpose(struct spasm_triplet *T)
{
	int *t = T->i; T->i = T->j; T->j = t;
	int d = T->n; T->n = T->m; T->m = d;
}

// Same observable result as the reference: entries keep file order inside a
// row, a repeated (i, j) is summed into its first occurrence, zeros vanish.
struct spasm_csr *spasm_hip_compress(const struct spasm_triplet *T)
{
	const int n = T->n, m = T->m;
	const i64 nz = T->nz;
	const i64 prime = T->field->p;
	const bool vals = (T->x != nullptr);
	struct spasm_csr *C = spasm_hip_csr_alloc(n, m, nz, prime, vals);
	std::vector<i64> fill((size_t) n + 1, 0);
	for (i64 k = 0; k < nz; k++) {
		if (T->i[k] >= n || T->j[k] >= m)
			die("spasm_hip_compress: entry (%d, %d) outside %d x %d", T->i[k], T->j[k], n, m);
		fill[T->i[k] + 1] += 1;
	}
	for (int i = 0; i < n; i++)
		fill[i + 1] += fill[i];
	std::vector<i64> start(fill.begin(), fill.end());
	std::vector<int> bj((size_t) (nz > 0 ? nz : 1));
	std::vector<spasm_ZZp> bx((size_t) (vals && nz > 0 ? nz : 1));
	for (i64 k = 0; k < nz; k++) {
		i64 dst = fill[T->i[k]]++;
		bj[dst] = T->j[k];
		if (vals)
			bx[dst] = T->x[k];
	}
	std::vector<i64> slot((size_t) (m > 0 ? m : 1), -1);
	i64 out = 0;
	C->p[0] = 0;
	for (int i = 0; i < n; i++) {
		const i64 row0 = out;
		for (i64 px = start[i]; px < start[i + 1]; px++) {
			const int j = bj[px];
			if (slot[j] < row0) {
				slot[j] = out;
				C->j[out] = j;
				if (vals)
					C->x[out] = bx[px];
				out += 1;
			} else if (vals) {
				C->x[slot[j]] = zp_balance(prime, (i64) C->x[slot[j]] + bx[px]);
			}
		}
		if (vals) {                 // squeeze out entries that cancelled
			i64 w = row0;
			for (i64 px = row0; px < out; px++) {
				if (C->x[px] == 0)
					continue;
				C->j[w] = C->j[px];
				C->x[w] = C->x[px];
				w += 1;
			}
			out = w;
		}
		for (i64 px = start[i]; px < start[i + 1]; px++)
			slot[bj[px]] = -1;          // positions moved: forget this row's columns
		C->p[i + 1] = out;
	}
	spasm_hip_csr_realloc(C, -1);
	return C;
}

struct spasm_csr *spasm_hip_transpose(const struct spasm_csr *A, int keep_values)
{
	const int n = A->n, m = A->m;
	const i64 nnz = A->p[n];
	const bool vals = keep_values && (A->x != nullptr);
	struct spasm_csr *T = spasm_hip_csr_alloc(m, n, nnz, A->field->p, vals);
	std::vector<i64> w((size_t) m + 1, 0);
	for (i64 px = 0; px < nnz; px++)
		w[A->j[px] + 1] += 1;
	for (int j = 0; j < m; j++)
		w[j + 1] += w[j];
	for (int j = 0; j <= m; j++)
		T->p[j] = w[j];
	for (int i = 0; i < n; i++)
		for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
			i64 dst = w[A->j[px]]++;
			T->j[dst] = i;
			if (vals)
				T->x[dst] = A->x[px];
		}
	return T;
}

}  // extern "C"
