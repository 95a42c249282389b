// Should this batch go through the SPARSE image (sparse_image.hip)?  It is made for Schur complements that stay sparse on
// many columns: R is then mostly zeros, and both the build and the rows of S cost what R and S hold, not r x Sm.
// SPASM_HIP_SPARSE_IMAGE=0 never, =1 whenever the factor has the plan (tests); else: full batches (>= 1,024 rows) whose
// expected density is under 12 % (the driver's estimate; unknown: 3 % assumed beyond 16,384 non-pivotal columns), when
//   build (if R is not there)   8 us per elimination level (the chain of hand-overs: measured 8.0 on mk14.b4, 2,000 levels)
//   rows of S + gather          27 ps per expected entry of S + 3 ns per row + 0.5 ms      (mk14.b4: 6.4e8 entries in 20 ms)
// is less than what the cost model of backsolve_wanted gives the dense image and the row-by-row kernels.
// Launchers and C ABI of the sparse Schur complement (see include/spasm_hip.h).
#include <algorithm>
#include <unordered_map>
#include <atomic>
#include <thread>
#include <cinttypes>
#include <mutex>
#include <vector>

#include "device_types.h"

namespace sh {
void launch_schur_lds(const SchurArgs &a, int table, bool wide, int blocks, hipStream_t stream);
size_t schur_lds_bytes(int table, bool wide);
void launch_finalize(const spasm_hip_dwork *W, int nrows, int sort_rows, hipStream_t stream);
void wave_dense_geometry(int rpad, int Sm, bool wide, int64_t *slot_bytes, int64_t *off_bm, int64_t *off_xn);
void group_geometry(int rpad, int Sm, bool wide, int64_t *slot_bytes, int64_t *off_bm);
int64_t regroup_scratch_ints(int nrows, int r);
void launch_regroup_rows(const SchurArgs &a, int *sortbuf, int *order, hipStream_t stream);
void launch_schur_group(const SchurArgs &a, unsigned char *scratch, int64_t slot_bytes, int64_t off_bm, bool wide,
                        uint32_t *dense_out, int64_t ldS, int blocks, hipStream_t stream, int watch, float min_eff,
                        long long min_w, int waves);
void launch_schur_wave_dense(const SchurArgs &a, unsigned char *scratch, int64_t slot_bytes, int64_t off_bm, int64_t off_xn,
                             bool wide, uint32_t *dense_out, int64_t ldS, int blocks, hipStream_t stream);
void launch_all_rows_to_list(int *list, int *count, int *row_len, int nrows, hipStream_t stream);
void schur_group_variant_name(int r, bool wide, int waves, char *out, size_t cap);
size_t pull_lds_bytes(int rpad, int Sm);
int64_t pull_slot_bytes(int rpad, int Sm);
void launch_schur_pull(const SchurArgs &a, unsigned char *scratch, int64_t slot_bytes, const uint64_t *cp, const uint2 *cent,
                       const int2 *lvl, int nlev, int blocks, hipStream_t stream);
bool backsolve_eligible(int r, int Sm, int64_t nnz_u, int64_t *bytes, int64_t prime);
void backsolve_plan(const FactPlan &P, spasm_hip_dfact *F, hipStream_t stream);
void backsolve_free(spasm_hip_dfact *F);
void backsolve_build(const spasm_hip_dfact *F, hipStream_t stream);
bool backsolve_stages_output(const spasm_hip_dfact *F, int64_t *row_bytes);
void launch_backsolve_apply(const SchurArgs &a, const spasm_hip_dfact *F, uint32_t *dense_out, int64_t ldS, hipStream_t stream,
                            BsDirectOut *direct);
bool backsolve_wanted(const spasm_hip_dfact *F, bool other_path_forced, int nrows);
bool sparse_image_possible(int64_t prime);
void sparse_image_plan(const FactPlan &P, spasm_hip_dfact *F, hipStream_t stream);
void sparse_image_plan_start(const FactPlan &P, spasm_hip_dfact *F);
bool sparse_image_planned(const spasm_hip_dfact *F, hipStream_t stream);
bool sparse_image_plan_expected(const spasm_hip_dfact *F);
void sparse_image_free(spasm_hip_dfact *F);
bool sparse_image_build(const spasm_hip_dfact *F, hipStream_t stream);
int64_t sparse_image_table_words(int64_t nrows, int nseg);
void launch_sparse_image_apply(const SchurArgs &a, const spasm_hip_dfact *F, uint32_t *fpool, uint32_t *fpool_v, int64_t fcap,
                               uint64_t *T, unsigned long long *block_sum, int64_t *Sp, int *Sj, int *Sx, int64_t cap, hipStream_t stream,
                               hipEvent_t ev_gather);
bool sparse_image_wanted(const spasm_hip_dfact *F, bool other_path_forced, int nrows);
void sparse_image_census(const spasm_hip_dfact *F, int64_t *out, hipStream_t stream);
int usable_cpus();          // host_pivots.cpp
// multi-GPU layer (dist_api.hip)
spasm_hip_comm *current_comm();
int comm_rank(const spasm_hip_comm *c);
int comm_world(const spasm_hip_comm *c);
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

template <typename T> void upload(T *dst, const T *src, int64_t count, hipStream_t s)
{
	if (count > 0)
		sh::h2d(dst, src, (size_t) count * sizeof(T), s);
}

}  // namespace

namespace sh {
namespace {
struct BigPool {
	std::mutex mutex;
	std::vector<std::pair<void *, size_t>> free_blocks;          // cached, not in use
	std::vector<std::pair<void *, size_t>> live;                 // handed out by big_alloc
	size_t cached = 0;
	std::unordered_map<void *, int> idle;                        // driver calls a cached block has sat through unused (big_age)
	// small blocks (< BIG_MIN), rounded up to a power of two: free lists per size class, and who is out
	std::vector<void *> small_free[32];
	std::unordered_map<void *, int> small_live;
	size_t small_cached = 0;
};
BigPool g_big;
constexpr size_t BIG_MIN = (size_t) 32 << 20;                    // from here on: whole blocks, best fit
constexpr size_t BIG_CACHE_CAP = (size_t) 96 << 30;
constexpr size_t SMALL_CACHE_CAP = (size_t) 8 << 30;
inline int small_class(size_t bytes)
{
	int c = 8;                          // 256 bytes
	while (((size_t) 1 << c) < bytes)
		c += 1;
	return c;
}
}  // namespace

// Device buffers of every size come from here.  hipMalloc itself is cheap on these boxes (3-20 us: it maps nothing), but a
// FRESH block is paid for on first touch -- erratically, up to seconds for tens of GB -- and hipFree takes 70-170 us for
// anything from 1 MB up (it waits for the device): a call of the driver makes a few hundred small buffers.  So freed
// blocks are parked and handed out again: small ones (< 32 MB) by size class, large ones by best fit.
static std::atomic<long long> g_device_mallocs{0};          // requests that the cache could not serve (tests)

void *big_alloc(size_t bytes)
{
	if (bytes == 0)
		bytes = 1;
	void *ptr = nullptr;
	const bool big = bytes >= BIG_MIN;          // (decided BEFORE the rounding: a small request of 17-32 MB is rounded to BIG_MIN itself, was
	                                            //  then booked as a big block, parked as one when freed, and never found again by the next
	                                            //  small request of its class: five hipMallocs of 32 MB in every driver call on mk15.b4)
	if (big) {
		// Sizes in steps of an eighth of their leading power of two, and any parked block of up to twice the size will do: the
		// large buffers of a driver call -- row pools, the Schur complement, accumulators, echelon stacks -- follow the pivot set of
		// the call, which depends on timing, and differ by tens of percent from one call to the next.  With exact sizes and a
		// window of 1.5x the third mk15.b4 call of a process still fetched 62 GB from the device (3.5 s inside one sparse round;
		// spasm_hip_echelonize_counters: block_cache_miss_bytes), the fourth 21 GB.
		static const bool classes = (1) != 0;
		size_t step = (size_t) 1 << 20;
		while ((step << 4) <= bytes)
			step <<= 1;
		if (classes)
			bytes = (bytes + step - 1) / step * step;
		std::lock_guard<std::mutex> guard(g_big.mutex);
		int best = -1;
		for (size_t t = 0; t < g_big.free_blocks.size(); t++) {
			const size_t have = g_big.free_blocks[t].second;
			// (from 256 MB on a parked block of up to FOUR times the size would do -- see below --: what the device charges for is a block taken
			//  fresh -- its first touch, 0.2 s per GB when the driver has to clear new memory: a 2.5 s sparse round in one mk15.b4
			//  call of ten, whose pools of 2.7 GB found only the 6.7 GB blocks of a call with a larger Schur complement parked --, not
			//  the untouched tail of a block that is larger than asked)
			// (... and in the end ANY parked block that is large enough, the smallest one first: the bench's first mk15.b4 call after
			//  the phase with the 19 GB pools of the fixed pivot set found those parked, more than four times what it wanted, took 9 GB
			//  fresh right after 100 GB had gone back to the device, and spent 2.9 s on their first touch)
			// (round 6: bounded again, at four times the size or 8 GB more, whichever is larger -- a 300 MB buffer could pin a 19 GB block
			//  for its lifetime and send the next 19 GB request to the device for 28 GB fresh; the bench's case above, a 9 GB request
			//  that finds 19 GB parked, still fits)
			const size_t most = !classes ? bytes + bytes / 2 : (bytes >= ((size_t) 256 << 20) ? std::max(4 * bytes, bytes + ((size_t) 8 << 30)) : 2 * bytes);
			if (have >= bytes && have <= most && (best < 0 || have < g_big.free_blocks[(size_t) best].second))
				best = (int) t;
		}
		if (best >= 0) {
			const auto blk = g_big.free_blocks[(size_t) best];
			g_big.free_blocks.erase(g_big.free_blocks.begin() + best);
			g_big.cached -= blk.second;
			g_big.idle.erase(blk.first);
			g_big.live.push_back(blk);
			return blk.first;
		}
	} else {
		const int c = small_class(bytes);
		bytes = (size_t) 1 << c;
		std::lock_guard<std::mutex> guard(g_big.mutex);
		if (!g_big.small_free[c].empty()) {
			ptr = g_big.small_free[c].back();
			g_big.small_free[c].pop_back();
			g_big.small_cached -= bytes;
			g_big.small_live[ptr] = c;
			return ptr;
		}
	}
	// A large block that has to come from the device is taken half again as large as asked: the pools of a driver call follow its
	// pivot set, and the SECOND call of a process on mk15.b4 found its Schur complement 20-30 % larger than the first one's often
	// enough -- 6 blocks, 11.7 GB taken fresh, 0.36 s of first touches inside one sparse round (the later calls then fit).  The
	// part nobody asked for is never touched and costs nothing but address space of a 288 GB device.
	g_device_mallocs += 1;
	size_t take = bytes;
	if (big && bytes >= ((size_t) 256 << 20)) {
		// (the headroom follows what the device has left: half again when eight times the request is free, nothing when less than
		//  twice is -- on a device smaller than these 288 GB, or shared, the headroom must not be what runs it out of memory)
		size_t free_b = 0, total_b = 0;
		if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
			(void) hipGetLastError();
			free_b = 0;
		}
		const size_t pct = free_b >= 8 * bytes ? 50 : free_b >= 4 * bytes ? 25 : free_b >= 2 * bytes ? 10 : 0;
		take = bytes + bytes / 100 * pct;
	}
	if (take > bytes && hipMalloc(&ptr, take) != hipSuccess) {
		(void) hipGetLastError();
		ptr = nullptr;
		take = bytes;
	}
	if (ptr == nullptr && hipMalloc(&ptr, bytes) != hipSuccess) {
		(void) hipGetLastError();
		big_trim(0);                         // cached blocks may be what is in the way
		HIP_CHECK(hipMalloc(&ptr, bytes));
	}
	if (ptr != nullptr && take > bytes)
		bytes = take;
	std::lock_guard<std::mutex> guard(g_big.mutex);
	if (big) {
		counters()[CNT_BIG_ALLOC_MISSES] += 1;
		counters()[CNT_BIG_ALLOC_MISS_BYTES] += (long long) bytes;
		if (verbose() >= 3)
			logmsg("[block cache] no parked block for %.1f MB: taken from the device (%.1f MB parked in %zu blocks)\n", 1e-6 * (double) bytes, 1e-6 * (double) g_big.cached,
			       g_big.free_blocks.size());
	}
	if (big)
		g_big.live.push_back({ptr, bytes});
	else
		g_big.small_live[ptr] = small_class(bytes);
	return ptr;
}

// hipMalloc for the few buffers that do not go through the cache (they are freed with hipFree by their owners): a failure
// first returns what the cache parks, then tries again
hipError_t malloc_or_trim(void **ptr, size_t bytes)
{
	hipError_t e = hipMalloc(ptr, bytes);
	if (e != hipSuccess) {
		(void) hipGetLastError();
		big_trim(0);
		e = hipMalloc(ptr, bytes);
	}
	return e;
}

// free / total device memory as the decisions of this library should see them: what the block cache parks counts as free
// (it is given back on demand)
void mem_info(size_t *free_b, size_t *total_b)
{
	HIP_CHECK(hipMemGetInfo(free_b, total_b));
	std::lock_guard<std::mutex> guard(g_big.mutex);
	*free_b += g_big.cached + g_big.small_cached;
}

void big_free(void *ptr)
{
	if (ptr == nullptr)
		return;
	{
		std::lock_guard<std::mutex> guard(g_big.mutex);
		auto it = g_big.small_live.find(ptr);
		if (it != g_big.small_live.end()) {
			const int c = it->second;
			g_big.small_live.erase(it);
			if (g_big.small_cached + ((size_t) 1 << c) <= SMALL_CACHE_CAP) {
				g_big.small_free[c].push_back(ptr);
				g_big.small_cached += (size_t) 1 << c;
				return;
			}
		} else {
			for (size_t t = 0; t < g_big.live.size(); t++)
				if (g_big.live[t].first == ptr) {
					const auto blk = g_big.live[t];
					g_big.live.erase(g_big.live.begin() + (long) t);
					if (g_big.cached + blk.second <= BIG_CACHE_CAP) {
						g_big.free_blocks.push_back(blk);
						g_big.cached += blk.second;
						return;
					}
					break;
				}
		}
	}
	(void) hipFree(ptr);
}

void big_trim(size_t keep_bytes)
{
	std::vector<std::pair<void *, size_t>> blocks;
	{
		std::lock_guard<std::mutex> guard(g_big.mutex);
		if (keep_bytes == 0) {
			blocks.swap(g_big.free_blocks);
			g_big.cached = 0;
			for (int c = 0; c < 32; c++) {
				for (void *q : g_big.small_free[c])
					blocks.push_back({q, (size_t) 1 << c});
				g_big.small_free[c].clear();
			}
			g_big.small_cached = 0;
		} else {
			// the largest blocks stay (they are the ones that cost: up to 40 ms per GB to get back, seconds at times)
			std::sort(g_big.free_blocks.begin(), g_big.free_blocks.end(), [](const auto &x, const auto &y) { return x.second > y.second; });
			size_t kept = 0;
			std::vector<std::pair<void *, size_t>> stay;
			for (auto &b : g_big.free_blocks) {
				if (kept + b.second <= keep_bytes) {
					kept += b.second;
					stay.push_back(b);
				} else {
					blocks.push_back(b);
				}
			}
			g_big.free_blocks.swap(stay);
			g_big.cached = kept;
		}
	}
	{
		std::lock_guard<std::mutex> guard(g_big.mutex);
		for (auto &b : blocks)
			g_big.idle.erase(b.first);
	}
	for (auto &b : blocks)
		(void) hipFree(b.first);
}

// End of a driver call: a cached block that nobody took during the last `max_idle` calls goes back to the device (the working
// set of a process that calls the driver again and again stays -- mk15.b4 parks 80 GB: row pools, the Schur complement, 22 GB of
// accumulators of the low-rank finish, the FIFOs of the pivot search --, what a one-off large call left behind does not).
void big_age(int max_idle)
{
	std::vector<std::pair<void *, size_t>> blocks;
	{
		std::lock_guard<std::mutex> guard(g_big.mutex);
		std::vector<std::pair<void *, size_t>> stay;
		for (auto &b : g_big.free_blocks) {
			const int age = ++g_big.idle[b.first];
			if (age > max_idle) {
				blocks.push_back(b);
				g_big.cached -= b.second;
				g_big.idle.erase(b.first);
			} else {
				stay.push_back(b);
			}
		}
		g_big.free_blocks.swap(stay);
	}
	for (auto &b : blocks)
		(void) hipFree(b.first);
}
}  // namespace sh

namespace {
// The dense accumulator scratch is large (tens of GB on big inputs) and expensive to allocate;
// host-level entry points park it here between calls instead of freeing it (it is all zero
// whenever no kernel is running).
struct ScratchCache {
	unsigned char *ptr = nullptr;
	int64_t bytes = 0;
};
ScratchCache g_scratch_cache;

void scratch_adopt(spasm_hip_dwork *W)
{
	if (g_scratch_cache.ptr != nullptr && W->d_scratch == nullptr) {
		W->d_scratch = g_scratch_cache.ptr;
		W->scratch_bytes = g_scratch_cache.bytes;
		g_scratch_cache.ptr = nullptr;
		g_scratch_cache.bytes = 0;
	}
}

void scratch_park(spasm_hip_dwork *W)
{
	if (W->d_scratch == nullptr)
		return;
	if (g_scratch_cache.ptr != nullptr)
		sh::big_free(g_scratch_cache.ptr);
	g_scratch_cache.ptr = W->d_scratch;
	g_scratch_cache.bytes = W->scratch_bytes;
	W->d_scratch = nullptr;
	W->scratch_bytes = 0;
}

int cu_count()
{
	static int cus = 0;
	if (cus == 0) {
		int dev = 0;
		HIP_CHECK(hipGetDevice(&dev));
		hipDeviceProp_t prop;
		HIP_CHECK(hipGetDeviceProperties(&prop, dev));
		cus = prop.multiProcessorCount;
	}
	return cus;
}

}  // namespace

// ---- device copies of host matrices -----------------------------------------------------------------
// A host-pointer entry point needs its matrix in HBM for the duration of the call.  Inside spasm_hip_echelonize the
// same matrices come back call after call (density estimate, Schur complement, dense finish of round k all read the A
// of round k, and the S that round k produced is the A of round k + 1): the driver switches residency on, and then a
// matrix is uploaded at most once, a Schur complement is kept where it was computed (all-gathered, with several
// GPUs) and never uploaded at all.  Entries are keyed by the address of the host struct; the driver forgets an entry
// before it frees the matrix, so an address is never reused behind the table's back.  Outside the driver every call
// uploads and frees its own copy, as before.
namespace sh {
struct ResidentEntry {
	const struct spasm_csr *host;
	i64 *p;
	int *j, *x;
	i64 nnz;
	bool host_stale;          // the host struct has its row pointers only: j and x were never downloaded (resident_materialize)
	// A Schur complement computed by column slabs (round 6): p / j / x hold THIS RANK'S slab -- all rows, its range of the columns,
	// column numbers of the whole matrix --, the host struct the row pointers of the whole rows (nnz: their total).  It becomes
	// whole rows (all-gatherv + stitching: resident_unslab) only when somebody needs whole rows: another sparse round, a download.
	bool slab = false;
	spasm_hip_comm *comm = nullptr;
	i64 local_nnz = 0;
};
static std::vector<ResidentEntry> g_resident;
// ---- host -> device copies ---------------------------------------------------------------------------
// hipMemcpyAsync from pageable memory pins the caller's pages for large copies, and what that costs depends on the history of
// the process: the 136 MB of mk15.b4 went over in 1 to 36 ms from one driver call to the next, the 5 MB of tables of mk13.b5's
// factor image in 0.7 or 19 ms (`factor_image_ms` 6.8 or 30 from one bench process to the next -- with the same library).  Copies of
// 256 KB and more go through pinned buffers of the library instead: 8 MB pieces, filled by up to four threads (each with two
// buffers of its own, so that a piece is on its way while the next one is filled), sent by hipMemcpyAsync on the caller's
// stream.  The source is consumed when the call returns.  SPASM_HIP_STAGED_H2D=0: the runtime's own path.
namespace {
constexpr size_t H2D_PIECE = (size_t) 8 << 20;
constexpr int H2D_THREADS = 4;
struct H2dLane {
	void *buf[2] = {nullptr, nullptr};
	hipEvent_t ev[2] = {nullptr, nullptr};
	bool busy[2] = {false, false};
};
struct H2dStage {
	std::mutex mutex;
	H2dLane lane[H2D_THREADS];
	bool ready = false, broken = false;
};
H2dStage g_h2d;
}  // namespace

void h2d(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
	if (bytes == 0)
		return;
	static const bool staged = (1) != 0;
	if (!staged || bytes < ((size_t) 256 << 10)) {
		HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
		return;
	}
	std::lock_guard<std::mutex> guard(g_h2d.mutex);
	if (!g_h2d.ready && !g_h2d.broken) {
		for (int t = 0; t < H2D_THREADS && !g_h2d.broken; t++)
			for (int b = 0; b < 2 && !g_h2d.broken; b++)
				if (hipHostMalloc(&g_h2d.lane[t].buf[b], H2D_PIECE, hipHostMallocDefault) != hipSuccess ||
				    hipEventCreateWithFlags(&g_h2d.lane[t].ev[b], hipEventDisableTiming) != hipSuccess) {
					(void) hipGetLastError();
					g_h2d.broken = true;
				}
		g_h2d.ready = !g_h2d.broken;
	}
	if (g_h2d.broken) {
		HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
		return;
	}
	const size_t pieces = (bytes + H2D_PIECE - 1) / H2D_PIECE;
	const int T = (int) std::min<size_t>((size_t) std::max(1, std::min(H2D_THREADS, usable_cpus())), pieces);
	int dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	auto work = [&](int t) {
		if (t > 0 && hipSetDevice(dev) != hipSuccess)
			die("h2d: hipSetDevice(%d) failed on a copy thread", dev);
		H2dLane &L = g_h2d.lane[t];
		int b = 0;
		for (size_t k = (size_t) t; k < pieces; k += (size_t) T, b ^= 1) {
			const size_t off = k * H2D_PIECE, n = std::min(H2D_PIECE, bytes - off);
			if (L.busy[b])
				HIP_CHECK(hipEventSynchronize(L.ev[b]));
			std::memcpy(L.buf[b], static_cast<const char *>(src) + off, n);
			HIP_CHECK(hipMemcpyAsync(static_cast<char *>(dst) + off, L.buf[b], n, hipMemcpyHostToDevice, stream));
			HIP_CHECK(hipEventRecord(L.ev[b], stream));
			L.busy[b] = true;
		}
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < T; t++)
		pool.emplace_back(work, t);
	work(0);
	for (auto &th : pool)
		th.join();
}

// the pinned buffers go back (spasm_hip_release_cached_memory); the next large copy makes new ones
void h2d_release()
{
	std::lock_guard<std::mutex> guard(g_h2d.mutex);
	for (int t = 0; t < H2D_THREADS; t++)
		for (int b = 0; b < 2; b++) {
			H2dLane &L = g_h2d.lane[t];
			if (L.busy[b] && L.ev[b] != nullptr)
				(void) hipEventSynchronize(L.ev[b]);
			L.busy[b] = false;
			if (L.buf[b] != nullptr)
				(void) hipHostFree(L.buf[b]);
			if (L.ev[b] != nullptr)
				(void) hipEventDestroy(L.ev[b]);
			L.buf[b] = nullptr;
			L.ev[b] = nullptr;
		}
	g_h2d.ready = false;
	g_h2d.broken = false;
}

// (the way back: 256 KB to 256 MB -- the pivots and labels of a search, the row pointers of a Schur complement -- through the same
//  buffers; larger ones -- a whole Schur complement for a host round -- keep the runtime's path, which pins once and streams)
void d2h(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
	if (bytes == 0)
		return;
	static const bool staged = (1) != 0;
	bool direct = !staged || bytes < ((size_t) 256 << 10) || bytes > ((size_t) 256 << 20);
	std::unique_lock<std::mutex> guard(g_h2d.mutex, std::defer_lock);
	if (!direct) {
		guard.lock();
		direct = !g_h2d.ready;          // (the buffers are made by the first staged copy to the device)
	}
	if (direct) {
		HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		return;
	}
	const size_t pieces = (bytes + H2D_PIECE - 1) / H2D_PIECE;
	const int T = (int) std::min<size_t>((size_t) std::max(1, std::min(H2D_THREADS, usable_cpus())), pieces);
	int dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	auto work = [&](int t) {
		if (t > 0 && hipSetDevice(dev) != hipSuccess)
			die("d2h: hipSetDevice(%d) failed on a copy thread", dev);
		H2dLane &L = g_h2d.lane[t];
		for (int b = 0; b < 2; b++)
			if (L.busy[b]) {          // (a copy TO the device may still be reading the buffer)
				HIP_CHECK(hipEventSynchronize(L.ev[b]));
				L.busy[b] = false;
			}
		size_t pend_off = 0, pend_n = 0;
		int pend_b = -1, b = 0;
		for (size_t k = (size_t) t; k < pieces; k += (size_t) T, b ^= 1) {
			const size_t off = k * H2D_PIECE, n = std::min(H2D_PIECE, bytes - off);
			HIP_CHECK(hipMemcpyAsync(L.buf[b], static_cast<const char *>(src) + off, n, hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipEventRecord(L.ev[b], stream));
			if (pend_b >= 0) {
				HIP_CHECK(hipEventSynchronize(L.ev[pend_b]));
				std::memcpy(static_cast<char *>(dst) + pend_off, L.buf[pend_b], pend_n);
			}
			pend_off = off;
			pend_n = n;
			pend_b = b;
		}
		if (pend_b >= 0) {
			HIP_CHECK(hipEventSynchronize(L.ev[pend_b]));
			std::memcpy(static_cast<char *>(dst) + pend_off, L.buf[pend_b], pend_n);
		}
	};
	std::vector<std::thread> pool;
	for (int t = 1; t < T; t++)
		pool.emplace_back(work, t);
	work(0);
	for (auto &th : pool)
		th.join();
}

static bool g_resident_on = false;
static bool g_lazy_download = false;
static i64 g_resident_uploads = 0, g_resident_hits = 0;

void resident_begin() { g_resident_on = true; }

void resident_forget(const struct spasm_csr *A)
{
	for (size_t t = 0; t < g_resident.size(); t++)
		if (g_resident[t].host == A) {
			sh::big_free(g_resident[t].p);
			big_free(g_resident[t].j);          // (may be the output arrays of a workspace, which come from the block cache)
			big_free(g_resident[t].x);
			g_resident.erase(g_resident.begin() + (long) t);
			return;
		}
}

void fact_cache_age();

void resident_end()
{
	while (!g_resident.empty())
		resident_forget(g_resident.back().host);
	g_resident_on = false;
	fact_cache_age();
	// The driver is done.  Its large blocks (images, row pools, accumulators, FIFOs: 60-90 GB on mk14.b4) STAY in the cache, up
	// to SPASM_HIP_KEEP_GB (default: the cap of the cache, 96; 0: everything goes back): allocating and freeing multi-GB blocks
	// is erratic on these boxes -- 0.1 to 1 s apiece, now and then -- and made five of eight consecutive mk14.b4 calls take
	// 1.0-2.3 s instead of 0.55 (keeping 32 or 64 GB did not help: whatever is handed back comes back slowly).
	// spasm_hip_release_cached_memory() gives everything back; a failed hipMalloc of the library's own does too (big_alloc).
	if ((0) == 0) {
		// (default: a third of the device memory, at most the cap of the cache -- a quarter, 72 GB here, was tried in round 4 and is
		//  less than one mk15.b4 call parks: every call then gave 10-20 GB back and paid 1-2 s to get them again, in whichever
		//  stage asked first --; and whatever sat unused through two calls in a row goes back whatever the total)
		size_t free_b = 0, total_b = 0;
		HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
		const int dflt = (int) std::min<size_t>(96, (total_b >> 30) / 3);
		big_age((2));
		big_trim((size_t) std::max(0, env_int("SPASM_HIP_KEEP_GB", dflt)) << 30);
	}
}

void resident_counters(i64 *uploads, i64 *hits)
{
	*uploads = g_resident_uploads;
	*hits = g_resident_hits;
}

// takes ownership of device arrays that hold the matrix `host` describes (a fresh Schur complement)
void resident_adopt(const struct spasm_csr *host, i64 *dp, int *dj, int *dx, bool host_stale)
{
	resident_forget(host);
	g_resident.push_back(ResidentEntry{host, dp, dj, dx, host->p[host->n], host_stale});
}

// The driver may ask spasm_hip_schur to leave the entries of S on the device (the host struct gets its row pointers only):
// whether the next round needs them on the host at all is decided by a census on the device first (resident_fl_census).
void resident_lazy_downloads(bool on) { g_lazy_download = on; }

void resident_adopt_slab(const struct spasm_csr *host, i64 *dp, int *dj, int *dx, i64 local_nnz, spasm_hip_comm *comm)
{
	resident_forget(host);
	ResidentEntry e{host, dp, dj, dx, host->p[host->n], true};
	e.slab = true;
	e.comm = comm;
	e.local_nnz = local_nnz;
	g_resident.push_back(e);
}

// the communicator of a matrix that is resident as column slabs, else null
spasm_hip_comm *resident_slab_comm(const struct spasm_csr *A)
{
	for (const ResidentEntry &e : g_resident)
		if (e.host == A && e.slab)
			return e.comm;
	return nullptr;
}

// slabs -> whole rows on every device (collective: every rank holds a slab of the same matrix and gets here at the same point)
static void resident_unslab(ResidentEntry &e, hipStream_t stream)
{
	if (!e.slab)
		return;
	const int n = e.host->n;
	const int world = comm_world(e.comm);
	i64 total = 0;
	int rows_all = 0;
	(void) comm_allgatherv_csr(e.comm, n, e.local_nnz, e.p, e.j, e.x, nullptr, nullptr, nullptr, -1, &rows_all, &total, stream);
	if (rows_all != n * world || total != e.nnz)
		die("column slabs: the ranks hold %d slab rows and %lld entries in all, %d and %lld expected", rows_all, (long long) total, n * world, (long long) e.nnz);
	i64 *gSp = dalloc<i64>((i64) rows_all + 1);
	int *gSj = dalloc<int>(total);
	int *gSx = dalloc<int>(total);
	if (comm_allgatherv_csr(e.comm, n, e.local_nnz, e.p, e.j, e.x, gSp, gSj, gSx, total, nullptr, nullptr, stream) != 0)
		die("column slabs: the all-gatherv failed");
	i64 *dSp = dalloc<i64>((i64) n + 1);
	int *dSj = static_cast<int *>(big_alloc((size_t) std::max<i64>(total, 1) * sizeof(int)));
	int *dSx = static_cast<int *>(big_alloc((size_t) std::max<i64>(total, 1) * sizeof(int)));
	int *d_len = dalloc<int>(n);
	unsigned long long *d_bs = dalloc<unsigned long long>((i64) (n + 1023) / 1024 + 16);
	int *d_ctr = dalloc<int>(CTR_COUNT);
	HIP_CHECK(hipMemsetAsync(d_ctr, 0, CTR_COUNT * sizeof(int), stream));
	launch_stitch_slabs(gSp, gSj, gSx, n, world, dSp, dSj, dSx, total, d_len, d_bs, d_ctr, stream);
	HIP_CHECK(hipStreamSynchronize(stream));
	sh::big_free(gSp);
	sh::big_free(gSj);
	sh::big_free(gSx);
	sh::big_free(d_len);
	sh::big_free(d_bs);
	sh::big_free(d_ctr);
	sh::big_free(e.p);
	big_free(e.j);
	big_free(e.x);
	e.p = dSp;
	e.j = dSj;
	e.x = dSx;
	e.slab = false;
	e.comm = nullptr;
	counters()[CNT_SLABS_GATHERED] += 1;
	if (verbose() >= 2)
		logmsg("[schur/hip] column slabs gathered and stitched into whole rows on every device (%lld entries): somebody needs whole rows\n", (long long) total);
}

// brings the entries of a lazily returned Schur complement to its host struct (no-op when they are there)
void resident_materialize(const struct spasm_csr *A)
{
	for (ResidentEntry &e : g_resident)
		if (e.host == A && e.slab)
			resident_unslab(e, nullptr);
	for (ResidentEntry &e : g_resident)
		if (e.host == A && e.host_stale) {
			if (e.nnz > 0) {
				HIP_CHECK(hipMemcpy(A->j, e.j, (size_t) e.nnz * sizeof(int), hipMemcpyDeviceToHost));
				HIP_CHECK(hipMemcpy(A->x, e.x, (size_t) e.nnz * sizeof(int), hipMemcpyDeviceToHost));
			}
			e.host_stale = false;
			return;
		}
}

__global__ __launch_bounds__(256) void fl_census_kernel(const i64 *Ap, const int *Aj, int n, uint32_t *bitmap)
{
	// one wave per row: the smallest column of the row (rows need not be sorted), one bit per distinct column
	const int row = (int) ((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
	if (row >= n)
		return;
	// (four loads in flight per lane: with one, the 757 M entries of mk15.b4's Schur complement took 7.6 ms -- 0.4 TB/s)
	int best = 0x7FFFFFFF;
	const i64 lo = Ap[row], hi = Ap[row + 1];
	i64 px = lo + lane;
	for (; px + 192 < hi; px += 256) {
		const int a = Aj[px], b = Aj[px + 64], c = Aj[px + 128], d = Aj[px + 192];
		best = min(min(best, min(a, b)), min(c, d));
	}
	for (; px < hi; px += 64)
		best = min(best, Aj[px]);
	for (int d = 32; d >= 1; d >>= 1)
		best = min(best, __shfl_xor(best, d));
	// (the rows of a dense Schur complement nearly all start in its first few columns: millions of atomics on the same word are
	//  served one after the other -- look first)
	if (lane == 0 && best != 0x7FFFFFFF && (__builtin_nontemporal_load(&bitmap[best >> 5]) & (1u << (best & 31))) == 0)
		atomicOr(&bitmap[best >> 5], 1u << (best & 31));
}

__global__ __launch_bounds__(256) void popcount_kernel(const uint32_t *bitmap, int nwords, int *out)
{
	int c = 0;
	for (int w = blockIdx.x * 256 + threadIdx.x; w < nwords; w += gridDim.x * 256)
		c += __popc(bitmap[w]);
	for (int d = 32; d >= 1; d >>= 1)
		c += __shfl_xor(c, d);
	if ((threadIdx.x & 63) == 0 && c != 0)
		atomicAdd(out, c);
}

// How many pivots would the first step of the structural search (Faugere-Lachartre: the leftmost entry of every row, one
// row per column -- spasm_pivots.c:35-80, host_pivots.cpp leftmost_entries) find on this device-resident matrix?  = the
// number of distinct leftmost columns.  -1 when the matrix is not resident.
__global__ __launch_bounds__(256) void row_min_column_kernel(const i64 *Ap, const int *Aj, int n, int *out)
{
	const int row = (int) ((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
	if (row >= n)
		return;
	int best = 0x7FFFFFFF;
	const i64 lo = Ap[row], hi = Ap[row + 1];
	i64 px = lo + lane;
	for (; px + 192 < hi; px += 256) {
		const int a = Aj[px], b = Aj[px + 64], c = Aj[px + 128], d = Aj[px + 192];
		best = min(min(best, min(a, b)), min(c, d));
	}
	for (; px < hi; px += 64)
		best = min(best, Aj[px]);
	for (int d = 32; d >= 1; d >>= 1)
		best = min(best, __shfl_xor(best, d));
	if (lane == 0)
		out[row] = best;
}

__global__ __launch_bounds__(256) void columns_to_bitmap_kernel(const int *col, int n, uint32_t *bitmap)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n)
		return;
	const int c = col[i];
	// (as in fl_census_kernel: most rows name the same few columns -- look before the atomic)
	if (c != 0x7FFFFFFF && (__builtin_nontemporal_load(&bitmap[c >> 5]) & (1u << (c & 31))) == 0)
		atomicOr(&bitmap[c >> 5], 1u << (c & 31));
}

__global__ __launch_bounds__(256) void row_lengths_kernel(const i64 *Sp, int n, int *len)
{
	const int i = blockIdx.x * 256 + threadIdx.x;
	if (i < n)
		len[i] = (int) (Sp[i + 1] - Sp[i]);
}

int resident_fl_census(const struct spasm_csr *A)
{
	for (const ResidentEntry &e : g_resident)
		if (e.host == A) {
			const int n = A->n, m = A->m;
			const int nwords = (m + 31) / 32 + 1;
			uint32_t *bm = nullptr;
			HIP_CHECK(sh::malloc_or_trim((void **) &bm, ((size_t) nwords + 1) * sizeof(uint32_t)));
			HIP_CHECK(hipMemset(bm, 0, ((size_t) nwords + 1) * sizeof(uint32_t)));
			if (e.slab && n > 0) {
				// column slabs: the leftmost entry of a row is the smallest of the leftmost entries of its pieces -- one minimum per row
				// over the ranks (4 bytes per row: nothing next to the rows themselves)
				int *d_left = dalloc<int>(n);
				hipLaunchKernelGGL(row_min_column_kernel, dim3((unsigned) (((i64) n * 64 + 255) / 256)), dim3(256), 0, nullptr, e.p, e.j, n, d_left);
				comm_allreduce_min_i32(e.comm, d_left, n, nullptr);
				hipLaunchKernelGGL(columns_to_bitmap_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, nullptr, d_left, n, bm);
				HIP_CHECK(hipStreamSynchronize(nullptr));
				sh::big_free(d_left);
			} else if (n > 0)
				hipLaunchKernelGGL(fl_census_kernel, dim3((unsigned) (((i64) n * 64 + 255) / 256)), dim3(256), 0, nullptr, e.p, e.j, n, bm);
			hipLaunchKernelGGL(popcount_kernel, dim3(64), dim3(256), 0, nullptr, bm, nwords, reinterpret_cast<int *>(bm + nwords));
			int count = 0;
			HIP_CHECK(hipMemcpy(&count, bm + nwords, sizeof(int), hipMemcpyDeviceToHost));
			sh::big_free(bm);
			return count;
		}
	return -1;
}

bool resident_enabled() { return g_resident_on; }

// A on its way to the device while the host does something else (the Faugere-Lachartre steps of a round: host_pivots.cpp): the
// matrix is resident by the time the first device stage of the round asks for it.  Only inside a driver call (resident_begin).
void resident_prefetch(const struct spasm_csr *A)
{
	if (!g_resident_on)
		return;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		(void) hipGetLastError();
		return;
	}
	DeviceMatrix up(A, nullptr);
}
// (for translation units that do not see device_types.h)
bool resident_prefetch_possible() { return g_resident_on; }
int resident_current_device()
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess)
		(void) hipGetLastError();
	return dev;
}
void resident_prefetch_matrix(const struct spasm_csr *A, int dev)          // (called on a thread of its own: the caller's device is not this thread's by default)
{
	if (hipSetDevice(dev) != hipSuccess) {
		(void) hipGetLastError();
		return;
	}
	resident_prefetch(A);
}

DeviceMatrix::DeviceMatrix(const struct spasm_csr *A, hipStream_t stream, bool slab_will_do)
{
	nnz = A->p[A->n];
	for (ResidentEntry &e : g_resident)
		if (e.host == A && e.nnz == nnz) {
			if (e.slab && !slab_will_do)
				resident_unslab(e, stream);          // (whoever reads whole rows: another sparse round, a density sample)
			if (e.slab)
				nnz = e.local_nnz;
			p = e.p;
			j = e.j;
			x = e.x;
			owned = false;
			g_resident_hits += 1;
			return;
		}
	HIP_CHECK(sh::malloc_or_trim((void **) &p, ((size_t) A->n + 1) * sizeof(i64)));
	HIP_CHECK(sh::malloc_or_trim((void **) &j, (size_t) (nnz > 0 ? nnz : 1) * sizeof(int)));
	HIP_CHECK(sh::malloc_or_trim((void **) &x, (size_t) (nnz > 0 ? nnz : 1) * sizeof(int)));
	h2d(p, A->p, ((size_t) A->n + 1) * sizeof(i64), stream);
	if (nnz > 0) {
		h2d(j, A->j, (size_t) nnz * sizeof(int), stream);
		h2d(x, A->x, (size_t) nnz * sizeof(int), stream);
	}
	HIP_CHECK(hipStreamSynchronize(stream));
	g_resident_uploads += 1;
	if (g_resident_on) {
		g_resident.push_back(ResidentEntry{A, p, j, x, nnz});
		owned = false;
	} else {
		owned = true;
	}
}

DeviceMatrix::~DeviceMatrix()
{
	if (owned) {
		sh::big_free(p);
		sh::big_free(j);
		sh::big_free(x);
	}
}
}  // namespace sh

// ---- one-entry cache of the factor image for the host-pointer wrappers -------------------------
// A round of the driver calls spasm_hip_schur_estimate_density, spasm_hip_schur and the dense-row
// functions on the same (U, qinv): planning + uploading the image once is enough.  The key is the
// device, the shape of U and a hash of ALL of U->p, U->j, U->x and qinv (one linear pass: far cheaper
// than planning + uploading), so a factor edited in place or reallocated at the same address is never
// mistaken for the cached one.  One mutex serialises the cache (host entry points may be called from
// several threads).
namespace sh {
struct FactCacheKey {
	int device = -1;
	int n = -1, m = -1;
	i64 nnz = -1, prime = -1;
	uint64_t sum = 0;
	bool operator==(const FactCacheKey &o) const
	{
		return device == o.device && n == o.n && m == o.m && nnz == o.nnz && prime == o.prime && sum == o.sum;
	}
};
// two slots, most recently used first: a call split by columns plans the image of its slab factor next to the image of the whole
// factor, which the driver comes back to (density samples, the finish) -- one slot had each evict the other
static std::mutex g_fact_mutex;
static FactCacheKey g_fact_key[2];
static spasm_hip_dfact *g_fact[2] = {nullptr, nullptr};
static long long g_fact_call[2] = {0, 0};          // the driver call (g_driver_calls) that last used the slot
static long long g_driver_calls = 0;

// the end of a driver call: the second slot goes when the call did not use it (an image can hold gigabytes -- the dense R of
// ch7-8.b5's factor, 15 GB, sat there through every later call of the process on other matrices)
void fact_cache_age()
{
	std::lock_guard<std::mutex> guard(g_fact_mutex);
	if (g_fact[1] != nullptr && g_fact_call[1] != g_driver_calls) {
		spasm_hip_dfact_destroy(g_fact[1]);
		g_fact[1] = nullptr;
	}
	g_driver_calls += 1;
}

static uint64_t hash_words(uint64_t h, const void *data, size_t bytes)
{
	// four independent multiply-xor lanes over 8-byte words (memory-bound), folded at the end
	const uint64_t *w = static_cast<const uint64_t *>(data);
	const size_t nw = bytes / 8;
	uint64_t l0 = h, l1 = h ^ 0x9E3779B97F4A7C15ULL, l2 = h ^ 0xC2B2AE3D27D4EB4FULL, l3 = h ^ 0x165667B19E3779F9ULL;
	size_t t = 0;
	for (; t + 4 <= nw; t += 4) {
		l0 = (l0 ^ w[t]) * 0x100000001B3ULL;
		l1 = (l1 ^ w[t + 1]) * 0x100000001B3ULL;
		l2 = (l2 ^ w[t + 2]) * 0x100000001B3ULL;
		l3 = (l3 ^ w[t + 3]) * 0x100000001B3ULL;
		l0 ^= l0 >> 29;
	}
	for (; t < nw; t++)
		l0 = (l0 ^ w[t]) * 0x100000001B3ULL;
	const unsigned char *tail = static_cast<const unsigned char *>(data) + nw * 8;
	for (size_t b = 0; b < bytes % 8; b++)
		l1 = (l1 ^ tail[b]) * 0x100000001B3ULL;
	uint64_t r = l0;
	r = (r ^ (l1 + (r << 6) + (r >> 2))) * 0x100000001B3ULL;
	r = (r ^ (l2 + (r << 6) + (r >> 2))) * 0x100000001B3ULL;
	r = (r ^ (l3 + (r << 6) + (r >> 2))) * 0x100000001B3ULL;
	return r;
}

spasm_hip_dfact *cached_dfact(const struct spasm_csr *U, const int *qinv, hipStream_t stream)
{
	std::lock_guard<std::mutex> guard(g_fact_mutex);
	FactCacheKey key;
	HIP_CHECK(hipGetDevice(&key.device));
	key.n = U->n;
	key.m = U->m;
	key.nnz = U->p[U->n];
	key.prime = U->field->p;
	uint64_t h = 1469598103934665603ULL;
	h = hash_words(h, qinv, (size_t) U->m * sizeof(int));
	h = hash_words(h, U->p, ((size_t) U->n + 1) * sizeof(i64));
	h = hash_words(h, U->j, (size_t) key.nnz * sizeof(int));
	h = hash_words(h, U->x, (size_t) key.nnz * sizeof(spasm_ZZp));
	key.sum = h;
	// (an image built without the tables of the pull experiment does not serve a call that asks for that kernel)
	for (int slot = 0; slot < 2; slot++)
		if (g_fact[slot] != nullptr && key == g_fact_key[slot] && (g_fact[slot]->has_pull || env_int("SPASM_HIP_PULL", 0) == 0)) {
			if (slot == 1) {
				std::swap(g_fact[0], g_fact[1]);
				std::swap(g_fact_key[0], g_fact_key[1]);
				std::swap(g_fact_call[0], g_fact_call[1]);
			}
			g_fact_call[0] = g_driver_calls;
			return g_fact[0];
		}
	if (g_fact[1] != nullptr)
		spasm_hip_dfact_destroy(g_fact[1]);
	g_fact[1] = g_fact[0];
	g_fact_key[1] = g_fact_key[0];
	g_fact_call[1] = g_fact_call[0];
	g_fact[0] = spasm_hip_dfact_create(U, qinv, stream);
	counters()[CNT_FACTOR_PLANS] += 1;
	g_fact_key[0] = key;
	g_fact_call[0] = g_driver_calls;
	return g_fact[0];
}
}  // namespace sh

namespace sh {
// Should this batch of `nrows` rows go through the back-substituted image?  A cost model fitted on thirteen generated
// factors (tools/sweep_cost.py, table in DESIGN.md section 3; 650 <= Sm <= 23,958, densities 0.4 % - 72 %), seconds:
//   build of R (only if it is not there yet)   max(15 us per chunk of 768 rows, (r + D) Sm e / 2.1 TB/s)      D = pivotal entries of U'
//   apply + expansion                          (P Sm e + 2 nrows ldR e + 8 nnz(S)) / 4.5 TB/s + 0.4 ms        P = pivotal entries of the rows
//   row by row                                 0.3 ms + 27 ps per (row, pivot) elimination
// The number of eliminations is what the row-by-row kernels measure on the density sample of the driver
// (spasm_hip_schur on 100 rows: bs.elim_hint); without a sample, 5 % of the pivots per row (the family: 1.6 % - 22 %).
// On full batches the image won on all thirteen (1.05x on mk13.b4 ... 11x on mk13.b5), so the rule of round 2 (Sm <= 8192
// or density >= 0.25) lost up to 1.7x on the four widest; a one-off batch of a few thousand rows does not pay for the build.
// SPASM_HIP_BACKSOLVE=0 never, =1 whenever the factor has a plan; tests that force a tier or the row-group kernel
// switch the image off; batches under 1024 rows (density samples, completion tests) only trigger a build when Sm <= 8192.
bool backsolve_wanted(const spasm_hip_dfact *F, bool other_path_forced, int nrows)
{
	const char *e = sh::env_get("SPASM_HIP_BACKSOLVE");
	const int mode = (e == nullptr || *e == 0) ? -1 : std::atoi(e);
	if (mode == 0 || !(F->bs.planned || F->bs_deferred))
		return false;
	if (F->bs.d_R == nullptr) {
		size_t free_b = 0, total_b = 0;
		sh::mem_info(&free_b, &total_b);
		if ((size_t) F->bs.r * (size_t) F->bs.ldR * 4 > free_b / 2)
			return false;
	}
	if (mode == 1)
		return true;
	if (other_path_forced)
		return false;
	if (F->bs.valid)
		return true;                      // R is there: a few of its rows per reduced row always beat an elimination
	const BsImage &B = F->bs;
	// small batches (the driver's density sample of 100 rows, completion tests): rows of R of at most 8,192 entries are
	// built at once -- at worst a small loss, and what follows on such a factor (the Schur complement, the dense finish)
	// will want R anyway; wider factors are sampled row by row, which is also what measures their eliminations per row
	if (nrows < 1024)
		return B.Sm <= 8192;
	const double eb = (F->prime < 65536) ? 2.0 : 4.0;
	const double r = (double) B.r, Sm = (double) B.Sm, n = (double) nrows;
	const double t_build = std::max(15e-6 * std::ceil(r / 768.0), (r + (double) B.ndeps) * Sm * eb / 2.1e12);
	const double density = (B.density_hint >= 0.0) ? std::min(1.0, B.density_hint) : 0.05;
	const double pivotal_per_row = 3.0;          // (entries of a row of A on pivotal columns: 2.9 - 5.1 in the family)
	const double t_apply = (pivotal_per_row * n * Sm * eb + 2.0 * n * (double) B.ldR * eb + 8.0 * density * n * Sm) / 4.5e12 + 0.4e-3;
	const double elim_per_row = (B.elim_hint >= 0.0) ? B.elim_hint : 0.05 * r;
	const double t_rows = 0.3e-3 + 27e-12 * elim_per_row * n;
	return t_build + t_apply <= t_rows;
}

// Should this batch go through the SPARSE image (sparse_image.hip)?  It is made for Schur complements that stay sparse on
// many columns: R is then mostly zeros, and both the build and the rows of S cost what R and S hold, not r x Sm.
// SPASM_HIP_SPARSE_IMAGE=0 never, =1 whenever the factor has the plan (tests); else: full batches (>= 1,024 rows) whose
// expected density is under 12 % (the driver's estimate; unknown: 3 % assumed beyond 16,384 non-pivotal columns), when the
// chain of level launches of the build (~10 us each) is small against what the other paths would take.
bool sparse_image_wanted(const spasm_hip_dfact *F, bool other_path_forced, int nrows)
{
	const SpImage &S = F->sp;
	const int mode = env_int("SPASM_HIP_SPARSE_IMAGE", -1);
	if (mode == 0 || !sparse_image_plan_expected(F) || S.failed)
		return false;
	if (mode == 1)
		return sparse_image_planned(F, nullptr);
	if (other_path_forced || env_int("SPASM_HIP_BACKSOLVE", -1) >= 0)
		return false;
	if (S.valid)
		return true;
	if (nrows < 1024)
		return false;          // (a density sample: the tables may still be on their way, and nobody waits for them here)
	const BsImage &B = F->bs;
	const double density = (B.density_hint >= 0.0) ? B.density_hint : (S.Sm > 16384 ? 0.03 : 1.0);
	if (density >= 0.12)
		return false;
	const double r = (double) S.r, Sm = (double) S.Sm, n = (double) nrows;
	const double eb = 2.0;
	const double t_dense = std::max(15e-6 * std::ceil(r / 768.0), (r + (double) S.ndeps) * Sm * eb / 2.1e12) +
	                       (3.0 * n * Sm * eb + 2.0 * n * Sm * eb + 8.0 * density * n * Sm) / 4.5e12 + 0.4e-3;
	const double elim_per_row = (B.elim_hint >= 0.0) ? B.elim_hint : 0.05 * r;
	const double t_rows = 0.3e-3 + 27e-12 * elim_per_row * n;
	const double t_other = (B.planned || F->bs_deferred) ? std::min(t_dense, t_rows) : t_rows;
	const double t_sparse = 8e-6 * (double) S.nlevels + 27e-12 * density * n * Sm + 3e-9 * n + 0.5e-3;
	return t_sparse < t_other && sparse_image_planned(F, nullptr);
}
}  // namespace sh

extern "C" {

void spasm_hip_shard(int n, int rank, int world, int *lo, int *hi);
int spasm_hip_dschur_allgatherv(spasm_hip_comm *c, const spasm_hip_dwork *W, i64 *d_Sp, int *d_Sj, spasm_ZZp *d_Sx, i64 cap,
                                int *total_rows, i64 *total_nnz, void *stream);
int spasm_hip_column_slab(const struct spasm_csr *A, const struct spasm_lu *fact, int part, int parts, struct spasm_csr **A_slab,
                          struct spasm_lu **fact_slab, int *cols);
void spasm_hip_lu_free(struct spasm_lu *N);

int spasm_hip_debug_plan(const struct spasm_csr *U, const int *qinv, int *label_of_row, int *lvl_end_of_row, int *lab,
                         int *info);

// (tests) a buffer of `bytes` bytes taken and given back twice: 1 when the second request was served by the cache, 0 when it went
// to the device again
int spasm_hip_debug_block_cache_roundtrip(size_t bytes)
{
	void *first = big_alloc(bytes);
	sh::big_free(first);
	const long long before = g_device_mallocs.load();
	void *second = big_alloc(bytes);
	sh::big_free(second);
	return g_device_mallocs.load() == before ? 1 : 0;
}

// everything the library parks on the device between calls goes back to it: the block cache, the accumulator scratch
void spasm_hip_release_cached_memory(void)
{
	big_trim(0);
	sh::h2d_release();
	if (g_scratch_cache.ptr != nullptr) {
		sh::big_free(g_scratch_cache.ptr);
		g_scratch_cache.ptr = nullptr;
		g_scratch_cache.bytes = 0;
	}
}

int spasm_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

// --------------------------------------------------------------------------
// factor image
// --------------------------------------------------------------------------
// Host part of the factor image (sh::FactPlan, device_types.h): checks U, computes the elimination levels,
// the column labels and the relabelled rows.  No GPU involved (unit-tested on
// the CPU through spasm_hip_debug_plan).
// Levels that somebody else has computed already (round 5, late): the device pivot search ends with a depth label for every
// column -- every other entry of a pivot row has a larger label than its pivot -- so "largest label minus the label of the
// row's pivot" is a valid height for every row of a factor made of that search's pivots alone.  The search leaves them here
// (spasm_hip_pivots_extract_structural: first round of a driver call); plan_factor takes them when they belong to the factor
// it is given AND a threaded pass confirms, entry by entry, that they order it (so a stale or foreign hint costs a millisecond,
// never a wrong schedule), and skips its own serial pass over the rows (8 ms on mk15.b4's 604,000).
extern "C++" {
namespace sh {
struct LevelHint {
	const void *U = nullptr;
	int rows = 0;
	std::vector<int> height;          // per row of U
};
static LevelHint g_level_hint;
static std::mutex g_level_hint_mutex;

void level_hint_set(const struct spasm_csr *U, int rows, std::vector<int> &&height)
{
	std::lock_guard<std::mutex> guard(g_level_hint_mutex);
	g_level_hint.U = U;
	g_level_hint.rows = rows;
	g_level_hint.height = std::move(height);
}
}  // namespace sh
}  // extern "C++"

static bool heights_from_hint(const struct spasm_csr *U, const int *qinv, std::vector<int> &height)
{
	const int r = U->n;
	{
		std::lock_guard<std::mutex> guard(sh::g_level_hint_mutex);
		if (sh::g_level_hint.U != (const void *) U || sh::g_level_hint.rows != r || (int) sh::g_level_hint.height.size() != r || r < 20000 ||
		    (1) == 0)
			return false;
		height = sh::g_level_hint.height;
	}
	const int T = std::max(1, std::min(16, usable_cpus()));
	std::atomic<int> bad{0};
	auto check = [&](int k0, int k1) {
		for (int k = k0; k < k1 && bad.load(std::memory_order_relaxed) == 0; k++) {
			const int h = height[k];
			if (h < 0) {
				bad.store(1);
				return;
			}
			for (i64 px = U->p[k] + 1; px < U->p[k + 1]; px++) {
				const int k2 = qinv[U->j[px]];
				if (k2 >= 0 && (k2 == k || height[k2] >= h)) {
					bad.store(1);
					return;
				}
			}
		}
	};
	sh::pool_run(T, [&](int t) { check((int) ((i64) r * t / T), (int) ((i64) r * (t + 1) / T)); });
	if (bad.load() != 0)
		return false;
	// the labels of a search have gaps: heights without empty levels (same order)
	int top = 0;
	for (int k = 0; k < r; k++)
		top = std::max(top, height[k]);
	std::vector<int> rank_of((size_t) top + 1, 0);
	for (int k = 0; k < r; k++)
		rank_of[(size_t) height[k]] = 1;
	int next = 0;
	for (int h = 0; h <= top; h++)
		if (rank_of[(size_t) h]) {
			rank_of[(size_t) h] = next;
			next += 1;
		}
	for (int k = 0; k < r; k++)
		height[k] = rank_of[(size_t) height[k]];
	return true;
}

static void plan_factor(const struct spasm_csr *U, const int *qinv, FactPlan &P)
{
	const int r = U->n, m = U->m;
	const i64 prime = U->field->p;
	P.m = m;
	P.r = r;
	P.prime = prime;
	double t_mark = wtime();
	auto lap = [&](const char *what) {
		if (verbose() >= 3)
			logmsg("[factor image/plan] %s %.2f ms\n", what, 1e3 * (wtime() - t_mark));
		t_mark = wtime();
	};
	for (int k = 0; k < r; k++) {
		if (U->p[k + 1] == U->p[k])
			die("row %d of U is empty", k);
		const int j = U->j[U->p[k]];
		if (j < 0 || j >= m || qinv[j] != k)
			die("row %d of U does not start with its pivot (column %d, qinv says row %d)", k, j, (j >= 0 && j < m) ? qinv[j] : -1);
		if (U->x[U->p[k]] != 1)
			die("pivot of row %d of U is not 1", k);
	}
	int npivcols = 0;
	for (int j = 0; j < m; j++)
		if (qinv[j] >= 0) {
			if (qinv[j] >= r)
				die("qinv[%d] = %d but U has %d rows", j, qinv[j], r);
			npivcols += 1;
		}
	if (npivcols != r)
		die("qinv marks %d pivotal columns but U has %d rows", npivcols, r);

	// height of every pivot row in the dependency DAG (row k depends on the
	// pivot rows of the pivotal columns it touches); iterative DFS
	std::vector<int> height((size_t) (r > 0 ? r : 1), 0);
	// The rows of a factor come in topological order as a rule (spasm_pivots.c:307-372 appends them that way, round after
	// round: a row only touches pivot columns of rows after it): then one pass from the last row to the first gives the
	// heights.  Any dependency on an earlier row sends the whole computation to the general search below.
	bool forward_only = true;
	const bool hinted = heights_from_hint(U, qinv, height);
	if (!hinted)
		std::fill(height.begin(), height.end(), 0);
	for (int k = r - 1; k >= 0 && forward_only && !hinted; k--) {
		int h = 0;
		for (i64 px = U->p[k] + 1; px < U->p[k + 1]; px++) {
			const int k2 = qinv[U->j[px]];
			if (k2 < 0)
				continue;
			if (k2 <= k) {
				forward_only = false;
				break;
			}
			h = std::max(h, height[k2] + 1);
		}
		height[k] = h;
	}
	if (!forward_only && !hinted) {
		std::fill(height.begin(), height.end(), 0);
		std::vector<unsigned char> state((size_t) (r > 0 ? r : 1), 0);   // 0 new, 1 open, 2 done
		std::vector<int> stk;
		std::vector<i64> pos;
		for (int root = 0; root < r; root++) {
			if (state[root])
				continue;
			stk.assign(1, root);
			pos.assign(1, U->p[root] + 1);
			state[root] = 1;
			while (!stk.empty()) {
				const int k = stk.back();
				bool down = false;
				for (i64 px = pos.back(); px < U->p[k + 1]; px++) {
					const int k2 = qinv[U->j[px]];
					if (k2 < 0)
						continue;
					if (k2 == k)
						die("row %d of U holds its pivot column twice", k);
					if (state[k2] == 1)
						die("the pivots of U are not triangular (cycle through rows %d and %d)", k, k2);
					if (state[k2] == 0) {
						pos.back() = px;        // come back to this entry
						state[k2] = 1;
						stk.push_back(k2);
						pos.push_back(U->p[k2] + 1);
						down = true;
						break;
					}
					if (height[k2] + 1 > height[k])
						height[k] = height[k2] + 1;
				}
				if (!down) {
					state[k] = 2;
					stk.pop_back();
					pos.pop_back();
				}
			}
		}
	}
	lap(hinted ? "checks + heights (levels of the device pivot search, verified)" : "checks + heights");
	int hmax = 0;
	for (int k = 0; k < r; k++)
		hmax = std::max(hmax, height[k]);
	const int nlev = (r > 0) ? hmax + 1 : 0;
	P.nlevels = nlev;
	// level = hmax - height; stable counting sort inside a level; every level
	// starts on a multiple of 32 labels so that a level is a whole number of
	// words of the pending bitmaps
	std::vector<int> lvl_count((size_t) nlev + 1, 0);
	for (int k = 0; k < r; k++)
		lvl_count[hmax - height[k]] += 1;
	P.lvl_count = lvl_count;
	std::vector<int> lvl_start((size_t) nlev + 1, 0);
	// (levels of fewer than 32 pivots are packed without padding, so that chains of tiny levels
	// do not inflate the label space; a bitmap word shared by several levels is marked MIXED)
	for (int l = 0; l < nlev; l++) {
		int begin = lvl_start[l];
		if (lvl_count[l] >= 32)
			begin = (begin + 31) / 32 * 32;
		lvl_start[l] = begin;
		lvl_start[l + 1] = begin + lvl_count[l];
	}
	const int rpad = (lvl_start[nlev] + 31) / 32 * 32;
	P.rpad = rpad;
	P.lvl_first.assign(lvl_start.begin(), lvl_start.begin() + nlev);
	P.label_of_row.assign((size_t) (r > 0 ? r : 1), 0);
	{
		std::vector<int> cursor(lvl_start.begin(), lvl_start.end());
		for (int k = 0; k < r; k++)
			P.label_of_row[k] = cursor[hmax - height[k]]++;
	}
	P.lvl_end.assign((size_t) (rpad > 0 ? rpad : 1), 0);
	P.lvl_end_w.assign((size_t) (rpad / 32 > 0 ? rpad / 32 : 1), 0);
	P.kof.assign((size_t) (rpad > 0 ? rpad : 1), -1);
	// labels in the gap before an aligned level belong to no row; give them the end of the gap
	{
		int prev_end = 0;
		std::vector<int> owners((size_t) (rpad / 32 > 0 ? rpad / 32 : 1), 0);    // levels touching each word
		for (int l = 0; l < nlev; l++) {
			for (int c = prev_end; c < lvl_start[l]; c++)
				P.lvl_end[c] = (uint32_t) lvl_start[l];
			for (int c = lvl_start[l]; c < lvl_start[l + 1]; c++)
				P.lvl_end[c] = (uint32_t) lvl_start[l + 1];
			if (lvl_count[l] > 0)
				for (int w = lvl_start[l] / 32; w <= (lvl_start[l + 1] - 1) / 32; w++)
					owners[w] += 1;
			prev_end = lvl_start[l + 1];
		}
		for (int c = prev_end; c < rpad; c++)
			P.lvl_end[c] = (uint32_t) rpad;
		for (int w = 0; w < rpad / 32; w++)
			P.lvl_end_w[w] = 0xFFFFFFFFu;                 // MIXED unless proven otherwise
		for (int l = 0; l < nlev; l++) {
			if (lvl_count[l] == 0)
				continue;
			// a level may claim its words when it starts on a word boundary and no other
			// level shares its last word
			const int w0 = lvl_start[l] / 32, w1 = (lvl_start[l + 1] - 1) / 32;
			if (lvl_start[l] % 32 != 0 || owners[w1] != 1)
				continue;
			for (int w = w0; w <= w1; w++)
				P.lvl_end_w[w] = (uint32_t) (w1 + 1);
		}
	}
	for (int k = 0; k < r; k++)
		P.kof[P.label_of_row[k]] = k;
	// column labels: pivotal columns get the label of their row, the others follow
	P.lab.assign((size_t) (m > 0 ? m : 1), 0);
	P.q.assign((size_t) (m - r > 0 ? m - r : 1), 0);
	{
		int np = 0;
		for (int j = 0; j < m; j++) {
			if (qinv[j] >= 0) {
				P.lab[j] = (uint32_t) P.label_of_row[qinv[j]];
			} else {
				P.lab[j] = (uint32_t) (rpad + np);
				P.q[np] = j;
				np += 1;
			}
		}
	}
	lap("levels + labels");
	// rows of U' in label order, pivot entry dropped, values * R mod p
	const i64 nnz = U->p[r] - r;
	// (v << 32) mod p without a 64-bit division per entry: q = floor(x * floor(2^64 / p) / 2^64) is the quotient or one less
	const uint64_t pu64 = (uint64_t) prime, barrett = ~0ull / pu64;
	auto shifted_mod = [&](uint64_t v) -> uint32_t {
		const uint64_t x = v << 32;
		const uint64_t qh = (uint64_t) (((unsigned __int128) x * barrett) >> 64);
		uint64_t rem = x - qh * pu64;
		while (rem >= pu64)
			rem -= pu64;
		return (uint32_t) rem;
	};
	P.rp.assign((size_t) rpad + 1, 0);
	P.ent.assign((size_t) (nnz > 0 ? nnz : 1), uint2{0, 0});
	const int T_fill = (r < 20000) ? 1 : std::max(1, std::min(16, usable_cpus()));
	auto in_threads = [&](auto &&body) {          // body(c0, c1) over equal ranges of the labels
		sh::pool_run(T_fill, [&](int t) { body((int) ((i64) rpad * t / T_fill), (int) ((i64) rpad * (t + 1) / T_fill)); });
	};
	{
		// (the lengths first -- a cache miss per label, by the threads --, then their running sum)
		in_threads([&](int c0, int c1) {
			for (int c = c0; c < c1; c++) {
				const int k = P.kof[c];
				P.rp[(size_t) c + 1] = (k >= 0) ? (uint64_t) (U->p[k + 1] - U->p[k] - 1) : 0;
			}
		});
		uint64_t w = 0;
		for (int c = 0; c < rpad; c++) {
			const uint64_t len = P.rp[(size_t) c + 1];
			P.rp[c] = w;
			w += len;
		}
		P.rp[rpad] = w;
	}
	lap("row pointers");
	// the first four entries of every row again, at a fixed place (label * 4): the row-group kernel
	// fetches them together with the accumulator line, without waiting for the row extent
	P.head.assign((size_t) (rpad > 0 ? rpad : 1) * 4, uint2{0xFFFFFFFFu, 0u});
	// The rows are gathered in label order -- a cache miss per row and per entry (the label of its column) -- by a few threads,
	// each a range of labels; the connected components below only need the labels and run beside them.
	std::atomic<int64_t> ndeps_all{0};
	auto fill = [&](int c0, int c1) {
		int64_t ndeps = 0;
		for (int c = c0; c < c1; c++) {
			// (a row of U per label: three cache misses a row -- its extent, its columns, its values -- and nothing else to do
			//  meanwhile: the rows sixteen and eight labels ahead are asked for now)
			if (c + 16 < c1 && P.kof[c + 16] >= 0)
				__builtin_prefetch(&U->p[P.kof[c + 16]]);
			if (c + 8 < c1 && P.kof[c + 8] >= 0) {
				const i64 pf = U->p[P.kof[c + 8]];
				__builtin_prefetch(&U->j[pf]);
				__builtin_prefetch(&U->x[pf]);
			}
			const int k = P.kof[c];
			if (k < 0)
				continue;
			uint64_t w = P.rp[c];
			for (i64 px = U->p[k] + 1; px < U->p[k + 1]; px++) {
				uint2 e;
				e.x = P.lab[U->j[px]];
				e.y = shifted_mod(zp_unsigned(prime, U->x[px]));
				ndeps += e.x < (uint32_t) rpad;
				P.ent[w++] = e;
			}
			const uint64_t len = P.rp[c + 1] - P.rp[c];
			for (uint64_t t = 0; t < len && t < 4; t++)
				P.head[(size_t) c * 4 + t] = P.ent[P.rp[c] + t];
		}
		ndeps_all += ndeps;
	};
	in_threads(fill);
	P.ndeps = ndeps_all.load();
	lap("entries + heads");

}

// What only the row-by-row kernels read, computed when they are first asked for (a quarter of the planning time, and the
// image paths never look at it): the connected components of the pivot graph -- rows of the matrix whose pivotal entries
// lie in different components share no elimination at all; the row-group kernel regroups rows by component when the order
// of the row list turns out to be unrelated to the structure -- and the largest number of rows of U' that hold one label
// (how many terms an accumulator may receive).  From the relabelled entries of the plan: no access to U.
// components = false: the column degrees only (what the per-row tiers need: the width of their sums) -- the connected components
// of the pivot graph are what the row-group kernel regroups its rows by, and a factor that only ever sees a density sample of
// 100 rows (every GL7d19-class round since the sparse image: mk15.b4) paid 10 ms of union-find for nothing
static void plan_row_tables(FactPlan &P, bool components = true)
{
	const int rpad = P.rpad;
	P.ncomp = 0;
	P.comp_largest = 0;
	P.maxdeg = 0;
	if (!components) {
		std::vector<int> deg((size_t) rpad + (size_t) (P.m - P.r) + 1, 0);
		const size_t count = (size_t) P.rp[rpad];
		for (size_t e = 0; e < count; e++)
			deg[P.ent[e].x] += 1;
		for (size_t t = 0; t < deg.size(); t++)
			P.maxdeg = std::max(P.maxdeg, deg[t]);
		return;
	}
	P.comp.assign((size_t) (rpad > 0 ? rpad : 1), 0);
	std::vector<uint32_t> parent((size_t) (rpad > 0 ? rpad : 1));
	for (int c = 0; c < rpad; c++)
		parent[c] = (uint32_t) c;
	auto find = [&](uint32_t x) {
		while (parent[x] != x) {
			parent[x] = parent[parent[x]];
			x = parent[x];
		}
		return x;
	};
	std::vector<int> deg((size_t) rpad + (size_t) (P.m - P.r) + 1, 0);
	for (int c = 0; c < rpad; c++) {
		if (P.kof[c] < 0)
			continue;
		for (uint64_t e = P.rp[c]; e < P.rp[c + 1]; e++) {
			const uint32_t t = P.ent[e].x;
			deg[t] += 1;
			if (t < (uint32_t) rpad) {
				const uint32_t a = find((uint32_t) c), b = find(t);
				if (a != b)
					parent[std::max(a, b)] = std::min(a, b);          // the root is the smallest label
			}
		}
	}
	for (size_t t = 0; t < deg.size(); t++)
		P.maxdeg = std::max(P.maxdeg, deg[t]);
	std::vector<int> members((size_t) (rpad > 0 ? rpad : 1), 0);
	for (int c = 0; c < rpad; c++) {
		P.comp[c] = find((uint32_t) c);
		if (P.kof[c] >= 0)
			members[P.comp[c]] += 1;
	}
	for (int c = 0; c < rpad; c++)
		if (members[c] > 0) {
			P.ncomp += 1;
			P.comp_largest = std::max(P.comp_largest, members[c]);
		}
}

// (tests) levels offered to plan_factor for this factor, as the device pivot search offers them: heights[k] per row of U
void spasm_hip_debug_level_hint(const struct spasm_csr *U, const int *heights)
{
	std::vector<int> h(heights, heights + U->n);
	sh::level_hint_set(U, U->n, std::move(h));
}

// CPU-only view of the plan, for tests: label of each row of U, end of the
// level of each label, label of each column.  Arrays sized r, r, m.
// label_of_row: r ints; lvl_end_of_row: r ints (end of the level of that row's label);
// lab: m ints; info[0] = rpad, info[1] = maxdeg.
int spasm_hip_debug_plan(const struct spasm_csr *U, const int *qinv, int *label_of_row, int *lvl_end_of_row, int *lab,
                         int *info)
{
	FactPlan P;
	plan_factor(U, qinv, P);
	for (int k = 0; k < P.r; k++) {
		label_of_row[k] = P.label_of_row[k];
		lvl_end_of_row[k] = (int) P.lvl_end[P.label_of_row[k]];
	}
	for (int j = 0; j < P.m; j++)
		lab[j] = (int) P.lab[j];
	info[0] = P.rpad;
	plan_row_tables(P);
	info[1] = P.maxdeg;
	return P.nlevels;
}

spasm_hip_dfact *spasm_hip_dfact_create(const struct spasm_csr *U, const int *qinv, void *stream_)
{
	hipStream_t stream = (hipStream_t) stream_;
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_dfact_create: no HIP device (this library has no CPU path)");
	const double t_begin = wtime();
	FactPlan P;
	plan_factor(U, qinv, P);
	const double t_planned = wtime();
	const int r = P.r, m = P.m, rpad = P.rpad;
	spasm_hip_dfact *F = new spasm_hip_dfact();
	F->m = m;
	F->r = r;
	F->rpad = rpad;
	F->Sm = m - r;
	F->prime = P.prime;
	F->mont = mont_setup(P.prime);
	F->nlevels = P.nlevels;
	F->nnz = (i64) P.rp[rpad];
	F->h_q = P.q;
	F->h_kof = P.kof;
	F->d_lab = dalloc<uint32_t>(m);
	F->d_q = dalloc<int>(m - r + 514);          // (padded: bs_apply_s16_kernel reads pairs over whole tiles of the padded row)
	F->d_rp = dalloc<uint64_t>(rpad + 1);
	F->d_ent = dalloc<uint2>(F->nnz);
	F->d_head = dalloc<uint2>((i64) rpad * 4);
	upload(F->d_head, P.head.data(), (i64) rpad * 4, stream);
	F->d_lvl_end = dalloc<uint32_t>(rpad);
	F->d_lvl_end_w = dalloc<uint32_t>(rpad / 32);
	F->d_kof = dalloc<int>(rpad);
	upload(F->d_lab, P.lab.data(), m, stream);
	upload(F->d_q, P.q.data(), m - r, stream);
	upload(F->d_rp, P.rp.data(), (i64) rpad + 1, stream);
	upload(F->d_ent, P.ent.data(), F->nnz, stream);
	upload(F->d_lvl_end, P.lvl_end.data(), rpad, stream);
	upload(F->d_lvl_end_w, P.lvl_end_w.data(), rpad / 32, stream);
	upload(F->d_kof, P.kof.data(), rpad, stream);
	// U' by target label and the label range of every level: what the pull variant of the row-group kernel reads (an
	// experiment, SPASM_HIP_PULL=1: the tables are only built for it -- a counting sort of U' on the host, 15 % of the
	// image's build time on mk13.b5)
	const bool want_pull = env_int("SPASM_HIP_PULL", 0) != 0;
	F->has_pull = want_pull;
	std::vector<uint64_t> cp(want_pull ? (size_t) rpad + (size_t) (m - r) + 1 : 1, 0);
	std::vector<uint2> cent((size_t) (want_pull && F->nnz > 0 ? F->nnz : 1));
	std::vector<int2> lvl((size_t) (want_pull && P.nlevels > 0 ? P.nlevels : 1));
	if (want_pull) {
		for (i64 e = 0; e < F->nnz; e++)
			cp[(size_t) P.ent[e].x + 1] += 1;
		for (size_t t = 0; t + 1 < cp.size(); t++)
			cp[t + 1] += cp[t];
		std::vector<uint64_t> cursor(cp.begin(), cp.end() - 1);
		for (int c = 0; c < rpad; c++)
			for (uint64_t e = P.rp[c]; e < P.rp[c + 1]; e++)
				cent[cursor[P.ent[e].x]++] = uint2{(uint32_t) c, P.ent[e].y};
		for (int l = 0; l < P.nlevels; l++)
			lvl[l] = int2{P.lvl_first[l], P.lvl_first[l] + P.lvl_count[l]};
	}
	F->d_cp = dalloc<uint64_t>((i64) cp.size());
	F->d_cent = dalloc<uint2>((i64) cent.size());
	F->d_lvl = dalloc<int2>((i64) lvl.size());
	upload(F->d_cp, cp.data(), (i64) cp.size(), stream);
	upload(F->d_cent, cent.data(), (i64) cent.size(), stream);
	upload(F->d_lvl, lvl.data(), (i64) lvl.size(), stream);
	HIP_CHECK(hipStreamSynchronize(stream));    // the host vectors die here
	const double t_uploaded = wtime();
	// back-substituted image (backsolve.hip): planned when the non-pivotal columns are few enough for dense rows
	// of R; R itself is computed by the first Schur complement that wants it
	// sparse image (sparse_image.hip): its dependency tables, for wide factors (R itself is built by the first batch that wants it)
	const bool plan_sparse = sparse_image_possible(F->prime) && r > 0 && m - r > 0 &&
	                         (env_int("SPASM_HIP_SPARSE_IMAGE", -1) == 1 || (m - r >= 8192 && (double) r * (double) (m - r) >= 5e8));
	// (large factors: by a thread of their own, started below once the plan stands where it will stay)
	const bool plan_sparse_async = plan_sparse && r >= 100000 && (1) != 0;
	if (plan_sparse && !plan_sparse_async)
		sparse_image_plan(P, F, stream);
	const double t_bs = wtime();
	if (verbose() >= 2 && F->sp.planned)
		logmsg("[factor image] tables of the sparse image: %.1f ms\n", 1e3 * (t_bs - t_uploaded));
	int64_t bs_bytes = 0;
	if (env_int("SPASM_HIP_BACKSOLVE", -1) != 0 && backsolve_eligible(r, m - r, F->nnz, &bs_bytes, F->prime)) {
		if (plan_sparse && env_int("SPASM_HIP_BACKSOLVE", -1) != 1 && (0) == 0) {
			// the plan of the dense image waits for a batch that wants it (backsolve_build); what the path choice reads is known now
			F->bs.r = r;
			F->bs.Sm = m - r;
			F->bs.ldR = ((int64_t) (m - r) + 511) / 512 * 512;
			F->bs.ndeps = P.ndeps;
			F->bs_deferred = true;
		} else {
			backsolve_plan(P, F, stream);
		}
	}
	if (verbose() >= 2)
		logmsg("[factor image] %d rows, %d levels: level schedule + relabelling %.1f ms, tables + upload %.1f ms, plan of the back-substitution %.1f ms\n",
		       r, F->nlevels, 1e3 * (t_planned - t_begin), 1e3 * (t_uploaded - t_planned), 1e3 * (wtime() - t_bs));
	// the host part of the image stays: the tables of the row-by-row kernels (ensure_row_tables) and a deferred plan of the dense
	// image are made from it when somebody asks
	F->host_plan = std::make_unique<FactPlan>(std::move(P));
	if (plan_sparse_async)
		sparse_image_plan_start(*F->host_plan, F);
	return F;
}

}  // extern "C"

namespace sh {
// components of the pivot graph + largest column degree, on first use by a row-by-row path (plan_row_tables)
void ensure_row_tables(const spasm_hip_dfact *F, hipStream_t stream, bool components)
{
	static std::mutex mutex;
	std::lock_guard<std::mutex> guard(mutex);
	if (F->row_tables || (!components && F->maxdeg > 0))
		return;
	if (!F->host_plan)
		die("ensure_row_tables: the factor image has lost its host plan");
	const double t0 = wtime();
	FactPlan &P = *F->host_plan;
	if (!components) {
		plan_row_tables(P, false);
		F->maxdeg = std::max(P.maxdeg, 1);
		if (verbose() >= 2)
			logmsg("[factor image] column degrees for the per-row kernels: %.1f ms\n", 1e3 * (wtime() - t0));
		return;
	}
	plan_row_tables(P);
	F->maxdeg = P.maxdeg;
	F->ncomp = P.ncomp;
	F->comp_largest = P.comp_largest;
	F->d_comp = dalloc<uint32_t>(F->rpad);
	upload(F->d_comp, P.comp.data(), (i64) F->rpad, stream);
	HIP_CHECK(hipStreamSynchronize(stream));
	F->row_tables = true;
	if (verbose() >= 2)
		logmsg("[factor image] tables of the row-by-row kernels (components of the pivot graph, column degrees): %.1f ms\n", 1e3 * (wtime() - t0));
}
}  // namespace sh

extern "C" {

void spasm_hip_dfact_destroy(spasm_hip_dfact *F)
{
	if (F == nullptr)
		return;
	backsolve_free(F);
	sparse_image_free(F);
	sh::big_free(F->d_lab);
	sh::big_free(F->d_q);
	sh::big_free(F->d_rp);
	sh::big_free(F->d_ent);
	sh::big_free(F->d_head);
	sh::big_free(F->d_comp);
	sh::big_free(F->d_lvl_end);
	sh::big_free(F->d_lvl_end_w);
	sh::big_free(F->d_kof);
	sh::big_free(F->d_cp);
	sh::big_free(F->d_cent);
	sh::big_free(F->d_lvl);
	delete F;
}

void spasm_hip_dfact_hint_density(spasm_hip_dfact *F, double density)
{
	if (F != nullptr)
		F->bs.density_hint = density;
}

void spasm_hip_dfact_hint_eliminations(spasm_hip_dfact *F, double per_row)
{
	if (F != nullptr)
		F->bs.elim_hint = per_row;
}

void spasm_hip_dfact_forget(spasm_hip_dfact *F)
{
	if (F != nullptr) {
		F->bs.valid = false;          // (the buffer stays: only the contents are forgotten)
		F->sp.valid = false;
	}
}

// what spasm_hip_echelonize does around a sparse round, for callers that time it (bench.py --gpus N): the cached images forget
// their derived state (R is rebuilt by the next call, as after a fresh factor) ...
void spasm_hip_forget_cached_images(void)
{
	std::lock_guard<std::mutex> guard(g_fact_mutex);
	for (int slot = 0; slot < 2; slot++)
		spasm_hip_dfact_forget(g_fact[slot]);
}

// fill of R as the sparse image holds it: out[0] entries, out[1] occupied 64-column tiles, out[2] non-empty
// fragments, out[3] (row, segment) pairs.  Returns 1 when the factor holds a valid sparse image, else 0 (out zeroed).
int spasm_hip_dfact_sparse_image_census(const spasm_hip_dfact *F, i64 *out, void *stream)
{
	sparse_image_census(F, out, (hipStream_t) stream);
	return F->sp.valid ? 1 : 0;
}

int spasm_hip_dfact_rank(const spasm_hip_dfact *F) { return F->r; }
int spasm_hip_dfact_levels(const spasm_hip_dfact *F) { return F->nlevels; }
i64 spasm_hip_dfact_nnz(const spasm_hip_dfact *F) { return F->nnz; }

// --------------------------------------------------------------------------
// workspace
// --------------------------------------------------------------------------
spasm_hip_dwork *spasm_hip_dwork_create(int max_rows, int m, i64 pool_entries)
{
	spasm_hip_dwork *W = new spasm_hip_dwork();
	W->max_rows = max_rows;
	W->m = m;
	W->pool_cap = pool_entries;
	const size_t pool_bytes = (size_t) (pool_entries > 0 ? pool_entries : 1) * sizeof(int);
	W->d_pool_j = (int *) big_alloc(pool_bytes);
	W->d_pool_x = (int *) big_alloc(pool_bytes);
	W->d_Sj = (int *) big_alloc(pool_bytes);
	W->d_Sx = (int *) big_alloc(pool_bytes);
	W->d_row_off = dalloc<int64_t>(max_rows);
	W->d_row_len = dalloc<int>(max_rows);
	W->d_ovf1 = dalloc<int>(max_rows);
	W->d_ovf2 = dalloc<int>(max_rows);
	W->d_Sp = dalloc<int64_t>((i64) max_rows + 1);
	W->d_blocksum = dalloc<int64_t>((max_rows + 1023) / 1024 + 1);
	W->d_ctr = dalloc<int>(CTR_COUNT);
	W->d_ctr64 = dalloc<unsigned long long>(C64_COUNT);
	for (int e = 0; e < 7; e++)
		HIP_CHECK(hipEventCreate(&W->ev[e]));
	return W;
}

void spasm_hip_dwork_destroy(spasm_hip_dwork *W)
{
	if (W == nullptr)
		return;
	big_free(W->d_pool_j);
	big_free(W->d_pool_x);
	big_free(W->d_Sj);
	big_free(W->d_Sx);
	sh::big_free(W->d_row_off);
	sh::big_free(W->d_row_len);
	sh::big_free(W->d_ovf1);
	sh::big_free(W->d_ovf2);
	sh::big_free(W->d_Sp);
	sh::big_free(W->d_blocksum);
	if (W->d_lb_status != nullptr)
		sh::big_free(W->d_lb_status);
	if (W->d_stage != nullptr)
		sh::big_free(W->d_stage);
	if (W->d_spT != nullptr)
		sh::big_free(W->d_spT);
	if (W->d_order != nullptr)
		sh::big_free(W->d_order);
	if (W->d_sortbuf != nullptr)
		sh::big_free(W->d_sortbuf);
	sh::big_free(W->d_ctr);
	sh::big_free(W->d_ctr64);
	sh::big_free(W->d_scratch);
	for (int e = 0; e < 7; e++)
		if (W->ev[e] != nullptr)
			(void) hipEventDestroy(W->ev[e]);
	delete W;
}

// --------------------------------------------------------------------------
// device Schur complement: three tiers (small LDS table, large LDS table,
// dense accumulator in HBM), then row pointers + gather/sort.
// --------------------------------------------------------------------------
}  // extern "C"

namespace {
int dschur_impl(const spasm_hip_dcsr *A, const int *d_rows, int nrows, const spasm_hip_dfact *F, spasm_hip_dwork *W,
                void *stream_, spasm_hip_schur_stats *stats, LOut *Lout)
{
	hipStream_t stream = (hipStream_t) stream_;
	const double t_enter = wtime();
	if (nrows > W->max_rows)
		die("spasm_hip_dschur: %d rows but the workspace was sized for %d", nrows, W->max_rows);
	if (A->m != F->m || W->m < F->m)
		die("spasm_hip_dschur: column count mismatch (A %d, factor %d, workspace %d)", A->m, F->m, W->m);
	// lazy 32-bit sums in the LDS tables: at most 6144 + 1 terms below p each
	const bool wide_lds = ((double) F->prime * 6146.0 >= 4294967296.0);
	// ... and in the dense accumulators: a column receives at most maxdeg + 1 terms, each below 2p (the
	// row-group kernel adds unreduced products)
	bool wide_dense = false;          // (set below, once it is known that a row-by-row path runs: it needs the tables of ensure_row_tables)
	const int sort_rows = (1);
	const int small_table = 1024, big_table = 8192;
	const int cus = cu_count();
	// tests: 1 = start at the large LDS table, 2 = dense accumulators only.  The large table is otherwise
	// skipped (one wave per CU: slower than the dense tier) unless SPASM_HIP_USE_BIG_TABLE=1.
	int force_tier = env_int("SPASM_HIP_FORCE_TIER", 0);
	if (Lout != nullptr)
		force_tier = 2;         // rows must not be restarted once coefficients have been recorded: no LDS tiers
	const bool use_big = (force_tier == 1) || (0);

	// row-group kernel (64 consecutive rows per wave, label-major state) for every row: default for
	// batches large enough to fill the GPU with groups; SPASM_HIP_GROUP=0/1 forces the choice.
	// Small batches (density samples, dense blocks) stay on the per-row tiers.
	int group_mode = env_int("SPASM_HIP_GROUP", -1);
	bool probe = false;            // auto: run a few groups first and look at their lane efficiency
	if (group_mode < 0) {
		group_mode = (nrows >= 64 * 32 && (force_tier == 0 || Lout != nullptr)) ? 1 : 0;
		probe = group_mode && Lout == nullptr && nrows >= env_int("SPASM_HIP_GROUP_WATCH_ROWS", 0);
	}
	int group_slots = 0, group_waves = 1;
	i64 group_slot_bytes = 0, group_off_bm = 0;
	// S = A_n - A_p R from the back-substituted image (backsolve.hip) when the factor has one: no accumulator scratch
	const bool other_forced = env_int("SPASM_HIP_FORCE_TIER", 0) != 0 || env_int("SPASM_HIP_GROUP", -1) >= 0;
	// S = A_n - A_p R from the SPARSE image (sparse_image.hip) when the Schur complement is expected to stay sparse; the image
	// is built on first use, and a build that finds R dense gives up: the other paths then take the batch
	bool want_sp = nrows > 0 && Lout == nullptr && sparse_image_wanted(F, other_forced, nrows);
	bool built_sp = false;
	if (want_sp && !F->sp.valid) {
		built_sp = sparse_image_build(F, stream);
		want_sp = built_sp;
	}
	const bool want_bs = !want_sp && nrows > 0 && Lout == nullptr && backsolve_wanted(F, other_forced, nrows);
	if (!want_bs && !want_sp) {
		// (the components of the pivot graph are what the row-GROUP kernel regroups its rows by: a density sample of 100 rows on the
		//  per-row tiers only needs the column degrees -- 10 ms of union-find on mk15.b4's factor that nothing ever read)
		ensure_row_tables(F, stream, group_mode != 0 || (1) == 0);
		wide_dense = (2.0 * (double) F->prime * ((double) F->maxdeg + 3.0) >= 4294967296.0);
	}
	// per-wave dense scratch, (re)allocated when the factor geometry needs more
	if (!want_bs && !want_sp) {
		i64 slot_bytes, off_bm, off_xn;
		wave_dense_geometry(F->rpad, F->Sm, wide_dense, &slot_bytes, &off_bm, &off_xn);
		int slots = (cus * 32);
		// accumulator slices may take up to half of the free HBM (288 GB parts: be generous), or what
		// SPASM_HIP_SCRATCH_GB says
		size_t free_b = 0, total_b = 0;
		sh::mem_info(&free_b, &total_b);
		i64 budget = (i64) ((free_b + (size_t) W->scratch_bytes) / 2);
		if (W->scratch_budget > 0)
			budget = std::min(budget, std::max(W->scratch_budget, W->scratch_bytes));
		if (env_int("SPASM_HIP_SCRATCH_GB", 0) > 0)
			budget = (i64) env_int("SPASM_HIP_SCRATCH_GB", 0) << 30;
		slots = (int) std::max<i64>(cus, std::min<i64>(slots, budget / slot_bytes));
		slots = std::max(1, std::min(slots, nrows));
		i64 need = slot_bytes * slots;
		if (group_mode) {
			group_geometry(F->rpad, F->Sm, wide_dense, &group_slot_bytes, &group_off_bm);
			const int ngroups = (nrows + 63) / 64;
			// a slice costs (rpad + Sm) * 256 B (512 B with 64-bit sums): on very wide matrices the budget holds
			// fewer slices than there are CUs (or none) and the chip would idle -- the per-row tiers need 4 B per
			// label per wave and run at full occupancy, so they take such batches
			const i64 slices_that_fit = budget / group_slot_bytes;
			if (slices_that_fit < 1 || (slices_that_fit < std::min<i64>(ngroups, cus / 2) && env_int("SPASM_HIP_GROUP", -1) < 0)) {
				group_mode = 0;
				probe = false;
			}
		}
		if (group_mode) {
			const int ngroups = (nrows + 63) / 64;
			// four (two) waves per group while there are at most 3 (12) groups per CU: with few groups the run time is
			// the chain of level rounds of one group, which the waves split between them (tools/probe_groups.py)
			// (slots = workgroups = accumulator slices: as many as are resident at two waves per SIMD; the others
			//  would only wait for a CU and find the queue of groups empty)
			// ... and the same when it is the BUDGET that keeps the groups in flight few (wide factors: a slice of mk14.b4 is
			// 80 MB, 297 of them fit the 24 GB of a one-shot call -- one wave each would leave the chip at one wave per CU:
			// 641 ms against 334 ms with four waves per group)
			const i64 in_flight = std::min<i64>(ngroups, budget / group_slot_bytes);
			group_waves = env_int("SPASM_HIP_GROUP_WAVES", in_flight <= cus * 3 ? 4 : in_flight <= cus * 12 ? 2 : 1);
			group_slots = (int) std::min<i64>((group_waves >= 4 ? cus * 2 : group_waves >= 2 ? cus * 4 : cus * 8),
			                                  budget / group_slot_bytes);
			group_slots = std::max(1, std::min(group_slots, ngroups));
			// with the automatic fallback the per-row tier may run in the same buffer afterwards
			need = probe ? std::max(need, group_slot_bytes * group_slots) : group_slot_bytes * group_slots;
		}
		if (need > W->scratch_bytes) {
			if (W->d_scratch != nullptr)
				sh::big_free(W->d_scratch);
			HIP_CHECK(sh::malloc_or_trim((void **) &W->d_scratch, (size_t) need));
			W->scratch_bytes = need;
			HIP_CHECK(hipMemsetAsync(W->d_scratch, 0, (size_t) need, stream));
		} else if (slot_bytes != W->slot_bytes || off_bm != W->off_bm || off_xn != W->off_xn) {
			// same buffer, other layout: it is all zero anyway (the kernels restore that invariant)
		}
		W->scratch_slots = slots;
		W->slot_bytes = slot_bytes;
		W->off_bm = off_bm;
		W->off_xn = off_xn;
	}

	HIP_CHECK(hipMemsetAsync(W->d_ctr, 0, CTR_COUNT * sizeof(int), stream));
	HIP_CHECK(hipMemsetAsync(W->d_ctr64, 0, C64_COUNT * sizeof(unsigned long long), stream));
	HIP_CHECK(hipEventRecord(W->ev[0], stream));

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
	a.comp = F->d_comp;
	a.lvl_end = F->d_lvl_end;
	a.lvl_end_w = F->d_lvl_end_w;
	a.r = F->rpad;              // kernels only see the padded label space
	a.Sm = F->m - F->r;
	a.m = F->m;
	a.F = to_dev(F->mont);
	a.pool_j = W->d_pool_j;
	a.pool_x = W->d_pool_x;
	a.pool_cap = W->pool_cap;
	a.row_off = W->d_row_off;
	a.row_len = W->d_row_len;
	a.ctr = W->d_ctr;
	a.ctr64 = W->d_ctr64;
	if (Lout != nullptr) {
		a.L_i = Lout->Li;
		a.L_j = Lout->Lj;
		a.L_x = Lout->Lx;
		a.L_cap = Lout->cap;
		a.kof = F->d_kof;
		a.row_orig = Lout->row_orig;
	}

	bool used_bs = false, built_bs = false, bs_direct = false, use_pull = false;
	int bs_staged_slices = 0;          // staged output of the back-substituted path: slices it ran in (0: not used)
	if (want_sp) {
		group_mode = 0;
		HIP_CHECK(hipEventRecord(W->ev[5], stream));
		a.list = nullptr;
		a.list_count = nullptr;
		a.done_ctr = CTR_DONE2;
		bs_direct = true;          // rows land in W->d_Sj / d_Sx in their final order: no gather pass from the pool
		const i64 twords = sparse_image_table_words(nrows, F->sp.nseg);
		if (W->spT_words < twords) {
			if (W->d_spT != nullptr)
				sh::big_free(W->d_spT);
			W->d_spT = dalloc<uint64_t>(twords);
			W->spT_words = twords;
		}
		if (W->d_lb_status == nullptr)
			W->d_lb_status = dalloc<unsigned long long>((i64) W->max_rows + 16 + 16 * 16);
		// the fragments of S go to the row pool of the workspace (4 bytes an entry: pool_cap entries fit pool_j)
		launch_sparse_image_apply(a, F, reinterpret_cast<uint32_t *>(W->d_pool_j), reinterpret_cast<uint32_t *>(W->d_pool_x), W->pool_cap, W->d_spT, W->d_lb_status, W->d_Sp, W->d_Sj,
		                          W->d_Sx, W->pool_cap, stream, W->ev[6]);
		HIP_CHECK(hipEventRecord(W->ev[3], stream));
		HIP_CHECK(hipEventRecord(W->ev[4], stream));
		goto eliminated;
	}
	if (want_bs) {
		// R is built on first use
		used_bs = true;
		group_mode = 0;
		if (!F->bs.valid) {
			backsolve_build(F, stream);
			built_bs = true;
		}
		HIP_CHECK(hipEventRecord(W->ev[5], stream));
		a.list = nullptr;
		a.list_count = nullptr;
		a.done_ctr = CTR_DONE2;
		// rows straight into W->d_Sj / d_Sx in their final order (offsets by look-back): no pool, no gather pass
		bs_direct = env_int("SPASM_HIP_BS_DIRECT", 1) != 0;
		if (bs_direct) {
			if (W->d_lb_status == nullptr)
				W->d_lb_status = dalloc<unsigned long long>((i64) W->max_rows + 16 + 16 * 16);
			// status words of the rows, then (on a fresh 128-byte line) the 16 ticket counters, one line each
			const size_t ticket_at = ((size_t) nrows + 16) / 16 * 16;
			HIP_CHECK(hipMemsetAsync(W->d_lb_status, 0, (ticket_at + 16 * 16) * sizeof(unsigned long long), stream));
			BsDirectOut out{W->d_lb_status, reinterpret_cast<int *>(W->d_lb_status + ticket_at), W->d_Sp, W->d_Sj, W->d_Sx, W->pool_cap};
			int64_t stage_row_bytes = 0;
			if (backsolve_stages_output(F, &stage_row_bytes) && nrows > 0) {
				// packed rows of the staged output: the whole batch when it fits SPASM_HIP_STAGE_GB (default 8), else slices
				const int64_t budget = (int64_t) env_int("SPASM_HIP_STAGE_GB", 8) << 30;
				int64_t rows_fit = std::max<int64_t>(1024, budget / stage_row_bytes);
				if (env_int("SPASM_HIP_STAGE_ROWS", 0) > 0)          // (tests: slices on small inputs)
					rows_fit = env_int("SPASM_HIP_STAGE_ROWS", 0);
				out.stage_rows = std::min<int64_t>(nrows, rows_fit);
				const int64_t need = out.stage_rows * stage_row_bytes;
				if (W->stage_bytes < need) {
					if (W->d_stage != nullptr)
						sh::big_free(W->d_stage);
					W->d_stage = dalloc<uint32_t>(need / 4);
					W->stage_bytes = need;
				}
				out.stage = W->d_stage;
				out.ev_expand = W->ev[6];
			}
			launch_backsolve_apply(a, F, nullptr, 0, stream, &out);
			bs_staged_slices = out.staged ? out.slices : 0;
		} else {
			launch_backsolve_apply(a, F, nullptr, 0, stream, nullptr);
		}
		HIP_CHECK(hipEventRecord(W->ev[3], stream));
		HIP_CHECK(hipEventRecord(W->ev[4], stream));
		goto eliminated;
	}
	if (nrows > 0) {
		// tier 0: small LDS table, many waves per CU
		a.list = nullptr;
		a.list_count = nullptr;
		a.ovf_list = W->d_ovf1;
		a.next_ctr = CTR_ROW_NEXT;
		a.ovf_ctr = CTR_OVF1;
		a.done_ctr = CTR_DONE0;
		const int per_cu0 = (int) std::min<size_t>(16, (size_t) (160 * 1024) / schur_lds_bytes(small_table, wide_lds));
		int blocks0 = std::min(cus * per_cu0, (nrows + 3) / 4);
		if (group_mode) {
			a.next_ctr = CTR_ROW_NEXT_G;
			a.done_ctr = CTR_DONE2;
			// auto mode: the kernel watches its own lane efficiency (eliminations / (64 * applied pivots)).
			// The break-even against the per-row kernel is near 0.15 (one coalesced atomic instruction per
			// pivot entry at 2.6 G/s against one scattered update per row at 23 G/s),
			// but the efficiency of a healthy batch starts low (the first eliminations of a row are its
			// own, the shared part of the reach comes later: mk13.b5 is under 0.15 for its first 10 ms),
			// so the kernel only bails out of hopeless batches: under 0.04 after 500,000 applied pivots.
			// Abandoned rows keep row_len == -1 and go to the per-row tiers.
			HIP_CHECK(hipMemsetAsync(W->d_row_len, 0xFF, (size_t) nrows * sizeof(int), stream));
			const float min_eff = (float) (4) / 100.0f;
			// (... per group in flight: the judgement used to fall after 500,000 pivots whatever the number of groups, i.e. after
			// 1,700 pivots of each of the 297 groups of a mk14.b4 call -- all of them still in the cheap private start of their
			// rows -- and sent a batch whose final efficiency is 0.60 to the per-row tier: 1.68 s instead of 0.35 s)
			// (SPASM_HIP_GROUP_MIN_PIVOTS, when set, is the threshold as it stands: tests judge early with it)
			const char *min_w_env = sh::env_get("SPASM_HIP_GROUP_MIN_PIVOTS");
			const long long min_w = (min_w_env != nullptr) ? std::atoll(min_w_env) : std::max<long long>(500000, 8192ll * group_slots);
			// several connected components in the pivot graph: rows of different components share nothing, so the
			// rows are grouped by component from the start (the order of the list is kept inside a component; with a
			// single or a dominant component -- mk13.b5: 109,966 of its 111,177 pivots -- nothing is done)
			const bool regroup_enabled = env_int("SPASM_HIP_GROUP_REGROUP", 1) != 0;
			auto regroup = [&]() {
				const int64_t need = regroup_scratch_ints(nrows, F->rpad);
				if (W->sortbuf_ints < need) {
					if (W->d_sortbuf != nullptr)
						sh::big_free(W->d_sortbuf);
					W->d_sortbuf = dalloc<int>(need);
					W->sortbuf_ints = need;
				}
				if (W->d_order == nullptr)
					W->d_order = dalloc<int>(W->max_rows);
				launch_regroup_rows(a, W->d_sortbuf, W->d_order, stream);
			};
			const bool grouped_first = regroup_enabled && F->ncomp > 1 && Lout == nullptr &&
			                           (i64) F->comp_largest * 10 < (i64) F->r * 9;          // (a giant component: the list order is what matters)
			if (grouped_first) {
				regroup();
				a.order = W->d_order;
			}
			use_pull = Lout == nullptr && env_int("SPASM_HIP_PULL", 0) != 0 && F->has_pull && pull_lds_bytes(F->rpad, F->Sm) <= (size_t) 150 * 1024;
			if (use_pull) {
				// left-looking numeric pass, no atomics (schur_pull.hip); its slices are the row-group kernel's without the bitmap
				const int per_cu = (int) std::max<size_t>(1, std::min<size_t>(8, (size_t) (160 * 1024) / (pull_lds_bytes(F->rpad, F->Sm) + 5 * 1024)));
				const int pull_slots = (int) std::max<i64>(1, std::min<i64>(std::min<i64>((nrows + 63) / 64, (i64) cus * per_cu),
				                                                          W->scratch_bytes / pull_slot_bytes(F->rpad, F->Sm)));
				probe = false;
				launch_schur_pull(a, W->d_scratch, pull_slot_bytes(F->rpad, F->Sm), F->d_cp, F->d_cent, F->d_lvl, F->nlevels, pull_slots, stream);
			} else
			launch_schur_group(a, W->d_scratch, group_slot_bytes, group_off_bm, wide_dense, nullptr, 0, group_slots, stream,
			                   probe ? 1 : 0, min_eff, min_w, group_waves);
			a.order = nullptr;
			a.skip_ctr = CTR_GROUP_ABORT;
			// did it give up?  (one small read-back: the call blocks at its end anyway, and nothing is launched
			// for nothing -- the kernels below also check the flags themselves)
			int gave_up = 0;
			if (probe) {
				HIP_CHECK(hipMemcpyAsync(&gave_up, W->d_ctr + CTR_GROUP_ABORT, sizeof(int), hipMemcpyDeviceToHost, stream));
				HIP_CHECK(hipStreamSynchronize(stream));
			}
			HIP_CHECK(hipEventRecord(W->ev[5], stream));
			if (!probe || !gave_up) {
				HIP_CHECK(hipEventRecord(W->ev[3], stream));
				HIP_CHECK(hipEventRecord(W->ev[4], stream));
				goto eliminated;
			}
			a.skip_done = 1;
			a.next_ctr = CTR_ROW_NEXT;
			a.done_ctr = CTR_DONE0;
		}
		if (force_tier == 0)
			launch_schur_lds(a, small_table, wide_lds, std::max(blocks0, 1), stream);
		else
			launch_all_rows_to_list(W->d_ovf1, W->d_ctr + CTR_OVF1, W->d_row_len, nrows, stream);
		HIP_CHECK(hipEventRecord(W->ev[3], stream));
		// tier 1 (optional): large LDS table, one wave per CU
		const int *last_list = W->d_ovf1;
		const int *last_count = W->d_ctr + CTR_OVF1;
		if (use_big) {
			a.list = W->d_ovf1;
			a.list_count = W->d_ctr + CTR_OVF1;
			a.ovf_list = W->d_ovf2;
			a.next_ctr = CTR_ROW_NEXT2;
			a.ovf_ctr = CTR_OVF2;
			a.done_ctr = CTR_DONE1;
			launch_schur_lds(a, big_table, wide_lds, cus, stream);
			last_list = W->d_ovf2;
			last_count = W->d_ctr + CTR_OVF2;
		}
		HIP_CHECK(hipEventRecord(W->ev[4], stream));
		// tier 2: dense accumulators in HBM, one wave per row, thousands of rows in flight
		a.list = last_list;
		a.list_count = last_count;
		a.ovf_list = nullptr;
		a.next_ctr = CTR_ROW_NEXT3;
		a.ovf_ctr = CTR_OVF2;
		a.done_ctr = CTR_DONE2;
		launch_schur_wave_dense(a, W->d_scratch, W->slot_bytes, W->off_bm, W->off_xn, wide_dense, nullptr, 0,
		                        W->scratch_slots, stream);
	}
eliminated:
	HIP_CHECK(hipEventRecord(W->ev[1], stream));
	if (!bs_direct)
		launch_finalize(W, nrows, sort_rows, stream);
	HIP_CHECK(hipEventRecord(W->ev[2], stream));

	int ctr[CTR_COUNT];
	unsigned long long ctr64[C64_COUNT];
	i64 total = 0;
	HIP_CHECK(hipMemcpyAsync(ctr, W->d_ctr, sizeof(ctr), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipMemcpyAsync(ctr64, W->d_ctr64, sizeof(ctr64), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipMemcpyAsync(&total, W->d_Sp + nrows, sizeof(i64), hipMemcpyDeviceToHost, stream));
	const double t_enqueued = wtime();
	HIP_CHECK(hipStreamSynchronize(stream));
	if (verbose() >= 3)
		logmsg("[schur/hip] %d rows: %.2f ms of host work up to the last launch, %.2f ms waiting for the device\n", nrows, 1e3 * (t_enqueued - t_enter),
		       1e3 * (wtime() - t_enqueued));
	W->last_rows = nrows;
	W->last_nnz = total;
	if (ctr[CTR_STATUS] & 4)
		die("spasm_hip_dschur: a row waited for its predecessors' lengths for too long (look-back of the direct output)");
	const int status = ctr[CTR_STATUS] & 3;      // bit 0: row pool exhausted, bit 1: L pool exhausted
#ifdef SPASM_GROUP_PROFILE
	if (group_mode) {
		static const char *const phase[8] = {"drain", "watch", "bitmap scan", "level end", "gather pending", "apply+scatter", "output count", "output write"};
		const double g = (double) ((nrows + 63) / 64);
		for (int q = 0; q < 8; q++)
			fprintf(stderr, "[spasm_hip profile] %-15s %12.0f per group\n", phase[q], (double) ctr64[C64_PROF0 + q] / g);
	}
#endif
	if (Lout != nullptr)
		Lout->used = (i64) ctr64[C64_LPOOL];
	if (stats != nullptr) {
		stats->nnz = total;
		stats->eliminations = (i64) ctr64[C64_ELIM];
		stats->entries_streamed = (i64) ctr64[C64_STREAM];
		stats->input_entries = (i64) ctr64[C64_INPUT];
		stats->group_pivots = (i64) ctr64[C64_WAVEPIV];
		stats->used_group_kernel = group_mode ? 1 : 0;
		stats->group_slots = group_mode ? group_slots : 0;
		stats->group_waves = group_mode ? group_waves : 0;
		stats->group_slot_bytes = group_mode ? group_slot_bytes : 0;
		stats->group_slots_wanted = group_mode ? (int) std::min<i64>((nrows + 63) / 64, (i64) cus * 8 / std::max(1, group_waves)) : 0;
		stats->rows = nrows;
		stats->rows_lds = ctr[CTR_DONE0];
		stats->rows_lds_big = ctr[CTR_DONE1];
		stats->rows_dense = ctr[CTR_DONE2];
		stats->status = status;
		HIP_CHECK(hipEventElapsedTime(&stats->ms_eliminate, W->ev[0], W->ev[1]));
		stats->ms_tier0 = stats->ms_tier1 = stats->ms_tier2 = stats->ms_group = 0.0f;
		stats->group_aborted = ctr[CTR_GROUP_ABORT] ? 1 : 0;
		stats->used_backsolve = used_bs ? 1 : 0;
		stats->backsolve_built = built_bs ? 1 : 0;
		stats->ms_backsolve = stats->ms_apply = 0.0f;
		stats->bytes_backsolve = stats->bytes_apply = 0;
		stats->kernel[0] = 0;
		stats->kernel_other[0] = 0;
		stats->kernel_expand[0] = 0;
		stats->ms_expand = stats->ms_pad = 0.0f;
		stats->bytes_expand = stats->bytes_staged = 0;
		stats->used_sparse_image = want_sp ? 1 : 0;
		stats->sparse_image_built = built_sp ? 1 : 0;
		stats->ms_sparse_build = stats->ms_sparse_apply = stats->ms_sparse_gather = 0.0f;
		stats->sparse_image_nnz = stats->sparse_image_ops_build = stats->sparse_image_ops_apply = 0;
		stats->sparse_image_levels = stats->sparse_image_launches = 0;
		stats->bytes_sparse_build = stats->bytes_sparse_apply = stats->bytes_sparse_gather = 0;
		if (want_sp) {
			const SpImage &P = F->sp;
			if (built_sp)
				HIP_CHECK(hipEventElapsedTime(&stats->ms_sparse_build, P.ev0, P.ev1));
			HIP_CHECK(hipEventElapsedTime(&stats->ms_sparse_apply, W->ev[5], W->ev[6]));
			HIP_CHECK(hipEventElapsedTime(&stats->ms_sparse_gather, W->ev[6], W->ev[1]));
			stats->sparse_image_nnz = P.nnz;
			stats->sparse_image_ops_build = P.ops_build;
			stats->sparse_image_ops_apply = (i64) ctr64[C64_STREAM];
			stats->sparse_image_levels = P.nlevels;
			stats->sparse_image_launches = P.launches;
			// algorithmic bytes.  Build: every fragment entry read once per row that uses it and written
			// once (4 B each), the fragment words (8 B per dependency and segment + 8 B per row and segment), the entries of U'.
			// Rows of S: the fragment entries read (4 B), a fragment word per pivotal entry and segment, the entries in and the
			// fragments of S out (4 B); gather: fragments in (4 B), pairs out (8 B), 8 B per (row, segment).
			stats->bytes_sparse_build = 4 * (P.ops_build + P.nnz) + 8 * (P.ndeps + (i64) P.r) * P.nseg + 8 * F->nnz;
			stats->bytes_sparse_apply = 4 * (i64) ctr64[C64_STREAM] + 8 * (i64) ctr64[C64_ELIM] * P.nseg + 8 * (i64) ctr64[C64_INPUT] + 4 * total +
			                            8 * (i64) nrows * P.nseg;
			stats->bytes_sparse_gather = 12 * total + 8 * (i64) nrows * P.nseg + 8 * (i64) nrows;
			const bool build_dominates = stats->ms_sparse_build > stats->ms_sparse_apply;
			snprintf(stats->kernel, sizeof(stats->kernel), "%s", build_dominates ? "sp_build_kernel" : "sp_apply_kernel");
			snprintf(stats->kernel_other, sizeof(stats->kernel_other), "%s", build_dominates ? "sp_apply_kernel" : "sp_build_kernel");
			snprintf(stats->kernel_expand, sizeof(stats->kernel_expand), "sp_gather_kernel");
		} else if (used_bs) {
			const BsImage &B = F->bs;
			if (built_bs) {
				HIP_CHECK(hipEventElapsedTime(&stats->ms_backsolve, B.ev0, B.ev1));
				// every row of R written once and read once per dependency; the entries of U' read once
				stats->bytes_backsolve = ((i64) B.r + B.ndeps) * (i64) B.Sm * B.elem_bytes + 8 * F->nnz;
			}
			HIP_CHECK(hipEventElapsedTime(&stats->ms_apply, W->ev[5], W->ev[1]));
			// one row of R per pivotal entry of the reduced rows, the entries in and out, 20 B per row
			stats->bytes_apply = (i64) ctr64[C64_ELIM] * (i64) B.Sm * B.elem_bytes + 8 * ((i64) ctr64[C64_INPUT] + total) + 20 * (i64) nrows;
			if (bs_staged_slices > 0) {
				// staged output: the entries of S leave through bs_expand_s16_kernel; with one slice the two kernels are timed apart
				stats->bytes_apply -= 8 * total;
				stats->bytes_expand = 8 * total;
				stats->bytes_staged = (i64) nrows * B.ldR * B.elem_bytes;
				snprintf(stats->kernel_expand, sizeof(stats->kernel_expand), "bs_expand_kernel<%d>", B.sgn ? 0 : B.elem_bytes == 2 ? 1 : 2);
				if (bs_staged_slices == 1) {
					HIP_CHECK(hipEventElapsedTime(&stats->ms_apply, W->ev[5], W->ev[6]));
					HIP_CHECK(hipEventElapsedTime(&stats->ms_expand, W->ev[6], W->ev[1]));
				}
			}
			char apply_name[32];
			if (B.sgn)
				snprintf(apply_name, sizeof(apply_name), "bs_apply_s16_kernel");
			else
				snprintf(apply_name, sizeof(apply_name), "bs_apply_kernel<%s,%s>", B.elem_bytes == 2 ? "true" : "false", B.plain ? "true" : "false");
			const bool build_dominates = stats->ms_backsolve > stats->ms_apply;
			snprintf(stats->kernel, sizeof(stats->kernel), "%s", build_dominates ? B.kernel_build : apply_name);
			snprintf(stats->kernel_other, sizeof(stats->kernel_other), "%s", build_dominates ? apply_name : B.kernel_build);
		} else if (group_mode && use_pull) {
			snprintf(stats->kernel, sizeof(stats->kernel), "schur_pull_kernel");
		} else if (group_mode && !stats->group_aborted) {
			schur_group_variant_name(F->rpad, wide_dense, group_waves, stats->kernel, sizeof(stats->kernel));
		} else {
			snprintf(stats->kernel, sizeof(stats->kernel), "schur_wave_dense_kernel<%s>", wide_dense ? "true" : "false");
		}
		if (nrows > 0 && !used_bs && !want_sp) {
			if (group_mode)
				HIP_CHECK(hipEventElapsedTime(&stats->ms_group, W->ev[0], W->ev[5]));
			HIP_CHECK(hipEventElapsedTime(&stats->ms_tier0, group_mode ? W->ev[5] : W->ev[0], W->ev[3]));
			HIP_CHECK(hipEventElapsedTime(&stats->ms_tier1, W->ev[3], W->ev[4]));
			HIP_CHECK(hipEventElapsedTime(&stats->ms_tier2, W->ev[4], W->ev[1]));
		}
		HIP_CHECK(hipEventElapsedTime(&stats->ms_finalize, W->ev[1], W->ev[2]));
		HIP_CHECK(hipEventElapsedTime(&stats->ms_total, W->ev[0], W->ev[2]));
		if (built_sp) {          // (the sparse image is built before the events of the call start: its time belongs to the call)
			stats->ms_total += stats->ms_sparse_build;
			stats->ms_eliminate += stats->ms_sparse_build;
		}
	}
	return status;
}
}  // namespace

extern "C" {

int spasm_hip_dschur(const spasm_hip_dcsr *A, const int *d_rows, int nrows, const spasm_hip_dfact *F,
                     spasm_hip_dwork *W, void *stream, spasm_hip_schur_stats *stats)
{
	return dschur_impl(A, d_rows, nrows, F, W, stream, stats, nullptr);
}

// only the row pointers of the last result (rows + 1 int64): what the ranks of a column split exchange (lengths)
void spasm_hip_dschur_row_pointers(const spasm_hip_dwork *W, i64 *d_Sp, void *stream_)
{
	hipStream_t stream = (hipStream_t) stream_;
	HIP_CHECK(hipMemcpyAsync(d_Sp, W->d_Sp, ((size_t) W->last_rows + 1) * sizeof(i64), hipMemcpyDefault, stream));
}

void spasm_hip_dschur_fetch(const spasm_hip_dwork *W, i64 *d_Sp, int *d_Sj, spasm_ZZp *d_Sx, void *stream_)
{
	hipStream_t stream = (hipStream_t) stream_;
	HIP_CHECK(hipMemcpyAsync(d_Sp, W->d_Sp, ((size_t) W->last_rows + 1) * sizeof(i64), hipMemcpyDefault, stream));
	if (W->last_nnz > 0) {
		HIP_CHECK(hipMemcpyAsync(d_Sj, W->d_Sj, (size_t) W->last_nnz * sizeof(int), hipMemcpyDefault, stream));
		HIP_CHECK(hipMemcpyAsync(d_Sx, W->d_Sx, (size_t) W->last_nnz * sizeof(int), hipMemcpyDefault, stream));
	}
	HIP_CHECK(hipStreamSynchronize(stream));
}

// --------------------------------------------------------------------------
// spasm_hip_schur over several GPUs, split by columns
// --------------------------------------------------------------------------
// Every rank: its slab problem (host: spasm_hip_column_slab -- A and U with the other ranks' non-pivotal columns deleted),
// its image, ALL n rows reduced on it (the ordinary one-GPU call below, communicator set aside), columns mapped back,
// all-gatherv of the slabs, stitched into whole rows on every device, then downloaded / kept resident like any result.
extern "C" {
static struct spasm_csr *schur_entry(const struct spasm_csr *A, const int *p, int n, const struct spasm_lu *fact, double est_density, struct spasm_triplet *L,
                                     const int *p_in, int *p_out, spasm_hip_dwork **keep_on_device);
}

static struct spasm_csr *schur_by_column_slabs(const struct spasm_csr *A, const int *p, int n, const struct spasm_lu *fact, double est_density,
                                               const int *p_in, int *p_out, spasm_hip_comm *comm)
{
	const double t0 = wtime();
	const int m = A->m, world = comm_world(comm), rank = comm_rank(comm);
	const i64 prime = A->field->p;
	hipStream_t stream = nullptr;
	struct spasm_csr *A_slab = nullptr;
	struct spasm_lu *F_slab = nullptr;
	std::vector<int> cols((size_t) (m > 0 ? m : 1));
	const int mm = spasm_hip_column_slab(A, fact, rank, world, &A_slab, &F_slab, cols.data());
	const double t_slab = wtime() - t0;
	// the slab's Schur complement, on this device only, and kept there: the all-gatherv reads it from the workspace of the call
	// (round 5 brought it to the host and sent it back: 1 / world of S over PCIe, twice)
	spasm_hip_dwork *W = nullptr;
	(void) schur_entry(A_slab, p, n, F_slab, est_density, nullptr, nullptr, nullptr, &W);
	const i64 snz = W->last_nnz;
	int *d_cols = dalloc<int>(mm);
	upload(d_cols, cols.data(), mm, stream);
	launch_map_columns(W->d_Sj, snz, d_cols, stream);
	HIP_CHECK(hipStreamSynchronize(stream));
	resident_forget(A_slab);
	spasm_hip_csr_free(A_slab);
	spasm_hip_lu_free(F_slab);
	// Between two rounds of the driver (residency on, entries left on the device) the slabs STAY slabs: what the next step reads of S
	// is column-separable -- the census of leftmost entries (a minimum per row over the ranks), the random combinations and the
	// dense rows of the finish (every rank forms its columns of them: dense_api.hip) -- and whole rows are only gathered when
	// another sparse round or the host asks for them (resident_unslab).  The ranks exchange the row lengths: 4 bytes a row.
	if (g_lazy_download && resident_enabled() && n >= 1024) {
		int *d_len = dalloc<int>(n);
		hipLaunchKernelGGL(row_lengths_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, W->d_Sp, n, d_len);
		comm_allreduce_sum_i32(comm, d_len, n, stream);
		std::vector<int> len((size_t) n);
		sh::d2h(len.data(), d_len, (size_t) n * sizeof(int), stream);
		sh::big_free(d_len);
		i64 total = 0;
		for (int i = 0; i < n; i++)
			total += len[(size_t) i];
		struct spasm_csr *S = spasm_hip_csr_alloc(n, m, total, prime, true);
		S->p[0] = 0;
		for (int i = 0; i < n; i++)
			S->p[i + 1] = S->p[i] + len[(size_t) i];
		resident_adopt_slab(S, W->d_Sp, W->d_Sj, W->d_Sx, snz, comm);
		counters()[CNT_SLABS_KEPT] += 1;
		W->d_Sp = nullptr;
		W->d_Sj = nullptr;
		W->d_Sx = nullptr;
		sh::big_free(d_cols);
		spasm_hip_dwork_destroy(W);
		if (p_out != nullptr)
			for (int k = 0; k < n; k++)
				p_out[k] = (p_in != nullptr) ? p_in[p[k]] : p[k];
		const double density = (n > 0 && m > 0) ? (double) total / ((double) m * n) : 0.0;
		logmsg("Schur complement: %d * %d [%" PRId64 " nz / density= %.3f], %.1fs (split by columns: rank %d of %d holds its slab of %d of the %d non-pivotal columns, "
		       "%" PRId64 " entries, on the device; slab problem %.2fs)\n", n, m, total, density, wtime() - t0, rank, world, mm - fact->U->n, m - fact->U->n, snz, t_slab);
		return S;
	}
	// stack of the slabs, then whole rows
	i64 total = 0;
	int rows_all = 0;
	(void) spasm_hip_dschur_allgatherv(comm, W, nullptr, nullptr, nullptr, -1, &rows_all, &total, stream);
	if (rows_all != n * world)
		die("spasm_hip_schur (columns): the ranks hold %d slab rows in all, %d expected", rows_all, n * world);
	i64 *gSp = dalloc<i64>((i64) rows_all + 1);
	int *gSj = dalloc<int>(total);
	int *gSx = dalloc<int>(total);
	if (spasm_hip_dschur_allgatherv(comm, W, gSp, gSj, gSx, total, nullptr, nullptr, stream) != 0)
		die("spasm_hip_schur (columns): all-gatherv of the slabs failed");
	i64 *dSp = dalloc<i64>((i64) n + 1);
	int *dSj = static_cast<int *>(big_alloc((size_t) std::max<i64>(total, 1) * sizeof(int)));
	int *dSx = static_cast<int *>(big_alloc((size_t) std::max<i64>(total, 1) * sizeof(int)));
	if (W->d_lb_status == nullptr)
		W->d_lb_status = dalloc<unsigned long long>((i64) W->max_rows + 16 + 16 * 16);
	HIP_CHECK(hipMemsetAsync(W->d_ctr, 0, CTR_COUNT * sizeof(int), stream));
	launch_stitch_slabs(gSp, gSj, gSx, n, world, dSp, dSj, dSx, total, W->d_row_len, W->d_lb_status, W->d_ctr, stream);
	struct spasm_csr *S = spasm_hip_csr_alloc(n, m, total, prime, true);
	HIP_CHECK(hipMemcpyAsync(S->p, dSp, ((size_t) n + 1) * sizeof(i64), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	if (S->p[n] != total)
		die("spasm_hip_schur (columns): the stitched rows hold %lld entries, the slabs %lld", (long long) S->p[n], (long long) total);
	const bool keep = resident_enabled() && n >= 1024;
	const bool lazy = keep && g_lazy_download;
	if (total > 0 && !lazy) {
		HIP_CHECK(hipMemcpy(S->j, dSj, (size_t) total * sizeof(int), hipMemcpyDeviceToHost));
		HIP_CHECK(hipMemcpy(S->x, dSx, (size_t) total * sizeof(int), hipMemcpyDeviceToHost));
	}
	if (keep) {
		resident_adopt(S, dSp, dSj, dSx, lazy);          // the next round's A is already on every device
	} else {
		sh::big_free(dSp);
		big_free(dSj);
		big_free(dSx);
	}
	sh::big_free(gSp);
	sh::big_free(gSj);
	sh::big_free(gSx);
	sh::big_free(d_cols);
	spasm_hip_dwork_destroy(W);
	if (p_out != nullptr)
		for (int k = 0; k < n; k++)
			p_out[k] = (p_in != nullptr) ? p_in[p[k]] : p[k];
	const double density = (n > 0 && m > 0) ? (double) total / ((double) m * n) : 0.0;
	logmsg("Schur complement: %d * %d [%" PRId64 " nz / density= %.3f], %.1fs (split by columns: rank %d of %d reduced every row on %d of the %d "
	       "non-pivotal columns; slab problem %.2fs)\n", n, m, total, density, wtime() - t0, rank, world, mm - fact->U->n, m - fact->U->n, t_slab);
	return S;
}

// --------------------------------------------------------------------------
// host-pointer drop-in for spasm_schur
// --------------------------------------------------------------------------
static struct spasm_csr *schur_entry(const struct spasm_csr *A, const int *p, int n, const struct spasm_lu *fact, double est_density, struct spasm_triplet *L,
                                     const int *p_in, int *p_out, spasm_hip_dwork **keep_on_device);

struct spasm_csr *spasm_hip_schur(const struct spasm_csr *A, const int *p, int n, const struct spasm_lu *fact,
                                  double est_density, struct spasm_triplet *L, const int *p_in, int *p_out)
{
	return schur_entry(A, p, n, fact, est_density, L, p_in, p_out, nullptr);
}

// keep_on_device != NULL: one device, no communicator, no download -- the workspace that holds the result (W->d_Sp / d_Sj / d_Sx,
// W->last_rows, W->last_nnz) is handed to the caller, who destroys it, and NULL is returned (the column split: a rank's slab of S
// goes from here into the all-gatherv without touching the host)
static struct spasm_csr *schur_entry(const struct spasm_csr *A, const int *p, int n, const struct spasm_lu *fact, double est_density, struct spasm_triplet *L,
                                     const int *p_in, int *p_out, spasm_hip_dwork **keep_on_device)
{
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_schur: no HIP device (this library has no CPU path)");
	if (p == nullptr)
		die("spasm_hip_schur: the row list p must not be NULL");
	const int m = A->m;
	const i64 prime = A->field->p;
	const double t0 = wtime();
	hipStream_t stream = nullptr;
	// one process per GPU with a communicator installed (dist_api.hip): this rank reduces rows [lo, hi) of the list, the
	// slices are reassembled on the devices.  Small batches (density samples) and calls that record L are not sharded:
	// every rank computes them whole, which keeps the ranks in step without a collective.
	spasm_hip_comm *comm = (keep_on_device != nullptr) ? nullptr : current_comm();
	int lo = 0, hi = n;
	const bool shard = comm != nullptr && L == nullptr && (comm_world(comm) > 1 || env_int("SPASM_HIP_SHARD_FORCE", 0) != 0) &&
	                   n >= env_int("SPASM_HIP_SHARD_MIN_ROWS", 2048) * comm_world(comm);
	// Two ways to share a batch.  By COLUMNS when the factor takes an image path -- the sparse image or
	// the dense one: the columns of R and of S never meet, so rank k builds only ITS slab of R and reduces all rows on it;
	// nothing is replicated --, by ROWS otherwise (the row-by-row kernels: rows never meet; with an image every rank would
	// rebuild all of R).  SPASM_HIP_SHARD=rows|columns forces one.  The choice is made BEFORE the image of the whole factor is
	// planned -- it hangs on the sizes alone (r, Sm, nnz(U), p: the rules of spasm_hip_dfact_create) --, so that a rank of a
	// column split never plans, hashes or uploads more than its own slab (round 4 planned the full image on every rank first).
	if (shard) {
		const int r_all = fact->U->n, Sm_all = m - fact->U->n;
		bool shard_cols = false;
		const char *how = sh::env_get("SPASM_HIP_SHARD");
		if (how != nullptr && std::strcmp(how, "columns") == 0) {
			shard_cols = true;
		} else if (how == nullptr || std::strcmp(how, "rows") != 0) {
			int64_t bs_bytes = 0;
			const bool has_sparse_plan = sparse_image_possible(prime) && r_all > 0 && Sm_all > 0 &&
			                             (env_int("SPASM_HIP_SPARSE_IMAGE", -1) == 1 || (Sm_all >= 8192 && (double) r_all * (double) Sm_all >= 5e8));
			const bool has_dense_plan = env_int("SPASM_HIP_BACKSOLVE", -1) != 0 && backsolve_eligible(r_all, Sm_all, fact->U->p[r_all] - r_all, &bs_bytes, prime);
			shard_cols = has_sparse_plan || has_dense_plan;
		}
		if (shard_cols && Sm_all >= comm_world(comm))
			return schur_by_column_slabs(A, p, n, fact, est_density, p_in, p_out, comm);
	}
	spasm_hip_dfact *F = cached_dfact(fact->U, fact->qinv, stream);
	if (est_density >= 0)
		F->bs.density_hint = est_density;      // (a dense result is cheaper through the back-substituted image)
	const double t_fact = wtime() - t0;
	// device image of A and of the row list
	const i64 annz = A->p[A->n];
	DeviceMatrix devA(A, stream);
	if (shard)
		spasm_hip_shard(n, comm_rank(comm), comm_world(comm), &lo, &hi);
	const int n_all = n;
	const int *p_all = p;
	p += lo;
	n = hi - lo;
	int *drows = dalloc<int>(n);
	upload(drows, p, n, stream);
	spasm_hip_dcsr dA{A->n, m, annz, devA.p, devA.j, devA.x};

	if (est_density < 0)
		est_density = 0.0;     // the pool below is grown on demand instead of being estimated
	i64 in_nnz = 0;
	for (int k = 0; k < n; k++)
		in_nnz += A->p[p[k] + 1] - A->p[p[k]];
	// est_density is relative to the non-pivotal columns (spasm_schur_estimate_density divides by m - r)
	i64 pool = std::max<i64>((i64) (est_density * n * (double) (m - F->r) * 1.3), 4 * in_nnz) + (i64) 4096 * 4096;
	const i64 pool_max = (i64) n * (i64) (m - F->r) + (i64) 4096 * 4096;
	pool = std::min(pool, pool_max);
	spasm_hip_schur_stats st{};
	spasm_hip_dwork *W = nullptr;
	const double t1 = wtime();
	double t_wcreate = 0.0;
	// A full batch through the sparse image: the pool of S is sized from the image itself.  The driver's estimate comes from
	// 100 rows (spasm_schur_estimate_density) and the rows of these Schur complements differ by orders of magnitude: mk15.b4
	// was given 2.54e9 entries for 1.86e9 in one call and too few in another -- and a pool that is too small means the whole
	// call again on a fresh block of twice the size (tens of GB that the device has to map: the 2.2 s sparse round in one
	// call of five of round 4's bench, where the others took 0.07).  Here R is built first and 16,384 rows spread over the
	// batch go through it (under a millisecond): their entries, scaled, + 15 % + what the waves strand in their arenas
	// (8,192 rows + 12 % still fell short once in fifteen mk14.b4 calls).
	double ms_sample = 0.0;
	if (L == nullptr && !shard && n >= 65536 && (1) != 0 &&
	    sparse_image_wanted(F, env_int("SPASM_HIP_FORCE_TIER", 0) != 0 || env_int("SPASM_HIP_GROUP", -1) >= 0, n) &&
	    (F->sp.valid || sparse_image_build(F, stream))) {
		const int ns = 16384;
		std::vector<int> sample((size_t) ns);
		for (int k = 0; k < ns; k++)
			sample[(size_t) k] = p[(i64) k * n / ns];
		int *d_sample = dalloc<int>(ns);
		upload(d_sample, sample.data(), ns, stream);
		const i64 spool_max = (i64) ns * (i64) (m - F->r) + (i64) 4096 * 4096;
		i64 spool = std::min(spool_max, std::max<i64>((i64) (4.0 * est_density * ns * (double) (m - F->r)), (i64) 1 << 24) + (i64) 4096 * 4096);
		for (;;) {
			spasm_hip_dwork *Ws = spasm_hip_dwork_create(ns, m, spool);
			spasm_hip_schur_stats sts{};
			const int rc = dschur_impl(&dA, d_sample, ns, F, Ws, stream, &sts, nullptr);
			spasm_hip_dwork_destroy(Ws);
			if (rc == 0) {
				ms_sample = sts.ms_total;
				if (sts.used_sparse_image && sts.nnz > 0) {
					const i64 sized = (i64) (1.15 * (double) sts.nnz / (double) ns * (double) n) + (i64) 48 * 1024 * 1024;
					if (verbose() >= 2)
						logmsg("[schur/hip] pool of S: %" PRId64 " entries from %d sampled rows through the sparse image (%.1f per row), %" PRId64 " from the driver's estimate\n",
						       sized, ns, (double) sts.nnz / ns, pool);
					pool = std::min(pool_max, sized);
					counters()[CNT_POOL_RESIZED] += 1;
				}
				break;
			}
			if (spool >= spool_max)
				break;
			spool = std::min(spool_max, 2 * spool);
		}
		sh::big_free(d_sample);
	}
	// L requested: pools for the elimination coefficients, grown on demand
	LOut lout;
	i64 lcap = (L != nullptr) ? std::max<i64>(16 * in_nnz, (i64) 1 << 24) : 0;
	int *d_row_orig = nullptr;
	if (L != nullptr) {
		std::vector<int> ro((size_t) (n > 0 ? n : 1));
		for (int k = 0; k < n; k++)
			ro[k] = (p_in != nullptr) ? p_in[p[k]] : p[k];
		d_row_orig = dalloc<int>(n);
		upload(d_row_orig, ro.data(), n, stream);
		HIP_CHECK(hipStreamSynchronize(stream));
	}
	for (;;) {
		if (L != nullptr) {
			lout.row_orig = d_row_orig;
			lout.cap = lcap;
			lout.Li = dalloc<int>(lcap);
			lout.Lj = dalloc<int>(lcap);
			lout.Lx = dalloc<int>(lcap);
			HIP_CHECK(hipMemsetAsync(lout.Li, 0xFF, (size_t) lcap * sizeof(int), stream));
		}
		const double tw0 = wtime();
		W = spasm_hip_dwork_create(n, m, pool);
		t_wcreate += wtime() - tw0;
		// one-shot call: allocating tens of GB costs more than the kernel gains from having every
		// row group resident at once (hipMalloc is ~30 ms per GB); the device-level API keeps its
		// workspace and takes the large budget
		W->scratch_budget = (i64) 24 << 30;
		scratch_adopt(W);
		const int rc = dschur_impl(&dA, drows, n, F, W, stream, &st, (L != nullptr) ? &lout : nullptr);
		if (rc == 0) {
			// what the row-by-row kernels measured feeds the path choice of the next, larger batch on the same factor
			if (!st.used_backsolve && !st.used_sparse_image && n >= 64 && st.eliminations > 0)
				F->bs.elim_hint = (double) st.eliminations / (double) n;
			break;
		}
		scratch_park(W);
		spasm_hip_dwork_destroy(W);
		if (L != nullptr) {
			sh::big_free(lout.Li);
			sh::big_free(lout.Lj);
			sh::big_free(lout.Lx);
		}
		if (rc & 1) {
			if (pool >= pool_max)
				die("spasm_hip_schur: pool of %" PRId64 " entries still too small", pool);
			pool = std::min(pool_max, 2 * pool + m);
			counters()[CNT_POOL_RETRIES] += 1;
			logmsg("[schur/hip] pool too small, retrying with %" PRId64 " entries\n", pool);
		}
		if (rc & 2) {
			lcap *= 4;
			logmsg("[schur/hip] L pool too small, retrying with %" PRId64 " entries\n", lcap);
		}
	}
	if (L != nullptr) {
		// bring the coefficient triplets back; slots never written still hold -1
		const i64 used = std::min(lout.used, lcap);
		std::vector<int> hi((size_t) (used > 0 ? used : 1)), hj((size_t) (used > 0 ? used : 1)), hx((size_t) (used > 0 ? used : 1));
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
		sh::big_free(lout.Li);
		sh::big_free(lout.Lj);
		sh::big_free(lout.Lx);
		sh::big_free(d_row_orig);
	}
	const double t_run = wtime() - t1;
	const double t2 = wtime();
	struct spasm_csr *S = nullptr;
	bool lazy = false;
	if (keep_on_device != nullptr) {
		scratch_park(W);
		sh::big_free(drows);
		W->last_rows = n;
		W->last_nnz = st.nnz;
		*keep_on_device = W;
		logmsg("Schur complement (kept on the device): %d * %d [%" PRId64 " nz], %.1fs (GPU kernels %.1f ms, %s; factor image %.2fs, alloc+run %.2fs)\n", n, m, (i64) st.nnz,
		       wtime() - t0, st.ms_total + ms_sample, st.kernel, t_fact, t_run);
		return nullptr;
	}
	if (shard) {
		// all-gatherv of the slices (sizes first, then exact-count broadcasts), then one download of the whole
		i64 total = 0;
		int rows_all = 0;
		(void) spasm_hip_dschur_allgatherv(comm, W, nullptr, nullptr, nullptr, -1, &rows_all, &total, stream);
		if (rows_all != n_all)
			die("spasm_hip_schur: the ranks reduced %d rows in all, %d expected", rows_all, n_all);
		i64 *gSp = dalloc<i64>((i64) n_all + 1);
		int *gSj = dalloc<int>(total);
		int *gSx = dalloc<int>(total);
		if (spasm_hip_dschur_allgatherv(comm, W, gSp, gSj, gSx, total, nullptr, nullptr, stream) != 0)
			die("spasm_hip_schur: all-gatherv of the slices failed");
		S = spasm_hip_csr_alloc(n_all, m, total, prime, true);
		HIP_CHECK(hipMemcpy(S->p, gSp, ((size_t) n_all + 1) * sizeof(i64), hipMemcpyDeviceToHost));
		if (total > 0) {
			HIP_CHECK(hipMemcpy(S->j, gSj, (size_t) total * sizeof(int), hipMemcpyDeviceToHost));
			HIP_CHECK(hipMemcpy(S->x, gSx, (size_t) total * sizeof(int), hipMemcpyDeviceToHost));
		}
		if (resident_enabled() && n_all >= 1024) {
			resident_adopt(S, gSp, gSj, gSx, false);          // the next round's A is already on every device
		} else {
			sh::big_free(gSp);
			sh::big_free(gSj);
			sh::big_free(gSx);
		}
	} else {
		S = spasm_hip_csr_alloc(n, m, st.nnz, prime, true);
		sh::d2h(S->p, W->d_Sp, ((size_t) n + 1) * sizeof(i64), stream);
		const bool keep = resident_enabled() && n >= 1024;
		lazy = keep && g_lazy_download && L == nullptr;
		if (st.nnz > 0 && !lazy) {
			HIP_CHECK(hipMemcpy(S->j, W->d_Sj, (size_t) st.nnz * sizeof(int), hipMemcpyDeviceToHost));
			HIP_CHECK(hipMemcpy(S->x, W->d_Sx, (size_t) st.nnz * sizeof(int), hipMemcpyDeviceToHost));
		}
		if (keep) {
			// keep the result where it was computed -- it is the A of the next round: the table takes the workspace's arrays
			// over (no copy; they are sized for the estimate, a little more than the result)
			resident_adopt(S, W->d_Sp, W->d_Sj, W->d_Sx, lazy);
			W->d_Sp = nullptr;
			W->d_Sj = nullptr;
			W->d_Sx = nullptr;
		}
	}
	if (p_out != nullptr)
		for (int k = 0; k < n_all; k++)
			p_out[k] = (p_in != nullptr) ? p_in[p_all[k]] : p_all[k];
	const double t_down = wtime() - t2;
	const double t3 = wtime();
	scratch_park(W);
	spasm_hip_dwork_destroy(W);
	sh::big_free(drows);
	const double density = (S->n > 0 && m > 0) ? (double) S->p[S->n] / ((double) m * S->n) : 0.0;
	logmsg("Schur complement: %d * %d [%" PRId64 " nz / density= %.3f], %.1fs (GPU kernels %.1f ms, %s%s; tiers %d/%d/%d; "
	       "factor image %.2fs, alloc+run %.2fs, download %.2fs%s, free %.2fs)\n", S->n, m, S->p[S->n], density, wtime() - t0,
	       st.ms_total + ms_sample, st.kernel, st.group_aborted ? " after the row-group kernel gave up" : "", st.rows_lds, st.rows_lds_big, st.rows_dense,
	       t_fact, t_run, t_down, lazy ? " (row pointers only: the entries stay on the device)" : "", wtime() - t3);
	if (verbose() >= 2)
		logmsg("[schur/hip] of alloc+run: %.2fs allocating the workspace (%" PRId64 " pool entries), scratch %.1f GB\n", t_wcreate, pool,
		       (double) g_scratch_cache.bytes / 1073741824.0);
	return S;
}

// ... and spasm_hip_schur the way the driver calls it between two rounds (host_echelonize.cpp): residency on, the entries of S left on
// the device (row pointers only come to the host), the communicator in force -- so with several ranks this IS the column / row split
// with its all-gatherv and its stitching.  Returns the entries of S; S itself is dropped.
i64 spasm_hip_schur_resident(const struct spasm_csr *A, const int *p, int n, const struct spasm_lu *fact, double est_density)
{
	const bool was_on = resident_enabled();
	if (!was_on)
		resident_begin();
	resident_lazy_downloads(true);
	struct spasm_csr *S = spasm_hip_schur(A, p, n, fact, est_density, nullptr, nullptr, nullptr);
	resident_lazy_downloads(false);
	const i64 nnz = S->p[S->n];
	resident_forget(S);
	spasm_hip_csr_free(S);
	if (!was_on) {
		resident_forget(A);
		g_resident_on = false;          // (not resident_end(): the block cache keeps its blocks for the next step)
	}
	return nnz;
}

}  // extern "C"
