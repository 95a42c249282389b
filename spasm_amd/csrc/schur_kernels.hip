// HIP kernels (gfx950 / CDNA4, wave64) for the sparse Schur complement.
//
// One wavefront reduces one row of A against the factor.  The running row
// lives in an LDS hash table private to the wave (keys = column labels,
// values = lazily reduced sums); pivotal columns still to be eliminated sit
// in an LDS pending list.  Labels are assigned so that pivots are sorted by
// elimination level: every pending pivot of the lowest level present can be
// eliminated in the same step, and the rows of U' they select are streamed
// with coalesced 8-byte loads, flattened over the 64 lanes.
// See DESIGN.md ("Kernels") for the algorithm and its byte accounting.
#include <type_traits>
#include "device_types.h"
#include "field_dev.h"

namespace sh {

namespace {

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr int ROWS_PER_GRAB = 4;
constexpr int POOL_CHUNK = 2048;        // entries reserved from the pool at a time

__device__ __forceinline__ uint64_t lanes_below(int lane) { return (1ull << lane) - 1ull; }

// L factor: append (row, pivot index, coefficient) for the lanes that have one.  Space comes from a
// per-wave arena refilled from the pool 4096 entries at a time; status bit 1 (value 2) = pool exhausted.
struct LArena {
	int64_t off = 0;
	int left = 0;
};

__device__ __forceinline__ void record_L(const SchurArgs &a, LArena &ar, bool has, int row, uint32_t label, uint32_t v,
                                         int lane, const MontDev &F)
{
	const uint64_t mask = __ballot(has);
	const int cnt = __popcll(mask);
	if (cnt == 0)
		return;
	if (cnt > ar.left) {
		unsigned long long got = 0;
		if (lane == 0)
			got = atomicAdd(&a.ctr64[C64_LPOOL], 4096ull);
		const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) got);
		const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t) (got >> 32));
		ar.off = (int64_t) (((uint64_t) hi << 32) | lo);
		ar.left = 4096;
		if (ar.off + 4096 > a.L_cap) {
			ar.left = 0;
			if (lane == 0)
				atomicOr(&a.ctr[CTR_STATUS], 2);
			return;
		}
	}
	if (has) {
		const int64_t dst = ar.off + __popcll(mask & ((1ull << lane) - 1ull));
		a.L_i[dst] = row;
		a.L_j[dst] = a.kof[label];
		a.L_x[dst] = (v > F.half) ? (int) (v - F.p) : (int) v;
	}
	ar.off += cnt;
	ar.left -= cnt;
}

__device__ __forceinline__ int wave_exclusive_scan(int v, int lane, int &total)
{
	int x = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		int y = __shfl_up(x, d);
		if (lane >= d)
			x += y;
	}
	total = __shfl(x, 63);
	return x - v;
}

// exclusive scan of small counts (v < 16) without LDS traffic: one ballot + mbcnt per bit plane
__device__ __forceinline__ int wave_exclusive_scan_small(int v, int &total)
{
	int pos = 0, tot = 0;
#pragma unroll
	for (int b = 0; b < 4; b++) {
		const uint64_t m = __ballot((v >> b) & 1);
		pos += (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)) << b;
		tot += __popcll(m) << b;
	}
	total = tot;
	return pos;
}

__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		uint32_t o = (uint32_t) __shfl_xor((int) v, d);
		v = (o < v) ? o : v;
	}
	return v;
}

template <bool WIDE> struct Acc { using type = uint32_t; };
template <> struct Acc<true> { using type = unsigned long long; };

template <int H> __device__ __forceinline__ uint32_t slot_of(uint32_t c)
{
	constexpr int LOG = (H == 1024) ? 10 : (H == 2048) ? 11 : (H == 4096) ? 12 : 13;
	static_assert(H == 1024 || H == 2048 || H == 4096 || H == 8192, "table size");
	return (c * 0x9E3779B1u) >> (32 - LOG);
}

// LDS image of one wave.  vals first (8-byte aligned when WIDE).
template <int H, bool WIDE> struct WaveLds {
	using V = typename Acc<WIDE>::type;
	static constexpr int PCAP = H / 2;
	V vals[H];
	uint32_t keys[H];
	uint32_t pend[2][PCAP];
	uint32_t cnt[4];
};

// x[c] += delta.  Returns true when c was not in the table before.  New
// pivotal labels (c < r) are appended to the pending list.
template <int H, bool WIDE>
__device__ __forceinline__ bool table_add(WaveLds<H, WIDE> __attribute__((address_space(3))) *T, uint32_t c, uint32_t delta,
                                          int which, uint32_t r)
{
	using V = typename Acc<WIDE>::type;
	volatile uint32_t __attribute__((address_space(3))) *keys = T->keys;
	uint32_t s = slot_of<H>(c);
	bool fresh = false;
	for (;;) {
		uint32_t k = keys[s];
		if (k == c)
			break;
		if (k == EMPTY) {
			uint32_t old = EMPTY;          // LDS compare-and-swap: `old` receives what was there
			(void) __hip_atomic_compare_exchange_strong(&T->keys[s], &old, c, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			if (old == EMPTY) {
				fresh = true;
				break;
			}
			if (old == c)
				break;
		}
		s = (s + 1) & (H - 1);
	}
	(void) __hip_atomic_fetch_add(&T->vals[s], (V) delta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	if (fresh && c < r) {
		uint32_t pos = __hip_atomic_fetch_add(&T->cnt[which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		((volatile uint32_t __attribute__((address_space(3))) *) T->pend[which])[pos] = c;
	}
	return fresh;
}

template <int H, bool WIDE>
__device__ __forceinline__ typename Acc<WIDE>::type table_get(WaveLds<H, WIDE> __attribute__((address_space(3))) *T, uint32_t c)
{
	using V = typename Acc<WIDE>::type;
	volatile uint32_t __attribute__((address_space(3))) *keys = T->keys;
	uint32_t s = slot_of<H>(c);
	for (int guard = 0; guard < H; guard++) {
		if (keys[s] == c)
			return ((volatile V __attribute__((address_space(3))) *) T->vals)[s];
		s = (s + 1) & (H - 1);
	}
	return 0;
}

}  // namespace

// --------------------------------------------------------------------------
// K1: one wave per row, LDS hash accumulator.
// --------------------------------------------------------------------------
template <int H, bool WIDE>
__global__ __launch_bounds__(64) void schur_lds_kernel(SchurArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	using L = WaveLds<H, WIDE>;
	using V = typename Acc<WIDE>::type;
	// (address_space(3): generic pointers to LDS become FLAT instructions, which count in vmcnt and lgkmcnt)
	typedef L __attribute__((address_space(3))) LdsL;
	LdsL *T = (LdsL *) lds_raw;
	constexpr int PCAP = L::PCAP;
	constexpr int CAPK = (H * 3) / 4 - 64;      // keys allowed before a batch of 64 inserts
	constexpr int CAPP = PCAP - 64;
	const int lane = threadIdx.x;
	const uint32_t r = (uint32_t) a.r;
	const MontDev F = a.F;
	if (a.skip_done && a.ctr[a.skip_ctr] == 0)
		return;                     // the row-group kernel finished the batch: nothing left for the per-row tiers
	const int total_rows = (a.list != nullptr) ? *a.list_count : a.nrows;

	int64_t arena_off = 0;
	int arena_left = 0;
	unsigned long long st_elim = 0, st_stream = 0, st_input = 0;
	int st_done = 0;

	for (;;) {
		int kbase = 0;
		if (lane == 0)
			kbase = atomicAdd(&a.ctr[a.next_ctr], ROWS_PER_GRAB);
		kbase = __builtin_amdgcn_readfirstlane(kbase);
		if (kbase >= total_rows)
			break;
		const int kend = (kbase + ROWS_PER_GRAB < total_rows) ? kbase + ROWS_PER_GRAB : total_rows;
		for (int kk = kbase; kk < kend; kk++) {
			const int k = (a.list != nullptr) ? a.list[kk] : kk;
			if (a.skip_done && a.row_len[k] != -1)
				continue;               // already produced by the row-group probe
			const int i = a.rows[k];

			// ---- reset the table ----
			for (int s = lane; s < H; s += 64) {
				T->keys[s] = EMPTY;
				T->vals[s] = 0;
			}
			if (lane < 4)
				T->cnt[lane] = 0;
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

			int nkeys = 0;
			bool overflow = false;

			// ---- scatter the input row (relabelled on the fly) ----
			const int64_t lo = a.Ap[i], hi = a.Ap[i + 1];
			st_input += (unsigned long long) (hi - lo);
			for (int64_t base = lo; base < hi; base += 64) {
				uint32_t pc = ((volatile uint32_t __attribute__((address_space(3))) *) T->cnt)[0];
				if (nkeys > CAPK || (int) pc > CAPP) {
					overflow = true;
					break;
				}
				const int64_t px = base + lane;
				bool fresh = false;
				if (px < hi) {
					const int j = a.Aj[px];
					const uint32_t c = a.lab[j];
					const uint32_t v = from_balanced(a.Ax[px], F);
					fresh = table_add<H, WIDE>(T, c, v, 0, r);
				}
				nkeys += __popcll(__ballot(fresh));
			}

			// ---- eliminate, one level per step ----
			int cur = 0;
			while (!overflow) {
				__builtin_amdgcn_wave_barrier();
				const int P = (int) ((volatile uint32_t __attribute__((address_space(3))) *) T->cnt)[cur];
				if (P == 0)
					break;
				volatile uint32_t __attribute__((address_space(3))) *pcur = T->pend[cur];
				uint32_t cmin = EMPTY;
				for (int t = lane; t < P; t += 64) {
					uint32_t c = pcur[t];
					cmin = (c < cmin) ? c : cmin;
				}
				cmin = wave_min(cmin);
				const uint32_t lend = a.lvl_end[__builtin_amdgcn_readfirstlane(cmin)];
				const int nxt = cur ^ 1;
				if (lane == 0)
					((volatile uint32_t __attribute__((address_space(3))) *) T->cnt)[nxt] = 0;
				__builtin_amdgcn_wave_barrier();

				for (int base = 0; base < P && !overflow; base += 64) {
					if ((int) ((volatile uint32_t __attribute__((address_space(3))) *) T->cnt)[nxt] > CAPP) {
						overflow = true;
						break;
					}
					const int t = base + lane;
					const bool valid = t < P;
					const uint32_t c = valid ? pcur[t] : EMPTY;
					const bool sel = valid && (c < lend);
					const bool keep = valid && !sel;
					// survivors move to the other list
					const uint64_t mk = __ballot(keep);
					if (mk != 0) {
						uint32_t off = 0;
						if (lane == 0)
							off = __hip_atomic_fetch_add(&T->cnt[nxt], (uint32_t) __popcll(mk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						off = __builtin_amdgcn_readfirstlane(off);
						if (keep)
							((volatile uint32_t __attribute__((address_space(3))) *) T->pend[nxt])[off + __popcll(mk & lanes_below(lane))] = c;
					}
					const uint64_t ms = __ballot(sel);
					if (ms == 0)
						continue;
					// coefficient and row extent of each selected pivot
					uint32_t v = 0;
					if (sel)
						v = reduce_sum(table_get<H, WIDE>(T, c), F);
					int len = 0;
					uint64_t start = 0;
					if (v != 0) {
						start = a.rp[c];
						len = (int) (a.rp[c + 1] - start);
					}
					const uint32_t w = F.p - v;           // -v mod p (only used when v != 0)
					st_elim += (unsigned long long) __popcll(__ballot(v != 0));
					int tot;
					const int excl = wave_exclusive_scan(len, lane, tot);
					st_stream += (unsigned long long) tot;
					const uint32_t start_lo = (uint32_t) start, start_hi = (uint32_t) (start >> 32);
					for (int f0 = 0; f0 < tot; f0 += 64) {
						uint32_t pc = ((volatile uint32_t __attribute__((address_space(3))) *) T->cnt)[nxt];
						if (nkeys > CAPK || (int) pc > CAPP) {
							overflow = true;
							break;
						}
						const int f = f0 + lane;
						int owner = 0;
#pragma unroll
						for (int step = 32; step >= 1; step >>= 1) {
							const int cand = owner + step;
							const int e = __shfl(excl, cand);
							if (e <= f)
								owner = cand;
						}
						const uint32_t o_lo = (uint32_t) __shfl((int) start_lo, owner);
						const uint32_t o_hi = (uint32_t) __shfl((int) start_hi, owner);
						const uint32_t o_w = (uint32_t) __shfl((int) w, owner);
						const int o_ex = __shfl(excl, owner);
						bool fresh = false;
						if (f < tot) {
							const uint64_t idx = (((uint64_t) o_hi << 32) | o_lo) + (uint64_t) (f - o_ex);
							const uint2 e = a.ent[idx];
							const uint32_t delta = montmul(o_w, e.y, F);
							fresh = table_add<H, WIDE>(T, e.x, delta, nxt, r);
						}
						nkeys += __popcll(__ballot(fresh));
					}
				}
				cur = nxt;
			}

			if (overflow) {
				if (lane == 0) {
					int pos = atomicAdd(&a.ctr[a.ovf_ctr], 1);
					a.ovf_list[pos] = k;
					a.row_len[k] = -2;
				}
				continue;
			}

			// ---- emit the non-pivotal, non-zero entries ----
			__builtin_amdgcn_wave_barrier();
			int count = 0;
			for (int s0 = 0; s0 < H; s0 += 64) {
				const int s = s0 + lane;
				const uint32_t key = ((volatile uint32_t __attribute__((address_space(3))) *) T->keys)[s];
				bool keep = (key != EMPTY) && (key >= r);
				if (keep)
					keep = reduce_sum(((volatile V __attribute__((address_space(3))) *) T->vals)[s], F) != 0;
				count += __popcll(__ballot(keep));
			}
			if (count > arena_left) {
				const int want = (count > POOL_CHUNK) ? count : POOL_CHUNK;
				unsigned long long got = 0;
				if (lane == 0)
					got = atomicAdd(&a.ctr64[C64_POOL], (unsigned long long) want);
				const uint32_t g_lo = __builtin_amdgcn_readfirstlane((uint32_t) got);
				const uint32_t g_hi = __builtin_amdgcn_readfirstlane((uint32_t) (got >> 32));
				arena_off = (int64_t) (((uint64_t) g_hi << 32) | g_lo);
				arena_left = want;
				if (arena_off + want > a.pool_cap) {
					arena_left = 0;
					if (lane == 0) {
						atomicOr(&a.ctr[CTR_STATUS], 1);
						a.row_len[k] = -1;
					}
					continue;
				}
			}
			int64_t wpos = arena_off;
			for (int s0 = 0; s0 < H; s0 += 64) {
				const int s = s0 + lane;
				const uint32_t key = ((volatile uint32_t __attribute__((address_space(3))) *) T->keys)[s];
				bool keep = (key != EMPTY) && (key >= r);
				uint32_t v = 0;
				if (keep) {
					v = reduce_sum(((volatile V __attribute__((address_space(3))) *) T->vals)[s], F);
					keep = v != 0;
				}
				const uint64_t mk = __ballot(keep);
				if (keep) {
					const int64_t dst = wpos + __popcll(mk & lanes_below(lane));
					a.pool_j[dst] = a.q[key - r];
					a.pool_x[dst] = to_balanced(v, F);
				}
				wpos += __popcll(mk);
			}
			if (lane == 0) {
				a.row_off[k] = arena_off;
				a.row_len[k] = count;
			}
			arena_off += count;
			arena_left -= count;
			st_done += 1;
		}
	}
	if (lane == 0) {
		atomicAdd(&a.ctr64[C64_ELIM], st_elim);
		atomicAdd(&a.ctr64[C64_STREAM], st_stream);
		atomicAdd(&a.ctr64[C64_INPUT], st_input);
		atomicAdd(&a.ctr[a.done_ctr], st_done);
	}
}

// --------------------------------------------------------------------------
// K3: row lengths -> row pointers (three small passes), then gather + sort.
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scan_block_sums(const int *row_len, int n, int64_t *blocksum)
{
	__shared__ int64_t part[256];
	const int tid = threadIdx.x;
	const int64_t base = (int64_t) blockIdx.x * 1024;
	int64_t s = 0;
	for (int t = 0; t < 4; t++) {
		int64_t idx = base + tid * 4 + t;
		if (idx < n) {
			int l = row_len[idx];
			s += (l > 0) ? l : 0;
		}
	}
	part[tid] = s;
	__syncthreads();
	for (int d = 128; d > 0; d >>= 1) {
		if (tid < d)
			part[tid] += part[tid + d];
		__syncthreads();
	}
	if (tid == 0)
		blocksum[blockIdx.x] = part[0];
}

__global__ __launch_bounds__(256) void scan_of_sums(int64_t *blocksum, int nblocks)
{
	// single block: exclusive scan of up to a few hundred thousand block sums
	__shared__ int64_t carry;
	__shared__ int64_t buf[256];
	const int tid = threadIdx.x;
	if (tid == 0)
		carry = 0;
	__syncthreads();
	for (int base = 0; base < nblocks; base += 256) {
		int64_t v = (base + tid < nblocks) ? blocksum[base + tid] : 0;
		buf[tid] = v;
		__syncthreads();
		for (int d = 1; d < 256; d <<= 1) {
			int64_t add = (tid >= d) ? buf[tid - d] : 0;
			__syncthreads();
			buf[tid] += add;
			__syncthreads();
		}
		int64_t incl = buf[tid];
		int64_t c = carry;
		__syncthreads();
		if (base + tid < nblocks)
			blocksum[base + tid] = c + incl - v;
		if (tid == 255)
			carry = c + incl;
		__syncthreads();
	}
}

__global__ __launch_bounds__(256) void scan_finish(const int *row_len, int n, const int64_t *blocksum, int64_t *Sp)
{
	__shared__ int64_t part[256];
	const int tid = threadIdx.x;
	const int64_t base = (int64_t) blockIdx.x * 1024;
	int l[4];
	int64_t s = 0;
	for (int t = 0; t < 4; t++) {
		int64_t idx = base + tid * 4 + t;
		int v = (idx < n) ? row_len[idx] : 0;
		l[t] = (v > 0) ? v : 0;
		s += l[t];
	}
	part[tid] = s;
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) {
		int64_t add = (tid >= d) ? part[tid - d] : 0;
		__syncthreads();
		part[tid] += add;
		__syncthreads();
	}
	int64_t run = blocksum[blockIdx.x] + part[tid] - s;
	for (int t = 0; t < 4; t++) {
		int64_t idx = base + tid * 4 + t;
		if (idx < n)
			Sp[idx] = run;
		run += l[t];
	}
	if (blockIdx.x == gridDim.x - 1 && tid == 255)
		Sp[n] = run;
}

// one wave per row: copy from the pool to its final place, sorted by column.
// Launched twice: rows of at most SORT_SMALL entries (small LDS footprint,
// many waves per CU), then the few longer ones with a 64 KiB sort buffer.
constexpr int SORT_SMALL = 1024;
constexpr int SORT_LDS_MAX = 8192;

template <int CAP>
__global__ __launch_bounds__(64) void gather_rows_kernel(const int *pool_j, const int *pool_x, const int64_t *row_off,
                                                        const int *row_len, int n, const int64_t *Sp, int *Sj, int *Sx,
                                                        int sort_rows, int len_lo, int len_hi)
{
	__shared__ int kj[CAP];
	__shared__ int kx[CAP];
	const int lane = threadIdx.x;
	for (int k = blockIdx.x; k < n; k += gridDim.x) {
		const int len = row_len[k];
		if (len <= 0)
			continue;
		const int64_t raw = row_off[k];
		const bool presorted = (raw >> 62) & 1;
		// rows that only need copying (already sorted, or sorting is off) all belong to the first launch
		// (small LDS footprint, many waves); the second launch takes the long rows that must be sorted
		const bool copy_only = !sort_rows || presorted;
		const bool mine = (len_lo == 0) ? (copy_only || len <= len_hi) : (!copy_only && len > len_lo);
		if (!mine)
			continue;
		const int64_t src = raw & ~(1LL << 62);
		const int64_t dst = Sp[k];
		if (copy_only || len > CAP) {
			for (int t = lane; t < len; t += 64) {
				Sj[dst + t] = pool_j[src + t];
				Sx[dst + t] = pool_x[src + t];
			}
			continue;
		}
		if (len <= 64) {
			int key = (lane < len) ? pool_j[src + lane] : 0x7FFFFFFF;
			int val = (lane < len) ? pool_x[src + lane] : 0;
#pragma unroll
			for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
				for (int jj = kk >> 1; jj > 0; jj >>= 1) {
					const int pk = __shfl_xor(key, jj);
					const int pv = __shfl_xor(val, jj);
					const bool asc = (lane & kk) == 0;
					const bool lower = (lane & jj) == 0;
					const bool want_min = (lower == asc);
					const bool take = want_min ? (pk < key) : (pk > key);
					if (take) {
						key = pk;
						val = pv;
					}
				}
			}
			if (lane < len) {
				Sj[dst + lane] = key;
				Sx[dst + lane] = val;
			}
			continue;
		}
		int n2 = 128;
		while (n2 < len)
			n2 <<= 1;
		for (int t = lane; t < n2; t += 64) {
			kj[t] = (t < len) ? pool_j[src + t] : 0x7FFFFFFF;
			kx[t] = (t < len) ? pool_x[src + t] : 0;
		}
		__syncthreads();
		for (int kk = 2; kk <= n2; kk <<= 1) {
			for (int jj = kk >> 1; jj > 0; jj >>= 1) {
				for (int t = lane; t < n2; t += 64) {
					const int o = t ^ jj;
					if (o > t) {
						const bool asc = (t & kk) == 0;
						const int a0 = kj[t], a1 = kj[o];
						if ((a0 > a1) == asc) {
							kj[t] = a1;
							kj[o] = a0;
							const int b0 = kx[t];
							kx[t] = kx[o];
							kx[o] = b0;
						}
					}
				}
				__syncthreads();
			}
		}
		for (int t = lane; t < len; t += 64) {
			Sj[dst + t] = kj[t];
			Sx[dst + t] = kx[t];
		}
		__syncthreads();
	}
}

// --------------------------------------------------------------------------
// launch helpers (called from schur_api.hip)
// --------------------------------------------------------------------------
template <int H, bool WIDE> static void launch_lds(const SchurArgs &a, int blocks, hipStream_t stream)
{
	using L = WaveLds<H, WIDE>;
	const size_t bytes = sizeof(L);
	static bool configured = false;
	if (!configured) {
		HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&schur_lds_kernel<H, WIDE>),
		                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes));
		configured = true;
	}
	hipLaunchKernelGGL((schur_lds_kernel<H, WIDE>), dim3(blocks), dim3(64), bytes, stream, a);
	HIP_CHECK(hipGetLastError());
}

void launch_schur_lds(const SchurArgs &a, int table, bool wide, int blocks, hipStream_t stream)
{
	if (table == 1024 && !wide)
		launch_lds<1024, false>(a, blocks, stream);
	else if (table == 1024 && wide)
		launch_lds<1024, true>(a, blocks, stream);
	else if (table == 8192 && !wide)
		launch_lds<8192, false>(a, blocks, stream);
	else if (table == 8192 && wide)
		launch_lds<8192, true>(a, blocks, stream);
	else
		die("no LDS kernel for table size %d", table);
}

size_t schur_lds_bytes(int table, bool wide)
{
	if (table == 1024)
		return wide ? sizeof(WaveLds<1024, true>) : sizeof(WaveLds<1024, false>);
	return wide ? sizeof(WaveLds<8192, true>) : sizeof(WaveLds<8192, false>);
}

// exclusive scan of max(row_len, 0) into Sp[0..n]
void launch_row_scan(const int *row_len, int n, int64_t *blocksum, int64_t *Sp, hipStream_t stream)
{
	if (n == 0) {
		HIP_CHECK(hipMemsetAsync(Sp, 0, sizeof(int64_t), stream));
		return;
	}
	const int nblocks = (n + 1023) / 1024;
	hipLaunchKernelGGL(scan_block_sums, dim3(nblocks), dim3(256), 0, stream, row_len, n, blocksum);
	hipLaunchKernelGGL(scan_of_sums, dim3(1), dim3(256), 0, stream, blocksum, nblocks);
	hipLaunchKernelGGL(scan_finish, dim3(nblocks), dim3(256), 0, stream, row_len, n, blocksum, Sp);
	HIP_CHECK(hipGetLastError());
}

void launch_finalize(const spasm_hip_dwork *W, int nrows, int sort_rows, hipStream_t stream)
{
	const int nblocks = (nrows + 1023) / 1024;
	if (nrows == 0) {
		HIP_CHECK(hipMemsetAsync(W->d_Sp, 0, sizeof(int64_t), stream));
		return;
	}
	hipLaunchKernelGGL(scan_block_sums, dim3(nblocks), dim3(256), 0, stream, W->d_row_len, nrows, W->d_blocksum);
	hipLaunchKernelGGL(scan_of_sums, dim3(1), dim3(256), 0, stream, W->d_blocksum, nblocks);
	hipLaunchKernelGGL(scan_finish, dim3(nblocks), dim3(256), 0, stream, W->d_row_len, nrows, W->d_blocksum, W->d_Sp);
	int blocks = nrows < 16384 ? nrows : 16384;
	hipLaunchKernelGGL((gather_rows_kernel<SORT_SMALL>), dim3(blocks), dim3(64), 0, stream, W->d_pool_j, W->d_pool_x,
	                   W->d_row_off, W->d_row_len, nrows, W->d_Sp, W->d_Sj, W->d_Sx, sort_rows, 0, SORT_SMALL);
	blocks = nrows < 512 ? nrows : 512;
	hipLaunchKernelGGL((gather_rows_kernel<SORT_LDS_MAX>), dim3(blocks), dim3(64), 0, stream, W->d_pool_j, W->d_pool_x,
	                   W->d_row_off, W->d_row_len, nrows, W->d_Sp, W->d_Sj, W->d_Sx, sort_rows, SORT_SMALL, 0x7FFFFFFF);
	HIP_CHECK(hipGetLastError());
}

}  // namespace sh

namespace sh {

__global__ void all_rows_to_list_kernel(int *list, int *count, int *row_len, int nrows)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < nrows) {
		list[t] = t;
		row_len[t] = -2;
	}
	if (t == 0)
		*count = nrows;
}

void launch_all_rows_to_list(int *list, int *count, int *row_len, int nrows, hipStream_t stream)
{
	hipLaunchKernelGGL(all_rows_to_list_kernel, dim3((nrows + 255) / 256), dim3(256), 0, stream, list, count, row_len, nrows);
	HIP_CHECK(hipGetLastError());
}

}  // namespace sh

// --------------------------------------------------------------------------
// K2w: dense accumulators in HBM, ONE WAVE per row, thousands of rows in
// flight.  For rows whose reach does not fit an LDS table (on mk13.b5 that is
// most of them: ~8000 eliminations and ~10^4 distinct columns per row).
//
// Per wave: xp[0..r) pivotal part, bm = bitmap of pending pivots, xn[0..Sm)
// non-pivotal part, all private to the wave, all zero between rows.  Updates
// are fire-and-forget wavefront-scope atomics (performed in the XCD's L2),
// every read of that state is an sc1 load (served by L2, never by the L1).
// A step = find the first pending label, take every pending label of its
// level, stream the rows of U' they select (flattened over the 64 lanes).
// --------------------------------------------------------------------------
namespace sh {

namespace {

constexpr uint32_t MIXED = 0xFFFFFFFFu;

template <typename V> __device__ __forceinline__ V ld_sc1(const V *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (SCOPE: wavefront for state private to one wave; workgroup when several waves of a workgroup update the same
//  lines -- on gfx950 both are the same instruction, the scope only tells the compiler what may race)
template <typename V, int SCOPE = __HIP_MEMORY_SCOPE_WAVEFRONT> __device__ __forceinline__ void add_ff(V *p, uint32_t delta)
{
	(void) __hip_atomic_fetch_add(p, (V) delta, __ATOMIC_RELAXED, SCOPE);
}

__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- software-managed vmcnt ------------------------------------------------
// On CDNA every vector memory instruction of a wave (loads, stores, atomics
// without return) retires in issue order and counts in vmcnt.  The compiler
// only knows how many instructions follow a load when that number is static;
// in the elimination loop it is not (rows have any number of entries), so it
// waits with vmcnt(0) before every use of a prefetched accumulator line --
// i.e. for the round trip of every atomic issued so far -- and a row group
// advances at one trip per memory latency.  The group kernel therefore keeps
// its prefetched lines away from the compiler: the loads are inline asm that
// writes ACCUMULATION registers (a0.., which the compiler never allocates in a
// kernel without MFMA and below 256 VGPRs), and a line is taken out with
// s_waitcnt vmcnt(n) + v_accvgpr_read, n = a lower bound, kept by the kernel,
// of the number of vector memory instructions issued after the load.  A lower
// bound is always safe: outstanding <= n <= (instructions issued after the
// load) means the load has retired.  (A value the compiler knows about cannot
// be used for this: it is free to copy a register before the wait.)
template <bool WIDE> __device__ __forceinline__ void ring_reserve()
{
	// make the kernel allocate the ring: a0..a47 (32-bit lines) or a0..a63 (64-bit lines)
	asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
	asm volatile("" ::: "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29",
	             "a30", "a31");
	asm volatile("" ::: "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45",
	             "a46", "a47");
	if constexpr (WIDE)
		asm volatile("" ::: "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61",
		             "a62", "a63");
}
template <int I> __device__ __forceinline__ void ring_issue(const uint32_t *p)
{
	static_assert(I < 64, "ring too large");
	asm volatile("global_load_dword a%1, %0, off sc1" : : "v"(p), "n"(I) : "memory");
}
template <int I> __device__ __forceinline__ void ring_issue(const unsigned long long *p)
{
	static_assert(I % 2 == 0 && I + 1 < 64, "64-bit tuples are even-aligned");
	asm volatile("global_load_dwordx2 a[%1:%2], %0, off sc1" : : "v"(p), "n"(I), "n"(I + 1) : "memory");
}
template <int I> __device__ __forceinline__ void ring_issue_x4(const void *p)          // read-only data: no sc1
{
	static_assert(I % 2 == 0 && I + 3 < 64, "128-bit tuples are even-aligned");
	asm volatile("global_load_dwordx4 a[%1:%2], %0, off" : : "v"(p), "n"(I), "n"(I + 3) : "memory");
}
template <int I> __device__ __forceinline__ void ring_issue_x2_ro(const void *p)
{
	static_assert(I % 2 == 0 && I + 1 < 64, "64-bit tuples are even-aligned");
	asm volatile("global_load_dwordx2 a[%1:%2], %0, off" : : "v"(p), "n"(I), "n"(I + 1) : "memory");
}
template <int I> __device__ __forceinline__ void ring_read2(uint32_t &x, uint32_t &y)
{
	asm volatile("v_accvgpr_read_b32 %0, a%2\n\tv_accvgpr_read_b32 %1, a%3" : "=v"(x), "=v"(y) : "n"(I), "n"(I + 1) : "memory");
}
template <int I> __device__ __forceinline__ void ring_issue_ro(const uint32_t *p)
{
	asm volatile("global_load_dword a%1, %0, off" : : "v"(p), "n"(I) : "memory");
}
template <int I> __device__ __forceinline__ void ring_read(uint32_t &x)
{
	asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(x) : "n"(I) : "memory");
}
template <int I> __device__ __forceinline__ void ring_read(unsigned long long &x)
{
	uint32_t lo, hi;
	asm volatile("v_accvgpr_read_b32 %0, a%2\n\tv_accvgpr_read_b32 %1, a%3" : "=v"(lo), "=v"(hi) : "n"(I), "n"(I + 1) : "memory");
	x = ((unsigned long long) hi << 32) | lo;
}

// wait until at most n (wave-uniform, held in an SGPR) vector memory instructions are outstanding
#define SPASM_W(K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
__device__ __forceinline__ void wait_vm_at_most(int n)
{
	n = __builtin_amdgcn_readfirstlane(n);
	if (n >= 64)            // the counter has 6 bits: with 64 younger instructions issued, the awaited one has retired
		return;
	if (n < 32) {
		if (n < 16) {
			if (n < 8) {
				if (n < 4) {
					if (n < 2) {
						if (n < 1) SPASM_W(0); else SPASM_W(1);
					} else {
						if (n < 3) SPASM_W(2); else SPASM_W(3);
					}
				} else {
					if (n < 6) SPASM_W(4); else SPASM_W(6);
				}
			} else {
				if (n < 12) {
					if (n < 10) SPASM_W(8); else SPASM_W(10);
				} else {
					if (n < 14) SPASM_W(12); else SPASM_W(14);
				}
			}
		} else {
			if (n < 24) {
				if (n < 20) SPASM_W(16); else SPASM_W(20);
			} else {
				if (n < 28) SPASM_W(24); else SPASM_W(28);
			}
		}
	} else {
		if (n < 48) {
			if (n < 40) {
				if (n < 36) SPASM_W(32); else SPASM_W(36);
			} else {
				if (n < 44) SPASM_W(40); else SPASM_W(44);
			}
		} else {
			if (n < 56) {
				if (n < 52) SPASM_W(48); else SPASM_W(52);
			} else {
				if (n < 60) SPASM_W(56); else SPASM_W(60);
			}
		}
	}
}
#undef SPASM_W

template <int I, int N, typename Fn> __device__ __forceinline__ void static_for(Fn &&f)
{
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		static_for<I + 1, N>(f);
	}
}

}  // namespace

struct WaveDenseArgs {
	SchurArgs a;
	unsigned char *scratch;   // per-wave slots
	int64_t slot_bytes;
	int64_t off_bm, off_xn;   // byte offsets of the bitmap and of xn inside a slot
	uint32_t *dense_out;      // non-null: dense rows (values in [0,p)) instead of the pool
	int64_t ldS;
};

template <bool WIDE>
__global__ __launch_bounds__(64) void schur_wave_dense_kernel(WaveDenseArgs d)
{
	using V = typename Acc<WIDE>::type;
	const SchurArgs &a = d.a;
	const int lane = threadIdx.x;
	const uint32_t r = (uint32_t) a.r;
	const int Sm = a.Sm;
	const MontDev F = a.F;
	if (a.skip_done && a.ctr[a.skip_ctr] == 0)
		return;
	const int total_rows = (a.list != nullptr) ? *a.list_count : a.nrows;

	unsigned char *slot = d.scratch + (int64_t) blockIdx.x * d.slot_bytes;
	V *xp = reinterpret_cast<V *>(slot);
	V *xn = reinterpret_cast<V *>(slot + d.off_xn);

	unsigned long long st_elim = 0, st_stream = 0, st_input = 0;
	int st_done = 0;
	LArena larena;

	for (;;) {
		int kk = 0;
		if (lane == 0)
			kk = atomicAdd(&a.ctr[a.next_ctr], 1);
		kk = __builtin_amdgcn_readfirstlane(kk);
		if (kk >= total_rows)
			break;
		const int k = (a.list != nullptr) ? a.list[kk] : kk;
		const int i = a.rows[k];
		const int row_to_record = (a.L_i != nullptr) ? a.row_orig[k] : 0;

		// ---- scatter the input row ----
		const int64_t lo = a.Ap[i], hi = a.Ap[i + 1];
		st_input += (unsigned long long) (hi - lo);
		for (int64_t px = lo + lane; px < hi; px += 64) {
			const uint32_t c = a.lab[a.Aj[px]];
			const uint32_t v = reduce_sum(from_balanced(a.Ax[px], F), F);
			if (c < r)
				add_ff(&xp[c], v);
			else
				add_ff(&xn[c - r], v);
		}

		// ---- eliminate level by level.  Pending pivots are the non-zero entries of xp at or after
		// the cursor (labels are sorted by level, contributions only go to later levels) ----
		uint32_t cursor = 0;
		for (;;) {
			drain_vmem();                       // every earlier update has been performed
			// first pending label at or after the cursor: sweep xp, 256 labels per trip
			uint32_t c0 = 0xFFFFFFFFu;
			for (uint32_t base = cursor & ~63u; base < r; base += 256) {
				V raw[4];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const uint32_t c = base + u * 64 + lane;
					raw[u] = (c < r && c >= cursor) ? ld_sc1(&xp[c]) : (V) 0;
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const uint64_t mask = __ballot(raw[u] != 0);
					if (mask != 0 && c0 == 0xFFFFFFFFu)
						c0 = base + u * 64 + (uint32_t) __builtin_ctzll(mask);
				}
				if (c0 != 0xFFFFFFFFu)
					break;
			}
			if (c0 == 0xFFFFFFFFu)
				break;
			const uint32_t lwe = a.lvl_end_w[c0 >> 5];
			const uint32_t lend = (lwe != MIXED) ? lwe * 32 : a.lvl_end[c0];

			// every pending pivot of [c0, lend), 64 labels at a time
			for (uint32_t base = c0 & ~63u; base < lend; base += 64) {
				const uint32_t c = base + lane;
				const bool inrange = (c >= c0) && (c < lend);
				const V raw = inrange ? ld_sc1(&xp[c]) : (V) 0;
				const bool sel = raw != 0;
				if (__ballot(sel) == 0)
					continue;
				uint32_t v = 0;
				uint64_t start = 0;
				int len = 0;
				if (sel) {
					start = a.rp[c];
					len = (int) (a.rp[c + 1] - start);
					v = reduce_sum(raw, F);
					__hip_atomic_store(&xp[c], (V) 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				}
				if (v == 0)
					len = 0;
				const uint32_t w_neg = F.p - v;
				st_elim += (unsigned long long) __popcll(__ballot(v != 0));
				if (a.L_i != nullptr)
					record_L(a, larena, v != 0, row_to_record, c, v, lane, F);
				int ftot;
				const int excl = wave_exclusive_scan(len, lane, ftot);
				st_stream += (unsigned long long) ftot;
				const uint32_t start_lo = (uint32_t) start, start_hi = (uint32_t) (start >> 32);
				for (int f0 = 0; f0 < ftot; f0 += 64) {
					const int f = f0 + lane;
					int owner = 0;
#pragma unroll
					for (int step = 32; step >= 1; step >>= 1) {
						const int cand = owner + step;
						const int e = __shfl(excl, cand);
						if (e <= f)
							owner = cand;
					}
					const uint32_t o_lo = (uint32_t) __shfl((int) start_lo, owner);
					const uint32_t o_hi = (uint32_t) __shfl((int) start_hi, owner);
					const uint32_t o_w = (uint32_t) __shfl((int) w_neg, owner);
					const int o_ex = __shfl(excl, owner);
					if (f < ftot) {
						const uint64_t idx = (((uint64_t) o_hi << 32) | o_lo) + (uint64_t) (f - o_ex);
						const uint2 e = a.ent[idx];
						const uint32_t delta = montmul(o_w, e.y, F);
						if (e.x < r)
							add_ff(&xp[e.x], delta);
						else
							add_ff(&xn[e.x - r], delta);
					}
				}
			}
			cursor = lend;
		}
		drain_vmem();

		// ---- the non-pivotal part is the row of S ----
		if (d.dense_out != nullptr) {
			uint32_t *out = d.dense_out + (int64_t) k * d.ldS;
			for (int t = lane; t < Sm; t += 64) {
				const V raw = ld_sc1(&xn[t]);
				out[t] = reduce_sum(raw, F);
				if (raw != 0)
					__hip_atomic_store(&xn[t], (V) 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
			}
			if (lane == 0)
				a.row_len[k] = Sm;
			st_done += 1;
			continue;
		}
		int count = 0;
		for (int t0 = 0; t0 < Sm; t0 += 256) {
			V raw[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int t = t0 + u * 64 + lane;
				raw[u] = (t < Sm) ? ld_sc1(&xn[t]) : (V) 0;
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const bool keep = (raw[u] != 0) && (reduce_sum(raw[u], F) != 0);
				count += __popcll(__ballot(keep));
			}
		}
		unsigned long long got = 0;
		if (lane == 0)
			got = atomicAdd(&a.ctr64[C64_POOL], (unsigned long long) count);
		const uint32_t g_lo = __builtin_amdgcn_readfirstlane((uint32_t) got);
		const uint32_t g_hi = __builtin_amdgcn_readfirstlane((uint32_t) (got >> 32));
		const int64_t off = (int64_t) (((uint64_t) g_hi << 32) | g_lo);
		const bool fits = off + count <= a.pool_cap;
		int64_t wpos = off;
		for (int t0 = 0; t0 < Sm; t0 += 256) {
			V raw[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int t = t0 + u * 64 + lane;
				raw[u] = (t < Sm) ? ld_sc1(&xn[t]) : (V) 0;
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int t = t0 + u * 64 + lane;
				uint32_t v = 0;
				if (raw[u] != 0) {
					v = reduce_sum(raw[u], F);
					__hip_atomic_store(&xn[t], (V) 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				}
				const bool keep = v != 0;
				const uint64_t mk = __ballot(keep);
				if (keep && fits) {
					const int64_t dst = wpos + __popcll(mk & lanes_below(lane));
					a.pool_j[dst] = a.q[t];
					a.pool_x[dst] = to_balanced(v, F);
				}
				wpos += __popcll(mk);
			}
		}
		if (lane == 0) {
			if (fits) {
				a.row_off[k] = off | (1LL << 62);       // sorted by column already
				a.row_len[k] = count;
			} else {
				atomicOr(&a.ctr[CTR_STATUS], 1);
				a.row_len[k] = -1;
			}
		}
		st_done += fits ? 1 : 0;
	}
	drain_vmem();
	if (lane == 0) {
		atomicAdd(&a.ctr64[C64_ELIM], st_elim);
		atomicAdd(&a.ctr64[C64_STREAM], st_stream);
		atomicAdd(&a.ctr64[C64_INPUT], st_input);
		atomicAdd(&a.ctr[a.done_ctr], st_done);
	}
}

// slot geometry: xp (rpad accumulators) | bitmap (rpad/32 words) | xn (Sm accumulators), 256-byte aligned parts
void wave_dense_geometry(int rpad, int Sm, bool wide, int64_t *slot_bytes, int64_t *off_bm, int64_t *off_xn)
{
	const int64_t vb = wide ? 8 : 4;
	auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
	*off_bm = up((int64_t) rpad * vb);
	*off_xn = *off_bm + up((int64_t) (rpad / 32 + 1) * 4);
	*slot_bytes = *off_xn + up((int64_t) Sm * vb) + 256;
}

void launch_schur_wave_dense(const SchurArgs &a, unsigned char *scratch, int64_t slot_bytes, int64_t off_bm, int64_t off_xn,
                             bool wide, uint32_t *dense_out, int64_t ldS, int blocks, hipStream_t stream)
{
	WaveDenseArgs d;
	d.a = a;
	d.scratch = scratch;
	d.slot_bytes = slot_bytes;
	d.off_bm = off_bm;
	d.off_xn = off_xn;
	d.dense_out = dense_out;
	d.ldS = ldS;
	if (wide)
		hipLaunchKernelGGL((schur_wave_dense_kernel<true>), dim3(blocks), dim3(64), 0, stream, d);
	else
		hipLaunchKernelGGL((schur_wave_dense_kernel<false>), dim3(blocks), dim3(64), 0, stream, d);
	HIP_CHECK(hipGetLastError());
}

}  // namespace sh

// --------------------------------------------------------------------------
// Regrouping for the row-group kernel when the pivot graph has several connected components: rows that are
// neighbours in the list but lie in different components share nothing and waste its lanes.  The rows are sorted by
// (component their pivotal entries fall in, position in the list): rows that can share eliminations come together,
// and inside a component the order of the list -- often meaningful -- is kept.  Rows with nothing to eliminate go
// last.  A bitonic sort of 64-bit keys in HBM (about 1 ms for 160,000 rows).
// --------------------------------------------------------------------------
namespace sh {

__global__ __launch_bounds__(256) void row_component_key_kernel(SchurArgs a, unsigned long long *keys, int npad)
{
	const int k = blockIdx.x * 256 + threadIdx.x;
	if (k >= npad)
		return;
	unsigned long long key = ~0ull;                   // padding
	if (k < a.nrows) {
		uint32_t comp = (uint32_t) a.r;               // nothing pivotal
		const int i = a.rows[k];
		for (int64_t px = a.Ap[i]; px < a.Ap[i + 1]; px++) {
			const uint32_t c = a.lab[a.Aj[px]];
			if (c < (uint32_t) a.r) {
				const uint32_t cc = a.comp[c];
				comp = (cc < comp) ? cc : comp;
			}
		}
		key = ((unsigned long long) comp << 32) | (unsigned int) k;
	}
	keys[k] = key;
}

__global__ __launch_bounds__(256) void bitonic_step_kernel(unsigned long long *keys, int npad, int size, int stride)
{
	const int t = blockIdx.x * 256 + threadIdx.x;
	const int partner = t ^ stride;
	if (t >= npad || partner <= t)
		return;
	const unsigned long long x = keys[t], y = keys[partner];
	const bool ascending = (t & size) == 0;
	if ((x > y) == ascending) {
		keys[t] = y;
		keys[partner] = x;
	}
}

// all the steps of one merge stage whose partners lie inside a block of 2048 keys (stride <= 1024), in LDS
constexpr int BITONIC_LOCAL = 2048;
__global__ __launch_bounds__(256) void bitonic_local_kernel(unsigned long long *keys, int npad, int size, int first_stride)
{
	__shared__ unsigned long long s[BITONIC_LOCAL];
	const int base = blockIdx.x * BITONIC_LOCAL;
	for (int t = threadIdx.x; t < BITONIC_LOCAL; t += 256)
		s[t] = (base + t < npad) ? keys[base + t] : ~0ull;
	__syncthreads();
	for (int stride = first_stride; stride > 0; stride >>= 1) {
		for (int q = threadIdx.x; q < BITONIC_LOCAL / 2; q += 256) {
			// the q-th pair of this step: t has bit `stride` clear
			const int t = ((q & ~(stride - 1)) << 1) | (q & (stride - 1));
			const int partner = t | stride;
			const unsigned long long x = s[t], y = s[partner];
			const bool ascending = ((base + t) & size) == 0;
			if ((x > y) == ascending) {
				s[t] = y;
				s[partner] = x;
			}
		}
		__syncthreads();
	}
	for (int t = threadIdx.x; t < BITONIC_LOCAL; t += 256)
		if (base + t < npad)
			keys[base + t] = s[t];
}

__global__ __launch_bounds__(256) void row_order_kernel(const unsigned long long *keys, int nrows, int *order)
{
	const int t = blockIdx.x * 256 + threadIdx.x;
	if (t < nrows)
		order[t] = (int) (unsigned int) keys[t];          // (the nrows real keys sort before the padding)
}

// sortbuf: 64-bit keys, padded to a power of two
int64_t regroup_scratch_ints(int nrows, int r)
{
	(void) r;
	int64_t npad = 1;
	while (npad < nrows)
		npad <<= 1;
	return 2 * npad + 8;
}

void launch_regroup_rows(const SchurArgs &a, int *sortbuf, int *order, hipStream_t stream)
{
	int npad = 1;
	while (npad < a.nrows)
		npad <<= 1;
	unsigned long long *keys = reinterpret_cast<unsigned long long *>(((uintptr_t) sortbuf + 7) & ~(uintptr_t) 7);
	const dim3 grid((npad + 255) / 256), block(256);
	hipLaunchKernelGGL(row_component_key_kernel, grid, block, 0, stream, a, keys, npad);
	for (int size = 2; size <= npad; size <<= 1) {
		int stride = size >> 1;
		for (; stride >= BITONIC_LOCAL; stride >>= 1)
			hipLaunchKernelGGL(bitonic_step_kernel, grid, block, 0, stream, keys, npad, size, stride);
		// (npad is a power of two: either one partial block or whole blocks)
		hipLaunchKernelGGL(bitonic_local_kernel, dim3((npad + BITONIC_LOCAL - 1) / BITONIC_LOCAL), block, 0, stream, keys, npad, size, stride);
	}
	hipLaunchKernelGGL(row_order_kernel, dim3((a.nrows + 255) / 256), block, 0, stream, keys, a.nrows, order);
	HIP_CHECK(hipGetLastError());
}

}  // namespace sh

// --------------------------------------------------------------------------
// K2g: "row-group" kernel.  A wave owns 64 CONSECUTIVE rows of the row list,
// lane = row.  The accumulators of the group are stored label-major:
// X[label][64] (one 256-byte line per label), so that eliminating pivot c in
// all the rows that hold it is a wave-uniform loop over the entries of U'[c]
// and every update is ONE coalesced 256-byte atomic instruction (lanes whose
// row does not hold c are masked).  Rows that are neighbours in the matrix
// have almost the same reach (mk13.b5: the union of 16 reaches is 1.07x one
// reach), which is what makes this layout pay: the random-access traffic of
// the per-row kernel is divided by the number of rows sharing a line.
// Pending pivots of the group = a bitmap in HBM, touched by one lane.
// --------------------------------------------------------------------------
namespace sh {

namespace {
constexpr int GR_ACT = 512;             // pending labels gathered per pass (GR_ACT / 32 bitmap words)
constexpr int GR_LBM_MAX_BYTES = 48 * 1024;   // largest pending bitmap kept in LDS
constexpr int GR_RB = 8;            // accumulator lines are loaded one block of this many pivots ahead
}

struct GroupArgs {
	SchurArgs a;
	int watch;                // 1: give up when the lane efficiency is hopeless (CTR_GROUP_ABORT)
	float min_eff;            // ... below this fraction of active lanes per applied pivot
	unsigned long long min_w; // ... judged once this many pivots have been applied over all groups
	unsigned char *scratch;
	int64_t slot_bytes;       // X | bitmap
	int64_t off_bm;
	uint32_t *dense_out;
	int64_t ldS;
	int touched_lds;          // LBM only: the bits of the non-pivotal labels sit in LDS too (they fit), else in HBM
};

// LBM: the pending bitmap of the group lives in LDS (rpad / 8 bytes, dynamic) instead of HBM.
#ifdef SPASM_GROUP_PROFILE
#define GR_TICK(q) do { const unsigned long long t_now = __builtin_readcyclecounter(); prof[q] += t_now - t_last; t_last = t_now; } while (0)
#else
#define GR_TICK(q) do { } while (0)
#endif

// NW: waves per row group (workgroup = 64 NW threads).  Every wave has lane = row; the waves share the pending
// bitmap and split the pivots of a level (and the input entries, and the output chunks) between them: a group
// is a chain of ~1000 dependent level rounds, and when there are fewer groups than the chip has room for, the
// length of that chain is the run time.  Wave 0 finds the level and lists its pending pivots while the others
// wait for their atomics to land; two workgroup barriers per round.
template <bool WIDE, bool LBM, int NW>
__global__ __launch_bounds__(64 * NW) void schur_group_kernel(GroupArgs d)
{
	// several waves of the workgroup update the same lines and bitmap words when NW > 1
	constexpr int SCOPE = (NW > 1) ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_WAVEFRONT;
	using V = typename Acc<WIDE>::type;
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
	__shared__ uint32_t act[GR_ACT * NW];      // [0, GR_ACT): pending pivots of the level; output: one list per wave
	__shared__ int ctl[8];
	__shared__ int cnt_w[NW][64];
	const SchurArgs &a = d.a;
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	// workgroup barrier that waits for the LDS queue only (__syncthreads would drain the atomics in flight too)
	auto wg_sync = [&]() {
		if (NW > 1)
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		else {
			__builtin_amdgcn_wave_barrier();
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
	};
	ring_reserve<WIDE>();
	const uint32_t r = (uint32_t) a.r;
	const int Sm = a.Sm;
	const MontDev F = a.F;
	const int nw = (int) (r / 32);
	const int ngroups = (a.nrows + 63) / 64;


	unsigned char *slot = d.scratch + (int64_t) blockIdx.x * d.slot_bytes;
	V *X = reinterpret_cast<V *>(slot);                 // X[label * 64 + lane]
	// LDS is addressed through address_space(3) pointers: a generic pointer would turn every access into a
	// FLAT instruction, which counts in vmcnt and makes the wave wait for all its outstanding atomics
	typedef uint32_t __attribute__((address_space(3))) lds_u32;
	lds_u32 *const act_l = (lds_u32 *) act;
	volatile int __attribute__((address_space(3))) *const ctl_l = (volatile int __attribute__((address_space(3))) *) ctl;
	lds_u32 *const bm_l = (lds_u32 *) lds_dyn;
	uint32_t *const bm_g = reinterpret_cast<uint32_t *>(slot + d.off_bm);
	if (LBM) {
		for (int w = threadIdx.x; w < nw + (d.touched_lds ? (Sm + 31) / 32 : 0); w += 64 * NW)
			bm_l[w] = 0;
		wg_sync();
	}
	auto bm_load = [&](int w) -> uint32_t { return LBM ? *(volatile lds_u32 *) (bm_l + w) : ld_sc1(&bm_g[w]); };
	auto bm_or = [&](uint32_t c) {
		if (LBM)
			(void) __hip_atomic_fetch_or(bm_l + (c >> 5), 1u << (c & 31), __ATOMIC_RELAXED, SCOPE);
		else
			(void) __hip_atomic_fetch_or(&bm_g[c >> 5], 1u << (c & 31), __ATOMIC_RELAXED, SCOPE);
	};
	auto bm_clear = [&](int w, uint32_t bits) {
		if (LBM)
			(void) __hip_atomic_fetch_and(bm_l + w, ~bits, __ATOMIC_RELAXED, SCOPE);
		else
			(void) __hip_atomic_fetch_and(&bm_g[w], ~bits, __ATOMIC_RELAXED, SCOPE);
	};
	auto act_load = [&](int t) -> uint32_t { return *(volatile lds_u32 *) (act_l + t); };
	// non-pivotal labels (>= r) that received something: bits r.. of the same bitmap; the output only visits
	// those lines
	const bool tl = LBM && d.touched_lds != 0;
	auto touch = [&](uint32_t c) {
		if (tl)
			(void) __hip_atomic_fetch_or(bm_l + (c >> 5), 1u << (c & 31), __ATOMIC_RELAXED, SCOPE);
		else
			(void) __hip_atomic_fetch_or(&bm_g[c >> 5], 1u << (c & 31), __ATOMIC_RELAXED, SCOPE);
	};
	auto touched_load = [&](int w) -> uint32_t { return tl ? *(volatile lds_u32 *) (bm_l + nw + w) : ld_sc1(&bm_g[nw + w]); };
	auto touched_clear = [&](int w) {
		if (tl)
			bm_l[nw + w] = 0;
		else
			bm_g[nw + w] = 0;
	};
	const int nwS = (Sm + 31) / 32;

	unsigned long long st_elim = 0, st_stream = 0, st_input = 0, st_wavepiv = 0;
	int st_done = 0;
	LArena larena;
#ifdef SPASM_GROUP_PROFILE
	unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long t_last = __builtin_readcyclecounter();
#endif

	for (;;) {
		if (threadIdx.x == 0) {
			ctl_l[0] = atomicAdd(&a.ctr[a.next_ctr], 1);
			ctl_l[1] = d.watch ? __hip_atomic_load(&a.ctr[CTR_GROUP_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
		}
		wg_sync();
		const int g = __builtin_amdgcn_readfirstlane(ctl_l[0]);
		if (g >= ngroups || __builtin_amdgcn_readfirstlane(ctl_l[1]) != 0)
			break;
		const int kpos = g * 64 + lane;
		const bool have_row = kpos < a.nrows;
		const int k = have_row ? (a.order != nullptr ? a.order[kpos] : kpos) : 0;       // position in the row list
		const int row_to_record = (a.L_i != nullptr && have_row) ? a.row_orig[k] : 0;

		// ---- scatter the 64 input rows (each lane its own) ----
		if (have_row) {
			const int i = a.rows[k];
			const int64_t lo = a.Ap[i], hi = a.Ap[i + 1];
			if (wv == 0)
				st_input += (unsigned long long) (hi - lo);
			for (int64_t px = lo + wv; px < hi; px += NW) {
				const uint32_t c = a.lab[a.Aj[px]];
				const uint32_t v = reduce_sum(from_balanced(a.Ax[px], F), F);
				add_ff<V, SCOPE>(&X[(int64_t) c * 64 + lane], v);
				if (c < r)
					bm_or(c);
				else
					touch(c);
			}
		}

		// ---- eliminate level by level ----
		// ctl[2]: 0 = apply the pivots listed in act[], 1 = no pivot left, 2 = the batch was abandoned
		// ctl[3]: number of listed pivots; ctl[4]: 1 = another chunk of the same level follows
		uint32_t cursor = 0;          // (wave 0)
		bool abandoned = false;
		int round = 0;
		// wave 0: state of the level being worked on, between the chunks of a wide level
		int wi = 0, wl = 0, wb = 0;
		uint32_t c0 = 0, lend = 0;
		bool more_chunks = false;
		for (;;) {
			GR_TICK(5);
			// barrier A: every wave's pending marks are in the bitmap and act[] is free.  With the bitmap in LDS
			// the atomics only have to land before the next accumulator lines are read (barrier B): they
			// complete while wave 0 looks for the level.
			if (!LBM)
				drain_vmem();
			wg_sync();
			GR_TICK(0);
			bool published = false;
			if (d.watch && st_wavepiv >= 256 / NW) {
				// publish progress every 256 applied pivots of the group; whoever publishes judges the batch: all
				// running groups contribute in proportion to their work, so the ratio is not biased towards the
				// cheap groups that finish first
				if (lane == 0) {
					const unsigned long long e = atomicAdd(&a.ctr64[C64_ELIM], st_elim) + st_elim;
					const unsigned long long w = atomicAdd(&a.ctr64[C64_WAVEPIV], st_wavepiv) + st_wavepiv;
					atomicAdd(&a.ctr64[C64_STREAM], st_stream);
					if (w > d.min_w && (double) e < (double) d.min_eff * 64.0 * (double) w)
						atomicOr(&a.ctr[CTR_GROUP_ABORT], 1);
				}
				st_elim = 0;
				st_wavepiv = 0;
				st_stream = 0;
				published = true;
			}
			if (wv == 0) {
				int status = 0;
				if (!more_chunks) {
					if (d.watch) {
						// (a global load per round would add its latency to the chain of the group)
						int stop = 0;
						if (lane == 0 && (published || (round & 15) == 0))
							stop = __hip_atomic_load(&a.ctr[CTR_GROUP_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						round += 1;
						if (__builtin_amdgcn_readfirstlane(stop) != 0)
							status = 2;
					}
					GR_TICK(1);
					if (status == 0) {
						wi = -1;
						uint32_t fbits = 0;
						const int wstart = (int) (cursor >> 5);
						for (int base = wstart; base < nw; base += 64) {
							const int w = base + lane;
							uint32_t bits = 0;
							if (w < nw) {
								bits = bm_load(w);
								if (w == wstart)
									bits &= ~((1u << (cursor & 31)) - 1u);
							}
							const uint64_t mask = __ballot(bits != 0);
							if (mask != 0) {
								const int fl = __builtin_ctzll(mask);
								wi = base + fl;
								fbits = (uint32_t) __shfl((int) bits, fl);
								break;
							}
						}
						GR_TICK(2);
						if (wi < 0) {
							status = 1;
						} else {
							c0 = (uint32_t) wi * 32 + (uint32_t) __builtin_ctz(fbits);
							// (scalar loads: a vector load would return after every atomic issued before it)
							typedef const uint32_t __attribute__((address_space(4))) *const_u32_ptr;
							const uint32_t lwe = ((const_u32_ptr) (uintptr_t) a.lvl_end_w)[wi];
							lend = (lwe != MIXED) ? lwe * 32 : ((const_u32_ptr) (uintptr_t) a.lvl_end)[c0];
							wl = (int) ((lend + 31) >> 5);
							wb = wi;
#ifdef SPASM_GROUP_PROFILE
							asm volatile("" : : "v"(lend) : "memory");
#endif
						}
					}
					GR_TICK(3);
				}
				int tot = 0;
				if (status == 0) {
					// pending labels of this chunk of the level -> act[]: lane l takes byte l % 4 of word wb + l / 4
					const int w = wb + (lane >> 2);
					const int sh = (lane & 3) * 8;
					uint32_t bits = 0;
					if (w < wl) {
						bits = bm_load(w);
						if (w == wi)
							bits &= ~((1u << (c0 & 31)) - 1u);
						if ((uint32_t) w * 32 + 32 > lend)
							bits &= (1u << (lend & 31)) - 1u;
						bits &= 0xFFu << sh;
						if (bits != 0)
							bm_clear(w, bits);
					}
					int pos = wave_exclusive_scan_small(__popc(bits), tot);
					uint32_t b = bits;
					while (b) {
						const int bit = __builtin_ctz(b);
						b &= b - 1;
						act_l[pos++] = (uint32_t) w * 32 + bit;
					}
					wb += GR_ACT / 32;
					more_chunks = wb < wl;
					if (!more_chunks)
						cursor = lend;
				}
				if (lane == 0) {
					ctl_l[2] = status;
					ctl_l[3] = tot;
				}
				GR_TICK(4);
			}
			// barrier B: the list is there, and so are the updates of the previous level: every wave has waited
			// for its own atomics (pivots of one level never update each other's lines, so this wait is only
			// needed once per level; it costs nothing when it has nothing to wait for)
			if (LBM)
				drain_vmem();
			wg_sync();
			GR_TICK(0);
			const int status = __builtin_amdgcn_readfirstlane(ctl_l[2]);
			if (status != 0) {
				abandoned = status == 2;
				break;
			}
			const int tot = __builtin_amdgcn_readfirstlane(ctl_l[3]);
			// this wave's share of the list
			const int share = (tot + NW - 1) / NW;
			const int t_lo = wv * share;
			const int t_hi = min(tot, t_lo + share);
			{
				// apply them: uniform loop over blocks of GR_RB pivots.  Everything a block needs is fetched with
				// vector loads one block ahead, into the two halves of the accumulation-register ring
				// (software-managed vmcnt, above), so that the atomics keep issuing back to back behind the loads:
				//   regs 0-3  row extents rp[c], rp[c+1]: lane u holds those of pivot u
				//   reg  4    the first four entries of every row: lane 8u + d holds dword d of the head of pivot u
				//   regs 6..  the accumulator lines X[c][lane]
				// Wave-uniform values are then picked out of these with v_readlane.  `issued` counts (a subset of)
				// the vector memory instructions issued so far.
				constexpr int NL = WIDE ? 2 : 1;
				constexpr int HALF_REGS = 6 + GR_RB * NL;
				static_assert(GR_RB == 8, "one head load covers 8 pivots x 8 dwords = 64 lanes");
				int issued = 0;
				int base_blk[2] = {0, 0};
				auto issue_block = [&](int b0, auto half) -> uint32_t {
					constexpr int R0 = decltype(half)::value * HALF_REGS;
					base_blk[decltype(half)::value] = issued;
					const int th = b0 + (lane >> 3), tr = b0 + (lane & 7);
					const uint32_t ch = (th < t_hi) ? act_load(th) : 0xFFFFFFFFu;      // pivot of this lane's head dword
					const uint32_t cr = (tr < t_hi) ? act_load(tr) : 0u;
					ring_issue_x4<R0>(a.rp + cr);
					ring_issue_ro<R0 + 4>(reinterpret_cast<const uint32_t *>(a.head) + (int64_t) (ch != 0xFFFFFFFFu ? ch : 0u) * 8 + (lane & 7));
					static_for<0, GR_RB>([&](auto uu) {
						constexpr int u = decltype(uu)::value;
						// slots past the end load line 0 (harmless, keeps the instruction count static)
						uint32_t c = __builtin_amdgcn_readlane(ch, 8 * u);
						c = (c != 0xFFFFFFFFu) ? c : 0u;
						ring_issue<R0 + 6 + u * NL>(&X[(int64_t) c * 64 + lane]);
					});
					issued += GR_RB + 2;
					return ch;
				};
				auto process_block = [&](int b0, auto half, const uint32_t ch) {
					constexpr int R0 = decltype(half)::value * HALF_REGS;
					constexpr int OV0 = 2 * HALF_REGS;       // 8 x 2 registers: entries 4.. of long rows
					const int base = base_blk[decltype(half)::value];
					// one wait for the whole block: its last line (slots past the end were loaded too) and
					// everything older.  The block was issued while the previous one was applied.
					wait_vm_at_most(issued - (base + 2 + GR_RB - 1) - 1);
					uint32_t hv;
					unsigned long long rp_lo, rp_hi;
					ring_read<R0 + 4>(hv);
					ring_read<R0>(rp_lo);
					ring_read<R0 + 2>(rp_hi);
					// pass 1: the multipliers (lane = row) of the 8 pivots, computed side by side (independent
					// chains of multiplications); long rows of applied pivots start fetching their entries 4.. now,
					// 64 per instruction (lane e holds entry 4 + e)
					V raw[GR_RB];
					uint32_t vv[GR_RB];
					static_for<0, GR_RB>([&](auto uu) {
						constexpr int u = decltype(uu)::value;
						ring_read<R0 + 6 + u * NL>(raw[u]);
					});
#pragma unroll
					for (int u = 0; u < GR_RB; u++)
						vv[u] = reduce_sum(raw[u], F);          // (0 stays 0)
					int seq_ov[GR_RB];
					uint32_t applied = 0;        // bit u: pivot u of the block was applied to some row
					static_for<0, GR_RB>([&](auto uu) {
						constexpr int u = decltype(uu)::value;
						seq_ov[u] = 0;
						if (b0 + u >= t_hi) {
							vv[u] = 0;       // (slots past the end hold line 0)
							return;
						}
						const uint32_t c = __builtin_amdgcn_readlane(ch, 8 * u);
						if (__ballot(raw[u] != 0) != 0) {
							if (raw[u] != 0)
								__hip_atomic_store(&X[(int64_t) c * 64 + lane], (V) 0, __ATOMIC_RELAXED, SCOPE);
							issued += 1;
						}
						const uint32_t v = vv[u];
						const uint64_t active = __ballot(v != 0);
						if (active == 0)
							return;
						applied |= 1u << u;
						const int nact = __popcll(active);
#ifdef SPASM_GROUP_PROFILE
						// 64-byte quarters of the line with an active lane (x 1000, on top of the watch phase)
						prof[1] += 1000ull * (((active & 0xFFFFull) != 0) + ((active & 0xFFFF0000ull) != 0) +
						                      ((active & 0xFFFF00000000ull) != 0) + ((active >> 48) != 0));
						prof[2] += 1000ull * ((nact + 15) / 16);      // (on top of the bitmap scan: quarters if the active lanes were packed)
#endif
						st_wavepiv += 1;
						st_elim += (unsigned long long) nact;
						if (a.L_i != nullptr)
							record_L(a, larena, v != 0, row_to_record, c, v, lane, F);
						if (__builtin_amdgcn_readlane(hv, 8 * u + 6) != 0xFFFFFFFFu) {
							// four head entries: the row may go on
							const uint32_t s_lo = __builtin_amdgcn_readlane((uint32_t) rp_lo, u);
							const uint32_t s_hi = __builtin_amdgcn_readlane((uint32_t) (rp_lo >> 32), u);
							const int L = (int) (__builtin_amdgcn_readlane((uint32_t) rp_hi, u) - s_lo);
							if (L > 4) {
								const uint64_t s0 = ((uint64_t) s_hi << 32) | s_lo;
								seq_ov[u] = issued;
								ring_issue_x2_ro<OV0 + 2 * u>(a.ent + s0 + (uint64_t) min(4 + lane, L - 1));
								issued += 1;
							}
						}
						st_stream += (unsigned long long) nact * 4ull;       // (rows longer than the head add theirs below)
					});
					// pass 2: the updates.  32-bit sums take the products unreduced, in (0, 2p): they are reduced
					// when read.
					static_for<0, GR_RB>([&](auto uu) {
						constexpr int u = decltype(uu)::value;
						if (((applied >> u) & 1u) == 0)
							return;
						const uint32_t v = vv[u];
						const uint32_t w_neg = F.p - v;
						// the four products first (independent), then the atomics under one exec mask
						uint32_t tgt[4], prod[4];
						int nvalid = 0;
#pragma unroll
						for (int q = 0; q < 4; q++) {
							tgt[q] = __builtin_amdgcn_readlane(hv, 8 * u + 2 * q);
							const uint32_t val = __builtin_amdgcn_readlane(hv, 8 * u + 2 * q + 1);
							prod[q] = WIDE ? montmul(w_neg, val, F) : montmul_lazy(w_neg, val, F);
							nvalid += (tgt[q] != 0xFFFFFFFFu) ? 1 : 0;
						}
						if (v != 0) {
#pragma unroll
							for (int q = 0; q < 4; q++)
								if (tgt[q] != 0xFFFFFFFFu)
									add_ff<V, SCOPE>(&X[(int64_t) tgt[q] * 64 + lane], prod[q]);
						}
						issued += nvalid;      // some lane has v != 0: the instructions are issued
						if (nvalid < 4)
							return;
						const uint32_t s_lo = __builtin_amdgcn_readlane((uint32_t) rp_lo, u);
						const uint32_t s_hi = __builtin_amdgcn_readlane((uint32_t) (rp_lo >> 32), u);
						const int L = (int) (__builtin_amdgcn_readlane((uint32_t) rp_hi, u) - s_lo);
						if (L <= 4)
							return;
						const uint64_t s0 = ((uint64_t) s_hi << 32) | s_lo;
						const int nact = __popcll(__ballot(v != 0));
						st_stream += (unsigned long long) nact * (unsigned long long) (L - 4);
						wait_vm_at_most(issued - seq_ov[u] - 1);
						for (int done = 4;;) {
							uint32_t ot, oc;
							ring_read2<OV0 + 2 * u>(ot, oc);
							const int n = min(L - done, 64);
							for (int e = 0; e < n; e++) {
								const uint32_t t = __builtin_amdgcn_readlane(ot, e);
								const uint32_t val = __builtin_amdgcn_readlane(oc, e);
								if (v != 0)
									add_ff<V, SCOPE>(&X[(int64_t) t * 64 + lane], WIDE ? montmul(w_neg, val, F) : montmul_lazy(w_neg, val, F));
							}
							issued += n;
							if (lane < n) {
								if (ot < r)
									bm_or(ot);
								else
									touch(ot);
							}
							done += n;
							if (done >= L)
								break;
							// rows with more than 68 entries: the next 64, through the same registers
							ring_issue_x2_ro<OV0 + 2 * u>(a.ent + s0 + (uint64_t) min(done + lane, L - 1));
							issued += 1;
							wait_vm_at_most(0);
						}
					});
					// the targets among the heads of the applied pivots become pending (pivotal ones) or touched:
					// one LDS and one global instruction for the block (lane 8u + 2q holds target q of pivot u)
					if (applied != 0 && (lane & 1) == 0 && ((applied >> (lane >> 3)) & 1u) && hv != 0xFFFFFFFFu) {
						if (hv < r)
							bm_or(hv);
						else
							touch(hv);
					}
				};
				using H0 = std::integral_constant<int, 0>;
				using H1 = std::integral_constant<int, 1>;
				if (t_lo < t_hi) {
					uint32_t chA = issue_block(t_lo, H0{}), chB = 0;
					for (int b0 = t_lo; b0 < t_hi; b0 += 2 * GR_RB) {
						if (b0 + GR_RB < t_hi)
							chB = issue_block(b0 + GR_RB, H1{});
						process_block(b0, H0{}, chA);
						if (b0 + GR_RB >= t_hi)
							break;
						if (b0 + 2 * GR_RB < t_hi)
							chA = issue_block(b0 + 2 * GR_RB, H0{});
						process_block(b0 + GR_RB, H1{}, chB);
					}
				}
			}
		}
		GR_TICK(5);
		drain_vmem();
		wg_sync();
		GR_TICK(0);
		if (abandoned) {
			// the batch went to the per-row kernels: restore the all-zero state of this slice and leave
			// (rows keep row_len == -1)
			// Only lines whose bit is set can be non-zero: a line is marked when it is written, and a pending pivot
			// whose bit was cleared has been consumed (and zeroed) within the same round -- rounds are never cut short.
			const int nwt = nw + nwS;
			for (int wb0 = wv * 64; wb0 < nwt; wb0 += 64 * NW) {
				const int w = wb0 + lane;
				const uint32_t bits = (w < nwt) ? (w < nw ? bm_load(w) : touched_load(w - nw)) : 0u;
				unsigned long long holders = __ballot(bits != 0);
				while (holders != 0) {
					const int l = __builtin_ctzll(holders);
					holders &= holders - 1;
					uint32_t b = (uint32_t) __shfl((int) bits, l);
					const int wq = wb0 + l;
					const int64_t first = (wq < nw) ? (int64_t) wq * 32 : (int64_t) r + (int64_t) (wq - nw) * 32;
					while (b != 0) {
						const int bit = __builtin_ctz(b);
						b &= b - 1;
						X[(first + bit) * 64 + lane] = 0;
					}
				}
			}
			wg_sync();
			for (int w = threadIdx.x; w < nw; w += 64 * NW) {
				if (LBM)
					bm_l[w] = 0;
				else
					bm_g[w] = 0;
			}
			for (int w = threadIdx.x; w < nwS; w += 64 * NW)
				touched_clear(w);
			drain_vmem();
			break;
		}

		// ---- output: lane = row, the non-pivotal labels in order are the sorted row ----
		V *Xn = X + (int64_t) r * 64;
		if (d.dense_out != nullptr) {
			for (int t0 = 16 * wv; t0 < Sm; t0 += 16 * NW) {
				V rv[16];
#pragma unroll
				for (int u = 0; u < 16; u++)
					rv[u] = (t0 + u < Sm) ? ld_sc1(&Xn[(int64_t) (t0 + u) * 64 + lane]) : (V) 0;
#pragma unroll
				for (int u = 0; u < 16; u++) {
					if (t0 + u >= Sm)
						continue;
					if (have_row)
						d.dense_out[(int64_t) k * d.ldS + t0 + u] = reduce_sum(rv[u], F);
					if (rv[u] != 0)
						__hip_atomic_store(&Xn[(int64_t) (t0 + u) * 64 + lane], (V) 0, __ATOMIC_RELAXED,
						                   SCOPE);
				}
			}
			for (int w = threadIdx.x; w < nwS; w += 64 * NW)
				touched_clear(w);
			if (have_row && wv == 0)
				a.row_len[k] = Sm;
			if (wv == 0)
				st_done += __popcll(__ballot(have_row));
			continue;
		}
		GR_TICK(5);
		// two passes over the touched lines only (a few percent of the Sm non-pivotal labels): count, then write.
		// Chunks of 512 labels, a contiguous range of chunks per wave; the touched labels of a chunk are listed in
		// this wave's part of act[] (lane l: byte l % 4 of word l / 4).
		lds_u32 *const my_act = act_l + wv * GR_ACT;
		auto list_touched = [&](int wb) -> int {
			const int w = wb + (lane >> 2);
			uint32_t bits = 0;
			if (w < nwS)
				bits = touched_load(w) & (0xFFu << ((lane & 3) * 8));
			int tot;
			int pos = wave_exclusive_scan_small(__popc(bits), tot);
			uint32_t b = bits;
			while (b) {
				const int bit = __builtin_ctz(b);
				b &= b - 1;
				my_act[pos++] = (uint32_t) w * 32 + bit;
			}
			__builtin_amdgcn_wave_barrier();
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // LDS only: a fence would also wait for the atomics in flight
			return __builtin_amdgcn_readfirstlane(tot);
		};
		auto my_act_load = [&](int t) -> uint32_t { return *(volatile lds_u32 *) (my_act + t); };
		constexpr int OB = 16;
		constexpr int CW = GR_ACT / 32;            // words per chunk
		const int nchunks = (nwS + CW - 1) / CW;
		const int ch_per = (nchunks + NW - 1) / NW;
		const int wb_lo = wv * ch_per * CW, wb_hi = min(nwS, (wv + 1) * ch_per * CW);
		int count = 0;
		for (int wb = wb_lo; wb < wb_hi; wb += CW) {
			const int tot = list_touched(wb);
			for (int b0 = 0; b0 < tot; b0 += OB) {
				V rv[OB];
#pragma unroll
				for (int u = 0; u < OB; u++) {
					const uint32_t t = (b0 + u < tot) ? __builtin_amdgcn_readfirstlane(my_act_load(b0 + u)) : 0xFFFFFFFFu;
					rv[u] = (t != 0xFFFFFFFFu) ? ld_sc1(&Xn[(int64_t) t * 64 + lane]) : (V) 0;
				}
#pragma unroll
				for (int u = 0; u < OB; u++)
					count += (rv[u] != 0 && reduce_sum(rv[u], F) != 0) ? 1 : 0;
			}
			__builtin_amdgcn_wave_barrier();
		}
		GR_TICK(6);
		// row lengths = sums over the waves; the entries of wave w come after those of the waves before it
		int before = 0, row_count = count;
		if (NW > 1) {
			((volatile int __attribute__((address_space(3))) *) &cnt_w[0][0])[wv * 64 + lane] = count;
			wg_sync();
			row_count = 0;
#pragma unroll
			for (int w2 = 0; w2 < NW; w2++) {
				const int cw = ((volatile int __attribute__((address_space(3))) *) &cnt_w[0][0])[w2 * 64 + lane];
				row_count += cw;
				before += (w2 < wv) ? cw : 0;
			}
		}
		int gtot;
		const int excl = wave_exclusive_scan(row_count, lane, gtot);
		if (wv == 0) {
			unsigned long long got = 0;
			if (lane == 0) {
				got = atomicAdd(&a.ctr64[C64_POOL], (unsigned long long) gtot);
				ctl_l[5] = (int) (uint32_t) got;
				ctl_l[6] = (int) (uint32_t) (got >> 32);
			}
		}
		wg_sync();
		const uint32_t g_lo = (uint32_t) __builtin_amdgcn_readfirstlane(ctl_l[5]);
		const uint32_t g_hi = (uint32_t) __builtin_amdgcn_readfirstlane(ctl_l[6]);
		const int64_t base_off = (int64_t) (((uint64_t) g_hi << 32) | g_lo);
		const bool fits = base_off + gtot <= a.pool_cap;
		int64_t wpos = base_off + excl + before;
		for (int wb = wb_lo; wb < wb_hi; wb += CW) {
			const int tot = list_touched(wb);
			if ((lane & 3) == 0 && wb + (lane >> 2) < nwS)
				touched_clear(wb + (lane >> 2));
			for (int b0 = 0; b0 < tot; b0 += OB) {
				V rv[OB];
				uint32_t tt[OB];
#pragma unroll
				for (int u = 0; u < OB; u++) {
					tt[u] = (b0 + u < tot) ? __builtin_amdgcn_readfirstlane(my_act_load(b0 + u)) : 0xFFFFFFFFu;
					rv[u] = (tt[u] != 0xFFFFFFFFu) ? ld_sc1(&Xn[(int64_t) tt[u] * 64 + lane]) : (V) 0;
				}
#pragma unroll
				for (int u = 0; u < OB; u++) {
					if (rv[u] == 0)
						continue;
					__hip_atomic_store(&Xn[(int64_t) tt[u] * 64 + lane], (V) 0, __ATOMIC_RELAXED, SCOPE);
					const uint32_t v = reduce_sum(rv[u], F);
					if (v != 0 && fits) {
						a.pool_j[wpos] = a.q[tt[u]];
						a.pool_x[wpos] = to_balanced(v, F);
						wpos += 1;
					}
				}
			}
			__builtin_amdgcn_wave_barrier();
		}
		GR_TICK(7);
		if (wv == 0) {
			if (have_row) {
				if (fits) {
					a.row_off[k] = (base_off + excl) | (1LL << 62);
					a.row_len[k] = row_count;
				} else {
					a.row_len[k] = -1;
				}
			}
			if (!fits && lane == 0)
				atomicOr(&a.ctr[CTR_STATUS], 1);
			st_done += fits ? __popcll(__ballot(have_row)) : 0;
		}
	}
	drain_vmem();
#ifdef SPASM_GROUP_PROFILE
	GR_TICK(7);
	if (threadIdx.x == 0)
		for (int q = 0; q < 8; q++)
			atomicAdd(&a.ctr64[C64_PROF0 + q], prof[q]);
#endif
	// per-lane statistics -> wave totals
	for (int dlt = 32; dlt >= 1; dlt >>= 1) {
		const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) st_input, dlt);
		const uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (st_input >> 32), dlt);
		st_input += ((unsigned long long) hi << 32) | lo;
	}
	if (lane == 0) {
		atomicAdd(&a.ctr64[C64_ELIM], st_elim);
		atomicAdd(&a.ctr64[C64_STREAM], st_stream);
		atomicAdd(&a.ctr64[C64_INPUT], st_input);
		atomicAdd(&a.ctr64[C64_WAVEPIV], st_wavepiv);
		atomicAdd(&a.ctr[a.done_ctr], st_done);
	}
}

void group_geometry(int rpad, int Sm, bool wide, int64_t *slot_bytes, int64_t *off_bm)
{
	const int64_t vb = wide ? 8 : 4;
	*off_bm = ((int64_t) rpad + Sm) * 64 * vb;
	*slot_bytes = *off_bm + ((int64_t) ((rpad + Sm) / 32 + 2) * 4 + 255) / 256 * 256;      // one bit per label
}

template <bool WIDE, bool LBM, int NW>
static void launch_group_variant(const GroupArgs &d, int blocks, size_t lds_bytes, hipStream_t stream)
{
	static bool configured = false;
	if (LBM && !configured) {
		HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&schur_group_kernel<WIDE, LBM, NW>),
		                              hipFuncAttributeMaxDynamicSharedMemorySize, GR_LBM_MAX_BYTES));
		configured = true;
	}
	hipLaunchKernelGGL((schur_group_kernel<WIDE, LBM, NW>), dim3(blocks), dim3(64 * NW), LBM ? lds_bytes : 0, stream, d);
}

// name of the variant launch_schur_group picks, as rocprofv3 prints it (the bench line quotes it)
void schur_group_variant_name(int r, bool wide, int waves, char *out, size_t cap)
{
	const size_t piv_bytes = ((size_t) r / 32 + 1) * 4;
	const bool lbm = piv_bytes <= (size_t) GR_LBM_MAX_BYTES;
	snprintf(out, cap, "schur_group_kernel<%s,%s,%d>", wide ? "true" : "false", lbm ? "true" : "false",
	         waves >= 4 ? 4 : waves >= 2 ? 2 : 1);
}

// waves: 1, 2 or 4 waves per row group (more when groups are few: the chain of a group is then the run time)
void launch_schur_group(const SchurArgs &a, unsigned char *scratch, int64_t slot_bytes, int64_t off_bm, bool wide,
                        uint32_t *dense_out, int64_t ldS, int blocks, hipStream_t stream, int watch, float min_eff,
                        long long min_w, int waves)
{
	GroupArgs d;
	d.a = a;
	d.watch = watch;
	d.min_eff = min_eff;
	d.min_w = (unsigned long long) min_w;
	d.scratch = scratch;
	d.slot_bytes = slot_bytes;
	d.off_bm = off_bm;
	d.dense_out = dense_out;
	d.ldS = ldS;
	// pending bitmap of the pivotal labels in LDS when it fits; the bits of the non-pivotal labels too when both do
	const size_t piv_bytes = ((size_t) a.r / 32 + 1) * 4;
	const size_t all_bytes = (((size_t) a.r + (size_t) a.Sm) / 32 + 2) * 4;
	const bool lbm = piv_bytes <= (size_t) GR_LBM_MAX_BYTES;
	d.touched_lds = (lbm && all_bytes <= (size_t) GR_LBM_MAX_BYTES && sh::env_get("SPASM_HIP_GROUP_TOUCHED_HBM") == nullptr) ? 1 : 0;     // (the knob: tests)
	const size_t bm_bytes = d.touched_lds ? all_bytes : piv_bytes;
	const int nwv = waves >= 4 ? 4 : waves >= 2 ? 2 : 1;
#define SPASM_LAUNCH_GROUP(WIDE_, LBM_, LDS_)                                                    \
	do {                                                                                        \
		if (nwv == 4)                                                                           \
			launch_group_variant<WIDE_, LBM_, 4>(d, blocks, LDS_, stream);                      \
		else if (nwv == 2)                                                                      \
			launch_group_variant<WIDE_, LBM_, 2>(d, blocks, LDS_, stream);                      \
		else                                                                                    \
			launch_group_variant<WIDE_, LBM_, 1>(d, blocks, LDS_, stream);                      \
	} while (0)
	if (lbm && wide)
		SPASM_LAUNCH_GROUP(true, true, bm_bytes);
	else if (lbm)
		SPASM_LAUNCH_GROUP(false, true, bm_bytes);
	else if (wide)
		SPASM_LAUNCH_GROUP(true, false, 0);
	else
		SPASM_LAUNCH_GROUP(false, false, 0);
#undef SPASM_LAUNCH_GROUP
	HIP_CHECK(hipGetLastError());
}

}  // namespace sh
