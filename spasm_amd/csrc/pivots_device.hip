// The greedy cycle-free pivot search (spasm_pivots.c:147-305, step 3 of host_pivots.cpp) on the device.
//
// The search is a breadth-first walk per candidate row through "the other columns of the pivot row of column c"; the host
// version spends 11-21 ns per visited pivot row and thread, 16 threads at most on these boxes (cgroup quota), and it is
// 65-80 % of an echelonization of the stand-in matrices.  Here one WAVEFRONT searches one row, 64 frontier columns per
// step, with the set of reached columns as one bit per column in LDS (m / 8 bytes: 47 KB for the 376,320 columns of
// ch8-8.b5, three searches per CU) and hundreds of searches in flight.  The transactions are those of the threaded host
// search (acyclic_greedy_threads): a wave explores against the pivots it can see, replays the journal of pivots committed
// meanwhile on its own marks, and commits -- but not under a lock, nor by a compare-and-swap on the length of the journal
// (the first version: with 2,048 searches in flight every commit sent hundreds of waiting waves back to replay one entry
// and fail their swap again, 720 attempts per pivot on mk13.b5, 20 us per commit).  A wave that has a pivot draws a
// TICKET (one atomic add, never refused), writes its proposal there, and looks at the tickets drawn between its last replay
// and its own: if one of them -- accepted, or still undecided -- falls on a column this search has marked, it withdraws
// (ABORTED), replays and goes on; otherwise its pivot is ACCEPTED.  Serialized by ticket number this is the sequential
// algorithm: an accepted ticket has seen every accepted ticket before it, either in a replay or as one that does not touch
// it.  Nobody waits for a chain: a ticket is decided a few loads after it was drawn, whatever the tickets before it do.
// A replay only moves past DECIDED tickets (it polls the few that are not: they are a few loads from their decision), so the
// record of an accepted pivot is complete before any search relies on it; a search that starts takes a lower bound of the
// decided prefix (PsCtrl::prefix, raised by whoever replays) as the point its first replay starts from.
// (Tried in round 4: one BACKWARD level -- the pivot rows that hold a candidate are known from the columns of A, so a candidate
//  can count as reached as soon as one of their pivot columns is MARKED, a level of the walk earlier.  Four searches in five
//  end without a pivot and make 70 % of the visits (mk15.b4: 4.3e9 of 5.9e9, 1,900 per row), so a level looked like a factor
//  of the branching.  It is not: the visits did not move (5.75e9 against 5.46e9; 1.05e10 against 1.29e10 on 19-entry random
//  rows) and the table look-ups cost 5-40 % -- the pivot graph of these matrices is thousands of levels deep and a walk is
//  long and thin, so a level is a few dozen visits, not three quarters of them.)
// (Also tried in round 4, HBM variant: the reached columns of a search in an LDS hash set -- 4,096 slots, open addressing --
//  until they outgrow half of it, which four searches in five never do, so that their probes never leave the CU.  Same time
//  (mk15.b4 0.50 s against 0.45; mk14.b4 with the bits forced to HBM 0.135 against 0.081 with one bit per column in LDS):
//  the HBM variant is not slow because of where the marks are.  A step is a chain of round trips -- 6.8 us with eight
//  searches per CU, 4 us with four -- and more searches in flight buy 15 %.)
// As with threads, the set of pivots depends on timing; it is always cycle-free -- and the host checks that the order it
// derives from the result is triangular before anything is built on it (host_pivots.cpp, Search::triangular).
//
// What waves hand to each other inside the launch -- the pivot records, the journal, the counter -- is written and read
// with agent-scope atomic stores and loads (write-through `sc1` stores, L1-bypassing `sc1` loads); a committer drains its
// record stores (s_waitcnt vmcnt(0)) before it stores the journal entry that announces them.  Everything else a wave
// touches in global memory is either read-only in the launch (A, the rows that had a pivot before) or its own (its FIFO;
// its marks when the matrix is too wide for the LDS -- set by atomics and read by L1-bypassing loads, since an atomic
// leaves the L1 as it was).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "device_types.h"

namespace sh {
namespace {

typedef int64_t i64;
typedef unsigned long long u64;

// record of a pivotal column, 16 bytes = ONE load per visited pivot row.  Bits [0, 3): 0 = not pivotal, 1..ENTS = that many
// other columns of its row follow, BITS bits each; 7 = the row is too long for that (or has nothing else): bits [3, 64) are
// where its entries start in A and the second word is their number (until round 4: the index of the row, and the walk paid
// a round trip for Ap[row] before it could ask for the entries -- one of three per step on a matrix of long rows).  Two formats: six columns of 20 bits (m <= 2^20; with the reached-bits in LDS: m <= 2^19), five of 25.
constexpr int REC_WORDS = 2, REC_LONG = 7;

template <int BITS> __host__ __device__ inline void rec_put(unsigned long long &lo, unsigned long long &hi, int k, unsigned long long v)
{
	const int off = 3 + BITS * k;
	if (off < 64) {
		lo |= v << off;
		if (off + BITS > 64)
			hi |= v >> (64 - off);
	} else {
		hi |= v << (off - 64);
	}
}

template <int BITS> __host__ __device__ inline int rec_get(unsigned long long lo, unsigned long long hi, int k)
{
	const int off = 3 + BITS * k;
	unsigned long long v;
	if (off < 64) {
		v = lo >> off;
		if (off + BITS > 64)
			v |= hi << (64 - off);
	} else {
		v = hi >> (off - 64);
	}
	return (int) (v & ((1ull << BITS) - 1ull));
}

struct PsCtrl {
	int tickets;         // tickets drawn = length of the journal
	int prefix;          // every ticket below is decided (a lower bound, raised by whoever replays)
	int next_row;
	int status;          // 0 ok, 1: a bounded spin gave up (the host search takes over)
	int overflowed;      // rows given up because their FIFO was full
	u64 visits, attempts, steps;
	u64 visits_won, rows_won, rows_lost;             // visits of the searches that ended with a pivot; rows with / without one
	u64 c_search, c_commit, c_total;                 // wave-cycles (s_memtime): in the walk, in replay + ticket, in all
	u64 t_start, t_first_exit, t_last_exit;          // wall_clock64() (100 MHz): first wave in, first wave out of rows, last wave out
	u64 longest_search;                              // ... and the longest time one row took
};

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

int env_int(const char *name, int dflt)
{
	const char *e = sh::env_get(name);
	return (e == nullptr || *e == 0) ? dflt : std::atoi(e);
}

__device__ __forceinline__ int ld_i32(const int *p) { return __hip_atomic_load(p, RLX_AGENT); }
__device__ __forceinline__ u64 ld_u64(const u64 *p) { return __hip_atomic_load(p, RLX_AGENT); }
__device__ __forceinline__ void st_i32(int *p, int v) { __hip_atomic_store(p, v, RLX_AGENT); }
__device__ __forceinline__ void st_u64(u64 *p, u64 v) { __hip_atomic_store(p, v, RLX_AGENT); }
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// one 16-byte L1-bypassing load (global_load_dwordx4 sc1), waited for on the spot
__device__ __forceinline__ void ld_rec(const u64 *p, u64 &lo, u64 &hi)
{
	u32x4 r;
	asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
	lo = (u64) r.x | ((u64) r.y << 32);
	hi = (u64) r.z | ((u64) r.w << 32);
}
// the record of a column and one word of the reached-bits (HBM variant) in flight together, waited for on the spot: a step of
// the walk is a chain of dependent round trips, and "are the candidates still unreached?" used to be one of its own
__device__ __forceinline__ void ld_rec_and_word(const u64 *p, const uint32_t *q, u64 &lo, u64 &hi, uint32_t &word)
{
	u32x4 r;
	uint32_t w;
	asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dword %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r), "=&v"(w) : "v"(p), "v"(q) : "memory");
	lo = (u64) r.x | ((u64) r.y << 32);
	hi = (u64) r.z | ((u64) r.w << 32);
	word = w;
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// records of the pivots the host found (Faugere-Lachartre), one thread per column
template <int REC_ENTS, int REC_BITS>
__global__ __launch_bounds__(256) void pivot_records_kernel(const i64 *Ap, const int *Aj, const int *qinv, int m, u64 *rec)
{
	const int col = blockIdx.x * 256 + threadIdx.x;
	if (col >= m)
		return;
	const int row = qinv[col];
	u64 lo = 0, hi = 0;
	if (row >= 0) {
		const i64 first = Ap[row], last = Ap[row + 1];
		const i64 others = last - first - 1;
		if (others > REC_ENTS || others <= 0) {
			lo = (u64) REC_LONG | ((u64) first << 3);
			hi = (u64) (last - first);
		} else {
			int len = 0;
			for (i64 px = first; px < last; px++) {
				const int j = Aj[px];
				if (j != col && len < REC_ENTS)
					rec_put<REC_BITS>(lo, hi, len++, (u64) j);
			}
			if (len > 0) {
				lo |= (u64) len;
			} else {
				lo = (u64) REC_LONG | ((u64) first << 3);
				hi = (u64) (last - first);
			}
		}
	}
	u64 *R = rec + (size_t) col * REC_WORDS;
	R[0] = lo;
	R[1] = hi;
}

// journal entry of a ticket, ONE 8-byte word: column | state << 32; 0 = not written yet
constexpr u64 PS_PENDING = 1, PS_ACCEPTED = 2, PS_ABORTED = 3;
__device__ __forceinline__ u64 entry(int col, u64 state) { return (u64) (uint32_t) col | (state << 32); }

constexpr int PS_ROWS_PER_GRAB = 8;
constexpr int PS_LIST = 512;
constexpr unsigned PS_SPIN_LIMIT = 1u << 24;
constexpr int PS_RING = 512;          // the last entries of a search's FIFO, mirrored in LDS

// one wavefront per workgroup.  The reached-bit of every column lives in LDS (GB = false: m / 8 bytes, cleared per row) or,
// for matrices too wide for that, in a private stretch of HBM (GB = true: set and read through the L2 -- atomics and
// L1-bypassing loads --, cleared after a row by walking its FIFO, which holds every column that was marked); then the
// candidate columns of the row (one per lane) in LDS.
template <bool GB, int REC_ENTS, int REC_BITS>
__global__ __launch_bounds__(64) void pivot_search_kernel(const i64 *Ap, const int *Aj, const int *pinv, int n, int m, int words, u64 *rec, u64 *jent, int *jrow,
                                                          PsCtrl *ctrl, int *fifo_all, int fifo_cap, int jcap, uint32_t *gbits, int list_cap, i64 annz)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t ps_lds[];
	uint32_t *bits = GB ? gbits + (size_t) blockIdx.x * (size_t) (words + 64) : ps_lds;
	int *cand = reinterpret_cast<int *>(ps_lds + (GB ? 0 : words));
	int *tmp = cand + 64;
	uint32_t *list = reinterpret_cast<uint32_t *>(tmp + 8);          // PS_LIST (lane, entry) pairs of the long rows of a step
	// The walks of these matrices are long and thin (a few dozen columns per step, thousands of levels): the next step's columns
	// are the ones just queued.  The last PS_RING entries of the FIFO are mirrored here, so that a step reads them without
	// waiting for its own stores to reach memory and come back (a round trip per step).
	int *ring = reinterpret_cast<int *>(list + list_cap + 64);          // (behind the 64 spare words)
	const int lane = threadIdx.x;
	const int spare = words + (GB ? 0 : 64 + 8 + list_cap) + lane;          // a word of this lane's own, for atomics that must do nothing
	auto bits_at = [&](int w) -> uint32_t {                       // (GB: what the atomics left in the L2, not what the L1 remembers)
		if constexpr (GB)
			return (uint32_t) ld_i32(reinterpret_cast<const int *>(bits + w));
		else
			return bits[w];
	};
	const u64 below = (1ull << lane) - 1ull;
	int *fifo = fifo_all + (size_t) blockIdx.x * fifo_cap;
	u64 visits = 0, attempts = 0, longest = 0, steps = 0, c_search = 0, c_commit = 0, visits_won = 0, rows_won = 0, rows_lost = 0;
	const u64 c_begin = clock64();
	bool dead = false;          // a bounded spin gave up somewhere: leave
	if (lane == 0)
		atomicMin(&ctrl->t_start, wall_clock64());

	for (;;) {
		int first = 0;
		if (lane == 0)
			first = atomicAdd(&ctrl->next_row, PS_ROWS_PER_GRAB);
		first = __shfl(first, 0);
		if (first >= n || dead)
			break;
		for (int i = first; i < min(n, first + PS_ROWS_PER_GRAB) && !dead; i++) {
			if (pinv[i] >= 0)
				continue;
			if (ld_i32(&ctrl->status) != 0) {
				dead = true;
				break;
			}
			const u64 t_row = wall_clock64();
			const u64 visits_before = visits;
			if constexpr (!GB)
				for (int w = lane * 4; w < words; w += 256)
					*reinterpret_cast<uint4 *>(bits + w) = make_uint4(0, 0, 0, 0);
			// every ticket below `seen` is decided, and the records of the accepted ones are complete
			int seen = ld_i32(&ctrl->prefix);
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			int head = 0, tail = 0, ncand = 0;
			bool overflow = false;
			auto push = [&](bool pred, int j) {
				const u64 mask = __ballot(pred);
				if (mask == 0)
					return;
				const int pos = tail + __popcll(mask & below);
				if (pred && pos < fifo_cap) {
					fifo[pos] = j;
					ring[pos & (PS_RING - 1)] = j;
				}
				tail += __popcll(mask);
				if (tail > fifo_cap) {
					tail = fifo_cap;
					overflow = true;
				}
			};
			auto reach = [&](bool valid, int j) {          // the column becomes reached; queued if it was not
				bool fresh = false;
				if (valid) {
					const uint32_t bit = 1u << (j & 31);
					fresh = (atomicOr(&bits[j >> 5], bit) & bit) == 0;
				}
				push(fresh, j);
			};
			const i64 row_lo = Ap[i], row_hi = Ap[i + 1];
			for (i64 px0 = row_lo; px0 < row_hi; px0 += 64) {
				const bool valid = px0 + lane < row_hi;
				const int j = valid ? Aj[px0 + lane] : 0;
				const int len = valid ? (int) (ld_u64(rec + (size_t) j * REC_WORDS) & 7ull) : 0;
				reach(valid && len != 0, j);
				const bool is_cand = valid && len == 0;
				const u64 mask = __ballot(is_cand);
				const int pos = ncand + __popcll(mask & below);
				if (is_cand && pos < 64)
					cand[pos] = j;
				// a row with more than 64 columns without a pivot: the others are not eligible, but they ARE entries of this row --
				// marked reached, so that a pivot committed on one of them meanwhile is explored like any pivotal entry
				// (through the FIFO like any reached column: it has no record yet, so walking it does nothing, and the FIFO lists every
				// marked column -- which is what the GB variant clears its bits by)
				reach(is_cand && pos >= 64, j);
				ncand = min(64, ncand + __popcll(mask));
			}
			// candidates still unreached (one per lane)
			auto alive = [&]() -> u64 {
				bool a = false;
				if (lane < ncand) {
					const int j = cand[lane];
					a = (bits_at(j >> 5) & (1u << (j & 31))) == 0;
				}
				return __ballot(a);
			};
			bool committed = false;
			for (;;) {
				u64 live = alive();
				const u64 c0 = clock64();
				while (head < tail && live != 0 && !overflow) {
					if (tail + (REC_ENTS + 1) * 64 > fifo_cap) {
						overflow = true;
						break;
					}
					const int cnt = min(64, tail - head);
					int c;
					if (tail - head <= PS_RING) {
						c = (lane < cnt) ? ring[(head + lane) & (PS_RING - 1)] : -1;
					} else {
						drain();                 // (the FIFO entries pushed by the steps before are in memory)
						c = (lane < cnt) ? fifo[head + lane] : -1;
					}
					head += cnt;
					u64 lo = 0, hi = 0;
					if constexpr (GB) {
						// the records of the step and the candidates' reached-bits (as the step before left them) in one round trip
						const int jc = (lane < ncand) ? cand[lane] : 0;
						uint32_t word;
						ld_rec_and_word(rec + (size_t) (c >= 0 ? c : 0) * REC_WORDS, bits + ((lane < ncand) ? (jc >> 5) : spare), lo, hi, word);
						if (c < 0)
							lo = hi = 0;
						live = __ballot(lane < ncand && (word & (1u << (jc & 31))) == 0);
						if (live == 0)
							break;
					} else {
						if (c >= 0)
							ld_rec(rec + (size_t) c * REC_WORDS, lo, hi);
					}
					// (a record of a long row is published second word first -- its length, never 0 -- then, drained, the first word;
					//  should a 16-byte load ever see the new first word beside the old second one, it is read again)
					for (unsigned spins = 0; __ballot((lo & 7ull) == (u64) REC_LONG && hi == 0) != 0; spins++) {
						if ((lo & 7ull) == (u64) REC_LONG && hi == 0)
							ld_rec(rec + (size_t) c * REC_WORDS, lo, hi);
						if (spins > 64) {
							dead = true;
							break;
						}
					}
					if (dead)
						break;
					const int len = (int) (lo & 7ull);
					const i64 long_off = (i64) (lo >> 3);
					const int long_len = (int) min((u64) (1 << 30), hi);
					visits += (u64) __popcll(__ballot(len != 0));
					steps += 1;
					// The step is bound by instruction issue, not by memory (512 searches in flight are as fast as 2,048, cached record
					// loads change nothing): the six columns of the records are marked by six LDS atomics issued back to back -- a lane
					// with no column there ORs nothing into a word of its own --, then queued with one prefix sum.
					int e[REC_ENTS];
					uint32_t bit[REC_ENTS], old[REC_ENTS];
					if constexpr (GB) {
						// (the marks are in HBM and the search is bound by the atomics the memory side takes, 30 G/s; five columns in
						//  six are marked already: look first -- L1-bypassing loads, issued together --, set only what looks clear)
						uint32_t seen_word[REC_ENTS];
#pragma unroll
						for (int t = 0; t < REC_ENTS; t++) {
							e[t] = rec_get<REC_BITS>(lo, hi, t);
							const bool ok = len != REC_LONG && t < len && e[t] < m;          // (e[t] < m: a record caught half written)
							bit[t] = ok ? 1u << (e[t] & 31) : 0u;
							seen_word[t] = ok ? bits_at(e[t] >> 5) : ~0u;
						}
#pragma unroll
						for (int t = 0; t < REC_ENTS; t++) {
							old[t] = ~0u;
							if ((bit[t] & ~seen_word[t]) != 0)
								old[t] = atomicOr(&bits[e[t] >> 5], bit[t]);
						}
					} else {
#pragma unroll
						for (int t = 0; t < REC_ENTS; t++) {
							e[t] = rec_get<REC_BITS>(lo, hi, t);
							const bool ok = len != REC_LONG && t < len && e[t] < m;          // (e[t] < m: a record caught half written)
							bit[t] = ok ? 1u << (e[t] & 31) : 0u;
							old[t] = atomicOr(&bits[ok ? (e[t] >> 5) : spare], bit[t]);
						}
					}
					int base = tail;
#pragma unroll
					for (int t = 0; t < REC_ENTS; t++) {
						const bool fresh = (bit[t] & ~old[t]) != 0;
						const u64 mask = __ballot(fresh);
						if (fresh) {
							const int pos = base + __popcll(mask & below);
							fifo[pos] = e[t];
							ring[pos & (PS_RING - 1)] = e[t];
						}
						base += __popcll(mask);
					}
					tail = base;
					// Pivot rows too long for a record (GL7d19 has ~19 entries per row: every one of them).  One row after the other with
					// the whole wave -- 19 lanes of 64 at work, one round of marking per visited row -- left the search at 1.4 G visits/s
					// on such a matrix; here the entries of all the long rows of the step are listed in LDS as (lane, entry) pairs, up
					// to list_cap at a time (PS_LIST when the matrix has such rows: the list is LDS that matrices of short rows would rather
					// spend on a fourth search per CU), and marked 64 per round whatever row they come from.
					{
						u64 longs = __ballot(len == REC_LONG && long_off + long_len <= annz);          // (beyond A: a record caught half written)
						i64 my_lo = 0;
						int my_len = 0;
						if ((longs >> lane) & 1) {
							my_lo = long_off;
							my_len = long_len;
						}
						while (longs != 0) {
							const bool in = (longs >> lane) & 1;
							int incl = in ? my_len : 0;          // entries of the waiting rows up to and including this lane's
							for (int d = 1; d < 64; d <<= 1) {
								const int v = __shfl_up(incl, d);
								if (lane >= d)
									incl += v;
							}
							const bool fits = in && incl <= list_cap;
							const u64 batch = __ballot(fits);          // (a prefix of the waiting lanes: incl grows with the lane)
							if (batch == 0) {
								// the first waiting row alone is longer than the list: the whole wave walks it
								const int l0 = __builtin_ctzll(longs);
								const i64 lo0 = ((i64) __shfl((int) (my_lo >> 32), l0) << 32) | (uint32_t) __shfl((int) (uint32_t) my_lo, l0);
								const i64 hi0 = lo0 + __shfl(my_len, l0);
								for (i64 px0 = lo0; px0 < hi0; px0 += 64) {
									const bool valid = px0 + lane < hi0;
									reach(valid, valid ? Aj[px0 + lane] : 0);
								}
								longs &= longs - 1;
								continue;
							}
							if (fits)
								for (int t = 0; t < my_len; t++)
									list[incl - my_len + t] = ((uint32_t) lane << 16) | (uint32_t) t;
							const int total = __shfl(incl, 63 - __builtin_clzll(batch));
							// four rounds of 64 entries at a time: their loads in flight together, their marks set back to back, one prefix
							// sum for the queue (as for the columns of the records)
							for (int g0 = 0; g0 < total && !overflow; g0 += 256) {
								if (tail + 256 > fifo_cap) {
									overflow = true;
									break;
								}
								int jj[4];
								uint32_t bt[4], od[4];
#pragma unroll
								for (int u = 0; u < 4; u++) {
									const int g = g0 + 64 * u + lane;
									const bool valid = g < total;
									const uint32_t ent = valid ? list[g] : 0u;
									const int owner = (int) (ent >> 16);
									const i64 lo1 = ((i64) __shfl((int) (my_lo >> 32), owner) << 32) | (uint32_t) __shfl((int) (uint32_t) my_lo, owner);
									jj[u] = valid ? Aj[lo1 + (ent & 0xFFFFu)] : 0;
									bt[u] = valid ? 1u << (jj[u] & 31) : 0u;
								}
								if constexpr (GB) {
									uint32_t sw[4];
#pragma unroll
									for (int u = 0; u < 4; u++)
										sw[u] = (bt[u] != 0) ? bits_at(jj[u] >> 5) : ~0u;
#pragma unroll
									for (int u = 0; u < 4; u++) {
										od[u] = ~0u;
										if ((bt[u] & ~sw[u]) != 0)
											od[u] = atomicOr(&bits[jj[u] >> 5], bt[u]);
									}
								} else {
#pragma unroll
									for (int u = 0; u < 4; u++)
										od[u] = atomicOr(&bits[(bt[u] != 0) ? (jj[u] >> 5) : spare], bt[u]);
								}
								int base2 = tail;
#pragma unroll
								for (int u = 0; u < 4; u++) {
									const bool fresh = (bt[u] & ~od[u]) != 0;
									const u64 mask = __ballot(fresh);
									if (fresh) {
										const int pos = base2 + __popcll(mask & below);
										fifo[pos] = jj[u];
										ring[pos & (PS_RING - 1)] = jj[u];
									}
									base2 += __popcll(mask);
								}
								tail = base2;
							}
							longs &= ~batch;
						}
					}
					if constexpr (!GB)
						live = alive();
				}
				if constexpr (GB)
					if (live != 0)
						live = alive();          // (the marks of the last step)
				const u64 c1 = clock64();
				c_search += c1 - c0;
				struct Tally {
					u64 &acc, from;
					__device__ ~Tally() { acc += clock64() - from; }
				} tally{c_commit, c1};
				if (live == 0 || overflow)
					break;                       // every candidate is reached: no pivot on this row
				// does column j of another search's pivot fall on our marks?  (a reached column, or a candidate of ours)
				auto touches = [&](int j, bool &was_reached) {
					const uint32_t bit = 1u << (j & 31);
					was_reached = (bits_at(j >> 5) & bit) != 0;
					if (was_reached)
						return true;
					for (int k = 0; k < ncand; k++)
						if (cand[k] == j)
							return true;
					return false;
				};
				// replay the tickets decided since the last look on our marks: an accepted pivot that fell on a candidate of ours
				// makes it reached, one on a reached column has a row we must explore; the others cannot be reached from this row
				const int target = min(jcap, ld_i32(&ctrl->tickets));
				const int seen_before = seen;
				bool touched = false;
				while (seen < target && !dead) {
					const int t = seen + lane;
					const bool valid = t < target;
					u64 g = 0;
					unsigned spins = 0;
					for (;;) {
						if (valid && (g >> 32) < PS_ACCEPTED)
							g = ld_u64(jent + t);
						if (__ballot(valid && (g >> 32) < PS_ACCEPTED) == 0)
							break;
						if (++spins > PS_SPIN_LIMIT) {
							dead = true;
							break;
						}
					}
					if (dead)
						break;
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
					const int j = (int) (uint32_t) g;
					bool hit = false;
					if (valid && (g >> 32) == PS_ACCEPTED) {
						bool was_reached;
						hit = touches(j, was_reached);
						if (hit && !was_reached)
							atomicOr(&bits[j >> 5], 1u << (j & 31));
					}
					push(hit, j);
					touched = touched || __ballot(hit) != 0;
					seen = min(target, seen + 64);
				}
				if (dead)
					break;
				if (seen > seen_before && lane == 0)
					atomicMax(&ctrl->prefix, seen);
				if (touched)
					continue;
				// the first candidate of the row that is still unreached goes on a ticket
				const int chosen = cand[__builtin_ctzll(live)];
				attempts += 1;
				int ticket = 0;
				if (lane == 0) {
					ticket = atomicAdd(&ctrl->tickets, 1);
					if (ticket < jcap) {
						st_i32(jrow + ticket, i);
						st_u64(jent + ticket, entry(chosen, PS_PENDING));
					}
				}
				ticket = __shfl(ticket, 0);
				if (ticket >= jcap) {                // (the journal is full: more withdrawals than anyone planned for)
					dead = true;
					break;
				}
				// the tickets drawn since the replay: written a few cycles after they were drawn, decided or not
				bool conflict = false;
				for (int t0 = seen; t0 < ticket && !dead; t0 += 64) {
					const int t = t0 + lane;
					const bool valid = t < ticket;
					u64 g = 0;
					unsigned spins = 0;
					for (;;) {
						if (valid && g == 0)
							g = ld_u64(jent + t);
						if (__ballot(valid && g == 0) == 0)
							break;
						if (++spins > PS_SPIN_LIMIT) {
							dead = true;
							break;
						}
					}
					bool hit = false;
					if (valid && !dead && (g >> 32) != PS_ABORTED) {
						bool was_reached;
						hit = touches((int) (uint32_t) g, was_reached);
					}
					conflict = conflict || __ballot(hit) != 0;
				}
				if (dead)
					break;
				if (conflict) {
					if (lane == 0)
						st_u64(jent + ticket, entry(chosen, PS_ABORTED));
					continue;
				}
				// accepted: its record, then the journal entry that announces it
				{
					const i64 others = row_hi - row_lo - 1;
					u64 *R = rec + (size_t) chosen * REC_WORDS;
					const u64 as_long = (u64) REC_LONG | ((u64) row_lo << 3), all_of_it = (u64) (row_hi - row_lo);
					if (others > REC_ENTS || others <= 0) {
						if (lane == 0) {
							st_u64(R + 1, all_of_it);
							drain();
							st_u64(R, as_long);
						}
					} else {
						const bool valid = row_lo + lane < row_hi;
						const int j = valid ? Aj[row_lo + lane] : 0;
						const bool other = valid && j != chosen;
						const u64 mask = __ballot(other);
						if (other)
							tmp[__popcll(mask & below)] = j;
						const int len = __popcll(mask);
						if (lane == 0) {
							u64 lo = 0, hi = 0;
							for (int t = 0; t < len; t++)
								rec_put<REC_BITS>(lo, hi, t, (u64) tmp[t]);
							st_u64(R + 1, len > 0 ? hi : all_of_it);
							drain();
							st_u64(R, len > 0 ? (lo | (u64) len) : as_long);
						}
					}
					if (lane == 0) {
						drain();
						st_u64(jent + ticket, entry(chosen, PS_ACCEPTED));
					}
				}
				committed = true;
				break;
			}
			if (committed) {
				visits_won += visits - visits_before;
				rows_won += 1;
			} else {
				rows_lost += 1;
			}
			if (overflow && lane == 0)
				atomicAdd(&ctrl->overflowed, 1);
			if constexpr (GB) {
				// the marks of this row go: every marked column is in the FIFO (a FIFO that overflowed lists only some: all words then)
				drain();
				if (overflow) {
					for (int w = lane; w < words; w += 64)
						bits[w] = 0;
				} else {
					for (int t = lane; t < tail; t += 64)
						bits[fifo[t] >> 5] = 0;
				}
				drain();
			}
			longest = max(longest, (u64) (wall_clock64() - t_row));
		}
	}
	if (dead && lane == 0)
		st_i32(&ctrl->status, 1);
	if (lane == 0) {
		const u64 now = wall_clock64();
		atomicMin(&ctrl->t_first_exit, now);
		atomicMax(&ctrl->t_last_exit, now);
		atomicMax(&ctrl->longest_search, longest);
		atomicAdd(&ctrl->visits, visits);
		atomicAdd(&ctrl->visits_won, visits_won);
		atomicAdd(&ctrl->rows_won, rows_won);
		atomicAdd(&ctrl->rows_lost, rows_lost);
		atomicAdd(&ctrl->steps, steps);
		atomicAdd(&ctrl->c_search, c_search);
		atomicAdd(&ctrl->c_commit, c_commit);
		atomicAdd(&ctrl->c_total, (u64) (clock64() - c_begin));
		atomicAdd(&ctrl->attempts, attempts);
	}
}

}  // namespace

// The search on the device.  pinv / qinv: the pivots found so far (row -> column, column -> row, -1 = none), extended in
// place.  Returns the number of new pivots, or -1 when the search does not apply here -- no device, the switch
// SPASM_HIP_PIVOT_SEARCH=host, more than 2^25 columns -- or gave up; the caller then
// runs the host search (which is the same algorithm: this is a matter of speed, the result is a valid set either way).
int device_acyclic_greedy(const struct spasm_csr *A, int *pinv, int *qinv)
{
	const int n = A->n, m = A->m;
	if (const char *e = sh::env_get("SPASM_HIP_PIVOT_SEARCH"))
		if (std::strcmp(e, "host") == 0)
			return -1;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		(void) hipGetLastError();
		return -1;
	}
	const int words = ((m + 31) / 32 + 255) / 256 * 256;          // (cleared 256 words at a time)
	// the reached-bits in LDS when one bit per column fits 64 KB (and a column fits 20 bits), else in HBM
	bool global_bits = (size_t) words * 4 + (64 + 8 + PS_LIST + 64 + PS_RING) * sizeof(int) > 64 * 1024 || m > (1 << 20);          // (room for the list, needed or not)
	if (const char *e = sh::env_get("SPASM_HIP_PIVOT_BITS"))
		global_bits = global_bits || std::strcmp(e, "global") == 0;
	// rows with more than six other entries have no 16-byte record: their entries go through a list in LDS (if there are any)
	int list_cap = 0;
	for (int i = 0; i < n && list_cap == 0; i++)
		if (A->p[i + 1] - A->p[i] > (global_bits ? 6 : 7))          // (the pivot and five or six others: what a record holds)
			list_cap = PS_LIST;
	const size_t lds = (global_bits ? (size_t) 0 : (size_t) words * 4) + (size_t) (64 + 8 + list_cap + 64 + PS_RING) * sizeof(int);
	if (n <= 0 || m <= 0 || m > (1 << 25))
		return -1;
	const double t0 = wtime();
	hipStream_t stream = nullptr;
	DeviceMatrix dA(A, stream);
	int dev = 0, cus = 256;
	HIP_CHECK(hipGetDevice(&dev));
	HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
	const int fifo_cap = m + 4096;
	// searches in flight per CU: 4 with the marks in LDS (the step is bound by instruction issue: more only adds speculation),
	// 8 with the marks in HBM (bound by memory latency: mk14.b5 6.4 s at 2, 4.2 at 4, 3.4 at 8)
	int per_cu = std::max(1, std::min(env_int("SPASM_HIP_PIVOT_WAVES_PER_CU", global_bits ? 8 : 4), (int) ((160 * 1024) / lds)));
	// (a search owns a FIFO of m + 4096 columns, and m / 8 bytes of marks with global_bits: at most 16 GB in all)
	const size_t per_search = (size_t) fifo_cap * sizeof(int) + (global_bits ? ((size_t) words + 64) * sizeof(uint32_t) : 0);
	while (per_cu > 1 && (size_t) cus * per_cu * per_search > ((size_t) 16 << 30))
		per_cu -= 1;
	const int grid = cus * per_cu;
	std::vector<void *> owned;
	auto dal = [&](size_t bytes) {
		void *ptr = big_alloc(bytes);
		owned.push_back(ptr);
		return ptr;
	};
	u64 *rec = (u64 *) dal((size_t) m * REC_WORDS * sizeof(u64));
	int *d_pinv = (int *) dal((size_t) n * sizeof(int));
	int *d_qinv = (int *) dal((size_t) m * sizeof(int));
	const size_t jcap = (size_t) 8 * n + 65536;          // tickets: one per accepted pivot (<= n) and one per withdrawal
	u64 *jent = (u64 *) dal(jcap * sizeof(u64));
	int *jrow = (int *) dal(jcap * sizeof(int));
	PsCtrl *ctrl = (PsCtrl *) dal(sizeof(PsCtrl));
	int *fifo = (int *) dal((size_t) grid * fifo_cap * sizeof(int));
	uint32_t *gbits = nullptr;
	if (global_bits) {
		gbits = (uint32_t *) dal((size_t) grid * ((size_t) words + 64) * sizeof(uint32_t));
		HIP_CHECK(hipMemsetAsync(gbits, 0, (size_t) grid * ((size_t) words + 64) * sizeof(uint32_t), stream));
	}
	const double t_alloc = wtime();
	HIP_CHECK(hipMemcpyAsync(d_pinv, pinv, (size_t) n * sizeof(int), hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(d_qinv, qinv, (size_t) m * sizeof(int), hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemsetAsync(jent, 0, jcap * sizeof(u64), stream));
	{
		PsCtrl init;
		std::memset(&init, 0, sizeof(init));
		init.t_start = init.t_first_exit = ~0ull;
		HIP_CHECK(hipMemcpyAsync(ctrl, &init, sizeof(PsCtrl), hipMemcpyHostToDevice, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
	}
	if (global_bits) {
		hipLaunchKernelGGL((pivot_records_kernel<5, 25>), dim3((m + 255) / 256), dim3(256), 0, stream, dA.p, dA.j, d_qinv, m, rec);
		hipLaunchKernelGGL((pivot_search_kernel<true, 5, 25>), dim3(grid), dim3(64), lds, stream, dA.p, dA.j, d_pinv, n, m, words, rec, jent, jrow, ctrl, fifo,
		                   fifo_cap, (int) jcap, gbits, list_cap, (i64) A->p[n]);
	} else {
		hipLaunchKernelGGL((pivot_records_kernel<6, 20>), dim3((m + 255) / 256), dim3(256), 0, stream, dA.p, dA.j, d_qinv, m, rec);
		hipLaunchKernelGGL((pivot_search_kernel<false, 6, 20>), dim3(grid), dim3(64), lds, stream, dA.p, dA.j, d_pinv, n, m, words, rec, jent, jrow, ctrl, fifo,
		                   fifo_cap, (int) jcap, gbits, list_cap, (i64) A->p[n]);
	}
	HIP_CHECK(hipGetLastError());
	PsCtrl c;
	HIP_CHECK(hipMemcpyAsync(&c, ctrl, sizeof(PsCtrl), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	const double t_run = wtime();
	int found = -1;
	if (c.status == 0) {
		const int tickets = std::min(c.tickets, (int) jcap);
		std::vector<u64> ent((size_t) std::max(tickets, 1));
		std::vector<int> rows((size_t) std::max(tickets, 1));
		if (tickets > 0) {
			HIP_CHECK(hipMemcpy(ent.data(), jent, (size_t) tickets * sizeof(u64), hipMemcpyDeviceToHost));
			HIP_CHECK(hipMemcpy(rows.data(), jrow, (size_t) tickets * sizeof(int), hipMemcpyDeviceToHost));
		}
		found = 0;
		for (int t = 0; t < tickets; t++) {
			const u64 state = ent[t] >> 32;
			const int col = (int) (uint32_t) ent[t];
			if (state == PS_ABORTED)
				continue;
			if (state != PS_ACCEPTED || rows[t] < 0 || rows[t] >= n || col < 0 || col >= m || pinv[rows[t]] != -1 || qinv[col] != -1) {
				// an inconsistent journal (a ticket left undecided, a row or a column taken twice): nothing of this search is
				// kept -- the pivots applied so far are taken back and the host search runs (the caller sees -1)
				std::fprintf(stderr, "[pivots] device search: ticket %d = (row %d, column %d, state %llu) is not a new pivot (a bug: please report); "
				                     "the %d pivots of this search are discarded, searching on the host\n", t, rows[t], col, (unsigned long long) state, found);
				for (int u = 0; u < t; u++) {
					if ((ent[u] >> 32) != PS_ACCEPTED)
						continue;
					const int cu = (int) (uint32_t) ent[u];
					if (rows[u] >= 0 && rows[u] < n && cu >= 0 && cu < m && pinv[rows[u]] == cu && qinv[cu] == rows[u]) {
						pinv[rows[u]] = -1;
						qinv[cu] = -1;
					}
				}
				found = -1;
				break;
			}
			pinv[rows[t]] = col;
			qinv[col] = rows[t];
			found += 1;
		}
		if (found >= 0 && sh::env_get("SPASM_HIP_PIVOT_STATS"))
			logmsg("[pivots] device: %d searches in flight (%d per CU, %zu bytes of LDS each%s), %llu pivot rows visited in %llu steps (%llu of them by the %llu searches that ended with a pivot, the rest by %llu that did not), %d tickets for %d pivots, "
			       "%d rows given up (FIFO full) [%.3fs: %.3f upload of A + allocations, %.3f kernels, %.3f journal; in the search kernel the first wave "
			       "ran out of rows after %.1f ms, the last one left after %.1f ms, the longest search of one row took %.1f ms; "
			       "of the waves' time %.0f %% in the walk, %.0f %% in replays and tickets]\n", grid, per_cu, lds, global_bits ? "; reached-bits in HBM" : "", c.visits, c.steps, c.visits_won, c.rows_won, c.rows_lost,
			       c.tickets, found, c.overflowed, wtime() - t0, t_alloc - t0, t_run - t_alloc, wtime() - t_run, 1e-5 * (double) (c.t_first_exit - c.t_start),
			       1e-5 * (double) (c.t_last_exit - c.t_start), 1e-5 * (double) c.longest_search, 100.0 * (double) c.c_search / (double) std::max<u64>(c.c_total, 1),
			       100.0 * (double) c.c_commit / (double) std::max<u64>(c.c_total, 1));
	} else {
		logmsg("[pivots] device search gave up (a wait ran out): the host search takes over\n");
	}
	for (void *ptr : owned)
		big_free(ptr);
	return found;
}

}  // namespace sh
