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
                                                          PsCtrl *ctrl, int *fifo_all, int fifo_cap, int jcap, uint32_t *gbits, int list_cap, i64 annz, const int *rowlist)
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
		for (int i0 = first; i0 < min(n, first + PS_ROWS_PER_GRAB) && !dead; i0++) {
			// (n counts the entries of the row list when there is one: the second pass of the labelled search below)
			const int i = (rowlist != nullptr) ? rowlist[i0] : i0;
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


// ------------------------------------------------------------------------------------------------------------------------------
// Round 5: the search with DEPTH LABELS -- fewer visits, not cheaper ones.
//
// Every column e carries a label D[e] (30 bits) and a state (2 bits: leaf / claimed / pivotal) in ONE 32-bit word, and the
// labels keep, at every instant, the invariant
//     (I)   for every pivotal column c and every other entry e of its pivot row:   D[e] > D[c].
// Labels only ever grow.  (I) makes the pivot graph acyclic by itself -- labels strictly increase along every edge -- and it
// answers most reachability questions of the greedy search (spasm_pivots.c:147-305) without a walk:
//   * column j can only be reached from pivotal column c when D[j] > D[c]: a candidate j of row i whose label does not exceed
//     the label of any pivotal entry of the row is unreachable -- accepted without visiting anything ("free");
//   * otherwise the walk only expands pivot rows whose label is below the largest label of a candidate that is still
//     unreached: everything else cannot lead to one (mk14.b4: 1.04e9 visited pivot rows -> 2.3e7; mk15.b4: 5.9e9 -> 7.3e7 in
//     the sequential simulation of this scheme).
// A pivot (i, j) is ADDED by restoring (I) for its row first: every entry of row i -- and, transitively, whatever hangs below
// it -- is raised above D[j] (the "cascade": a breadth-first list of (column, new label) items, children = parent + 1, built
// without touching anything and then applied from the largest value down, so that a column is never raised before the columns
// its row holds: (I) holds between any two stores); meeting j itself on the way means j IS reachable (a pivot published
// meanwhile, or a pruned walk that was wrong about stale labels): nothing was written, the candidate is dropped.  Then ONE
// compare-and-swap on j's word -- same label, leaf -> claimed -- decides: it fails when anybody raised j or took it meanwhile
// (the row is searched again), and when it succeeds (I) holds for the new row at that instant.  The record of the row is
// written, drained, and the state becomes pivotal.  Nobody relies on anybody's marks: what the walk believes only decides
// which cascades are attempted, never whether the pivot set is cycle-free -- that is (I), kept by single-word atomics.
// A leaf is raised by compare-and-swap on (label, leaf): a raise can never slip past the moment a column becomes pivotal;
// a raiser that finds its leaf pivotal gives up its attempt (what it raised so far is harmless: labels may be too large,
// never too small) and searches the row again.
// Cascades are mostly tiny (half of the accepted pivots need none beyond the leaves of their row) but heavy-tailed (3.5e6
// items in the sequential simulation of mk15.b4): a row whose candidate sits more than `gap_max` above its lowest pivotal
// entry, or whose cascade outgrows `casc_cap` items, is DEFERRED -- a second pass gives the deferred rows (a percent of the
// rows that end with a pivot) to the ticket search above, against the pivots of this pass.
struct PlCtrl {
	int next_row, status, ndeferred, pad;
	u64 visits, visits_won, rows_won, rows_lost, steps;
	u64 free_accepts, walk_accepts, casc_items, casc_steps, casc_wasted;
	u64 deferred_gap, deferred_cap, deferred_retry, restarts, cycles;
	u64 t_start, t_first_exit, t_last_exit, longest_search;
};

constexpr int PL_HASH = 1024;                  // pending values of a cascade (direct-mapped, lossy: a miss costs a duplicate item)
constexpr int PL_SET_LOG = 11, PL_SET = 1 << PL_SET_LOG;          // the set of reached columns of the labelled search (HS): slots ...
constexpr int PL_SET_LIMIT = PL_SET - 7 * 64 - 64;                  // ... and how many columns a row may reach before it is deferred (a step adds up to 6 x 64)
constexpr uint32_t PL_CLAIMED = 1u, PL_PIVOTAL = 3u;
constexpr int PL_EXPANDED = 1 << 30, PL_NOOP = 1 << 29, PL_VALUE = (1 << 29) - 1;
constexpr int PL_MAX_LABEL = (1 << 29) - (1 << 21);          // (room above for the values of a cascade: casc_cap <= 2^20 items)

__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p) { return __hip_atomic_load(p, RLX_AGENT); }
__device__ __forceinline__ int wave_min(int v)
{
	for (int d = 32; d > 0; d >>= 1)
		v = min(v, __shfl_xor(v, d));
	return v;
}
__device__ __forceinline__ int wave_max(int v)
{
	for (int d = 32; d > 0; d >>= 1)
		v = max(v, __shfl_xor(v, d));
	return v;
}

// labels of the pivots the host found: state from qinv, then sweeps of D[e] = max(D[e], D[c] + 1) over the pivot rows until
// nothing moves (the Faugere-Lachartre pivots of these matrices are 5-6 levels deep)
__global__ __launch_bounds__(256) void pivot_labels_init_kernel(const int *qinv, int m, uint32_t *lab)
{
	const int col = blockIdx.x * 256 + threadIdx.x;
	if (col < m)
		lab[col] = (qinv[col] >= 0) ? PL_PIVOTAL : 0u;
}

// after the searches: every column that got a pivot is pivotal in its label word (the ticket search does not touch the words)
__global__ __launch_bounds__(256) void pivot_labels_states_kernel(const int *qinv, int m, uint32_t *lab)
{
	const int col = blockIdx.x * 256 + threadIdx.x;
	if (col < m)
		lab[col] = (lab[col] & ~3u) | ((qinv[col] >= 0) ? PL_PIVOTAL : 0u);
}

__global__ __launch_bounds__(256) void pivot_labels_relax_kernel(const i64 *Ap, const int *Aj, const int *qinv, int m, uint32_t *lab, int *changed)
{
	const int col = blockIdx.x * 256 + threadIdx.x;
	if (col >= m)
		return;
	const int row = qinv[col];
	if (row < 0)
		return;
	const uint32_t need = (ld_u32(lab + col) >> 2) + 1u;
	bool moved = false;
	for (i64 px = Ap[row]; px < Ap[row + 1]; px++) {
		const int e = Aj[px];
		if (e == col)
			continue;
		const uint32_t word = (need << 2) | ((qinv[e] >= 0) ? PL_PIVOTAL : 0u);
		if (ld_u32(lab + e) < word)
			moved = atomicMax(lab + e, word) < word || moved;
	}
	if (moved)
		*changed = 1;
}

// The same fixed point by WORK LISTS (round 5, at the end of the searches): the pivots of the ticket search hang chains of hundreds
// of levels under their rows, and a sweep over all the pivot rows moves such a chain down by ONE level -- 424 sweeps of 600,000 rows,
// 19 ms on mk15.b4, for 445,000 labels raised in all.  One sweep that also LISTS the pivotal columns whose label it raised, then
// rounds that take the listed columns, relax their pivot rows again and list what THAT raised, until a list stays empty: every
// pivot row has then been relaxed with the final label of its column (its last visit came after the last raise of its column,
// which listed it).  q[0 .. cap) the list, qn[0 .. 3) the counters of three consecutive rounds, qn[3] = 1: a list overflowed (the
// caller falls back to the sweeps / the host).
__global__ __launch_bounds__(256) void pivot_labels_relax_list_kernel(const i64 *Ap, const int *Aj, const int *qinv, int m, uint32_t *lab, int *q, int cap, int *qn)
{
	const int col = blockIdx.x * 256 + threadIdx.x;
	if (col >= m)
		return;
	const int row = qinv[col];
	if (row < 0)
		return;
	const uint32_t need = (ld_u32(lab + col) >> 2) + 1u;
	for (i64 px = Ap[row]; px < Ap[row + 1]; px++) {
		const int e = Aj[px];
		if (e == col)
			continue;
		const bool pivotal = qinv[e] >= 0;
		const uint32_t word = (need << 2) | (pivotal ? PL_PIVOTAL : 0u);
		if (ld_u32(lab + e) < word && atomicMax(lab + e, word) < word && pivotal) {
			const int at = atomicAdd(&qn[0], 1);
			if (at < cap)
				q[at] = e;
			else
				qn[3] = 1;
		}
	}
}

// one round of the chase: the pivot rows of the columns listed by the round before (qin[0 .. cnt[k % 3])) are relaxed again, the
// pivotal columns this raises are listed for the next round (qout, cnt[(k + 1) % 3]); the third counter is zeroed for the round
// after.  No host in between: the rounds are launched in batches and the counters looked at after each batch.
__global__ __launch_bounds__(256) void pivot_labels_round_kernel(const i64 *Ap, const int *Aj, const int *qinv, uint32_t *lab, const int *qin, int *qout, int cap, int *cnt,
                                                                 int k)
{
	const int nin = min(cnt[k % 3], cap);
	if (blockIdx.x == 0 && threadIdx.x == 0)
		cnt[(k + 2) % 3] = 0;
	int *nout = cnt + (k + 1) % 3;
	for (int i = blockIdx.x * 256 + threadIdx.x; i < nin; i += gridDim.x * 256) {
		const int col = qin[i];
		const int row = qinv[col];
		const uint32_t need = (ld_u32(lab + col) >> 2) + 1u;
		for (i64 px = Ap[row]; px < Ap[row + 1]; px++) {
			const int e = Aj[px];
			if (e == col)
				continue;
			const bool pivotal = qinv[e] >= 0;
			const uint32_t word = (need << 2) | (pivotal ? PL_PIVOTAL : 0u);
			if (ld_u32(lab + e) < word && atomicMax(lab + e, word) < word && pivotal) {
				const int at = atomicAdd(nout, 1);
				if (at < cap)
					qout[at] = e;
				else
					cnt[3] = 1;
			}
		}
	}
}

template <bool GB, int REC_ENTS, int REC_BITS, bool HS = false>
__global__ __launch_bounds__(64) void pivot_label_search_kernel(const i64 *Ap, const int *Aj, int *pinv, int n, int m, int words, u64 *rec, uint32_t *lab, PlCtrl *ctrl,
                                                                int *fifo_all, int fifo_cap, uint32_t *gbits, int *deferred, i64 annz, int gap_max, int casc_cap,
                                                                const int *rowlist)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t ps_lds[];
	uint32_t *bits = GB ? gbits + (size_t) blockIdx.x * (size_t) (words + 64) : ps_lds;
	int *cand = reinterpret_cast<int *>(ps_lds + (GB ? 0 : words));
	int *clab = cand + 64;           // the labels of the candidates, as read when the row was
	int *tmp = clab + 64;
	int *ring = tmp + 8;             // the last PS_RING entries of the FIFO / of the cascade list (column ...
	int *ringv = ring + PS_RING;     // ... and value)
	int *hk = ringv + PS_RING;       // pending values of the cascade: column, value
	int *hv = hk + PL_HASH;
	// HS: the reached columns of a row as a SET in LDS (open addressing, column + 1, 0 = free) instead of one bit per column in HBM.
	// The walks of this kernel are pruned -- 14 pivot rows visited per row on mk15.b4 -- and with the marks in HBM every step of a
	// walk was four dependent trips to memory (record, word of marks, atomic on it, marks of the candidates) where the LDS takes
	// one; 2,048 searches x 84 KB of marks were also the one large randomly accessed footprint of the kernel.  A row that reaches more
	// than PL_SET_LIMIT columns is deferred like one that outgrows its FIFO (the ticket search keeps its marks in HBM).
	int *hs = hv + PL_HASH;
	static_assert(!HS || GB, "the set replaces the marks in HBM");
	const int lane = threadIdx.x;
	auto bits_at = [&](int w) -> uint32_t {
		if constexpr (GB)
			return ld_u32(bits + w);
		else
			return bits[w];
	};
	const u64 below = (1ull << lane) - 1ull;
	int *fifo = fifo_all + (size_t) blockIdx.x * fifo_cap;
	u64 visits = 0, visits_won = 0, rows_won = 0, rows_lost = 0, steps = 0, longest = 0;
	u64 free_accepts = 0, walk_accepts = 0, casc_items = 0, casc_steps = 0, casc_wasted = 0, deferred_gap = 0, deferred_cap = 0, deferred_retry = 0, restarts = 0, cycles = 0;
	bool dead = false;
	if (lane == 0)
		atomicMin(&ctrl->t_start, wall_clock64());
	if constexpr (!GB)
		for (int w = lane * 4; w < words; w += 256)
			*reinterpret_cast<uint4 *>(bits + w) = make_uint4(0, 0, 0, 0);
	if constexpr (HS)
		for (int w = lane * 4; w < PL_SET; w += 256)
			*reinterpret_cast<int4 *>(hs + w) = make_int4(0, 0, 0, 0);
	// (HS) column j joins the set: true when it was not there.  The lanes of a call may hold the same column: one of them wins.
	auto set_insert = [&](bool valid, int j) -> bool {
		bool fresh = false;
		if (valid) {
			uint32_t slot = ((uint32_t) j * 2654435761u) >> (32 - PL_SET_LOG);
			for (int probe = 0; probe < PL_SET; probe++) {
				const int old = atomicCAS(&hs[slot], 0, j + 1);
				if (old == 0) {
					fresh = true;
					break;
				}
				if (old == j + 1)
					break;
				slot = (slot + 1) & (PL_SET - 1);
			}
		}
		return fresh;
	};
	auto set_has = [&](int j) -> bool {
		uint32_t slot = ((uint32_t) j * 2654435761u) >> (32 - PL_SET_LOG);
		for (int probe = 0; probe < PL_SET; probe++) {
			const int old = hs[slot];
			if (old == 0)
				return false;
			if (old == j + 1)
				return true;
			slot = (slot + 1) & (PL_SET - 1);
		}
		return true;
	};

	for (;;) {
		int first = 0;
		if (lane == 0)
			first = atomicAdd(&ctrl->next_row, PS_ROWS_PER_GRAB);
		first = __shfl(first, 0);
		if (first >= n || dead)
			break;
		for (int i0 = first; i0 < min(n, first + PS_ROWS_PER_GRAB) && !dead; i0++) {
			// (n counts the entries of the row list when there is one: a later pass, on the rows an earlier one deferred)
			const int i = (rowlist != nullptr) ? rowlist[i0] : i0;
			if (pinv[i] >= 0)
				continue;
			if (ld_i32(&ctrl->status) != 0) {
				dead = true;
				break;
			}
			const u64 t_row = wall_clock64();
			const u64 visits_before = visits;
			const i64 row_lo = Ap[i], row_hi = Ap[i + 1];
			int outcome = -1;          // 0: no pivot on this row, 1: pivot, 2: deferred to the next pass
			bool was_free = false;
			for (int attempt = 0; outcome < 0; attempt++) {
				// (the reached-bits are all clear here)
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
					if constexpr (HS) {
						fresh = set_insert(valid, j);
					} else if (valid) {
						const uint32_t bit = 1u << (j & 31);
						fresh = (atomicOr(&bits[j >> 5], bit) & bit) == 0;
					}
					push(fresh, j);
					if constexpr (HS)
						overflow = overflow || tail > PL_SET_LIMIT;
				};
				// the entries of the row: pivotal (or being committed) ones start the walk, the others are candidates
				int hi_lab = 0x7fffffff;
				for (i64 px0 = row_lo; px0 < row_hi; px0 += 64) {
					const bool valid = px0 + lane < row_hi;
					const int j = valid ? Aj[px0 + lane] : 0;
					const uint32_t w = valid ? ld_u32(lab + j) : 0u;
					const bool piv = valid && (w & 3u) != 0;
					if (piv)
						hi_lab = min(hi_lab, (int) (w >> 2));
					reach(piv, j);
					const bool is_cand = valid && !piv;
					const u64 mask = __ballot(is_cand);
					const int pos = ncand + __popcll(mask & below);
					if (is_cand && pos < 64) {
						cand[pos] = j;
						clab[pos] = (int) (w >> 2);
					}
					reach(is_cand && pos >= 64, j);          // (beyond 64 candidates: entries of the row, not eligible)
					ncand = min(64, ncand + __popcll(mask));
				}
				hi_lab = wave_min(hi_lab);
				auto alive = [&]() -> u64 {
					bool a = false;
					if (lane < ncand) {
						const int j = cand[lane];
						if constexpr (HS)
							a = !set_has(j);
						else
							a = (bits_at(j >> 5) & (1u << (j & 31))) == 0;
					}
					return __ballot(a);
				};
				bool again = false;          // this attempt failed on somebody else's progress: search the row again
				while (outcome < 0 && !again && !dead) {
					u64 live = alive();
					if (live == 0) {
						outcome = 0;
						break;
					}
					int mine = ((live >> lane) & 1) ? clab[lane] : 0x7fffffff;
					int lmin = wave_min(mine);
					int lmax = wave_max(((live >> lane) & 1) ? clab[lane] : -1);
					was_free = lmin <= hi_lab;
					// the walk, pruned: a pivot row at or above the largest label of a live candidate cannot lead to one
					while (!was_free && head < tail && live != 0 && !overflow) {
						if (tail + (REC_ENTS + 1) * 64 > (HS ? min(fifo_cap, PL_SET_LIMIT) : fifo_cap)) {
							overflow = true;
							break;
						}
						const int cnt = min(64, tail - head);
						int c;
						if (tail - head <= PS_RING) {
							c = (lane < cnt) ? ring[(head + lane) & (PS_RING - 1)] : -1;
						} else {
							drain();
							c = (lane < cnt) ? fifo[head + lane] : -1;
						}
						head += cnt;
						u64 lo = 0, hi = 0;
						uint32_t w = 0;
						ld_rec_and_word(rec + (size_t) (c >= 0 ? c : 0) * REC_WORDS, lab + (c >= 0 ? c : 0), lo, hi, w);
						bool open = c >= 0 && (w & 3u) == PL_PIVOTAL && (int) (w >> 2) < lmax;
						// (the record is in memory before the state says pivotal; the two loads are not ordered: read it again)
						for (unsigned spins = 0; __ballot(open && ((lo & 7ull) == 0 || ((lo & 7ull) == (u64) REC_LONG && hi == 0))) != 0; spins++) {
							if (open && ((lo & 7ull) == 0 || ((lo & 7ull) == (u64) REC_LONG && hi == 0)))
								ld_rec(rec + (size_t) c * REC_WORDS, lo, hi);
							if (spins > 4096) {
								dead = true;
								break;
							}
						}
						if (dead)
							break;
						const int len = open ? (int) (lo & 7ull) : 0;
						const i64 long_off = (i64) (lo >> 3);
						const int long_len = (int) min((u64) (1 << 30), hi);
						visits += (u64) __popcll(__ballot(len != 0));
						steps += 1;
						int e[REC_ENTS];
						uint32_t bit[REC_ENTS], old[REC_ENTS];
						if constexpr (HS) {
#pragma unroll
							for (int t = 0; t < REC_ENTS; t++) {
								e[t] = rec_get<REC_BITS>(lo, hi, t);
								const bool ok = len != 0 && len != REC_LONG && t < len && e[t] < m;
								bit[t] = set_insert(ok, e[t]) ? 1u : 0u;
								old[t] = 0u;
							}
						} else if constexpr (GB) {
							uint32_t seen_word[REC_ENTS];
#pragma unroll
							for (int t = 0; t < REC_ENTS; t++) {
								e[t] = rec_get<REC_BITS>(lo, hi, t);
								const bool ok = len != 0 && len != REC_LONG && t < len && e[t] < m;
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
								const bool ok = len != 0 && len != REC_LONG && t < len && e[t] < m;
								bit[t] = ok ? 1u << (e[t] & 31) : 0u;
								old[t] = ~0u;
								if (ok)
									old[t] = atomicOr(&bits[e[t] >> 5], bit[t]);
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
						// rows too long for a record: the whole wave walks them, one after the other
						for (u64 longs = __ballot(len == REC_LONG && long_off + long_len <= annz); longs != 0 && !overflow; longs &= longs - 1) {
							const int l0 = __builtin_ctzll(longs);
							const i64 lo0 = ((i64) __shfl((int) (long_off >> 32), l0) << 32) | (uint32_t) __shfl((int) (uint32_t) long_off, l0);
							const i64 hi0 = lo0 + __shfl(long_len, l0);
							for (i64 px0 = lo0; px0 < hi0 && !overflow; px0 += 64) {
								const bool valid = px0 + lane < hi0;
								reach(valid, valid ? Aj[px0 + lane] : 0);
							}
						}
						const u64 now_live = alive();
						if (now_live != live) {
							live = now_live;
							mine = ((live >> lane) & 1) ? clab[lane] : 0x7fffffff;
							lmin = wave_min(mine);
							lmax = wave_max(((live >> lane) & 1) ? clab[lane] : -1);
						}
					}
					if (dead)
						break;
					if (overflow) {
						outcome = 2;
						deferred_cap += 1;
						break;
					}
					if (live == 0) {
						outcome = 0;
						break;
					}
					// the live candidate with the smallest label: the least to raise
					const int chosen_lane = __builtin_ctzll(__ballot(((live >> lane) & 1) && mine == lmin));
					const int chosen = cand[chosen_lane], dj = lmin;
					if ((hi_lab != 0x7fffffff && dj - hi_lab > gap_max) || dj >= PL_MAX_LABEL) {
						outcome = 2;
						deferred_gap += 1;
						break;
					}
					// ---- the cascade: (column, value) items, breadth first; the list is sorted by value (children = parent + 1 and
					//      a step only takes items of one value), kept behind the FIFO of the walk
					for (int t = lane; t < PL_HASH; t += 64)
						hk[t] = -1;
					const int lbase = (tail + 63) & ~63;
					const int lroom = min(casc_cap, (fifo_cap - lbase) / 2 - 64);
					int lhead = 0, ltail = 0;
					bool cyc = false, capped = lroom < 64;
					auto lpush = [&](bool pred, int x, int v) {
						const u64 mask = __ballot(pred);
						if (mask == 0)
							return;
						const int pos = ltail + __popcll(mask & below);
						if (pred && pos < lroom) {
							fifo[lbase + 2 * pos] = x;
							fifo[lbase + 2 * pos + 1] = v;
							ring[pos & (PS_RING - 1)] = x;
							ringv[pos & (PS_RING - 1)] = v;
						}
						ltail += __popcll(mask);
						if (ltail > lroom) {
							ltail = lroom;
							capped = true;
						}
					};
					// does (e, v) still have to be listed?  not when an item of e with at least that value is pending
					auto want = [&](int e, int v) -> bool {
						const int slot = (int) (((uint32_t) e * 2654435761u) >> 22);
						if (hk[slot] == e && hv[slot] >= v)
							return false;
						hk[slot] = e;
						hv[slot] = v;
						return true;
					};
					for (i64 px0 = row_lo; px0 < row_hi && !capped; px0 += 64) {
						const bool valid = px0 + lane < row_hi;
						const int x = valid ? Aj[px0 + lane] : 0;
						const uint32_t w = valid ? ld_u32(lab + x) : ~0u;
						lpush(valid && x != chosen && (int) (w >> 2) <= dj, x, dj + 1);
					}
					while (lhead < ltail && !cyc && !capped && !dead) {
						int cnt = min(64, ltail - lhead);
						int x = -1, v = 0;
						if (ltail - lhead <= PS_RING) {
							if (lane < cnt) {
								x = ring[(lhead + lane) & (PS_RING - 1)];
								v = ringv[(lhead + lane) & (PS_RING - 1)];
							}
						} else {
							drain();
							if (lane < cnt) {
								x = fifo[lbase + 2 * (lhead + lane)];
								v = fifo[lbase + 2 * (lhead + lane) + 1];
							}
						}
						const int v0 = __shfl(v, 0);
						cnt = __popcll(__ballot(lane < cnt && v == v0));          // (a prefix: the list is sorted)
						const bool active = lane < cnt;
						const int my_index = lhead + lane;
						lhead += cnt;
						casc_steps += 1;
						u64 lo = 0, hi = 0;
						uint32_t w = 0;
						ld_rec_and_word(rec + (size_t) (active ? x : 0) * REC_WORDS, lab + (active ? x : 0), lo, hi, w);
						const bool pivotal = active && (w & 3u) == PL_PIVOTAL;
						bool expand = pivotal && (int) (w >> 2) < v0;
						if (pivotal)
							fifo[lbase + 2 * my_index + 1] = v0 | (expand ? PL_EXPANDED : PL_NOOP);
						// (a later item of the same column carries more: that one looks at the row)
						if (expand) {
							const int slot = (int) (((uint32_t) x * 2654435761u) >> 22);
							if (hk[slot] == x && hv[slot] > v0)
								expand = false;
						}
						for (unsigned spins = 0; __ballot(expand && ((lo & 7ull) == 0 || ((lo & 7ull) == (u64) REC_LONG && hi == 0))) != 0; spins++) {
							if (expand && ((lo & 7ull) == 0 || ((lo & 7ull) == (u64) REC_LONG && hi == 0)))
								ld_rec(rec + (size_t) x * REC_WORDS, lo, hi);
							if (spins > 4096) {
								dead = true;
								break;
							}
						}
						if (dead)
							break;
						const int len = expand ? (int) (lo & 7ull) : 0;
						const i64 long_off = (i64) (lo >> 3);
						const int long_len = (int) min((u64) (1 << 30), hi);
						int e[REC_ENTS];
						uint32_t ew[REC_ENTS];
#pragma unroll
						for (int t = 0; t < REC_ENTS; t++) {
							e[t] = rec_get<REC_BITS>(lo, hi, t);
							const bool ok = len != 0 && len != REC_LONG && t < len && e[t] < m;
							ew[t] = ok ? ld_u32(lab + e[t]) : ~0u;
						}
#pragma unroll
						for (int t = 0; t < REC_ENTS; t++) {
							bool need = ew[t] != ~0u && (int) (ew[t] >> 2) <= v0;
							cyc = cyc || __ballot(need && e[t] == chosen) != 0;
							need = need && want(e[t], v0 + 1);
							lpush(need, e[t], v0 + 1);
						}
						for (u64 longs = __ballot(len == REC_LONG && long_off + long_len <= annz); longs != 0 && !capped; longs &= longs - 1) {
							const int l0 = __builtin_ctzll(longs);
							const i64 lo0 = ((i64) __shfl((int) (long_off >> 32), l0) << 32) | (uint32_t) __shfl((int) (uint32_t) long_off, l0);
							const i64 hi0 = lo0 + __shfl(long_len, l0);
							const int self = __shfl(x, l0);
							for (i64 px0 = lo0; px0 < hi0 && !capped; px0 += 64) {
								const bool valid = px0 + lane < hi0;
								const int ee = valid ? Aj[px0 + lane] : 0;
								const uint32_t eww = (valid && ee != self) ? ld_u32(lab + ee) : ~0u;
								bool need = eww != ~0u && (int) (eww >> 2) <= v0;
								cyc = cyc || __ballot(need && ee == chosen) != 0;
								need = need && want(ee, v0 + 1);
								lpush(need, ee, v0 + 1);
							}
						}
					}
					if (dead)
						break;
					casc_items += (u64) ltail;
					if (cyc) {
						// the candidate IS reachable (through a pivot published meanwhile): nothing was written; it counts as reached
						cycles += 1;
						casc_wasted += (u64) ltail;
						reach(lane == 0, chosen);
						continue;
					}
					if (capped) {
						casc_wasted += (u64) ltail;
						outcome = 2;
						deferred_cap += 1;
						break;
					}
					// ---- applied from the largest value down: all items of one value, then -- once they are in memory -- the next
					drain();
					bool conflict = false;
					for (int pos = ltail; pos > 0 && !conflict;) {
						const int lo_idx = max(0, pos - 64);
						const int k = lo_idx + lane;
						const bool valid = k < pos;
						const int x = valid ? fifo[lbase + 2 * k] : 0;
						const int vv = valid ? fifo[lbase + 2 * k + 1] : 0;
						const int vtop = __shfl(vv, pos - 1 - lo_idx) & PL_VALUE;
						const u64 group = __ballot(valid && (vv & PL_VALUE) == vtop);          // (a suffix: the list is sorted)
						const bool mine_now = (group >> lane) & 1;
						bool bad = false;
						if (mine_now && (vv & PL_NOOP) == 0) {
							if (vv & PL_EXPANDED) {
								atomicMax(lab + x, ((uint32_t) vtop << 2) | PL_PIVOTAL);
							} else {
								uint32_t w = ld_u32(lab + x);
								for (unsigned spins = 0;;) {
									const uint32_t st = w & 3u;
									if (st == PL_PIVOTAL) {          // it got a row meanwhile, which must rise first: give up the attempt
										bad = true;
										break;
									}
									if (st == PL_CLAIMED) {
										if (++spins > (1u << 20)) {
											bad = true;
											break;
										}
										w = ld_u32(lab + x);
										continue;
									}
									if ((int) (w >> 2) >= vtop)
										break;
									const uint32_t old = atomicCAS(lab + x, w, (uint32_t) vtop << 2);
									if (old == w)
										break;
									w = old;
								}
							}
						}
						drain();
						conflict = __ballot(bad) != 0;
						pos = lo_idx + __builtin_ctzll(group);
					}
					if (conflict) {
						again = true;
						break;
					}
					// ---- the commit: j's word from (dj, leaf) to (dj, claimed), or not at all
					int ok = 0;
					if (lane == 0)
						ok = atomicCAS(lab + chosen, (uint32_t) dj << 2, ((uint32_t) dj << 2) | PL_CLAIMED) == ((uint32_t) dj << 2);
					ok = __shfl(ok, 0);
					if (!ok) {
						again = true;
						break;
					}
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
								drain();          // (second word first: a reader that sees the new first word sees all of the record)
								st_u64(R, len > 0 ? (lo | (u64) len) : as_long);
							}
						}
						if (lane == 0) {
							drain();
							atomicOr(lab + chosen, 2u);
							st_i32(pinv + i, chosen);
						}
					}
					outcome = 1;
				}
				// the marks of this attempt go: every marked column is in the FIFO
				drain();
				if constexpr (HS) {
					if (tail > 0)
						for (int w = lane * 4; w < PL_SET; w += 256)
							*reinterpret_cast<int4 *>(hs + w) = make_int4(0, 0, 0, 0);
				} else if (overflow || (!GB && tail > words / 2)) {
					if constexpr (GB) {
						for (int w = lane; w < words; w += 64)
							bits[w] = 0;
					} else {
						for (int w = lane * 4; w < words; w += 256)
							*reinterpret_cast<uint4 *>(bits + w) = make_uint4(0, 0, 0, 0);
					}
				} else {
					for (int t = lane; t < tail; t += 64)
						bits[fifo[t] >> 5] = 0;
				}
				drain();
				if (dead)
					break;
				if (again) {
					restarts += 1;
					if (attempt >= 6) {
						outcome = 2;
						deferred_retry += 1;
					}
				}
			}
			if (dead)
				break;
			if (outcome == 1) {
				rows_won += 1;
				visits_won += visits - visits_before;
				if (was_free)
					free_accepts += 1;
				else
					walk_accepts += 1;
			} else if (outcome == 0) {
				rows_lost += 1;
			} else if (lane == 0) {
				deferred[atomicAdd(&ctrl->ndeferred, 1)] = i;
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
		atomicAdd(&ctrl->free_accepts, free_accepts);
		atomicAdd(&ctrl->walk_accepts, walk_accepts);
		atomicAdd(&ctrl->casc_items, casc_items);
		atomicAdd(&ctrl->casc_steps, casc_steps);
		atomicAdd(&ctrl->casc_wasted, casc_wasted);
		atomicAdd(&ctrl->deferred_gap, deferred_gap);
		atomicAdd(&ctrl->deferred_cap, deferred_cap);
		atomicAdd(&ctrl->deferred_retry, deferred_retry);
		atomicAdd(&ctrl->restarts, restarts);
		atomicAdd(&ctrl->cycles, cycles);
	}
}

}  // namespace

// The search on the device.  pinv / qinv: the pivots found so far (row -> column, column -> row, -1 = none), extended in
// place.  Returns the number of new pivots, or -1 when the search does not apply here -- no device, the switch
// SPASM_HIP_PIVOT_SEARCH=host, more than 2^25 columns -- or gave up; the caller then
// runs the host search (which is the same algorithm: this is a matter of speed, the result is a valid set either way).
// col_label (optional): when the labelled search ran, the final depth label of every column -- D[e] > D[c] for every other
// entry e of the pivot row of every pivotal column c, re-established for ALL pivots (the ticket search's too) by sweeps that
// only end when nothing violates it: pivotal rows sorted by the label of their pivot are in triangular order, and the sweeps
// having ended IS the proof that the pivot set is cycle-free (on a cycle the labels would grow for ever).  Left empty when the
// labels are not there (ticket search alone) or the sweeps did not end: the caller then orders and checks on the host.
int device_acyclic_greedy(const struct spasm_csr *A, int *pinv, int *qinv, std::vector<int> *col_label)
{
	if (col_label != nullptr)
		col_label->clear();
	const int n = A->n, m = A->m;
	if (const char *e = sh::env_get("SPASM_HIP_PIVOT_SEARCH"))
		if (std::strcmp(e, "host") == 0)
			return -1;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
		(void) hipGetLastError();
		return -1;
	}
	if (n <= 0 || m <= 0 || m > (1 << 25))
		return -1;
	// the labelled search (round 5) first, the ticket search for the rows it defers; SPASM_HIP_PIVOT_LABELS=0: the ticket search alone
	bool labels = env_int("SPASM_HIP_PIVOT_LABELS", 1) != 0;
	const int gap_max = std::max(0, env_int("SPASM_HIP_PIVOT_GAP", 64));
	const int casc_cap = std::max(64, std::min(1 << 20, env_int("SPASM_HIP_PIVOT_CASCADE", 8192)));
	const int casc_cap_late = std::max(casc_cap, std::min(1 << 20, env_int("SPASM_HIP_PIVOT_CASCADE_LATE", 65536)));
	const int words = ((m + 31) / 32 + 255) / 256 * 256;          // (cleared 256 words at a time)
	const size_t lds_labels_extra = (size_t) (64 + 64 + 8 + 2 * PS_RING + 2 * PL_HASH) * sizeof(int);
	// the reached-bits in LDS when one bit per column fits 64 KB (and a column fits 20 bits), else in HBM
	bool global_bits = (size_t) words * 4 + (64 + 8 + PS_LIST + 64 + PS_RING) * sizeof(int) > 64 * 1024 || m > (1 << 20);          // (room for the list, needed or not)
	if (labels && (size_t) words * 4 + lds_labels_extra > 64 * 1024)
		global_bits = true;
	if (const char *e = sh::env_get("SPASM_HIP_PIVOT_BITS"))
		global_bits = global_bits || std::strcmp(e, "global") == 0;
	// rows with more than six other entries have no 16-byte record: their entries go through a list in LDS (if there are any)
	int list_cap = 0;
	for (int i = 0; i < n && list_cap == 0; i++)
		if (A->p[i + 1] - A->p[i] > (global_bits ? 6 : 7))          // (the pivot and five or six others: what a record holds)
			list_cap = PS_LIST;
	const size_t lds = (global_bits ? (size_t) 0 : (size_t) words * 4) + (size_t) (64 + 8 + list_cap + 64 + PS_RING) * sizeof(int);
	// (round 5, late) with the marks in HBM the labelled search keeps the reached columns of a row as a set in LDS instead
	// (SPASM_HIP_PIVOT_REACHED_SET: 0 never, 1 first pass only -- the second pass then takes what outgrew the set with its marks in HBM
	//  instead of leaving it to the ticket search --, 2 both passes)
	const int hash_set_mode = env_int("SPASM_HIP_PIVOT_REACHED_SET", 2);
	const bool hash_set = labels && global_bits && hash_set_mode != 0;
	const size_t lds_labels = (global_bits ? (size_t) 0 : (size_t) words * 4) + lds_labels_extra + (hash_set ? (size_t) PL_SET * sizeof(int) : 0);
	const double t0 = wtime();
	hipStream_t stream = nullptr;
	DeviceMatrix dA(A, stream);
	int dev = 0, cus = 256;
	HIP_CHECK(hipGetDevice(&dev));
	HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
	// A search of the ticket kernel may reach every column: a FIFO of m + 4096.  The walks of the labelled search are short (a few
	// dozen visits per row, pruned): 32,768 columns, then the cascade list behind them; a row that outgrows that is deferred to
	// the ticket search, which then runs few searches at a time (its rows are a percent of all).  What this saves is device
	// memory that a driver call has to get first: 5.8 GB of FIFOs on mk15.b4 were 0.1-0.3 s of hipMalloc per call.
	const int fifo_cap = m + 4096;
	const int fifo_cap_labels = std::min(m + 4096, env_int("SPASM_HIP_PIVOT_LABEL_FIFO", 32768)) + 2 * casc_cap_late + 256;
	// searches in flight per CU: 4 with the marks in LDS (the step is bound by instruction issue: more only adds speculation),
	// 8 with the marks in HBM (bound by memory latency: mk14.b5 6.4 s at 2, 4.2 at 4, 3.4 at 8)
	const int per_cu_wanted = (global_bits ? 8 : 4);
	int per_cu = std::max(1, std::min(per_cu_wanted, (int) ((160 * 1024) / lds)));
	int per_cu_labels = std::max(1, std::min(per_cu_wanted, (int) ((160 * 1024) / lds_labels)));
	// (a search owns a FIFO of m + 4096 columns, and m / 8 bytes of marks with global_bits: at most 16 GB in all)
	const size_t per_search = (size_t) fifo_cap * sizeof(int) + (global_bits ? ((size_t) words + 64) * sizeof(uint32_t) : 0);
	while (per_cu > 1 && (size_t) cus * per_cu * per_search > ((size_t) 16 << 30))
		per_cu -= 1;
	const size_t per_search_labels = (size_t) fifo_cap_labels * sizeof(int) + (global_bits ? ((size_t) words + 64) * sizeof(uint32_t) : 0);
	while (per_cu_labels > 1 && (size_t) cus * per_cu_labels * per_search_labels > ((size_t) 16 << 30))
		per_cu_labels -= 1;
	// (behind the labelled search the ticket search gets a fraction of the rows: 512 searches at a time are plenty)
	const int grid = labels ? std::min(cus * per_cu, std::max(64, (512))) : cus * per_cu;
	const int grid_labels = cus * per_cu_labels, grid_max = std::max(grid, labels ? grid_labels : 0);
	std::vector<void *> owned;
	auto dal = [&](size_t bytes) {
		void *ptr = big_alloc(bytes);
		owned.push_back(ptr);
		return ptr;
	};
	auto release = [&]() {
		for (void *ptr : owned)
			big_free(ptr);
		owned.clear();
	};
	u64 *rec = (u64 *) dal((size_t) m * REC_WORDS * sizeof(u64));
	int *d_pinv = (int *) dal((size_t) n * sizeof(int));
	int *d_qinv = (int *) dal((size_t) m * sizeof(int));
	int *fifo = labels ? (int *) dal((size_t) grid_labels * fifo_cap_labels * sizeof(int)) : nullptr;          // (the ticket search allocates its own when it runs)
	uint32_t *gbits = nullptr;
	if (global_bits) {
		gbits = (uint32_t *) dal((size_t) grid_max * ((size_t) words + 64) * sizeof(uint32_t));
		HIP_CHECK(hipMemsetAsync(gbits, 0, (size_t) grid_max * ((size_t) words + 64) * sizeof(uint32_t), stream));
	}
	const double t_alloc = wtime();
	sh::h2d(d_pinv, pinv, (size_t) n * sizeof(int), stream);
	sh::h2d(d_qinv, qinv, (size_t) m * sizeof(int), stream);
	if (global_bits)
		hipLaunchKernelGGL((pivot_records_kernel<5, 25>), dim3((m + 255) / 256), dim3(256), 0, stream, dA.p, dA.j, d_qinv, m, rec);
	else
		hipLaunchKernelGGL((pivot_records_kernel<6, 20>), dim3((m + 255) / 256), dim3(256), 0, stream, dA.p, dA.j, d_qinv, m, rec);
	HIP_CHECK(hipGetLastError());
	const bool stats = sh::env_get("SPASM_HIP_PIVOT_STATS") != nullptr;
	int found = 0;

	// ---- first pass: the labelled search
	int *d_deferred = nullptr;
	int ndeferred = 0;
	double t_labels_init = 0.0, t_labels = 0.0;
	std::vector<int> mine;          // the rows that got their pivot in the first pass
	uint32_t *lab = nullptr;        // the label words of the labelled search
	int ticket_pivots = 0;          // pivots the ticket search added behind it
	int *d_changed_final = nullptr;
	if (labels) {
		const double ta = wtime();
		lab = (uint32_t *) dal((size_t) m * sizeof(uint32_t));
		PlCtrl *pctrl = (PlCtrl *) dal(sizeof(PlCtrl) + 64);
		int *d_changed = reinterpret_cast<int *>(reinterpret_cast<char *>(pctrl) + sizeof(PlCtrl));
		d_changed_final = d_changed;
		hipLaunchKernelGGL(pivot_labels_init_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, d_qinv, m, lab);
		// (sweeps until nothing moves, looked at every fourth; the pivots of the Faugere-Lachartre steps are a few levels deep on
		//  boundary matrices -- a matrix on which they are thousands deep goes to the ticket search alone)
		int sweeps = 0;
		for (bool moving = true; moving && labels;) {
			HIP_CHECK(hipMemsetAsync(d_changed, 0, sizeof(int), stream));
			for (int t = 0; t < 4; t++)
				hipLaunchKernelGGL(pivot_labels_relax_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, dA.p, dA.j, d_qinv, m, lab, d_changed);
			sweeps += 4;
			int changed = 0;
			HIP_CHECK(hipMemcpyAsync(&changed, d_changed, sizeof(int), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
			moving = changed != 0;
			if (moving && sweeps >= 8192)
				labels = false;
		}
		t_labels_init = wtime() - ta;
		// Two labelled passes.  The first one defers what would hang long chains under the rows in flight (gap) or costs a long
		// cascade (cap); the second one takes the deferred rows again with the gap rule off and cascades of up to 65,536 items --
		// nine in ten of them have lost their candidate meanwhile and end after a pruned walk of a few dozen visits (mk15.b5:
		// 347,000 rows deferred, 33,000 of them end with a pivot; through the ticket search they cost 120,000 visits apiece:
		// 6-9 s of a 10-12 s call).  What the second pass defers goes to the ticket search.
		const int npasses = std::max(1, std::min(2, env_int("SPASM_HIP_PIVOT_LABEL_PASSES", 2)));
		int nrows_pass = n, late_cap_now = casc_cap_late;
		const int *rowlist_pass = nullptr;
		for (int pass = 0; pass < npasses && labels; pass++) {
			const double tp = wtime();
			const int gap_pass = (pass == 0) ? gap_max : (1 << 30);
			const int cap_pass = (pass == 0) ? casc_cap : late_cap_now;
			int *d_out = (int *) dal((size_t) std::max(nrows_pass, 1) * sizeof(int));
			PlCtrl init;
			std::memset(&init, 0, sizeof(init));
			init.t_start = init.t_first_exit = ~0ull;
			HIP_CHECK(hipMemcpyAsync(pctrl, &init, sizeof(PlCtrl), hipMemcpyHostToDevice, stream));
			const int grid_pass = std::max(1, std::min(grid_labels, (nrows_pass + PS_ROWS_PER_GRAB - 1) / PS_ROWS_PER_GRAB));
			if (hash_set && (pass == 0 || hash_set_mode >= 2))
				hipLaunchKernelGGL((pivot_label_search_kernel<true, 5, 25, true>), dim3(grid_pass), dim3(64), lds_labels, stream, dA.p, dA.j, d_pinv, nrows_pass, m, words, rec, lab, pctrl,
				                   fifo, fifo_cap_labels, gbits, d_out, (i64) A->p[n], gap_pass, cap_pass, rowlist_pass);
			else if (global_bits)
				hipLaunchKernelGGL((pivot_label_search_kernel<true, 5, 25>), dim3(grid_pass), dim3(64), lds_labels, stream, dA.p, dA.j, d_pinv, nrows_pass, m, words, rec, lab, pctrl,
				                   fifo, fifo_cap_labels, gbits, d_out, (i64) A->p[n], gap_pass, cap_pass, rowlist_pass);
			else
				hipLaunchKernelGGL((pivot_label_search_kernel<false, 6, 20>), dim3(grid_pass), dim3(64), lds_labels, stream, dA.p, dA.j, d_pinv, nrows_pass, m, words, rec, lab, pctrl,
				                   fifo, fifo_cap_labels, gbits, d_out, (i64) A->p[n], gap_pass, cap_pass, rowlist_pass);
			HIP_CHECK(hipGetLastError());
			PlCtrl c;
			HIP_CHECK(hipMemcpyAsync(&c, pctrl, sizeof(PlCtrl), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
			t_labels += wtime() - tp;
			if (c.status != 0) {
				logmsg("[pivots] labelled device search gave up (a wait ran out): the host search takes over\n");
				for (int i : mine) {
					qinv[pinv[i]] = -1;
					pinv[i] = -1;
				}
				release();
				return -1;
			}
			counters()[CNT_PIVOT_VISITS] += (long long) c.visits;
			counters()[CNT_PIVOT_VISITS_WON] += (long long) c.visits_won;
			counters()[CNT_PIVOT_CASCADE_ITEMS] += (long long) c.casc_items;
			counters()[CNT_PIVOT_ROWS_WON] += (long long) c.rows_won;
			counters()[CNT_PIVOT_ROWS_LOST] += (long long) c.rows_lost;
			counters()[CNT_PIVOT_FREE_ACCEPTS] += (long long) c.free_accepts;
			if (stats)
				logmsg("[pivots] device, labelled search, pass %d on %d rows: %d searches in flight (%d per CU, %zu bytes of LDS each%s), %llu pivot rows visited in %llu steps (%llu of them on the %llu rows that ended with a pivot: "
				       "%llu accepted on their labels alone, %llu after a walk; %llu rows without one), cascades: %llu items in %llu steps (%llu of them thrown away: %llu candidates found reachable, "
				       "%llu attempts repeated), %d rows deferred (%llu label gap > %d, %llu cascade > %d items, %llu retries) [%.3f s; first wave out of rows after %.1f ms, last one after %.1f ms, longest row %.1f ms]\n",
				       pass + 1, nrows_pass, grid_pass, per_cu_labels, lds_labels, (hash_set && (pass == 0 || hash_set_mode >= 2)) ? "; reached columns as a set in LDS" : global_bits ? "; reached-bits in HBM" : "", c.visits, c.steps, c.visits_won, c.rows_won, c.free_accepts, c.walk_accepts, c.rows_lost,
				       c.casc_items, c.casc_steps, c.casc_wasted, c.cycles, c.restarts, c.ndeferred, c.deferred_gap, gap_pass, c.deferred_cap, cap_pass, c.deferred_retry, wtime() - tp,
				       1e-5 * (double) (c.t_first_exit - c.t_start), 1e-5 * (double) (c.t_last_exit - c.t_start), 1e-5 * (double) c.longest_search);
			ndeferred = c.ndeferred;
			d_deferred = d_out;
			rowlist_pass = d_out;
			nrows_pass = ndeferred;
			// Is a second labelled pass worth it?  Its long cascades are a latency tail (milliseconds per wave), the ticket search
			// costs a closure per row; measured on five matrices (tools/probe_pivot_passes.sh): with MANY rows deferred (mk15.b5:
			// 300,000) it is what makes the call (cascades up to 65,536 items: 8 s of ticket search -> 0.7); with few, cascades up
			// to 16,384 items leave the ticket search a dozen pivots, few enough for the labels to settle afterwards and replace
			// the host's depth-first search (mk15.b4 104 -> 79 ms, mk14.b4 58 -> 46) -- unless the rows were deferred by the gap
			// rule rather than by the cap (mk13.b5 65 : 1, ch8-8.b5 4 : 1): those sit far above their rows, their cascades are the
			// longest, and the ticket search takes them faster (mk13.b5 20 against 27-36 ms, ch8-8.b5 76 against 93-114).
			// (SPASM_HIP_PIVOT_SECOND_PASS_ALWAYS: 1 = the pass runs whatever deferred the rows, 2 = ... and as with many rows)
			const int second_always = env_int("SPASM_HIP_PIVOT_SECOND_PASS_ALWAYS", 0);
			const bool many = second_always >= 2 || ndeferred >= 32768;
			late_cap_now = many ? casc_cap_late : std::min(casc_cap_late, std::max(casc_cap, 16384));
			if (!many && c.deferred_gap > 3 * c.deferred_cap && second_always == 0)
				break;
		}
		if (labels) {
			// its pivots: the rows whose entry of pinv it filled
			std::vector<int> got((size_t) n);
			sh::d2h(got.data(), d_pinv, (size_t) n * sizeof(int), stream);
			bool fine = true;
			for (int i = 0; i < n && fine; i++) {
				if (pinv[i] >= 0 || got[i] < 0)
					continue;
				const int col = got[i];
				if (col >= m || qinv[col] != -1) {
					std::fprintf(stderr, "[pivots] labelled device search: (row %d, column %d) is not a new pivot (a bug: please report); its pivots are discarded, searching on the host\n", i, col);
					fine = false;
					break;
				}
				pinv[i] = col;
				qinv[col] = i;
				mine.push_back(i);
			}
			if (!fine) {
				for (int i : mine) {
					qinv[pinv[i]] = -1;
					pinv[i] = -1;
				}
				release();
				return -1;
			}
			found = (int) mine.size();
			counters()[CNT_PIVOT_DEFERRED] += (long long) ndeferred;
			if (stats)
				logmsg("[pivots] device, labelled search: %d pivots, %d rows left to the ticket search [%.3fs: %.3f upload of A + allocations, %.3f initial labels (%d sweeps), %.3f search]\n", found, ndeferred,
				       wtime() - t0, t_alloc - t0, t_labels_init, sweeps, t_labels);
		}
	}

	// ---- the ticket search: on the rows the first pass deferred, or on all of them
	if (!labels || ndeferred > 0) {
		const double tb = wtime();
		const int nrows = labels ? ndeferred : n;
		const int *rowlist = labels ? d_deferred : nullptr;
		const size_t jcap = (size_t) 8 * nrows + 65536;          // tickets: one per accepted pivot (<= n) and one per withdrawal
		u64 *jent = (u64 *) dal(jcap * sizeof(u64));
		int *jrow = (int *) dal(jcap * sizeof(int));
		PsCtrl *ctrl = (PsCtrl *) dal(sizeof(PsCtrl));
		HIP_CHECK(hipMemsetAsync(jent, 0, jcap * sizeof(u64), stream));
		{
			PsCtrl init;
			std::memset(&init, 0, sizeof(init));
			init.t_start = init.t_first_exit = ~0ull;
			HIP_CHECK(hipMemcpyAsync(ctrl, &init, sizeof(PsCtrl), hipMemcpyHostToDevice, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
		}
		const int grid2 = std::max(1, std::min(grid, (nrows + PS_ROWS_PER_GRAB - 1) / PS_ROWS_PER_GRAB));
		if (fifo == nullptr || (size_t) grid2 * fifo_cap > (size_t) grid_labels * fifo_cap_labels)
			fifo = (int *) dal((size_t) grid2 * fifo_cap * sizeof(int));
		if (global_bits)
			hipLaunchKernelGGL((pivot_search_kernel<true, 5, 25>), dim3(grid2), dim3(64), lds, stream, dA.p, dA.j, d_pinv, nrows, m, words, rec, jent, jrow, ctrl, fifo,
			                   fifo_cap, (int) jcap, gbits, list_cap, (i64) A->p[n], rowlist);
		else
			hipLaunchKernelGGL((pivot_search_kernel<false, 6, 20>), dim3(grid2), dim3(64), lds, stream, dA.p, dA.j, d_pinv, nrows, m, words, rec, jent, jrow, ctrl, fifo,
			                   fifo_cap, (int) jcap, gbits, list_cap, (i64) A->p[n], rowlist);
		HIP_CHECK(hipGetLastError());
		PsCtrl c;
		HIP_CHECK(hipMemcpyAsync(&c, ctrl, sizeof(PsCtrl), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		const double t_run = wtime();
		int found2 = -1;
		if (c.status == 0) {
			const int tickets = std::min(c.tickets, (int) jcap);
			std::vector<u64> ent((size_t) std::max(tickets, 1));
			std::vector<int> rows((size_t) std::max(tickets, 1));
			if (tickets > 0) {
				HIP_CHECK(hipMemcpy(ent.data(), jent, (size_t) tickets * sizeof(u64), hipMemcpyDeviceToHost));
				HIP_CHECK(hipMemcpy(rows.data(), jrow, (size_t) tickets * sizeof(int), hipMemcpyDeviceToHost));
			}
			found2 = 0;
			for (int t = 0; t < tickets; t++) {
				const u64 state = ent[t] >> 32;
				const int col = (int) (uint32_t) ent[t];
				if (state == PS_ABORTED)
					continue;
				if (state != PS_ACCEPTED || rows[t] < 0 || rows[t] >= n || col < 0 || col >= m || pinv[rows[t]] != -1 || qinv[col] != -1) {
					// an inconsistent journal (a ticket left undecided, a row or a column taken twice): nothing of this search is
					// kept -- the pivots applied so far are taken back and the host search runs (the caller sees -1)
					std::fprintf(stderr, "[pivots] device search: ticket %d = (row %d, column %d, state %llu) is not a new pivot (a bug: please report); "
					                     "the %d pivots of this search are discarded, searching on the host\n", t, rows[t], col, (unsigned long long) state, found2);
					for (int u = 0; u < t; u++) {
						if ((ent[u] >> 32) != PS_ACCEPTED)
							continue;
						const int cu = (int) (uint32_t) ent[u];
						if (rows[u] >= 0 && rows[u] < n && cu >= 0 && cu < m && pinv[rows[u]] == cu && qinv[cu] == rows[u]) {
							pinv[rows[u]] = -1;
							qinv[cu] = -1;
						}
					}
					found2 = -1;
					break;
				}
				pinv[rows[t]] = col;
				qinv[col] = rows[t];
				found2 += 1;
			}
			if (found2 >= 0) {
				counters()[CNT_PIVOT_VISITS] += (long long) c.visits;
				counters()[CNT_PIVOT_VISITS_WON] += (long long) c.visits_won;
				counters()[CNT_PIVOT_ROWS_WON] += (long long) c.rows_won;
				counters()[CNT_PIVOT_ROWS_LOST] += (long long) c.rows_lost;
			}
			if (found2 >= 0 && stats)
				logmsg("[pivots] device%s: %d searches in flight (%d per CU, %zu bytes of LDS each%s), %llu pivot rows visited in %llu steps (%llu of them by the %llu searches that ended with a pivot, the rest by %llu that did not), %d tickets for %d pivots, "
				       "%d rows given up (FIFO full) [%.3fs: %.3f kernels, %.3f journal; in the search kernel the first wave "
				       "ran out of rows after %.1f ms, the last one left after %.1f ms, the longest search of one row took %.1f ms; "
				       "of the waves' time %.0f %% in the walk, %.0f %% in replays and tickets]\n", labels ? ", ticket search on the deferred rows" : "", grid2, per_cu, lds, global_bits ? "; reached-bits in HBM" : "", c.visits, c.steps, c.visits_won, c.rows_won, c.rows_lost,
				       c.tickets, found2, c.overflowed, wtime() - tb, t_run - tb, wtime() - t_run, 1e-5 * (double) (c.t_first_exit - c.t_start),
				       1e-5 * (double) (c.t_last_exit - c.t_start), 1e-5 * (double) c.longest_search, 100.0 * (double) c.c_search / (double) std::max<u64>(c.c_total, 1),
				       100.0 * (double) c.c_commit / (double) std::max<u64>(c.c_total, 1));
		} else {
			logmsg("[pivots] device search gave up (a wait ran out): the host search takes over\n");
		}
		if (found2 < 0) {
			// (the pivots of the first pass stand on their own -- they are cycle-free by their labels -- but the caller's contract
			//  is all or nothing: they are taken back too)
			for (int i : mine) {
				qinv[pinv[i]] = -1;
				pinv[i] = -1;
			}
			found = -1;
		} else {
			found += found2;
			ticket_pivots = found2;
		}
	}
	// ---- the labels of the final pivot set (see the head of this function)
	// (only when the ticket search added few pivots: every one of them hangs a long chain of sweeps under its row -- mk13.b5,
	//  450 of them: not settled after 30 ms, which were then lost)
	if (found >= 0 && labels && lab != nullptr && col_label != nullptr && ticket_pivots <= (64) &&
	    env_int("SPASM_HIP_PIVOT_ORDER_BY_LABELS", 1) != 0) {
		const double tl = wtime();
		sh::h2d(d_qinv, qinv, (size_t) m * sizeof(int), stream);
		hipLaunchKernelGGL(pivot_labels_states_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, d_qinv, m, lab);
		int sweeps = 0;
		bool settled = false;
		const double patience = 1e-3 * (double) (30);
		// one sweep that lists what it raised, then the rounds of the chase in batches of 64 launches (see the kernels)
		if (env_int("SPASM_HIP_PIVOT_ORDER_CHASE", 1) != 0) {
			const int qcap = 1 << 22;
			int *d_q0 = (int *) dal((2 * (size_t) qcap + 4) * sizeof(int));
			int *d_q1 = d_q0 + qcap, *d_cnt = d_q1 + qcap;
			HIP_CHECK(hipMemsetAsync(d_cnt, 0, 4 * sizeof(int), stream));
			hipLaunchKernelGGL(pivot_labels_relax_list_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, dA.p, dA.j, d_qinv, m, lab, d_q0, qcap, d_cnt);
			int k = 0, listed = 0;
			const int max_rounds = (1 << 16);
			for (;;) {
				for (int t = 0; t < 64; t++, k++)
					hipLaunchKernelGGL(pivot_labels_round_kernel, dim3(256), dim3(256), 0, stream, dA.p, dA.j, d_qinv, lab, (k & 1) ? d_q1 : d_q0, (k & 1) ? d_q0 : d_q1, qcap, d_cnt, k);
				int cnt[4] = {0, 0, 0, 1};
				HIP_CHECK(hipMemcpyAsync(cnt, d_cnt, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
				HIP_CHECK(hipStreamSynchronize(stream));
				listed += cnt[k % 3];
				if (cnt[3] != 0 || k >= max_rounds || wtime() - tl > patience)
					break;
				if (cnt[k % 3] == 0) {          // (nothing listed for the next round)
					settled = true;
					break;
				}
			}
			sweeps = -k;          // (logged as minus the number of rounds)
			(void) listed;
		}
		// (without the chase: a sweep moves the labels one level down the chains that hang under the pivots of the ticket search: a dozen of them on
		//  mk15.b4 -- 64-330 sweeps, 4-15 ms, against 35 + 8 ms of depth-first search and check on the host --, three thousand
		//  on mk15.b5 -- 6,700 sweeps, 0.87 s: given up after 30 ms, the host then orders as before)
		const int limit = (16384);
		while (!settled && sweeps >= 0 && sweeps < limit && wtime() - tl < patience) {
			HIP_CHECK(hipMemsetAsync(d_changed_final, 0, sizeof(int), stream));
			for (int t = 0; t < 8; t++)
				hipLaunchKernelGGL(pivot_labels_relax_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, dA.p, dA.j, d_qinv, m, lab, d_changed_final);
			sweeps += 8;
			int changed = 1;
			HIP_CHECK(hipMemcpyAsync(&changed, d_changed_final, sizeof(int), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
			settled = changed == 0;
		}
		if (settled) {
			std::vector<uint32_t> words((size_t) m);
			sh::d2h(words.data(), lab, (size_t) m * sizeof(uint32_t), stream);
			col_label->resize((size_t) m);
			for (int j = 0; j < m; j++)
				(*col_label)[(size_t) j] = (int) (words[(size_t) j] >> 2);
		}
		if (stats || verbose() >= 3)
			logmsg("[pivots] labels of the final pivot set: %d sweeps (< 0: rounds of the chase), %s [%.1f ms]\n", sweeps, settled ? "settled" : "NOT settled (the host orders and checks)", 1e3 * (wtime() - tl));
	}
	release();
	return found;
}

}  // namespace sh
