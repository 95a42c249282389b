// Sparse back-substituted factor image: S = A_n - A_p R with R = U_pp^-1 U_pn kept SPARSE.
//
// backsolve.hip stores R dense (r x Sm 16-bit entries): right when the Schur complement is going to be dense anyway
// (mk13.b5: 72 %), wasteful when it stays sparse -- mk14.b4: 273,000 x 42,000 entries = 23 GB built and gathered for a
// Schur complement that is 3.7 % dense, and no image at all beyond 131,072 non-pivotal columns.  The reference never
// forms R: it solves x = a U_pp^-1 row by row (spasm_schur.c:86-171 -> spasm_triangular.c:109-146 -> spasm_reach.c:21-135),
// and what that costs on a GPU is the random read-modify-write traffic of the row-group kernel.
// Here R is formed once per factor, like in backsolve.hip, but as sparse rows:
//
//   * the non-pivotal columns are cut into SEGMENTS of SP_SEG = 4,096 columns;
//   * a row of R is a set of FRAGMENTS, one per segment it has entries in: 4-byte entries (column inside the segment |
//     signed 16-bit value << 16, sorted by column) in a bump-allocated pool of 1-12 chunks; frag[c * nseg + g] = where and
//     how long;
//   * sp_build_kernel<true>: ONE cooperative launch of as many waves as the chip holds (9.6 KB of LDS each: seventeen per CU).
//     A ticket is a ROW of R (round 6): the wave reads the row's lists once -- one 64-byte head for a row of at most seven
//     entries -- and takes it through its segments, R[c][g] = U_n[c][g] - sum_t u_ct R[t][g], publishing every fragment word as
//     it goes.  Rows are handed out from the last to the first, i.e. after the rows they depend on (what spasm_reach's
//     depth-first search orders for one row, the level schedule orders for all of them); a segment whose dependencies are not
//     there yet polls their fragment words, and a row that depends on another walks the same segments one behind it.  While a
//     segment runs, the words of the next one are already asked for and, when they are there, the fragments they name fetched
//     beside the emit.  sp_build_kernel<false> is one (row, segment) task per workgroup and one launch per elimination level
//     (the fall-back when the cooperative launch is refused or the watchdog fires);
//   * the sum of a segment is formed in 8 KB of LDS: 16-bit accumulators, one read-modify-write per entry of a fragment (the
//     entries of a fragment have distinct columns and a wave's LDS accesses are served in order: no atomics).  A lane that finds
//     its accumulator at ZERO lists its column (a ballot gives the new columns their places in the list): no bitmap on the path
//     of a multiply-add.  Emit: the listed columns with a non-zero sum set their bits in a bitmap, ONE packed scan of its words'
//     popcounts gives every column its rank -- sorted output without sorting --, the entries go to the wave's arena of the pool,
//     the fragment word is published.  Work scales with nnz(R), not with r x Sm;
//   * sp_apply_kernel: one wave per reduced row (tickets of four rows, whose metadata is fetched together), through the
//     segments in which the row can hold anything (per row of R a mask of its non-empty segments): the same accumulation over
//     the pivotal entries of the row of A -- the fragments of segment g + 1 in flight while segment g is added up and emitted --,
//     fragments of S into a pool; a scan of the row lengths and sp_gather_kernel (a workgroup takes 256 rows through the
//     segments with the segment's piece of q in LDS) put the rows in their final place (W->d_Sp / d_Sj / d_Sx, columns sorted).
//
// Arithmetic: signed 16-bit representatives with the fp32 reduction of sgn_dev.h (p <= 44,927; 42013 -- the reference's
// default -- qualifies); every multiply-add is reduced at once (|x| <= B, |c v| <= (p/2) B: the sum fits 31 bits).
// Exact mod p, so S is the matrix the other paths compute, bit for bit (tests/test_gpu_sparse_image.py).
#include <algorithm>
#include <cinttypes>
#include <chrono>
#include <thread>
#include <type_traits>
#include <vector>

#include "device_types.h"
#include "field_dev.h"
#include "sgn_dev.h"

namespace sh {

int usable_cpus();          // host_pivots.cpp: the hardware threads, cut down to the CPU quota of the control group

namespace {

constexpr int SEGW = SP_SEG / 2;                       // 32-bit words of a segment's accumulators
constexpr uint64_t LEN_MASK = (1ull << SP_LEN_BITS) - 1;
constexpr uint64_t OFF_MASK = (1ull << SP_OFF_BITS) - 1;
constexpr int SHARD_STRIDE = 16;                        // 64-bit words per shard: one 128-byte line

int env_sp(const char *name, int dflt)
{
	const char *e = sh::env_get(name);
	return (e == nullptr || *e == 0) ? dflt : std::atoi(e);
}

template <typename T> T *dalloc(int64_t count)
{
	return static_cast<T *>(sh::big_alloc((size_t) (count > 0 ? count : 1) * sizeof(T)));
}

template <typename T> void upload(T *dst, const std::vector<T> &src, hipStream_t s)
{
	if (!src.empty())
		sh::h2d(dst, src.data(), src.size() * sizeof(T), s);
}

// A value every lane holds alike, told to the compiler: branches on it are scalar branches.  (Without this the loop
// of the single-launch driver -- whose exits hang on values loaded per lane from uniform addresses -- was compiled into
// exec-masked loops in which lane 0, the lane that publishes, left on its own: tasks were redone for ever and nothing was
// published.  Wave-uniform control flow must be visibly uniform.)
__device__ __forceinline__ uint64_t sp_uniform(uint64_t v)
{
	const uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) v);
	const uint32_t hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (v >> 32));
	return ((uint64_t) hi << 32) | lo;
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src)
{
	const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, src);
	const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), src);
	return ((uint64_t) hi << 32) | lo;
}

// ---- operations of lane 0 that leave no divergent branch behind --------------------------------------
// `if (lane == 0) store(...)` at the end of a loop body is a divergent branch whose join is the loop header; the compiler
// (ROCm 7.2, structurizer) has turned that into a loop nest in which lane 0 leaves first and the other 63 lanes go round
// again -- with readfirstlane / ballots that then see a wave without lane 0: the single-launch build re-ran its task for
// ever and never published (found with the watchdog below: one ticket drawn per wave, every fragment pending).  So lane 0's
// stores and atomics are single instructions under a temporary exec mask: the control flow stays uniform.  All 64 lanes
// are active wherever these are called.
__device__ __forceinline__ void l0_store_u64(uint64_t *p, uint64_t v)
{
	uint64_t saved;
	asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_store_dwordx2 %1, %2, off\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void l0_store_u64_sc1(uint64_t *p, uint64_t v)
{
	uint64_t saved;
	asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_store_dwordx2 %1, %2, off sc1\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void l0_store_i32(int *p, int v)
{
	uint64_t saved;
	asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_store_dword %1, %2, off\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void l0_store_i32_sc1(int *p, int v)
{
	uint64_t saved;
	asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_store_dword %1, %2, off sc1\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void l0_atomic_add_u64(unsigned long long *p, unsigned long long v)
{
	uint64_t saved;
	asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add_x2 %1, %2, off\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ void l0_atomic_smax_i32(int *p, int v)
{
	uint64_t saved;
	asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_smax %1, %2, off\n\ts_mov_b64 exec, %0" : "=&s"(saved) : "v"(p), "v"(v) : "memory");
}

// returning forms: the value lane 0 got, in every lane
__device__ __forceinline__ int l0_atomic_add_i32_ret(int *p, int v)
{
	uint64_t saved;
	int r;
	asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %1, %2, %3, off sc0\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, %0"
	             : "=&s"(saved), "=&v"(r)
	             : "v"(p), "v"(v)
	             : "memory");
	return __builtin_amdgcn_readfirstlane(r);
}

__device__ __forceinline__ unsigned long long l0_atomic_add_u64_ret(unsigned long long *p, unsigned long long v)
{
	uint64_t saved;
	unsigned long long r;
	asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add_x2 %1, %2, %3, off sc0\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, %0"
	             : "=&s"(saved), "=&v"(r)
	             : "v"(p), "v"(v)
	             : "memory");
	return sp_uniform(r);
}

// SPASM_HIP_SPARSE_IMAGE_PROFILE=1: shader-clock cycles per stage, summed over the waves (the kernels take a null pointer otherwise)
struct SpStamp {
	unsigned long long *out;
	unsigned long long last;
	unsigned long long acc[8];
	__device__ __forceinline__ void begin(unsigned long long *p)
	{
		out = p;
		for (int q = 0; q < 8; q++)
			acc[q] = 0;
		last = (p != nullptr) ? __builtin_amdgcn_s_memtime() : 0;
	}
	__device__ __forceinline__ void mark(int slot)
	{
		if (out != nullptr) {
			const unsigned long long t = __builtin_amdgcn_s_memtime();
			acc[slot] += t - last;
			last = t;
		}
	}
	__device__ __forceinline__ void flush(int lane)
	{
		if (out != nullptr && lane == 0)
			for (int q = 0; q < 8; q++)
				atomicAdd(&out[q], acc[q]);
	}
};

// ---- a wave's LDS: the accumulators of one segment, one bit per column that was touched, and the list of those columns ----------
// Round 6.  The kernels of the sparse image are bound by the vector instructions they issue per (row, segment) pair (SQ counters of
// round 5: 262 of them, the vector unit busy 56 % of the time, half of it finding out WHICH columns a segment touched: a scan of
// the bitmap, a loop per lane over the bits of its words -- the fill is clustered, so one lane walked 14 bits on average while
// most lanes had none -- and a second pass over the list).  Now:
//   * the FIRST visitor of a column lists it: the atomic OR that marks the column returns the old word, a ballot of the lanes
//     that set a new bit gives every new column its place in the list (unsorted, each column once, no imbalance);
//   * sorted output without sorting: the rank of column c among the touched columns is prefix[word of c] + popcount(bits of
//     that word below c) -- the prefixes come from ONE packed scan of the bitmap's popcounts and stay in registers (a lane
//     fetches the one it needs with ds_bpermute).  Columns whose sum came back to zero (common: boundary maps) are taken out of
//     the bitmap by a first pass over the list, so the ranks have no holes;
//   * no swizzle of the accumulators (it served the lane-by-lane walk of dense segments, which now goes 64 consecutive
//     columns at a time -- conflict-free as it stands).
// Segments that touch more than LISTCAP columns take that 64-columns-at-a-time walk over the bitmap.
constexpr int BMW = SP_SEG / 32;          // bitmap words
constexpr int WPL = BMW / 64;             // bitmap words per lane (2 or 4)
static_assert(WPL == 2 || WPL == 4, "segments of 4,096 or 8,192 columns");
#ifndef SPASM_SP_LISTCAP
#define SPASM_SP_LISTCAP 464              // 8,192 + 512 + 928 bytes per wave: seventeen waves per CU
#endif
constexpr int LISTCAP = SPASM_SP_LISTCAP;
struct __attribute__((aligned(16))) WaveLds {
	uint32_t acc[SEGW];               // 16-bit accumulators, two per word, column c at halfword c
	uint32_t bm[BMW];
	uint16_t list[LISTCAP];
};
// every odd prime below 2^32 (round 5): 32-bit accumulators, 8-byte fragment entries (column, plain residue in [0, p)),
// coefficients in Montgomery form (c * 2^32 mod p: one montmul per multiply-add).  16 KB + 512 + 928 bytes: nine waves per CU.
struct __attribute__((aligned(16))) WaveLds32 {
	uint32_t acc[SP_SEG];
	uint32_t bm[BMW];
	uint16_t list[LISTCAP];
};

template <typename LDS> __device__ __forceinline__ void sp_lds_init(LDS &L, int lane)
{
	uint4 *a4 = reinterpret_cast<uint4 *>(L.acc);
#pragma unroll
	for (int t = 0; t < (int) (sizeof(L.acc) / 16 / 64); t++)
		a4[t * 64 + lane] = uint4{0u, 0u, 0u, 0u};
#pragma unroll
	for (int t = 0; t < WPL; t++)
		L.bm[64 * t + lane] = 0;
}

__device__ __forceinline__ int sp_acc_get(WaveLds &L, uint32_t c) { return (int) reinterpret_cast<short *>(L.acc)[c]; }
__device__ __forceinline__ void sp_acc_zero(WaveLds &L, uint32_t c) { reinterpret_cast<short *>(L.acc)[c] = 0; }
__device__ __forceinline__ uint32_t sp_acc_get(WaveLds32 &L, uint32_t c) { return L.acc[c]; }
__device__ __forceinline__ void sp_acc_zero(WaveLds32 &L, uint32_t c) { L.acc[c] = 0; }

template <bool SC1> __device__ __forceinline__ uint32_t sp_ld(const uint32_t *p)
{
	if constexpr (SC1)
		return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // (global_load_dword sc1: not from this CU's L1)
	else
		return *p;
}

template <bool SC1> __device__ __forceinline__ void sp_st(uint32_t *p, uint32_t v)
{
	if constexpr (SC1)
		__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);             // (write-through)
	else
		*p = v;
}

template <bool SC1> __device__ __forceinline__ uint64_t sp_ld64(const uint64_t *p)
{
	if constexpr (SC1)
		return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else
		return *p;
}

template <bool SC1> __device__ __forceinline__ void sp_st64(uint64_t *p, uint64_t v)
{
	if constexpr (SC1)
		__hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else
		*p = v;
}

// A lane whose accumulator was ZERO before its update lists its column (first: the lanes that found a zero; nl: the length of the
// list, wave-uniform; beyond LISTCAP it is no longer written and the segment will be walked 64 columns at a time).  An accumulator
// is zero when its column was never touched -- or when its sum has come back to zero for the time being: such a column can be
// listed twice, which the emit pass copes with (both copies find the same sum and the same rank).  No bitmap, hence no atomic,
// on the path of a multiply-add: the clustered fill put ten lanes of a batch on the same bitmap word, one after the other.
template <typename LDS> __device__ __forceinline__ void sp_list_new(LDS &L, uint32_t c, bool first, uint32_t &nl)
{
	const uint64_t firsts = __ballot(first);
	const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t) (firsts >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) firsts, nl));
	if (first && pos < (uint32_t) LISTCAP)
		L.list[pos] = (uint16_t) c;
	nl += (uint32_t) __popcll(firsts);
}

// acc[column] += coef * value for one entry (column | value << 16) per active lane (the columns of the active lanes are distinct;
// coef and value are not zero mod p); the sum is reduced at once.  All 64 lanes come here.
__device__ __forceinline__ void sp_entry(WaveLds &L, uint32_t e, int coef, bool act, const SgnDev &G, uint32_t &nl)
{
	const uint32_t c = e & 0xFFFFu;
	short *a = reinterpret_cast<short *>(L.acc) + c;
	int old = 1;
	if (act)
		old = (int) *a;
	int t = old;
	asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(t) : "v"(e), "v"(coef));
	t = sgn_reduce(t, G);
	if (act)
		*a = (short) t;
	sp_list_new(L, c, old == 0, nl);
}

// two batches of ONE fragment (distinct columns): both reads in flight together
__device__ __forceinline__ void sp_entry2(WaveLds &L, uint32_t e0, uint32_t e1, int coef, bool act0, bool act1, const SgnDev &G, uint32_t &nl)
{
	const uint32_t c0 = e0 & 0xFFFFu, c1 = e1 & 0xFFFFu;
	short *acc = reinterpret_cast<short *>(L.acc);
	int old0 = 1, old1 = 1;
	if (act0)
		old0 = (int) acc[c0];
	if (act1)
		old1 = (int) acc[c1];
	int t0 = old0, t1 = old1;
	asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(t0) : "v"(e0), "v"(coef));
	asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(t1) : "v"(e1), "v"(coef));
	t0 = sgn_reduce(t0, G);
	t1 = sgn_reduce(t1, G);
	if (act0)
		acc[c0] = (short) t0;
	if (act1)
		acc[c1] = (short) t1;
	sp_list_new(L, c0, old0 == 0, nl);
	sp_list_new(L, c1, old1 == 0, nl);
}

// acc[column] += value (an entry of the row itself, not zero); value in [-p/2, p/2]
__device__ __forceinline__ void sp_own_entry(WaveLds &L, uint32_t c, int val, bool act, const SgnDev &G, uint32_t &nl)
{
	int old = 1;
	if (act) {
		short *a = reinterpret_cast<short *>(L.acc) + c;
		old = (int) *a;
		*a = (short) sgn_canonical(old + val, G);
	}
	sp_list_new(L, act ? c : 0u, old == 0, nl);
}

__device__ __forceinline__ uint32_t sp_addmod(uint32_t a, uint32_t b, uint32_t p)          // a, b in [0, p), any p < 2^32
{
	uint32_t s = a + b;
	if (s < a || s >= p)
		s -= p;
	return s;
}

// the 32-bit variant: acc[column] += coef * value for one entry (column | value << 32), coef in Montgomery form
__device__ __forceinline__ void sp_entry(WaveLds32 &L, uint64_t e, uint32_t coef, bool act, const MontDev &F, uint32_t &nl)
{
	const uint32_t c = (uint32_t) e & 0xFFFFu;
	uint32_t old = 1;
	if (act) {
		uint32_t *a = L.acc + c;
		old = *a;
		*a = sp_addmod(old, montmul(coef, (uint32_t) (e >> 32), F), F.p);
	}
	sp_list_new(L, c, old == 0, nl);
}

__device__ __forceinline__ void sp_entry2(WaveLds32 &L, uint64_t e0, uint64_t e1, uint32_t coef, bool act0, bool act1, const MontDev &F, uint32_t &nl)
{
	sp_entry(L, e0, coef, act0, F, nl);
	sp_entry(L, e1, coef, act1, F, nl);
}

__device__ __forceinline__ void sp_own_entry(WaveLds32 &L, uint32_t c, uint32_t val, bool act, const MontDev &F, uint32_t &nl)
{
	uint32_t old = 1;
	if (act) {
		uint32_t *a = L.acc + c;
		old = *a;
		*a = sp_addmod(old, val, F.p);
	}
	sp_list_new(L, act ? c : 0u, old == 0, nl);
}

// where a fragment lies: chunk 0 of the pool (the only one unless a build ran out of room) is a kernel argument in scalar
// registers; the table of the other chunks is only read for fragments that live there
__device__ __forceinline__ const uint32_t *frag_base(const SpPools &P, uint64_t f)
{
	const uint32_t k = (uint32_t) (f >> (SP_LEN_BITS + SP_OFF_BITS)) & 15u;
	const uint32_t *b = P.base[0];
	if (k != 0)
		b = P.base[k];
	return b;
}

__device__ __forceinline__ const uint32_t *frag_ptr(const SpPools &P, uint64_t f) { return frag_base(P, f) + ((f >> SP_LEN_BITS) & OFF_MASK); }

__device__ __forceinline__ const uint64_t *frag_ptr64(const SpPools &P, uint64_t f)
{
	return reinterpret_cast<const uint64_t *>(frag_base(P, f)) + ((f >> SP_LEN_BITS) & OFF_MASK);
}

template <typename ENT, bool SC1> __device__ __forceinline__ ENT sp_ldent(const ENT *p)
{
	if constexpr (sizeof(ENT) == 8)
		return sp_ld64<SC1>(p);
	else
		return sp_ld<SC1>(p);
}

template <typename ENT> __device__ __forceinline__ const ENT *frag_ptr_as(const SpPools &P, uint64_t f)
{
	return reinterpret_cast<const ENT *>(frag_base(P, f)) + ((f >> SP_LEN_BITS) & OFF_MASK);
}

// acc += coef * fragment for the lanes of `live` (their fragment words f are not empty): a wave-uniform loop over those lanes.
// Four fragments at a time: the first 128 entries of all four are in flight together (a reduced row combines 3-5 rows of R per
// segment, each 1-3 batches long: the stage is the latency of these loads).
// ENT: uint32_t (column | signed 16-bit value << 16) with LDS = WaveLds, COEF = int, FLD = SgnDev; uint64_t (column | residue << 32)
// with WaveLds32, uint32_t (Montgomery form), MontDev.
template <bool SC1, typename ENT, typename LDS, typename COEF, typename FLD>
__device__ __forceinline__ void sp_accumulate_live(LDS &L, uint64_t live, uint64_t f, COEF coef, const SpPools &pools, int lane, const FLD &G,
                                                   unsigned long long &ops, uint32_t &nl)
{
	while (live != 0) {
		const ENT *src[4];
		int len[4];
		COEF cf[4];
		ENT h0[4], h1[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			len[u] = 0;
			cf[u] = 0;
			src[u] = reinterpret_cast<const ENT *>(pools.base[0]);
			if (live != 0) {
				const int s = __builtin_ctzll(live);
				live &= live - 1;
				const uint64_t fc = readlane64(f, s);
				cf[u] = (COEF) __builtin_amdgcn_readlane((int) coef, s);
				src[u] = frag_ptr_as<ENT>(pools, fc);
				len[u] = (int) (fc & LEN_MASK);
			}
			// (no branch around the loads: a lane without an entry reads the first word of its fragment, or of the pool)
			h0[u] = sp_ldent<ENT, SC1>(src[u] + (lane < len[u] ? lane : 0));
			h1[u] = sp_ldent<ENT, SC1>(src[u] + (lane + 64 < len[u] ? lane + 64 : 0));
		}
#pragma unroll
		for (int u = 0; u < 4; u++) {
			if (len[u] == 0)
				continue;
			ops += (unsigned long long) len[u];
			if (len[u] > 64)
				sp_entry2(L, h0[u], h1[u], cf[u], lane < len[u], lane + 64 < len[u], G, nl);
			else
				sp_entry(L, h0[u], cf[u], lane < len[u], G, nl);
			for (int i0 = 128; i0 < len[u]; i0 += 128) {
				const bool a0 = i0 + lane < len[u], a1 = i0 + 64 + lane < len[u];
				const ENT e0 = sp_ldent<ENT, SC1>(src[u] + (a0 ? i0 + lane : 0)), e1 = sp_ldent<ENT, SC1>(src[u] + (a1 ? i0 + 64 + lane : 0));
				sp_entry2(L, e0, e1, cf[u], a0, a1, G, nl);
			}
		}
	}
}

template <bool SC1>
__device__ __forceinline__ void sp_accumulate(WaveLds &L, uint64_t f, int coef, bool take, const SpPools &pools, int lane, const SgnDev &G,
                                              unsigned long long &ops, uint32_t &nl)
{
	sp_accumulate_live<SC1, uint32_t>(L, __ballot(take && (f & LEN_MASK) != 0), f, coef, pools, lane, G, ops, nl);
}

template <bool SC1>
__device__ __forceinline__ void sp_accumulate(WaveLds32 &L, uint64_t f, uint32_t coef, bool take, const SpPools &pools, int lane, const MontDev &F,
                                              unsigned long long &ops, uint32_t &nl)
{
	sp_accumulate_live<SC1, uint64_t>(L, __ballot(take && (f & LEN_MASK) != 0), f, coef, pools, lane, F, ops, nl);
}

// The first four fragments of a (row, segment) pair, fetched AHEAD: sp_apply_kernel issues the loads of segment g + 1 before it
// adds up and emits segment g (the fragment words of g + 1 are in registers by then), so that the latency of these loads -- what
// the wave spent most of its "adding" time waiting for -- runs beside the LDS work of the segment before.  Every group issues
// its eight loads whether or not there are fragments for them (an absent one reads the first word of the pool): the wait counts
// are static.  Lanes whose fragments did not fit (a row combining more than four rows of R in one segment) are left in `rest`
// for sp_accumulate_live.
template <typename ENT, typename COEF> struct SpGroup {
	ENT h0[4], h1[4];
	const ENT *src[4];
	int len[4];
	COEF cf[4];
	uint64_t rest;
};

template <bool SC1, typename ENT, typename COEF>
__device__ __forceinline__ void sp_group_issue(SpGroup<ENT, COEF> &Q, uint64_t f, COEF coef, bool take, const SpPools &pools, int lane)
{
	uint64_t live = __ballot(take && (f & LEN_MASK) != 0);
#pragma unroll
	for (int u = 0; u < 4; u++) {
		int len = 0;
		COEF cf = 0;
		const ENT *src = reinterpret_cast<const ENT *>(pools.base[0]);
		if (live != 0) {
			const int s = __builtin_ctzll(live);
			live &= live - 1;
			const uint64_t fc = readlane64(f, s);
			cf = (COEF) __builtin_amdgcn_readlane((int) coef, s);
			src = frag_ptr_as<ENT>(pools, fc);
			len = (int) (fc & LEN_MASK);
		}
		Q.len[u] = len;
		Q.cf[u] = cf;
		Q.src[u] = src;
		Q.h0[u] = sp_ldent<ENT, SC1>(src + (lane < len ? lane : 0));
		Q.h1[u] = sp_ldent<ENT, SC1>(src + (lane + 64 < len ? lane + 64 : 0));
	}
	Q.rest = live;
}

template <bool SC1, typename ENT, typename LDS, typename COEF, typename FLD>
__device__ __forceinline__ void sp_group_consume(LDS &L, const SpGroup<ENT, COEF> &Q, uint64_t f, COEF coef, const SpPools &pools, int lane, const FLD &G,
                                                 unsigned long long &ops, uint32_t &nl)
{
#pragma unroll
	for (int u = 0; u < 4; u++) {
		const int len = Q.len[u];
		if (len == 0)
			continue;
		ops += (unsigned long long) len;
		if (len > 64)
			sp_entry2(L, Q.h0[u], Q.h1[u], Q.cf[u], lane < len, lane + 64 < len, G, nl);
		else
			sp_entry(L, Q.h0[u], Q.cf[u], lane < len, G, nl);
		for (int i0 = 128; i0 < len; i0 += 128) {
			const bool a0 = i0 + lane < len, a1 = i0 + 64 + lane < len;
			const ENT e0 = sp_ldent<ENT, SC1>(Q.src[u] + (a0 ? i0 + lane : 0)), e1 = sp_ldent<ENT, SC1>(Q.src[u] + (a1 ? i0 + 64 + lane : 0));
			sp_entry2(L, e0, e1, Q.cf[u], a0, a1, G, nl);
		}
	}
	if (Q.rest != 0)
		sp_accumulate_live<SC1, ENT>(L, Q.rest, f, coef, pools, lane, G, ops, nl);
}

// inclusive prefix sum over the 64 lanes (DPP: shifts inside the rows of 16 lanes, then the row totals handed on)
__device__ __forceinline__ int wave_incl_scan(int x)
{
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);          // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);          // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);          // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);          // row_shr:8
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);          // row_bcast:15 into rows 1 and 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);          // row_bcast:31 into rows 2 and 3
	return x;
}

// The columns of the segment whose sums are not zero, in column order, through put(position, column, raw sum); the accumulators
// (and the bitmap this pass uses) go back to zero.  nl = length of the list (sp_list_new).  Returns the number of entries.
// All 64 lanes come here; put is called by the lanes that hold an entry (a column listed twice is put twice: same place, same word).
// Sorted output without sorting: the listed columns with a non-zero sum set their bits in the bitmap; the rank of column c is
// prefix[word of c] + popcount(bits of that word below c) -- the prefixes come from ONE packed scan of the words' popcounts and
// stay in registers (a lane fetches the one it needs with ds_bpermute).
template <typename LDS, typename PUT> __device__ __forceinline__ int sp_emit_sorted(LDS &L, uint32_t nl, int lane, PUT put)
{
	if (nl <= (uint32_t) LISTCAP) {
		// lane l holds the words 2l, 2l + 1 (, 128 + 2l, 129 + 2l) of the bitmap -- one 64-bit LDS access --: how many columns stand
		// before its pair, and how many its first word holds (16 bits each in pa; pb: the second pair of a segment of 8,192 columns)
		uint32_t pa = 0, pb = 0;
		int total = 0;
		uint64_t *bm64 = reinterpret_cast<uint64_t *>(L.bm);
		auto prefixes = [&]() {
			const uint64_t wa = bm64[lane], wb = (WPL > 2) ? bm64[64 + lane] : 0ull;
			const int ca = __popcll(wa), cb = __popcll(wb);
			const int is = wave_incl_scan(ca | (cb << 16));          // (a field holds at most 64 * 64)
			const int ts = __builtin_amdgcn_readlane(is, 63);
			const int ta = ts & 0xFFFF;
			total = ta + (ts >> 16);
			pa = (uint32_t) ((is & 0xFFFF) - ca) | ((uint32_t) __popc((uint32_t) wa) << 16);
			pb = (uint32_t) (ta + (is >> 16) - cb) | ((uint32_t) __popc((uint32_t) wb) << 16);
		};
		// the rank of column c among the columns of the bitmap (all 64 lanes call this: ds_bpermute reads the lanes' registers)
		auto rank_of = [&](uint32_t c) -> uint32_t {
			const uint32_t w = c >> 5;
			const uint32_t bits = L.bm[w];
			uint32_t pk = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (((w >> 1) & 63u) << 2), (int) pa);
			if (WPL > 2) {
				const uint32_t pk2 = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (((w >> 1) & 63u) << 2), (int) pb);
				pk = (w & 128u) ? pk2 : pk;
			}
			const uint32_t before = (pk & 0xFFFFu) + ((w & 1u) ? (pk >> 16) : 0u);
			return before + (uint32_t) __popc(bits & ((1u << (c & 31u)) - 1u));
		};
		if (nl <= 64) {
			// the usual case: one batch; the listed columns and their sums stay in registers between the two steps
			const bool in0 = (uint32_t) lane < nl;
			const uint32_t c0 = in0 ? (uint32_t) L.list[lane] : 0u;
			auto v0 = sp_acc_get(L, c0);
			if (!in0)
				v0 = 0;
			if (v0 != 0)
				atomicOr(&L.bm[c0 >> 5], 1u << (c0 & 31u));
			if (in0)
				sp_acc_zero(L, c0);
			prefixes();
			const uint32_t r0 = rank_of(c0);
			if (v0 != 0)
				put(r0, c0, v0);
		} else if (nl <= 128) {
			const bool in0 = true, in1 = (uint32_t) lane + 64 < nl;
			const uint32_t c0 = (uint32_t) L.list[lane], c1 = in1 ? (uint32_t) L.list[lane + 64] : 0u;
			auto v0 = sp_acc_get(L, c0), v1 = sp_acc_get(L, c1);
			if (!in1)
				v1 = 0;
			if (v0 != 0)
				atomicOr(&L.bm[c0 >> 5], 1u << (c0 & 31u));
			if (v1 != 0)
				atomicOr(&L.bm[c1 >> 5], 1u << (c1 & 31u));
			if (in0)
				sp_acc_zero(L, c0);
			if (in1)
				sp_acc_zero(L, c1);
			prefixes();
			const uint32_t r0 = rank_of(c0), r1 = rank_of(c1);
			if (v0 != 0)
				put(r0, c0, v0);
			if (v1 != 0)
				put(r1, c1, v1);
		} else {
			for (uint32_t i0 = 0; i0 < nl; i0 += 64) {
				const uint32_t i = i0 + (uint32_t) lane;
				if (i < nl) {
					const uint32_t c = L.list[i];
					if (sp_acc_get(L, c) != 0)
						atomicOr(&L.bm[c >> 5], 1u << (c & 31u));
				}
			}
			// (LDS operations of a wave are served in order: the reads see the bits)
			prefixes();
			for (uint32_t i0 = 0; i0 < nl; i0 += 64) {
				const uint32_t i = i0 + (uint32_t) lane;
				const bool in = i < nl;
				const uint32_t c = in ? (uint32_t) L.list[i] : 0u;
				auto v = sp_acc_get(L, c);
				if (!in)
					v = 0;
				const uint32_t rank = rank_of(c);
				if (v != 0)
					put(rank, c, v);
			}
			// (a column listed twice must show its sum to both copies: the accumulators go back to zero last)
			for (uint32_t i0 = 0; i0 < nl; i0 += 64) {
				const uint32_t i = i0 + (uint32_t) lane;
				if (i < nl)
					sp_acc_zero(L, (uint32_t) L.list[i]);
			}
		}
		bm64[lane] = 0;
		if (WPL > 2)
			bm64[64 + lane] = 0;
		return total;
	}
	// many columns: 64 consecutive columns at a time, straight from the accumulators
	uint32_t w = 0;
	for (int ch = 0; ch < SP_SEG / 64; ch++) {
		const uint32_t c = (uint32_t) ch * 64u + (uint32_t) lane;
		const auto v = sp_acc_get(L, c);
		const uint64_t nz = __ballot(v != 0);
		if (nz == 0)
			continue;
		const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t) (nz >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) nz, w));
		if (v != 0) {
			sp_acc_zero(L, c);
			put(pos, c, v);
		}
		w += (uint32_t) __popcll(nz);
	}
	return (int) w;
}

// the same walk without output: a segment that cannot be written (no room) still leaves its accumulators at zero
template <typename LDS> __device__ __forceinline__ void sp_discard(LDS &L, uint32_t nl, int lane)
{
	(void) sp_emit_sorted(L, nl, lane, [](uint32_t, uint32_t, auto) {});
}

__device__ __forceinline__ uint32_t sp_hash(uint32_t c, uint32_t g)
{
	uint32_t h = c * 0x9E3779B1u + g * 0x85EBCA77u;
	h ^= h >> 15;
	h *= 0x2C1B3C6Du;
	return h >> 24;          // SP_SHARDS = 256
}

// ---------------------------------------------------------------------------------------------------
// build of R
// ---------------------------------------------------------------------------------------------------
// Two drivers for the same task -- (row c, segment g): R[c][g] = U_n[c][g] - sum_t u_ct R[t][g]:
//   * sp_build_kernel<false>: one launch per elimination level, from the last level to the first (a launch boundary is what
//     makes the fragments of the later levels visible);
//   * sp_build_kernel<true> (default): ONE launch of as many waves as the chip holds at once.  Tasks are handed out in
//     order -- rows from the last to the first, i.e. every row after the rows it depends on -- by ticket counters; a task
//     whose dependencies are not there yet polls their fragment words.  Whoever waits, waits for a task with a smaller
//     number, which a running wave holds or has finished: no deadlock as long as all waves of the grid are resident.  The
//     chain of levels then costs a hand-over between two CUs per level (a few us) instead of a launch and the slowest task
//     of the level (17-18 us measured), and rows of different levels overlap wherever the dependencies allow.
//     Visibility: the entries of a fragment are written through (sc1), the wave drains its stores, then publishes the
//     fragment word (sc1); readers poll the word (sc1) and read the entries past their L1 (sc1).
constexpr uint64_t FRAG_PENDING = ~0ull;                 // not computed yet (length bits all ones: never a valid word)
constexpr uint64_t FRAG_FAILED = ~0ull - 1;              // could not be computed in this launch (no room, or a dependency failed)
constexpr int SP_TICKETS = 16, SP_TICKET_STRIDE = 32;    // ticket counters, one 128-byte line each

struct SpBuildArgs {
	const uint64_t *segmask;      // persistent: per row the segments whose words are pending (nullptr: all of them)
	const uint2 *head;            // 8 words per row: the lists of a row of at most seven entries (word 0: counts; 0xFFFFFFFF: see the lists below)
	const uint64_t *dep_rp;
	const uint2 *dep;
	const uint64_t *np_rp;
	const uint2 *np;
	uint64_t *frag;
	int nseg;
	int row_lo, row_hi;           // level-by-level: compact rows of this level; persistent: all rows (0, r)
	int level, chunk;
	SpPools pools;
	uint32_t *chunk_base;         // the chunk fragments are written to
	unsigned long long *shard;
	int *ovf_level;               // level by level: largest level in which a reservation failed (-1: none); persistent: 1 when any task failed
	unsigned long long shard_sub; // entries of a shard of the current chunk (shard s: [s, s + 1) * shard_sub)
	int arena;                    // persistent: entries a wave reserves at a time from the cursor of shard 0, which then serves the whole
	                              // chunk (0: every fragment reserves its own room from the shard its hash picks)
	int *ticket;                  // persistent: SP_TICKETS counters
	int retry;                    // persistent: not the first launch of this build (some words are done already)
	int *abort_flag;              // persistent: a wave waited too long (the grid is not resident?): everybody gives up
	long long poll_limit;         // ... polls of one batch of dependencies before that happens
	unsigned long long *prof;     // SPASM_HIP_SPARSE_IMAGE_PROFILE=1: 8 cycle counters (ticket, metadata, polling, adding, reservation, emit, publication)
	int *dbg;                     // SPASM_HIP_SPARSE_IMAGE_DEBUG=1: 4 ints per workgroup (stage, task, detail, polls), read by the host's watchdog
	SgnDev G;
	MontDev M;                    // the 32-bit variant: coefficients in Montgomery form, values plain residues
};

__global__ __launch_bounds__(64) void sp_reset_shards_kernel(unsigned long long *shard, unsigned long long sub, int clear_stats)
{
	const int s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= SP_SHARDS)
		return;
	shard[s * SHARD_STRIDE + 0] = (unsigned long long) s * sub;
	shard[s * SHARD_STRIDE + 1] = (unsigned long long) (s + 1) * sub;
	if (clear_stats) {
		shard[s * SHARD_STRIDE + 2] = 0;          // entries of fragments added up (the work of the build)
		shard[s * SHARD_STRIDE + 3] = 0;          // entries written
		shard[s * SHARD_STRIDE + 4] = 0;          // non-empty (row, segment) pairs
		shard[s * SHARD_STRIDE + 5] = 0;          // entries reserved (touched columns: the fill plus what cancelled)
	}
}

// before a build: the word of every (row, segment) pair that can hold something is PENDING, the others are published as empty
__global__ __launch_bounds__(256) void sp_init_frag_kernel(uint64_t *frag, const uint64_t *segmask, int nseg, int64_t n)
{
	for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t) gridDim.x * blockDim.x) {
		const int64_t c = t / nseg;
		const int g = (int) (t - c * nseg);
		frag[t] = ((segmask[c] >> g) & 1ull) ? FRAG_PENDING : 0ull;
	}
}

// failed words back to pending (before the launch that retries them with more room)
__global__ __launch_bounds__(256) void sp_reset_failed_kernel(uint64_t *frag, int64_t n)
{
	for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t) gridDim.x * blockDim.x)
		if (frag[t] == FRAG_FAILED)
			frag[t] = FRAG_PENDING;
}

__device__ __forceinline__ void sp_dbg(const SpBuildArgs &b, int lane, int stage, long long task, int detail)
{
#ifndef SP_NO_DBG
	if (b.dbg != nullptr) {
		int *d = b.dbg + 4 * (size_t) blockIdx.x;
		l0_store_i32_sc1(d + 0, stage);
		l0_store_i32_sc1(d + 1, (int) task);
		l0_store_i32_sc1(d + 2, detail);
	}
#endif
}

// what a wave of the single-launch build carries from task to task: its arena in the pool, its share of the statistics
struct SpWaveState {
	long long ar_cur = 0, ar_end = 0;
	unsigned long long ops = 0, nnz = 0, frags = 0, reserved = 0;
};

template <bool PERSISTENT> __device__ __forceinline__ void sp_publish(uint64_t *fout, uint64_t word, int lane)
{
	if (PERSISTENT) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the entries of the fragment have left
		l0_store_u64_sc1(fout, word);
	} else {
		l0_store_u64(fout, word);
	}
}

// one task: the fragment of (row c, segment g).  One exit, one publication.
// the 64-byte head of row c: lanes 0..7 (see sp_build_task)
__device__ __forceinline__ uint2 sp_row_head(const SpBuildArgs &b, int c, int lane)
{
	return (lane < 8 && b.head != nullptr) ? b.head[(uint64_t) c * 8 + lane] : uint2{0xFFFFFFFFu, 0u};
}

// what a (row, segment) pair of the build leaves behind once its sums stand in the accumulators (nl columns listed): room in the
// pool, the fragment, the statistics, and -- one exit, one publication -- its fragment word
template <bool PERSISTENT, bool W32, typename LDS>
__device__ __forceinline__ void sp_build_finish(const SpBuildArgs &b, LDS &L, int c, int g, int lane, SpStamp &st, SpWaveState &ws, uint32_t nl, unsigned long long ops,
                                                bool touched, bool failed)
{
	uint64_t *fout = b.frag + (uint64_t) c * b.nseg + g;
	st.mark(1);
	uint64_t word = 0;
	int cnt = 0;
	if (touched) {
		if (PERSISTENT)
			sp_dbg(b, lane, 4, (long long) c * b.nseg + g, failed ? 1 : 0);
		const int ub = (int) nl;          // touched columns: an upper bound of the entries (a sum that came back to zero gives none)
		const uint32_t sh = sp_hash((uint32_t) c, (uint32_t) g);
		unsigned long long *S = b.shard + (size_t) sh * SHARD_STRIDE;
		unsigned long long off = 0;
		const bool arenas = PERSISTENT && b.arena > 0;
		if (!failed && ub > 0) {
			if (arenas) {
				// room from the wave's own arena: no atomic on the path of a task (a new arena every few dozen fragments), and
				// the arena moves on by what was WRITTEN -- columns whose sums cancelled strand nothing
				if (ws.ar_cur + ub > ws.ar_end) {
					const long long want = (b.arena > ub) ? b.arena : ub;
					ws.ar_cur = (long long) l0_atomic_add_u64_ret(&b.shard[0], (unsigned long long) want);
					ws.ar_end = ws.ar_cur + want;
					if ((unsigned long long) ws.ar_end > (unsigned long long) SP_SHARDS * b.shard_sub) {
						ws.ar_end = ws.ar_cur;          // (the chunk is full)
						failed = true;
					}
				}
				off = (unsigned long long) ws.ar_cur;
			} else {
				off = l0_atomic_add_u64_ret(&S[0], (unsigned long long) ub);
				if (off + (unsigned long long) ub > (unsigned long long) (sh + 1) * b.shard_sub)
					failed = true;
			}
		}
		st.mark(4);
		if (failed) {
			sp_discard(L, nl, lane);
		} else if (ub > 0) {
			if constexpr (W32) {
				uint64_t *dst = reinterpret_cast<uint64_t *>(b.chunk_base) + off;
				cnt = sp_emit_sorted(L, nl, lane, [&](uint32_t pos, uint32_t col, uint32_t v) { sp_st64<PERSISTENT>(dst + pos, (uint64_t) col | ((uint64_t) v << 32)); });
			} else {
				// (values as they stand in the accumulators, |v| <= B: the readers' arithmetic takes them)
				uint32_t *dst = b.chunk_base + off;
				cnt = sp_emit_sorted(L, nl, lane, [&](uint32_t pos, uint32_t col, int v) { sp_st<PERSISTENT>(dst + pos, col | ((uint32_t) v << 16)); });
			}
			if (cnt > 0)
				word = ((uint64_t) b.chunk << (SP_LEN_BITS + SP_OFF_BITS)) | ((uint64_t) off << SP_LEN_BITS) | (uint64_t) cnt;
			if (arenas)
				ws.ar_cur += cnt;
		}
		st.mark(5);
		if (!failed) {
			if (PERSISTENT) {
				ws.ops += ops;
				ws.nnz += (unsigned long long) cnt;
				ws.frags += (cnt > 0) ? 1ull : 0ull;
				ws.reserved += (unsigned long long) (arenas ? cnt : ub);
			} else {
				l0_atomic_add_u64(&S[2], ops);
				l0_atomic_add_u64(&S[3], (unsigned long long) cnt);
				l0_atomic_add_u64(&S[4], (cnt > 0) ? 1ull : 0ull);
				l0_atomic_add_u64(&S[5], (unsigned long long) ub);
			}
		}
	}
	if (failed) {
		l0_atomic_smax_i32(b.ovf_level, PERSISTENT ? 1 : b.level);
		word = PERSISTENT ? FRAG_FAILED : 0;
	}
	sp_publish<PERSISTENT>(fout, word, lane);
	st.mark(6);
}


template <bool PERSISTENT, bool W32, typename LDS>
__device__ __forceinline__ void sp_build_task(const SpBuildArgs &b, LDS &L, int c, int g, int lane, SpStamp &st, SpWaveState &ws, const uint2 hd)
{
	const uint32_t col0 = (uint32_t) g * SP_SEG;
	const SgnDev G = b.G;
	const MontDev M = b.M;
	uint64_t *fout = b.frag + (uint64_t) c * b.nseg + g;
	unsigned long long ops = 0;
	uint32_t nl = 0;                               // columns of the segment listed so far (sp_list_new)
	bool touched = false, failed = false;          // (wave-uniform: they only ever change on ballots)
	// up to 64 non-pivotal entries of the row (lane: index among the non-pivotal columns, value): those of this segment
	auto own_entries = [&](bool have, uint32_t x, int val) {
		const uint32_t idx = x - col0;
		const bool in = have && idx < (uint32_t) SP_SEG;
		if (__ballot(in) != 0) {
			touched = true;
			if constexpr (W32)
				sp_own_entry(L, idx, (uint32_t) val, in, M, nl);
			else
				sp_own_entry(L, idx, val, in, G, nl);
		}
	};
	// up to 64 pivotal entries of the row (lane: compact row of the pivot, negated coefficient): minus coefficient times their fragments
	auto dependencies = [&](bool have, uint32_t row, int coef, int detail) {
		uint64_t f = 0;
		const uint64_t *fin = have ? b.frag + (uint64_t) row * b.nseg + g : b.frag;
		st.mark(1);
		if (PERSISTENT) {
			sp_dbg(b, lane, 2, (long long) c * b.nseg + g, detail);
			f = have ? __hip_atomic_load(fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
			long long polls = 0;
			while (__ballot(f == FRAG_PENDING) != 0) {
				polls += 1;
				int gave_up = (polls > b.poll_limit) ? 1 : 0;
				if ((polls & 255) == 0)
					gave_up |= __builtin_amdgcn_readfirstlane(__hip_atomic_load(b.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
				if (gave_up != 0) {
					l0_store_i32_sc1(b.abort_flag, 1);
					f = (f == FRAG_PENDING) ? FRAG_FAILED : f;
				} else {
					__builtin_amdgcn_s_sleep(1);
					if (f == FRAG_PENDING)
						f = __hip_atomic_load(fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			if (__ballot(f == FRAG_FAILED) != 0)
				failed = true;
		} else if (have) {
			f = *fin;
		}
		st.mark(2);
		if (!failed && __ballot((f & LEN_MASK) != 0) != 0) {
			touched = true;
			if (PERSISTENT)
				sp_dbg(b, lane, 3, (long long) c * b.nseg + g, detail);
			if constexpr (W32)
				sp_accumulate<PERSISTENT>(L, f, (uint32_t) coef, true, b.pools, lane, M, ops, nl);
			else
				sp_accumulate<PERSISTENT>(L, f, coef, true, b.pools, lane, G, ops, nl);
			st.mark(3);
		}
	};
	// The row's lists.  Rows of U' are short (mk15.b4: five entries on average): a row of at most seven entries has them ALL in
	// its 64-byte head -- word 0: the counts, then the pivotal entries, then the non-pivotal ones -- one trip to memory where the
	// four row pointers and the two lists took three, one after the other (metadata: 26 % of the build's wave-cycles).
	const uint32_t counts = (uint32_t) __builtin_amdgcn_readfirstlane((int) hd.x);
	if (counts != 0xFFFFFFFFu) {
		const int nd = (int) (counts & 0xFFFFu), nn = (int) (counts >> 16);
		own_entries(lane > nd && lane <= nd + nn, hd.x, (int) hd.y);
		if (nd > 0)
			dependencies(lane >= 1 && lane <= nd, hd.x, (int) hd.y, 0);
	} else {
		const uint64_t d0 = sp_uniform(b.dep_rp[c]), d1 = sp_uniform(b.dep_rp[c + 1]), n0 = sp_uniform(b.np_rp[c]), n1 = sp_uniform(b.np_rp[c + 1]);
		for (uint64_t e = n0; e < n1; e += 64) {
			const bool have = e + lane < n1;
			const uint2 en = have ? b.np[e + lane] : uint2{0u, 0u};
			own_entries(have, en.x, (int) en.y);
		}
		for (uint64_t e = d0; e < d1; e += 64) {
			const bool have = e + lane < d1;
			const uint2 de = have ? b.dep[e + lane] : uint2{0u, 0u};
			dependencies(have, de.x, (int) de.y, (int) (e - d0));
		}
	}
	sp_build_finish<PERSISTENT, W32>(b, L, c, g, lane, st, ws, nl, ops, touched, failed);
}

// A row whose lists stand in its head (at most seven entries: lanes 1..nd its pivotal entries, the next nn lanes its non-pivotal
// ones), through the segments of `mask`.  While segment g runs, the fragment words of the dependencies for the NEXT segment are
// already asked for; when they have all arrived by the time g's sums are formed, the fragments they name are fetched before g is
// emitted -- their latency runs beside the emit, the drain of its stores and the publication.
template <bool W32, typename LDS>
__device__ __forceinline__ void sp_build_row_inline(const SpBuildArgs &b, LDS &L, int c, const uint2 hd, uint64_t mask, int lane, SpStamp &st, SpWaveState &ws)
{
	using ENT = typename std::conditional<W32, uint64_t, uint32_t>::type;
	using COEF = typename std::conditional<W32, uint32_t, int>::type;
	const SgnDev G = b.G;
	const MontDev M = b.M;
	const uint32_t counts = (uint32_t) __builtin_amdgcn_readfirstlane((int) hd.x);
	const int nd = (int) (counts & 0xFFFFu), nn = (int) (counts >> 16);
	const bool isdep = lane >= 1 && lane <= nd, isnp = lane > nd && lane <= nd + nn;
	const uint64_t *fin = b.frag + (isdep ? (uint64_t) hd.x * b.nseg : 0);
	const uint64_t *fown = b.frag + (uint64_t) c * b.nseg;
	const COEF coef = (COEF) hd.y;
	int g = __builtin_ctzll(mask);
	uint64_t m = mask & (mask - 1);
	uint64_t f = isdep ? __hip_atomic_load(fin + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
	SpGroup<ENT, COEF> pre;
	bool have_pre = false;
	for (;;) {
		const int gn = (m != 0) ? __builtin_ctzll(m) : -1;
		uint64_t fn = (isdep && gn >= 0) ? __hip_atomic_load(fin + gn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
		bool skip = false;
		if (b.retry != 0)          // (a launch that retries after a pool extension finds most fragments done)
			skip = sp_uniform(__hip_atomic_load(fown + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != FRAG_PENDING;
		if (!skip) {
			unsigned long long ops = 0;
			uint32_t nl = 0;
			bool touched = false, failed = false;
			st.mark(1);
			sp_dbg(b, lane, 2, (long long) c * b.nseg + g, 0);
			long long polls = 0;
			while (__ballot(f == FRAG_PENDING) != 0) {
				polls += 1;
				int gave_up = (polls > b.poll_limit) ? 1 : 0;
				if ((polls & 255) == 0)
					gave_up |= __builtin_amdgcn_readfirstlane(__hip_atomic_load(b.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
				if (gave_up != 0) {
					l0_store_i32_sc1(b.abort_flag, 1);
					f = (f == FRAG_PENDING) ? FRAG_FAILED : f;
				} else {
					__builtin_amdgcn_s_sleep(1);
					if (f == FRAG_PENDING)
						f = __hip_atomic_load(fin + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			if (__ballot(f == FRAG_FAILED) != 0)
				failed = true;
			st.mark(2);
			const uint32_t idx = hd.x - (uint32_t) g * SP_SEG;
			const bool in = isnp && idx < (uint32_t) SP_SEG;
			if (__ballot(in) != 0) {
				touched = true;
				if constexpr (W32)
					sp_own_entry(L, idx, (uint32_t) hd.y, in, M, nl);
				else
					sp_own_entry(L, idx, (int) hd.y, in, G, nl);
			}
			if (!failed && __ballot(isdep && (f & LEN_MASK) != 0) != 0) {
				touched = true;
				sp_dbg(b, lane, 3, (long long) c * b.nseg + g, 0);
				if constexpr (W32) {
					if (have_pre)
						sp_group_consume<true>(L, pre, f, coef, b.pools, lane, M, ops, nl);
					else
						sp_accumulate<true>(L, f, coef, isdep, b.pools, lane, M, ops, nl);
				} else {
					if (have_pre)
						sp_group_consume<true>(L, pre, f, coef, b.pools, lane, G, ops, nl);
					else
						sp_accumulate<true>(L, f, coef, isdep, b.pools, lane, G, ops, nl);
				}
				st.mark(3);
			}
			have_pre = false;
			if (gn >= 0 && __ballot(fn == FRAG_PENDING || fn == FRAG_FAILED) == 0 && __ballot(isdep && (fn & LEN_MASK) != 0) != 0) {
				sp_group_issue<true>(pre, fn, coef, isdep, b.pools, lane);
				have_pre = true;
			}
			sp_build_finish<true, W32>(b, L, c, g, lane, st, ws, nl, ops, touched, failed);
		} else {
			have_pre = false;
		}
		if (gn < 0)
			break;
		g = gn;
		m &= m - 1;
		f = fn;
	}
}

template <bool PERSISTENT, bool W32 = false> __global__ __launch_bounds__(64) void sp_build_kernel(SpBuildArgs b)
{
	__shared__ typename std::conditional<W32, WaveLds32, WaveLds>::type L;
	const int lane = threadIdx.x;
	sp_lds_init(L, lane);
	SpStamp st;
	st.begin(b.prof);
	if constexpr (!PERSISTENT) {
		const int task = blockIdx.x;
		const int c = b.row_lo + task / b.nseg;
		SpWaveState ws;
		sp_build_task<false, W32>(b, L, c, task - (c - b.row_lo) * b.nseg, lane, st, ws, sp_row_head(b, c, lane));
	} else {
		// A ticket is a ROW (round 6; a (row, segment) pair before): the wave reads the row's lists once and takes it through its
		// segments, publishing every fragment word as it goes -- a row that depends on this one walks the same segments in the same
		// order, one behind: the chain of levels advances by a segment (2-3 us), not by a task with its ticket, its lists and its
		// publication (7.5 us), and there are 18 times fewer tickets.  Rows are handed out from the last to the first, i.e. every row
		// after the rows it depends on: whoever waits, waits for a row with a larger number, which a running wave holds or has finished.
		const int q = (int) (blockIdx.x % SP_TICKETS);
		const long long ntasks = (long long) (b.row_hi - b.row_lo);
		SpWaveState ws;
		for (;;) {
			const int j = l0_atomic_add_i32_ret(b.ticket + q * SP_TICKET_STRIDE, 1);
			const long long t = (long long) j * SP_TICKETS + q;
			st.mark(0);
			sp_dbg(b, lane, 1, t, j);
			if (t >= ntasks)
				break;
			const int c = b.row_hi - 1 - (int) t;          // rows from the last to the first
			const uint2 hd = sp_row_head(b, c, lane);
			const uint64_t *fown = b.frag + (uint64_t) c * b.nseg;
			const bool inline_row = (uint32_t) __builtin_amdgcn_readfirstlane((int) hd.x) != 0xFFFFFFFFu;
			if (b.segmask != nullptr && inline_row) {
				const uint64_t m = sp_uniform(b.segmask[c]);
				if (m != 0)
					sp_build_row_inline<W32>(b, L, c, hd, m, lane, st, ws);
			} else if (b.segmask != nullptr) {
				// (the segments in which the row can hold anything: the words of the others were published as empty before the launch)
				for (uint64_t m = sp_uniform(b.segmask[c]); m != 0; m &= m - 1) {
					const int g = __builtin_ctzll(m);
					// (a launch that retries after a pool extension finds most fragments done)
					const uint64_t cur = sp_uniform(__hip_atomic_load(fown + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
					if (cur == FRAG_PENDING)
						sp_build_task<true, W32>(b, L, c, g, lane, st, ws, hd);
				}
			} else {
				for (int g = 0; g < b.nseg; g++) {
					const uint64_t cur = sp_uniform(__hip_atomic_load(fown + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
					if (cur == FRAG_PENDING)
						sp_build_task<true, W32>(b, L, c, g, lane, st, ws, hd);
				}
			}
			sp_dbg(b, lane, 5, t, 0);
		}
		sp_dbg(b, lane, 9, 0, 0);
		// the wave's share of the statistics (and what its last arena strands counts as reserved)
		unsigned long long *S = b.shard + (size_t) (blockIdx.x % SP_SHARDS) * SHARD_STRIDE;
		l0_atomic_add_u64(&S[2], ws.ops);
		l0_atomic_add_u64(&S[3], ws.nnz);
		l0_atomic_add_u64(&S[4], ws.frags);
		l0_atomic_add_u64(&S[5], ws.reserved + (unsigned long long) (ws.ar_end - ws.ar_cur));
	}
	st.flush(lane);
}

// the 64-byte head of every row (sp_build_task): word 0 = (pivotal entries | non-pivotal entries << 16, 0), then the pivotal entries
// (compact row, coefficient), then the non-pivotal ones (index, value); a row of more than seven entries keeps its lists (word 0 = all ones)
__global__ __launch_bounds__(256) void sp_make_heads_kernel(const uint64_t *dep_rp, const uint2 *dep, const uint64_t *np_rp, const uint2 *np, int r, uint2 *head)
{
	const int n = blockIdx.x * 256 + threadIdx.x;
	if (n >= r)
		return;
	const uint64_t d0 = dep_rp[n], nd = dep_rp[n + 1] - d0, n0 = np_rp[n], nn = np_rp[n + 1] - n0;
	uint2 *h = head + (size_t) 8 * (size_t) n;
	if (nd + nn > 7) {
		h[0] = uint2{0xFFFFFFFFu, 0u};
		return;
	}
	h[0] = uint2{(uint32_t) nd | ((uint32_t) nn << 16), 0u};
	for (uint64_t e = 0; e < nd; e++)
		h[1 + e] = dep[d0 + e];
	for (uint64_t e = 0; e < nn; e++)
		h[1 + nd + e] = np[n0 + e];
	for (uint64_t e = 1 + nd + nn; e < 8; e++)
		h[e] = uint2{0u, 0u};
}

// which segments of every row of R hold anything: one bit per segment, (nseg + 63) / 64 words per row
__global__ __launch_bounds__(256) void sp_rowmask_kernel(const uint64_t *frag, int r, int nseg, uint64_t *mask)
{
	const int nmw = (nseg + 63) >> 6;
	for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < (int64_t) r * nmw; t += (int64_t) gridDim.x * blockDim.x) {
		const int64_t c = t / nmw;
		const int b = (int) (t - c * nmw);
		uint64_t m = 0;
		for (int g = 64 * b; g < nseg && g < 64 * b + 64; g++)
			m |= ((frag[(uint64_t) c * nseg + g] & LEN_MASK) != 0 ? 1ull : 0ull) << (g & 63);
		mask[t] = m;
	}
}

// entries of R: the lengths of all fragments (after a build that redid levels the counters of the shards count those twice)
__global__ __launch_bounds__(256) void sp_sum_frag_kernel(const uint64_t *frag, int64_t n, unsigned long long *out)
{
	unsigned long long s = 0;
	for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t) gridDim.x * blockDim.x)
		s += frag[t] & LEN_MASK;
	for (int sft = 32; sft >= 1; sft >>= 1)
		s += ((unsigned long long) (uint32_t) __shfl_xor((int) (uint32_t) (s >> 32), sft) << 32) + (uint32_t) __shfl_xor((int) (uint32_t) s, sft);
	if ((threadIdx.x & 63) == 0)
		atomicAdd(out, s);
}

// census of R for DESIGN.md: entries, occupied 64-column tiles (what a tile-sparse dense form would store), fragments
template <bool W32> __global__ __launch_bounds__(256) void sp_census_kernel(const uint64_t *frag, int64_t n, SpPools pools, unsigned long long *out)
{
	const int lane = threadIdx.x & 63;
	const int64_t wave = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t) gridDim.x * blockDim.x) >> 6;
	unsigned long long entries = 0, tiles = 0, frags = 0;
	for (int64_t t = wave; t < n; t += nwaves) {
		const uint64_t f = frag[t];
		const int len = (int) (f & LEN_MASK);
		if (len == 0)
			continue;
		const uint32_t *src = frag_ptr(pools, f);
		const uint64_t *src64 = frag_ptr64(pools, f);
		frags += (lane == 0);
		for (int i = lane; i < len; i += 64) {
			const uint32_t c = (W32 ? (uint32_t) src64[i] : src[i]) & 0xFFFFu;
			const uint32_t prev = (i > 0) ? ((W32 ? (uint32_t) src64[i - 1] : src[i - 1]) & 0xFFFFu) : 0xFFFFFFFFu;
			entries += 1;
			tiles += (i == 0) || ((c >> 6) != (prev >> 6));
		}
	}
	for (int sft = 32; sft >= 1; sft >>= 1) {
		entries += (unsigned long long) (uint32_t) __shfl_xor((int) (uint32_t) entries, sft) | ((unsigned long long) (uint32_t) __shfl_xor((int) (uint32_t) (entries >> 32), sft) << 32);
		tiles += (unsigned long long) (uint32_t) __shfl_xor((int) (uint32_t) tiles, sft) | ((unsigned long long) (uint32_t) __shfl_xor((int) (uint32_t) (tiles >> 32), sft) << 32);
		frags += (unsigned long long) (uint32_t) __shfl_xor((int) (uint32_t) frags, sft) | ((unsigned long long) (uint32_t) __shfl_xor((int) (uint32_t) (frags >> 32), sft) << 32);
	}
	if (lane == 0) {
		atomicAdd(&out[0], entries);
		atomicAdd(&out[1], tiles);
		atomicAdd(&out[2], frags);
	}
}

// ---------------------------------------------------------------------------------------------------
// rows of S
// ---------------------------------------------------------------------------------------------------
#ifndef SPASM_SP_ROW_BATCH
#define SPASM_SP_ROW_BATCH 4
#endif
constexpr int SP_ROW_BATCH = SPASM_SP_ROW_BATCH;          // rows of a ticket of sp_apply_kernel

struct SpApplyArgs {
	SchurArgs a;
	const int *col;               // column -> compact id of its pivot row, or r + index among the non-pivotal columns
	int r, nseg;
	const uint64_t *frag;
	const uint64_t *rowmask;      // per row of R, (nseg + 63) / 64 words: its non-empty segments
	SpPools pools;
	SgnDev G;
	uint32_t *fpool;              // fragments of S (column inside the segment | value << 16, values in [-p/2, p/2])
	uint32_t *fpool_v;            // the 32-bit variant: fpool holds the columns, this the values in [0, p)
	int64_t fcap;
	uint64_t *T;                  // nseg x nrows (segment-major: the gather walks a segment): offset << SP_LEN_BITS | length of the fragment of (row, segment)
	uint32_t *D;                  // nseg x nrows: entries of the row in the segments before this one (where the gather puts the fragment)
	unsigned long long *block_sum;// sum of the lengths of every block of 1024 rows (zeroed before the launch)
	int arena;                    // entries a wave reserves from the pool at a time (0: every fragment on its own)
	int *ticket;                  // SP_TICKETS counters handing out the rows (zeroed before the launch)
	unsigned long long *prof;     // SPASM_HIP_SPARSE_IMAGE_PROFILE=1: 8 cycle counters (row of A, fragment words, adding, reservation, emit, words out)
};

// One wave per row of the batch, segment after segment: the entries of the row of A are read and relabelled once (rows
// of at most 64 entries -- longer ones go through them once per segment), the fragment words of segment g + 1 are in
// flight while segment g is added up, and the row's fragments of S land one behind the other in the wave's arena.  The
// words of T and D stay in registers (lane g holds those of segment g) and leave in one store per row.
template <bool W32> __global__ __launch_bounds__(64) void sp_apply_kernel(SpApplyArgs d)
{
	__shared__ typename std::conditional<W32, WaveLds32, WaveLds>::type L;
	const SchurArgs &a = d.a;
	const int lane = threadIdx.x;
	const SgnDev G = d.G;
	const MontDev F = a.F;
	const int nrows = a.nrows, nseg = d.nseg;
	unsigned long long st_input = 0, ops = 0;
	int st_done = 0, st_piv = 0;
	long long ar_cur = 0, ar_end = 0;            // the wave's arena in the fragment pool
	bool pool_full = false;
	sp_lds_init(L, lane);
	SpStamp st;
	st.begin(d.prof);
	// what a touched (row, segment) leaves behind: its fragment in the pool; returns its word of T and the number of entries
	auto finish = [&](uint32_t nl, int &cnt) -> uint64_t {
		const int ub = (int) nl;
		cnt = 0;
		if (ar_cur + ub > ar_end) {
			const long long want = (d.arena > ub) ? d.arena : ub;
			const unsigned long long got = l0_atomic_add_u64_ret(&a.ctr64[C64_POOL], (unsigned long long) want);
			ar_cur = (long long) got;
			ar_end = ar_cur + want;
			if (ar_end > d.fcap) {
				pool_full = true;
				ar_end = ar_cur;
			}
		}
		if (ar_cur + ub > ar_end) {
			sp_discard(L, nl, lane);
			return 0;
		}
		st.mark(3);
		if constexpr (W32) {
			uint32_t *oc = d.fpool + ar_cur, *ov = d.fpool_v + ar_cur;
			cnt = sp_emit_sorted(L, nl, lane, [&](uint32_t pos, uint32_t col, uint32_t v) {
				oc[pos] = col;
				ov[pos] = v;
			});
		} else {
			uint32_t *out = d.fpool + ar_cur;
			cnt = sp_emit_sorted(L, nl, lane, [&](uint32_t pos, uint32_t col, int v) { out[pos] = col | ((uint32_t) sgn_canonical(v, G) << 16); });
		}
		st.mark(4);
		const uint64_t word = (cnt > 0) ? (((uint64_t) ar_cur << SP_LEN_BITS) | (uint64_t) cnt) : 0;
		ar_cur += cnt;
		return word;
	};
	// rows are handed out by ticket counters (their costs differ by orders of magnitude: with a fixed share per wave the
	// kernel ran 1.6 times as long as its average wave)
	const int q = (int) (blockIdx.x % SP_TICKETS);
	// A ticket is a batch of SP_ROW_BATCH consecutive rows, and what the rows need before their first multiply-add -- row number,
	// extent, entries, the pivot rows of their columns, the occupied segments of those -- is fetched for the whole batch at once:
	// five trips to memory per batch where every row made them one after the other (7 % of the kernel's wave-cycles).
	constexpr int RB = SP_ROW_BATCH;
	for (;;) {
		const long long k0 = ((long long) l0_atomic_add_i32_ret(d.ticket + q * SP_TICKET_STRIDE, 1) * SP_TICKETS + q) * RB;
		if (k0 >= nrows)
			break;
		const bool rl = lane < RB && k0 + lane < nrows;
		const int i_l = rl ? a.rows[k0 + lane] : 0;
		const int64_t lo_l = rl ? a.Ap[i_l] : 0, hi_l = rl ? a.Ap[i_l + 1] : 0;
		int64_t lo_u[RB], hi_u[RB];
		int aj_u[RB], ax_u[RB];
		uint32_t cid_u[RB];
		uint64_t rm_u[RB];
		const int nmw0 = (nseg + 63) >> 6;
#pragma unroll
		for (int u = 0; u < RB; u++) {
			lo_u[u] = (int64_t) readlane64((uint64_t) lo_l, u);
			hi_u[u] = (int64_t) readlane64((uint64_t) hi_l, u);
			const bool have = hi_u[u] - lo_u[u] <= 64 && lo_u[u] + lane < hi_u[u];
			aj_u[u] = have ? a.Aj[lo_u[u] + lane] : -1;
			ax_u[u] = have ? a.Ax[lo_u[u] + lane] : 0;
		}
#pragma unroll
		for (int u = 0; u < RB; u++)
			cid_u[u] = (aj_u[u] >= 0) ? (uint32_t) d.col[aj_u[u]] : 0xFFFFFFFFu;
#pragma unroll
		for (int u = 0; u < RB; u++)
			rm_u[u] = (cid_u[u] < (uint32_t) d.r) ? d.rowmask[(uint64_t) cid_u[u] * nmw0] : 0;
#pragma unroll 1
		for (int u = 0; u < RB && k0 + u < nrows; u++) {
		const int k = (int) k0 + u;
		// (u is wave-uniform: the selections are scalar compares)
		int64_t lo = lo_u[0], hi = hi_u[0];
		uint32_t cid_pre = cid_u[0];
		int ax_pre = ax_u[0];
		uint64_t rm_pre = rm_u[0];
#pragma unroll
		for (int v = 1; v < RB; v++)
			if (u == v) {
				lo = lo_u[v];
				hi = hi_u[v];
				cid_pre = cid_u[v];
				ax_pre = ax_u[v];
				rm_pre = rm_u[v];
			}
		st_input += (unsigned long long) (hi - lo);
		int total = 0;
		uint64_t tw = 0;          // lane (g mod 64) holds the word of T of segment g ...
		uint32_t td = 0;          // ... and the entries of the row before it
		// (T and D are segment-major: row k's words of segments [64 b, 64 b + 64) leave in one store, one line per lane)
		auto keep = [&](int g, uint64_t word, int cnt) {
			if (lane == (g & 63)) {
				tw = word;
				td = (uint32_t) total;
			}
			total += cnt;
		};
		auto flush = [&](int b) {
			const int g = 64 * b + lane;
			if (g < nseg) {
				d.T[(uint64_t) g * nrows + k] = tw;
				d.D[(uint64_t) g * nrows + k] = td;
			}
			tw = 0;
			td = 0;
		};
		if (hi - lo <= 64) {
			// the row in registers: lane e holds entry e
			const uint32_t cid = cid_pre;
			int bal = 0;          // the entry: balanced (16-bit variant) / plain residue (32-bit variant)
			int ncoef = 0;        // minus the entry, as the arithmetic wants a coefficient: negated balanced / Montgomery form of p - a
			if (lo + lane < hi) {
				const uint32_t av = reduce_sum(from_balanced(ax_pre, F), F);
				if constexpr (W32) {
					bal = (int) av;
					ncoef = (int) montmul(av == 0 ? 0u : F.p - av, F.r2, F);
				} else {
					bal = sgn_from_residue(av, G);
					ncoef = -bal;
				}
			}
			const bool piv = cid < (uint32_t) d.r && bal != 0;
			const bool own = cid != 0xFFFFFFFFu && cid >= (uint32_t) d.r;
			const uint32_t idx_all = cid - (uint32_t) d.r;          // (own entries: index among the non-pivotal columns)
			const int own_seg = own ? (int) (idx_all / (uint32_t) SP_SEG) : -1;
			const uint64_t *fin = d.frag + (piv ? (uint64_t) cid * nseg : 0);
			st_piv += piv ? 1 : 0;
			using ENT = typename std::conditional<W32, uint64_t, uint32_t>::type;
			using COEF = typename std::conditional<W32, uint32_t, int>::type;
			const int nmw = (nseg + 63) >> 6;
			const uint64_t piv_lanes = __ballot(piv), own_lanes = __ballot(own);
			for (int b = 0; b < nmw; b++) {
				// the segments of this block of 64 in which the row can hold anything: where one of its rows of R has entries,
				// and where its own non-pivotal entries fall -- no other segment is visited
				const uint64_t rm = (b == 0) ? rm_pre : (piv ? d.rowmask[(uint64_t) cid * nmw + b] : 0);
				uint64_t mask = 0;
				for (uint64_t pl = piv_lanes; pl != 0; pl &= pl - 1)
					mask |= readlane64(rm, __builtin_ctzll(pl));
				for (uint64_t ol = own_lanes; ol != 0; ol &= ol - 1) {
					const int sg = __builtin_amdgcn_readlane(own_seg, __builtin_ctzll(ol)) - 64 * b;
					if (sg >= 0 && sg < 64)
						mask |= 1ull << sg;
				}
				st.mark(0);
				if (mask != 0) {
					int g = 64 * b + __builtin_ctzll(mask);
					uint64_t m = mask & (mask - 1);
					int gn = (m != 0) ? 64 * b + __builtin_ctzll(m) : -1;
					uint64_t f = piv ? fin[g] : 0;
					uint64_t fn = (piv && gn >= 0) ? fin[gn] : 0;
					SpGroup<ENT, COEF> cur, nxt;
					sp_group_issue<false>(cur, f, (COEF) ncoef, piv, d.pools, lane);
					for (;;) {
						const uint64_t m2 = (m != 0) ? (m & (m - 1)) : 0;
						const int gnn = (m2 != 0) ? 64 * b + __builtin_ctzll(m2) : -1;
						const uint64_t fnn = (piv && gnn >= 0) ? fin[gnn] : 0;
						sp_group_issue<false>(nxt, fn, (COEF) ncoef, piv, d.pools, lane);          // (after the last segment: fn = 0, eight loads of the pool's first word)
						const uint32_t idx = idx_all - (uint32_t) g * SP_SEG;
						const bool in = own && idx < (uint32_t) SP_SEG;
						st.mark(1);
						uint32_t nl = 0;
						int cnt = 0;
						if constexpr (W32) {
							if (__ballot(in) != 0)
								sp_own_entry(L, idx, (uint32_t) bal, in, F, nl);
							sp_group_consume<false>(L, cur, f, (COEF) ncoef, d.pools, lane, F, ops, nl);
						} else {
							if (__ballot(in) != 0)
								sp_own_entry(L, idx, bal, in, G, nl);
							sp_group_consume<false>(L, cur, f, (COEF) ncoef, d.pools, lane, G, ops, nl);
						}
						st.mark(2);
						const uint64_t word = (nl != 0) ? finish(nl, cnt) : 0;
						keep(g, word, cnt);
						st.mark(5);
						if (gn < 0)
							break;
						g = gn;
						gn = gnn;
						m = m2;
						f = fn;
						fn = fnn;
						cur = nxt;
					}
				}
				flush(b);
			}
		} else {
			for (int g = 0; g < nseg; g++) {
				const uint32_t col0 = (uint32_t) g * SP_SEG;
				uint32_t nl = 0;
				bool touched = false;
				for (int64_t base = lo; base < hi; base += 64) {
					uint64_t f = 0;
					int bal = 0, ncoef = 0;
					uint32_t idx = 0xFFFFFFFFu;
					bool piv = false;
					if (base + lane < hi) {
						const uint32_t cid = (uint32_t) d.col[a.Aj[base + lane]];
						const uint32_t av = reduce_sum(from_balanced(a.Ax[base + lane], F), F);
						if constexpr (W32) {
							bal = (int) av;
							ncoef = (int) montmul(av == 0 ? 0u : F.p - av, F.r2, F);
						} else {
							bal = sgn_from_residue(av, G);
							ncoef = -bal;
						}
						if (cid >= (uint32_t) d.r) {
							idx = cid - (uint32_t) d.r - col0;
						} else if (bal != 0) {
							piv = true;
							f = d.frag[(uint64_t) cid * nseg + g];
							st_piv += (g == 0) ? 1 : 0;
						}
					}
					const bool in = idx < (uint32_t) SP_SEG;
					if ((__ballot(in) | __ballot((f & LEN_MASK) != 0)) == 0)
						continue;
					touched = true;
					if constexpr (W32) {
						if (__ballot(in) != 0)
							sp_own_entry(L, idx, (uint32_t) bal, in, F, nl);
						sp_accumulate<false>(L, f, (uint32_t) ncoef, piv, d.pools, lane, F, ops, nl);
					} else {
						if (__ballot(in) != 0)
							sp_own_entry(L, idx, bal, in, G, nl);
						sp_accumulate<false>(L, f, ncoef, piv, d.pools, lane, G, ops, nl);
					}
				}
				uint64_t word = 0;
				int cnt = 0;
				if (touched)
					word = finish(nl, cnt);
				keep(g, word, cnt);
				if ((g & 63) == 63 || g + 1 == nseg)
					flush(g >> 6);
			}
		}
		st_done += 1;
		l0_store_i32(a.row_len + k, total);
		l0_atomic_add_u64(&d.block_sum[k >> 10], (unsigned long long) total);
		}
	}
	st.flush(lane);
	for (int sft = 32; sft >= 1; sft >>= 1)
		st_piv += __shfl_xor(st_piv, sft);
	if (lane == 0) {
		atomicAdd(&a.ctr64[C64_INPUT], st_input);
		atomicAdd(&a.ctr64[C64_STREAM], ops);
		atomicAdd(&a.ctr64[C64_ELIM], (unsigned long long) st_piv);
		atomicAdd(&a.ctr[a.done_ctr], st_done);
		if (pool_full)
			atomicOr(&a.ctr[CTR_STATUS], 1);
	}
}

struct SpGatherArgs {
	const uint64_t *T;
	const uint32_t *D;
	const uint32_t *fpool;
	const uint32_t *fpool_v;      // 32-bit variant: the values (fpool: the columns)
	uint32_t p;
	int nrows, nseg, Sm;
	const int64_t *Sp;
	int *Sj, *Sx;
	int64_t cap;
	const int *q;                 // index among the non-pivotal columns -> column
};

// The fragments of S as (column, value) pairs at their final place.  A workgroup takes SP_GATHER_ROWS rows through the segments one
// after the other, with the 4,096 columns of the segment's piece of q in LDS (by row, every entry paid a trip to the L1 for its
// column: 9.9 -> 6.4 ms measured without the look-up in round 5); where a fragment goes inside its row is D, written by the apply
// kernel.  The pieces of a row are written by ONE workgroup, one behind the other: the lines they share meet in one L2 (a grid of
// (segment, row block) workgroups -- the first version of this kernel -- had them written from eight: 6.9 ms against round 5's 4.4
// on mk14.b4).
constexpr int SP_GATHER_ROWS = 256;
#ifndef SPASM_SP_GATHER_UNROLL
#define SPASM_SP_GATHER_UNROLL 8
#endif
constexpr int SP_GATHER_UNROLL = SPASM_SP_GATHER_UNROLL;
template <bool W32> __global__ __launch_bounds__(256) void sp_gather_kernel(SpGatherArgs e)
{
	__shared__ int qs[SP_SEG];
	const int k_lo = (int) blockIdx.x * SP_GATHER_ROWS;
	const int k_hi = (k_lo + SP_GATHER_ROWS < e.nrows) ? k_lo + SP_GATHER_ROWS : e.nrows;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int k = k_lo + 64 * wave + lane;
	int64_t off = 0;
	bool row_ok = false;
	if (k < k_hi) {
		off = e.Sp[k];
		const int64_t end = e.Sp[k + 1];
		row_ok = end <= e.cap && end != off;
	}
	for (int g = 0; g < e.nseg; g++) {
		const int ncols = (e.Sm - g * SP_SEG < SP_SEG) ? e.Sm - g * SP_SEG : SP_SEG;
		__syncthreads();
		for (int t = threadIdx.x; t < ncols; t += 256)
			qs[t] = e.q[(int64_t) g * SP_SEG + t];
		__syncthreads();
		uint64_t t = 0;
		int64_t dst = 0;
		if (row_ok) {
			t = e.T[(uint64_t) g * e.nrows + k];
			dst = off + (int64_t) e.D[(uint64_t) g * e.nrows + k];
		}
		uint64_t live = __ballot((t & LEN_MASK) != 0);
		// SP_GATHER_UNROLL fragments at a time, their first 64 entries in flight together (one at a time, a wave spent its time waiting for
		// one load after the other: 40 M fragments of 20-60 entries)
		while (live != 0) {
			const uint32_t *src[SP_GATHER_UNROLL];
			int len[SP_GATHER_UNROLL];
			int64_t w[SP_GATHER_UNROLL];
			uint32_t h[SP_GATHER_UNROLL], hv[SP_GATHER_UNROLL];
#pragma unroll
			for (int u = 0; u < SP_GATHER_UNROLL; u++) {
				len[u] = 0;
				w[u] = 0;
				src[u] = e.fpool;
				if (live != 0) {
					const int s = __builtin_ctzll(live);
					live &= live - 1;
					const uint64_t tt = readlane64(t, s);
					w[u] = (int64_t) readlane64((uint64_t) dst, s);
					len[u] = (int) (tt & LEN_MASK);
					src[u] = e.fpool + (tt >> SP_LEN_BITS);
				}
				h[u] = src[u][lane < len[u] ? lane : 0];
				hv[u] = 0;
				if constexpr (W32)
					hv[u] = (e.fpool_v + (src[u] - e.fpool))[lane < len[u] ? lane : 0];
			}
#pragma unroll
			for (int u = 0; u < SP_GATHER_UNROLL; u++) {
				if (len[u] == 0)
					continue;
				int *oj = e.Sj + w[u], *ox = e.Sx + w[u];
				if constexpr (W32) {
					const uint32_t *sv = e.fpool_v + (src[u] - e.fpool);
					if (lane < len[u]) {
						oj[lane] = qs[h[u] & 0xFFFFu];
						ox[lane] = (hv[u] > e.p / 2) ? (int) (hv[u] - e.p) : (int) hv[u];          // balanced representative (spasm_ZZp)
					}
					for (int i = lane + 64; i < len[u]; i += 64) {
						const uint32_t v = sv[i];
						oj[i] = qs[src[u][i] & 0xFFFFu];
						ox[i] = (v > e.p / 2) ? (int) (v - e.p) : (int) v;
					}
				} else {
					if (lane < len[u]) {
						oj[lane] = qs[h[u] & 0xFFFFu];
						ox[lane] = (int) h[u] >> 16;
					}
					for (int i = lane + 64; i < len[u]; i += 64) {
						const uint32_t en = src[u][i];
						oj[i] = qs[en & 0xFFFFu];
						ox[i] = (int) en >> 16;
					}
				}
			}
		}
	}
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// every odd prime below 2^32: signed 16-bit entries up to 44,927, (column, 32-bit residue) entries beyond
// (SPASM_HIP_SPARSE_IMAGE_WIDE=1 takes the 32-bit variant for small primes too: tests)
bool sparse_image_possible(int64_t prime)
{
	return prime >= 3 && prime < ((int64_t) 1 << 32) && (prime & 1) != 0 && env_sp("SPASM_HIP_SPARSE_IMAGE", -1) != 0;
}

// dependency tables of the build: per compact row (level order) its pivotal entries (compact row, negated balanced
// coefficient) and its non-pivotal entries (index among the non-pivotal columns, balanced value)
// The tables as the host makes them.  Round 5: on a large factor they are made by a thread of their own while the caller goes on
// (sparse_image_plan_start) -- the driver's next step after planning a factor is the density sample of 100 rows, which never
// looks at them: 15 ms of mk15.b4's call that nobody waited for -- and reach the device when somebody asks for the image
// (sparse_image_planned).  The thread reads the FactPlan only (kof, lab, rp, ent, lvl_count: nothing ensure_row_tables or the
// deferred plan of the dense image write) and makes no HIP call.
struct SpPending {
	std::thread worker;
	double t_start = 0.0, t_done = 0.0;
	std::vector<int> colmap;
	std::vector<uint64_t> dep_rp, np_rp, segmask;
	std::vector<uint2> dep, np;
};

// what the path choice reads: known at once
void sparse_image_plan_sizes(const FactPlan &P, spasm_hip_dfact *F)
{
	SpImage &S = F->sp;
	S.r = P.r;
	S.Sm = P.m - P.r;
	S.nseg = (S.Sm + SP_SEG - 1) / SP_SEG;
	S.nlevels = P.nlevels;
	S.wide = !sgn_eligible(P.prime) || env_sp("SPASM_HIP_SPARSE_IMAGE_WIDE", 0) != 0;
	S.lvl_lo.assign((size_t) P.nlevels + 1, 0);
	for (int l = 0; l < P.nlevels; l++)
		S.lvl_lo[l + 1] = S.lvl_lo[l] + P.lvl_count[l];
	S.ndeps = P.ndeps;
	S.nnp = (int64_t) P.rp[P.rpad] - P.ndeps;
}

static void sparse_image_plan_host(const FactPlan &P, bool wide, SpPending &H)
{
	const int r = P.r, rpad = P.rpad, m = P.m;
	const int64_t prime = P.prime;
	std::vector<int> cid((size_t) (rpad > 0 ? rpad : 1), -1), label_of((size_t) (r > 0 ? r : 1), 0);
	{
		int n = 0;
		for (int c = 0; c < rpad; c++)
			if (P.kof[c] >= 0) {
				cid[c] = n;
				label_of[n] = c;
				n += 1;
			}
		if (n != r)
			die("sparse_image_plan: %d labelled rows, %d expected", n, r);
	}
	std::vector<int> &colmap = H.colmap;
	colmap.assign((size_t) (m > 0 ? m : 1), 0);
	for (int j = 0; j < m; j++)
		colmap[j] = (P.lab[j] < (uint32_t) rpad) ? cid[P.lab[j]] : r + (int) (P.lab[j] - (uint32_t) rpad);
	// the image keeps values in Montgomery form (value * 2^32 mod p): back to plain residues, then balanced
	uint64_t unmont = 1;
	{
		const uint64_t R1 = (uint64_t) ((1ull << 32) % (uint64_t) prime);
		int64_t t0 = 0, t1 = 1, r0 = prime, r1 = (int64_t) R1;
		while (r1 != 0) {
			const int64_t qq = r0 / r1;
			const int64_t t2 = t0 - qq * t1, r2 = r0 - qq * r1;
			t0 = t1;
			t1 = t2;
			r0 = r1;
			r1 = r2;
		}
		if (r0 != 1)
			die("sparse_image_plan: 2^32 is not invertible mod %lld", (long long) prime);
		unmont = (uint64_t) ((t0 % prime + prime) % prime);
	}
	auto balanced = [&](uint32_t y_mont) -> int32_t {
		const int64_t v = (int64_t) (((unsigned __int128) y_mont * unmont) % (uint64_t) prime);
		return (int32_t) ((v > prime / 2) ? v - prime : v);
	};
	// two passes over the rows of U', by a few threads on large factors (604,000 rows and 3 M entries on mk15.b4: 16-20 ms with
	// one thread and a push_back per entry, a 128-bit division per value): count, then fill.  The entries of boundary
	// matrices are +-1: their Montgomery forms are recognised, no division.
	std::vector<uint64_t> &dep_rp = H.dep_rp, &np_rp = H.np_rp;
	dep_rp.assign((size_t) r + 1, 0);
	np_rp.assign((size_t) r + 1, 0);
	const uint32_t mont_one = (uint32_t) ((1ull << 32) % (uint64_t) prime), mont_minus_one = (uint32_t) ((uint64_t) prime - mont_one);
	const int T = (r < 50000) ? 1 : std::max(1, std::min(8, usable_cpus()));
	auto for_rows = [&](auto &&body) { sh::pool_run(T, [&](int t) { body((int) ((int64_t) r * t / T), (int) ((int64_t) r * (t + 1) / T)); }); };
	for_rows([&](int n_lo, int n_hi) {
		for (int n = n_lo; n < n_hi; n++) {
			const int c = label_of[n];
			uint64_t nd = 0;
			for (uint64_t e = P.rp[c]; e < P.rp[c + 1]; e++)
				nd += P.ent[e].x < (uint32_t) rpad;
			dep_rp[n + 1] = nd;
			np_rp[n + 1] = (P.rp[c + 1] - P.rp[c]) - nd;
		}
	});
	for (int n = 0; n < r; n++) {
		dep_rp[n + 1] += dep_rp[n];
		np_rp[n + 1] += np_rp[n];
	}
	std::vector<uint2> &dep = H.dep, &np = H.np;
	dep.resize((size_t) dep_rp[r]);
	np.resize((size_t) np_rp[r]);
	for_rows([&](int n_lo, int n_hi) {
		for (int n = n_lo; n < n_hi; n++) {
			const int c = label_of[n];
			uint64_t wd = dep_rp[n], wn = np_rp[n];
			for (uint64_t e = P.rp[c]; e < P.rp[c + 1]; e++) {
				const uint2 en = P.ent[e];
				const bool pivotal = en.x < (uint32_t) rpad;
				if (wide) {
					// (the image's values ARE Montgomery forms: the negated coefficient as it stands, the non-pivotal value as a plain residue)
					if (pivotal)
						dep[wd++] = uint2{(uint32_t) cid[en.x], en.y == 0 ? 0u : (uint32_t) ((uint64_t) prime - en.y)};
					else
						np[wn++] = uint2{en.x - (uint32_t) rpad, en.y == mont_one ? 1u : en.y == mont_minus_one ? (uint32_t) (prime - 1)
						                                                         : (uint32_t) (((unsigned __int128) en.y * unmont) % (uint64_t) prime)};
					continue;
				}
				const int32_t v = en.y == mont_one ? 1 : en.y == mont_minus_one ? -1 : balanced(en.y);
				if (pivotal)
					dep[wd++] = uint2{(uint32_t) cid[en.x], (uint32_t) (-v)};
				else
					np[wn++] = uint2{en.x - (uint32_t) rpad, (uint32_t) v};
			}
		}
	});
	// Which (row, segment) pairs can hold anything at all: a row's own non-pivotal entries and the segments of the rows it depends
	// on (rows with larger compact ids: one pass from the last row to the first).  Half of the pairs of the generated families
	// are empty (mk15.b4: 47-49 %), and their tasks used to read the row's lists and poll its dependencies to find that out
	// (metadata: 38 % of the build's wave-cycles): their fragment words are published as empty before the build starts.
	std::vector<uint64_t> &segmask = H.segmask;
	segmask.clear();
	if ((m - r + SP_SEG - 1) / SP_SEG <= 64) {
		segmask.assign((size_t) r, 0);
		for (int n = r - 1; n >= 0; n--) {
			uint64_t mk = 0;
			for (uint64_t e = np_rp[n]; e < np_rp[n + 1]; e++)
				mk |= 1ull << (np[e].x / (uint32_t) SP_SEG);
			for (uint64_t e = dep_rp[n]; e < dep_rp[n + 1]; e++)
				mk |= segmask[dep[e].x];
			segmask[(size_t) n] = mk;
		}
	}
}

static void sparse_image_plan_device(SpImage &S, SpPending &H, int m, hipStream_t stream)
{
	const int r = S.r;
	if ((int64_t) H.dep.size() != S.ndeps || (int64_t) H.np.size() != S.nnp)
		die("sparse_image_plan: %zu + %zu entries in the tables, %lld + %lld counted by the plan of the factor", H.dep.size(), H.np.size(), (long long) S.ndeps, (long long) S.nnp);
	S.d_col = dalloc<int>(m);
	S.d_dep_rp = dalloc<uint64_t>((int64_t) r + 1);
	S.d_dep = dalloc<uint2>((int64_t) H.dep.size());
	S.d_np_rp = dalloc<uint64_t>((int64_t) r + 1);
	S.d_np = dalloc<uint2>((int64_t) H.np.size());
	S.d_head = dalloc<uint2>((int64_t) 8 * (r > 0 ? r : 1));
	upload(S.d_col, H.colmap, stream);
	upload(S.d_dep_rp, H.dep_rp, stream);
	upload(S.d_dep, H.dep, stream);
	upload(S.d_np_rp, H.np_rp, stream);
	upload(S.d_np, H.np, stream);
	if (!H.segmask.empty()) {
		S.d_segmask = dalloc<uint64_t>((int64_t) r);
		upload(S.d_segmask, H.segmask, stream);
	}
	// the 64-byte heads of the rows are made on the device, from the lists that are there now (on the host they were 38 MB to
	// fill and send for mk15.b4's factor: 5 of the 22 ms the caller of the first Schur complement partly waits for)
	if (r > 0)
		hipLaunchKernelGGL(sp_make_heads_kernel, dim3((unsigned) ((r + 255) / 256)), dim3(256), 0, stream, S.d_dep_rp, S.d_dep, S.d_np_rp, S.d_np, r, S.d_head);
	HIP_CHECK(hipStreamSynchronize(stream));          // the host vectors die here
	S.planned = true;
	S.valid = false;
	S.failed = false;
}

// the tables, here and now
void sparse_image_plan(const FactPlan &P, spasm_hip_dfact *F, hipStream_t stream)
{
	sparse_image_plan_sizes(P, F);
	SpPending H;
	sparse_image_plan_host(P, F->sp.wide, H);
	sparse_image_plan_device(F->sp, H, P.m, stream);
}

// ... or by a thread of their own: P must stay where it is until sparse_image_planned / sparse_image_free has been called
void sparse_image_plan_start(const FactPlan &P, spasm_hip_dfact *F)
{
	sparse_image_plan_sizes(P, F);
	auto H = std::make_shared<SpPending>();
	const bool wide = F->sp.wide;
	const FactPlan *plan = &P;
	SpPending *raw = H.get();
	H->t_start = wtime();
	H->worker = std::thread([plan, wide, raw]() {
		sparse_image_plan_host(*plan, wide, *raw);
		raw->t_done = wtime();
	});
	F->sp.pending = H;
	F->sp.pending_m = P.m;
}

// is there a plan?  (waits for the thread and sends its tables to the device the first time)
bool sparse_image_planned(const spasm_hip_dfact *F, hipStream_t stream)
{
	SpImage &S = F->sp;
	if (S.pending) {
		std::shared_ptr<SpPending> H = S.pending;
		const double t_ask = wtime();
		if (H->worker.joinable())
			H->worker.join();
		const double t_joined = wtime();
		S.pending.reset();
		sparse_image_plan_device(S, *H, S.pending_m, stream);
		if (verbose() >= 2)
			logmsg("[factor image] tables of the sparse image: %.1f ms by a thread beside the caller, who waited %.1f ms for them; upload %.1f ms\n", 1e3 * (H->t_done - H->t_start),
			       1e3 * (t_joined - t_ask), 1e3 * (wtime() - t_joined));
	}
	return S.planned;
}

// (a plan that is on its way counts for the decisions that only need to know whether there will be one)
bool sparse_image_plan_expected(const spasm_hip_dfact *F) { return F->sp.planned || (bool) F->sp.pending; }

static void sparse_image_drop_chunks(SpImage &S)
{
	for (int k = 0; k < S.nchunks; k++) {
		sh::big_free(S.d_chunk[k]);
		S.d_chunk[k] = nullptr;
		S.chunk_cap[k] = 0;
	}
	S.nchunks = 0;
}

void sparse_image_free(spasm_hip_dfact *F)
{
	SpImage &S = F->sp;
	if (S.pending) {
		if (S.pending->worker.joinable())
			S.pending->worker.join();
		S.pending.reset();
	}
	sparse_image_drop_chunks(S);
	sh::big_free(S.d_col);
	sh::big_free(S.d_dep_rp);
	sh::big_free(S.d_dep);
	sh::big_free(S.d_np_rp);
	sh::big_free(S.d_np);
	sh::big_free(S.d_segmask);
	sh::big_free(S.d_head);
	sh::big_free(S.d_rowmask);
	sh::big_free(S.d_frag);
	sh::big_free(S.d_shard);
	if (S.ev0 != nullptr)
		(void) hipEventDestroy(S.ev0);
	if (S.ev1 != nullptr)
		(void) hipEventDestroy(S.ev1);
	S = SpImage{};
}

// (re)computes the fragments of R on `stream`; synchronises the stream (the host has to learn whether the pool held).
// Returns false when R turned out not to be sparse (the pool budget ran out): the caller takes another path.
bool sparse_image_build(const spasm_hip_dfact *F, hipStream_t stream)
{
	SpImage &S = F->sp;
	if (!sparse_image_planned(F, stream))
		die("sparse_image_build: the factor has no plan for the sparse image");
	S.valid = false;
	if (S.ev0 == nullptr) {
		HIP_CHECK(hipEventCreate(&S.ev0));
		HIP_CHECK(hipEventCreate(&S.ev1));
	}
	const int64_t nfrag = (int64_t) S.r * S.nseg;
	if (S.d_frag == nullptr)
		S.d_frag = dalloc<uint64_t>(nfrag);
	if (S.d_shard == nullptr)
		S.d_shard = dalloc<unsigned long long>((int64_t) SP_SHARDS * SHARD_STRIDE + 512);
	// small words behind the shards: [0, 512) the ticket counters, 512 the abort flag, 513 the overflow word
	int *d_sync = reinterpret_cast<int *>(S.d_shard + (size_t) SP_SHARDS * SHARD_STRIDE);
	int *d_abort = d_sync + SP_TICKETS * SP_TICKET_STRIDE, *d_ovf = d_abort + 1;
	// Room: the pool grows by chunks.  A build that runs out of room allocates the next chunk (twice the size) and redoes
	// what failed; what it may take in all is bounded -- an R that needs more than half the bytes of its dense form is not
	// sparse, and the other paths are the better ones for it.
	size_t free_b = 0, total_b = 0;
	sh::mem_info(&free_b, &total_b);
	int64_t held = 0;
	const int64_t esize = S.wide ? 8 : 4;          // bytes of a fragment entry
	for (int k = 0; k < S.nchunks; k++)
		held += S.chunk_cap[k] * esize;
	int64_t budget = std::min<int64_t>((int64_t) ((free_b + (size_t) held) / 3), std::max<int64_t>((int64_t) S.r * (int64_t) S.Sm, (int64_t) 256 << 20));
	if (env_sp("SPASM_HIP_SPARSE_IMAGE_GB", 0) > 0)
		budget = (int64_t) env_sp("SPASM_HIP_SPARSE_IMAGE_GB", 0) << 30;
	// a rebuild of an image that needed several chunks: one chunk of the size that is known now
	if (S.nchunks > 1 && S.pool_used > 0) {
		sparse_image_drop_chunks(S);
		held = 0;
	}
	if (S.nchunks == 0) {
		int64_t cap = (S.pool_used > 0) ? S.pool_used + S.pool_used / 4 + ((int64_t) SP_SHARDS << 14)
		                                 : std::max<int64_t>((int64_t) 16 << 20, std::max<int64_t>(64 * (F->nnz + S.r), (int64_t) 1536 * S.r));
		// (first guess: the rows of R of the generated families hold 550-870 entries on average, and a segment reserves room for
		//  every column it touched, cancelled or not; a guess that is too small costs a second launch, one too large only address space)
		if (env_sp("SPASM_HIP_SPARSE_IMAGE_CHUNK", 0) > 0)          // (tests: pool extensions on small inputs)
			cap = env_sp("SPASM_HIP_SPARSE_IMAGE_CHUNK", 0);
		cap = std::min<int64_t>(cap, std::max<int64_t>(budget / esize, (int64_t) SP_SHARDS * 64));
		cap = (cap + SP_SHARDS - 1) / SP_SHARDS * SP_SHARDS;
		S.d_chunk[0] = dalloc<uint32_t>(cap * (esize / 4));
		S.chunk_cap[0] = cap;
		S.nchunks = 1;
	}
	SpBuildArgs b{};
	b.head = S.d_head;
	b.dep_rp = S.d_dep_rp;
	b.dep = S.d_dep;
	b.np_rp = S.d_np_rp;
	b.np = S.d_np;
	b.frag = S.d_frag;
	b.nseg = S.nseg;
	b.shard = S.d_shard;
	b.ovf_level = d_ovf;
	b.ticket = d_sync;
	b.abort_flag = d_abort;
	b.poll_limit = (long long) (1 << 21);
	unsigned long long *d_prof = nullptr;
	if (env_sp("SPASM_HIP_SPARSE_IMAGE_PROFILE", 0) != 0) {
		d_prof = dalloc<unsigned long long>(8);
		HIP_CHECK(hipMemsetAsync(d_prof, 0, 8 * sizeof(unsigned long long), stream));
	}
	b.prof = d_prof;
	b.G = sgn_setup(S.wide ? 3 : F->prime);
	b.M = to_dev(F->mont);
	// the persistent driver needs every wave of its grid resident at once
	int dev = 0, cus = 0, per_cu = 0;
	HIP_CHECK(hipGetDevice(&dev));
	HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
	if (S.wide)
		HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sp_build_kernel<true, true>, 64, 0));
	else
		HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sp_build_kernel<true, false>, 64, 0));
	per_cu = std::min(per_cu, ((int) std::min<size_t>(32, (size_t) (160 * 1024) / (S.wide ? sizeof(WaveLds32) : sizeof(WaveLds)))));
	bool persistent = env_sp("SPASM_HIP_SPARSE_IMAGE_PERSISTENT", 1) != 0 && per_cu >= 1;          // (0: the level-by-level fall-back, for the stress runs)
	HIP_CHECK(hipEventRecord(S.ev0, stream));
	auto next_chunk = [&](int chunk) -> bool {          // room for another attempt?  (allocates chunk + 1 when it is not there)
		int64_t total = 0;
		for (int k = 0; k < S.nchunks; k++)
			total += S.chunk_cap[k] * esize;
		const int64_t cap = S.chunk_cap[S.nchunks - 1] * 2;
		if (chunk + 1 < S.nchunks)
			return true;
		if (chunk + 1 >= SP_MAX_CHUNKS || total + cap * esize > budget) {
			if (verbose() >= 2)
				logmsg("[sparse image] gave up: %.2f GB of fragments do not hold R (budget %.2f GB) -- R is not sparse\n", 1e-9 * (double) total,
				       1e-9 * (double) budget);
			return false;
		}
		S.d_chunk[S.nchunks] = dalloc<uint32_t>(cap * (esize / 4));
		S.chunk_cap[S.nchunks] = cap;
		S.nchunks += 1;
		counters()[CNT_SP_CHUNK_EXTENSIONS] += 1;
		return true;
	};
	auto set_chunk = [&](int chunk, bool first) {
		for (int k = 0; k < SP_MAX_CHUNKS; k++)
			b.pools.base[k] = S.d_chunk[k < S.nchunks ? k : 0];
		b.chunk = chunk;
		b.chunk_base = S.d_chunk[chunk];
		b.shard_sub = (unsigned long long) (S.chunk_cap[chunk] / SP_SHARDS);
		hipLaunchKernelGGL(sp_reset_shards_kernel, dim3(SP_SHARDS / 64), dim3(64), 0, stream, S.d_shard,
		                   (unsigned long long) (S.chunk_cap[chunk] / SP_SHARDS), first ? 1 : 0);
	};
	int chunk = 0;
	S.launches = 0;
	bool ok = true;
	if (persistent) {
		if (S.d_segmask != nullptr) {
			hipLaunchKernelGGL(sp_init_frag_kernel, dim3(2048), dim3(256), 0, stream, S.d_frag, S.d_segmask, S.nseg, nfrag);
			b.segmask = S.d_segmask;
		} else {
			HIP_CHECK(hipMemsetAsync(S.d_frag, 0xFF, (size_t) nfrag * sizeof(uint64_t), stream));          // every fragment pending
			b.segmask = nullptr;
		}
		b.row_lo = 0;
		b.row_hi = S.r;
		const int64_t ntasks = S.r;          // (a ticket is a row)
		const int blocks = (int) std::max<int64_t>(1, std::min<int64_t>(ntasks, (int64_t) cus * per_cu));
		for (bool first = true;; first = false) {
			set_chunk(chunk, first);
			// (arenas of 16,384 entries when the chunk is large enough for every wave to strand one; tests with tiny chunks: none)
			b.arena = (S.chunk_cap[chunk] >= (int64_t) blocks * 16384 * 4 && (1) != 0) ? 16384 : 0;
			b.retry = first ? 0 : 1;
			HIP_CHECK(hipMemsetAsync(d_sync, 0, (size_t) (SP_TICKETS * SP_TICKET_STRIDE + 2) * sizeof(int), stream));
			int *d_dbg = nullptr;
			if ((0) != 0) {
				d_dbg = dalloc<int>((int64_t) blocks * 4);
				HIP_CHECK(hipMemsetAsync(d_dbg, 0, (size_t) blocks * 4 * sizeof(int), stream));
			}
			b.dbg = d_dbg;
			// A COOPERATIVE launch: the tasks of this kernel wait for each other, which only ends when every wave of the grid is
			// resident -- the runtime then refuses a grid that is not (an error at launch time instead of a stall), and does not
			// start it beside work that holds part of the chip.
			{
				void *kargs[1] = {(void *) &b};
				const hipError_t le = hipLaunchCooperativeKernel(S.wide ? reinterpret_cast<void *>(sp_build_kernel<true, true>) : reinterpret_cast<void *>(sp_build_kernel<true, false>),
				                                                 dim3(blocks), dim3(64), kargs, 0, stream);
				if (le != hipSuccess) {
					(void) hipGetLastError();
					logmsg("[sparse image] the single-launch build cannot be resident here (%s); building level by level\n", hipGetErrorString(le));
					counters()[CNT_SP_BUILD_ABORTS] += 1;
					if (d_dbg != nullptr)
						sh::big_free(d_dbg);
					persistent = false;
					chunk = 0;
					break;
				}
			}
			S.launches += 1;
			{
				// second line of defence, a watchdog: a launch that is still running after `patience` seconds is told to give up (the abort flag, written
				// from a stream of its own); what the ticket counters and the wave marks say goes to stderr.  The host sleeps
				// between its looks at the stream.
				// (patience grows with the factor: 5 s -- a hundred times the longest build measured, 0.05 s for 3 M dependencies -- plus
				//  a second per million dependencies, and at least 20 s for the 32-bit variant, which runs nine waves per CU)
				const double t_launch = wtime(), patience = std::max(S.wide ? 20.0 : 5.0, 5.0 + 1e-6 * (double) S.ndeps);
				while (hipStreamQuery(stream) == hipErrorNotReady && wtime() - t_launch < patience)
					std::this_thread::sleep_for(std::chrono::microseconds(wtime() - t_launch < 2e-3 ? 5 : 50));
				if (hipStreamQuery(stream) == hipErrorNotReady) {
					hipStream_t side;
					HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
					std::vector<int> hs((size_t) SP_TICKETS * SP_TICKET_STRIDE + 2);
					HIP_CHECK(hipMemcpyAsync(hs.data(), d_sync, hs.size() * sizeof(int), hipMemcpyDeviceToHost, side));
					HIP_CHECK(hipStreamSynchronize(side));
					fprintf(stderr, "[sparse image] the build kernel is still running after %.0f s (%lld tasks, %d waves); tickets drawn per class:", patience,
					        (long long) ntasks, blocks);
					for (int q = 0; q < SP_TICKETS; q++)
						fprintf(stderr, " %d", hs[(size_t) q * SP_TICKET_STRIDE]);
					fprintf(stderr, "; abort %d, overflow %d\n", hs[(size_t) SP_TICKETS * SP_TICKET_STRIDE], hs[(size_t) SP_TICKETS * SP_TICKET_STRIDE + 1]);
					if (d_dbg != nullptr) {
						std::vector<int> hd((size_t) blocks * 4);
						HIP_CHECK(hipMemcpyAsync(hd.data(), d_dbg, hd.size() * sizeof(int), hipMemcpyDeviceToHost, side));
						HIP_CHECK(hipStreamSynchronize(side));
						int hist[10] = {0};
						for (int w = 0; w < blocks; w++)
							hist[std::max(0, std::min(9, hd[(size_t) w * 4]))] += 1;
						fprintf(stderr, "[sparse image/debug] waves by stage: not started %d, ticket %d, polling %d, adding %d, emit %d, task done %d, exited %d\n",
						        hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[9]);
						int shown = 0;
						for (int w = 0; w < blocks && shown < 24; w++)
							if (hd[(size_t) w * 4] != 9 && hd[(size_t) w * 4] != 0) {
								fprintf(stderr, "[sparse image/debug]   wave %d: stage %d, task %d, detail %d\n", w, hd[(size_t) w * 4], hd[(size_t) w * 4 + 1], hd[(size_t) w * 4 + 2]);
								shown += 1;
							}
					}
					// a sample of the fragment words: how many are still pending?
					{
						const int64_t ns = std::min<int64_t>(nfrag, 1 << 20);
						std::vector<uint64_t> hf((size_t) ns);
						HIP_CHECK(hipMemcpyAsync(hf.data(), S.d_frag + (nfrag - ns), (size_t) ns * sizeof(uint64_t), hipMemcpyDeviceToHost, side));
						HIP_CHECK(hipStreamSynchronize(side));
						int64_t pend = 0, first_pending = -1;
						for (int64_t t = ns - 1; t >= 0; t--)
							if (hf[(size_t) t] == FRAG_PENDING) {
								pend += 1;
								if (first_pending < 0)
									first_pending = (nfrag - 1) - ((nfrag - ns) + t);
							}
						fprintf(stderr, "[sparse image] of the first %lld tasks, %lld are pending; the first pending one is task %lld\n", (long long) ns, (long long) pend,
						        (long long) first_pending);
					}
					const int one = 1;
					HIP_CHECK(hipMemcpyAsync(d_abort, &one, sizeof(int), hipMemcpyHostToDevice, side));
					HIP_CHECK(hipStreamSynchronize(side));
					(void) hipStreamDestroy(side);
				}
			}
			int flags[2] = {0, 0};          // abort, overflow
			HIP_CHECK(hipMemcpyAsync(flags, d_abort, sizeof(flags), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
			if (d_dbg != nullptr)
				sh::big_free(d_dbg);
			if (verbose() >= 3)
				logmsg("[sparse image] launch %d on chunk %d (%lld entries): abort %d, overflow %d\n", S.launches, chunk, (long long) S.chunk_cap[chunk], flags[0], flags[1]);
			if (flags[0] != 0) {
				// a wave waited for a dependency for seconds: the grid was not resident (something else holds the chip?).  The
				// level-by-level driver needs no such thing.
				logmsg("[sparse image] the single-launch build gave up waiting; building level by level\n");
				counters()[CNT_SP_BUILD_ABORTS] += 1;
				persistent = false;
				chunk = 0;
				break;
			}
			if (flags[1] == 0)
				break;
			if (!next_chunk(chunk)) {
				ok = false;
				break;
			}
			chunk += 1;
			hipLaunchKernelGGL(sp_reset_failed_kernel, dim3(1024), dim3(256), 0, stream, S.d_frag, nfrag);
		}
	}
	if (!persistent) {
		int from_level = S.nlevels - 1;
		for (bool first = true;; first = false) {
			set_chunk(chunk, first);
			HIP_CHECK(hipMemsetAsync(d_ovf, 0xFF, sizeof(int), stream));
			for (int l = from_level; l >= 0; l--) {
				b.row_lo = S.lvl_lo[l];
				b.row_hi = S.lvl_lo[l + 1];
				b.level = l;
				const int64_t ntasks = (int64_t) (b.row_hi - b.row_lo) * S.nseg;
				if (ntasks <= 0)
					continue;
				if (ntasks > 0x7FFFFFFFll)
					die("sparse_image_build: level %d has %lld (row, segment) pairs", l, (long long) ntasks);
				if (S.wide)
					hipLaunchKernelGGL((sp_build_kernel<false, true>), dim3((unsigned) ntasks), dim3(64), 0, stream, b);
				else
					hipLaunchKernelGGL((sp_build_kernel<false, false>), dim3((unsigned) ntasks), dim3(64), 0, stream, b);
				S.launches += 1;
			}
			HIP_CHECK(hipGetLastError());
			int ovf = -1;
			HIP_CHECK(hipMemcpyAsync(&ovf, d_ovf, sizeof(int), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
			if (ovf < 0)
				break;
			if (!next_chunk(chunk)) {          // out of room in level `ovf`
				ok = false;
				break;
			}
			chunk += 1;
			from_level = ovf;
		}
	}
	HIP_CHECK(hipEventRecord(S.ev1, stream));
	if (!ok) {
		// R is not sparse: the other paths take the batch -- on this very call, and they size themselves by the memory that is
		// free: everything the attempt holds goes back first (the pool chunks: up to a third of the HBM; the fragment words).
		// `failed` is for good on this factor, so nothing here is needed again.
		S.failed = true;
		HIP_CHECK(hipStreamSynchronize(stream));
		sparse_image_drop_chunks(S);
		for (void **ptr : {(void **) &S.d_frag, (void **) &S.d_shard, (void **) &S.d_col, (void **) &S.d_dep_rp, (void **) &S.d_dep, (void **) &S.d_np_rp, (void **) &S.d_np, (void **) &S.d_segmask, (void **) &S.d_rowmask, (void **) &S.d_head}) {
			sh::big_free(*ptr);          // (the tables of the plan too: the dependencies and the non-pivotal entries of every row)
			*ptr = nullptr;
		}
		if (d_prof != nullptr)
			sh::big_free(d_prof);
		return false;
	}
	std::vector<unsigned long long> h((size_t) SP_SHARDS * SHARD_STRIDE);
	HIP_CHECK(hipMemcpyAsync(h.data(), S.d_shard, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	S.ops_build = 0;
	S.nnz = 0;
	int64_t reserved = 0;
	for (int s = 0; s < SP_SHARDS; s++) {
		S.ops_build += (int64_t) h[(size_t) s * SHARD_STRIDE + 2];
		S.nnz += (int64_t) h[(size_t) s * SHARD_STRIDE + 3];
		reserved += (int64_t) h[(size_t) s * SHARD_STRIDE + 5];
	}
	S.pool_used = reserved;          // (what a rebuild needs in one chunk: a segment reserves room for every column it touched)
	if (chunk > 0 && !persistent) {
		// levels were redone: their fragments were counted twice
		HIP_CHECK(hipMemsetAsync(S.d_shard, 0, sizeof(unsigned long long), stream));
		hipLaunchKernelGGL(sp_sum_frag_kernel, dim3(1024), dim3(256), 0, stream, S.d_frag, nfrag, S.d_shard);
		unsigned long long exact = 0;
		HIP_CHECK(hipMemcpyAsync(&exact, S.d_shard, sizeof(exact), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		S.nnz = (int64_t) exact;
	}
	if (d_prof != nullptr) {
		unsigned long long hp[8];
		HIP_CHECK(hipMemcpyAsync(hp, d_prof, sizeof(hp), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		static const char *const name[7] = {"ticket", "metadata", "polling", "adding", "reservation", "emit", "publication"};
		double tot = 0;
		for (int q = 0; q < 7; q++)
			tot += (double) hp[q];
		fprintf(stderr, "[sparse image/profile] build, wave cycles by stage:");
		for (int q = 0; q < 7; q++)
			fprintf(stderr, " %s %.1f %%;", name[q], 100.0 * (double) hp[q] / std::max(1.0, tot));
		fprintf(stderr, " %.3g cycles in all\n", tot);
		sh::big_free(d_prof);
	}
	if (S.d_rowmask == nullptr)
		S.d_rowmask = dalloc<uint64_t>((int64_t) S.r * ((S.nseg + 63) / 64));
	hipLaunchKernelGGL(sp_rowmask_kernel, dim3(1024), dim3(256), 0, stream, S.d_frag, S.r, S.nseg, S.d_rowmask);
	S.valid = true;
	S.failed = false;
	S.builds += 1;
	if (verbose() >= 2) {
		double gb = 0;
		for (int k = 0; k < S.nchunks; k++)
			gb += 4e-9 * (double) S.chunk_cap[k];
		logmsg("[sparse image] R: %d rows x %d columns in %d segments, %lld entries (%.3f %% of the dense form, %.1f per row), %lld multiply-adds, %d "
		       "launch(es) (%s), %d pool chunk(s) of %.2f GB in all\n",
		       S.r, S.Sm, S.nseg, (long long) S.nnz, 100.0 * (double) S.nnz / std::max(1.0, (double) S.r * (double) S.Sm),
		       (double) S.nnz / std::max(1, S.r), (long long) S.ops_build, S.launches, persistent ? "tasks handed out in order" : "level by level", S.nchunks, gb);
	}
	return true;
}

// 64-bit words of the tables of a Schur call: T (one word per (row, segment)) and D (32 bits each) behind it
int64_t sparse_image_table_words(int64_t nrows, int nseg) { return nrows * nseg + (nrows * nseg + 1) / 2 + 8; }

// S rows from the sparse image, as sparse rows in W's final arrays.
//   fpool / fcap: room for the fragments of S (4-byte entries; the 32-bit variant: their columns, and fpool_v their values);
//   T: nrows * nseg words; block_sum: (nrows + 1023) / 1024 words
void launch_sparse_image_apply(const SchurArgs &a, const spasm_hip_dfact *F, uint32_t *fpool, uint32_t *fpool_v, int64_t fcap,
                               uint64_t *T, unsigned long long *block_sum, int64_t *Sp, int *Sj, int *Sx, int64_t cap, hipStream_t stream,
                               hipEvent_t ev_gather)
{
	// (T: nrows * nseg 64-bit words, then as many 32-bit words of D: sparse_image_table_words)
	const SpImage &S = F->sp;
	if (!S.valid)
		die("launch_sparse_image_apply: the sparse image has not been built");
	int dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	hipDeviceProp_t prop;
	HIP_CHECK(hipGetDeviceProperties(&prop, dev));
	SpApplyArgs d{};
	d.a = a;
	d.col = S.d_col;
	d.r = S.r;
	d.nseg = S.nseg;
	d.frag = S.d_frag;
	d.rowmask = S.d_rowmask;
	for (int k = 0; k < SP_MAX_CHUNKS; k++)
		d.pools.base[k] = S.d_chunk[k < S.nchunks ? k : 0];
	d.G = sgn_setup(S.wide ? 3 : F->prime);
	d.fpool = fpool;
	d.fpool_v = fpool_v;
	d.fcap = fcap;
	d.T = T;
	d.D = reinterpret_cast<uint32_t *>(T + (size_t) a.nrows * S.nseg);
	d.block_sum = block_sum;
	const int64_t ntasks = (int64_t) a.nrows;          // (a wave takes a row through all its segments)
	if (ntasks <= 0)
		return;
	// as many waves per CU as its LDS holds (19.5 KB each: eight), every wave a workgroup of its own
	const int per_cu = std::max(1, std::min(32, ((int) std::min<size_t>(32, (size_t) (160 * 1024) / (S.wide ? sizeof(WaveLds32) : sizeof(WaveLds))))));
	const int blocks = (int) std::min<int64_t>(ntasks, (int64_t) prop.multiProcessorCount * per_cu);
	// a wave reserves the room of its fragments 8,192 entries at a time (what the ~4,000 waves strand at the end must stay small
	// against a pool sized from a density estimate: 32,768 apiece were 126 M entries, and a retry of the whole call); a pool
	// too small even for that: fragment by fragment
	d.arena = (fcap >= (int64_t) blocks * 8192 * 8) ? 8192 : 0;
	// (block_sum: (nrows + 1023) / 1024 words rounded up to 16, then 256 words of ticket counters)
	const int nblocks = (a.nrows + 1023) / 1024;
	d.ticket = reinterpret_cast<int *>(block_sum + (size_t) (nblocks + 15) / 16 * 16);
	HIP_CHECK(hipMemsetAsync(d.ticket, 0, (size_t) SP_TICKETS * SP_TICKET_STRIDE * sizeof(int), stream));
	HIP_CHECK(hipMemsetAsync(block_sum, 0, (size_t) nblocks * sizeof(unsigned long long), stream));
	HIP_CHECK(hipMemsetAsync(Sp, 0, sizeof(int64_t), stream));
	unsigned long long *d_prof = nullptr;
	if (env_sp("SPASM_HIP_SPARSE_IMAGE_PROFILE", 0) != 0) {
		d_prof = dalloc<unsigned long long>(8);
		HIP_CHECK(hipMemsetAsync(d_prof, 0, 8 * sizeof(unsigned long long), stream));
	}
	d.prof = d_prof;
	if (S.wide)
		hipLaunchKernelGGL(sp_apply_kernel<true>, dim3(blocks), dim3(64), 0, stream, d);
	else
		hipLaunchKernelGGL(sp_apply_kernel<false>, dim3(blocks), dim3(64), 0, stream, d);
	HIP_CHECK(hipGetLastError());
	if (d_prof != nullptr) {
		unsigned long long hp[8];
		HIP_CHECK(hipMemcpyAsync(hp, d_prof, sizeof(hp), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		static const char *const name[6] = {"row of A", "fragment words", "adding", "reservation", "emit", "words out"};
		double tot = 0;
		for (int q = 0; q < 6; q++)
			tot += (double) hp[q];
		fprintf(stderr, "[sparse image/profile] rows of S (%d waves), wave cycles by stage:", blocks);
		for (int q = 0; q < 6; q++)
			fprintf(stderr, " %s %.1f %%;", name[q], 100.0 * (double) hp[q] / std::max(1.0, tot));
		fprintf(stderr, " %.3g cycles in all\n", tot);
		sh::big_free(d_prof);
	}
	if (ev_gather != nullptr)
		HIP_CHECK(hipEventRecord(ev_gather, stream));
	launch_scan_lengths(a.row_len, a.nrows, block_sum, Sp, cap, a.ctr, stream);
	SpGatherArgs e{T, d.D, fpool, fpool_v, (uint32_t) F->prime, a.nrows, S.nseg, S.Sm, Sp, Sj, Sx, cap, a.q};
	const int gblocks = (a.nrows + SP_GATHER_ROWS - 1) / SP_GATHER_ROWS;
	if (S.wide)
		hipLaunchKernelGGL(sp_gather_kernel<true>, dim3(gblocks), dim3(256), 0, stream, e);
	else
		hipLaunchKernelGGL(sp_gather_kernel<false>, dim3(gblocks), dim3(256), 0, stream, e);
	HIP_CHECK(hipGetLastError());
}

// entries of R, occupied 64-column tiles, non-empty fragments, (row, segment) pairs -- of the image as it stands
void sparse_image_census(const spasm_hip_dfact *F, int64_t *out, hipStream_t stream)
{
	const SpImage &S = F->sp;
	out[0] = out[1] = out[2] = out[3] = 0;
	if (!S.valid)
		return;
	SpPools pools;
	for (int k = 0; k < SP_MAX_CHUNKS; k++)
		pools.base[k] = S.d_chunk[k < S.nchunks ? k : 0];
	unsigned long long *d = dalloc<unsigned long long>(4);
	HIP_CHECK(hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), stream));
	if (S.wide)
		hipLaunchKernelGGL(sp_census_kernel<true>, dim3(2048), dim3(256), 0, stream, S.d_frag, (int64_t) S.r * S.nseg, pools, d);
	else
		hipLaunchKernelGGL(sp_census_kernel<false>, dim3(2048), dim3(256), 0, stream, S.d_frag, (int64_t) S.r * S.nseg, pools, d);
	unsigned long long h[4] = {0, 0, 0, 0};
	HIP_CHECK(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	sh::big_free(d);
	out[0] = (int64_t) h[0];
	out[1] = (int64_t) h[1];
	out[2] = (int64_t) h[2];
	out[3] = (int64_t) S.r * S.nseg;
}

}  // namespace sh
