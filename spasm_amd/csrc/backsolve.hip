// Back-substituted factor image and the Schur complement computed from it.
//
// The reference reduces every non-pivotal row a of A by a sparse triangular solve against U
// (spasm_schur.c:86-171 -> spasm_triangular.c:109-146): x = a U_pp^-1, then s = a_n - x U_pn.  The
// same product can be bracketed the other way round:
//
//       S = A_n - A_p (U_pp^-1 U_pn) = A_n - A_p R,
//
// where row c of R is the non-pivotal part of the fully reduced pivot row c (the rows spasm_rref would
// produce, spasm_rref.c:25).  R is dense, r x Sm (Sm = number of non-pivotal columns).  When Sm is small
// -- a Schur complement that is going to be dense anyway -- building R costs nnz(U') * Sm multiply-adds,
// streaming and free of atomics, and every reduced row is then a combination of the few rows of R its
// pivotal entries select.  mk13.b5: 130183 x 4952 (1.3 GB in 16-bit entries) against 3.97e9 (row, pivot) eliminations.
// Arithmetic mod p is exact, so the result is the same matrix, bit for bit.
//
// R[c] = U_n[c] - sum_{t pivotal in U'[c]} u_ct R[t], and t always lies in a later elimination level
// than c.  The columns of R are independent: one workgroup owns a slab of 32 or 64 columns and walks the
// rows from the last level to the first, with no communication between workgroups at all.  Inside a
// workgroup the chain of levels runs in LDS: rows are cut into chunks of <= RING consecutive rows;
//   phase A: every row of the chunk gathers what it needs from outside the chunk (rows of R that are
//            final, in HBM) into an LDS ring,
//   phase B: level by level, rows pick up their dependencies inside the chunk from the ring -- every WAVE
//            on its own columns of every row, so that the levels need no barrier,
//   phase C: the ring is written back to R.
// The rows of S then come from bs_apply_*: one wave per row, a few rows of R each.
#include <algorithm>
#include <cinttypes>
#include <type_traits>
#include <thread>
#include <vector>

#include "device_types.h"
#include "field_dev.h"
#include "sgn_dev.h"

namespace sh {

int usable_cpus();          // host_pivots.cpp

namespace {

constexpr int BS_RING = 768;       // rows per chunk (the LDS ring holds RING rows of one slab)
constexpr int BS_NEARCAP = 1024;   // dependencies inside a chunk that do not fit the pass table (rows with more than two)
constexpr int BS_PASSCAP = 80;     // phase-B passes of a chunk (32 rows of one level each)
constexpr int BS_PASSROWS = 32;
// an empty slot of the pass table points at a spare row of the ring (index BS_RING) with coefficient 0: the signed kernels
// run every lane through the same reads, multiply-adds and write, without a branch
#define BS_EMPTY_ENTRY uint4{(uint32_t) BS_RING, (uint32_t) BS_RING | ((uint32_t) BS_RING << 16), 0u, 0u}
constexpr uint32_t BS_NONE = 0xFFFFFFFFu;

int env_bs(const char *name, int dflt)
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

struct BsArgs {
	void *R;                      // r x ldR elements of T (uint16_t when p < 2^16, else uint32_t)
	int64_t ldR;
	int nchunks;
	const BsChunk *chunk;
	const int *chunk_extra;       // per chunk: 1 when some row has more than two dependencies outside the chunk
	const uint4 *ptab;
	const uint2 *near;
	const uint4 *far_head;
	const uint64_t *far_rp;
	const uint2 *far;
	const uint64_t *np_rp;
	const uint2 *np;
	const int *np_row;            // compact row of every entry of np
	int plain;                    // coefficients (y) are plain residues (p < 2^16), not Montgomery form
	int sparse_init;              // 1: the kernel scatters U_n itself (few entries); 0: R was pre-filled by bs_init_kernel
	int r;
	int dbg;                      // timing experiments only (SPASM_HIP_BS_DEBUG): bit 0/1/2 = skip phase A/B/C (wrong results)
	unsigned long long *prof;     // SPASM_HIP_BS_PROFILE=1: 8 counters, shader-clock cycles of workgroup 0 per stage of a chunk, summed over the chunks
	int sgn;                      // signed 16-bit entries (the SGN kernels); coefficients are negated balanced residues
	MontDev F;
	SgnDev G;
};

// ---- storage of R ---------------------------------------------------------------------------------
// A lane always moves 32-bit WORDS.  PACKED = false: a word is one entry of R (any odd p < 2^32).  PACKED = true
// (p < 2^16; 42013, the reference's default, qualifies): a word holds two consecutive 16-bit entries -- half the
// traffic, half the LDS, the same number of memory instructions.
template <bool PACKED> struct Word {
	static constexpr int CPL = PACKED ? 2 : 1;            // columns per lane
	using Elem = typename std::conditional<PACKED, uint16_t, uint32_t>::type;
};

// x - v * y, component-wise.  Coefficients y are in Montgomery form (value * 2^32 mod p) for p >= 2^16; for p < 2^16
// (PLAIN) they are the plain residues: v * y < 2^32 is one full-rate 24-bit multiply and a Barrett reduction (one
// quarter-rate multiply instead of the four of a Montgomery product), bm = floor(2^32 / p).
template <bool PACKED, bool PLAIN> __device__ __forceinline__ uint32_t w_submul(uint32_t x, uint32_t v, uint32_t y, const MontDev &F, uint32_t bm)
{
	auto small = [&](uint32_t xe, uint32_t ve) -> uint32_t {
		const uint32_t t = __umul24(ve, y);                  // ve, y < p < 2^16
		const uint32_t q = __umulhi(t, bm);                   // floor(t / p) - 2 <= q <= floor(t / p); q < p
		uint32_t rem = t - __umul24(q, F.p);
		rem = (rem >= F.p) ? rem - F.p : rem;
		rem = (rem >= F.p) ? rem - F.p : rem;
		return (xe >= rem) ? xe - rem : xe + F.p - rem;
	};
	if constexpr (PACKED) {
		static_assert(PLAIN, "packed entries only exist for p < 2^16");
		return small(x & 0xFFFFu, v & 0xFFFFu) | (small(x >> 16, v >> 16) << 16);
	} else if constexpr (PLAIN) {
		return small(x, v);
	} else {
		return submod(x, montmul(v, y, F), F);
	}
}

// R <- U_n (values out of Montgomery form); R was zeroed before.  Only for factors whose rows hold many non-pivotal
// entries: otherwise the backsolve kernel scatters them itself and R is never pre-filled.
template <typename T> __global__ __launch_bounds__(256) void bs_init_kernel(BsArgs b)          // (b.plain: entries hold plain residues)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= b.r)
		return;
	T *row = static_cast<T *>(b.R) + (int64_t) c * b.ldR;
	for (uint64_t e = b.np_rp[c]; e < b.np_rp[c + 1]; e++) {
		const uint2 en = b.np[e];
		row[en.x] = b.sgn ? (T) (uint16_t) (int16_t) sgn_from_residue(en.y, b.G) : (T) (b.plain ? en.y : montmul(en.y, 1u, b.F));
	}
}

// A fixed number of values that must stay in REGISTERS across a long stretch of code.  A local array would do in
// principle, but an array of more than 16 dwords is not turned into registers by the compiler (it becomes scratch memory,
// i.e. global memory behind the cache: measured on the metadata of backsolve_kernel, whose 20-dword table part went
// through scratch_store / scratch_load and cost two memory round trips per chunk); members of a struct, reached with
// compile-time indices, always are.
template <typename T, int N> struct RegFile {
	T head;
	RegFile<T, N - 1> tail;
	template <int I> __device__ __forceinline__ T &at()
	{
		if constexpr (I == 0)
			return head;
		else
			return tail.template at<I - 1>();
	}
};
template <typename T> struct RegFile<T, 0> {
};

template <int I, int N, typename Fn> __device__ __forceinline__ void bs_static_for(Fn &&f)
{
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		bs_static_for<I + 1, N>(f);
	}
}

// LPR lanes (words) per row of a slab: a row of a slab is LPR * 4 bytes (128 B with LPR = 32: whole cache lines, half
// as many requests as 64-byte segments -- the kernel is bound by the rate of such requests).
// NW waves per workgroup; a wave instruction covers 64 / LPR rows.
// RING / PASSROWS / PASSCAP: rows per chunk, rows per phase-B pass, passes per chunk -- the plan (backsolve_plan) is built
// for the values of the kernel that will run it (BsImage::ring, passrows, passcap).
template <bool PACKED, int LPR, int NW, int RING = BS_RING, int PASSROWS = BS_PASSROWS, int PASSCAP = BS_PASSCAP> struct BsGeom {
	static constexpr int RING_ = RING, PASSROWS_ = PASSROWS, PASSCAP_ = PASSCAP;
	static constexpr int THREADS = 64 * NW;
	static constexpr int RS = 64 / LPR;
	static constexpr int CW = LPR * Word<PACKED>::CPL;     // columns per slab
	static constexpr int ROWS_PER_ITER = RS * NW;
	static constexpr int ITERS = RING / ROWS_PER_ITER;
	// rows in flight per lane in phase A.  Eight waves (two per SIMD: 256 registers each) take a whole chunk in ONE pass --
	// a pass is a round trip to memory, and a chunk has little else to hide it behind
	static constexpr int UNR = (NW >= 8 && ITERS <= 24) ? ITERS : (ITERS % 12 == 0) ? 12 : 8;
	static constexpr int LANES_USED = RS * LPR;            // (LPR need not divide 64: the lanes beyond RS whole rows idle in phases A and C)
	static_assert(RING % ROWS_PER_ITER == 0 && ITERS % UNR == 0, "phase A is unrolled in passes of UNR rows");
	static constexpr int N_NEAR = (BS_NEARCAP + THREADS - 1) / THREADS;      // metadata words a thread carries for the next chunk
	static constexpr int N_ROW = (RING + THREADS - 1) / THREADS;
	static constexpr int N_PTAB = (PASSCAP * PASSROWS + THREADS - 1) / THREADS;
	static constexpr size_t FH_BYTES = (size_t) RING * sizeof(uint4);
	static constexpr size_t PTAB_BYTES = (size_t) PASSCAP * PASSROWS * sizeof(uint4);
	static constexpr int RSTR = LPR + 1;                    // row stride of the ring in words: odd, so that a wave instruction over
	                                                        // consecutive rows AND one over consecutive words both spread over the banks
	static constexpr size_t LDS_BYTES = FH_BYTES + PTAB_BYTES + (size_t) BS_NEARCAP * sizeof(uint2) + (size_t) (RING + 1) * RSTR * 4;          // (+ the spare row)
};

template <bool PACKED, bool PLAIN, int LPR, int NW, bool SGN = false, int RING = BS_RING, int PASSROWS = BS_PASSROWS, int PASSCAP = BS_PASSCAP, int WGS_PER_CU = 1>
__global__ __launch_bounds__(64 * NW, WGS_PER_CU) void backsolve_kernel(BsArgs b)
{
	// an empty slot of the pass table points at the spare row of the ring (index RING) with coefficient 0
	const uint4 EMPTY_ENTRY = uint4{(uint32_t) RING, (uint32_t) RING | ((uint32_t) RING << 16), 0u, 0u};
	static_assert(!SGN || (PACKED && PLAIN), "signed entries are packed 16-bit entries");
	const uint32_t bm = PLAIN ? (uint32_t) (0x100000000ull / b.F.p) : 0u;
	const SgnDev G = b.G;
	using Geo = BsGeom<PACKED, LPR, NW, RING, PASSROWS, PASSCAP>;
	using Elem = typename Word<PACKED>::Elem;
	extern __shared__ __attribute__((aligned(16))) unsigned char bs_lds[];
	uint4 *fh = reinterpret_cast<uint4 *>(bs_lds);                          // first two outside dependencies of every row
	uint4 *ptab = reinterpret_cast<uint4 *>(bs_lds + Geo::FH_BYTES);       // phase-B passes
	uint2 *near = reinterpret_cast<uint2 *>(bs_lds + Geo::FH_BYTES + Geo::PTAB_BYTES);
	uint32_t *ring = reinterpret_cast<uint32_t *>(near + BS_NEARCAP);
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	// workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains the vector-memory counter,
	// i.e. the metadata of the NEXT chunk that was asked for at the top of this one: the first barrier of every chunk then
	// costs a round trip to memory (170 chunks x ~1 us on mk13.b5).  Between the phases only the ring (LDS) is handed
	// from wave to wave; the one place where global stores must be visible to other waves (after phase C) keeps
	// __syncthreads().
	auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
	const int rs = lane / LPR, wl = lane % LPR;           // row slot of the lane, its word inside the row (phases A and C)
	constexpr int RSTR = Geo::RSTR;                        // words between two rows of the ring
	// phase B splits the COLUMNS among the waves: a wave owns WPW words of every row, a wave instruction covers RSB rows
	constexpr int WPW = LPR / NW, RSB = 64 / WPW;
	static_assert(WPW >= 1 && LPR % NW == 0 && 64 % WPW == 0, "every wave takes the same number of words of a row through phase B");
	static_assert(PASSROWS % RSB == 0, "a pass is a whole number of wave instructions");
	const int rsb = lane / WPW, wlb = wave * WPW + lane % WPW;
	const MontDev F = b.F;
	const int64_t ldw = b.ldR / Word<PACKED>::CPL;       // row stride of R in words
	uint32_t *Rs = static_cast<uint32_t *>(b.R) + (int64_t) blockIdx.x * LPR + wl;
	// a row of R starts at row * ldw words; ldw is a multiple of 256 (rows are padded to 512 columns), and row * (ldw / 256)
	// fits 32 bits for any R below 4 TB: one 32-bit multiply and a shift instead of a 64-bit multiply per load
	const uint32_t ldw256 = (uint32_t) (ldw >> 8);
	auto row_at = [&](uint32_t row) -> uint32_t * { return Rs + ((uint64_t) (row * ldw256) << 8); };
	const int slot0 = wave * Geo::RS + rs;          // this lane's row slot within an iteration
	// lanes that hold a word of a row in phases A and C: all of them when LPR divides 64; and the word must exist (the last
	// slab of a row may reach beyond the padded row when LPR does not divide its length)
	const bool lane_ok = lane < Geo::LANES_USED && (int64_t) blockIdx.x * LPR + wl < ldw;
	const int col_lo = blockIdx.x * Geo::CW;        // columns [col_lo, col_lo + CW) belong to this workgroup

	// metadata of a chunk (the same for every slab: served by the L2) travels through registers: the loads for chunk
	// k + 1 are issued when chunk k starts and land in LDS when it is done
	RegFile<uint2, Geo::N_NEAR> m_near;
	RegFile<uint4, Geo::N_ROW> m_fh;
	RegFile<uint4, Geo::N_PTAB> m_ptab;
	// the first THREADS non-pivotal entries of the next chunk (mk13.b5: 61 per chunk), one per thread; the rest -- if any --
	// is read when the chunk starts
	uint2 m_np = uint2{0u, 0u};
	int m_np_row = 0;
	auto load_meta = [&](const BsChunk &c) {
		{
			const int cnt = c.npn & 0x7FFFFFFF;
			const uint2 v = b.np[(tid < cnt) ? c.np0 + tid : 0];
			const int vr = b.np_row[(tid < cnt) ? c.np0 + tid : 0];
			m_np = v;
			m_np_row = vr;
		}
		bs_static_for<0, Geo::N_NEAR>([&](auto qq) {
			constexpr int q = decltype(qq)::value;
			const int t = tid + q * Geo::THREADS;
			// (the load is unconditional, from a valid address, and the VALUE is selected: `cond ? *p : constant` makes the
			//  compiler select between p and the address of the constant, which then has to live in scratch memory, and turns
			//  the load into a FLAT one)
			const uint2 v = b.near[(t < c.nnear) ? c.near0 + t : 0];
			m_near.template at<q>() = (t < c.nnear) ? v : uint2{0u, 0u};
		});
		bs_static_for<0, Geo::N_ROW>([&](auto qq) {
			constexpr int q = decltype(qq)::value;
			const int t = tid + q * Geo::THREADS;
			const uint4 v = b.far_head[(t < c.hi - c.lo) ? c.lo + t : 0];
			m_fh.template at<q>() = (t < c.hi - c.lo) ? v : uint4{BS_NONE, 0u, BS_NONE, 0u};
		});
		bs_static_for<0, Geo::N_PTAB>([&](auto qq) {
			constexpr int q = decltype(qq)::value;
			const int t = tid + q * Geo::THREADS;
			const uint4 v = b.ptab[(t < c.npass * PASSROWS) ? (int64_t) c.pass0 * PASSROWS + t : 0];
			m_ptab.template at<q>() = (t < c.npass * PASSROWS) ? v : uint4{(uint32_t) RING, (uint32_t) RING | ((uint32_t) RING << 16), 0u, 0u};
		});
	};
	auto store_meta = [&]() {
		bs_static_for<0, Geo::N_NEAR>([&](auto qq) {
			constexpr int q = decltype(qq)::value;
			if (tid + q * Geo::THREADS < BS_NEARCAP)
				near[tid + q * Geo::THREADS] = m_near.template at<q>();
		});
		bs_static_for<0, Geo::N_ROW>([&](auto qq) {
			constexpr int q = decltype(qq)::value;
			if (tid + q * Geo::THREADS < RING)
				fh[tid + q * Geo::THREADS] = m_fh.template at<q>();
		});
		bs_static_for<0, Geo::N_PTAB>([&](auto qq) {
			constexpr int q = decltype(qq)::value;
			if (tid + q * Geo::THREADS < PASSCAP * PASSROWS)
				ptab[tid + q * Geo::THREADS] = m_ptab.template at<q>();
		});
	};
	// chunk descriptors travel one chunk ahead too: the descriptor of chunk k + 1 is in registers when chunk k starts, so
	// that the loads of its metadata (whose addresses it holds) can be issued at once instead of after a round trip
	// (read with VECTOR loads -- every lane the same 32 bytes: a scalar load shares its counter with the LDS, and the first
	//  LDS access after it would wait for the round trip)
	static_assert(sizeof(BsChunk) == 32, "a chunk descriptor is two 16-byte loads");
	int vzero;
	asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));          // (a zero the compiler cannot see through: keeps the address in a VGPR)
	uint4 raw_a = uint4{0u, 0u, 0u, 0u}, raw_b = uint4{0u, 0u, 0u, 0u};
	auto fetch_chunk = [&](int k) {
		const uint4 *src = reinterpret_cast<const uint4 *>(b.chunk + k) + vzero;
		raw_a = src[0];
		raw_b = src[1];
	};
	auto fetched_chunk = [&]() {
		BsChunk c;
		c.lo = __builtin_amdgcn_readfirstlane((int) raw_a.x);
		c.hi = __builtin_amdgcn_readfirstlane((int) raw_a.y);
		c.pass0 = __builtin_amdgcn_readfirstlane((int) raw_a.z);
		c.npass = __builtin_amdgcn_readfirstlane((int) raw_a.w);
		c.near0 = __builtin_amdgcn_readfirstlane((int) raw_b.x);
		c.nnear = __builtin_amdgcn_readfirstlane((int) raw_b.y);
		c.np0 = __builtin_amdgcn_readfirstlane((int) raw_b.z);
		c.npn = __builtin_amdgcn_readfirstlane((int) raw_b.w);
		return c;
	};
	BsChunk ch_next{};
	if (b.nchunks > 0) {
		fetch_chunk(0);
		ch_next = fetched_chunk();
		load_meta(ch_next);
		store_meta();
	}
	uint2 c_np = m_np;            // (chunk 0's own first entries)
	int c_np_row = m_np_row;
	if (b.nchunks > 1)
		fetch_chunk(1);
	__syncthreads();

	// stage timer (SPASM_HIP_BS_PROFILE): thread 0 of workgroup 0 reads the shader clock at every stage boundary
	unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long t_last = 0;
	const bool profiling = b.prof != nullptr && blockIdx.x == 0 && tid == 0;
	auto tick = [&](int q) {
		if (profiling) {
			const unsigned long long now = __builtin_readcyclecounter();
			pt[q] += now - t_last;
			t_last = now;
		}
	};
	if (profiling)
		t_last = __builtin_readcyclecounter();
	for (int k = 0; k < b.nchunks; k++) {
		const BsChunk ch = ch_next;
		const int nrows = ch.hi - ch.lo;
		if (k > 0) {
			c_np = m_np;
			c_np_row = m_np_row;
		}
		if (k + 1 < b.nchunks) {
			ch_next = fetched_chunk();
			load_meta(ch_next);
		}
		if (k + 2 < b.nchunks)
			fetch_chunk(k + 2);
		tick(0);          // descriptor + issue of the next chunk's metadata

		if (b.sparse_init) {
			// the rows start as U_n: few entries, scattered from the list (R itself is never read for them)
			for (int t = tid; t < nrows * RSTR; t += Geo::THREADS)
				ring[t] = 0;
			lds_barrier();
			const int np_count = ch.npn & 0x7FFFFFFF;
			auto put = [&](const uint2 en, int row) {
				const int cc = (int) en.x - col_lo;
				if (cc >= 0 && cc < Geo::CW)
					reinterpret_cast<Elem *>(ring)[(row - ch.lo) * (RSTR * Word<PACKED>::CPL) + cc] =
						SGN ? (Elem) (uint16_t) (int16_t) sgn_from_residue(en.y, G) : (Elem) (PLAIN ? en.y : montmul(en.y, 1u, F));
			};
			if (tid < np_count)
				put(c_np, c_np_row);
			for (int e = Geo::THREADS + tid; e < np_count; e += Geo::THREADS)
				put(b.np[ch.np0 + e], b.np_row[ch.np0 + e]);
			lds_barrier();
		}
		tick(1);          // start of the rows (zero + scatter of the non-pivotal entries)

		// ---- phase A: own row + dependencies outside the chunk, up to 3 * UNR loads in flight per lane ----
		for (int pass = 0; pass < Geo::ITERS / Geo::UNR; pass++) {
			if (pass * Geo::UNR * Geo::ROWS_PER_ITER >= nrows || (b.dbg & 1))
				break;
			uint32_t acc[Geo::UNR], v0[Geo::UNR], v1[Geo::UNR];
#pragma unroll
			for (int u = 0; u < Geo::UNR; u++) {
				const int s = (pass * Geo::UNR + u) * Geo::ROWS_PER_ITER + slot0;
				const bool ok = s < nrows && lane_ok;
				const uint4 h = fh[ok ? s : 0];
				const int c = ch.lo + (ok ? s : 0);
				acc[u] = (b.sparse_init || !ok) ? 0u : *row_at((uint32_t) c);
				if constexpr (LPR <= 16 && 64 % LPR == 0) {
					// no branch around the loads: an absent dependency reads the chunk's own first row -- a valid address -- and
					// is multiplied by the coefficient 0 of its slot, or skipped, below (64-byte rows: 2.84 -> 2.75 ms on mk13.b5,
					// 9.2 -> 6.1 ms with 32-bit entries; with 128-byte rows the wasted lines cost more than the branches)
					v0[u] = *row_at((ok && h.x != BS_NONE) ? h.x : (uint32_t) ch.lo);
					v1[u] = *row_at((ok && h.z != BS_NONE) ? h.z : (uint32_t) ch.lo);
				} else {
					v0[u] = (ok && h.x != BS_NONE) ? *row_at(h.x) : 0u;
					v1[u] = (ok && h.z != BS_NONE) ? *row_at(h.z) : 0u;
				}
			}
#pragma unroll
			for (int u = 0; u < Geo::UNR; u++) {
				const int s = (pass * Geo::UNR + u) * Geo::ROWS_PER_ITER + slot0;
				if (s < nrows && lane_ok) {
					const uint4 h = fh[s];
					uint32_t x = b.sparse_init ? ring[s * RSTR + wl] : acc[u];
					if constexpr (SGN) {
						// (an absent dependency was loaded as 0: its term vanishes whatever the coefficient slot holds)
						if (h.x != BS_NONE) {
							int lo, hi;
							sgn_unpack(x, lo, hi);
							sgn_mad(v0[u], (int) h.y, lo, hi);
							sgn_mad(v1[u], (int) h.w, lo, hi);
							x = sgn_pack(sgn_reduce(lo, G), sgn_reduce(hi, G));
						}
					} else {
						if (h.x != BS_NONE)
							x = w_submul<PACKED, PLAIN>(x, v0[u], h.y, F, bm);
						if (h.z != BS_NONE)
							x = w_submul<PACKED, PLAIN>(x, v1[u], h.w, F, bm);
					}
					ring[s * RSTR + wl] = x;
				}
			}
		}
		lds_barrier();
		tick(2);          // phase A
		if (ch.npn < 0) {          // (bit 31 of the descriptor)
			// rows with more than two outside dependencies (long rows of U): the rest of their lists
			for (int s = slot0; s < nrows; s += Geo::ROWS_PER_ITER) {
				const int c = ch.lo + s;
				const uint64_t e0 = b.far_rp[c], e1 = b.far_rp[c + 1];
				if (e0 == e1 || !lane_ok)
					continue;
				uint32_t x = ring[s * RSTR + wl];
				if constexpr (SGN) {
					int lo, hi;
					sgn_unpack(x, lo, hi);
					for (uint64_t e = e0; e < e1; e += 4) {          // four terms per reduction
#pragma unroll
						for (int t = 0; t < 4; t++) {
							const uint2 en = (e + t < e1) ? b.far[e + t] : uint2{0u, 0u};
							sgn_mad((e + t < e1) ? *row_at(en.x) : 0u, (int) en.y, lo, hi);
						}
						lo = sgn_reduce(lo, G);
						hi = sgn_reduce(hi, G);
					}
					x = sgn_pack(lo, hi);
				} else {
					for (uint64_t e = e0; e < e1; e++) {
						const uint2 en = b.far[e];
						x = w_submul<PACKED, PLAIN>(x, *row_at(en.x), en.y, F, bm);
					}
				}
				ring[s * RSTR + wl] = x;
			}
			lds_barrier();
		}

		tick(3);          // long lists of outside dependencies
		// ---- phase B: the chain of levels, in LDS ----
		// Columns never meet in a triangular solve, so every wave takes ITS columns (WPW words of every row) through all the
		// levels of the chunk on its own: no barrier between the levels -- the LDS serves a wave's reads and writes in
		// order.  The rows come as a flat table of passes (32 rows of one level each, levels in order): one 16-byte entry
		// holds everything a row with one or two dependencies inside the chunk needs, and the entry of the next trip is
		// in flight during the arithmetic of this one.  (Every wave decodes every entry: the table is what keeps that cheap.)
		{
			const int niter = ((b.dbg & 2) ? 0 : ch.npass) * (PASSROWS / RSB);
			uint4 e = (niter > 0) ? ptab[rsb] : EMPTY_ENTRY;
			for (int it = 0; it < niter; it++) {
				const uint4 e_next = (it + 1 < niter) ? ptab[(it + 1) * RSB + rsb] : EMPTY_ENTRY;
				const int cnt = (int) (e.x >> 16);                    // cnt != 0: this lane has a row in the pass
				if constexpr (SGN) {
					// no branch: an empty slot reads and writes the spare row.  (With a branch around the reads the compiler has
					// to wait for the write of a pass before it may look at the next table entry.)
					const int slot = (int) (e.x & 0xFFFFu);
					const uint32_t x0 = ring[slot * RSTR + wlb];
					const uint32_t v0 = ring[(e.y & 0xFFFFu) * RSTR + wlb];
					const uint32_t v1 = ring[(e.y >> 16) * RSTR + wlb];          // (one dependency: the same row again, coefficient 0)
					int lo, hi;
					sgn_unpack(x0, lo, hi);
					sgn_mad(v0, (int) e.z, lo, hi);
					sgn_mad(v1, (cnt > 2) ? 0 : (int) e.w, lo, hi);          // (more than two: .w is the offset of the others in the list)
					if (__ballot(cnt > 2) != 0) {
						if (cnt > 2) {
							const int rest = cnt - 1;
							for (int j = 0; j < rest; j += 4) {
								uint2 em[4];
								uint32_t wm[4];
#pragma unroll
								for (int t = 0; t < 4; t++)
									em[t] = (j + t < rest) ? near[e.w + j + t] : uint2{(uint32_t) slot, 0u};
#pragma unroll
								for (int t = 0; t < 4; t++)
									wm[t] = ring[em[t].x * RSTR + wlb];
								lo = sgn_reduce(lo, G);          // (one term or four are in already: four more need a fresh start)
								hi = sgn_reduce(hi, G);
#pragma unroll
								for (int t = 0; t < 4; t++)
									sgn_mad(wm[t], (int) em[t].y, lo, hi);
							}
						}
					}
					ring[slot * RSTR + wlb] = sgn_pack(sgn_reduce(lo, G), sgn_reduce(hi, G));
				} else if (cnt != 0) {
					const int slot = (int) (e.x & 0xFFFFu);
					uint32_t x = ring[slot * RSTR + wlb];
					const uint32_t v0 = ring[(e.y & 0xFFFFu) * RSTR + wlb];
					if (cnt <= 2) {
						const uint32_t v1 = ring[(e.y >> 16) * RSTR + wlb];          // (one dependency: the same row again, coefficient 0)
						if constexpr (SGN) {
							int lo, hi;
							sgn_unpack(x, lo, hi);
							sgn_mad(v0, (int) e.z, lo, hi);
							sgn_mad(v1, (int) e.w, lo, hi);
							x = sgn_pack(sgn_reduce(lo, G), sgn_reduce(hi, G));
						} else {
							x = w_submul<PACKED, PLAIN>(x, v0, e.z, F, bm);
							if (cnt == 2)
								x = w_submul<PACKED, PLAIN>(x, v1, e.w, F, bm);
						}
					} else {
						// the first dependency inline, the others in the list
						const int rest = cnt - 1;
						if constexpr (SGN) {
							int lo, hi;
							sgn_unpack(x, lo, hi);
							sgn_mad(v0, (int) e.z, lo, hi);
							for (int j = 0; j < rest; j += 4) {
								uint2 em[4];
								uint32_t wm[4];
#pragma unroll
								for (int t = 0; t < 4; t++)
									em[t] = (j + t < rest) ? near[e.w + j + t] : uint2{(uint32_t) slot, 0u};
#pragma unroll
								for (int t = 0; t < 4; t++)
									wm[t] = ring[em[t].x * RSTR + wlb];
								lo = sgn_reduce(lo, G);          // (one term or four are in already: four more need a fresh start)
								hi = sgn_reduce(hi, G);
#pragma unroll
								for (int t = 0; t < 4; t++)
									sgn_mad(wm[t], (int) em[t].y, lo, hi);
							}
							x = sgn_pack(sgn_reduce(lo, G), sgn_reduce(hi, G));
						} else {
							x = w_submul<PACKED, PLAIN>(x, v0, e.z, F, bm);
							for (int j = 0; j < rest; j++) {
								const uint2 en = near[e.w + j];
								x = w_submul<PACKED, PLAIN>(x, ring[en.x * RSTR + wlb], en.y, F, bm);
							}
						}
					}
					ring[slot * RSTR + wlb] = x;
				}
				e = e_next;
			}
		}
		lds_barrier();
		tick(4);          // phase B

		// ---- phase C: write the chunk back ----
		for (int s = slot0; s < ((b.dbg & 4) ? 0 : nrows); s += Geo::ROWS_PER_ITER)
			if (lane_ok)
				*row_at((uint32_t) (ch.lo + s)) = ring[s * RSTR + wl];
		__syncthreads();          // (workgroup-scope release/acquire: later chunks read these rows; LDS metadata is free)
		tick(5);          // phase C
		if (k + 1 < b.nchunks)
			store_meta();
		__syncthreads();
		tick(6);          // metadata of the next chunk into LDS
	}
	if (profiling)
		for (int q = 0; q < 8; q++)
			b.prof[q] = pt[q];
}

// --------------------------------------------------------------------------
// S = A_n - A_p R, one wave per row.  The row of S is accumulated in LDS (one word per lane and tile), the pivotal
// entries of the input row are collected in a small LDS list and applied tile by tile.
// --------------------------------------------------------------------------
constexpr int AP_LIST = 64;        // pivotal entries applied per pass (one batch of input entries)
constexpr int AP_TU = 4;           // 64-word tiles per inner step

struct ApplyArgs {
	SchurArgs a;
	const void *R;
	int64_t ldR;                  // in entries
	const int *col;               // column -> compact id
	int r;                        // rows of R
	int Smpad;                    // Sm rounded up to whole tile groups (= ldR)
	int seg_words;                // bs_apply_s16_kernel: words of a row a wave holds in LDS at a time (a multiple of 64 * AP_TU; >= the row: one segment)
	int waves;                    // waves per workgroup
	size_t wave_bytes;            // LDS per wave: row buffer + list
	uint32_t *dense_out;
	int64_t ldS;
	// direct sparse output (no pool, no gather pass): rows are handed out in order through a ticket, every row
	// publishes its length and finds its offset by looking back over the rows before it (single-pass chained scan)
	int direct;
	unsigned long long *status;   // per row: flag << 62 | value; flag 1: value = length of the row, 2: = offset past the row
	int *ticket;                  // LB_TICKETS counters, LB_TICKET_STRIDE ints apart, zeroed before the launch
	int ntickets;                 // counters in use (<= LB_TICKETS, <= workgroups)
	int64_t *Sp;
	int *Sj;
	int *Sx;
	int64_t cap;
	int dbg;                      // timing experiments only (SPASM_HIP_BS_DEBUG): bit 3 = no loads of R, 4 = no look-back, 5 = no output stores
	int sgn;                      // R holds signed 16-bit entries (SgnDev)
	SgnDev G;
	unsigned long long *block_sum; // staged output: sum of the lengths of every block of SCAN_BLOCK rows (zeroed before the launch)
	uint32_t *stage;              // staged output (bs_apply_s16_kernel): the packed row of S, Smpad / 2 words per row, and its
	                              // number of entries in a.row_len; bs_expand_kernel writes the sparse rows afterwards
};

constexpr unsigned long long LB_FLAG_LEN = 1ull << 62, LB_FLAG_END = 2ull << 62, LB_VALUE = (1ull << 62) - 1;
constexpr int LB_PER_LANE = 1;             // predecessors inspected per lane and poll (64 per wave: 8 -> 4.23, 4 -> 3.93, 2 -> 3.69, 1 -> 3.50 ms)


// Rows are handed out in (nearly) increasing order, each to a wave that starts it at once: whoever waits for a row in the
// look-back below knows a running wave holds it.  One counter would serve every wave of the chip -- returning atomics on
// one address are served one after the other at the memory side, and a wave waited ~20 us for its row (0.9 ms of a
// 3.5 ms kernel on mk13.b5).  LB_TICKETS counters on separate cache lines each hand out the rows of one residue class;
// the classes advance at the same pace (every class is served by workgroups spread over the whole chip).
constexpr int LB_TICKETS = 16, LB_TICKET_STRIDE = 32;          // (ints: one 128-byte line per counter)

__device__ __forceinline__ int next_ticket(const ApplyArgs &d, int lane)
{
	const int c = (int) (blockIdx.x % (unsigned) d.ntickets);
	int t = 0;
	if (lane == 0)
		t = atomicAdd(d.ticket + c * LB_TICKET_STRIDE, 1);
	t = __builtin_amdgcn_readfirstlane(t);
	const long long k = (long long) t * d.ntickets + c;
	return (k < (long long) d.a.nrows) ? (int) k : d.a.nrows;
}

// Ordered output without a second pass (single-pass chained scan): row k publishes its length, adds up the lengths of
// the rows before it, back to the nearest row that already knows where it ends, and publishes its own end.  Rows are
// handed out by next_ticket().  Returns the offset of row k.
__device__ __forceinline__ unsigned long long lookback_offset(const ApplyArgs &d, int k, int count, int lane, bool &lost)
{
	if (lane == 0)
		__hip_atomic_store(&d.status[k], LB_FLAG_LEN | (unsigned long long) count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	unsigned long long prior = 0;
	long long polls = 0;
	if (d.dbg & 16)
		prior = (unsigned long long) k * 3600ull;
	for (int j = (d.dbg & 16) ? -1 : k - 1; j >= 0;) {
		unsigned long long val[LB_PER_LANE];
#pragma unroll
		for (int u = 0; u < LB_PER_LANE; u++) {
			const int idx = j - (u * 64 + lane);
			val[u] = (idx >= 0) ? __hip_atomic_load(&d.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_FLAG_END;
		}
		// position (in look-back order) of the nearest row that knows its end; every row nearer must have a length
		int first_end = 64 * LB_PER_LANE;
		bool hole = false;
		unsigned long long sum = 0;
#pragma unroll
		for (int u = LB_PER_LANE - 1; u >= 0; u--) {
			const uint64_t m_end = __ballot((val[u] >> 62) == 2);
			if (m_end != 0)
				first_end = u * 64 + __builtin_ctzll(m_end);
		}
#pragma unroll
		for (int u = 0; u < LB_PER_LANE; u++) {
			const int pos = u * 64 + lane;
			const bool counts = pos <= first_end;
			hole = hole || (counts && (val[u] >> 62) == 0);
			sum += counts ? (val[u] & LB_VALUE) : 0ull;
		}
		if (__ballot(hole) != 0) {          // a row in between has not published its length yet: look again
			if (++polls > (1ll << 24)) {
				lost = true;
				break;
			}
			__builtin_amdgcn_s_sleep(2);
			continue;
		}
		// wave-wide sum (64-bit, two halves through DPP-free shuffles)
		for (int sft = 32; sft >= 1; sft >>= 1) {
			const uint32_t lo32 = (uint32_t) __shfl_xor((int) (uint32_t) sum, sft);
			const uint32_t hi32 = (uint32_t) __shfl_xor((int) (uint32_t) (sum >> 32), sft);
			sum += ((unsigned long long) hi32 << 32) | lo32;
		}
		prior += sum;
		if (first_end < 64 * LB_PER_LANE)
			break;
		j -= 64 * LB_PER_LANE;
	}
	if (lane == 0) {
		__hip_atomic_store(&d.status[k], LB_FLAG_END | ((prior + (unsigned long long) count) & LB_VALUE), __ATOMIC_RELAXED,
		                   __HIP_MEMORY_SCOPE_AGENT);
		d.Sp[k] = (int64_t) prior;
		if (k == d.a.nrows - 1)
			d.Sp[d.a.nrows] = (int64_t) (prior + (unsigned long long) count);
	}
	return prior;
}

template <bool PACKED, bool PLAIN> __global__ __launch_bounds__(512) void bs_apply_kernel(ApplyArgs d)
{
	const uint32_t bm = PLAIN ? (uint32_t) (0x100000000ull / d.a.F.p) : 0u;
	constexpr int CPL = Word<PACKED>::CPL;
	using Elem = typename Word<PACKED>::Elem;
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const SchurArgs &a = d.a;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const MontDev F = a.F;
	const int Sm = a.Sm;
	const int nwords = d.Smpad / CPL;                     // a multiple of 64 * AP_TU / CPL >= 128
	uint2 *plist = reinterpret_cast<uint2 *>(lds_raw + (size_t) wave * d.wave_bytes);
	uint32_t *xw = reinterpret_cast<uint32_t *>(plist + AP_LIST);
	Elem *xe = reinterpret_cast<Elem *>(xw);
	const uint32_t *R = static_cast<const uint32_t *>(d.R);
	const int64_t ldw = d.ldR / CPL;
	unsigned long long st_input = 0, st_piv = 0;
	int st_done = 0;

	for (int k = blockIdx.x * d.waves + wave;; k += gridDim.x * d.waves) {
		if (d.direct)
			k = next_ticket(d, lane);
		if (k >= a.nrows)
			break;
		const int i = a.rows[k];
		const int64_t lo = a.Ap[i], hi = a.Ap[i + 1];
		st_input += (unsigned long long) (hi - lo);
		for (int t = lane; t < nwords; t += 64)
			xw[t] = 0;
		int npl = 0;                         // entries waiting in plist (wave-uniform)
		for (int64_t base = lo;; base += 64) {
			// non-pivotal entries go straight into the row buffer, pivotal ones are queued
			bool piv = false;
			uint32_t cid = 0, v = 0;
			if (base + lane < hi) {
				cid = (uint32_t) d.col[a.Aj[base + lane]];
				v = reduce_sum(from_balanced(a.Ax[base + lane], F), F);
				if (cid >= (uint32_t) d.r) {
					const uint32_t t = cid - (uint32_t) d.r;
					uint32_t sum = (uint32_t) xe[t] + v;
					if (sum < v || sum >= F.p)
						sum -= F.p;
					xe[t] = (Elem) sum;
				} else {
					piv = v != 0;
				}
			}
			const uint64_t mk = __ballot(piv);
			if (piv)
				plist[npl + __popcll(mk & ((1ull << lane) - 1ull))] = uint2{cid, PLAIN ? v : montmul(v, F.r2, F)};
			npl += __popcll(mk);
			st_piv += (unsigned long long) __popcll(mk);
			const bool last = base + 64 >= hi;
			if (npl > 0 && (last || npl + 64 > AP_LIST)) {
				// apply the queued pivotal entries: x[tile] -= sum_e a_e R[e][tile]
				for (int t0 = 0; t0 < nwords; t0 += 64 * AP_TU) {
					uint32_t acc[AP_TU];
#pragma unroll
					for (int u = 0; u < AP_TU; u++)
						acc[u] = (t0 + u * 64 < nwords) ? xw[t0 + u * 64 + lane] : 0u;
					// four rows of R at a time: up to 16 loads in flight per lane (a row of mk13.b5 has three pivotal entries:
					// one round trip per tile group)
					for (int e = 0; e < npl; e += 4) {
						uint2 pe[4];
						uint32_t w[4][AP_TU];
#pragma unroll
						for (int q = 0; q < 4; q++) {
							pe[q] = (e + q < npl) ? plist[e + q] : uint2{0u, 0u};          // (coefficient 0: row 0 of R, no effect)
							const uint32_t *rq = R + (int64_t) pe[q].x * ldw + t0 + lane;
#pragma unroll
							for (int u = 0; u < AP_TU; u++)
								w[q][u] = (d.dbg & 8) ? (uint32_t) (lane + u) : (e + q < npl && t0 + u * 64 < nwords) ? rq[u * 64] : 0u;
						}
#pragma unroll
						for (int q = 0; q < 4; q++)
#pragma unroll
							for (int u = 0; u < AP_TU; u++)
								acc[u] = w_submul<PACKED, PLAIN>(acc[u], w[q][u], pe[q].y, F, bm);
					}
#pragma unroll
					for (int u = 0; u < AP_TU; u++)
						if (t0 + u * 64 < nwords)
							xw[t0 + u * 64 + lane] = acc[u];
				}
				npl = 0;
			}
			if (last)
				break;
		}

		// ---- output ----
		if (d.dense_out != nullptr) {
			uint32_t *out = d.dense_out + (int64_t) k * d.ldS;
			for (int t = lane; t < Sm; t += 64)
				out[t] = xe[t];
			if (lane == 0)
				a.row_len[k] = Sm;
			st_done += 1;
			continue;
		}
		// lane l of a tile holds columns CPL * (t0 + l) ..: entries come out sorted by column
		int count = 0;
		for (int t0 = 0; t0 < nwords; t0 += 64) {
			const uint32_t w = xw[t0 + lane];
			if constexpr (PACKED)
				count += __popcll(__ballot((w & 0xFFFFu) != 0)) + __popcll(__ballot((w >> 16) != 0));
			else
				count += __popcll(__ballot(w != 0));
		}
		if (d.stage != nullptr) {          // staged output: the row as it stands + its length (see bs_expand_kernel)
			uint32_t *out = d.stage + (int64_t) k * nwords;
			for (int t = lane; t < nwords; t += 64)
				out[t] = xw[t];
			if (lane == 0) {
				a.row_len[k] = count;
				atomicAdd(&d.block_sum[k / 1024], (unsigned long long) count);          // (SCAN_BLOCK rows per block)
			}
			st_done += 1;
			continue;
		}
		int64_t off = 0;
		int *out_j = a.pool_j, *out_x = a.pool_x;
		bool fits;
		if (d.direct) {
			bool lost = false;
			const unsigned long long prior = lookback_offset(d, k, count, lane, lost);
			if (lane == 0 && lost)
				atomicOr(&a.ctr[CTR_STATUS], 4);
			off = (int64_t) prior;
			out_j = d.Sj;
			out_x = d.Sx;
			fits = !lost && off + count <= d.cap;
		} else {
			unsigned long long got = 0;
			if (lane == 0)
				got = atomicAdd(&a.ctr64[C64_POOL], (unsigned long long) count);
			const uint32_t g_lo = __builtin_amdgcn_readfirstlane((uint32_t) got);
			const uint32_t g_hi = __builtin_amdgcn_readfirstlane((uint32_t) (got >> 32));
			off = (int64_t) (((uint64_t) g_hi << 32) | g_lo);
			fits = off + count <= a.pool_cap;
		}
		if (fits && !(d.dbg & 32)) {
			int64_t wpos = off;
			const uint64_t below = (1ull << lane) - 1ull;
			for (int t0 = 0; t0 < nwords; t0 += 64) {
				const uint32_t w = xw[t0 + lane];
				if constexpr (PACKED) {
					const uint32_t v0 = w & 0xFFFFu, v1 = w >> 16;
					const uint64_t m0 = __ballot(v0 != 0), m1 = __ballot(v1 != 0);
					const int before = __popcll(m0 & below) + __popcll(m1 & below);
					const int c0 = 2 * (t0 + lane);
					if (v0 != 0) {
						out_j[wpos + before] = a.q[c0];
						out_x[wpos + before] = to_balanced(v0, F);
					}
					if (v1 != 0) {
						const int64_t dst = wpos + before + (v0 != 0 ? 1 : 0);
						out_j[dst] = a.q[c0 + 1];
						out_x[dst] = to_balanced(v1, F);
					}
					wpos += __popcll(m0) + __popcll(m1);
				} else {
					const uint64_t mk = __ballot(w != 0);
					if (w != 0) {
						const int64_t dst = wpos + __popcll(mk & below);
						out_j[dst] = a.q[t0 + lane];
						out_x[dst] = to_balanced(w, F);
					}
					wpos += __popcll(mk);
				}
			}
		}
		if (lane == 0) {
			if (fits) {
				if (!d.direct) {
					a.row_off[k] = off | (1LL << 62);       // sorted by column already
					a.row_len[k] = count;
				}
			} else {
				atomicOr(&a.ctr[CTR_STATUS], 1);
				if (!d.direct)
					a.row_len[k] = -1;
			}
		}
		st_done += fits ? 1 : 0;
	}
	if (lane == 0) {
		atomicAdd(&a.ctr64[C64_INPUT], st_input);
		atomicAdd(&a.ctr64[C64_ELIM], st_piv);          // rows of R combined (not the reference's count of eliminations)
		atomicAdd(&a.ctr[a.done_ctr], st_done);
	}
}

// The same with signed 16-bit entries (SgnDev):
//   * one v_mad_i32_i16 per term and component, one reduction per four rows of R;
//   * the rows of R of the next tile group are in flight while the current one is multiplied;
//   * the entries are counted while the last batch is applied (no separate pass over the row), the output loop works on
//     32-bit offsets from a uniform base.
__global__ __launch_bounds__(512) void bs_apply_s16_kernel(ApplyArgs d)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const SchurArgs &a = d.a;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const MontDev F = a.F;
	const SgnDev G = d.G;
	const int Sm = a.Sm;
	const int nwords_row = d.Smpad / 2;                   // a multiple of 64 * AP_TU
	// A row wider than the LDS a wave may have (mk14.b4: 42,356 non-pivotal columns = 85 KB) is produced in SEGMENTS of
	// seg_words words: the few entries of the input row are walked once per segment, the rows of R are read segment by
	// segment -- every byte of R once, as before.  Staged and dense output only (the offset of a sparse row is known once
	// its last segment is).
	const int nseg = (nwords_row + d.seg_words - 1) / d.seg_words;
	uint2 *plist = reinterpret_cast<uint2 *>(lds_raw + (size_t) wave * d.wave_bytes);
	uint32_t *xw = reinterpret_cast<uint32_t *>(plist + AP_LIST);
	short *xe = reinterpret_cast<short *>(xw);
	const uint32_t *R = static_cast<const uint32_t *>(d.R);
	const uint32_t ldw256 = (uint32_t) (d.ldR / 2 / 256);          // row stride of R in units of 256 words (rows are padded to 512 columns):
	                                                               // the offset of a row in these units fits 32 bits for any R below 4 TB
	const int2 *q2 = reinterpret_cast<const int2 *>(a.q);          // (the array is padded to whole tile groups)
	unsigned long long st_input = 0, st_piv = 0;
	int st_done = 0;

	for (int k = blockIdx.x * d.waves + wave;; k += gridDim.x * d.waves) {
		if (d.direct) {
			k = next_ticket(d, lane);
		}
		if (k >= a.nrows)
			break;
		const int i = a.rows[k];
		const int64_t lo = a.Ap[i], hi = a.Ap[i + 1];
		st_input += (unsigned long long) (hi - lo);
		int total = 0;                       // entries of the segments done so far
	  for (int seg = 0; seg < nseg; seg++) {
		const int w0 = seg * d.seg_words, nwords = min(d.seg_words, nwords_row - w0);
		for (int t = lane; t < nwords; t += 64)
			xw[t] = 0;
		int npl = 0;                         // entries waiting in plist (wave-uniform)
		int count = -1;                      // entries of the finished segment, once known
		for (int64_t base = lo;; base += 64) {
			bool piv = false;
			uint32_t cid = 0;
			int bal = 0;
			if (base + lane < hi) {
				cid = (uint32_t) d.col[a.Aj[base + lane]];
				bal = sgn_from_residue(reduce_sum(from_balanced(a.Ax[base + lane], F), F), G);
				if (cid >= (uint32_t) d.r) {
					const uint32_t t = cid - (uint32_t) d.r - 2u * (uint32_t) w0;          // (unsigned: columns before the segment wrap around)
					if (t < 2u * (uint32_t) nwords)
						xe[t] = (short) sgn_canonical((int) xe[t] + bal, G);
				} else {
					piv = bal != 0;
				}
			}
			const uint64_t mk = __ballot(piv);
			if (piv)
				plist[npl + __popcll(mk & ((1ull << lane) - 1ull))] = uint2{cid * ldw256, (uint32_t) (-bal)};          // (where the row of R starts, in units of 256 words)
			npl += __popcll(mk);
			st_piv += (seg == 0) ? (unsigned long long) __popcll(mk) : 0ull;
			const bool last = base + 64 >= hi;
			if (npl > 0 && (last || npl + 64 > AP_LIST)) {
				// apply the queued pivotal entries: x[tile group] += sum_e (-a_e) R[e][tile group], four rows of R at a
				// time.  A unit = (tile group, four entries of the list); the loads of unit u + 1 are issued before unit
				// u is multiplied.
				const int ne = (npl + 3) >> 2;
				const int nunits = (nwords / (64 * AP_TU)) * ne;
				int cnt = 0;
				int alo[AP_TU], ahi[AP_TU];
				uint32_t wa[4][AP_TU], wb[4][AP_TU];
				int ca[4], cb[4];
				int pf_t0 = 0, pf_e = 0;                 // unit the next issue() will load
				// (no branch in here: the wait counts of the multiplications are static, and a path that skips an issue
				//  would force them all to the count of that path.  Past the last unit the last tile group is read again.)
				const int t0_last = nwords - 64 * AP_TU;
				const int64_t row_mask = (d.dbg & 8) ? 0 : -1;
				auto issue = [&](uint32_t (&w)[4][AP_TU], int (&cf)[4]) {
					const int t0c = (pf_t0 < t0_last) ? pf_t0 : t0_last;
#pragma unroll
					for (int q = 0; q < 4; q++) {
						const int idx = (pf_e + q < npl) ? pf_e + q : npl - 1;
						const uint2 pe = plist[idx];
						cf[q] = (pf_e + q < npl) ? (int) pe.y : 0;          // (coefficient 0: no effect)
						const uint32_t *rq = R + ((((uint64_t) pe.x) << 8) & (uint64_t) row_mask) + w0 + t0c + lane;
#pragma unroll
						for (int u = 0; u < AP_TU; u++)
							w[q][u] = rq[u * 64];          // (rows are padded to whole tile groups)
					}
					pf_e += 4;
					const bool wrap = pf_e >= npl;
					pf_e = wrap ? 0 : pf_e;
					pf_t0 += wrap ? 64 * AP_TU : 0;
				};
				int t0 = 0, e = 0;                       // unit being multiplied
				auto multiply = [&](const uint32_t (&w)[4][AP_TU], const int (&cf)[4]) {
					if (e == 0) {
#pragma unroll
						for (int u = 0; u < AP_TU; u++)
							sgn_unpack(xw[t0 + u * 64 + lane], alo[u], ahi[u]);
					} else {                 // four terms were added already
#pragma unroll
						for (int u = 0; u < AP_TU; u++) {
							alo[u] = sgn_reduce(alo[u], G);
							ahi[u] = sgn_reduce(ahi[u], G);
						}
					}
#pragma unroll
					for (int q = 0; q < 4; q++)
#pragma unroll
						for (int u = 0; u < AP_TU; u++)
							sgn_mad(w[q][u], cf[q], alo[u], ahi[u]);
					e += 4;
					if (e >= npl) {
#pragma unroll
						for (int u = 0; u < AP_TU; u++) {
							const int rl = sgn_reduce(alo[u], G), rh = sgn_reduce(ahi[u], G);
							xw[t0 + u * 64 + lane] = sgn_pack(rl, rh);
							cnt += __popcll(__ballot(rl != 0)) + __popcll(__ballot(rh != 0));
						}
						e = 0;
						t0 += 64 * AP_TU;
					}
				};
				issue(wa, ca);
				for (int u = 0; u < nunits; u += 2) {
					issue(wb, cb);
					multiply(wa, ca);
					issue(wa, ca);
					if (u + 1 < nunits)
						multiply(wb, cb);
				}
				if (last)
					count = cnt;
				npl = 0;
			}
			if (last)
				break;
		}

		// ---- output ----
		if (d.dense_out != nullptr) {
			uint32_t *out = d.dense_out + (int64_t) k * d.ldS + 2 * w0;
			for (int t = lane; t < min(2 * nwords, Sm - 2 * w0); t += 64) {
				const int v = (int) xe[t];
				out[t] = (uint32_t) (v < 0 ? v + G.p : v);
			}
			if (seg + 1 < nseg)
				continue;
			if (lane == 0)
				a.row_len[k] = Sm;
			st_done += 1;
			continue;
		}
		if (count < 0) {                     // no pivotal entry in the last batch: the segment was not swept
			count = 0;
			for (int t0 = 0; t0 < nwords; t0 += 64) {
				const uint32_t w = xw[t0 + lane];
				count += __popcll(__ballot((w & 0xFFFFu) != 0)) + __popcll(__ballot((w >> 16) != 0));
			}
		}
		total += count;
		if (d.stage != nullptr) {
			uint32_t *out = d.stage + (int64_t) k * nwords_row + w0;
			for (int t = lane; t < nwords; t += 64)
				out[t] = xw[t];
			if (seg + 1 < nseg)
				continue;
			if (lane == 0) {
				a.row_len[k] = total;
				atomicAdd(&d.block_sum[k / 1024], (unsigned long long) total);          // (SCAN_BLOCK rows per block)
			}
			st_done += 1;
			continue;
		}
		// (sparse rows straight from the LDS: one segment, the launcher sees to it)
		int64_t off = 0;
		int *out_j = a.pool_j, *out_x = a.pool_x;
		bool fits;
		if (d.direct) {
			bool lost = false;
			const unsigned long long prior = lookback_offset(d, k, count, lane, lost);
			if (lane == 0 && lost)
				atomicOr(&a.ctr[CTR_STATUS], 4);
			off = (int64_t) prior;
			out_j = d.Sj;
			out_x = d.Sx;
			fits = !lost && off + count <= d.cap;
		} else {
			unsigned long long got = 0;
			if (lane == 0)
				got = atomicAdd(&a.ctr64[C64_POOL], (unsigned long long) count);
			const uint32_t g_lo = __builtin_amdgcn_readfirstlane((uint32_t) got);
			const uint32_t g_hi = __builtin_amdgcn_readfirstlane((uint32_t) (got >> 32));
			off = (int64_t) (((uint64_t) g_hi << 32) | g_lo);
			fits = off + count <= a.pool_cap;
		}
		if (fits && !(d.dbg & 32)) {
			// lane l of a tile holds columns 2 (t0 + l) and 2 (t0 + l) + 1: entries come out sorted by column
			int *oj = out_j + off, *ox = out_x + off;
			uint32_t wpos = 0;
			for (int t0 = 0; t0 < nwords; t0 += 64) {
				int v0, v1;
				sgn_unpack(xw[t0 + lane], v0, v1);
				v0 = sgn_canonical(v0, G);
				v1 = sgn_canonical(v1, G);
				const uint64_t m0 = __ballot(v0 != 0), m1 = __ballot(v1 != 0);
				const int2 qq = q2[t0 + lane];
				uint32_t dst = wpos;
				dst = __builtin_amdgcn_mbcnt_hi((uint32_t) (m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m0, dst));
				dst = __builtin_amdgcn_mbcnt_hi((uint32_t) (m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m1, dst));
				if (v0 != 0) {
					oj[dst] = qq.x;
					ox[dst] = v0;
					dst += 1;
				}
				if (v1 != 0) {
					oj[dst] = qq.y;
					ox[dst] = v1;
				}
				wpos += (uint32_t) (__popcll(m0) + __popcll(m1));
			}
		}
		if (lane == 0) {
			if (fits) {
				if (!d.direct) {
					a.row_off[k] = off | (1LL << 62);       // sorted by column already
					a.row_len[k] = count;
				}
			} else {
				atomicOr(&a.ctr[CTR_STATUS], 1);
				if (!d.direct)
					a.row_len[k] = -1;
			}
		}
		st_done += fits ? 1 : 0;
	  }          // segments
	}
	if (lane == 0) {
		atomicAdd(&a.ctr64[C64_INPUT], st_input);
		atomicAdd(&a.ctr64[C64_ELIM], st_piv);
		atomicAdd(&a.ctr[a.done_ctr], st_done);
	}
}

// ---- staged sparse output (signed 16-bit entries) --------------------------------------------------
// Writing the rows of S in order from ONE kernel makes every row wait for the lengths of the rows before it -- with
// 3584 rows in flight a row waits for the slowest of its several hundred running predecessors, a third of its own time
// on mk13.b5 -- and the waiting rows keep their LDS.  Two kernels instead: bs_apply_s16_kernel leaves the packed row
// (2 bytes per column) and its length, one small scan turns lengths into offsets, bs_expand_kernel streams the
// packed rows out as (column, value) pairs.  Nobody waits for anybody; the price is one write and one read of the
// packed rows (a third of the bytes of the result).
// offsets from lengths.  The apply kernel has added every length to the sum of its block of SCAN_BLOCK rows
// (block_sum); workgroup g adds up the sums of the blocks before it and scans its own block.
constexpr int SCAN_BLOCK = 1024;

__global__ __launch_bounds__(SCAN_BLOCK) void bs_scan_lengths_kernel(const int *len, int n, const unsigned long long *block_sum, int64_t *Sp, int64_t cap,
                                                                      int *ctr)
{
	__shared__ long long part[SCAN_BLOCK / 64];
	__shared__ long long s_base;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int g = blockIdx.x;
	auto wave_sum = [](long long v) -> long long {
		for (int sft = 32; sft >= 1; sft >>= 1) {
			const uint32_t lo32 = (uint32_t) __shfl_xor((int) (uint32_t) v, sft);
			const uint32_t hi32 = (uint32_t) __shfl_xor((int) (uint32_t) ((unsigned long long) v >> 32), sft);
			v += (long long) (((unsigned long long) hi32 << 32) | lo32);
		}
		return v;
	};
	// blocks before this one
	long long before = 0;
	for (int b = tid; b < g; b += SCAN_BLOCK)
		before += (long long) block_sum[b];
	before = wave_sum(before);
	if (lane == 0)
		part[wave] = before;
	__syncthreads();
	if (tid == 0) {
		long long t = Sp[0];                 // (where this slice of rows starts)
		for (int w = 0; w < SCAN_BLOCK / 64; w++)
			t += part[w];
		s_base = t;
	}
	__syncthreads();
	const long long base = s_base;
	__syncthreads();
	// inclusive scan of the block: inside a wave, then over the waves
	const int k = g * SCAN_BLOCK + tid;
	const long long mine = (k < n) ? (long long) len[k] : 0;
	long long incl = mine;
	for (int sft = 1; sft < 64; sft <<= 1) {
		const uint32_t lo32 = (uint32_t) __shfl_up((int) (uint32_t) incl, sft);
		const uint32_t hi32 = (uint32_t) __shfl_up((int) (uint32_t) ((unsigned long long) incl >> 32), sft);
		if (lane >= sft)
			incl += (long long) (((unsigned long long) hi32 << 32) | lo32);
	}
	if (lane == 63)
		part[wave] = incl;
	__syncthreads();
	long long waves_before = 0;
	for (int w = 0; w < wave; w++)
		waves_before += part[w];
	const long long end = base + waves_before + incl;
	if (k < n) {
		Sp[k + 1] = end;
		if (k == n - 1 && end > cap)
			atomicOr(&ctr[CTR_STATUS], 1);
	}
}

struct ExpandArgs {
	const uint32_t *stage;
	int nwords, nrows;
	const int64_t *Sp;
	int *Sj, *Sx;
	int64_t cap;
	const int *q;
	SgnDev G;                     // (p, -p, p / 2: every format uses them)
	int dbg;
};

// FMT 0: signed 16-bit entries, two per word; 1: residues in 16 bits, two per word; 2: residues, one per word

template <int FMT> __global__ __launch_bounds__(256) void bs_expand_kernel(ExpandArgs e)
{
	constexpr int TU = 4;                // tiles per group (rows are padded to whole groups of four tiles)
	// The entries of a group are compacted in LDS and leave as dense, contiguous stores (64 consecutive entries per
	// instruction).  Written straight from the lanes that hold them, an instruction covers every other slot of a
	// 512-byte span and the holes are filled by the next one: the same bytes took 1.74 ms instead of 1.1.
	__shared__ int stage_j[4][128 * TU], stage_x[4][128 * TU];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int wave = (int) ((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int) ((gridDim.x * blockDim.x) >> 6);
	const int2 *q2 = reinterpret_cast<const int2 *>(e.q) + lane;          // (two columns per word)
	const int *q1 = e.q + lane;                                             // (one column per word: FMT 2)
	const SgnDev G = e.G;
	const uint32_t pu = (uint32_t) G.p, halfu = (uint32_t) G.half;
	auto balanced = [&](uint32_t v) -> int { return (v > halfu) ? (int) (v - pu) : (int) v; };          // residue -> [-p/2, p/2]
	const int last = e.nwords - 64 * TU;
	int *sj = stage_j[wv], *sx = stage_x[wv];
	for (int k = wave; k < e.nrows; k += nwaves) {
		const int64_t off = e.Sp[k], end = e.Sp[k + 1];
		if (end > e.cap || end == off)
			continue;                    // (the scan has raised the overflow flag) / empty row
		const uint32_t *row = e.stage + (int64_t) k * e.nwords + lane;
		int *oj = e.Sj + off, *ox = e.Sx + off;
		uint32_t wpos = 0;
		if (e.dbg & 128) {          // timing experiment: the same bytes as dense, contiguous stores (wrong results)
			uint32_t acc = 0;
			for (int t0 = 0; t0 < e.nwords; t0 += 64)
				acc += row[t0];
			for (int64_t t = lane; t < end - off; t += 64) {
				oj[t] = (int) acc;
				ox[t] = (int) acc;
			}
			continue;
		}
		// The loads of the next group are issued BEFORE the stores of this one: stores and loads share one in-order
		// counter.  (No branch around the issue -- the wait counts are static; past the end the last group is read again.)
		uint32_t wa[TU], wb[TU];
		int2 qa[TU], qb[TU];
		auto issue = [&](int t0, uint32_t (&w)[TU], int2 (&qq)[TU]) {
			const int t = (t0 < last) ? t0 : last;
#pragma unroll
			for (int u = 0; u < TU; u++)
				w[u] = __builtin_nontemporal_load(row + t + 64 * u);
#pragma unroll
			for (int u = 0; u < TU; u++)
				qq[u] = (FMT == 2) ? int2{q1[t + 64 * u], 0} : q2[t + 64 * u];
		};
		// lane l of a tile holds columns 2 (t0 + l) and 2 (t0 + l) + 1: entries come out sorted by column
		auto emit = [&](const uint32_t (&w)[TU], const int2 (&qq)[TU]) {
			uint32_t gpos = 0;
#pragma unroll
			for (int u = 0; u < TU; u++) {
				int v0, v1;
				if constexpr (FMT == 0) {
					sgn_unpack(w[u], v0, v1);
					v0 = sgn_canonical(v0, G);
					v1 = sgn_canonical(v1, G);
				} else if constexpr (FMT == 1) {
					v0 = balanced(w[u] & 0xFFFFu);
					v1 = balanced(w[u] >> 16);
				} else {
					v0 = balanced(w[u]);
					v1 = 0;
				}
				const uint64_t m0 = __ballot(v0 != 0), m1 = __ballot(v1 != 0);
				uint32_t dst = gpos;
				dst = __builtin_amdgcn_mbcnt_hi((uint32_t) (m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m0, dst));
				dst = __builtin_amdgcn_mbcnt_hi((uint32_t) (m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m1, dst));
				if (v0 != 0) {
					sj[dst] = qq[u].x;
					sx[dst] = v0;
					dst += 1;
				}
				if (v1 != 0) {
					sj[dst] = qq[u].y;
					sx[dst] = v1;
				}
				gpos += (uint32_t) (__popcll(m0) + __popcll(m1));
			}
			for (uint32_t t = lane; t < gpos; t += 64) {
				oj[wpos + t] = sj[t];
				ox[wpos + t] = sx[t];
			}
			wpos += gpos;
		};
		issue(0, wa, qa);
		for (int t0 = 0; t0 < e.nwords; t0 += 2 * 64 * TU) {
			issue(t0 + 64 * TU, wb, qb);
			emit(wa, qa);
			issue(t0 + 2 * 64 * TU, wa, qa);
			if (t0 + 64 * TU < e.nwords)
				emit(wb, qb);
		}
	}
}

// Dense rows for FEW, LONG input rows (the completion test combines every remaining row: ~10 rows of ~10^5 entries):
// one wave per row would leave the chip idle, so a workgroup takes (row, group of 256 words) and its waves split the
// entries of the row; partial sums meet in LDS.
constexpr int AW_NW = 8;

template <bool PACKED, bool PLAIN> __global__ __launch_bounds__(64 * AW_NW) void bs_apply_wide_kernel(ApplyArgs d)
{
	const uint32_t bm = PLAIN ? (uint32_t) (0x100000000ull / d.a.F.p) : 0u;
	constexpr int CPL = Word<PACKED>::CPL;
	__shared__ uint32_t part[AW_NW][AP_TU * 64];
	const SchurArgs &a = d.a;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const MontDev F = a.F;
	const int k = blockIdx.y;
	const int w0 = blockIdx.x * 64 * AP_TU;                 // first word of this workgroup's tile group
	const int nwords = d.Smpad / CPL;
	const uint32_t *R = static_cast<const uint32_t *>(d.R);
	const int64_t ldw = d.ldR / CPL;
	const int i = a.rows[k];
	const int64_t lo = a.Ap[i], hi = a.Ap[i + 1];
	uint32_t acc[AP_TU];
#pragma unroll
	for (int u = 0; u < AP_TU; u++)
		acc[u] = 0;
	for (int64_t base = lo + 64 * wave; base < hi; base += 64 * AW_NW) {
		uint32_t cid = 0xFFFFFFFFu, v = 0;
		if (base + lane < hi) {
			cid = (uint32_t) d.col[a.Aj[base + lane]];
			v = reduce_sum(from_balanced(a.Ax[base + lane], F), F);
		}
		// non-pivotal entries of this tile group: added by the lane that owns the column
		const bool np_here = v != 0 && cid != 0xFFFFFFFFu && cid >= (uint32_t) d.r;
		uint64_t todo = __ballot(np_here);
		while (todo != 0) {
			const int src = __builtin_ctzll(todo);
			todo &= todo - 1;
			const uint32_t t = (uint32_t) __shfl((int) cid, src) - (uint32_t) d.r;      // column among the non-pivotal ones
			const uint32_t val = (uint32_t) __shfl((int) v, src);
			const int word = (int) (t / CPL) - w0;
			if (word >= 0 && word < 64 * AP_TU && (word & 63) == lane) {
				const int u = word >> 6;
#pragma unroll
				for (int q = 0; q < AP_TU; q++)
					if (q == u) {
						if constexpr (PACKED) {
							const int sh = (t & 1) ? 16 : 0;
							uint32_t part_v = ((acc[q] >> sh) & 0xFFFFu) + val;
							if (part_v >= F.p)
								part_v -= F.p;
							acc[q] = (acc[q] & ~(0xFFFFu << sh)) | (part_v << sh);
						} else {
							uint32_t sum = acc[q] + val;
							if (sum < val || sum >= F.p)
								sum -= F.p;
							acc[q] = sum;
						}
					}
			}
		}
		// pivotal entries: their rows of R, tile group by tile group
		const bool piv = v != 0 && cid < (uint32_t) d.r;
		const uint32_t ay = piv ? (PLAIN ? v : montmul(v, F.r2, F)) : 0u;
		todo = __ballot(piv);
		while (todo != 0) {
			const int s0 = __builtin_ctzll(todo);
			todo &= todo - 1;
			int s1 = -1;
			if (todo != 0) {
				s1 = __builtin_ctzll(todo);
				todo &= todo - 1;
			}
			const uint32_t c0 = (uint32_t) __shfl((int) cid, s0), y0 = (uint32_t) __shfl((int) ay, s0);
			const uint32_t *r0 = R + (int64_t) c0 * ldw + w0 + lane;
			uint32_t x0[AP_TU], x1[AP_TU];
			// (signed 16-bit entries are brought back to residues: this kernel runs for a handful of rows)
			auto residues = [&](uint32_t w) -> uint32_t {
				if constexpr (PACKED) {
					if (d.sgn) {
						int l, h;
						sgn_unpack(w, l, h);
						l = (l < 0) ? l + d.G.p : l;
						h = (h < 0) ? h + d.G.p : h;
						return (uint32_t) l | ((uint32_t) h << 16);
					}
				}
				return w;
			};
#pragma unroll
			for (int u = 0; u < AP_TU; u++)
				x0[u] = (w0 + u * 64 < nwords) ? residues(r0[u * 64]) : 0u;
			uint32_t y1 = 0;
			if (s1 >= 0) {
				const uint32_t c1 = (uint32_t) __shfl((int) cid, s1);
				y1 = (uint32_t) __shfl((int) ay, s1);
				const uint32_t *r1 = R + (int64_t) c1 * ldw + w0 + lane;
#pragma unroll
				for (int u = 0; u < AP_TU; u++)
					x1[u] = (w0 + u * 64 < nwords) ? residues(r1[u * 64]) : 0u;
			}
#pragma unroll
			for (int u = 0; u < AP_TU; u++) {
				acc[u] = w_submul<PACKED, PLAIN>(acc[u], x0[u], y0, F, bm);
				if (s1 >= 0)
					acc[u] = w_submul<PACKED, PLAIN>(acc[u], x1[u], y1, F, bm);
			}
		}
	}
#pragma unroll
	for (int u = 0; u < AP_TU; u++)
		part[wave][u * 64 + lane] = acc[u];
	__syncthreads();
	if (wave == 0) {
		uint32_t *out = d.dense_out + (int64_t) k * d.ldS;
#pragma unroll
		for (int u = 0; u < AP_TU; u++) {
			uint32_t lo_s = 0, hi_s = 0;
			for (int w = 0; w < AW_NW; w++) {
				const uint32_t x = part[w][u * 64 + lane];
				if constexpr (PACKED) {
					lo_s += x & 0xFFFFu;
					hi_s += x >> 16;
					if (lo_s >= F.p)
						lo_s -= F.p;
					if (hi_s >= F.p)
						hi_s -= F.p;
				} else {
					const uint32_t sum = lo_s + x;
					lo_s = (sum < x || sum >= F.p) ? sum - F.p : sum;
				}
			}
			const int word = w0 + u * 64 + lane;
			if constexpr (PACKED) {
				if (2 * word < a.Sm)
					out[2 * word] = lo_s;
				if (2 * word + 1 < a.Sm)
					out[2 * word + 1] = hi_s;
			} else {
				if (word < a.Sm)
					out[word] = lo_s;
			}
		}
		if (lane == 0 && blockIdx.x == 0) {
			a.row_len[k] = a.Sm;
			atomicAdd(&a.ctr[a.done_ctr], 1);
			atomicAdd(&a.ctr64[C64_INPUT], (unsigned long long) (hi - lo));
		}
	}
}

template <bool PACKED, bool PLAIN, int LPR, int NW, bool SGN = false, int RING = BS_RING, int PASSROWS = BS_PASSROWS, int PASSCAP = BS_PASSCAP, int WGS_PER_CU = 1>
void launch_backsolve_variant(const BsArgs &b, int Sm, hipStream_t stream, const BsImage &B)
{
	using G = BsGeom<PACKED, LPR, NW, RING, PASSROWS, PASSCAP>;
	if (B.ring != RING || B.passrows != PASSROWS || B.passcap != PASSCAP)
		die("backsolve: the plan was built for chunks of %d rows, passes of %d rows, %d passes; this kernel takes %d / %d / %d", B.ring, B.passrows,
		    B.passcap, RING, PASSROWS, PASSCAP);
	static bool configured = false;
	if (!configured) {
		HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&backsolve_kernel<PACKED, PLAIN, LPR, NW, SGN, RING, PASSROWS, PASSCAP, WGS_PER_CU>),
		                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) G::LDS_BYTES));
		configured = true;
	}
	hipLaunchKernelGGL((backsolve_kernel<PACKED, PLAIN, LPR, NW, SGN, RING, PASSROWS, PASSCAP, WGS_PER_CU>), dim3((unsigned) ((Sm + G::CW - 1) / G::CW)),
	                   dim3(64 * NW), G::LDS_BYTES, stream, b);
}

template <bool PACKED, bool PLAIN> void launch_apply_variant(const ApplyArgs &d, int blocks, size_t lds, hipStream_t stream)
{
	static size_t configured = 0;
	if (lds > configured) {
		HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&bs_apply_kernel<PACKED, PLAIN>),
		                              hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		configured = lds;
	}
	hipLaunchKernelGGL((bs_apply_kernel<PACKED, PLAIN>), dim3(blocks), dim3(64 * d.waves), lds, stream, d);
}

}  // namespace

// offsets from row lengths + per-block sums (bs_scan_lengths_kernel), for the other producers of staged rows (sparse_image.hip)
void launch_scan_lengths(const int *len, int n, const unsigned long long *block_sum, int64_t *Sp, int64_t cap, int *ctr, hipStream_t stream)
{
	if (n <= 0)
		return;
	hipLaunchKernelGGL(bs_scan_lengths_kernel, dim3((n + SCAN_BLOCK - 1) / SCAN_BLOCK), dim3(SCAN_BLOCK), 0, stream, len, n, block_sum, Sp, cap, ctr);
}

// --------------------------------------------------------------------------
// host plan
// --------------------------------------------------------------------------
// Is the back-substituted image worth having for this factor?  Memory: r x Sm words.  Work: nnz(U') * Sm.
bool backsolve_eligible(int r, int Sm, int64_t nnz_u, int64_t *bytes, int64_t prime)
{
	const int64_t ldR = ((int64_t) Sm + 511) / 512 * 512;
	*bytes = (int64_t) r * ldR * 4;          // (2 bytes per entry when p < 2^16)
	if (r <= 0 || Sm <= 0)
		return false;
	// the apply kernels keep one row of S in LDS (96 KB); the one for signed 16-bit entries (p <= 44,927) goes through wider
	// rows in segments
	const bool segments = prime < 65536 && sgn_eligible(prime) && env_bs("SPASM_HIP_BS_SIGNED", 1) != 0 && true &&
	                      env_bs("SPASM_HIP_BS_STAGED", 1) != 0 && (1) != 0;
	if (Sm > (segments ? 131072 : 24576))
		return false;
	if ((double) (nnz_u + r) * (double) Sm > 1.5e11)
		return false;
	return true;
}

void backsolve_plan(const FactPlan &P, spasm_hip_dfact *F, hipStream_t stream)
{
	BsImage &B = F->bs;
	const int r = P.r, rpad = P.rpad, m = P.m;
	B.r = r;
	B.Sm = m - r;
	B.ldR = ((int64_t) B.Sm + 511) / 512 * 512;          // whole tile groups of the apply kernels (64 * AP_TU words of two entries): the padding stays zero
	// p < 2^16: coefficients are kept as plain residues (the kernels multiply with 24-bit products + Barrett); small p: signed
	// 16-bit entries of R, coefficients of the dependencies are NEGATED balanced residues (SgnDev above)
	B.plain = P.prime < 65536;
	B.sgn = B.plain && sgn_eligible(P.prime) && env_bs("SPASM_HIP_BS_SIGNED", 1) != 0 && true;
	// Shape of the build kernel (the plan is cut for it).  0 = 128-byte slab rows, 16 waves; 1 = 128 B, 8 waves; 2 = 64 B, 8
	// waves; 3 = 32-byte slab rows (16 columns), 8 waves, passes of 64 rows, TWO workgroups per CU.  The build is a chain
	// (DESIGN.md section 5): a workgroup takes the same time whatever the width of its slab, phase A being bound by the
	// instruction issue of ONE CU and phase B by LDS latency; so narrower slabs -- half the phase-A instructions per
	// workgroup, a whole level of up to 64 rows per pass -- and two workgroups sharing a CU, one in its issue-bound phase
	// while the other waits on the LDS, use the chip better as long as all of them are resident at once.
	{
		int cus = 0, dev = 0;
		HIP_CHECK(hipGetDevice(&dev));
		hipDeviceProp_t prop;
		HIP_CHECK(hipGetDeviceProperties(&prop, dev));
		cus = prop.multiProcessorCount;
		const bool packed = B.sgn || (P.prime < 65536 && true);
		const int slabs_small = packed ? (B.Sm + 31) / 32 : (B.Sm + 15) / 16;
		const int slabs_narrow = (B.Sm + 15) / 16;
		int shape = slabs_small <= (packed ? 2 * cus : cus) ? 2 : 0;
		if (B.sgn && slabs_narrow <= 2 * cus && (0) != 0)
			shape = 3;
		// 4 / 5 = slabs of 10 / 12 words, as many waves as words (every wave ONE word of every row through phase B: passes of 64
		// rows), chunks of 1,260 / 1,200 rows.  A workgroup's time is the vector work of its slab on ONE CU (phases A and B are
		// ~61 and ~45 instructions per row and word) plus ~8,000 cycles of latency per chunk: narrower
		// slabs mean less work per workgroup and more workgroups (207 instead of 155 on mk13.b5, on a chip of 256 CUs), thinner
		// rows put more of them into the same LDS (fewer chunks), and the slab must stay wide enough for every workgroup to
		// have a CU of its own -- 8 words would be 310 workgroups.
		const int words = (B.Sm + 1) / 2;
		// (twelve waves = three per SIMD on all four; ten leave two SIMDs with three and two with two, and the fuller ones set
		//  the pace: 2.18 against 2.26 ms on mk13.b5, 2.66 for the 64-byte slabs of round 2)
		if (B.sgn && (1) != 0 && (words + 11) / 12 <= cus)
			shape = 5;
		shape = env_bs("SPASM_HIP_BS_SHAPE", shape);
		if (shape >= 3 && !B.sgn)
			shape = 2;
		B.shape = shape;
		B.ring = (shape == 4) ? 1260 : (shape == 5) ? 1200 : BS_RING;
		B.passrows = (shape >= 3) ? 64 : BS_PASSROWS;
		B.passcap = (shape == 3) ? 31 : (shape == 4) ? 48 : (shape == 5) ? 44 : BS_PASSCAP;
	}
	const int PLAN_RING = B.ring, PLAN_PASSROWS = B.passrows, PLAN_PASSCAP = B.passcap;
	const uint4 PLAN_EMPTY = uint4{(uint32_t) PLAN_RING, (uint32_t) PLAN_RING | ((uint32_t) PLAN_RING << 16), 0u, 0u};
	// compact ids: labels that hold a row, in label (= level) order
	std::vector<int> cid((size_t) (rpad > 0 ? rpad : 1), -1);
	std::vector<int> label_of((size_t) (r > 0 ? r : 1), 0);
	{
		int n = 0;
		for (int c = 0; c < rpad; c++)
			if (P.kof[c] >= 0) {
				cid[c] = n;
				label_of[n] = c;
				n += 1;
			}
		if (n != r)
			die("backsolve_plan: %d labelled rows, %d expected", n, r);
	}
	// level of every compact row (levels are consecutive runs of labelled rows)
	std::vector<int> level((size_t) (r > 0 ? r : 1), 0);
	{
		int n = 0;
		for (int l = 0; l < P.nlevels; l++)
			for (int t = 0; t < P.lvl_count[l]; t++)
				level[n++] = l;
	}
	std::vector<int> colmap((size_t) (m > 0 ? m : 1), 0);
	for (int j = 0; j < m; j++)
		colmap[j] = (P.lab[j] < (uint32_t) rpad) ? cid[P.lab[j]] : r + (int) (P.lab[j] - (uint32_t) rpad);

	uint64_t unmont = 1;                 // 2^-32 mod p
	if (B.plain) {
		const uint64_t R1 = (uint64_t) ((1ull << 32) % (uint64_t) P.prime);
		// inverse by Fermat (p is prime in every use; for a composite odd modulus the extended Euclid below still works)
		int64_t t0 = 0, t1 = 1, r0 = P.prime, r1 = (int64_t) R1;
		while (r1 != 0) {
			const int64_t qq = r0 / r1;
			const int64_t t2 = t0 - qq * t1, r2 = r0 - qq * r1;
			t0 = t1;
			t1 = t2;
			r0 = r1;
			r1 = r2;
		}
		if (r0 != 1)
			die("backsolve_plan: 2^32 is not invertible mod %lld", (long long) P.prime);
		unmont = (uint64_t) ((t0 % P.prime + P.prime) % P.prime);
	}
	// (x mod p for x < 2^64 without a division per entry: q = floor(x * floor(2^64 / p) / 2^64) is the quotient or one less)
	const uint64_t pu64 = (uint64_t) P.prime, barrett = ~0ull / pu64;
	auto mod_p = [&](uint64_t x) -> uint32_t {
		uint64_t rem = x - (uint64_t) (((unsigned __int128) x * barrett) >> 64) * pu64;
		while (rem >= pu64)
			rem -= pu64;
		return (uint32_t) rem;
	};
	auto coeff = [&](uint32_t y_mont) -> uint32_t { return B.plain ? mod_p((uint64_t) y_mont * unmont) : y_mont; };
	auto dep_coeff = [&](uint32_t y_mont) -> uint32_t {
		const uint32_t c = coeff(y_mont);
		if (!B.sgn)
			return c;
		const int64_t bal = ((int64_t) c > P.prime / 2) ? (int64_t) c - P.prime : (int64_t) c;
		return (uint32_t) (int32_t) (-bal);
	};
	double t_mark = wtime();
	auto lap = [&](const char *what) {
		if (verbose() >= 3)
			logmsg("[factor image/back-substitution plan] %s %.2f ms\n", what, 1e3 * (wtime() - t_mark));
		t_mark = wtime();
	};
	// split every row into pivotal dependencies (compact ids) and non-pivotal entries: counted, then filled, by a few threads
	// (the entries of boundary matrices are +-1: their Montgomery forms are recognised, no reduction)
	std::vector<uint64_t> dep_rp((size_t) r + 1, 0), np_rp((size_t) r + 1, 0);
	std::vector<uint2> dep, np;
	std::vector<int> np_row;
	{
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
		dep.resize((size_t) dep_rp[r]);
		np.resize((size_t) np_rp[r]);
		np_row.resize((size_t) np_rp[r]);
		const uint32_t mont_one = (uint32_t) ((1ull << 32) % (uint64_t) P.prime), mont_minus_one = (uint32_t) ((uint64_t) P.prime - mont_one);
		const uint32_t one_c = coeff(mont_one), minus_c = coeff(mont_minus_one), one_d = dep_coeff(mont_one), minus_d = dep_coeff(mont_minus_one);
		for_rows([&](int n_lo, int n_hi) {
			for (int n = n_lo; n < n_hi; n++) {
				const int c = label_of[n];
				uint64_t wd = dep_rp[n], wn = np_rp[n];
				for (uint64_t e = P.rp[c]; e < P.rp[c + 1]; e++) {
					const uint2 en = P.ent[e];
					if (en.x < (uint32_t) rpad) {
						dep[wd++] = uint2{(uint32_t) cid[en.x], en.y == mont_one ? one_d : en.y == mont_minus_one ? minus_d : dep_coeff(en.y)};
					} else {
						np[wn] = uint2{en.x - (uint32_t) rpad, en.y == mont_one ? one_c : en.y == mont_minus_one ? minus_c : coeff(en.y)};
						np_row[wn] = n;
						wn += 1;
					}
				}
			}
		});
	}

	lap("dependencies / non-pivotal entries");
	// chunks, from the last row to the first
	std::vector<BsChunk> chunks;
	std::vector<int> chunk_extra;
	std::vector<uint4> ptab;
	std::vector<uint2> near;
	std::vector<uint4> far_head((size_t) (r > 0 ? r : 1), uint4{BS_NONE, 0u, BS_NONE, 0u});
	std::vector<uint64_t> far_rp((size_t) r + 1, 0);
	std::vector<uint2> far;
	std::vector<uint64_t> far_cnt((size_t) (r > 0 ? r : 1), 0);
	// first pass: chunk boundaries (a chunk grows downwards while its rows, the passes of its rows that have
	// dependencies inside the chunk -- 32 rows of one level each -- and the dependencies beyond the second fit the LDS
	// arrays of the kernel)
	int hi = r;
	while (hi > 0) {
		BsChunk ch{};
		ch.hi = hi;
		int lo = hi, nnear = 0, npass = 0, in_level = 0, last_level = -1;
		while (lo > 0 && hi - lo < PLAN_RING) {
			const int c = lo - 1;
			int nc = 0;
			for (uint64_t e = dep_rp[c]; e < dep_rp[c + 1]; e++)
				nc += dep[e].x < (uint32_t) hi;
			const bool new_pass = nc > 0 && (level[c] != last_level || in_level % PLAN_PASSROWS == 0);
			const int extra = nc > 2 ? nc - 1 : 0;
			if (nnear + extra > BS_NEARCAP || nc > 65535 || (new_pass && npass + 1 > PLAN_PASSCAP))
				break;                           // (the first row of a chunk never has dependencies inside it)
			if (nc > 0) {
				if (level[c] != last_level)
					in_level = 0;
				npass += new_pass ? 1 : 0;
				in_level += 1;
				last_level = level[c];
				nnear += extra;
			}
			lo = c;
		}
		ch.lo = lo;
		ch.npass = npass;
		ch.nnear = nnear;
		chunks.push_back(ch);
		chunk_extra.push_back(0);
		hi = lo;
	}
	// second pass: the tables of every chunk, slots relative to its first row.  The first pass has counted the passes and the list
	// entries of every chunk: the tables are sized at once (empty table slots are what the padding of a pass is) and the chunks
	// filled by a few threads, each at its own place (4.1 of the 7.2 ms of mk13.b5's image went into this loop and its push_backs).
	{
		int64_t total_pass = 0, total_near = 0;
		for (BsChunk &ch : chunks) {
			ch.pass0 = (int) total_pass;
			ch.near0 = (int) total_near;
			total_pass += ch.npass;
			total_near += ch.nnear;
		}
		ptab.assign((size_t) total_pass * (size_t) PLAN_PASSROWS, PLAN_EMPTY);
		near.assign((size_t) total_near, uint2{0u, 0u});
		auto fill_chunk = [&](size_t k) {
			BsChunk &ch = chunks[k];
			size_t pt = (size_t) ch.pass0 * (size_t) PLAN_PASSROWS;          // next slot of the table of passes
			int nn = 0;                                                       // list entries of the chunk so far
			int last_level = -1, extra = 0, in_level = 0;
			for (int c = ch.hi - 1; c >= ch.lo; c--) {
				int nc = 0, nf = 0;
				for (uint64_t e = dep_rp[c]; e < dep_rp[c + 1]; e++) {
					if (dep[e].x < (uint32_t) ch.hi) {
						nc += 1;
					} else {
						if (nf == 0) {
							far_head[c].x = dep[e].x;
							far_head[c].y = dep[e].y;
						} else if (nf == 1) {
							far_head[c].z = dep[e].x;
							far_head[c].w = dep[e].y;
						} else {
							far_cnt[c] += 1;
							extra = 1;
						}
						nf += 1;
					}
				}
				if (nc == 0)
					continue;
				if (level[c] != last_level || in_level % PLAN_PASSROWS == 0) {
					// a new pass: the rest of the previous one stays empty
					pt = (pt + (size_t) PLAN_PASSROWS - 1) / (size_t) PLAN_PASSROWS * (size_t) PLAN_PASSROWS;
					if (level[c] != last_level)
						in_level = 0;
					last_level = level[c];
				}
				in_level += 1;
				uint4 en{(uint32_t) (c - ch.lo) | ((uint32_t) nc << 16), 0u, 0u, 0u};
				int seen = 0;
				for (uint64_t e = dep_rp[c]; e < dep_rp[c + 1]; e++) {
					if (dep[e].x >= (uint32_t) ch.hi)
						continue;
					const uint32_t slot = dep[e].x - (uint32_t) ch.lo;
					if (seen == 0) {
						en.y = slot | (slot << 16);          // (one dependency: the second slot repeats it with coefficient 0)
						en.z = dep[e].y;
						if (nc > 2)
							en.w = (uint32_t) nn;
					} else if (nc == 2) {
						en.y = (en.y & 0xFFFFu) | (slot << 16);
						en.w = dep[e].y;
					} else {
						if (nn >= ch.nnear)
							die("backsolve_plan: chunk %zu holds more list entries than the %d counted", k, ch.nnear);
						near[(size_t) ch.near0 + (size_t) nn] = uint2{slot, dep[e].y};
						nn += 1;
					}
					seen += 1;
				}
				if (pt >= (size_t) (ch.pass0 + ch.npass) * (size_t) PLAN_PASSROWS)
					die("backsolve_plan: chunk %zu needs more than the %d passes counted", k, ch.npass);
				ptab[pt++] = en;
			}
			const int passes = (int) ((pt + (size_t) PLAN_PASSROWS - 1) / (size_t) PLAN_PASSROWS) - ch.pass0;
			if (passes != ch.npass || nn != ch.nnear)
				die("backsolve_plan: chunk %zu was counted differently on the second pass (%d passes against %d, %d list entries against %d)", k, passes, ch.npass, nn,
				    ch.nnear);
			chunk_extra[k] = extra;
			// everything the kernel needs to start a chunk sits in its descriptor (no dependent loads at the top of a chunk)
			ch.np0 = (int) std::min<uint64_t>(np_rp[ch.lo], 0x7FFFFFFFull);
			ch.npn = (int) std::min<uint64_t>(np_rp[ch.hi] - np_rp[ch.lo], 0x7FFFFFFFull) | (extra ? (int) 0x80000000u : 0);
		};
		sh::pool_run((int) chunks.size(), [&](int k) { fill_chunk((size_t) k); });
	}
	if (np.size() >= 0x7FFFFFFFull)
		die("backsolve_plan: %zu non-pivotal entries in the factor (the chunk descriptors index them with 31 bits)", np.size());
	// dependencies beyond the first two outside the chunk, CSR by row
	for (int c = 0; c < r; c++)
		far_rp[c + 1] = far_rp[c] + far_cnt[c];
	far.assign((size_t) (far_rp[r] > 0 ? far_rp[r] : 1), uint2{0, 0});
	{
		// chunk of a row: chunks are stored from the last rows to the first
		std::vector<int> chunk_hi((size_t) (r > 0 ? r : 1), 0);
		for (size_t k = 0; k < chunks.size(); k++)
			for (int c = chunks[k].lo; c < chunks[k].hi; c++)
				chunk_hi[c] = chunks[k].hi;
		for (int c = 0; c < r; c++) {
			if (far_cnt[c] == 0)
				continue;
			uint64_t w = far_rp[c];
			int nf = 0;
			for (uint64_t e = dep_rp[c]; e < dep_rp[c + 1]; e++)
				if (dep[e].x >= (uint32_t) chunk_hi[c]) {
					if (nf >= 2)
						far[w++] = dep[e];
					nf += 1;
				}
		}
	}

	if ((0)) {
		// shape of the plan (tuning aid)
		int64_t cnt_hist[6] = {0, 0, 0, 0, 0, 0}, empty = 0;
		for (size_t t = 0; t < ptab.size(); t++) {
			const int c = (int) (ptab[t].x >> 16);
			cnt_hist[c < 5 ? c : 5] += 1;
			empty += c == 0;
		}
		int64_t far2 = 0, far1 = 0;
		for (int c = 0; c < r; c++) {
			far1 += far_head[c].x != BS_NONE;
			far2 += far_head[c].z != BS_NONE;
		}
		fprintf(stderr, "[bs plan] r %d, Sm %d, levels %d, chunks %zu, passes %zu (%lld empty slots), rows with dependencies inside their chunk: 1: %lld, 2: %lld, 3: %lld, "
		        "4: %lld, 5+: %lld; list %zu, far heads %lld + %lld, far rest %llu, np %zu\n",
		        r, B.Sm, P.nlevels, chunks.size(), ptab.size() / PLAN_PASSROWS, (long long) empty, (long long) cnt_hist[1], (long long) cnt_hist[2], (long long) cnt_hist[3],
		        (long long) cnt_hist[4], (long long) cnt_hist[5], near.size(), (long long) far1, (long long) far2, (unsigned long long) far_rp[r], np.size());
	}
	lap("chunks, passes, near and far tables");
	B.nchunks = (int) chunks.size();
	B.nnear = (int64_t) near.size();
	B.nfar = (int64_t) far_rp[r];
	B.nnp = (int64_t) np.size();
	B.ndeps = (int64_t) dep.size();
	B.d_col = dalloc<int>(m);
	B.d_chunk = dalloc<BsChunk>((int64_t) chunks.size());
	B.d_ptab = dalloc<uint4>((int64_t) ptab.size());
	B.d_near = dalloc<uint2>((int64_t) near.size());
	B.d_far_head = dalloc<uint4>(r);
	B.d_far_rp = dalloc<uint64_t>((int64_t) r + 1);
	B.d_far = dalloc<uint2>((int64_t) far.size());
	B.d_np_rp = dalloc<uint64_t>((int64_t) r + 1);
	B.d_np = dalloc<uint2>((int64_t) np.size());
	B.d_np_row = dalloc<int>((int64_t) np_row.size());
	B.d_chunk_extra = dalloc<int>((int64_t) chunk_extra.size());
	lap("device buffers");
	upload(B.d_col, colmap, stream);
	upload(B.d_chunk, chunks, stream);
	upload(B.d_chunk_extra, chunk_extra, stream);
	upload(B.d_ptab, ptab, stream);
	upload(B.d_near, near, stream);
	upload(B.d_far_head, far_head, stream);
	upload(B.d_far_rp, far_rp, stream);
	upload(B.d_far, far, stream);
	upload(B.d_np_rp, np_rp, stream);
	upload(B.d_np, np, stream);
	upload(B.d_np_row, np_row, stream);
	HIP_CHECK(hipStreamSynchronize(stream));      // the host vectors die here
	lap("uploads");
	B.planned = true;
	B.valid = false;
}

void backsolve_free(spasm_hip_dfact *F)
{
	BsImage &B = F->bs;
	big_free(B.d_R);
	sh::big_free(B.d_col);
	sh::big_free(B.d_chunk);
	sh::big_free(B.d_chunk_extra);
	sh::big_free(B.d_ptab);
	sh::big_free(B.d_near);
	sh::big_free(B.d_far_head);
	sh::big_free(B.d_far_rp);
	sh::big_free(B.d_far);
	sh::big_free(B.d_np_rp);
	sh::big_free(B.d_np);
	sh::big_free(B.d_np_row);
	if (B.ev0 != nullptr)
		(void) hipEventDestroy(B.ev0);
	if (B.ev1 != nullptr)
		(void) hipEventDestroy(B.ev1);
	B = BsImage{};
}

// (re)computes R on `stream`.  The caller synchronises before reading the events.
void backsolve_build(const spasm_hip_dfact *F, hipStream_t stream)
{
	BsImage &B = F->bs;
	if (!B.planned && F->bs_deferred && F->host_plan) {
		// (a factor with the tables of the sparse image: the plan of the dense one was left for the batch that wants it)
		backsolve_plan(*F->host_plan, const_cast<spasm_hip_dfact *>(F), stream);
		F->bs_deferred = false;
	}
	if (!B.planned)
		die("backsolve_build: the factor has no back-substitution plan");
	// R is stored in 16 bits when the prime allows (42013, the reference's default, does): half the traffic, half the LDS
	const bool packed = B.sgn || (F->prime < 65536 && true);
	const int elem = packed ? 2 : 4;
	const size_t bytes = (size_t) B.r * (size_t) B.ldR * (size_t) elem;
	if (B.d_R != nullptr && B.elem_bytes != elem) {
		big_free(B.d_R);
		B.d_R = nullptr;
	}
	bool fresh = false;
	if (B.d_R == nullptr) {
		B.d_R = big_alloc(bytes);          // (from the block cache: tens of GB on a wide factor)
		B.elem_bytes = elem;
		fresh = true;
	}
	if (B.ev0 == nullptr) {
		HIP_CHECK(hipEventCreate(&B.ev0));
		HIP_CHECK(hipEventCreate(&B.ev1));
	}
	BsArgs b{};
	b.R = B.d_R;
	b.ldR = B.ldR;
	b.nchunks = B.nchunks;
	b.chunk = B.d_chunk;
	b.chunk_extra = B.d_chunk_extra;
	b.ptab = B.d_ptab;
	b.near = B.d_near;
	b.far_head = B.d_far_head;
	b.far_rp = B.d_far_rp;
	b.far = B.d_far;
	b.np_rp = B.d_np_rp;
	b.np = B.d_np;
	b.np_row = B.d_np_row;
	b.r = B.r;
	b.plain = B.plain ? 1 : 0;
	b.sgn = B.sgn ? 1 : 0;
	b.G = sgn_setup(F->prime);
	b.F = to_dev(F->mont);
	b.dbg = (0);
	b.prof = nullptr;
	if (env_bs("SPASM_HIP_BS_PROFILE", 0)) {
		b.prof = dalloc<unsigned long long>(8);
		HIP_CHECK(hipMemsetAsync(b.prof, 0, 8 * sizeof(unsigned long long), stream));
	}
	// few non-pivotal entries per row of U (mk13.b5: 0.08): the kernel scatters them into its LDS ring itself and R is
	// neither zeroed nor read for them; many (factors that already hold dense rows): R is pre-filled instead
	b.sparse_init = (B.nnp <= 4 * (int64_t) B.r && env_bs("SPASM_HIP_BS_SPARSE_INIT", 1) != 0) ? 1 : 0;
	HIP_CHECK(hipEventRecord(B.ev0, stream));
	if (fresh || !b.sparse_init)        // (the padding columns beyond the last slab must be zero: once is enough for them)
		HIP_CHECK(hipMemsetAsync(B.d_R, 0, bytes, stream));
	if (!b.sparse_init) {
		if (packed)
			hipLaunchKernelGGL(bs_init_kernel<uint16_t>, dim3((B.r + 255) / 256), dim3(256), 0, stream, b);
		else
			hipLaunchKernelGGL(bs_init_kernel<uint32_t>, dim3((B.r + 255) / 256), dim3(256), 0, stream, b);
	}
	// shape of a workgroup: 0 = 128-byte slab rows, 16 waves; 1 = 128 B, 8 waves; 2 = 64 B, 8 waves.  The chain of levels
	// (phase B) is bound by instruction issue -- every wave runs every step -- so the small workgroup wins while all its
	// slabs are resident at once (mk13.b5: 4.4 against 4.9 ms); with more slabs than that, whole cache lines per request
	// matter more (phase A is bound by the rate of memory requests)
	const int shape = B.shape;
	if (shape == 3)
		snprintf(B.kernel_build, sizeof(B.kernel_build), "backsolve_kernel<true,true,8,8,true,768,64,31,2>");
	else if (shape == 4)
		snprintf(B.kernel_build, sizeof(B.kernel_build), "backsolve_kernel<true,true,10,10,true,1260,64,48,1>");
	else if (shape == 5)
		snprintf(B.kernel_build, sizeof(B.kernel_build), "backsolve_kernel<true,true,12,12,true,1200,64,44,1>");
	else
		snprintf(B.kernel_build, sizeof(B.kernel_build), "backsolve_kernel<%s,%s,%d,%d,%s,%d,%d,%d,1>", packed ? "true" : "false", B.plain ? "true" : "false",
		         shape == 2 ? 16 : 32, (shape == 0 || (!packed && B.plain && shape == 1)) ? 16 : 8, B.sgn ? "true" : "false", BS_RING, BS_PASSROWS,
		         BS_PASSCAP);          // (as rocprofv3 prints it: bench.py looks the counters of the profile passes up by this name)
	if (B.sgn) {
		if (shape == 3)
			launch_backsolve_variant<true, true, 8, 8, true, 768, 64, 31, 2>(b, B.Sm, stream, B);
		else if (shape == 4)
			launch_backsolve_variant<true, true, 10, 10, true, 1260, 64, 48, 1>(b, B.Sm, stream, B);
		else if (shape == 5)
			launch_backsolve_variant<true, true, 12, 12, true, 1200, 64, 44, 1>(b, B.Sm, stream, B);
		else if (shape == 1)
			launch_backsolve_variant<true, true, 32, 8, true>(b, B.Sm, stream, B);
		else if (shape == 2)
			launch_backsolve_variant<true, true, 16, 8, true>(b, B.Sm, stream, B);
		else
			launch_backsolve_variant<true, true, 32, 16, true>(b, B.Sm, stream, B);
	} else if (packed) {
		if (shape == 1)
			launch_backsolve_variant<true, true, 32, 8>(b, B.Sm, stream, B);
		else if (shape == 2)
			launch_backsolve_variant<true, true, 16, 8>(b, B.Sm, stream, B);
		else
			launch_backsolve_variant<true, true, 32, 16>(b, B.Sm, stream, B);
	} else if (B.plain) {          // (p < 2^16 with 32-bit entries of R: the SPASM_HIP_BS_PACKED=0 knob)
		if (shape == 2)
			launch_backsolve_variant<false, true, 16, 8>(b, B.Sm, stream, B);
		else
			launch_backsolve_variant<false, true, 32, 16>(b, B.Sm, stream, B);
	} else {
		if (shape == 1)
			launch_backsolve_variant<false, false, 32, 8>(b, B.Sm, stream, B);
		else if (shape == 2)
			launch_backsolve_variant<false, false, 16, 8>(b, B.Sm, stream, B);
		else
			launch_backsolve_variant<false, false, 32, 16>(b, B.Sm, stream, B);
	}
	HIP_CHECK(hipGetLastError());
	HIP_CHECK(hipEventRecord(B.ev1, stream));
	B.valid = true;
	B.builds += 1;
	if (b.prof != nullptr) {
		unsigned long long h[8];
		HIP_CHECK(hipMemcpyAsync(h, b.prof, sizeof(h), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		static const char *const stage[7] = {"descriptor + metadata issue", "row start (zero + scatter)", "phase A", "long outside lists", "phase B", "phase C",
		                                     "metadata into LDS"};
		unsigned long long tot = 0;
		for (int q = 0; q < 7; q++)
			tot += h[q];
		fprintf(stderr, "[bs profile] %s, %d chunks, workgroup 0, shader-clock cycles per chunk:", B.kernel_build, B.nchunks);
		for (int q = 0; q < 7; q++)
			fprintf(stderr, " %s %.0f (%.0f%%);", stage[q], (double) h[q] / B.nchunks, 100.0 * (double) h[q] / (double) (tot ? tot : 1));
		fprintf(stderr, " total %.0f\n", (double) tot / B.nchunks);
		sh::big_free(b.prof);
	}
	if ((0) && b.sparse_init && bytes < ((size_t) 1 << 30)) {
		// debugging aid: the same build with R pre-filled (the other way of starting the rows), compared entry by entry
		void *R2 = nullptr;
		HIP_CHECK(sh::malloc_or_trim(&R2, bytes));
		BsArgs c = b;
		c.R = R2;
		c.sparse_init = 0;
		HIP_CHECK(hipMemsetAsync(R2, 0, bytes, stream));
		if (packed)
			hipLaunchKernelGGL(bs_init_kernel<uint16_t>, dim3((B.r + 255) / 256), dim3(256), 0, stream, c);
		else
			hipLaunchKernelGGL(bs_init_kernel<uint32_t>, dim3((B.r + 255) / 256), dim3(256), 0, stream, c);
		if (B.sgn)
			launch_backsolve_variant<true, true, 32, 16, true>(c, B.Sm, stream, B);
		else if (packed)
			launch_backsolve_variant<true, true, 32, 16>(c, B.Sm, stream, B);
		else if (B.plain)
			launch_backsolve_variant<false, true, 32, 16>(c, B.Sm, stream, B);
		else
			launch_backsolve_variant<false, false, 32, 16>(c, B.Sm, stream, B);
		std::vector<unsigned char> h1(bytes), h2(bytes);
		HIP_CHECK(hipMemcpyAsync(h1.data(), B.d_R, bytes, hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipMemcpyAsync(h2.data(), R2, bytes, hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		int shown = 0;
		int64_t bad = 0;
		for (int64_t t = 0; t < (int64_t) B.r * B.ldR; t++) {
			const uint32_t x1 = packed ? ((uint16_t *) h1.data())[t] : ((uint32_t *) h1.data())[t];
			const uint32_t x2 = packed ? ((uint16_t *) h2.data())[t] : ((uint32_t *) h2.data())[t];
			if (x1 != x2) {
				bad += 1;
				if (shown++ < 12)
					fprintf(stderr, "[bs check] row %lld col %lld: sparse-init %u, pre-filled %u (r %d, Sm %d, ldR %lld, chunks %d)\n",
					        (long long) (t / B.ldR), (long long) (t % B.ldR), x1, x2, B.r, B.Sm, (long long) B.ldR, B.nchunks);
			}
		}
		fprintf(stderr, "[bs check] %lld entries differ\n", (long long) bad);
		sh::big_free(R2);
	}
}

// staged sparse output available for this factor?  *row_bytes = size of one packed row of the staging buffer
bool backsolve_stages_output(const spasm_hip_dfact *F, int64_t *row_bytes)
{
	const BsImage &B = F->bs;
	*row_bytes = B.ldR * (B.valid ? B.elem_bytes : 4);          // (asked after the build: the entry size is known)
	return B.planned && env_bs("SPASM_HIP_BS_STAGED", 1) != 0;
}

// S rows from R: sparse rows into the pool of `a` (dense_out == nullptr) or dense rows.
void launch_backsolve_apply(const SchurArgs &a, const spasm_hip_dfact *F, uint32_t *dense_out, int64_t ldS, hipStream_t stream,
                            BsDirectOut *direct)
{
	const BsImage &B = F->bs;
	if (!B.valid)
		die("launch_backsolve_apply: R has not been built");
	ApplyArgs d{};
	d.a = a;
	d.R = B.d_R;
	d.ldR = B.ldR;
	d.col = B.d_col;
	d.r = B.r;
	static_assert(2 * 64 * AP_TU == 512, "ldR is padded to whole tile groups of the apply kernels");
	d.Smpad = (int) B.ldR;
	d.dense_out = dense_out;
	d.ldS = ldS;
	d.dbg = (0);
	d.sgn = B.sgn ? 1 : 0;
	d.G = sgn_setup(F->prime);
	if (direct != nullptr && dense_out == nullptr) {
		d.direct = 1;
		d.status = direct->status;
		d.ticket = direct->ticket;
		d.Sp = direct->Sp;
		d.Sj = direct->Sj;
		d.Sx = direct->Sx;
		d.cap = direct->cap;
	}
	const bool packed = B.elem_bytes == 2;
	int dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	hipDeviceProp_t prop;
	HIP_CHECK(hipGetDeviceProperties(&prop, dev));
	// few long rows asked for as dense rows (the completion test: combinations of every remaining row): workgroups split
	// the columns AND the entries of a row; otherwise one wave per row
	if (dense_out != nullptr && a.nrows > 0 && a.nrows < prop.multiProcessorCount && a.avg_row_entries >= 1024) {
		const int nwords = d.Smpad / (packed ? 2 : 1);
		const dim3 grid((unsigned) ((nwords + 64 * AP_TU - 1) / (64 * AP_TU)), (unsigned) a.nrows);
		if (packed)
			hipLaunchKernelGGL((bs_apply_wide_kernel<true, true>), grid, dim3(64 * AW_NW), 0, stream, d);
		else if (B.plain)
			hipLaunchKernelGGL((bs_apply_wide_kernel<false, true>), grid, dim3(64 * AW_NW), 0, stream, d);
		else
			hipLaunchKernelGGL((bs_apply_wide_kernel<false, false>), grid, dim3(64 * AW_NW), 0, stream, d);
		HIP_CHECK(hipGetLastError());
		return;
	}
	// rows of more than 24,576 columns are produced in segments of 8,192 columns (signed 16-bit entries, staged or dense
	// output): 16.5 KB of LDS per wave, eight waves per CU, where the whole row would leave one wave per CU
	const bool staged_call = direct != nullptr && direct->stage != nullptr && dense_out == nullptr;
	int seg_cols = d.Smpad;
	if (B.sgn && d.Smpad > 24576 && (staged_call || dense_out != nullptr))
		seg_cols = 8192;
	d.seg_words = seg_cols / 2;
	const size_t per_wave = ((size_t) seg_cols * (size_t) B.elem_bytes + (size_t) AP_LIST * sizeof(uint2) + 15) / 16 * 16;
	if (per_wave > 150 * 1024)
		die("launch_backsolve_apply: %d non-pivotal columns do not fit the LDS row buffer", B.Sm);
	// as many waves as fit half of a CU's LDS (two workgroups per CU), at most 8, at least 1
	const int waves = (int) std::max<size_t>(1, std::min<size_t>(8, (size_t) (76 * 1024) / per_wave));
	d.waves = waves;
	d.wave_bytes = per_wave;
	const size_t lds = per_wave * (size_t) waves;
	int blocks = std::max(1, std::min((a.nrows + waves - 1) / waves, prop.multiProcessorCount * 8));
	if (d.direct) {
		// look-back output: a row waits for rows held by OTHER workgroups, so every workgroup of the grid must be resident
		// (a workgroup that has not started cannot publish the lengths its ticket class owes): at most what the LDS of the
		// chip holds at once
		const int per_cu = (int) std::max<size_t>(1, (size_t) (160 * 1024) / std::max<size_t>(lds, 1));
		blocks = std::min(blocks, prop.multiProcessorCount * per_cu);
	}
	d.ntickets = std::min(LB_TICKETS, blocks);
	// one launch of the apply kernel that fits the arithmetic of R
	auto launch_apply = [&](const ApplyArgs &dd, int nblocks_apply) {
		if (B.sgn) {
			static size_t configured = 0;
			if (lds > configured) {
				HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&bs_apply_s16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
				configured = lds;
			}
			hipLaunchKernelGGL(bs_apply_s16_kernel, dim3(nblocks_apply), dim3(64 * dd.waves), lds, stream, dd);
		} else if (packed) {
			launch_apply_variant<true, true>(dd, nblocks_apply, lds, stream);
		} else if (B.plain) {
			launch_apply_variant<false, true>(dd, nblocks_apply, lds, stream);
		} else {
			launch_apply_variant<false, false>(dd, nblocks_apply, lds, stream);
		}
	};
	if (direct != nullptr && direct->stage != nullptr && dense_out == nullptr) {
		// staged output: rows as they stand + lengths, offsets by a scan, sparse rows by a streaming kernel; slices of
		// stage_rows rows when the staging buffer is smaller than the batch
		const int nwords = d.Smpad / (packed ? 2 : 1);
		HIP_CHECK(hipMemsetAsync(direct->Sp, 0, sizeof(int64_t), stream));
		auto expand_slice = [&](int64_t r0, int n, const uint32_t *stage, const unsigned long long *block_sum, hipStream_t s2) {
			const int nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
			hipLaunchKernelGGL(bs_scan_lengths_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, s2, a.row_len + r0, n, block_sum, direct->Sp + r0, direct->cap, a.ctr);
			ExpandArgs e{stage, nwords, n, direct->Sp + r0, direct->Sj, direct->Sx, direct->cap, a.q, d.G, d.dbg};
			const int blocks3 = std::max(1, std::min((n + 3) / 4, prop.multiProcessorCount * 8));
			if (B.sgn)
				hipLaunchKernelGGL(bs_expand_kernel<0>, dim3(blocks3), dim3(256), 0, s2, e);
			else if (packed)
				hipLaunchKernelGGL(bs_expand_kernel<1>, dim3(blocks3), dim3(256), 0, s2, e);
			else
				hipLaunchKernelGGL(bs_expand_kernel<2>, dim3(blocks3), dim3(256), 0, s2, e);
		};
		// (round 5, measured and dropped: the batch cut into 2 / 4 / 8 parts, each with its own part of the staging buffer, the parts
		//  applied one after the other on the caller's stream while a second stream scanned and expanded each part as soon as it was
		//  there -- expansion of part k beside the apply of part k + 1, two kernels bound by the memory at 4.5 and 4.1 TB/s of the
		//  8 the device has.  The step of mk13.b5 went from 4.63 ms to 4.74 / 5.19 / 6.30: two workgroup shapes that both want
		//  the LDS and the memory queues of the same CUs take more from each other than the overlap gives back.)
		for (int64_t r0 = 0; r0 < a.nrows; r0 += direct->stage_rows) {
			const int n = (int) std::min<int64_t>(direct->stage_rows, a.nrows - r0);
			ApplyArgs d2 = d;
			d2.direct = 0;
			d2.stage = direct->stage;
			d2.a.rows = a.rows + r0;
			d2.a.row_len = a.row_len + r0;
			d2.a.nrows = n;
			const int blocks2 = std::max(1, std::min((n + waves - 1) / waves, prop.multiProcessorCount * 8));
			const int nblocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
			d2.block_sum = direct->status;          // (the look-back words are not used by the staged output)
			HIP_CHECK(hipMemsetAsync(d2.block_sum, 0, (size_t) nblocks * sizeof(unsigned long long), stream));
			launch_apply(d2, blocks2);
			if (direct->ev_expand != nullptr)
				HIP_CHECK(hipEventRecord(direct->ev_expand, stream));
			expand_slice(r0, n, direct->stage, d2.block_sum, stream);
			direct->staged = true;
			direct->slices += 1;
		}
	} else {
		launch_apply(d, blocks);
	}
	HIP_CHECK(hipGetLastError());
}

}  // namespace sh
