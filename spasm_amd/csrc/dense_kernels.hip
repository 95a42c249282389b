// Dense reduced row echelon form mod p on the GPU (replaces the FFLAS-FFPACK
// calls behind spasm_ffpack_rref, spasm_ffpack.cpp:23-49).
//
// Blocked Gauss-Jordan with column-rank-profile pivots, rows stay in place:
//   for each panel of NB columns
//     1. panel step: k <= NB new pivots (rows rho, columns gamma) and the n x k multipliers M such that the
//        composite row transformation of the panel is T = I + M e_rho^T.  Default: a tournament over the free
//        rows + one Gauss-Jordan of a 64 x 128 block (see "Tournament panel step" below); the older kernels
//        eliminate column by column (one workgroup, or several with a grid barrier per column);
//     2. gather B = A[rho, rest] (the k new pivot rows, old values);
//     3. trailing update A[:, rest] += M B  (GEMM mod p), deferred over super-panels of four panels (K <= 256).
//        For p < 2^16 the GEMM runs on the matrix cores: operands are split into two signed base-256 digits and
//        multiplied with v_mfma_i32_32x32x32_i8 (four digit products, i32 accumulators, recombined mod p).
//        Larger primes use a tiled 64-bit VALU kernel.
// Values are canonical representatives in [0, p) stored as u32.
#include <algorithm>
#include <cstring>
#include <type_traits>
#include "device_types.h"
#include "field_dev.h"

namespace sh {

namespace {

constexpr int NB = 64;                  // panel width
constexpr int PW = 2 * NB;              // panel + selector columns
constexpr int PANEL_THREADS = 1024;

__device__ uint32_t invmod(uint32_t a, const MontDev &F)
{
	// Fermat: a^(p-2)
	uint32_t result = 1 % F.p, base = a;
	uint32_t e = F.p - 2;
	while (e) {
		if (e & 1)
			result = mulmod(result, base, F);
		base = mulmod(base, base, F);
		e >>= 1;
	}
	return result;
}

}  // namespace

struct PanelArgs {
	uint32_t *A;          // n x m, leading dimension ld
	int64_t ld;
	int n, m;
	int c0, width;        // panel columns [c0, c0 + width)
	uint32_t *P;          // workspace n x PW: panel copy | selector columns
	int *is_pivot_row;    // n flags
	int *pivrow;          // global list: pivot row of the t-th pivot
	int *pivcol;          // ... and its column
	int *rank;            // running rank (device scalar)
	int *knew;            // out: pivots found in this panel
	int *rho;             // out: their rows (NB)
	MontDev F;
};

// One workgroup.  Thread t owns rows t, t + 1024, ...  The panel copy P is COLUMN-major
// (P[c * n + i]) so that the threads of a wave touch consecutive addresses in every loop.
__global__ __launch_bounds__(PANEL_THREADS) void rref_panel_kernel(PanelArgs g)
{
	__shared__ uint32_t piv[PW];
	__shared__ int s_row;
	__shared__ int s_k;
	const int tid = threadIdx.x;
	const MontDev F = g.F;
	const int n = g.n, W = g.width;
	const int64_t nn = n;
	// copy the panel in (transposing), clear the selector columns
	for (int64_t t = tid; t < nn * W; t += PANEL_THREADS) {
		const int64_t i = t / W;
		const int c = (int) (t % W);
		g.P[(int64_t) c * nn + i] = g.A[i * g.ld + g.c0 + c];
	}
	for (int64_t t = tid; t < nn * NB; t += PANEL_THREADS)
		g.P[(int64_t) NB * nn + t] = 0;
	if (tid == 0)
		s_k = 0;
	__syncthreads();
	for (int c = 0; c < W; c++) {
		// lowest free row with a non-zero in column c
		if (tid == 0)
			s_row = 0x7FFFFFFF;
		__syncthreads();
		const uint32_t *Pc = g.P + (int64_t) c * nn;
		int mine = 0x7FFFFFFF;
		for (int i = tid; i < n; i += PANEL_THREADS)
			if (Pc[i] != 0 && !g.is_pivot_row[i]) {
				mine = i;
				break;
			}
		if (mine != 0x7FFFFFFF)
			atomicMin(&s_row, mine);
		__syncthreads();
		const int rho = s_row;
		if (rho == 0x7FFFFFFF)
			continue;
		const int k = s_k;
		// the scaled pivot row: panel columns >= c and the selector columns 0..k (k = its own, value 1)
		{
			const uint32_t inv = invmod(Pc[rho], F);
			for (int cc = tid; cc < PW; cc += PANEL_THREADS) {
				uint32_t v = 0;
				if (cc >= c && cc < W)
					v = g.P[(int64_t) cc * nn + rho];
				else if (cc >= NB && cc < NB + k)
					v = g.P[(int64_t) cc * nn + rho];
				else if (cc == NB + k)
					v = 1;
				piv[cc] = (v != 0) ? mulmod(v, inv, F) : 0u;
			}
		}
		__syncthreads();
		// eliminate column c from every other row; the pivot row itself becomes piv[]
		for (int i = tid; i < n; i += PANEL_THREADS) {
			if (i == rho) {
				for (int cc = c; cc < W; cc++)
					g.P[(int64_t) cc * nn + i] = piv[cc];
				for (int cc = NB; cc <= NB + k; cc++)
					g.P[(int64_t) cc * nn + i] = piv[cc];
				continue;
			}
			const uint32_t f = Pc[i];
			if (f == 0)
				continue;
			for (int cc = c; cc < W; cc++) {
				if (piv[cc] == 0)
					continue;
				uint32_t *x = g.P + (int64_t) cc * nn + i;
				*x = submod(*x, mulmod(f, piv[cc], F), F);
			}
			for (int cc = NB; cc <= NB + k; cc++) {
				if (piv[cc] == 0)
					continue;
				uint32_t *x = g.P + (int64_t) cc * nn + i;
				*x = submod(*x, mulmod(f, piv[cc], F), F);
			}
		}
		if (tid == 0) {
			g.is_pivot_row[rho] = 1;
			const int rk = *g.rank;
			g.pivrow[rk] = rho;
			g.pivcol[rk] = g.c0 + c;
			*g.rank = rk + 1;
			g.rho[k] = rho;
			s_k = k + 1;
		}
		__syncthreads();
	}
	// write the eliminated panel back; M = T J - J: subtract the selector ones
	const int k = s_k;
	for (int64_t t = tid; t < nn * W; t += PANEL_THREADS) {
		const int64_t i = t / W;
		const int c = (int) (t % W);
		g.A[i * g.ld + g.c0 + c] = g.P[(int64_t) c * nn + i];
	}
	__syncthreads();
	if (tid < k) {
		uint32_t *Pr = g.P + (int64_t) (NB + tid) * nn + g.rho[tid];
		*Pr = submod(*Pr, 1u, F);
	}
	if (tid == 0)
		*g.knew = k;
}

// ---- the same panel step spread over several workgroups (tall blocks: one CU cannot stream the
// panel fast enough).  Row i belongs to workgroup (i / COOP_THREADS) % gridDim.x.  Two grid-wide
// barriers per pivot: after the candidate search (everybody then reads the pivot row) and before
// the elimination (the owner overwrites the pivot row with its scaled image).  The barrier is a
// monotonic counter with the agent-scope release / acquire recipe of the CDNA guide (Guideline 16);
// spins are bounded: on a timeout the error flag is raised and every later barrier falls through.
constexpr int COOP_THREADS = 256;

__device__ __forceinline__ void grid_barrier(unsigned int *counter, unsigned int target, int *err)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		long spins = 0;
		while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
			__builtin_amdgcn_s_sleep(2);
			if (++spins > 20000000L || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
				__hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				break;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	__syncthreads();
}

struct CoopPanelArgs {
	PanelArgs g;
	unsigned int *barrier;    // zeroed before the launch
	int *cand;                // NB ints, 0x7FFFFFFF before the launch
	int *err;
};

__global__ __launch_bounds__(COOP_THREADS) void rref_panel_coop_kernel(CoopPanelArgs ca)
{
	const PanelArgs &g = ca.g;
	__shared__ uint32_t piv[PW];
	__shared__ int s_min;
	const int tid = threadIdx.x;
	const MontDev F = g.F;
	const int n = g.n, W = g.width;
	const int64_t nn = n;
	const int G = gridDim.x;
	const int stride = G * COOP_THREADS;
	const int first = blockIdx.x * COOP_THREADS + tid;          // rows first, first + stride, ...
	unsigned int phase = 0;
	// copy the panel in (each workgroup its own rows), clear the selector columns
	for (int i = first; i < n; i += stride) {
		for (int c = 0; c < W; c++)
			g.P[(int64_t) c * nn + i] = g.A[(int64_t) i * g.ld + g.c0 + c];
		for (int c = NB; c < PW; c++)
			g.P[(int64_t) c * nn + i] = 0;
	}
	int k = 0;
	for (int c = 0; c < W; c++) {
		if (tid == 0)
			s_min = 0x7FFFFFFF;
		__syncthreads();
		const uint32_t *Pc = g.P + (int64_t) c * nn;
		int mine = 0x7FFFFFFF;
		for (int i = first; i < n; i += stride)
			if (Pc[i] != 0 && !g.is_pivot_row[i]) {
				mine = i;
				break;
			}
		if (mine != 0x7FFFFFFF)
			atomicMin(&s_min, mine);
		__syncthreads();
		if (tid == 0 && s_min != 0x7FFFFFFF)
			atomicMin(&ca.cand[c], s_min);
		grid_barrier(ca.barrier, (++phase) * (unsigned int) G, ca.err);
		const int rho = __hip_atomic_load(&ca.cand[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (rho == 0x7FFFFFFF || rho >= n)
			continue;
		{
			const uint32_t inv = invmod(Pc[rho], F);
			for (int cc = tid; cc < PW; cc += COOP_THREADS) {
				uint32_t v = 0;
				if (cc >= c && cc < W)
					v = g.P[(int64_t) cc * nn + rho];
				else if (cc >= NB && cc < NB + k)
					v = g.P[(int64_t) cc * nn + rho];
				else if (cc == NB + k)
					v = 1;
				piv[cc] = (v != 0) ? mulmod(v, inv, F) : 0u;
			}
		}
		grid_barrier(ca.barrier, (++phase) * (unsigned int) G, ca.err);      // everybody holds piv[]
		for (int i = first; i < n; i += stride) {
			if (i == rho) {
				for (int cc = c; cc < W; cc++)
					g.P[(int64_t) cc * nn + i] = piv[cc];
				for (int cc = NB; cc <= NB + k; cc++)
					g.P[(int64_t) cc * nn + i] = piv[cc];
				g.is_pivot_row[rho] = 1;
				continue;
			}
			const uint32_t f = Pc[i];
			if (f == 0)
				continue;
			for (int cc = c; cc < W; cc++) {
				if (piv[cc] == 0)
					continue;
				uint32_t *x = g.P + (int64_t) cc * nn + i;
				*x = submod(*x, mulmod(f, piv[cc], F), F);
			}
			for (int cc = NB; cc <= NB + k; cc++) {
				if (piv[cc] == 0)
					continue;
				uint32_t *x = g.P + (int64_t) cc * nn + i;
				*x = submod(*x, mulmod(f, piv[cc], F), F);
			}
		}
		if (blockIdx.x == 0 && tid == 0) {
			const int rk = *g.rank;
			g.pivrow[rk] = rho;
			g.pivcol[rk] = g.c0 + c;
			*g.rank = rk + 1;
			g.rho[k] = rho;
		}
		k += 1;
	}
	// write the eliminated panel back (own rows); M = T J - J
	for (int i = first; i < n; i += stride)
		for (int c = 0; c < W; c++)
			g.A[(int64_t) i * g.ld + g.c0 + c] = g.P[(int64_t) c * nn + i];
	grid_barrier(ca.barrier, (++phase) * (unsigned int) G, ca.err);
	if (blockIdx.x == 0) {
		if (tid < k) {
			uint32_t *Pr = g.P + (int64_t) (NB + tid) * nn + g.rho[tid];
			*Pr = submod(*Pr, 1u, F);
		}
		if (tid == 0)
			*g.knew = k;
	}
}

// --------------------------------------------------------------------------
// Tournament panel step (default).  The column-by-column panel kernels above
// pay one grid-wide synchronisation per column.  Instead:
//   1. rref_free_list: the rows that hold no pivot yet;
//   2. rref_select_kernel, a tree of them: a workgroup takes 256 candidate rows (one per thread, their 64 panel
//      entries column-major in LDS), runs a fraction-free forward elimination with "first non-zero" pivoting and
//      passes on the <= 64 rows it picked -- a basis of the row space of its 256 rows restricted to the panel.
//      The root of the tree therefore holds k <= 64 ACTUAL rows rho spanning what every free row has in the panel;
//   3. rref_block_gj (one workgroup): Gauss-Jordan of [ A[rho, panel] | I ]: pivot columns gamma (the column rank
//      profile of the panel: leftmost, because the k rows span everything) and Ginv with Ginv A[rho, gamma] = I;
//   4. rref_multipliers: M = -A[:, gamma] Ginv (+ Ginv on the pivot rows), n x k, so that T = I + M e_rho^T;
//   5. the usual gather + GEMM update, from the first column of the panel on.
// Everything is exact arithmetic: any basis gives the same (unique) reduced echelon form.
// --------------------------------------------------------------------------
constexpr int SEL_ROWS = 256;

// The candidates of a try: at most 64 rows without pivot, picked by one wave.  Without a list of live rows: the first ones
// from *hint on (*hint moves up to the first of them).  With one (rref_mark_dead below: the rows that are not zero from the
// super-panel on, in no particular order): rows SPREAD over the list.  On a block the flow really produces -- the first
// dense rows of mk13.b5's Schur complement: 4,096 x 4,952, rank 1,583 -- the first 64 free rows are soon all zero rows (they
// depend on the pivots found so far and stay "free" for ever), and 64 CONSECUTIVE live rows have rank ~40 on a panel where 64
// rows spread over the block have 64: the try failed on 58 of 64 panels and the tournament ran.
__device__ __forceinline__ int pick_candidates(const int *flags, int n, int *hint, const int *live_list, const int *live_count, int *out, int lane)
{
	int found = 0;
	const int cnt = (live_list != nullptr) ? *live_count : 0;
	if (cnt >= NB) {
		const int stride = cnt / NB;
		for (int round = 0; round < stride && found < NB; round++) {
			const int pos = lane * stride + round;
			const int row = (pos < cnt) ? live_list[pos] : -1;
			const bool fr = row >= 0 && flags[row] == 0;          // (a row of the list may have become a pivot row since)
			const unsigned long long mk = __ballot(fr);
			const int at = found + __popcll(mk & ((1ull << lane) - 1ull));
			if (fr && at < NB)
				out[at] = row;
			found += __popcll(mk);
		}
		return min(found, NB);
	}
	int first = -1;
	for (int base = *hint; base < n && found < NB; base += 64) {
		const int i = base + lane;
		const bool fr = i < n && flags[i] == 0;
		const unsigned long long mk = __ballot(fr);
		const int pos = found + __popcll(mk & ((1ull << lane) - 1ull));
		if (fr && pos < NB)
			out[pos] = i;
		if (first < 0 && mk != 0)
			first = base + __builtin_ctzll(mk);
		found += __popcll(mk);
	}
	if (lane == 0)
		*hint = (first >= 0) ? first : n;
	return min(found, NB);
}

__global__ __launch_bounds__(64) void rref_first_free(const int *flags, int n, int *hint, int *out, int *count, const int *live_list, const int *live_count)
{
	const int found = pick_candidates(flags, n, hint, live_list, live_count, out, (int) threadIdx.x);
	if (threadIdx.x == 0)
		*count = found;
}

// Rows without pivot that are zero from column c_from on can never hold one: they leave the free rows for good (flags = 2);
// the others are listed (any order).  One wave per row; a live row usually says so in its first few hundred entries.
__global__ __launch_bounds__(256) void rref_mark_dead(const uint32_t *A, int64_t ld, int n, int m, int c_from, int *flags, int *live_list, int *live_count)
{
	const int lane = threadIdx.x & 63;
	const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
	for (int i = wave; i < n; i += nwaves) {
		if (flags[i] != 0)
			continue;
		bool live = false;
		for (int j0 = c_from; j0 < m && !live; j0 += 256) {
			bool nz = false;
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int j = j0 + 64 * u + lane;
				nz |= j < m && A[(int64_t) i * ld + j] != 0;
			}
			live = __ballot(nz) != 0;
		}
		if (lane == 0) {
			if (live)
				live_list[atomicAdd(live_count, 1)] = i;
			else
				flags[i] = 2;
		}
	}
}

// list[0 .. count) = rows i with flags[i] == 0, increasing.  One workgroup.  (skip: see rref_select_kernel)
__global__ __launch_bounds__(1024) void rref_free_list(const int *flags, int n, int *list, int *count, const int *skip, int *reset = nullptr)
{
	__shared__ int part[1024];
	const int tid = threadIdx.x;
	if (reset != nullptr) {
		// (a panel step that starts here -- the try and the selection on the first rows were not even launched: they fail on
		//  this block --: "not done yet" for the kernels that follow; reset[0] = *full = *skip, reset[4] = *gj_done)
		if (tid == 0) {
			reset[0] = 0;
			reset[4] = 0;
		}
		__syncthreads();
	}
	if (skip != nullptr && *skip != 0)
		return;
	const int per = (n + 1023) / 1024;
	const int lo = min(n, tid * per), hi = min(n, lo + per);
	int c = 0;
	for (int i = lo; i < hi; i++)
		c += flags[i] == 0;
	part[tid] = c;
	__syncthreads();
	for (int d = 1; d < 1024; d <<= 1) {
		const int v = (tid >= d) ? part[tid - d] : 0;
		__syncthreads();
		part[tid] += v;
		__syncthreads();
	}
	int pos = part[tid] - c;
	for (int i = lo; i < hi; i++)
		if (flags[i] == 0)
			list[pos++] = i;
	if (tid == 1023)
		*count = part[1023];
}

// pv * x - f * y mod p, up to a common non-zero factor (fraction-free elimination step).
//   SMALL (p < 46341, so that 2 p^2 < 2^32): 24-bit multiplies (full rate) and one Barrett reduction;
//   otherwise two Montgomery products (a common factor 2^-32, which does not matter here).
template <bool SMALL> struct ElimArith {
	uint32_t p, pinv, pp, m;
	__device__ explicit ElimArith(const MontDev &F) : p(F.p), pinv(F.pinv), pp(SMALL ? F.p * F.p : 0u), m(SMALL ? (uint32_t) (0x100000000ull / F.p) : 0u) {}
	__device__ __forceinline__ uint32_t mulsub(uint32_t pv, uint32_t x, uint32_t f, uint32_t y) const
	{
		if constexpr (SMALL) {
			const uint32_t v = __umul24(pv, x) + (pp - __umul24(f, y));       // in (0, 2 p^2) < 2^32
			const uint32_t q = __umulhi(v, m);                                // floor(v / p) - 2 <= q <= floor(v / p)
			uint32_t rem = v - __umul24(q, p);
			rem = (rem >= p) ? rem - p : rem;
			rem = (rem >= p) ? rem - p : rem;
			return rem;
		} else {
			const MontDev F{p, pinv, 0u, 0u, 0u};
			return submod(montmul(pv, x, F), montmul(f, y, F), F);
		}
	}
};

template <int I, int N, typename Fn> __device__ __forceinline__ void dense_static_for(Fn &&f)
{
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		dense_static_for<I + 1, N>(f);
	}
}

// cand_in: n_in row indices (-1 = nothing); n_in_dev (if not null) overrides n_in.  Output: 64 slots per workgroup.
// One row per thread, kept in 64 registers (the column loop is unrolled: static register indices); the pivot row
// of a step is broadcast through LDS.
template <bool SMALL>
__global__ __launch_bounds__(SEL_ROWS) void rref_select_kernel(const uint32_t *A, int64_t ld, int c0, int width, const int *cand_in,
                                                               int n_in, const int *n_in_dev, int *cand_out, MontDev F,
                                                               const int *skip, int *full)
{
	__shared__ __attribute__((aligned(16))) uint32_t prow[NB];
	__shared__ unsigned long long wave_nz[SEL_ROWS / 64];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const ElimArith<SMALL> E(F);
	// skip: the first 256 free rows already gave a pivot in every column of the panel (tournament not needed)
	if (skip != nullptr && *skip != 0)
		return;
	if (n_in_dev != nullptr)
		n_in = (full != nullptr) ? min(*n_in_dev, SEL_ROWS) : *n_in_dev;
	const int idx = blockIdx.x * SEL_ROWS + tid;
	const int row = (idx < n_in) ? cand_in[idx] : -1;
	if (blockIdx.x * SEL_ROWS >= n_in) {          // nothing for this workgroup
		if (tid < NB)
			cand_out[blockIdx.x * NB + tid] = -1;
		if (full != nullptr && tid == 0)
			*full = 0;
		return;
	}
	uint32_t x[NB];
	{
		const uint32_t *src = A + (int64_t) (row >= 0 ? row : 0) * ld + c0;
#pragma unroll
		for (int j = 0; j < NB; j++)
			x[j] = (row >= 0 && j < width) ? src[j] : 0u;
	}
	bool selected = false;
	int nsel = 0;
	dense_static_for<0, NB>([&](auto cc) {
		constexpr int col = decltype(cc)::value;
		if (col >= width)
			return;
		const uint32_t mine = x[col];
		const unsigned long long nz = __ballot(!selected && mine != 0);
		if (lane == 0)
			wave_nz[wave] = nz;
		__syncthreads();
		int winner = -1;
#pragma unroll
		for (int q = SEL_ROWS / 64 - 1; q >= 0; q--) {
			const unsigned long long w = wave_nz[q];
			if (w != 0)
				winner = q * 64 + __builtin_ctzll(w);
		}
		if (winner < 0) {
			__syncthreads();            // (wave_nz is rewritten by the next column)
			return;
		}
		if (tid == winner) {
#pragma unroll
			for (int j = col; j < NB; j++)
				prow[j] = x[j];
			selected = true;
			cand_out[blockIdx.x * NB + nsel] = row;
		}
		nsel += 1;
		__syncthreads();
		if (!selected && mine != 0) {
			const uint32_t pv = prow[col];
#pragma unroll
			for (int j = col + 1; j < NB; j++)
				x[j] = E.mulsub(pv, x[j], mine, prow[j]);
		}
		__syncthreads();                // (prow is rewritten by the next column)
	});
	if (tid >= nsel && tid < NB)
		cand_out[blockIdx.x * NB + tid] = -1;
	if (full != nullptr && tid == 0)
		*full = (nsel == width) ? 1 : 0;
}

// The first 64 free rows alone, four lanes per row (16 of the 64 panel entries each): the usual case is that they
// already give a pivot in every column of the panel, and a short step matters more than many candidates (the step
// time of rref_select_kernel is set by the 64 entries one thread owns).  Same elimination, same output format;
// *full = 1 when every column of the panel got a pivot -- the tournament over all free rows then returns at once.
template <bool SMALL>
__global__ __launch_bounds__(256) void rref_select_first(const uint32_t *A, int64_t ld, int c0, int width, const int *cand_in,
                                                         const int *n_in_dev, int *cand_out, MontDev F, int *full, const int *gj_done)
{
	__shared__ __attribute__((aligned(16))) uint32_t prow[NB];
	__shared__ unsigned long long wave_nz[4];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int slot = tid >> 2, q = tid & 3;
	const ElimArith<SMALL> E(F);
	if (gj_done != nullptr && *gj_done != 0) {          // rref_block_gj in try mode already did the panel
		if (tid == 0)
			*full = 1;                                  // (the tournament kernels return at once on this)
		return;
	}
	const int n_in = min(*n_in_dev, NB);
	const int row = (slot < n_in) ? cand_in[slot] : -1;
	uint32_t x[16];
	{
		const uint32_t *src = A + (int64_t) (row >= 0 ? row : 0) * ld + c0 + 16 * q;
#pragma unroll
		for (int e = 0; e < 16; e++)
			x[e] = (row >= 0 && 16 * q + e < width) ? src[e] : 0u;
	}
	bool selected = false;
	int nsel = 0;
	dense_static_for<0, NB>([&](auto cc) {
		constexpr int col = decltype(cc)::value;
		constexpr int qc = col / 16, ec = col % 16;
		if (col >= width)
			return;
		// my row's entry in this column sits in the lane of quarter qc
		const uint32_t f = (uint32_t) __builtin_amdgcn_mov_dpp((int) x[ec], qc * 0x55, 0xf, 0xf, true);    // quad_perm: all four lanes read lane qc
		const unsigned long long nz = __ballot(!selected && f != 0);
		if (lane == 0)
			wave_nz[wave] = nz;
		__syncthreads();
		int winner = -1;          // a thread of the first row that qualifies (rows are in thread order)
#pragma unroll
		for (int w = 3; w >= 0; w--) {
			const unsigned long long v = wave_nz[w];
			if (v != 0)
				winner = w * 64 + __builtin_ctzll(v);
		}
		if (winner < 0) {
			__syncthreads();            // (wave_nz is rewritten by the next column)
			return;
		}
		if (slot == (winner >> 2)) {
#pragma unroll
			for (int e = 0; e < 16; e++)
				prow[16 * q + e] = x[e];
			selected = true;
			if (q == 0)
				cand_out[nsel] = row;
		}
		nsel += 1;
		__syncthreads();
		if (!selected && f != 0) {
			const uint32_t pv = prow[col];
			if (q > qc) {
#pragma unroll
				for (int e = 0; e < 16; e++)
					x[e] = E.mulsub(pv, x[e], f, prow[16 * q + e]);
			} else if (q == qc) {
#pragma unroll
				for (int e = ec + 1; e < 16; e++)
					x[e] = E.mulsub(pv, x[e], f, prow[16 * qc + e]);
				x[ec] = 0;
			}
		}
		__syncthreads();                // (prow is rewritten by the next column)
	});
	if (tid >= nsel && tid < NB)
		cand_out[tid] = -1;
	if (tid == 0)
		*full = (nsel == width) ? 1 : 0;
}

struct BlockGjArgs {
	const uint32_t *A;
	int64_t ld;
	int n, c0, width;
	const int *cand;       // NB slots, -1 = nothing: independent rows (output of the root of the tournament)
	const int *cand_first; // ... or of the first 256 free rows, when *full says they gave a pivot in every column
	const int *full;
	const unsigned short *invtab;   // p < 46341: inverses of 0 .. p-1 (global copy of the LDS table)
	uint32_t *Ginv;        // NB x NB, Montgomery form (value * 2^32 mod p), row s = pivot s, column r = candidate r
	int *gamma;            // NB: pivot columns inside the panel, increasing
	int *is_pivot_row;
	int *pivrow, *pivcol, *rank, *knew, *rho;
	int *cand_pivot;       // NB: pivot index of candidate r (every candidate becomes a pivot row)
	// try mode (mode == 1): Gauss-Jordan straight on the first free rows (try_rows[0 .. *try_count)), before any selection.
	// When they give a pivot in every column of the panel -- the usual case while the block is not exhausted -- the
	// panel step is done (*gj_done = 1) and the selection kernels and the regular call return at once; otherwise
	// nothing is written.  try_state[0] = failures in a row: after two the try is skipped (a block of low rank, or
	// rows that are already reduced, would pay for it on every panel), and taken up again every eighth panel.
	int mode;
	const int *try_rows, *try_count;
	int *gj_done, *try_state;
	int panel_index;
	// optimistic super-panels (driver): the panel steps are enqueued without the selection kernels; a try that cannot
	// finish its panel raises *abort (1 + abort_value) and every later kernel of the super-panel returns at once
	int *abort;
	int abort_value;
	// (passes that run the tries one panel ahead of the updates: a word per panel -- abort points at this panel's --, and the try that
	//  fails raises the words of the LATER panels and the summary word the host reads: the kernels of the panels before it, which may not
	//  have started yet, must run)
	int *abort_words, *abort_summary;
	int abort_count;
	// ... and rref_try_inverse finds the first free rows itself (what rref_first_free does in its own launch otherwise):
	// ff_flags != null: scan from *ff_hint, write try_rows / *try_count through ff_out / ff_count
	const int *ff_flags;
	int *ff_hint, *ff_out, *ff_count;
	const int *live_list, *live_count;          // rref_mark_dead's list of this super-panel, or null
	int *done_word;        // rref_try_inverse: counted up when the kernel is through (null: nobody polls)
	// rref_try_inverse: the 64 x 64 block of the candidates as rref_lookahead left it (row-major, residues), instead of A's -- the
	// update of the panel before has not reached A yet
	const uint32_t *alt;
	// mode == 2 (SMALL only): a try that takes what its candidates give -- k <= 64 pivots, the other columns of the panel skipped
	// as dependent -- and leaves the proof to the multiplier kernel (MultArgs::verify_flags): after the step every row without a
	// pivot must be zero on the whole panel, which says that no skipped column could have had a pivot anywhere (then the
	// pivot columns are the leftmost possible ones: the step is the regular step).  Only the candidates that got a pivot are
	// passed on (knew = the number of pivots; knew[1] = the same = what rref_rollback takes back when the proof fails).
	MontDev F;
};

// inverses mod p of 1 .. p-1 (p < 46341), for the small-prime Gauss-Jordan below
__global__ __launch_bounds__(256) void rref_inverse_table(unsigned short *tab, MontDev F)
{
	const uint32_t a = blockIdx.x * 256 + threadIdx.x;
	if (a < F.p)
		tab[a] = (a == 0) ? 0 : (unsigned short) invmod(a, F);
}

// One workgroup of 128 NPAR threads, Gauss-Jordan of [R | I] (64 x 128).  Thread t keeps column t % 128 of the rows
// r = t / 128 mod NPAR in 64 / NPAR registers; per step the column being eliminated and the pivot row go through LDS.
//   SMALL (p < 46341): the table of inverses sits in LDS (2 p bytes, dynamic), pivot rows are normalised, an
//   update is one 24-bit multiply + Barrett and only touches the columns where the pivot row is non-zero;
//   otherwise: fraction-free steps (two Montgomery products per element), one inversion per pivot at the end.
template <bool SMALL, int NPAR>
__global__ __launch_bounds__(128 * NPAR) void rref_block_gj(BlockGjArgs g)
{
	extern __shared__ __attribute__((aligned(16))) unsigned short invtab[];
	__shared__ uint32_t colbuf[2][NB];
	__shared__ uint32_t prow[PW];
	__shared__ uint32_t diag[NB];
	__shared__ int s_rows[NB], s_gamma[NB], s_prow_of[NB];
	__shared__ int s_k;
	const int tid = threadIdx.x;
	const MontDev F = g.F;
	const ElimArith<SMALL> E(F);
	if (g.mode == 0 && *g.gj_done != 0)
		return;                       // the try on the first free rows already did this panel
	if (g.mode == 2) {
		if (*g.abort != 0)
			return;
		if (tid < 64 && g.ff_flags != nullptr) {          // (null: the candidates were selected by the launch before)
			const int found = pick_candidates(g.ff_flags, g.n, g.ff_hint, g.live_list, g.live_count, g.ff_out, tid);
			if (tid == 0)
				*g.ff_count = found;
		}
		if (tid == 0)
			g.knew[1] = 0;
		__syncthreads();
		if (*g.try_count < 1 || g.width < NB) {
			if (tid == 0) {
				*g.gj_done = 0;
				*g.abort = 1 + g.abort_value;
			}
			return;
		}
	}
	if (g.mode == 1) {
		const bool skip = g.try_state[0] >= 2 && (g.panel_index & 7) != 0;
		if (skip || *g.try_count < g.width || g.width < NB) {
			if (tid == 0)
				*g.gj_done = 0;
			return;
		}
	}
	if (tid < NB) {
		const int *cand = (g.mode >= 1) ? g.try_rows : (*g.full != 0) ? g.cand_first : g.cand;
		const int c = (g.mode >= 1 && tid >= *g.try_count) ? -1 : cand[tid];
		const unsigned long long have = __ballot(c >= 0);
		if (c >= 0)
			s_rows[__popcll(have & ((1ull << tid) - 1ull))] = c;
		if (tid == 0)
			s_k = __popcll(have);
		s_prow_of[tid] = -1;          // pivot index held by row r
	}
	__syncthreads();
	const int k = s_k;
	if (k == 0) {
		if (tid == 0)
			*g.knew = 0;
		return;
	}
	if constexpr (SMALL) {
		// (2 p bytes, copied 16 at a time; both copies are padded)
		const uint4 *src = reinterpret_cast<const uint4 *>(g.invtab);
		uint4 *dst = reinterpret_cast<uint4 *>(invtab);
		for (uint32_t t = tid; t < (2 * F.p + 15) / 16; t += 128 * NPAR)
			dst[t] = src[t];
	}
	constexpr int RPT = NB / NPAR;                          // rows per thread
	const int j = tid & (PW - 1), par = tid >> 7;          // my column; my rows: r = NPAR i + par
	uint32_t x[RPT];
#pragma unroll
	for (int i = 0; i < RPT; i++) {
		const int r = NPAR * i + par;
		uint32_t v = 0;
		if (r < k) {
			if (j < NB)
				v = (j < g.width) ? g.A[(int64_t) s_rows[r] * g.ld + g.c0 + j] : 0u;
			else
				v = (j - NB == r) ? 1u : 0u;
		}
		x[i] = v;
	}
	const uint32_t bm = SMALL ? (uint32_t) (0x100000000ull / F.p) : 0u;
	auto barrett = [&](uint32_t v) -> uint32_t {           // v < 2^32 -> v mod p (p < 2^16)
		const uint32_t q = __umulhi(v, bm);
		uint32_t rem = v - __umul24(q, F.p);
		rem = (rem >= F.p) ? rem - F.p : rem;
		rem = (rem >= F.p) ? rem - F.p : rem;
		return rem;
	};
	// Two barriers per column: every wave finds the pivot row of the column by itself (one ballot over the column's
	// 64 entries, which the owner of the column published during the previous step), the owners of that row
	// publish it (normalised), barrier, everybody updates its registers and the owner of the NEXT column publishes
	// it, barrier.
	const int lane = tid & 63;
	int npiv = 0;
	if (j == 0) {
#pragma unroll
		for (int i = 0; i < RPT; i++)
			colbuf[0][NPAR * i + par] = x[i];
	}
	__syncthreads();
	for (int col = 0; col < g.width && npiv < k; col++) {
		const uint32_t *cb = colbuf[col & 1];
		uint32_t *cb_next = colbuf[(col + 1) & 1];
		const unsigned long long nz = __ballot(lane < k && s_prow_of[lane] < 0 && cb[lane] != 0);
		const int pr = (nz != 0) ? (int) __builtin_ctzll(nz) : -1;
		if (pr < 0) {
			// no pivot in this column (uniform): only the next column has to be published
			if (j == col + 1) {
#pragma unroll
				for (int i = 0; i < RPT; i++)
					cb_next[NPAR * i + par] = x[i];
			}
			__syncthreads();
			continue;
		}
		const uint32_t pv = cb[pr];
		if ((pr % NPAR) == par) {
			uint32_t mine = 0;
#pragma unroll
			for (int i = 0; i < RPT; i++)
				mine = (i == (pr / NPAR)) ? x[i] : mine;
			if constexpr (SMALL) {
				// normalise the pivot row (in its registers too)
				mine = barrett(__umul24(mine, (uint32_t) invtab[pv]));
#pragma unroll
				for (int i = 0; i < RPT; i++)
					x[i] = (i == (pr / NPAR)) ? mine : x[i];
			}
			prow[j] = mine;
		}
		__syncthreads();
		if (tid == 0) {           // (read again only after the barrier below)
			s_prow_of[pr] = npiv;
			s_gamma[npiv] = col;
		}
		const uint32_t pj = prow[j];
		if constexpr (SMALL) {
			if (pj != 0) {
#pragma unroll
				for (int i = 0; i < RPT; i++) {
					const int r = NPAR * i + par;
					const uint32_t f = cb[r];
					if (r != pr && f != 0)
						x[i] = barrett(x[i] + __umul24(F.p - f, pj));          // < p + p^2 < 2^32
				}
			}
		} else {
#pragma unroll
			for (int i = 0; i < RPT; i++) {
				const int r = NPAR * i + par;
				const uint32_t f = cb[r];
				if (r != pr && f != 0)
					x[i] = (j == col) ? 0u : E.mulsub(pv, x[i], f, pj);
			}
		}
		if (j == col + 1) {
#pragma unroll
			for (int i = 0; i < RPT; i++)
				cb_next[NPAR * i + par] = x[i];
		}
		npiv += 1;
		__syncthreads();
	}
	// (the candidates are independent: npiv == k; a defect would show as a wrong rank in the tests)
	if (g.mode == 1) {
		const bool ok = npiv == g.width && npiv == k;
		if (tid == 0) {
			*g.gj_done = ok ? 1 : 0;
			g.try_state[0] = ok ? 0 : g.try_state[0] + 1;
		}
		if (!ok)
			return;                   // nothing written: the selection kernels and the regular call take over
	}
	if (g.mode == 2) {
		if (tid == 0)
			*g.gj_done = 1;
		if (npiv == 0) {
			// no pivot among the candidates: the multiplier kernel checks that NO free row has anything on this panel
			if (tid == 0)
				*g.knew = 0;
			return;
		}
	}
	const bool compact = g.mode == 2;          // pass on the candidates that hold a pivot, under their pivot index
	if constexpr (SMALL) {
		// pivot rows are normalised: the right half is Ginv
#pragma unroll
		for (int i = 0; i < RPT; i++) {
			const int r = NPAR * i + par;
			if (r < k && j >= NB && j - NB < k && s_prow_of[r] >= 0) {
				// (compact: a pivot row is a combination of candidates that hold a pivot -- the others only ever RECEIVE)
				const int col_of = compact ? s_prow_of[j - NB] : j - NB;
				if (col_of >= 0)
					g.Ginv[s_prow_of[r] * NB + col_of] = montmul(x[i], F.r2, F);          // Montgomery form
			}
		}
	} else {
		// scale: pivot row s has d at gamma_s, zeros at the other pivot columns; Ginv[s][r] = aug[s][r] / d
#pragma unroll
		for (int i = 0; i < RPT; i++) {
			const int r = NPAR * i + par;
			if (r < k && s_prow_of[r] >= 0 && j == s_gamma[s_prow_of[r]])
				diag[r] = x[i];
		}
		__syncthreads();
		if (tid < k && s_prow_of[tid] >= 0)
			diag[tid] = invmod(diag[tid], F);
		__syncthreads();
#pragma unroll
		for (int i = 0; i < RPT; i++) {
			const int r = NPAR * i + par;
			if (r < k && j >= NB && j - NB < k && s_prow_of[r] >= 0)
				// Montgomery form of (aug / d): mulmod gives the plain product, one more montmul by r2 lifts it
				g.Ginv[s_prow_of[r] * NB + (j - NB)] = montmul(mulmod(x[i], diag[r], F), F.r2, F);
		}
	}
	const int base = *g.rank;
	__syncthreads();
	if (tid < npiv)
		g.gamma[tid] = s_gamma[tid];
	if (tid < k) {
		const int s = s_prow_of[tid];
		if (!compact) {
			g.rho[tid] = s_rows[tid];
			g.cand_pivot[tid] = s;
		} else if (s >= 0) {
			g.rho[s] = s_rows[tid];
			g.cand_pivot[s] = s;
		}
		if (s >= 0) {
			g.is_pivot_row[s_rows[tid]] = 1;
			g.pivrow[base + s] = s_rows[tid];
			g.pivcol[base + s] = g.c0 + s_gamma[s];
		}
	}
	if (tid == 0) {
		*g.rank = base + npiv;
		*g.knew = compact ? npiv : k;
		if (compact)
			g.knew[1] = npiv;
	}
}

// ---- device-side hand-offs between the two streams of a pass that runs its tries one panel ahead --------------------------
// Events between the streams cost ~10 us per hop (a record and a wait are barrier packets of their own: the trace showed 11 us
// between the end of a try and the start of the lookahead behind it on the SAME queue).  Instead: the producer's workgroups count
// themselves in on a word once their stores have left (every wave's stores drained, the workgroup's barrier, an agent-scope
// release by one lane, then the atomic), the consumer's workgroups poll it with one lane (relaxed agent-scope loads), take an
// agent-scope acquire and meet at a barrier before any of them reads.  (MI355X_MICROARCH.md, "Workgroup dispatch, XCD
// placement & inter-workgroup visibility": the per-XCD L2s are not coherent with each other.)  Every exit of a producer
// signals -- an aborted pass must not leave anybody polling.
// A consumer can only be served if its producer is on the device at the same time.  Tools that let one kernel run at a time
// (rocprofv3 --pmc, AMD_SERIALIZE_KERNEL) break that: the host asks handoff_probe once per pair of streams whether two kernels of
// the two streams DO meet (no -> the pass runs without the lookahead), and every wait is bounded all the same: a wait that runs
// out raises g_handoff_stuck, later waits return at once, and the call fails loudly instead of hanging.
__device__ int g_handoff_stuck;
constexpr long long HANDOFF_PATIENCE = 200000000ll;          // ticks of the 100 MHz wall clock: two seconds (a wait is tens of microseconds)

__device__ __forceinline__ bool handoff_poll(const int *word, int value, long long patience)
{
	unsigned int spins = 0;
	long long t0 = 0;
	while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
		__builtin_amdgcn_s_sleep(4);
		if ((++spins & 255u) == 0) {
			const long long now = wall_clock64();
			if (t0 == 0)
				t0 = now;
			else if (now - t0 > patience || __hip_atomic_load(&g_handoff_stuck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
				return false;
		}
	}
	return true;
}

__device__ __forceinline__ void handoff_wait(const int *word, int value)
{
	if (word == nullptr)
		return;
	if (threadIdx.x == 0) {
		if (!handoff_poll(word, value, HANDOFF_PATIENCE))
			__hip_atomic_store(&g_handoff_stuck, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
	__syncthreads();
}

// The question to the device: do a kernel of one stream and a kernel of the other run side by side?  Each of the two raises its
// own word and waits (3 ms at most) for the other's; met[side] = 1 when it saw it.
__global__ void handoff_probe(int *words, int side, int *met)
{
	__hip_atomic_store(words + side, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	met[side] = handoff_poll(words + (side ^ 1), 1, 300000ll) ? 1 : 0;
}

__device__ __forceinline__ void handoff_signal(int *word)
{
	if (word == nullptr)
		return;
	__syncthreads();          // (after every wave's s_waitcnt vmcnt(0): the compiler puts it before the barrier)
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		__hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

__device__ __forceinline__ void raise_abort(const BlockGjArgs &g)
{
	*g.abort = 1 + g.abort_value;
	if (g.abort_words != nullptr) {
		for (int t = g.abort_value + 1; t < g.abort_count; t++)
			g.abort_words[t] = 1 + g.abort_value;
		*g.abort_summary = 1 + g.abort_value;
	}
}

// the pivots of a mode-2 try whose proof failed are taken back (the matrix itself was not touched: the update returned at once)
__global__ __launch_bounds__(64) void rref_rollback(const int *rho, int *knew, int *flags, int *rank)
{
	const int committed = knew[1];
	if ((int) threadIdx.x < committed)
		flags[rho[threadIdx.x]] = 0;
	__syncthreads();
	if (threadIdx.x == 0) {
		*rank -= committed;
		knew[1] = 0;
	}
}

// The usual case of the try mode above -- 64 free rows, 64 columns, a pivot in every column, a small prime -- is the
// inversion of a 64 x 64 block.  Four waves, lane = row, sixteen columns per wave in registers as signed representatives
// with deferred reduction (|x| <= p/2 + p/64 after a reduction through fp32, four multiply-adds of such values fit 32
// bits: p <= 44934), IN PLACE: the column being eliminated becomes the column of the inverse.  One barrier per
// column -- the wave that holds the column publishes it, the pivot row and the inverse of the pivot -- where the general
// kernel needs its 1024 threads, two barriers and a round of LDS traffic per column: 22 us instead of 92 us per panel.
// Writes what rref_block_gj writes in mode 1; a column without pivot leaves everything to the regular path.
__device__ __forceinline__ int gj_reduce(int t, int negp, float invp)
{
	const int q = (int) __builtin_rintf((float) t * invp);
	return t + __mul24(q, negp);
}

__device__ __forceinline__ void try_inverse_body(const BlockGjArgs &g)
{
	extern __shared__ __attribute__((aligned(16))) unsigned short invtab[];
	__shared__ int fcol[2][NB], s_pr[2], s_pv[2], s_inv[2];
	__shared__ int prow[NB], sigma[NB], prow_of[NB], s_rows[NB];
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const MontDev F = g.F;
	if (g.abort != nullptr && *g.abort != 0)
		return;
	if (g.ff_flags != nullptr) {
		if (w == 0) {
			const int found = pick_candidates(g.ff_flags, g.n, g.ff_hint, g.live_list, g.live_count, g.ff_out, lane);
			if (lane == 0)
				*g.ff_count = found;
		}
		__syncthreads();
	}
	const bool skip = g.abort == nullptr && g.try_state[0] >= 2 && (g.panel_index & 7) != 0;
	if (skip || *g.try_count < NB || g.width < NB) {
		if (tid == 0) {
			*g.gj_done = 0;
			if (g.abort != nullptr)
				raise_abort(g);
		}
		return;
	}
	{
		// (eight loads in flight per thread: the plain loop waits for every 16 bytes -- 21 round trips for p = 42013)
		const uint4 *src = reinterpret_cast<const uint4 *>(g.invtab);
		uint4 *dst = reinterpret_cast<uint4 *>(invtab);
		const uint32_t total = (2 * F.p + 15) / 16;
		for (uint32_t t0 = tid; t0 < total; t0 += 8 * 256) {
			uint4 v[8];
#pragma unroll
			for (int u = 0; u < 8; u++)
				v[u] = (t0 + u * 256 < total) ? src[t0 + u * 256] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
			for (int u = 0; u < 8; u++)
				if (t0 + u * 256 < total)
					dst[t0 + u * 256] = v[u];
		}
	}
	const int p = (int) F.p, negp = -p, half = (int) F.half;
	const float invp = 1.0f / (float) p;
	const int row = g.try_rows[lane];
	int x[16];
#pragma unroll
	for (int j = 0; j < 16; j++) {
		const int v = (g.alt != nullptr) ? (int) g.alt[lane * NB + 16 * w + j] : (int) g.A[(int64_t) row * g.ld + g.c0 + 16 * w + j];
		x[j] = (v > half) ? v - p : v;
	}
	if (tid < NB)
		s_rows[tid] = row;
	__syncthreads();
	bool used = false;
	for (int o = 0; o < 4; o++) {
#pragma unroll
		for (int jc = 0; jc < 16; jc++) {
			const int c = 16 * o + jc, par = c & 1;
			if (w == o) {
				const int fc = gj_reduce(x[jc], negp, invp);
				const unsigned long long cand = __ballot(!used && fc != 0);
				const int pr = (cand != 0) ? (int) __builtin_ctzll(cand) : -1;
				const int pv = __builtin_amdgcn_readlane(fc, pr < 0 ? 0 : pr);
				fcol[par][lane] = fc;
				if (lane == 0) {
					int inv = (pr < 0) ? 0 : (int) invtab[pv < 0 ? pv + p : pv];
					inv = (inv > half) ? inv - p : inv;
					s_pr[par] = pr;
					s_pv[par] = pv;
					s_inv[par] = inv;
				}
			}
			__syncthreads();
			// (everything the owner published, in one round of LDS reads)
			const int pr = s_pr[par], pv = s_pv[par], inv = s_inv[par], fmine = fcol[par][lane];
			if (pr < 0) {                    // no pivot in this column: the regular path takes the panel
				if (tid == 0) {
					*g.gj_done = 0;
					g.try_state[0] = g.try_state[0] + 1;
					if (g.abort != nullptr)
						raise_abort(g);
				}
				return;
			}
			if (w == o)
				x[jc] = (lane == pr) ? 1 : 0;          // in place: this column becomes the column of the inverse that row pr stands for
			// my sixteen entries of the pivot row, normalised (lanes 0 .. 15), back through LDS for everybody's update
			if (lane == pr) {
#pragma unroll
				for (int j = 0; j < 16; j++)
					prow[16 * w + j] = x[j];
			}
			if (lane < 16) {
				int t = gj_reduce(prow[16 * w + lane], negp, invp);
				t = gj_reduce(__mul24(t, inv), negp, invp);
				prow[16 * w + lane] = t;
			}
			const int fneg = (lane == pr) ? 1 - pv : -fmine;          // (row pr: x = pv * prow, so x + (1 - pv) prow = prow)
#pragma unroll
			for (int j = 0; j < 16; j++)
				x[j] += __mul24(fneg, prow[16 * w + j]);
			used = used || lane == pr;
			if (tid == 0) {
				sigma[c] = pr;
				prow_of[pr] = c;
			}
			if ((jc & 3) == 3) {
#pragma unroll
				for (int j = 0; j < 16; j++)
					x[j] = gj_reduce(x[j], negp, invp);
			}
		}
	}
	__syncthreads();
	// X[i][c] = G[i][sigma(c)] with G R = (permutation): Ginv[s][r] = G[sigma(s)][r], Montgomery form
	{
		const int s = prow_of[lane];
#pragma unroll
		for (int j = 0; j < 16; j++) {
			int v = x[j];          // (reduced at jc == 15)
			v = (v < 0) ? v + p : v;
			v = (v >= p) ? v - p : v;
			g.Ginv[s * NB + sigma[16 * w + j]] = montmul((uint32_t) v, F.r2, F);
		}
	}
	const int base = *g.rank;
	__syncthreads();
	if (tid < NB) {
		const int s = prow_of[tid];
		g.gamma[tid] = tid;
		g.rho[tid] = s_rows[tid];
		g.cand_pivot[tid] = s;
		g.is_pivot_row[s_rows[tid]] = 1;
		g.pivrow[base + s] = s_rows[tid];
		g.pivcol[base + s] = g.c0 + s;
	}
	if (tid == 0) {
		*g.rank = base + NB;
		*g.knew = NB;
		*g.gj_done = 1;
		g.try_state[0] = 0;
	}
}

__global__ __launch_bounds__(256) void rref_try_inverse(BlockGjArgs g)
{
	// (Tried: s_setprio 3 here, in rref_lookahead and in rref_mult_gather -- the chain shares its SIMDs with the far update of the
	//  super-panel before, twelve waves of matrix instructions per CU, and a try takes 75 us instead of 44 beside them.  No
	//  difference: 7.54-7.66 against 7.54-7.84 ms.)
	try_inverse_body(g);
	handoff_signal(g.done_word);          // (every thread of the workgroup leaves the body the same way)
}

// (Tried: the same elimination BLK = 4 columns per round -- every wave factors the 64 x 4 block by itself, two barriers
//  per round instead of two per column.  115 us against 88 us per panel: the 64 x 128 x 64 elimination is ~5 M
//  lane-instructions, i.e. bound by the instruction issue of ONE compute unit, not by its 128 barriers.)

// M[i][r] = -sum_s A[i, c0 + gamma_s] Ginv[s][r]  (+ Ginv[s][r] on the row that became pivot s), stored where the
// update kernels read it: P[(NB + r) * n + i].  Thread = (row, 4 consecutive r).
// signed base-256 digits of the balanced representative (see the matrix-core update below)
__device__ __forceinline__ void split_digits(uint32_t v, const MontDev &F, int &hi, int &lo)
{
	int b = (v > F.half) ? (int) v - (int) F.p : (int) v;
	lo = ((b + 128) & 255) - 128;
	hi = (b - lo) >> 8;
}

// SMALL16 (p < 2^16): plain 24-bit products summed in 64 bits, one reduction per multiplier.
struct MultArgs {
	const uint32_t *A;
	int64_t ld;
	int n, m, c0;
	const uint32_t *Ginv;
	const int *gamma, *knew;
	uint32_t *P;
	MontDev F;
	const int *rho, *cand_pivot;
	signed char *Mh, *Ml;
	uint32_t *Zblk;               // M as 64 columns of Z (residues), or null
	int64_t ldz;
	const int *abort;
	// the proof of a mode-2 try (BlockGjArgs): not null = the rows' flags (0 = no pivot, alive); after the step every such row
	// must be zero on the 64 columns of the panel, else *abort_w = 1 + abort_value (SMALL16 only)
	const int *verify_flags;
	int *abort_w;
	int abort_value;
	const int *wait_word;         // rref_mult_gather: the try of this panel counts itself in here (null: ordered by the stream)
	const int *wait_word2;        // ... and before it ends: the workgroups of the lookahead of this panel
	int wait_value2;
};

template <bool SMALL16> __device__ __forceinline__ void multipliers_body(const int bx, const MultArgs &g)
{
	const uint32_t *A = g.A;
	const int64_t ld = g.ld;
	const int n = g.n, m = g.m, c0 = g.c0;
	const uint32_t *Ginv = g.Ginv;
	const int *gamma = g.gamma, *knew = g.knew, *rho = g.rho, *cand_pivot = g.cand_pivot;
	uint32_t *P = g.P, *Zblk = g.Zblk;
	const MontDev F = g.F;
	signed char *Mh = g.Mh, *Ml = g.Ml;
	const int64_t ldz = g.ldz;
	if (g.abort != nullptr && *g.abort != 0)
		return;
	__shared__ uint32_t sG[NB][NB + 1];
	__shared__ int sgam[NB], srho[NB], spiv[NB];
	const int k = *knew;
	const int tid = threadIdx.x;
	if (k == 0) {
		// no pivot in this panel: all-zero digit planes (the update kernels multiply every set)
		if (SMALL16 && g.verify_flags != nullptr) {
			// (a mode-2 try that found nothing: right if no row without a pivot has an entry on the panel)
			const int i = bx * 64 + (tid & 63), q = tid >> 6;
			bool bad = false;
			if (i < n && g.verify_flags[i] == 0)
				for (int u = 0; u < 16; u++)
					bad = bad || (c0 + q * 16 + u < m && A[(int64_t) i * ld + c0 + q * 16 + u] != 0);
			if (bad)
				*g.abort_w = 1 + g.abort_value;
		}
		if (Mh != nullptr) {
			const int i = bx * 64 + (tid & 63), q = tid >> 6;
			if (i < n) {
				*reinterpret_cast<int4 *>(Mh + (int64_t) i * 64 + q * 16) = make_int4(0, 0, 0, 0);
				*reinterpret_cast<int4 *>(Ml + (int64_t) i * 64 + q * 16) = make_int4(0, 0, 0, 0);
				if (Zblk != nullptr)
					for (int u = 0; u < 16; u++)
						Zblk[(int64_t) i * ldz + q * 16 + u] = 0u;
			}
		}
		return;
	}
	if (tid < NB) {
		srho[tid] = (tid < k) ? rho[tid] : -1;
		spiv[tid] = (tid < k) ? cand_pivot[tid] : -1;
	}
	if (tid < NB)
		sgam[tid] = (tid < k) ? gamma[tid] : 0;
	// Ginv and the panel entries of the workgroup's 64 rows (read row by row, 256 contiguous bytes each): all 32 loads of a
	// thread are issued before the first store to LDS (written as two plain copy loops, every load was waited for on its own)
	__shared__ uint32_t tile[64][NB + 1];
	{
		uint32_t gv[NB * NB / 256], tv[64 * NB / 256];
#pragma unroll
		for (int u = 0; u < NB * NB / 256; u++) {
			const int t = tid + 256 * u;
			gv[u] = (t / NB < k && t % NB < k) ? Ginv[t] : 0u;
		}
#pragma unroll
		for (int u = 0; u < 64 * NB / 256; u++) {
			const int t = tid + 256 * u;
			const int irow = bx * 64 + t / NB, cc = t % NB;
			tv[u] = (irow < n && c0 + cc < m) ? A[(int64_t) irow * ld + c0 + cc] : 0u;
		}
#pragma unroll
		for (int u = 0; u < NB * NB / 256; u++) {
			const int t = tid + 256 * u;
			sG[t / NB][t % NB] = (SMALL16 && gv[u] != 0) ? montmul(gv[u], 1u, F) : gv[u];   // (plain / Montgomery form)
		}
#pragma unroll
		for (int u = 0; u < 64 * NB / 256; u++) {
			const int t = tid + 256 * u;
			tile[t / NB][t % NB] = tv[u];
		}
	}
	__syncthreads();
	// 64 rows per workgroup: thread (tid & 63) = row, (tid >> 6) = quarter of the r range
	const int i = bx * 64 + (tid & 63);
	const int q = tid >> 6;
	const bool verify = SMALL16 && g.verify_flags != nullptr;          // (uniform)
	if (i >= n && !verify)
		return;
	const bool active = i < n;
	uint32_t acc[16];
#pragma unroll
	for (int u = 0; u < 16; u++)
		acc[u] = 0;
	if constexpr (SMALL16) {
		unsigned long long wide[16];
#pragma unroll
		for (int u = 0; u < 16; u++)
			wide[u] = 0;
		for (int s = 0; s < k; s++) {
			const uint32_t a = tile[tid & 63][sgam[s]];
			if (a == 0)
				continue;
#pragma unroll
			for (int u = 0; u < 16; u++)
				wide[u] += (uint32_t) __umul24(a, sG[s][q * 16 + u]);          // < 2^32 each (HIP declares __umul24 as int), at most 64 of them
		}
#pragma unroll
		for (int u = 0; u < 16; u++)
			acc[u] = reduce_sum(wide[u], F);
	} else {
		for (int s = 0; s < k; s++) {
			const uint32_t a = tile[tid & 63][sgam[s]];
			if (a == 0)
				continue;
#pragma unroll
			for (int u = 0; u < 16; u++) {
				uint32_t v = acc[u] + montmul(a, sG[s][q * 16 + u], F);
				acc[u] = (v >= F.p || v < acc[u]) ? v - F.p : v;
			}
		}
	}
	// M = -acc; the row that became pivot s gets + Ginv[s][.] (see the definition of T above)
	int my_pivot = -1;
	for (int r = 0; r < k; r++)
		my_pivot = (active && srho[r] == i) ? spiv[r] : my_pivot;
	uint32_t mine[16];
	unsigned int wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
#pragma unroll
	for (int u = 0; u < 16; u++) {
		const int r = q * 16 + u;
		uint32_t mval = (acc[u] == 0) ? 0u : F.p - acc[u];
		if (my_pivot >= 0) {
			const uint32_t g = SMALL16 ? sG[my_pivot][r] : montmul(sG[my_pivot][r], 1u, F);
			mval += g;
			mval = (mval >= F.p || mval < g) ? mval - F.p : mval;
		}
		if (r < k && active)
			P[(int64_t) (NB + r) * n + i] = mval;
		if (r >= k)
			mval = 0;
		mine[u] = mval;
		if (Zblk != nullptr && active)
			Zblk[(int64_t) i * ldz + r] = SMALL16 ? mval : 0u;          // (only the matrix-core path, p < 2^16, keeps Z)
		int hi, lo;
		split_digits(mval, F, hi, lo);
		wh[u >> 2] |= (unsigned int) (hi & 255) << (8 * (u & 3));
		wl[u >> 2] |= (unsigned int) (lo & 255) << (8 * (u & 3));
	}
	if (Mh != nullptr && active) {
		*reinterpret_cast<int4 *>(Mh + (int64_t) i * 64 + q * 16) = make_int4((int) wh[0], (int) wh[1], (int) wh[2], (int) wh[3]);
		*reinterpret_cast<int4 *>(Ml + (int64_t) i * 64 + q * 16) = make_int4((int) wl[0], (int) wl[1], (int) wl[2], (int) wl[3]);
	}
	if constexpr (SMALL16) {
		if (verify) {
			// the proof of a mode-2 try: new row = old row + M[i] (old pivot rows), on the 64 columns of the panel; sG is done with
			// and becomes the multipliers of the 64 rows, the old pivot rows of the panel come in beside it
			__shared__ uint32_t sR[NB][NB + 1];
			__syncthreads();
#pragma unroll
			for (int u = 0; u < 16; u++)
				sG[tid & 63][q * 16 + u] = mine[u];
			for (int t = tid; t < NB * NB; t += 256) {
				const int tt = t / NB, cc = t % NB;
				sR[tt][cc] = (tt < k && c0 + cc < m) ? A[(int64_t) srho[tt] * ld + c0 + cc] : 0u;
			}
			__syncthreads();
			bool bad = false;
			if (active && g.verify_flags[i] == 0) {
#pragma unroll 4
				for (int u = 0; u < 16; u++) {
					const int cc = q * 16 + u;
					unsigned long long w = tile[tid & 63][cc];
					for (int t = 0; t < k; t++)
						w += (unsigned long long) sG[tid & 63][t] * sR[t][cc];          // < 2^32 each, at most 64 of them
					bad = bad || reduce_sum(w, F) != 0;
				}
			}
			if (bad)
				*g.abort_w = 1 + g.abort_value;
		}
	}
}

template <bool SMALL16> __global__ __launch_bounds__(256) void rref_multipliers(MultArgs g)
{
	multipliers_body<SMALL16>((int) blockIdx.x, g);
}


// B[t, :] = A[rho[t], c1:]  (old values of the new pivot rows), k x mr, row-major ld = mr
__global__ void rref_gather_pivot_rows(const uint32_t *A, int64_t ld, int c1, int mr, const int *rho, const int *knew,
                                       uint32_t *B)
{
	const int k = *knew;
	const int64_t total = (int64_t) k * mr;
	for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t) gridDim.x * blockDim.x) {
		const int row = (int) (t / mr), col = (int) (t % mr);
		B[t] = A[(int64_t) rho[row] * ld + c1 + col];
	}
}


// ---- one panel ahead ---------------------------------------------------------------------------------------------------
// The chain of an optimistic super-panel is try(i) -> multipliers(i) -> update(i) -> try(i + 1), three dependent launches per
// panel -- and all try(i + 1) reads of update(i) is ONE 64 x 64 tile: the next panel's columns of its 64 candidate rows.
// rref_lookahead forms that tile by itself, from what try(i) left (Ginv, the pivot rows rho) and from A as update(i - 1) left it:
//     M = -A[cand, panel i] Ginv            (the candidates hold no pivot: no +Ginv term -- multipliers_body)
//     X = A[cand, panel i + 1] + M A[rho, panel i + 1]
// so that try(i + 1) runs on its own stream BESIDE multipliers(i) and update(i), which bring A itself up to date.  It also picks
// the candidates (what rref_try_inverse does for itself otherwise).  One workgroup; p < 2^16, a full panel (k = 64, gamma = identity).
struct LookArgs {
	const uint32_t *A;
	int64_t ld;
	int n, c0;                    // panel i starts at c0, panel i + 1 at c0 + NB
	const uint32_t *Ginv;         // of panel i (Montgomery form)
	const int *rho;               // pivot rows of panel i, by candidate (column r of M multiplies row rho[r])
	const int *knew;
	const int *flags;
	int *hint, *out_rows, *out_count;
	uint32_t *alt;                // NB x NB: the tile of the next try
	const int *abort;
	const int *wait_word;         // the update of the panel before counts its workgroups in here ...
	int wait_value;               // ... all of them
	int *done_word;               // this kernel's workgroups count themselves in
	MontDev F;
};

constexpr int LOOK_WGS = 16, LOOK_COLS = NB / LOOK_WGS;          // workgroups of rref_lookahead, columns of the tile each of them forms

__device__ __forceinline__ void lookahead_body(const LookArgs &g);

__global__ __launch_bounds__(256) void rref_lookahead(LookArgs g)
{
	lookahead_body(g);
	handoff_signal(g.done_word);
}

__device__ __forceinline__ void lookahead_body(const LookArgs &g)
{
	// X = A[cand, next] - A[cand, panel] (Ginv A[rho, next]): workgroup b forms LOOK_COLS columns of the tile -- first
	// V = Ginv A[rho, its columns] (64 x 4), then its columns of X -- and every workgroup picks the candidates for itself
	// (the same rows: a pure function of the flags); workgroup 0 alone writes them out.
	__shared__ uint32_t sA[NB][NB + 1], sG[NB][NB + 1];
	__shared__ uint32_t sB[NB][LOOK_COLS], sV[NB][LOOK_COLS];
	__shared__ int srow[NB];
	__shared__ int s_found, s_hint;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int col0 = (int) blockIdx.x * LOOK_COLS;
	const MontDev F = g.F;
	if (g.abort != nullptr && *g.abort != 0)
		return;
	handoff_wait(g.wait_word, g.wait_value);          // (A as the update of the panel before left it)
	if (tid == 0)
		s_hint = *g.hint;
	__syncthreads();
	if (w == 0) {
		const int found = pick_candidates(g.flags, g.n, &s_hint, nullptr, nullptr, srow, lane);
		if (lane == 0)
			s_found = found;
	}
	__syncthreads();
	if (blockIdx.x == 0) {
		if (tid < NB)
			g.out_rows[tid] = srow[tid];
		if (tid == 0) {
			*g.out_count = s_found;
			*g.hint = s_hint;
		}
	}
	if (s_found < NB || *g.knew != NB)
		return;                          // (the try that follows gives up by itself: fewer than 64 candidates)
	{
		// all loads of a thread in flight together: Ginv (row-major), the candidates' panel entries, the pivot rows' and the
		// candidates' entries on this workgroup's columns of the next panel
		uint32_t gv[NB * NB / 256], av[NB * NB / 256];
#pragma unroll
		for (int u = 0; u < NB * NB / 256; u++) {
			const int t = tid + 256 * u;
			gv[u] = g.Ginv[t];
			av[u] = g.A[(int64_t) srow[t / NB] * g.ld + g.c0 + t % NB];
		}
		const int rr = tid / LOOK_COLS, cc = tid % LOOK_COLS;          // (256 threads = 64 rows x 4 columns)
		const uint32_t bv = g.A[(int64_t) g.rho[rr] * g.ld + g.c0 + NB + col0 + cc];
#pragma unroll
		for (int u = 0; u < NB * NB / 256; u++) {
			const int t = tid + 256 * u;
			sG[t / NB][t % NB] = (gv[u] != 0) ? montmul(gv[u], 1u, F) : 0u;          // plain residues
			sA[t / NB][t % NB] = av[u];
		}
		sB[rr][cc] = bv;
	}
	__syncthreads();
	const int i = tid / LOOK_COLS, c = tid % LOOK_COLS;
	{
		// V[s][c] = sum_r Ginv[s][r] A[rho[r]][next column c]  (column r of the multipliers multiplies the row that is candidate r)
		unsigned long long acc = 0;
#pragma unroll 8
		for (int r = 0; r < NB; r++)
			acc += (uint32_t) __umul24(sG[i][r], sB[r][c]);          // < 2^32 each, 64 of them
		sV[i][c] = reduce_sum(acc, F);
	}
	__syncthreads();
	{
		unsigned long long acc = 0;
#pragma unroll 8
		for (int sidx = 0; sidx < NB; sidx++)
			acc += (uint32_t) __umul24(sA[i][sidx], sV[sidx][c]);
		const uint32_t sub = reduce_sum(acc, F);
		const uint32_t old = g.A[(int64_t) srow[i] * g.ld + g.c0 + NB + col0 + c];
		g.alt[i * NB + col0 + c] = (old >= sub) ? old - sub : old + F.p - sub;
	}
}

// ---- trailing update, generic prime: C[i, j] += sum_t M[i, t] B[t, j]  (64-bit VALU) ----
constexpr int GT_ROWS = 64, GT_COLS = 64;

__global__ __launch_bounds__(256) void rref_update_valu(uint32_t *A, int64_t ld, int n, int c1, int mr, const uint32_t *P,
                                                        const uint32_t *B, const int *knew, MontDev F)
{
	__shared__ uint32_t sM[GT_ROWS][NB + 1];
	__shared__ uint32_t sB[NB][GT_COLS + 1];
	const int k = *knew;
	if (k == 0)
		return;
	const int tid = threadIdx.x;
	const int row0 = blockIdx.y * GT_ROWS, col0 = blockIdx.x * GT_COLS;
	for (int t = tid; t < GT_ROWS * NB; t += 256) {
		const int cc = t / GT_ROWS, rr = t % GT_ROWS;       // consecutive threads: consecutive rows of a column of M
		const int i = row0 + rr;
		sM[rr][cc] = (i < n && cc < k) ? P[(int64_t) (NB + cc) * n + i] : 0u;
	}
	for (int t = tid; t < NB * GT_COLS; t += 256) {
		const int rr = t / GT_COLS, cc = t % GT_COLS;
		const int j = col0 + cc;
		sB[rr][cc] = (rr < k && j < mr) ? B[(int64_t) rr * mr + j] : 0u;
	}
	__syncthreads();
	const int tx = tid % 16, ty = tid / 16;       // 16 x 16 threads, 4 x 4 outputs each
	const bool small = F.p < 65536u;
	for (int a = 0; a < 4; a++)
		for (int b = 0; b < 4; b++) {
			const int rr = ty * 4 + a, cc = tx * 4 + b;
			const int i = row0 + rr, j = col0 + cc;
			if (i >= n || j >= mr)
				continue;
			unsigned long long acc = 0;
			if (small) {
				for (int t = 0; t < k; t++)
					acc += (unsigned long long) (sM[rr][t] * sB[t][cc]);     // < 2^32 each
			} else {
				for (int t = 0; t < k; t++)
					acc += mulmod(sM[rr][t], sB[t][cc], F);
			}
			acc = reduce_sum(acc, F);
			uint32_t *dst = A + (int64_t) i * ld + c1 + j;
			uint32_t s = *dst + (uint32_t) acc;
			if (s < (uint32_t) acc || s >= F.p)
				s -= F.p;
			*dst = s;
		}
}

// ---- trailing update on the matrix cores, p < 2^16 ----
// Digits: v in [0,p) -> balanced b = v or v - p in (-p/2, p/2]; b = hi * 256 + lo with lo in
// [-128, 127], |hi| <= 128 (hi = 128 cannot occur for p < 2^16: |b| < 32768 -> hi in [-128, 127]).
// One wave computes a 32 x 32 tile of C with v_mfma_i32_32x32x32_i8 over K = 64 (two MFMAs per
// digit pair, four digit pairs).  A-operand lane map (i8, 32x32x32): lane l holds row (l & 31),
// k = 16 * (l >> 5) ... wait for the exact map see below: 8 consecutive k per lane half-block.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));


// Workgroup = 4 waves = a 64 x 64 tile of C (2 x 2 wave tiles of 32 x 32).  K = NB = 64.
// LDS holds the digit planes of M (64 x 64) and B (64 x 64) as int8.
__global__ __launch_bounds__(256) void rref_update_mfma(uint32_t *A, int64_t ld, int n, int c1, int mr, const uint32_t *P,
                                                        const uint32_t *B, const int *knew, MontDev F)
{
	__shared__ __attribute__((aligned(16))) signed char Mhi[64][64 + 16], Mlo[64][64 + 16];   // [row][k]
	__shared__ __attribute__((aligned(16))) signed char Bhi[64][64 + 16], Blo[64][64 + 16];   // [col][k]  (transposed)
	const int k = *knew;
	if (k == 0)
		return;
	const int tid = threadIdx.x;
	const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
	for (int t = tid; t < 64 * 64; t += 256) {
		const int kk = t / 64, rr = t % 64;                 // consecutive threads: consecutive rows of a column of M
		const int i = row0 + rr;
		const uint32_t v = (i < n && kk < k) ? P[(int64_t) (NB + kk) * n + i] : 0u;
		int hi, lo;
		split_digits(v, F, hi, lo);
		Mhi[rr][kk] = (signed char) hi;
		Mlo[rr][kk] = (signed char) lo;
	}
	for (int t = tid; t < 64 * 64; t += 256) {
		const int kk = t / 64, cc = t % 64;          // coalesced read of B rows
		const int j = col0 + cc;
		const uint32_t v = (kk < k && j < mr) ? B[(int64_t) kk * mr + j] : 0u;
		int hi, lo;
		split_digits(v, F, hi, lo);
		Bhi[cc][kk] = (signed char) hi;
		Blo[cc][kk] = (signed char) lo;
	}
	__syncthreads();
	const int wave = tid >> 6, lane = tid & 63;
	const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;      // this wave's 32 x 32 tile
	// i8 32x32x32: lane l supplies A[row = l & 31][k = 16 * (l >> 5) + 0..15] and
	// B[k = 16 * (l >> 5) + 0..15][col = l & 31] as 16 bytes (4 VGPRs)
	const int rsel = lane & 31, khalf = (lane >> 5) * 16;
	v16i acc_hh = {0}, acc_hl = {0}, acc_lh = {0}, acc_ll = {0};
#pragma unroll
	for (int ks = 0; ks < 64; ks += 32) {
		const v4i a_hi = *reinterpret_cast<const v4i *>(&Mhi[wr + rsel][ks + khalf]);
		const v4i a_lo = *reinterpret_cast<const v4i *>(&Mlo[wr + rsel][ks + khalf]);
		const v4i b_hi = *reinterpret_cast<const v4i *>(&Bhi[wc + rsel][ks + khalf]);
		const v4i b_lo = *reinterpret_cast<const v4i *>(&Blo[wc + rsel][ks + khalf]);
		acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_hi, b_hi, acc_hh, 0, 0, 0);
		acc_hl = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_hi, b_lo, acc_hl, 0, 0, 0);
		acc_lh = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_lo, b_hi, acc_lh, 0, 0, 0);
		acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_lo, b_lo, acc_ll, 0, 0, 0);
	}
	// C/D map of the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
	// |sum| <= 64 * (p/2)^2 < p * 2^26 for p < 2^16: shift by that multiple of p to stay non-negative
	const long long offset = (long long) F.p << 26;
#pragma unroll
	for (int reg = 0; reg < 16; reg++) {
		const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
		const int cc = lane & 31;
		const int i = row0 + wr + rr, j = col0 + wc + cc;
		if (i >= n || j >= mr)
			continue;
		const long long s = (long long) acc_hh[reg] * 65536 + ((long long) acc_hl[reg] + (long long) acc_lh[reg]) * 256 +
		                    (long long) acc_ll[reg] + offset;
		const uint32_t mred = reduce_sum((unsigned long long) s, F);
		uint32_t *dst = A + (int64_t) i * ld + c1 + j;
		uint32_t sum = *dst + mred;
		if (sum >= F.p)
			sum -= F.p;
		*dst = sum;
	}
}

// out[0] |= 1 when a row without pivot has a non-zero entry in columns >= c_from (when none has, the echelon form
// is complete: what is left of the block is zero)
__global__ __launch_bounds__(256) void rref_free_nonzero(const uint32_t *A, int64_t ld, int n, int m, int c_from, const int *flags, int *out)
{
	const int lane = threadIdx.x & 63;
	const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
	for (int i = wave; i < n; i += nwaves) {
		if (flags[i] != 0)
			continue;
		if (*(volatile int *) out != 0)
			return;
		// (a row that is not zero usually says so in its first few hundred entries)
		for (int j0 = c_from; j0 < m; j0 += 256) {
			bool nz = false;
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const int j = j0 + 64 * u + lane;
				nz |= j < m && A[(int64_t) i * ld + j] != 0;
			}
			if (__ballot(nz) != 0) {
				if (lane == 0)
					atomicOr(out, 1);
				return;
			}
		}
	}
}

// ---- super-panels: the trailing update of several panels in one pass (K up to 64 * SPW) ----
// Consecutive panels T_1 .. T_S (T_i = I + M_i E_i^T, E_i^T = the rows rho_i) are factored with the columns of their
// super-panel kept up to date (K = 64 updates of a few hundred columns); beyond the super-panel
//     T_S .. T_1 C = C + sum_i M'_i C[rho_i],        M'_i = T_S .. T_{i+1} M_i,
// with the ORIGINAL rows C[rho_i]: the multipliers M_i of the earlier panels are kept as one more block of columns (Z)
// that every later panel of the super-panel updates like the matrix itself (matrix-core path).  C is read and
// written once per super-panel and reduced mod p once per element.  (The VALU path for larger primes forms
// B_i = rows rho_i of (T_{i-1} .. T_1 C) = C[rho_i] + sum_{j<i} M_j[rho_i, :] B_j instead: rref_trailing_B.)
constexpr int MAXSETS = 8;
struct UpdSets {
	const uint32_t *P[MAXSETS];     // multipliers of set s: P[s][(NB + t) * n + i]
	const uint32_t *B[MAXSETS];     // k_s x mr, row-major
	const int *knew[MAXSETS];
	// the same as signed base-256 digits (matrix-core path): M8[plane][i][64], B8[plane][col][64], zero beyond k_s
	const signed char *Mh[MAXSETS], *Ml[MAXSETS], *Bh[MAXSETS], *Bl[MAXSETS];
	int nsets;
	// a second block of columns updated by the same launch (the multipliers of the earlier panels of the super-panel, see
	// the driver): column tiles tiles1 .. of the grid address Z2 (row stride ld2, mr2 columns); their B planes follow
	// the first block's at column tiles1 * 64
	uint32_t *Z2;
	int64_t ld2;
	int tiles1, mr2;
	const int *abort;               // optimistic super-panels: not null and raised -> the launch does nothing
	const int *wait_word;           // rref_update_mfma_multi: the lookahead of this panel has read A when this word holds wait_value
	int wait_value;
	int *done_word;                 // ... and its workgroups count themselves in here
};


// B[t, :] = A[rho[t], c1 : c1 + mr] (old values of the new pivot rows) as digit planes, one thread per column; the
// 32-bit copy is written too when B is not null (VALU update)
struct GatherArgs {
	const uint32_t *A;
	int64_t ld;
	int c1, mr;
	const int *rho, *knew;
	signed char *Bh, *Bl;
	uint32_t *B;
	MontDev F;
	const uint32_t *Z2;
	int64_t ld2;
	int tiles1, mr2;
	const int *abort;
};

__device__ __forceinline__ void gather_split_body(const int bx, const GatherArgs &g)
{
	__shared__ int srho[NB];
	const uint32_t *A = g.A, *Z2 = g.Z2;
	const int64_t ld = g.ld, ld2 = g.ld2;
	const int c1 = g.c1, mr = g.mr, tiles1 = g.tiles1, mr2 = g.mr2;
	const int *rho = g.rho, *knew = g.knew;
	signed char *Bh = g.Bh, *Bl = g.Bl;
	uint32_t *B = g.B;
	const MontDev F = g.F;
	if (g.abort != nullptr && *g.abort != 0)
		return;
	const int k = *knew;
	if (threadIdx.x < NB)
		srho[threadIdx.x] = (threadIdx.x < k) ? rho[threadIdx.x] : 0;
	__syncthreads();
	int col = bx * 256 + threadIdx.x;
	const uint32_t *src = A + c1;
	int64_t lds = ld;
	int plane_col = col;
	if (Z2 != nullptr && col >= tiles1 * 64) {          // second block of columns (see UpdSets)
		col -= tiles1 * 64;
		if (col >= mr2)
			return;
		src = Z2;
		lds = ld2;
	} else if (col >= mr) {
		return;
	}
	int4 *dh = reinterpret_cast<int4 *>(Bh + (int64_t) plane_col * 64), *dl = reinterpret_cast<int4 *>(Bl + (int64_t) plane_col * 64);
#pragma unroll
	for (int part = 0; part < 4; part++) {
		unsigned int wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
#pragma unroll
		for (int b = 0; b < 16; b++) {
			const int kk = part * 16 + b;
			const uint32_t v = (kk < k) ? src[(int64_t) srho[kk] * lds + col] : 0u;
			if (B != nullptr && kk < k)
				B[(int64_t) kk * mr + col] = v;
			int hi, lo;
			split_digits(v, F, hi, lo);
			wh[b >> 2] |= (unsigned int) (hi & 255) << (8 * (b & 3));
			wl[b >> 2] |= (unsigned int) (lo & 255) << (8 * (b & 3));
		}
		dh[part] = make_int4((int) wh[0], (int) wh[1], (int) wh[2], (int) wh[3]);
		dl[part] = make_int4((int) wl[0], (int) wl[1], (int) wl[2], (int) wl[3]);
	}
}

__global__ __launch_bounds__(256) void rref_gather_split_B(GatherArgs g)
{
	gather_split_body((int) blockIdx.x, g);
}

// the same for every set of a super-panel in one launch (blockIdx.y = set)
struct GatherSets {
	const int *rho[MAXSETS], *knew[MAXSETS];
	signed char *Bh[MAXSETS], *Bl[MAXSETS];
};

__global__ __launch_bounds__(256) void rref_gather_split_sets(const uint32_t *A, int64_t ld, int c1, int mr, GatherSets gs, MontDev F)
{
	GatherArgs g{A, ld, c1, mr, nullptr, nullptr, nullptr, nullptr, nullptr, F, nullptr, 0, 0, 0, nullptr};
#pragma unroll
	for (int s = 0; s < MAXSETS; s++)          // (static indices into the kernel arguments)
		if (s == (int) blockIdx.y) {
			g.rho = gs.rho[s];
			g.knew = gs.knew[s];
			g.Bh = gs.Bh[s];
			g.Bl = gs.Bl[s];
		}
	gather_split_body((int) blockIdx.x, g);
}

// both in one launch (they depend on the same Gauss-Jordan block and not on each other; a launch in a chain of dependent
// launches costs ~8 us of idle device): workgroups 0 .. nmult - 1 compute multipliers, the others gather
template <bool SMALL16> __global__ __launch_bounds__(256) void rref_mult_gather(MultArgs a, GatherArgs b, int nmult)
{
	handoff_wait(a.wait_word, 1);          // (the try of this panel, on the other stream)
	if ((int) blockIdx.x < nmult)
		multipliers_body<SMALL16>((int) blockIdx.x, a);
	else
		gather_split_body((int) blockIdx.x - nmult, b);
	// the update that follows this kernel on its stream rewrites what the lookahead of this panel reads: nobody leaves before the
	// lookahead is through (it started with this kernel and takes a quarter of its time: nobody waits here in practice, and the
	// update -- a grid that can fill the chip -- has no reason to poll)
	handoff_wait(a.wait_word2, a.wait_value2);
}

// digit planes of the accumulated multipliers Z[:, 64 s .. 64 s + 63] (set s), one thread per (row, 16 columns)
__global__ __launch_bounds__(256) void rref_split_Z(const uint32_t *Z, int64_t ldz, int n, int nsets, signed char *M8, int64_t set_stride, MontDev F)
{
	const int i = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
	const int s = blockIdx.y;
	if (i >= n || s >= nsets)
		return;
	signed char *Mh = M8 + (int64_t) s * set_stride, *Ml = Mh + (int64_t) n * 64;
	unsigned int wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
#pragma unroll
	for (int u = 0; u < 16; u++) {
		int hi, lo;
		split_digits(Z[(int64_t) i * ldz + 64 * s + q * 16 + u], F, hi, lo);
		wh[u >> 2] |= (unsigned int) (hi & 255) << (8 * (u & 3));
		wl[u >> 2] |= (unsigned int) (lo & 255) << (8 * (u & 3));
	}
	*reinterpret_cast<int4 *>(Mh + (int64_t) i * 64 + q * 16) = make_int4((int) wh[0], (int) wh[1], (int) wh[2], (int) wh[3]);
	*reinterpret_cast<int4 *>(Ml + (int64_t) i * 64 + q * 16) = make_int4((int) wl[0], (int) wl[1], (int) wl[2], (int) wl[3]);
}

// digits of B (k x mr, row-major with row stride ldb >= mr): one thread per column.  (The planes hold mr columns: with the
// stride as the bound, the columns of the padding of a matrix with ld > m were written past the end of the plane -- into the
// first columns of the next one, racing with their own writers.)
__global__ __launch_bounds__(256) void rref_split_B(const uint32_t *B, int64_t ldb, int mr, const int *knew, signed char *Bh, signed char *Bl, MontDev F)
{
	const int col = blockIdx.x * 256 + threadIdx.x;
	if (col >= mr)
		return;
	const int k = *knew;
	int4 *dh = reinterpret_cast<int4 *>(Bh + (int64_t) col * 64), *dl = reinterpret_cast<int4 *>(Bl + (int64_t) col * 64);
#pragma unroll
	for (int part = 0; part < 4; part++) {
		unsigned int wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
#pragma unroll
		for (int b = 0; b < 16; b++) {
			const int kk = part * 16 + b;
			const uint32_t v = (kk < k) ? B[(int64_t) kk * ldb + col] : 0u;
			int hi, lo;
			split_digits(v, F, hi, lo);
			wh[b >> 2] |= (unsigned int) (hi & 255) << (8 * (b & 3));
			wl[b >> 2] |= (unsigned int) (lo & 255) << (8 * (b & 3));
		}
		dh[part] = make_int4((int) wh[0], (int) wh[1], (int) wh[2], (int) wh[3]);
		dl[part] = make_int4((int) wl[0], (int) wl[1], (int) wl[2], (int) wl[3]);
	}
}

// Bt_i[t, :] = A[rho_i[t], c1:] + sum_{j < i} M_j[rho_i[t], :] B_j.   grid (column chunks, NB), 256 threads
template <bool SMALL16>
__global__ __launch_bounds__(256) void rref_trailing_B(const uint32_t *A, int64_t ld, int n, int c1, int mr, UpdSets S,
                                                       const int *rho_i, const int *knew_i, uint32_t *Bt, MontDev F)
{
	__shared__ uint32_t mrow[4][NB];
	const int t = blockIdx.y;
	if (t >= *knew_i)
		return;
	const int row = rho_i[t];
	const int tid = threadIdx.x;
	{
		const int j = tid >> 6, s = tid & 63;
		uint32_t v = 0;
		if (j < S.nsets && s < *S.knew[j]) {
			v = S.P[j][(int64_t) (NB + s) * n + row];
			if (!SMALL16)
				v = montmul(v, F.r2, F);          // Montgomery form: montmul(v, b) = v b
		}
		mrow[j][s] = v;
	}
	__syncthreads();
	const int col = blockIdx.x * 256 + tid;
	if (col >= mr)
		return;
	const uint32_t a = A[(int64_t) row * ld + c1 + col];
	if (SMALL16) {
		unsigned long long acc = a;                 // < 2^16 + 256 * 2^32
		for (int j = 0; j < S.nsets; j++) {
			const int kj = *S.knew[j];
			const uint32_t *Bj = S.B[j] + col;
			for (int s = 0; s < kj; s++)
				acc += (unsigned long long) mrow[j][s] * Bj[(int64_t) s * mr];
		}
		Bt[(int64_t) t * mr + col] = reduce_sum(acc, F);
	} else {
		uint32_t acc = a;
		for (int j = 0; j < S.nsets; j++) {
			const int kj = *S.knew[j];
			const uint32_t *Bj = S.B[j] + col;
			for (int s = 0; s < kj; s++) {
				const uint32_t v = acc + montmul(mrow[j][s], Bj[(int64_t) s * mr], F);
				acc = (v >= F.p || v < acc) ? v - F.p : v;
			}
		}
		Bt[(int64_t) t * mr + col] = acc;
	}
}

// C[:, c1:] += sum_s M_s B_s on the matrix cores (p <= 65279), 64 x 64 tile per workgroup, one epilogue.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void rref_update_mfma_multi(uint32_t *A, int64_t ld, int n, int c1, int mr, UpdSets S, MontDev F)
{
	__shared__ __attribute__((aligned(16))) signed char Mhi[64][64 + 16], Mlo[64][64 + 16];   // [row][k]
	__shared__ __attribute__((aligned(16))) signed char Bhi[64][64 + 16], Blo[64][64 + 16];   // [col][k]  (transposed)
	if (S.abort != nullptr && *S.abort != 0) {
		handoff_signal(S.done_word);
		return;
	}
	handoff_wait(S.wait_word, S.wait_value);
	const int tid = threadIdx.x;
	const int row0 = blockIdx.y * 64;
	int col0 = blockIdx.x * 64;
	const int bcol0 = col0;                       // column of this tile in the B planes
	if (S.Z2 != nullptr && (int) blockIdx.x >= S.tiles1) {          // second block of columns (see UpdSets)
		A = S.Z2;
		ld = S.ld2;
		c1 = 0;
		mr = S.mr2;
		col0 -= S.tiles1 * 64;
	}
	const int wave = tid >> 6, lane = tid & 63;
	const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;      // this wave's 32 x 32 tile
	const int rsel = lane & 31, khalf = (lane >> 5) * 16;
	v16i acc_hh = {0}, acc_hl = {0}, acc_lh = {0}, acc_ll = {0};
	// a workgroup is a chain of dependent loads (few workgroups fit on a CU): this thread's 16 entries of C are
	// fetched first, and the digit planes of set s + 1 while set s is multiplied.  Sets that found no pivot have
	// all-zero planes (rref_multipliers, rref_split_B): they are multiplied like the others rather than tested for.
	uint32_t cval[16];
#pragma unroll
	for (int reg = 0; reg < 16; reg++) {
		const int i = row0 + wr + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), j = col0 + wc + (lane & 31);
		cval[reg] = (i < n && j < mr) ? A[(int64_t) i * ld + c1 + j] : 0u;
	}
	// 64 rows (columns) x 64 digit bytes per plane: thread t moves 16 bytes of row (column) t / 4
	const int rr_t = tid >> 2, part = (tid & 3) * 16;
	const int i_t = row0 + rr_t, j_t = col0 + rr_t;
	const int jb_t = bcol0 + rr_t;
	const int4 zero = make_int4(0, 0, 0, 0);
	int4 mh, ml, bh, bl;
	auto fetch = [&](int s) {
		mh = (i_t < n) ? *reinterpret_cast<const int4 *>(S.Mh[s] + (int64_t) i_t * 64 + part) : zero;
		ml = (i_t < n) ? *reinterpret_cast<const int4 *>(S.Ml[s] + (int64_t) i_t * 64 + part) : zero;
		bh = (j_t < mr) ? *reinterpret_cast<const int4 *>(S.Bh[s] + (int64_t) jb_t * 64 + part) : zero;
		bl = (j_t < mr) ? *reinterpret_cast<const int4 *>(S.Bl[s] + (int64_t) jb_t * 64 + part) : zero;
	};
	fetch(0);
#pragma unroll
	for (int s = 0; s < MAXSETS; s++) {          // (static indices into the kernel arguments)
		if (s >= S.nsets)
			break;
		if (s > 0)
			__syncthreads();          // the previous set's tiles have been consumed
		*reinterpret_cast<int4 *>(&Mhi[rr_t][part]) = mh;
		*reinterpret_cast<int4 *>(&Mlo[rr_t][part]) = ml;
		*reinterpret_cast<int4 *>(&Bhi[rr_t][part]) = bh;
		*reinterpret_cast<int4 *>(&Blo[rr_t][part]) = bl;
		__syncthreads();
		if (s + 1 < MAXSETS && s + 1 < S.nsets)
			fetch(s + 1 < MAXSETS ? s + 1 : MAXSETS - 1);
#pragma unroll
		for (int ks = 0; ks < 64; ks += 32) {
			const v4i a_hi = *reinterpret_cast<const v4i *>(&Mhi[wr + rsel][ks + khalf]);
			const v4i a_lo = *reinterpret_cast<const v4i *>(&Mlo[wr + rsel][ks + khalf]);
			const v4i b_hi = *reinterpret_cast<const v4i *>(&Bhi[wc + rsel][ks + khalf]);
			const v4i b_lo = *reinterpret_cast<const v4i *>(&Blo[wc + rsel][ks + khalf]);
			acc_hh = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_hi, b_hi, acc_hh, 0, 0, 0);
			acc_hl = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_hi, b_lo, acc_hl, 0, 0, 0);
			acc_lh = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_lo, b_hi, acc_lh, 0, 0, 0);
			acc_ll = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_lo, b_lo, acc_ll, 0, 0, 0);
		}
	}
	// recombination and reduction in double precision: |digit sums| <= 512 * 128 * 128 = 2^23, so the recombined
	// value (< 2^40 in magnitude) and q * p are exact; the quotient estimate is off by at most one
	const double pd = (double) F.p, invp = 1.0 / pd;
#pragma unroll
	for (int reg = 0; reg < 16; reg++) {
		const int rr = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
		const int cc = lane & 31;
		const int i = row0 + wr + rr, j = col0 + wc + cc;
		if (i >= n || j >= mr)
			continue;
		const double sd = fma((double) acc_hh[reg], 65536.0, fma((double) (acc_hl[reg] + acc_lh[reg]), 256.0, (double) acc_ll[reg]));
		const double qd = floor(sd * invp);
		double rd = fma(-qd, pd, sd);
		rd = (rd < 0.0) ? rd + pd : rd;
		rd = (rd >= pd) ? rd - pd : rd;
		const uint32_t mred = (uint32_t) rd;
		uint32_t sum = cval[reg] + mred;
		if (sum >= F.p)
			sum -= F.p;
		A[(int64_t) i * ld + c1 + j] = sum;
	}
	handoff_signal(S.done_word);
}


// echelon rows to the top, in pivot-column order (rows that hold no pivot are zero after the
// full elimination): tmp[t, :] = A[pivrow[t], :], then copied back.
// (one workgroup walks whole rows, 16 bytes per thread when the rows allow it: the element-wise version spent its time
//  on 64-bit divisions)
__device__ __forceinline__ void copy_row(uint32_t *dst, const uint32_t *src, int m, bool zero)
{
	const bool vec = (m % 4) == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16) == 0;
	if (vec) {
		uint4 *d4 = reinterpret_cast<uint4 *>(dst);
		const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
		for (int j = threadIdx.x; j < m / 4; j += blockDim.x)
			d4[j] = zero ? make_uint4(0u, 0u, 0u, 0u) : s4[j];
	} else {
		for (int j = threadIdx.x; j < m; j += blockDim.x)
			dst[j] = zero ? 0u : src[j];
	}
}

// (a row that is already in its place -- the usual case: the first free row is the pivot row of its column with probability
//  1 - 1/p -- is neither copied out nor back)
__global__ __launch_bounds__(256) void rref_rows_to_tmp(const uint32_t *A, int64_t ld, int m, const int *pivrow, int rank, uint32_t *tmp)
{
	for (int t = blockIdx.x; t < rank; t += gridDim.x)
		if (pivrow[t] != t)
			copy_row(tmp + (int64_t) t * m, A + (int64_t) pivrow[t] * ld, m, false);
}

__global__ __launch_bounds__(256) void rref_tmp_to_rows(uint32_t *A, int64_t ld, int n, int m, int rank, const uint32_t *tmp, const int *pivrow)
{
	for (int i = blockIdx.x; i < n; i += gridDim.x)
		if (i >= rank || pivrow[i] != i)
			copy_row(A + (int64_t) i * ld, tmp + (int64_t) (i < rank ? i : 0) * m, m, i >= rank);
}

// ---- driver: everything resident on the device.  On return rows 0..rank-1 of A are the reduced
// echelon rows (pivot columns increasing, listed in d_pivcol), the other rows are zero. ----
int device_rref(int64_t prime, int n, int m, uint32_t *dA, int64_t ld, int *d_pivcol, hipStream_t stream, int use_mfma,
                float *ms_update)
{
	if (n == 0 || m == 0)
		return 0;
	const double t_entry = wtime();
	// work buffers are kept between calls (per host thread; SPASM_HIP_RREF_CACHE=0: allocated and freed every time): 28
	// hipMalloc / hipFree pairs and the half-gigabyte row buffer of the final permutation were 2 of the 11 ms of a
	// 4096 x 32768 block, and the dense finish calls this several times in a row
	struct Cache {
		int dev = -1;
		std::vector<std::pair<void *, size_t>> slots;
	};
	static thread_local Cache cache;
	const bool use_cache = sh::env_get("SPASM_HIP_RREF_CACHE") == nullptr || std::atoi(sh::env_get("SPASM_HIP_RREF_CACHE")) != 0;
	size_t next_slot = 0;
	{
		int dev = 0;
		HIP_CHECK(hipGetDevice(&dev));
		if (cache.dev != dev) {
			for (auto &sl : cache.slots)
				sh::big_free(sl.first);
			cache.slots.clear();
			cache.dev = dev;
		}
	}
	std::vector<void *> owned;
	auto ws_malloc = [&](void **ptr, size_t bytes) {
		if (!use_cache || bytes > ((size_t) 4 << 30)) {
			HIP_CHECK(sh::malloc_or_trim(ptr, bytes));
			owned.push_back(*ptr);
			return;
		}
		if (next_slot == cache.slots.size())
			cache.slots.push_back({nullptr, 0});
		auto &sl = cache.slots[next_slot++];
		if (sl.second < bytes) {
			sh::big_free(sl.first);
			sl.second = bytes + bytes / 4;
			HIP_CHECK(sh::malloc_or_trim(&sl.first, sl.second));
		}
		*ptr = sl.first;
	};
	auto ws_free = [&](void *ptr) {
		for (size_t t = 0; t < owned.size(); t++)
			if (owned[t] == ptr) {
				sh::big_free(ptr);
				owned[t] = nullptr;
				return;
			}
	};
	const Mont M = mont_setup(prime);
	const MontDev F = to_dev(M);
	uint32_t *P = nullptr, *B = nullptr;
	int *flags = nullptr, *pivrow = nullptr, *rank_d = nullptr, *knew = nullptr, *rho = nullptr;
	const int rmax = (n < m) ? n : m;
	ws_malloc((void **) &P, (size_t) n * PW * sizeof(uint32_t));
	ws_malloc((void **) &B, (size_t) NB * (size_t) m * sizeof(uint32_t));
	ws_malloc((void **) &flags, (size_t) n * sizeof(int));
	ws_malloc((void **) &pivrow, (size_t) rmax * sizeof(int) + 64);
	ws_malloc((void **) &rank_d, 64);
	ws_malloc((void **) &knew, 64);
	ws_malloc((void **) &rho, NB * sizeof(int));
	HIP_CHECK(hipMemsetAsync(flags, 0, (size_t) n * sizeof(int), stream));
	HIP_CHECK(hipMemsetAsync(rank_d, 0, 64, stream));
	// tall blocks: the panel step is spread over several workgroups (SPASM_HIP_COOP_ROWS rows and up)
	int coop_min_rows = 2048;
	if (const char *e = sh::env_get("SPASM_HIP_COOP_ROWS"))
		coop_min_rows = std::atoi(e);
	// panel step: tournament (default) or the column-by-column kernels (SPASM_HIP_RREF_PANEL=columns)
	bool tournament = true;
	const bool small_prime = prime < 46341;          // 2 p^2 < 2^32: the panel kernels use 24-bit multiplies
	if (const char *e = sh::env_get("SPASM_HIP_RREF_PANEL"))
		tournament = std::strcmp(e, "columns") != 0;
	int *candA = nullptr, *candB = nullptr, *free_count = nullptr, *gamma = nullptr, *cand_first = nullptr, *full_flag = nullptr, *first64 = nullptr, *cand_pivot = nullptr, *live_list = nullptr;
	uint32_t *Ginv = nullptr, *P4 = nullptr, *Bt4 = nullptr, *Zacc = nullptr;
	int *rho4 = nullptr, *knew4 = nullptr;
	signed char *M8 = nullptr, *B8 = nullptr;
	unsigned short *invtab = nullptr;
	size_t invtab_bytes = 0;
	if (tournament) {
		const size_t cand_len = (size_t) std::max(n, ((n + SEL_ROWS - 1) / SEL_ROWS) * NB) + NB;
		ws_malloc((void **) &candA, cand_len * sizeof(int));
		ws_malloc((void **) &candB, cand_len * sizeof(int));
		ws_malloc((void **) &free_count, 64);
		ws_malloc((void **) &gamma, 2 * NB * sizeof(int));
		ws_malloc((void **) &cand_first, NB * sizeof(int));
		ws_malloc((void **) &first64, NB * sizeof(int));
		ws_malloc((void **) &live_list, (size_t) n * sizeof(int));
		ws_malloc((void **) &cand_pivot, 2 * NB * sizeof(int));
		HIP_CHECK(hipMemsetAsync(free_count, 0, 64, stream));          // [4]: scan hint of rref_first_free
		ws_malloc((void **) &P4, (size_t) 2 * MAXSETS * (size_t) n * PW * sizeof(uint32_t));
		ws_malloc((void **) &Bt4, (size_t) 4 * (size_t) NB * (size_t) m * sizeof(uint32_t));
		ws_malloc((void **) &rho4, 2 * MAXSETS * NB * sizeof(int));
		if (small_prime) {
			invtab_bytes = ((size_t) prime * 2 + 15) / 16 * 16;
			ws_malloc((void **) &invtab, invtab_bytes + 64);
			hipLaunchKernelGGL(rref_inverse_table, dim3(((unsigned) prime + 255) / 256), dim3(256), 0, stream, invtab, F);
			static bool configured = false;
			if (!configured) {
				HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rref_block_gj<true, 8>),
				                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
				HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rref_try_inverse),
				                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
				configured = true;
			}
		}
		// digit planes: per set M (2 x n x 64) and trailing B (2 x m x 64); one more B pair for the super-panel's own columns
		// (M planes: one per panel of two super-panels -- the far update of the previous one may still read its own -- and
		//  as many again for the accumulated multipliers Z; B planes: one per set, one more for a panel's own update, which
		//  covers the rest of the super-panel and the Z columns)
		ws_malloc((void **) &M8, (size_t) 4 * MAXSETS * 2 * (size_t) n * 64);
		ws_malloc((void **) &B8, (size_t) MAXSETS * 2 * (size_t) m * 64 + (size_t) 2 * (2 * MAXSETS * NB + 64) * 64);
		ws_malloc((void **) &Zacc, (size_t) n * (size_t) (MAXSETS * NB) * sizeof(uint32_t));
		ws_malloc((void **) &knew4, 2 * MAXSETS * 16 * sizeof(int));
		ws_malloc((void **) &full_flag, 256);
		HIP_CHECK(hipMemsetAsync(full_flag, 0, 256, stream));          // [0] full, [4] gj_done, [8] try_state, [12] abort, [16 .. 16 + MAXSETS) abort by panel
		ws_malloc((void **) &Ginv, 2 * NB * NB * sizeof(uint32_t));
	}
	unsigned int *coop_barrier = nullptr;
	int *coop_cand = nullptr, *coop_err = nullptr;
	ws_malloc((void **) &coop_barrier, 64);
	ws_malloc((void **) &coop_cand, NB * sizeof(int));
	ws_malloc((void **) &coop_err, 64);
	HIP_CHECK(hipMemsetAsync(coop_err, 0, 64, stream));
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (ms_update != nullptr) {
		HIP_CHECK(hipEventCreate(&e0));
		HIP_CHECK(hipEventCreate(&e1));
	}
	float total_update = 0.f;
	const bool mfma_ok = use_mfma && prime <= 65279;      // two signed base-256 digits must fit int8
	hipStream_t stream2 = nullptr;
	hipEvent_t ev_near = nullptr, ev_far = nullptr;
	const bool two_streams = tournament && mfma_ok && !sh::env_get("SPASM_HIP_RREF_ONE_STREAM");
	if (two_streams) {
		HIP_CHECK(hipEventCreate(&ev_near));
		HIP_CHECK(hipEventCreate(&ev_far));
	}
	// one panel ahead (rref_lookahead): the tries of an optimistic super-panel on a stream of their own, beside the multipliers and
	// updates of the panel before.  Streams and events are kept between calls (per host thread).
	struct Ahead {
		int dev = -1;
		hipStream_t s_try = nullptr, s_far = nullptr;
		hipEvent_t ev_try[MAXSETS] = {}, ev_upd[MAXSETS] = {}, ev_look[MAXSETS] = {}, ev_start = nullptr;
	};
	static thread_local Ahead ahead;
	bool use_ahead = tournament && mfma_ok && prime < 65536 && (sh::env_get("SPASM_HIP_RREF_LOOKAHEAD") == nullptr || std::atoi(sh::env_get("SPASM_HIP_RREF_LOOKAHEAD")) != 0);
	uint32_t *alt_tile = nullptr;
	if (use_ahead || two_streams) {
		int dev = 0;
		HIP_CHECK(hipGetDevice(&dev));
		if (ahead.dev != dev) {
			// (Tried: disjoint compute units for the two streams (hipExtStreamCreateWithCUMask: sixteen for the tries, the rest for the far
			//  updates) -- the trace shows a try at 75-79 us instead of 44 whenever the far update of the super-panel before shares its
			//  compute unit.  The call went from 8.3 to 16 ms: masked queues are served far more slowly on this stack.  Plain streams.)
			HIP_CHECK(hipStreamCreateWithFlags(&ahead.s_far, hipStreamNonBlocking));
			HIP_CHECK(hipStreamCreateWithFlags(&ahead.s_try, hipStreamNonBlocking));
			for (int t = 0; t < MAXSETS; t++) {
				HIP_CHECK(hipEventCreateWithFlags(&ahead.ev_try[t], hipEventDisableTiming));
				HIP_CHECK(hipEventCreateWithFlags(&ahead.ev_upd[t], hipEventDisableTiming));
				HIP_CHECK(hipEventCreateWithFlags(&ahead.ev_look[t], hipEventDisableTiming));
			}
			HIP_CHECK(hipEventCreateWithFlags(&ahead.ev_start, hipEventDisableTiming));
			ahead.dev = dev;
		}
		if (use_ahead) {
			// the hand-offs need a kernel of each stream on the device at the same time: asked once per pair of streams (handoff_probe)
			struct Probed {
				hipStream_t main = nullptr, tries = nullptr;
				bool met = false, asked = false;
			};
			static thread_local Probed probed;
			if (!probed.asked || probed.main != stream || probed.tries != ahead.s_try) {
				int *pw = nullptr, met[2] = {0, 0};
				ws_malloc((void **) &pw, 4 * sizeof(int));
				HIP_CHECK(hipMemsetAsync(pw, 0, 4 * sizeof(int), stream));
				HIP_CHECK(hipEventRecord(ahead.ev_start, stream));
				HIP_CHECK(hipStreamWaitEvent(ahead.s_try, ahead.ev_start, 0));
				hipLaunchKernelGGL(handoff_probe, dim3(1), dim3(1), 0, stream, pw, 0, pw + 2);
				hipLaunchKernelGGL(handoff_probe, dim3(1), dim3(1), 0, ahead.s_try, pw, 1, pw + 2);
				HIP_CHECK(hipEventRecord(ahead.ev_look[0], ahead.s_try));
				HIP_CHECK(hipStreamWaitEvent(stream, ahead.ev_look[0], 0));
				HIP_CHECK(hipMemcpyAsync(met, pw + 2, sizeof(met), hipMemcpyDeviceToHost, stream));
				HIP_CHECK(hipStreamSynchronize(stream));
				ws_free(pw);
				probed.main = stream;
				probed.tries = ahead.s_try;
				probed.met = met[0] != 0 && met[1] != 0;
				probed.asked = true;
				if (!probed.met && sh::verbose() >= 1)
					fprintf(stderr, "[spasm_hip] dense RREF: kernels of two streams do not run side by side here (a tool that serialises launches?): no lookahead\n");
			}
			use_ahead = probed.met;
		}
		if (use_ahead)
			ws_malloc((void **) &alt_tile, (size_t) 2 * NB * NB * sizeof(uint32_t) + 3 * MAXSETS * sizeof(int));          // (two tiles, by parity of the panel; then the hand-off words)
		if (two_streams)
			stream2 = ahead.s_far;
	}
	auto timed = [&](auto &&launch) {
		if (ms_update != nullptr)
			HIP_CHECK(hipEventRecord(e0, stream));
		launch();
		if (ms_update != nullptr) {
			HIP_CHECK(hipEventRecord(e1, stream));
			HIP_CHECK(hipEventSynchronize(e1));
			float ms;
			HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
			total_update += ms;
		}
	};
	if (sh::env_get("SPASM_HIP_RREF_TIMING")) {
		HIP_CHECK(hipStreamSynchronize(stream));
		fprintf(stderr, "[rref timing] allocations + setup: %.3f ms\n", 1e3 * (wtime() - t_entry));
	}
	int stat_opt_ok = 0, stat_regular = 0, stat_aborts = 0, stat_marked = 0;          // panels done by the optimistic pass / the regular way; super-panels
	if (tournament) {
		const bool small16 = prime < 65536;
		const bool try_first = true;
		// the 64 x 64 inversion kernel for the try: signed representatives with deferred reduction need 4 B^2 + B < 2^31, B = p/2 + p/64 + 1
		bool fast_try = small_prime && (4 * (prime / 2 + prime / 64 + 1) * (prime / 2 + prime / 64 + 1) + (prime / 2 + prime / 64 + 1) <= 0x7FFFFFFFll);
		// panels per super-panel: eight on the matrix cores (K = 512 per pass over the matrix), four with VALU updates
		const int SPW = mfma_ok ? MAXSETS : 4;
		const int64_t ldz = (int64_t) MAXSETS * NB;
		signed char *Bown_h = B8 + (size_t) MAXSETS * 2 * (size_t) m * 64, *Bown_l = Bown_h + (size_t) (2 * MAXSETS * NB + 64) * 64;
		bool far_pending = false;
		const bool optimistic_enabled = true;
		bool optimistic_ok = optimistic_enabled;
		int optimistic_skip = 0;
		// a try has failed in this call: from the next super-panel on the zero rows are retired and the candidates of the tries
		// are spread over the live ones (rref_mark_dead, pick_candidates)
		bool deficient = false;
		const bool retire_rows = true;
		uint32_t *set_P[MAXSETS] = {};
		int *set_rho[MAXSETS] = {}, *set_knew[MAXSETS] = {};
		signed char *set_Mh[MAXSETS] = {}, *set_Ml[MAXSETS] = {};
		for (int sp0 = 0, spi = 0; sp0 < m; sp0 += SPW * NB, spi++) {
			if (spi == 1 || spi == 2 || (spi > 0 && spi % 4 == 0)) {
				// done when every row holds a pivot, or when the rows that do not are zero from here on (a block of
				// low rank: most of its panels would find nothing).  First on the columns of this super-panel, which the far
				// update of the previous one -- still running on the second stream -- does not write: a free row that is not
				// zero there settles the question, and the panels start beside that update (the trace of round 6 showed the
				// chain idle for the whole of it, 0.55 ms, at each of these checks); only when they are all zero there is the
				// update waited for and the rest looked at.
				int rk_nz[2] = {0, 0};
				const int own_end = std::min(m, sp0 + SPW * NB);
				const bool split = far_pending && own_end < m;          // (no update in flight: one look at everything)
				for (int pass = 0; pass < 2; pass++) {
					const bool rest = pass == 1;
					if (rest)
						HIP_CHECK(hipStreamWaitEvent(stream, ev_far, 0));
					HIP_CHECK(hipMemsetAsync(free_count + 8, 0, sizeof(int), stream));
					hipLaunchKernelGGL(rref_free_nonzero, dim3(512), dim3(256), 0, stream, dA, ld, n, (split && !rest) ? own_end : m, rest ? own_end : sp0, flags, free_count + 8);
					HIP_CHECK(hipMemcpyAsync(&rk_nz[0], rank_d, sizeof(int), hipMemcpyDeviceToHost, stream));
					HIP_CHECK(hipMemcpyAsync(&rk_nz[1], free_count + 8, sizeof(int), hipMemcpyDeviceToHost, stream));
					HIP_CHECK(hipStreamSynchronize(stream));
					if (rk_nz[0] >= n || rk_nz[1] != 0 || !split)
						break;
				}
				if (rk_nz[0] >= n || rk_nz[1] == 0)
					break;
			}
			bool have_live = false;
			int live_rows = -1;
			auto retire = [&]() {
				// (reads the columns from sp0 on: the far update of the previous super-panel must have landed -- the overlap of that
				//  update with the panel steps is given up here, on blocks whose panel steps are the slow part anyway)
				if (far_pending)
					HIP_CHECK(hipStreamWaitEvent(stream, ev_far, 0));
				HIP_CHECK(hipMemsetAsync(free_count + 10, 0, sizeof(int), stream));
				hipLaunchKernelGGL(rref_mark_dead, dim3(512), dim3(256), 0, stream, dA, ld, n, m, sp0, flags, live_list, free_count + 10);
				HIP_CHECK(hipMemcpyAsync(&live_rows, free_count + 10, sizeof(int), hipMemcpyDeviceToHost, stream));
				HIP_CHECK(hipStreamSynchronize(stream));
				have_live = true;
			};
			if (deficient && retire_rows) {
				retire();
				if (live_rows == 0)
					break;                           // every row holds a pivot or is zero from here on
			}
			const int sp_end = std::min(m, sp0 + SPW * NB);
			const int mrT = m - sp_end;              // columns beyond the super-panel
			UpdSets S{};
			int *abort_d = full_flag + 12;          // optimistic super-panels: raised by a try that cannot finish its panel
			hipEvent_t wait_before_update = nullptr;          // (phase 2: the update kernel waits for this event -- the lookahead has read A)
			bool abort_by_panel = false;                      // (passes with the tries one panel ahead: see BlockGjArgs::abort_words)
			// hand-off words of such a pass (handoff_wait / handoff_signal): by panel, the try is through / workgroups of the lookahead / of the update
			int *hand_try = use_ahead ? reinterpret_cast<int *>(alt_tile + (size_t) 2 * NB * NB) : nullptr, *hand_look = hand_try + MAXSETS, *hand_upd = hand_try + 2 * MAXSETS;
			int upd_workgroups[MAXSETS] = {};
			bool look_follows = false;                        // (phase 2 of a panel whose lookahead runs: the update waits for it)
			int *abort_pp = full_flag + 16;
			auto run_panel = [&](int c0, int nsets, bool optimistic, int phase = 0, bool with_alt = false) {
				hipStream_t stream_keep = stream;
				hipStream_t stream = (phase == 1) ? ahead.s_try : stream_keep;          // (the launches below name `stream`)
				const int *abort_c = optimistic ? (abort_by_panel ? abort_pp + nsets : abort_d) : nullptr;
				const int par = nsets & 1;          // (Ginv, gamma, cand_pivot: two copies -- the try of panel i + 1 may run beside the multipliers of panel i)
				const int width = std::min(NB, m - c0);
				// (sets alternate between two halves by super-panel: the far update of the previous super-panel may
				//  still be reading its multipliers on the second stream)
				const int slot = (spi & 1) * MAXSETS + nsets;
				uint32_t *P_s = P4 + (size_t) slot * (size_t) n * PW;
				int *rho_s = rho4 + slot * NB, *knew_s = knew4 + slot * 16;
				// (regular panel of a block on which the tries fail: straight to the tournament, three launches fewer)
				const bool straight = !optimistic && have_live;
				if (!optimistic && !straight)
					hipLaunchKernelGGL(rref_first_free, dim3(1), dim3(64), 0, stream, flags, n, free_count + 4, first64, free_count + 2, have_live ? live_list : nullptr,
					                   free_count + 10);
				// Gauss-Jordan straight on the first 64 free rows (try mode): when they give a pivot in every column of the
				// panel (the usual case while the block is not exhausted) everything up to the regular call returns at once
				BlockGjArgs bg;
				bg.A = dA;
				bg.ld = ld;
				bg.n = n;
				bg.c0 = c0;
				bg.width = width;
				bg.cand = nullptr;
				bg.cand_first = cand_first;
				bg.full = full_flag;
				bg.Ginv = Ginv + (size_t) par * NB * NB;
				bg.gamma = gamma + par * NB;
				bg.is_pivot_row = flags;
				bg.pivrow = pivrow;
				bg.pivcol = d_pivcol;
				bg.rank = rank_d;
				bg.knew = knew_s;
				bg.rho = rho_s;
				bg.cand_pivot = cand_pivot + par * NB;
				bg.F = F;
				bg.invtab = invtab;
				bg.mode = 1;
				bg.try_rows = first64;
				bg.try_count = free_count + 2;
				bg.gj_done = full_flag + 4;
				bg.try_state = full_flag + 8;
				bg.panel_index = c0 / NB;
				bg.abort = optimistic ? const_cast<int *>(abort_c) : nullptr;
				bg.abort_value = nsets;
				bg.abort_words = (optimistic && abort_by_panel) ? abort_pp : nullptr;
				bg.abort_summary = abort_d;
				bg.abort_count = MAXSETS;
				bg.ff_flags = optimistic ? flags : nullptr;
				bg.ff_hint = free_count + 4;
				bg.ff_out = first64;
				bg.ff_count = free_count + 2;
				bg.live_list = have_live ? live_list : nullptr;
				bg.live_count = free_count + 10;
				bg.alt = nullptr;
				bg.done_word = (phase == 1) ? hand_try + nsets : nullptr;
				if (with_alt) {
					bg.alt = alt_tile + (size_t) (nsets & 1) * NB * NB;
					bg.ff_flags = nullptr;          // (rref_lookahead picked the candidates: first64 / free_count + 2)
				}
				// (a block on which tries fail -- dependent columns, or rows that depend on each other --: the try that takes what its
				//  candidates give, with the proof in the multiplier kernel; see BlockGjArgs)
				const bool take_what_comes = optimistic && have_live && small_prime && small16 && mfma_ok && width == NB;
				if (phase == 2) {
					;                                // (the try ran on its own stream)
				} else if (take_what_comes) {
					bg.mode = 2;
					hipLaunchKernelGGL((rref_block_gj<true, 8>), dim3(1), dim3(1024), invtab_bytes, stream, bg);
					bg.mode = 1;
				} else if (try_first && !straight) {
					if (fast_try)
						hipLaunchKernelGGL(rref_try_inverse, dim3(1), dim3(256), invtab_bytes, stream, bg);
					else if (small_prime)
						hipLaunchKernelGGL((rref_block_gj<true, 8>), dim3(1), dim3(1024), invtab_bytes, stream, bg);
					else
						hipLaunchKernelGGL((rref_block_gj<false, 8>), dim3(1), dim3(1024), 0, stream, bg);
				}
				if (phase == 1)
					return;
				if (!optimistic) {
				// the first 64 free rows alone, by selection (when the try was skipped or failed)
				if (straight)
					;
				else if (small_prime)
					hipLaunchKernelGGL(rref_select_first<true>, dim3(1), dim3(256), 0, stream, dA, ld, c0, width, first64, free_count + 2,
					                   cand_first, F, full_flag, full_flag + 4);
				else
					hipLaunchKernelGGL(rref_select_first<false>, dim3(1), dim3(256), 0, stream, dA, ld, c0, width, first64, free_count + 2,
					                   cand_first, F, full_flag, full_flag + 4);
				// (everything from here to the Gauss-Jordan block returns at once when that was enough)
				hipLaunchKernelGGL(rref_free_list, dim3(1), dim3(1024), 0, stream, flags, n, candA, free_count, full_flag, straight ? full_flag : nullptr);
				int n_in = have_live ? std::max(live_rows, 1) : n;          // (the free rows are among the live ones: fewer levels on a block that is running out)
				const int *count_dev = free_count;
				int *src = candA, *dst = candB;
				for (;;) {
					const int wgs = (n_in + SEL_ROWS - 1) / SEL_ROWS;
					if (small_prime)
						hipLaunchKernelGGL(rref_select_kernel<true>, dim3(wgs), dim3(SEL_ROWS), 0, stream, dA, ld, c0, width, src, n_in,
						                   count_dev, dst, F, full_flag, nullptr);
					else
						hipLaunchKernelGGL(rref_select_kernel<false>, dim3(wgs), dim3(SEL_ROWS), 0, stream, dA, ld, c0, width, src, n_in,
						                   count_dev, dst, F, full_flag, nullptr);
					// (the first level reads candA and writes candB; candA is then free)
					std::swap(src, dst);
					n_in = wgs * NB;
					count_dev = nullptr;
					if (wgs == 1)
						break;
				}
				bg.cand = src;
				bg.mode = 0;
				if (small_prime)
					hipLaunchKernelGGL((rref_block_gj<true, 8>), dim3(1), dim3(1024), invtab_bytes, stream, bg);
				else
					hipLaunchKernelGGL((rref_block_gj<false, 8>), dim3(1), dim3(1024), 0, stream, bg);
				}
				signed char *Mh_s = M8 + (size_t) slot * 2 * (size_t) n * 64, *Ml_s = Mh_s + (size_t) n * 64;
				MultArgs ma{dA, ld, n, m, c0, Ginv + (size_t) par * NB * NB, gamma + par * NB, knew_s, P_s, F, rho_s, cand_pivot + par * NB, mfma_ok ? Mh_s : nullptr, mfma_ok ? Ml_s : nullptr,
				            mfma_ok ? Zacc + (size_t) nsets * NB : nullptr /* M_s becomes block `nsets` of Z */, ldz, abort_c,
				            take_what_comes ? flags : nullptr, abort_d, nsets};
				ma.wait_word = (phase == 2) ? hand_try + nsets : nullptr;
				ma.wait_word2 = (phase == 2 && look_follows) ? hand_look + nsets : nullptr;
				ma.wait_value2 = LOOK_WGS;
				const int nmult = (n + 63) / 64;
				// the columns of the super-panel, from this panel on, and (matrix cores) the multipliers of its earlier
				// panels, blocks 0 .. nsets - 1 of Z: K = 64 update now
				const int mr_sp = sp_end - c0;
				UpdSets one{};
				one.P[0] = P_s;
				one.B[0] = B;
				one.knew[0] = knew_s;
				one.nsets = 1;
				one.abort = abort_c;
				int tiles = (mr_sp + 63) / 64;
				if (mfma_ok) {
					// multipliers and the digit planes of the pivot rows in one launch
					const int tiles1 = (mr_sp + 63) / 64, mr2 = nsets * NB;
					GatherArgs ga{dA, ld, c0, mr_sp, rho_s, knew_s, Bown_h, Bown_l, nullptr, F, Zacc, ldz, tiles1, mr2, abort_c};
					const int ngather = (tiles1 * 64 + mr2 + 255) / 256;
					if (small16)
						hipLaunchKernelGGL(rref_mult_gather<true>, dim3(nmult + ngather), dim3(256), 0, stream, ma, ga, nmult);
					else
						hipLaunchKernelGGL(rref_mult_gather<false>, dim3(nmult + ngather), dim3(256), 0, stream, ma, ga, nmult);
					one.Mh[0] = Mh_s;
					one.Ml[0] = Ml_s;
					one.Bh[0] = Bown_h;
					one.Bl[0] = Bown_l;
					if (mr2 > 0) {
						one.Z2 = Zacc;
						one.ld2 = ldz;
						one.tiles1 = tiles1;
						one.mr2 = mr2;
						tiles = tiles1 + mr2 / 64;
					}
				} else {
					if (small16)
						hipLaunchKernelGGL(rref_multipliers<true>, dim3(nmult), dim3(256), 0, stream, ma);
					else
						hipLaunchKernelGGL(rref_multipliers<false>, dim3(nmult), dim3(256), 0, stream, ma);
					hipLaunchKernelGGL(rref_gather_pivot_rows, dim3(512), dim3(256), 0, stream, dA, ld, c0, mr_sp, rho_s, knew_s, B);
				}
				if (wait_before_update != nullptr)
					HIP_CHECK(hipStreamWaitEvent(stream, wait_before_update, 0));
				if (phase == 2) {
					one.done_word = hand_upd + nsets;
					upd_workgroups[nsets] = tiles * ((n + 63) / 64);
				}
				timed([&]() {
					dim3 grid(tiles, (n + 63) / 64);
					if (mfma_ok)
						hipLaunchKernelGGL(rref_update_mfma_multi, grid, dim3(256), 0, stream, dA, ld, n, c0, mr_sp, one, F);
					else
						hipLaunchKernelGGL(rref_update_valu, grid, dim3(256), 0, stream, dA, ld, n, c0, mr_sp, P_s, B, knew_s, F);
				});
				set_P[nsets] = P_s;
				set_rho[nsets] = rho_s;
				set_knew[nsets] = knew_s;
				set_Mh[nsets] = Mh_s;
				set_Ml[nsets] = Ml_s;
				HIP_CHECK(hipGetLastError());
			};
			// Optimistic pass: first-free rows, try, multipliers, update -- five launches per panel instead of twelve, none of
			// the selection kernels (they return at once when the try succeeds, but a launch is a launch: 30 us per panel).
			// The host looks at the flag once per super-panel; the panels from the one that raised it on are redone the
			// regular way, and the next super-panel is not attempted optimistically.
			const int npanels = (sp_end - sp0 + NB - 1) / NB;
			int first_regular = 0;
			if (optimistic_ok && fast_try && mfma_ok && try_first && ms_update == nullptr) {
				// passes of optimistic panel steps from `start` on; a pass ends at the panel whose try (or proof) failed.  The first
				// failure of a call switches to the tries that take what comes (rows retired, candidates spread: BlockGjArgs mode 2)
				// and the pass is taken up again at that panel; a failure in that mode sends the one panel the regular way.
				int start = 0, fallbacks = 0;
				while (start < npanels) {
					HIP_CHECK(hipMemsetAsync(abort_d, 0, sizeof(int), stream));
					// (knew[1] of a slot = the pivots a mode-2 try has taken, for rref_rollback: nothing yet)
					HIP_CHECK(hipMemset2DAsync(knew4 + (size_t) (spi & 1) * MAXSETS * 16 + 1, 16 * sizeof(int), 0, sizeof(int), MAXSETS, stream));
					// one panel ahead: plain optimistic passes only (full panels, no list of live rows, p < 2^16)
					// (... and blocks of at most 6,144 rows: beyond, the multipliers and the update
					//  of a panel take longer than its try -- the chain is no longer what the call waits for, and two streams of kernels that
					//  poll each other only get in the way: 16,384 x 16,384 went from 40 to 53 ms with it)
					const int look_rows = 6144;
					const bool look = use_ahead && !have_live && small16 && npanels - start >= 2 && n <= look_rows;
					if (!look) {
						for (int i = start; i < npanels; i++)
							run_panel(sp0 + i * NB, i, true);
					} else {
						abort_by_panel = true;
						HIP_CHECK(hipMemsetAsync(abort_pp, 0, MAXSETS * sizeof(int), stream));
						HIP_CHECK(hipMemsetAsync(hand_try, 0, 3 * MAXSETS * sizeof(int), stream));
						HIP_CHECK(hipEventRecord(ahead.ev_start, stream));
						HIP_CHECK(hipStreamWaitEvent(ahead.s_try, ahead.ev_start, 0));
						// The try stream: try(start), look(start), try(start + 1), ... back to back, no event between them.  The main
						// stream: multipliers(i) (its workgroups wait for try(i)'s word), update(i) (waits for look(i): it REWRITES the
						// next panel's columns of the candidates and of the pivot rows, which look(i) reads), ...  look(i) waits for
						// the workgroups of update(i - 1).  No cycle: every wait is for a kernel that is earlier in BOTH orders.
						for (int i = start; i < npanels; i++) {
							const int c0 = sp0 + i * NB;
							run_panel(c0, i, true, 1, i > start);          // try(i): from A for the first panel of the pass, from look(i - 1)'s tile after that
							const bool next_full = i + 1 < npanels && c0 + 2 * NB <= m;
							if (next_full) {
								const int slot = (spi & 1) * MAXSETS + i;
								LookArgs la{dA, ld, n, c0, Ginv + (size_t) (i & 1) * NB * NB, rho4 + slot * NB, knew4 + slot * 16, flags, free_count + 4, first64, free_count + 2,
								            alt_tile + (size_t) ((i + 1) & 1) * NB * NB, abort_pp + i, (i > start) ? hand_upd + (i - 1) : nullptr, (i > start) ? upd_workgroups[i - 1] : 0,
								            hand_look + i, F};
								hipLaunchKernelGGL(rref_lookahead, dim3(LOOK_WGS), dim3(256), 0, ahead.s_try, la);
							}
							look_follows = next_full;
							run_panel(c0, i, true, 2);
							look_follows = false;
							if (!next_full && i + 1 < npanels) {
								// (a narrow last panel: its try reads A itself, after update(i))
								HIP_CHECK(hipEventRecord(ahead.ev_upd[i], stream));
								HIP_CHECK(hipStreamWaitEvent(ahead.s_try, ahead.ev_upd[i], 0));
							}
						}
						abort_by_panel = false;
					}
					int raised = 0;
					HIP_CHECK(hipMemcpyAsync(&raised, abort_d, sizeof(int), hipMemcpyDeviceToHost, stream));
					HIP_CHECK(hipStreamSynchronize(stream));
					const int stop = raised != 0 ? raised - 1 : npanels;
					stat_opt_ok += stop - start;
					start = stop;
					if (raised == 0)
						break;
					stat_aborts += 1;
					if (have_live) {
						// (a mode-2 try whose proof failed has taken its pivots already: they go back before the panel is redone)
						const int slot = (spi & 1) * MAXSETS + stop;
						if (sh::env_get("SPASM_HIP_RREF_TIMING")) {
							int took = 0, tc = 0;
							HIP_CHECK(hipMemcpy(&took, knew4 + slot * 16 + 1, sizeof(int), hipMemcpyDeviceToHost));
							HIP_CHECK(hipMemcpy(&tc, free_count + 2, sizeof(int), hipMemcpyDeviceToHost));
							fprintf(stderr, "[rref timing] super-panel %d: the try of panel %d took %d pivots from %d candidates (%d live rows) and the proof failed\n", spi, stop, took, tc,
							        live_rows);
						}
						hipLaunchKernelGGL(rref_rollback, dim3(1), dim3(64), 0, stream, rho4 + slot * NB, knew4 + slot * 16, flags, rank_d);
						run_panel(sp0 + stop * NB, stop, false);
						stat_regular += 1;
						start = stop + 1;
						fallbacks += 1;
						if (fallbacks >= 3)
							break;                   // (the rest of the super-panel the regular way)
					} else if (retire_rows && small_prime && small16) {
						deficient = true;
						retire();
					} else {
						break;
					}
				}
				first_regular = start;
				if (start < npanels)
					optimistic_skip = (deficient && retire_rows) ? 0 : 2;          // (2: the next super-panel is not attempted)
			}
			if (optimistic_skip > 0)
				optimistic_skip -= 1;
			optimistic_ok = optimistic_enabled && optimistic_skip == 0;
			for (int i = first_regular; i < npanels; i++)
				run_panel(sp0 + i * NB, i, false);
			stat_regular += npanels - first_regular;
			stat_marked += have_live;
			const int nsets = npanels;
			if (mrT > 0) {
				// beyond the super-panel.  What is read here -- the rows rho_i beyond the super-panel -- is written by the far
				// update of the previous super-panel on the second stream: wait for it here, not earlier -- the panel steps
				// above (latency-bound, one workgroup most of the time) ran beside it.
				if (far_pending)
					HIP_CHECK(hipStreamWaitEvent(stream, ev_far, 0));
				if (mfma_ok) {
					// C += sum_i M'_i C[rho_i]: M'_i = block i of Z (digit planes), C[rho_i] = the rows as they stand
					signed char *MZ = M8 + (size_t) (2 * MAXSETS + (spi & 1) * MAXSETS) * 2 * (size_t) n * 64;
					hipLaunchKernelGGL(rref_split_Z, dim3((n + 63) / 64, nsets), dim3(256), 0, stream, Zacc, ldz, n, nsets, MZ, (int64_t) 2 * n * 64, F);
					GatherSets gs{};
					for (int s = 0; s < nsets; s++) {
						signed char *Bh_s = B8 + (size_t) s * 2 * (size_t) m * 64, *Bl_s = Bh_s + (size_t) m * 64;
						gs.rho[s] = set_rho[s];
						gs.knew[s] = set_knew[s];
						gs.Bh[s] = Bh_s;
						gs.Bl[s] = Bl_s;
						S.knew[s] = set_knew[s];
						S.Mh[s] = MZ + (size_t) s * 2 * (size_t) n * 64;
						S.Ml[s] = S.Mh[s] + (size_t) n * 64;
						S.Bh[s] = Bh_s;
						S.Bl[s] = Bl_s;
					}
					hipLaunchKernelGGL(rref_gather_split_sets, dim3((mrT + 255) / 256, nsets), dim3(256), 0, stream, dA, ld, sp_end, mrT, gs, F);
				} else {
					for (int s = 0; s < nsets; s++) {
						S.nsets = s;          // the sets before this one
						uint32_t *Bt_s = Bt4 + (size_t) s * (size_t) NB * (size_t) m;
						dim3 grid((mrT + 255) / 256, NB);
						if (small16)
							hipLaunchKernelGGL(rref_trailing_B<true>, grid, dim3(256), 0, stream, dA, ld, n, sp_end, mrT, S, set_rho[s], set_knew[s], Bt_s, F);
						else
							hipLaunchKernelGGL(rref_trailing_B<false>, grid, dim3(256), 0, stream, dA, ld, n, sp_end, mrT, S, set_rho[s], set_knew[s], Bt_s, F);
						S.P[s] = set_P[s];
						S.B[s] = Bt_s;
						S.knew[s] = set_knew[s];
					}
				}
				S.nsets = nsets;
				// near part (the next super-panel's own columns) on this stream, far part on the second one
				const int near = (mfma_ok && stream2 != nullptr) ? std::min(mrT, SPW * NB) : mrT;
				// (a 64 x 128 tile kernel, two waves per SIMD, 256 registers, was built in round 5 and took 24 % less serialised time for the
				//  updates -- and 5 % MORE for the call, with and without the lookahead of round 6: removed.  Round 6 also built a 128 x 128
				//  tile kernel for the far part -- 64 x 64 per wave, three accumulators per 32 x 32 tile, half the LDS bytes per matrix
				//  instruction, operands read one phase ahead, two LDS buffers, ONE workgroup per CU (320 registers): bit-identical and no
				//  faster, 5.08 against 5.00 ms serialised.  With K = 512 a workgroup is prologue (its 64 KB of C, the first planes),
				//  eight short sets whose planes arrive later than the 0.43 us the set before takes to multiply, and an epilogue; with one
				//  workgroup per CU these phases have nothing to hide behind.  What the far update needs is several workgroups per CU in
				//  different phases, which is what the 64 x 64 kernel has: NOTES/round6.md.)
				timed([&]() {
					dim3 grid((near + 63) / 64, (n + 63) / 64);
					if (mfma_ok) {
						hipLaunchKernelGGL(rref_update_mfma_multi, grid, dim3(256), 0, stream, dA, ld, n, sp_end, near, S, F);
					} else {
						for (int s = 0; s < nsets; s++)
							hipLaunchKernelGGL(rref_update_valu, grid, dim3(256), 0, stream, dA, ld, n, sp_end, mrT, S.P[s], S.B[s], S.knew[s], F);
					}
				});
				far_pending = false;
				if (near < mrT) {
					UpdSets Sf = S;
					for (int s = 0; s < nsets; s++) {
						Sf.Bh[s] = S.Bh[s] + (size_t) near * 64;
						Sf.Bl[s] = S.Bl[s] + (size_t) near * 64;
					}
					HIP_CHECK(hipEventRecord(ev_near, stream));
					HIP_CHECK(hipStreamWaitEvent(stream2, ev_near, 0));
					dim3 grid((mrT - near + 63) / 64, (n + 63) / 64);
					if (ms_update != nullptr)
						HIP_CHECK(hipEventRecord(e0, stream2));
					hipLaunchKernelGGL(rref_update_mfma_multi, grid, dim3(256), 0, stream2, dA, ld, n, sp_end + near, mrT - near, Sf, F);
					HIP_CHECK(hipEventRecord(ev_far, stream2));
					far_pending = true;
					if (ms_update != nullptr) {          // (timing runs serialise the two streams)
						HIP_CHECK(hipEventSynchronize(ev_far));
						float ms;
						HIP_CHECK(hipEventElapsedTime(&ms, e0, ev_far));
						total_update += ms;
					}
				}
				HIP_CHECK(hipGetLastError());
			}
		}
		if (far_pending)
			HIP_CHECK(hipStreamWaitEvent(stream, ev_far, 0));
	} else {
		for (int c0 = 0; c0 < m; c0 += NB) {
			if (c0 > 0 && (c0 / NB) % 8 == 0) {           // every row already holds a pivot: the rest is reduced
				int rk = 0;
				HIP_CHECK(hipMemcpyAsync(&rk, rank_d, sizeof(int), hipMemcpyDeviceToHost, stream));
				HIP_CHECK(hipStreamSynchronize(stream));
				if (rk >= n)
					break;
			}
			const int width = (m - c0 < NB) ? m - c0 : NB;
			PanelArgs g;
			g.A = dA;
			g.ld = ld;
			g.n = n;
			g.m = m;
			g.c0 = c0;
			g.width = width;
			g.P = P;
			g.is_pivot_row = flags;
			g.pivrow = pivrow;
			g.pivcol = d_pivcol;
			g.rank = rank_d;
			g.knew = knew;
			g.rho = rho;
			g.F = F;
			if (n >= coop_min_rows) {
				CoopPanelArgs ca;
				ca.g = g;
				ca.barrier = coop_barrier;
				ca.cand = coop_cand;
				ca.err = coop_err;
				HIP_CHECK(hipMemsetAsync(coop_barrier, 0, sizeof(unsigned int), stream));
				HIP_CHECK(hipMemsetAsync(coop_cand, 0x7F, NB * sizeof(int), stream));     // 0x7F7F7F7F >= any row index
				int G = (n + COOP_THREADS - 1) / COOP_THREADS;
				if (G > 64)
					G = 64;
				hipLaunchKernelGGL(rref_panel_coop_kernel, dim3(G), dim3(COOP_THREADS), 0, stream, ca);
			} else {
				hipLaunchKernelGGL(rref_panel_kernel, dim3(1), dim3(PANEL_THREADS), 0, stream, g);
			}
			const int c1 = c0 + width;
			const int mr = m - c1;
			if (mr > 0) {
				hipLaunchKernelGGL(rref_gather_pivot_rows, dim3(512), dim3(256), 0, stream, dA, ld, c1, mr, rho, knew, B);
				dim3 grid((mr + 63) / 64, (n + 63) / 64);
				if (ms_update != nullptr)
					HIP_CHECK(hipEventRecord(e0, stream));
				if (mfma_ok)
					hipLaunchKernelGGL(rref_update_mfma, grid, dim3(256), 0, stream, dA, ld, n, c1, mr, P, B, knew, F);
				else
					hipLaunchKernelGGL(rref_update_valu, grid, dim3(256), 0, stream, dA, ld, n, c1, mr, P, B, knew, F);
				if (ms_update != nullptr) {
					HIP_CHECK(hipEventRecord(e1, stream));
					HIP_CHECK(hipEventSynchronize(e1));
					float ms;
					HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
					total_update += ms;
				}
			}
			HIP_CHECK(hipGetLastError());
		}
	}
	double t_loop = 0.0;
	if (sh::env_get("SPASM_HIP_RREF_TIMING")) {
		HIP_CHECK(hipStreamSynchronize(stream));
		if (stream2 != nullptr)
			HIP_CHECK(hipStreamSynchronize(stream2));
		t_loop = wtime();
		fprintf(stderr, "[rref timing] up to the end of the panels: %.3f ms (%d panels by the optimistic passes, %d the regular way, %d super-panels gave up, %d with retired rows)\n",
		        1e3 * (t_loop - t_entry), stat_opt_ok, stat_regular, stat_aborts, stat_marked);
	}
	int rank = 0, coop_failed = 0;
	HIP_CHECK(hipMemcpyAsync(&rank, rank_d, sizeof(int), hipMemcpyDeviceToHost, stream));
	if (stream2 != nullptr) {
		HIP_CHECK(hipStreamSynchronize(stream2));          // (the stream itself is kept for the next call)
		(void) hipEventDestroy(ev_near);
		(void) hipEventDestroy(ev_far);
	}
	HIP_CHECK(hipMemcpyAsync(&coop_failed, coop_err, sizeof(int), hipMemcpyDeviceToHost, stream));
	int stuck = 0;
	if (use_ahead)
		HIP_CHECK(hipMemcpyFromSymbolAsync(&stuck, HIP_SYMBOL(g_handoff_stuck), sizeof(int), 0, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	if (stuck) {
		const int zero = 0;
		HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_handoff_stuck), &zero, sizeof(int)));
	}
	if (tournament) {
		ws_free(candA);
		ws_free(candB);
		ws_free(free_count);
		ws_free(gamma);
		ws_free(cand_first);
		ws_free(first64);
		ws_free(live_list);
		ws_free(cand_pivot);
		ws_free(P4);
		ws_free(Bt4);
		ws_free(Zacc);
		ws_free(rho4);
		ws_free(M8);
		ws_free(invtab);
		ws_free(B8);
		ws_free(knew4);
		ws_free(full_flag);
		ws_free(Ginv);
	}
	ws_free(coop_barrier);
	ws_free(coop_cand);
	ws_free(coop_err);
	if (coop_failed)
		die("dense RREF: a grid-wide barrier timed out (cooperative panel kernel)");
	if (stuck)
		die("dense RREF: a hand-off between the two streams of a lookahead pass timed out (SPASM_HIP_EXPERIMENT=1 SPASM_HIP_RREF_LOOKAHEAD=0 runs without them)");
	if (rank > 0) {
		uint32_t *tmp = nullptr;
		ws_malloc((void **) &tmp, (size_t) rank * (size_t) m * sizeof(uint32_t));
		hipLaunchKernelGGL(rref_rows_to_tmp, dim3(std::min(rank, 4096)), dim3(256), 0, stream, dA, ld, m, pivrow, rank, tmp);
		hipLaunchKernelGGL(rref_tmp_to_rows, dim3(std::min(n, 4096)), dim3(256), 0, stream, dA, ld, n, m, rank, tmp, pivrow);
		HIP_CHECK(hipStreamSynchronize(stream));
		ws_free(tmp);
	}
	if (ms_update != nullptr) {
		*ms_update = total_update;
		(void) hipEventDestroy(e0);
		(void) hipEventDestroy(e1);
	}
	ws_free(P);
	ws_free(B);
	ws_free(flags);
	ws_free(pivrow);
	ws_free(rank_d);
	ws_free(knew);
	ws_free(rho);
	if (sh::env_get("SPASM_HIP_RREF_TIMING"))
		fprintf(stderr, "[rref timing] whole call: %.3f ms\n", 1e3 * (wtime() - t_entry));
	return rank;
}

}  // namespace sh

// --------------------------------------------------------------------------
// Echelon rows by ROW panels: extends k reduced echelon rows E (rows [0, k) of M, pivot columns piv[0..k), identity on
// them) by the Sn rows below them.  This is what the dense / low-rank finish needs on a WIDE remainder (ch8-8.b5: stacks of
// 8,000 rows x 104,000 columns of rank ~4,000): device_rref above walks the 1,635 column panels of such a stack, finds two
// or three pivots in each and pays a pass over the trailing matrix per super-panel of 512 COLUMNS (0.35 - 0.5 s per
// stack, 2.5 pivots per panel).  Here a panel is 64 ROWS; its pivots are taken wherever its rows have their leftmost
// entries (any pivot will do for an echelon basis -- the column rank profile is not asked for), so a panel yields up to 64
// pivots and the matrix is passed over once per 64 PIVOTS:
//   A. Y -= Y[:, piv(E)] E                                     (matrix cores, K = k)
//   B. for every panel of 64 rows of Y, until none of its rows is left without a pivot or non-zero:
//        leftmost non-zero column of every row still waiting -> the first 256 pivot-free columns from the smallest one on,
//        plus those 64 columns themselves; Gauss-Jordan of the 64 x 320 block in LDS (one workgroup) -> T (64 x 64) and
//        the new pivot columns J;
//        every row of the matrix, in ONE launch on the matrix cores:  C_i += sum_u M'[i][u] P_u  with the OLD panel rows P_u,
//          M' = T - I on the panel's own rows, M'[i] = -C_i[J] T_J elsewhere (E and the earlier panels included: they stay reduced)
//   C. the rows that hold a pivot move up behind E, in panel order.
// p <= 65279 (two signed base-256 digits per operand, as rref_update_mfma_multi).  Returns the new number of echelon rows.
// --------------------------------------------------------------------------
namespace sh {

namespace {
constexpr int RP_ROWS = 64;
constexpr int RP_WIN = 256;

// leftmost non-zero column of every waiting row of the panel: one wave per (row, chunk of 1024 columns)
__global__ __launch_bounds__(64) void rowpanel_leftmost(const uint32_t *P, int64_t ld, int m, const int *state, int *left)
{
	const int row = blockIdx.y, lane = threadIdx.x;
	if (state[row] != -1)
		return;                              // has a pivot (>= 0) or is known to be zero (-2)
	const int c0 = blockIdx.x * 1024;
	if (c0 >= atomicMin(&left[row], 0x7FFFFFFF))
		return;                              // (a chunk further left already holds a non-zero entry)
	int best = 0x7FFFFFFF;
	for (int q = 0; q < 16; q++) {
		const int c = c0 + q * 64 + lane;
		const uint32_t v = (c < m) ? P[(int64_t) row * ld + c] : 0u;
		const uint64_t mask = __ballot(v != 0);
		if (mask != 0) {
			best = c0 + q * 64 + __builtin_ctzll(mask);
			break;
		}
	}
	if (lane == 0 && best != 0x7FFFFFFF)
		atomicMin(&left[row], best);
}

// Gauss-Jordan of the panel restricted to RP_COLS = 320 of its columns, in row order, with the row transformation
// accumulated in T (64 x 64, starts as the identity).  The columns: the first 256 that hold no pivot yet from the smallest
// leftmost entry of the waiting rows on -- dense rows (random combinations reduced by the echelon rows) have their first
// entries there and get up to 64 pivots out of it; the pivot columns of the echelon rows and of the earlier panels, zero
// in every waiting row, are skipped --, and the leftmost entry of EVERY waiting row (left[t]) -- sparse rows (the rows of a
// sparse Schur complement themselves) have theirs far apart: the window alone caught one row in four on a ch8-8.b5 block
// (288 steps for its 64 panels, each a pass over the whole stack; 155 with their own columns added); the leftmost entries
// alone collapse to one or two distinct columns on dense rows (3,082 steps), five columns per row from its leftmost entry
// on are no better than the window (272).  A row takes the first of these columns where it is still non-zero once the rows
// before it have been eliminated; a row that is zero on all of them waits for the next step.
// state[t] becomes the pivot column of row t when it finds one here; newpiv[t] = that column for the rows that found a
// pivot in THIS call, -1 for the others.
constexpr int RP_COLS = RP_WIN + RP_ROWS;

// (1,024 threads: a column of the block per thread of a group of 256, the 64 rows dealt out to the four groups -- the
//  elimination of a pivot column from 63 rows is what a step costs, 64 pivots a call on dense rows: 1.08 ms with 256 threads)
constexpr int RP_THREADS = 1024, RP_GROUPS = RP_THREADS / 256;

__global__ __launch_bounds__(RP_THREADS) void rowpanel_window(const uint32_t *P, int64_t ld, int m, int rows, const int *left, int *state, int *newpiv,
                                                              uint32_t *T_out, unsigned char *is_piv, MontDev F, const unsigned short *invtab)
{
	// p < 2^16 here (device_echelon_extend): products of residues fit 32 bits -- one full-rate 24-bit multiply and a Barrett
	// reduction (two quarter-rate multiplies) where the Montgomery mulmod took eight, the elimination a x - f y as ONE reduction of
	// x + (p - f) y < p + p^2, and the inverse of a pivot out of the table of rref_inverse_table instead of a Fermat power of 2 x 16
	// dependent mulmods computed by every thread (round 6: the kernel took 742 us per full panel, 11.6 us per pivot, and was
	// 190 ms of the 450 ms of the no-greedy flow on mk13.b5)
	const uint32_t bm = (uint32_t) (0x100000000ull / F.p);
	const bool q24 = F.p >= 256;          // v / p < 2^24 for every v < 2^32
	auto red = [&](uint32_t v) -> uint32_t {
		const uint32_t q = __umulhi(v, bm);
		uint32_t rem = v - (q24 ? __umul24(q, F.p) : q * F.p);
		rem = (rem >= F.p) ? rem - F.p : rem;
		rem = (rem >= F.p) ? rem - F.p : rem;
		return rem;
	};
	extern __shared__ uint32_t rp_lds[];
	uint32_t(*W)[RP_COLS + 1] = reinterpret_cast<uint32_t(*)[RP_COLS + 1]>(rp_lds);
	uint32_t(*T)[RP_ROWS + 1] = reinterpret_cast<uint32_t(*)[RP_ROWS + 1]>(rp_lds + RP_ROWS * (RP_COLS + 1));
	__shared__ int cols[RP_COLS], s_state[RP_ROWS], s_pick[5], s_w0, s_count, wcnt[4];
	__shared__ uint32_t fac[RP_ROWS];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int col = tid & 255, grp = tid >> 8;          // (a group is four whole waves: everything below that depends on grp is wave-uniform)
	if (tid < RP_ROWS) {
		const int st = state[tid];
		s_state[tid] = st;
		const int c = (st == -1 && tid < rows) ? left[tid] : 0x7FFFFFFF;
		cols[RP_WIN + tid] = (c < m) ? c : -1;
		newpiv[tid] = -1;
		int w0 = (c < m) ? c : 0x7FFFFFFF;          // the smallest leftmost entry (wave 0 holds the 64 rows)
		for (int d = 32; d >= 1; d >>= 1)
			w0 = min(w0, __shfl_xor(w0, d));
		if (tid == 0) {
			s_w0 = w0;
			s_count = 0;
		}
	}
	if (grp == 0)
		cols[col] = -1;
	__syncthreads();
	for (int base = s_w0, chunk = 0; base < m && chunk < 128; base += 256, chunk++) {
		const int have = s_count;
		if (have >= RP_WIN)
			break;                           // (uniform: shared, read after a barrier)
		const int c = base + col;
		const bool free_col = grp == 0 && c < m && is_piv[c] == 0;
		const uint64_t mask = __ballot(free_col);
		if (grp == 0 && lane == 0)
			wcnt[wave] = __popcll(mask);
		__syncthreads();
		int before = have, all = 0;
		for (int w2 = 0; w2 < 4; w2++) {
			if (w2 < wave)
				before += wcnt[w2];
			all += wcnt[w2];
		}
		const int pos = before + __popcll(mask & ((1ull << lane) - 1ull));
		if (free_col && pos < RP_WIN)
			cols[pos] = c;
		__syncthreads();
		if (tid == 0)
			s_count = have + all;
		__syncthreads();
	}
	__syncthreads();
	for (int t = grp; t < RP_ROWS; t += RP_GROUPS) {
		W[t][col] = (cols[col] >= 0 && t < rows) ? P[(int64_t) t * ld + cols[col]] : 0u;          // (the last panel may be short: its missing rows are zero, state -2)
		if (col < RP_ROWS) {
			W[t][RP_WIN + col] = (cols[RP_WIN + col] >= 0 && t < rows) ? P[(int64_t) t * ld + cols[RP_WIN + col]] : 0u;
			T[t][col] = (t == col) ? 1u : 0u;
		}
	}
	__syncthreads();
	for (int t = 0; t < RP_ROWS; t++) {
		if (s_state[t] != -1)
			continue;                        // (uniform: shared)
		// first non-zero entry of row t among the columns: the 256 of the window (one per thread of group 0), then the 64 leftmost entries (wave 0)
		if (grp == 0) {
			const uint64_t mask = __ballot(W[t][col] != 0);
			if (lane == 0)
				s_pick[wave] = (mask != 0) ? wave * 64 + __builtin_ctzll(mask) : 0x7FFFFFFF;
			if (wave == 0) {
				const uint64_t mask2 = __ballot(W[t][RP_WIN + lane] != 0);
				if (lane == 0)
					s_pick[4] = (mask2 != 0) ? RP_WIN + __builtin_ctzll(mask2) : 0x7FFFFFFF;
			}
		}
		__syncthreads();
		const int c = min(min(min(s_pick[0], s_pick[1]), min(s_pick[2], s_pick[3])), s_pick[4]);
		if (c == 0x7FFFFFFF) {
			__syncthreads();
			continue;                        // zero on every column looked at: the row waits for the next step
		}
		const uint32_t inv = (uint32_t) invtab[W[t][c]];
		const uint32_t wt = red(__umul24(W[t][col], inv));
		const uint32_t wt2 = (col < RP_ROWS) ? red(__umul24(W[t][RP_WIN + col], inv)) : 0u, tt = (col < RP_ROWS) ? red(__umul24(T[t][col], inv)) : 0u;
		if (tid < RP_ROWS)
			fac[tid] = (tid == t) ? 0u : W[tid][c];
		__syncthreads();
		if (grp == 0) {
			W[t][col] = wt;
			if (col < RP_ROWS) {
				W[t][RP_WIN + col] = wt2;
				T[t][col] = tt;
			}
		}
		for (int s2 = grp; s2 < RP_ROWS; s2 += RP_GROUPS) {
			const uint32_t f = fac[s2];
			if (f == 0)
				continue;                    // (uniform: a group is whole waves)
			const uint32_t nf = F.p - f;
			W[s2][col] = red(W[s2][col] + __umul24(nf, wt));
			if (col < RP_ROWS) {
				W[s2][RP_WIN + col] = red(W[s2][RP_WIN + col] + __umul24(nf, wt2));
				T[s2][col] = red(T[s2][col] + __umul24(nf, tt));
			}
		}
		if (tid == 0) {
			s_state[t] = cols[c];
			newpiv[t] = cols[c];
			is_piv[cols[c]] = 1;
		}
		__syncthreads();
	}
	if (tid < RP_ROWS)
		state[tid] = s_state[tid];
	for (int e = tid; e < RP_ROWS * RP_ROWS; e += RP_THREADS)
		T_out[e] = T[e / RP_ROWS][e % RP_ROWS];
}

__global__ __launch_bounds__(256) void rowpanel_mark_pivots(const int *piv, int k, unsigned char *is_piv)
{
	const int t = blockIdx.x * 256 + threadIdx.x;
	if (t < k)
		is_piv[piv[t]] = 1;
}

// digit planes of the multipliers of one panel step, for every row i of the matrix (n rows):
//   rows of the panel (i in [r0, r0 + 64)):  M'[i] = T[i - r0] - e_{i - r0}
//   other rows:                              M'[i][u] = - sum_{t new} C[i][J_t] T[t][u]
__global__ __launch_bounds__(256) void rowpanel_multipliers(const uint32_t *C, int64_t ld, int n, int r0, const int *newpiv, const uint32_t *T,
                                                            signed char *Mh, signed char *Ml, MontDev F)
{
	__shared__ uint32_t sT[RP_ROWS][RP_ROWS + 1];
	__shared__ int sJ[RP_ROWS];
	const int tid = threadIdx.x;
	for (int e = tid; e < RP_ROWS * RP_ROWS; e += 256)
		sT[e / RP_ROWS][e % RP_ROWS] = T[e];
	if (tid < RP_ROWS)
		sJ[tid] = newpiv[tid];
	__syncthreads();
	const int i = blockIdx.x * 64 + (tid & 63), q = tid >> 6;          // row, quarter of the 64 multipliers
	if (i >= n)
		return;
	uint32_t out[16];
	if (i >= r0 && i < r0 + RP_ROWS) {
		for (int u = 0; u < 16; u++) {
			const uint32_t v = sT[i - r0][q * 16 + u];
			out[u] = (i - r0 == q * 16 + u) ? submod(v, 1u, F) : v;
		}
	} else {
		unsigned long long acc[16];
		for (int u = 0; u < 16; u++)
			acc[u] = 0;
		for (int t = 0; t < RP_ROWS; t++) {
			const int j = sJ[t];
			if (j < 0)
				continue;                    // (uniform)
			const uint32_t c = C[(int64_t) i * ld + j];
			if (c == 0)
				continue;
			const uint32_t neg = F.p - c;
			for (int u = 0; u < 16; u++)
				acc[u] += (unsigned long long) neg * sT[t][q * 16 + u];          // < 64 * 2^32: fits
		}
		for (int u = 0; u < 16; u++)
			out[u] = (uint32_t) (acc[u] % F.p);
	}
	unsigned int wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
	for (int u = 0; u < 16; u++) {
		int hi, lo;
		split_digits(out[u], F, hi, lo);
		wh[u >> 2] |= (unsigned int) (hi & 255) << (8 * (u & 3));
		wl[u >> 2] |= (unsigned int) (lo & 255) << (8 * (u & 3));
	}
	*reinterpret_cast<int4 *>(Mh + (int64_t) i * 64 + q * 16) = make_int4((int) wh[0], (int) wh[1], (int) wh[2], (int) wh[3]);
	*reinterpret_cast<int4 *>(Ml + (int64_t) i * 64 + q * 16) = make_int4((int) wl[0], (int) wl[1], (int) wl[2], (int) wl[3]);
}

// digit planes of M'[i][u] = - C[i][piv[u]]  (0 where piv[u] < 0 and on the rows [skip_lo, skip_hi)): what clears the pivot
// columns of up to 64 rows that are reduced among themselves from the rows of C.  blockIdx.y = set (its pivots: piv + 64 * set).
__global__ __launch_bounds__(256) void rowpanel_neg_columns(const uint32_t *C, int64_t ld, int n, int skip_lo, int skip_hi, const int *piv, signed char *Mplanes,
                                                            int64_t set_stride, int64_t low_offset, MontDev F)
{
	__shared__ int sJ[RP_ROWS];
	const int tid = threadIdx.x;
	if (tid < RP_ROWS)
		sJ[tid] = piv[blockIdx.y * RP_ROWS + tid];
	__syncthreads();
	const int i = blockIdx.x * 64 + (tid & 63), q = tid >> 6;          // row, quarter of the 64 multipliers
	if (i >= n)
		return;
	const bool skip = i >= skip_lo && i < skip_hi;
	unsigned int wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
	for (int u = 0; u < 16; u++) {
		const int j = sJ[q * 16 + u];
		const uint32_t c = (skip || j < 0) ? 0u : C[(int64_t) i * ld + j];
		int hi, lo;
		split_digits(c == 0 ? 0u : F.p - c, F, hi, lo);
		wh[u >> 2] |= (unsigned int) (hi & 255) << (8 * (u & 3));
		wl[u >> 2] |= (unsigned int) (lo & 255) << (8 * (u & 3));
	}
	signed char *Mh = Mplanes + blockIdx.y * set_stride, *Ml = Mh + low_offset;
	*reinterpret_cast<int4 *>(Mh + (int64_t) i * 64 + q * 16) = make_int4((int) wh[0], (int) wh[1], (int) wh[2], (int) wh[3]);
	*reinterpret_cast<int4 *>(Ml + (int64_t) i * 64 + q * 16) = make_int4((int) wl[0], (int) wl[1], (int) wl[2], (int) wl[3]);
}

// Z[i][t] = - Y[i][piv[t0 + t]]  for the rows of Y and a run of echelon pivots (step A), 0 beyond the run
__global__ __launch_bounds__(256) void rowpanel_gather_neg(const uint32_t *Y, int64_t ld, int n, const int *piv, int t0, int count, uint32_t *Z, int64_t ldz, MontDev F)
{
	const int i = blockIdx.x * 4 + (threadIdx.x >> 6), t = blockIdx.y * 64 + (threadIdx.x & 63);
	if (i >= n || t >= (int) ldz)
		return;
	uint32_t v = 0;
	if (t < count) {
		const uint32_t y = Y[(int64_t) i * ld + piv[t0 + t]];
		v = (y == 0) ? 0u : F.p - y;
	}
	Z[(int64_t) i * ldz + t] = v;
}

__global__ __launch_bounds__(256) void rowpanel_copy_rows(uint32_t *dst, int64_t ldd, const uint32_t *src, int64_t lds, int m, const int *src_row, int count)
{
	for (int t = blockIdx.x; t < count; t += gridDim.x)
		copy_row(dst + (int64_t) t * ldd, src + (int64_t) src_row[t] * lds, m, false);
}
}  // namespace

// is any entry of these rows not zero?  (one wave per row, 256 columns at a time; everybody leaves once somebody has said yes)
__global__ __launch_bounds__(256) void rows_any_nonzero_kernel(const uint32_t *P, int64_t ld, int rows, int m, int *out)
{
	const int lane = threadIdx.x & 63;
	const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (gridDim.x * 256) >> 6;
	for (int i = wave; i < rows; i += nwaves) {
		if (*(volatile int *) out != 0)
			return;
		// (sixteen loads in flight per lane: with four, a row of 71,154 columns was 278 dependent round trips)
		for (int j0 = 0; j0 < m; j0 += 1024) {
			bool nz = false;
#pragma unroll
			for (int u = 0; u < 16; u++) {
				const int j = j0 + 64 * u + lane;
				nz |= j < m && P[(int64_t) i * ld + j] != 0;
			}
			if (__ballot(nz) != 0) {
				if (lane == 0)
					atomicOr(out, 1);
				return;
			}
		}
	}
}

int device_echelon_extend(int64_t prime, int m, uint32_t *dM, int64_t ld, int k, int Sn, int *d_piv, hipStream_t stream)
{
	if (Sn <= 0 || m <= 0)
		return k;
	if (prime > 65279)
		die("device_echelon_extend: p = %lld (the row-panel echelon form runs on the matrix cores: p <= 65279)", (long long) prime);
	const MontDev F = to_dev(mont_setup(prime));
	const int n = k + Sn;
	uint32_t *Y = dM + (int64_t) k * ld;
	std::vector<void *> owned;
	auto dal = [&](size_t bytes) {
		void *ptr = big_alloc(bytes);
		owned.push_back(ptr);
		return ptr;
	};
	const int SETS = MAXSETS;
	signed char *Mplanes = (signed char *) dal((size_t) SETS * 2 * n * 64);               // per set: Mh [n][64] | Ml [n][64]
	signed char *Bplanes = (signed char *) dal((size_t) SETS * 2 * (size_t) m * 64);       // per set: Bh [m][64] | Bl [m][64]
	uint32_t *Z = (uint32_t *) dal((size_t) Sn * 64 * SETS * sizeof(uint32_t));
	int *d_cnt = (int *) dal(SETS * sizeof(int));
	int *d_state = (int *) dal(RP_ROWS * sizeof(int));
	int *d_left = (int *) dal(RP_ROWS * sizeof(int));
	int *d_newpiv = (int *) dal(RP_ROWS * sizeof(int));
	uint32_t *d_T = (uint32_t *) dal(RP_ROWS * RP_ROWS * sizeof(uint32_t));
	unsigned char *d_ispiv = (unsigned char *) dal((size_t) m);
	HIP_CHECK(hipMemsetAsync(d_ispiv, 0, (size_t) m, stream));
	unsigned short *d_invtab = (unsigned short *) dal(((size_t) prime * 2 + 79) / 16 * 16);          // inverses of 0 .. p - 1 (rowpanel_window)
	hipLaunchKernelGGL(rref_inverse_table, dim3(((unsigned) prime + 255) / 256), dim3(256), 0, stream, d_invtab, F);
	if (k > 0)
		hipLaunchKernelGGL(rowpanel_mark_pivots, dim3((k + 255) / 256), dim3(256), 0, stream, d_piv, k, d_ispiv);
	const size_t win_lds = ((size_t) RP_ROWS * (RP_COLS + 1) + (size_t) RP_ROWS * (RP_ROWS + 1)) * sizeof(uint32_t);
	static bool configured = false;
	if (!configured) {
		HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rowpanel_window), hipFuncAttributeMaxDynamicSharedMemorySize, (int) win_lds));
		configured = true;
	}
	auto planes_of_set = [&](int s, signed char *&Mh, signed char *&Ml, signed char *&Bh, signed char *&Bl) {
		Mh = Mplanes + (size_t) s * 2 * n * 64;
		Ml = Mh + (size_t) n * 64;
		Bh = Bplanes + (size_t) s * 2 * (size_t) m * 64;
		Bl = Bh + (size_t) m * 64;
	};

	const double t_start = wtime();
	int total_iters = 0;
	// ---- A. the rows of Y lose the pivot columns of E ----
	for (int t0 = 0; t0 < k; t0 += 64 * SETS) {
		const int count = std::min(k - t0, 64 * SETS);
		const int nsets = (count + 63) / 64;
		const int64_t ldz = (int64_t) 64 * nsets;
		hipLaunchKernelGGL(rowpanel_gather_neg, dim3((Sn + 3) / 4, nsets), dim3(256), 0, stream, Y, ld, Sn, d_piv, t0, count, Z, ldz, F);
		UpdSets S{};
		std::vector<int> cnt((size_t) SETS, 0);
		for (int s = 0; s < nsets; s++)
			cnt[s] = std::min(64, count - 64 * s);
		HIP_CHECK(hipMemcpyAsync(d_cnt, cnt.data(), SETS * sizeof(int), hipMemcpyHostToDevice, stream));
		// (the planes of M for n = Sn rows: the first Sn * 64 bytes of each half)
		hipLaunchKernelGGL(rref_split_Z, dim3((Sn + 63) / 64, nsets), dim3(256), 0, stream, Z, ldz, Sn, nsets, Mplanes, (int64_t) 2 * Sn * 64, F);
		for (int s = 0; s < nsets; s++) {
			signed char *Bh = Bplanes + (size_t) s * 2 * (size_t) m * 64, *Bl = Bh + (size_t) m * 64;
			hipLaunchKernelGGL(rref_split_B, dim3((m + 255) / 256), dim3(256), 0, stream, dM + (int64_t) (t0 + 64 * s) * ld, ld, m, d_cnt + s, Bh, Bl, F);
			S.Mh[s] = Mplanes + (size_t) s * 2 * Sn * 64;
			S.Ml[s] = S.Mh[s] + (size_t) Sn * 64;
			S.Bh[s] = Bh;
			S.Bl[s] = Bl;
		}
		S.nsets = nsets;
		hipLaunchKernelGGL(rref_update_mfma_multi, dim3((m + 63) / 64, (Sn + 63) / 64), dim3(256), 0, stream, Y, ld, Sn, 0, m, S, F);
		HIP_CHECK(hipStreamSynchronize(stream));          // (cnt dies here)
	}

	const double t_A = wtime();
	// ---- B. panels of 64 rows of Y ----
	// A panel in progress only touches its own 64 rows; the rows of a finished panel are reduced among themselves, so its
	// pivot columns leave every other row with the entries of these columns as multipliers, in one pass.  Finished panels
	// wait (at most SETS of them: rows [pend_base, pend_base + 64 npend) of Y, pivots in d_pend) and are kept reduced by
	// one another -- (i) and (iii) below, on 64 (npend) rows --; the pass over the whole stack, which is what costs, is
	// made once for SETS panels, with SETS x 64 multipliers a row.
	std::vector<int> pivot_of((size_t) Sn, -1);          // pivot column of every row of Y, -1: none (the row became zero)
	const int panel_rows[2] = {64, Sn % RP_ROWS};          // d_cnt[0]: a full panel, d_cnt[1]: the last, short one
	HIP_CHECK(hipMemcpyAsync(d_cnt, panel_rows, sizeof(panel_rows), hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	int *d_pend = (int *) dal((size_t) SETS * RP_ROWS * sizeof(int));
	int npend = 0, pend_base = 0, passes = 0;
	// The blocks of a low-rank finish: 4,096 combinations of rank 70 -- two panels find the pivots, sixty-two find their rows zero,
	// one launch and one round trip to the host each (55-120 us: 3.5-5 ms per block, 775 of the 1,000 small copies of a driver call
	// on mk15.b4); and the block after the last pivots is zero from the start.  So: is everything from row r_from on zero already?
	// Asked once before the panels, and again at the first panel that comes out all zero after pivots have been found (with the
	// waiting panels passed on first, which is what makes the rows behind them zero).
	int *d_any = (int *) dal(sizeof(int));
	int zero_checks = 0;
	auto rest_is_zero = [&](int r_from) -> bool {
		if (r_from >= Sn)
			return true;
		int any = 1;
		HIP_CHECK(hipMemsetAsync(d_any, 0, sizeof(int), stream));
		hipLaunchKernelGGL(rows_any_nonzero_kernel, dim3(512), dim3(256), 0, stream, Y + (int64_t) r_from * ld, ld, Sn - r_from, m, d_any);
		HIP_CHECK(hipMemcpyAsync(&any, d_any, sizeof(int), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		zero_checks += 1;
		return any == 0;
	};
	bool found_since_check = false;
	const bool all_zero_at_start = rest_is_zero(0);
	auto split_panel = [&](int row0, int set) {          // the rows [row0, row0 + 64) of Y as the B planes of `set`
		signed char *Mh, *Ml, *Bh, *Bl;
		planes_of_set(set, Mh, Ml, Bh, Bl);
		hipLaunchKernelGGL(rref_split_B, dim3((m + 255) / 256), dim3(256), 0, stream, Y + (int64_t) row0 * ld, ld, m, d_cnt + (Sn - row0 >= RP_ROWS ? 0 : 1), Bh, Bl, F);
	};
	auto clear_columns = [&](uint32_t *C, int rows, int skip_lo, int skip_hi, const int *piv, int nsets) {          // C -= C[:, piv] * (B planes of the sets)
		hipLaunchKernelGGL(rowpanel_neg_columns, dim3((rows + 63) / 64, nsets), dim3(256), 0, stream, C, ld, rows, skip_lo, skip_hi, piv, Mplanes, (int64_t) 2 * n * 64,
		                   (int64_t) n * 64, F);
		UpdSets S{};
		for (int s = 0; s < nsets; s++) {
			signed char *Mh, *Ml, *Bh, *Bl;
			planes_of_set(s, Mh, Ml, Bh, Bl);
			S.Mh[s] = Mh;
			S.Ml[s] = Ml;
			S.Bh[s] = Bh;
			S.Bl[s] = Bl;
		}
		S.nsets = nsets;
		hipLaunchKernelGGL(rref_update_mfma_multi, dim3((m + 63) / 64, (rows + 63) / 64), dim3(256), 0, stream, C, ld, rows, 0, m, S, F);
	};
	auto flush = [&]() {          // the pivots of the waiting panels leave every other row of the stack
		if (npend == 0)
			return;
		for (int s = 0; s < npend; s++)
			split_panel(pend_base + RP_ROWS * s, s);
		clear_columns(dM, n, k + pend_base, k + pend_base + RP_ROWS * npend, d_pend, npend);
		passes += 1;
		npend = 0;
	};
	// A block of FULL rank is the other extreme (the 4,096 combinations of a finish whose remainder still has thousands of
	// dimensions: the no-greedy flow on mk13.b5, four blocks of 4,096 x 17,185): every row panel finds its 64 pivots, and 64 of them
	// one after the other -- window, multipliers, updates, two round trips to the host each -- took 62-70 ms per block where the
	// column-panel RREF of the same rows takes 5.  After two full panels in a row the rest of the block goes to device_rref (the
	// waiting panels are passed on first, so that the rest is reduced by everything found so far), and the new pivot columns
	// leave the rows above in passes of 512.  Only where the block is not much wider than tall: on a remainder of 70,000 columns
	// a rest of low rank would pay a try per 64 columns.
	int full_run = 0, rest_rows = 0, rest_rank = 0;
	auto rest_by_columns = [&](int rest0) {
		const int nr = Sn - rest0;
		uint32_t *Yr = Y + (int64_t) rest0 * ld;
		int *d_pivr = (int *) dal((size_t) m * sizeof(int));
		int *d_pad = (int *) dal((size_t) SETS * RP_ROWS * sizeof(int));
		const int rnew = device_rref(prime, nr, m, Yr, ld, d_pivr, stream, 1, nullptr);
		rest_rows = nr;
		rest_rank = rnew;
		if (rnew <= 0)
			return;
		std::vector<int> pv((size_t) rnew);
		HIP_CHECK(hipMemcpyAsync(pv.data(), d_pivr, (size_t) rnew * sizeof(int), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		for (int t = 0; t < rnew; t++)
			pivot_of[(size_t) rest0 + t] = pv[(size_t) t];
		const int above = k + rest0;          // E and the panels done: they lose the new pivot columns
		for (int t0 = 0; t0 < rnew && above > 0; t0 += RP_ROWS * SETS) {
			const int count = std::min(rnew - t0, RP_ROWS * SETS), nsets = (count + RP_ROWS - 1) / RP_ROWS;
			std::vector<int> pad((size_t) SETS * RP_ROWS, -1);
			for (int t = 0; t < count; t++)
				pad[(size_t) t] = pv[(size_t) t0 + t];
			const int last = count - RP_ROWS * (nsets - 1);
			HIP_CHECK(hipMemcpyAsync(d_pad, pad.data(), pad.size() * sizeof(int), hipMemcpyHostToDevice, stream));
			HIP_CHECK(hipMemcpyAsync(d_cnt + 2, &last, sizeof(int), hipMemcpyHostToDevice, stream));
			for (int s2 = 0; s2 < nsets; s2++) {
				signed char *Mh, *Ml, *Bh, *Bl;
				planes_of_set(s2, Mh, Ml, Bh, Bl);
				hipLaunchKernelGGL(rref_split_B, dim3((m + 255) / 256), dim3(256), 0, stream, Yr + (int64_t) (t0 + RP_ROWS * s2) * ld, ld, m, d_cnt + (s2 + 1 < nsets ? 0 : 2), Bh, Bl, F);
			}
			clear_columns(dM, above, -1, -1, d_pad, nsets);
			HIP_CHECK(hipStreamSynchronize(stream));          // (pad and last die here)
			passes += 1;
		}
	};
	for (int r0 = 0; r0 < Sn && !all_zero_at_start; r0 += RP_ROWS) {
		const int rows_here = std::min(RP_ROWS, Sn - r0);
		uint32_t *P = Y + (int64_t) r0 * ld;
		if (npend > 0) {          // (i) the panel loses the pivot columns of the waiting panels
			for (int s = 0; s < npend; s++)
				split_panel(pend_base + RP_ROWS * s, s);
			clear_columns(P, rows_here, -1, -1, d_pend, npend);
		}
		int state[RP_ROWS];
		for (int t = 0; t < RP_ROWS; t++)
			state[t] = (t < rows_here) ? -1 : -2;
		HIP_CHECK(hipMemcpyAsync(d_state, state, sizeof(state), hipMemcpyHostToDevice, stream));
		// (the leftmost entries for the NEXT step are looked up behind the update of this one, on the device's own copy of the
		//  states: one round trip to the host per step instead of two)
		auto look = [&]() {
			HIP_CHECK(hipMemsetAsync(d_left, 0x7F, RP_ROWS * sizeof(int), stream));          // 0x7F7F7F7F: larger than any column
			hipLaunchKernelGGL(rowpanel_leftmost, dim3((m + 1023) / 1024, rows_here), dim3(64), 0, stream, P, ld, m, d_state, d_left);
		};
		int left[RP_ROWS];
		look();
		HIP_CHECK(hipMemcpyAsync(left, d_left, sizeof(left), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		for (int iter = 0;; iter++) {          // (ii) its rows find their pivots
			bool changed = false, waiting = false;
			for (int t = 0; t < rows_here; t++) {
				if (state[t] != -1)
					continue;
				if (left[t] >= m) {
					state[t] = -2;           // the row is zero: it depended on the rows before it
					changed = true;
				} else {
					waiting = true;
				}
			}
			if (changed)
				HIP_CHECK(hipMemcpyAsync(d_state, state, sizeof(state), hipMemcpyHostToDevice, stream));
			if (!waiting)
				break;                       // every row of the panel has a pivot or is zero
			if (iter > RP_ROWS + 2)
				die("device_echelon_extend: a panel did not finish in %d window steps", iter);
			// the OLD rows of the panel as digit planes, then T and the new pivots, then P <- T P
			signed char *Mh, *Ml, *Bh, *Bl;
			planes_of_set(0, Mh, Ml, Bh, Bl);
			split_panel(r0, 0);
			hipLaunchKernelGGL(rowpanel_window, dim3(1), dim3(RP_THREADS), win_lds, stream, P, ld, m, rows_here, d_left, d_state, d_newpiv, d_T, d_ispiv, F, d_invtab);
			total_iters += 1;
			hipLaunchKernelGGL(rowpanel_multipliers, dim3(1), dim3(256), 0, stream, P, ld, rows_here, 0, d_newpiv, d_T, Mh, Ml, F);
			UpdSets S{};
			S.Mh[0] = Mh;
			S.Ml[0] = Ml;
			S.Bh[0] = Bh;
			S.Bl[0] = Bl;
			S.nsets = 1;
			hipLaunchKernelGGL(rref_update_mfma_multi, dim3((m + 63) / 64, 1), dim3(256), 0, stream, P, ld, rows_here, 0, m, S, F);
			look();
			HIP_CHECK(hipMemcpyAsync(state, d_state, sizeof(state), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipMemcpyAsync(left, d_left, sizeof(left), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
		}
		HIP_CHECK(hipStreamSynchronize(stream));          // (state may have just gone up)
		for (int t = 0; t < rows_here; t++)
			pivot_of[(size_t) r0 + t] = (state[t] >= 0) ? state[t] : -1;
		bool any = false;
		for (int t = 0; t < rows_here; t++)
			any = any || state[t] >= 0;
		if (!any && found_since_check) {
			// the first panel of zero rows behind panels with pivots: those pivots leave the whole stack now, and if that
			// leaves nothing in the rows behind this panel, the block is done
			flush();
			found_since_check = false;
			if (rest_is_zero(r0 + RP_ROWS))
				break;
		}
		found_since_check = found_since_check || any;
		{
			int got = 0;
			for (int t = 0; t < rows_here; t++)
				got += state[t] >= 0;
			full_run = (got == RP_ROWS) ? full_run + 1 : 0;
		}
		if (!any && npend == 0)
			continue;                        // nothing but zero rows, and no panel waits: nothing to pass on
		if (npend > 0) {          // (iii) the waiting panels lose its pivot columns
			split_panel(r0, 0);
			clear_columns(Y + (int64_t) pend_base * ld, RP_ROWS * npend, -1, -1, d_state, 1);
		} else {
			pend_base = r0;
		}
		HIP_CHECK(hipMemcpyAsync(d_pend + (size_t) npend * RP_ROWS, d_state, RP_ROWS * sizeof(int), hipMemcpyDeviceToDevice, stream));
		npend += 1;
		if (npend == SETS)
			flush();
		const int rest0 = r0 + RP_ROWS;
		if (full_run >= 2 && Sn - rest0 >= 512 && (int64_t) m <= 8 * (int64_t) (Sn - rest0)) {
			flush();
			rest_by_columns(rest0);
			break;
		}
	}
	flush();
	HIP_CHECK(hipStreamSynchronize(stream));

	const double t_B = wtime();
	// ---- C. the rows with a pivot move up behind E ----
	std::vector<int> src, piv_new;
	for (int i = 0; i < Sn; i++)
		if (pivot_of[i] >= 0) {
			src.push_back(i);
			piv_new.push_back(pivot_of[i]);
		}
	const int rr = (int) src.size();
	bool in_place = true;
	for (int t = 0; t < rr; t++)
		in_place = in_place && src[t] == t;
	if (rr > 0 && !in_place) {
		uint32_t *tmp = (uint32_t *) dal((size_t) rr * m * sizeof(uint32_t));
		int *d_src = (int *) dal((size_t) rr * sizeof(int));
		HIP_CHECK(hipMemcpyAsync(d_src, src.data(), (size_t) rr * sizeof(int), hipMemcpyHostToDevice, stream));
		hipLaunchKernelGGL(rowpanel_copy_rows, dim3(std::min(rr, 4096)), dim3(256), 0, stream, tmp, (int64_t) m, Y, ld, m, d_src, rr);
		HIP_CHECK(hipMemcpy2DAsync(Y, (size_t) ld * sizeof(uint32_t), tmp, (size_t) m * sizeof(uint32_t), (size_t) m * sizeof(uint32_t), (size_t) rr,
		                           hipMemcpyDeviceToDevice, stream));
	}
	if (rr > 0)
		HIP_CHECK(hipMemcpyAsync(d_piv + k, piv_new.data(), (size_t) rr * sizeof(int), hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	for (void *ptr : owned)
		big_free(ptr);
	if (verbose() >= 2)
		logmsg("[echelon rows] %d + %d rows x %d: reduction by the echelon rows %.1f ms, %d panels in %d steps and %d passes (%d looks at the rows left; %d rows by column panels: %d pivots) %.1f ms, compaction %.1f ms; %d new\n",
		       k, Sn, m, 1e3 * (t_A - t_start), (Sn + RP_ROWS - 1) / RP_ROWS, total_iters, passes, zero_checks, rest_rows, rest_rank, 1e3 * (t_B - t_A), 1e3 * (wtime() - t_B), rr);
	return k + rr;
}

}  // namespace sh

// --------------------------------------------------------------------------
// Random linear combinations of sparse rows, formed on the device (the first
// half of spasm_schur_dense_randomized, spasm_schur.c:380-400): Y[k, :] =
// sum_t c(k,t) A[rows(k,t), :] into a dense 64-bit accumulator, then packed
// to CSR so that the dense-row elimination kernel can take it as input.
// Coefficients come from a counter-based generator (splitmix64 of (salt, k,
// t)); the reference's own stream depends on rand() and thread timing.
// --------------------------------------------------------------------------
namespace sh {

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

__device__ __forceinline__ uint32_t uniform_below(uint64_t h, uint32_t bound)
{
	return (uint32_t) (((h >> 32) * (uint64_t) bound) >> 32);
}

}  // namespace

// one wave per (combination k, term t).  w > 0: term t picks a random row of the list (coefficient 1
// for t = 0); w == 0: term t is row t of the list, every row taken, random coefficient.
// (colmap != nullptr: the accumulators only span the non-pivotal columns -- column j goes to colmap[j] - base; the caller has
//  made sure that the rows hold no other column.  The same in the two kernels below.)
//  MAPPED is a template parameter, not a test of the pointer: with `colmap != nullptr ? colmap[j] - base : j` the compiler
//  turned the choice into a select and loaded colmap[j] from a null pointer.)
template <bool MAPPED>
__global__ __launch_bounds__(64) void combine_rows_kernel(const int64_t *Ap, const int *Aj, const int *Ax, const int *rows,
                                                         int nrows, int N, int w, int m, uint64_t salt,
                                                         unsigned long long *Y, MontDev F, const uint32_t *colmap, uint32_t base)
{
	const int lane = threadIdx.x;
	const int64_t terms = (w > 0) ? (int64_t) w : (int64_t) nrows;
	const int64_t total = (int64_t) N * terms;
	for (int64_t task = blockIdx.x; task < total; task += gridDim.x) {
		const int k = (int) (task / terms);
		const int64_t t = task % terms;
		const uint64_t h = mix64(salt ^ mix64(((uint64_t) k << 32) ^ (uint64_t) t));
		int i;
		uint32_t coeff;
		if (w > 0) {
			i = rows[uniform_below(mix64(h ^ 0x51ED270B1ULL), (uint32_t) nrows)];
			coeff = (t == 0) ? 1u : uniform_below(h, F.p);
		} else {
			i = rows[t];
			coeff = uniform_below(h, F.p);
		}
		if (coeff == 0)
			continue;
		const uint32_t cm = montmul(coeff, F.r2, F);          // coefficient in Montgomery form
		unsigned long long *Yk = Y + (int64_t) k * m;
		for (int64_t px = Ap[i] + lane; px < Ap[i + 1]; px += 64) {
			const int a = Ax[px];
			const uint32_t v = (a < 0) ? (uint32_t) a + F.p : (uint32_t) a;
			const int j = Aj[px];
			int jc = j;
			if constexpr (MAPPED)
				jc = (int) (colmap[j] - base);
			atomicAdd(&Yk[jc], (unsigned long long) montmul(cm, v % F.p, F));
		}
	}
}

// The few combinations of ALL the rows that end a low-rank finish (w == 0, N <= 16: spasm_echelonize_test_completion takes
// nine), over a large input -- the 673,000 rows and 1.06e9 entries of mk14.b4's Schur complement: one wave per (combination,
// row) reads every row N times and sends N atomic requests per entry, 6.3e9 of them, 0.16-0.19 s at the rate the memory
// side takes them.  Here a wave takes a row ONCE, sixteen lanes per entry -- one per combination -- and the accumulator is
// laid out [column][16]: the sixteen adds of an entry fall into one 128-byte line and travel as one or two requests.
// transpose_combinations_kernel puts the result where the other kernels expect it ([combination][column]).
template <bool MAPPED>
__global__ __launch_bounds__(64) void combine_all_rows_kernel(const int64_t *Ap, const int *Aj, const int *Ax, const int *rows, int nrows, int N, uint64_t salt,
                                                             unsigned long long *Yt, MontDev F, const uint32_t *colmap, uint32_t base)
{
	const int lane = threadIdx.x, k = lane & 15, e = lane >> 4;
	for (int64_t t = blockIdx.x; t < nrows; t += gridDim.x) {
		const uint64_t h = mix64(salt ^ mix64(((uint64_t) k << 32) ^ (uint64_t) t));          // (the generator of combine_rows_kernel)
		const uint32_t coeff = (k < N) ? uniform_below(h, F.p) : 0u;
		const uint32_t cm = montmul(coeff, F.r2, F);
		const int i = rows[t];
		for (int64_t px = Ap[i] + e; px < Ap[i + 1]; px += 4) {
			const int a = Ax[px];
			const uint32_t v = (a < 0) ? (uint32_t) a + F.p : (uint32_t) a;
			int j = Aj[px];
			if constexpr (MAPPED)
				j = (int) (colmap[j] - base);
			if (coeff != 0)
				atomicAdd(&Yt[(int64_t) j * 16 + k], (unsigned long long) montmul(cm, v % F.p, F));
		}
	}
}

// The same for p < 2^16 WITHOUT an atomic per term at the memory side (92 ms on mk14.b4: 1.4e9 atomic requests, the largest
// kernel of the whole call).  A workgroup takes CB_ROWS rows and walks the columns in blocks of CB_COLS: rows are sorted by
// column, so every row keeps a cursor; the terms of a block are added up in LDS -- 32-bit sums: a term is < 2^16 and a
// workgroup adds at most CB_ROWS = 512 of them to a sum -- and only the sums that are not zero go to Y ([combination]
// [column], 64-bit), one atomic each: 42,000 occupied columns x 9 per workgroup instead of a thousand entries x 9 per row.
// Needs sorted rows (a row with a column out of order falls behind its cursor: the caller checks the flag and redoes the
// batch with combine_all_rows_kernel).
#ifndef SPASM_CB_ROWS
#define SPASM_CB_ROWS 512
#endif
#ifndef SPASM_CB_COLS
#define SPASM_CB_COLS 1024
#endif
#ifndef SPASM_CB_THREADS
#define SPASM_CB_THREADS 512
#endif
// (round 5: 512 rows by 512 threads, one row per thread, three workgroups per CU.  With 2,048 rows by 256 threads -- eight rows per
//  thread, 8 waves per CU -- the kernel took 31 ms on the 8.2e8 entries of mk15.b4's Schur complement and 13 on mk14.b4's; a thread's
//  next entries stand in registers, so what counts is the waves in flight: 1,024 x 512: 15.4 / 7.2, 512 x 512: 13.7 / 4.0,
//  1,024 x 1,024: 19.7 / 4.7, column blocks of 512 or 2,048: no better)
constexpr int CB_ROWS = SPASM_CB_ROWS, CB_COLS = SPASM_CB_COLS, CB_THREADS = SPASM_CB_THREADS, CB_PER_THREAD = CB_ROWS / CB_THREADS;

template <bool MAPPED>
__global__ __launch_bounds__(CB_THREADS) void combine_all_rows_blocked_kernel(const int64_t *Ap, const int *Aj, const int *Ax, const int *rows, int nrows, int N, int m,
                                                                              uint64_t salt, unsigned long long *Y, int *unsorted, MontDev F, const uint32_t *colmap,
                                                                              uint32_t cbase, int64_t annz)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t cb_lds[];
	uint32_t *acc = cb_lds;                                                            // [CB_COLS][N]
	unsigned short *coef = reinterpret_cast<unsigned short *>(cb_lds + CB_COLS * N);     // [CB_ROWS][N], Montgomery form
	const int tid = threadIdx.x;
	const int64_t base = (int64_t) blockIdx.x * CB_ROWS;
	for (int e = tid; e < CB_ROWS * N; e += CB_THREADS) {
		const int64_t t = base + e / N;
		const int k = e % N;
		uint32_t cm = 0;
		if (t < nrows) {
			const uint64_t h = mix64(salt ^ mix64(((uint64_t) k << 32) ^ (uint64_t) t));          // (the generator of combine_rows_kernel)
			cm = montmul(uniform_below(h, F.p), F.r2, F);
		}
		coef[e] = (unsigned short) cm;
	}
	for (int e = tid; e < CB_COLS * N; e += CB_THREADS)
		acc[e] = 0;
	int64_t cur[CB_PER_THREAD], end[CB_PER_THREAD];
	int prev[CB_PER_THREAD];
	// A thread walks ITS rows: with one 4-byte load per entry every load instruction of a wave touched 64 different lines for
	// 256 useful bytes, and the lines did not survive in the L1 until their next entry was wanted (2,048 rows x two arrays per
	// workgroup) -- 48 ms for the 1e9 entries of mk15.b4's Schur complement, sixteen times the bytes.  The next four entries of
	// every row are kept in registers, fetched by aligned 16-byte loads (a row's entries are contiguous).
	int4 bj[CB_PER_THREAD], bx[CB_PER_THREAD];
	auto refill = [&](int q) {
		const int64_t b = cur[q] & ~(int64_t) 3;
		if (b + 4 <= annz) {
			bj[q] = *reinterpret_cast<const int4 *>(Aj + b);
			bx[q] = *reinterpret_cast<const int4 *>(Ax + b);
		} else {          // (the last entries of the arrays: one by one)
			bj[q] = make_int4(b < annz ? Aj[b] : 0, b + 1 < annz ? Aj[b + 1] : 0, b + 2 < annz ? Aj[b + 2] : 0, 0);
			bx[q] = make_int4(b < annz ? Ax[b] : 0, b + 1 < annz ? Ax[b + 1] : 0, b + 2 < annz ? Ax[b + 2] : 0, 0);
		}
	};
#pragma unroll
	for (int q = 0; q < CB_PER_THREAD; q++) {
		const int64_t t = base + tid + (int64_t) q * CB_THREADS;
		const int i = (t < nrows) ? rows[t] : 0;
		cur[q] = (t < nrows) ? Ap[i] : 0;
		end[q] = (t < nrows) ? Ap[i + 1] : 0;
		prev[q] = -1;
		bj[q] = bx[q] = make_int4(0, 0, 0, 0);
		if (cur[q] < end[q])
			refill(q);
	}
	__syncthreads();
	bool bad = false;
	for (int lo = 0; lo < m; lo += CB_COLS) {
		const int hi = min(m, lo + CB_COLS);
		bool any = false;
#pragma unroll
		for (int q = 0; q < CB_PER_THREAD; q++) {
			const unsigned short *cf = coef + (size_t) (tid + q * CB_THREADS) * N;
			while (cur[q] < end[q]) {
				const int sub = (int) (cur[q] & 3);
				int j = (sub == 0) ? bj[q].x : (sub == 1) ? bj[q].y : (sub == 2) ? bj[q].z : bj[q].w;
				if constexpr (MAPPED)
					j = (int) (colmap[j] - cbase);          // (monotone in the column: rows stay sorted)
				if (j >= hi)
					break;
				bad = bad || j <= prev[q];
				prev[q] = j;
				const int a = (sub == 0) ? bx[q].x : (sub == 1) ? bx[q].y : (sub == 2) ? bx[q].z : bx[q].w;
				const uint32_t v = ((a < 0) ? (uint32_t) a + F.p : (uint32_t) a) % F.p;
				if (j >= lo)
					for (int k = 0; k < N; k++)
						atomicAdd(&acc[(j - lo) * N + k], montmul((uint32_t) cf[k], v, F));
				cur[q] += 1;
				any = true;
				if (sub == 3 && cur[q] < end[q])
					refill(q);
			}
		}
		// (a block nobody of this workgroup touched: nothing to flush, nothing to clear)
		if (__syncthreads_or(any)) {
			for (int e = tid; e < (hi - lo) * N; e += CB_THREADS) {
				const uint32_t sum = acc[e];
				if (sum != 0) {
					atomicAdd(&Y[(int64_t) (e % N) * m + lo + e / N], (unsigned long long) sum);
					acc[e] = 0;
				}
			}
			__syncthreads();
		}
	}
	if (bad)
		atomicOr(unsorted, 1);
}

__global__ __launch_bounds__(256) void transpose_combinations_kernel(const unsigned long long *Yt, int N, int m, unsigned long long *Y)
{
	const int64_t idx = (int64_t) blockIdx.x * 256 + threadIdx.x;          // (column, combination): reads are contiguous
	if (idx >= (int64_t) m * 16)
		return;
	const int j = (int) (idx >> 4), k = (int) (idx & 15);
	if (k < N)
		Y[(int64_t) k * m + j] = Yt[idx];
}

// per row of the dense accumulator: number of entries that are non-zero mod p
__global__ __launch_bounds__(256) void dense_count_kernel(const unsigned long long *Y, int N, int m, uint32_t p, int *row_len)
{
	__shared__ int part[256];
	const int k = blockIdx.x;
	int c = 0;
	for (int j = threadIdx.x; j < m; j += 256)
		c += (Y[(int64_t) k * m + j] % p) != 0;
	part[threadIdx.x] = c;
	__syncthreads();
	for (int d = 128; d > 0; d >>= 1) {
		if ((int) threadIdx.x < d)
			part[threadIdx.x] += part[threadIdx.x + d];
		__syncthreads();
	}
	if (threadIdx.x == 0)
		row_len[k] = part[0];
}

// pack row k (sorted by column); one workgroup per row
template <bool MAPPED>
__global__ __launch_bounds__(256) void dense_pack_kernel(const unsigned long long *Y, int N, int m, uint32_t p,
                                                        const int64_t *Sp, int *Sj, int *Sx, const int *unmap)
{
	__shared__ int wave_tot[4];
	const int k = blockIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	int64_t wpos = Sp[k];
	for (int base = 0; base < m; base += 256) {
		const int j = base + threadIdx.x;
		const uint32_t v = (j < m) ? (uint32_t) (Y[(int64_t) k * m + j] % p) : 0u;
		const bool keep = v != 0;
		const uint64_t mk = __ballot(keep);
		if (lane == 0)
			wave_tot[wave] = __popcll(mk);
		__syncthreads();
		int before = 0, all = 0;
		for (int wv = 0; wv < 4; wv++) {
			if (wv < wave)
				before += wave_tot[wv];
			all += wave_tot[wv];
		}
		if (keep) {
			const int64_t dst = wpos + before + __popcll(mk & ((1ull << lane) - 1ull));
			int col = j;
			if constexpr (MAPPED)          // (accumulators over the non-pivotal columns only: back to the column of the matrix)
				col = unmap[j];
			Sj[dst] = col;
			// balanced representative (for p > 2^31 a residue >= 2^31 would read back as a negative int and be shifted by p)
			Sx[dst] = (v > p / 2) ? (int) (v - p) : (int) v;
		}
		wpos += all;
		__syncthreads();
	}
}

// ---- echelon rows (k x m words, values in [0, p), pivot of row t on column piv[t] with value 1) as rows of U ----
// entries of row t: the pivot first, then the others in increasing column; columns go through q (index among the non-
// pivotal columns of the factor the rows were reduced by -> column of the matrix); balanced representatives.
__global__ __launch_bounds__(256) void echelon_count_kernel(const uint32_t *M, int64_t ld, int m, int k, int *row_len)
{
	__shared__ int part[256];
	const int t = blockIdx.x;
	int c = 0;
	for (int j = threadIdx.x; j < m; j += 256)
		c += M[(int64_t) t * ld + j] != 0;
	part[threadIdx.x] = c;
	__syncthreads();
	for (int d = 128; d > 0; d >>= 1) {
		if ((int) threadIdx.x < d)
			part[threadIdx.x] += part[threadIdx.x + d];
		__syncthreads();
	}
	if (threadIdx.x == 0)
		row_len[t] = part[0];
}

__global__ __launch_bounds__(256) void echelon_pack_kernel(const uint32_t *M, int64_t ld, int m, int k, const int *piv, const int *q, uint32_t p,
                                                           const int64_t *Sp, int *Uj, int *Ux)
{
	__shared__ int wave_tot[4];
	const int t = blockIdx.x;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int jp = piv[t];
	int64_t wpos = Sp[t];
	if (threadIdx.x == 0) {
		Uj[wpos] = q[jp];
		Ux[wpos] = 1;
	}
	wpos += 1;
	for (int base = 0; base < m; base += 256) {
		const int j = base + threadIdx.x;
		const uint32_t v = (j < m && j != jp) ? M[(int64_t) t * ld + j] : 0u;
		const bool keep = v != 0;
		const uint64_t mk = __ballot(keep);
		if (lane == 0)
			wave_tot[wave] = __popcll(mk);
		__syncthreads();
		int before = 0, all = 0;
		for (int wv = 0; wv < 4; wv++) {
			if (wv < wave)
				before += wave_tot[wv];
			all += wave_tot[wv];
		}
		if (keep) {
			const int64_t dst = wpos + before + __popcll(mk & ((1ull << lane) - 1ull));
			Uj[dst] = q[j];
			Ux[dst] = (v > p / 2) ? (int) (v - p) : (int) v;
		}
		wpos += all;
		__syncthreads();
	}
}

void launch_echelon_count(const uint32_t *M, int64_t ld, int m, int k, int *row_len, hipStream_t stream)
{
	if (k > 0)
		hipLaunchKernelGGL(echelon_count_kernel, dim3(k), dim3(256), 0, stream, M, ld, m, k, row_len);
}

void launch_echelon_pack(const uint32_t *M, int64_t ld, int m, int k, const int *piv, const int *q, uint32_t p, const int64_t *Sp, int *Uj,
                         int *Ux, hipStream_t stream)
{
	if (k > 0)
		hipLaunchKernelGGL(echelon_pack_kernel, dim3(k), dim3(256), 0, stream, M, ld, m, k, piv, q, p, Sp, Uj, Ux);
}

void launch_combine(const int64_t *Ap, const int *Aj, const int *Ax, const int *rows, int nrows, int N, int w, int m,
                    uint64_t salt, unsigned long long *Y, const Mont &M, hipStream_t stream, const uint32_t *colmap, uint32_t base, int64_t annz)
{
	const int64_t terms = (w > 0) ? (int64_t) w : (int64_t) nrows;
	const int64_t total = (int64_t) N * terms;
	if (w <= 0 && N <= 16 && nrows >= 4096 && M.p < 65536) {
		// every row, few combinations, p < 2^16: block sums in LDS, one atomic per occupied (column, combination) and workgroup
		const size_t lds = (size_t) CB_COLS * N * sizeof(uint32_t) + (size_t) CB_ROWS * N * sizeof(unsigned short);
		static size_t configured = 0;
		if (lds > configured) {
			HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&combine_all_rows_blocked_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
			HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&combine_all_rows_blocked_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
			configured = lds;
		}
		int *d_flag = nullptr;
		HIP_CHECK(sh::malloc_or_trim((void **) &d_flag, sizeof(int)));
		HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(int), stream));
		if (colmap != nullptr)
			hipLaunchKernelGGL(combine_all_rows_blocked_kernel<true>, dim3((unsigned) ((nrows + CB_ROWS - 1) / CB_ROWS)), dim3(CB_THREADS), lds, stream, Ap, Aj, Ax, rows, nrows,
			                   N, m, salt, Y, d_flag, to_dev(M), colmap, base, annz);
		else
			hipLaunchKernelGGL(combine_all_rows_blocked_kernel<false>, dim3((unsigned) ((nrows + CB_ROWS - 1) / CB_ROWS)), dim3(CB_THREADS), lds, stream, Ap, Aj, Ax, rows, nrows,
			                   N, m, salt, Y, d_flag, to_dev(M), colmap, base, annz);
		HIP_CHECK(hipGetLastError());
		int flag = 0;
		HIP_CHECK(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		sh::big_free(d_flag);
		if (flag == 0 && sh::env_get("SPASM_HIP_COMBINE_CHECK")) {
			// (tests) the same combinations by the kernel that sends an atomic per term: the sums must agree mod p
			const size_t count = (size_t) N * m;
			unsigned long long *Y2 = (unsigned long long *) big_alloc(count * sizeof(unsigned long long));
			unsigned long long *Yt = (unsigned long long *) big_alloc((size_t) m * 16 * sizeof(unsigned long long));
			HIP_CHECK(hipMemsetAsync(Y2, 0, count * sizeof(unsigned long long), stream));
			HIP_CHECK(hipMemsetAsync(Yt, 0, (size_t) m * 16 * sizeof(unsigned long long), stream));
			if (colmap != nullptr)
				hipLaunchKernelGGL(combine_all_rows_kernel<true>, dim3((unsigned) std::min<int64_t>(nrows, 65536)), dim3(64), 0, stream, Ap, Aj, Ax, rows, nrows, N, salt, Yt,
				                   to_dev(M), colmap, base);
			else
				hipLaunchKernelGGL(combine_all_rows_kernel<false>, dim3((unsigned) std::min<int64_t>(nrows, 65536)), dim3(64), 0, stream, Ap, Aj, Ax, rows, nrows, N, salt, Yt,
				                   to_dev(M), colmap, base);
			hipLaunchKernelGGL(transpose_combinations_kernel, dim3((unsigned) (((int64_t) m * 16 + 255) / 256)), dim3(256), 0, stream, Yt, N, m, Y2);
			std::vector<unsigned long long> a(count), b(count);
			HIP_CHECK(hipMemcpyAsync(a.data(), Y, count * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipMemcpyAsync(b.data(), Y2, count * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
			size_t differ = 0, nonzero = 0;
			for (size_t t = 0; t < count; t++) {
				differ += (a[t] % M.p) != (b[t] % M.p);
				nonzero += (a[t] % M.p) != 0;
			}
			logmsg("[combine check] %d combinations of %d rows: %zu sums differ mod p (%zu non-zero)\n", N, nrows, differ, nonzero);
			if (differ != 0)
				die("combine_all_rows_blocked_kernel and combine_all_rows_kernel disagree on %zu sums", differ);
			big_free(Y2);
			big_free(Yt);
		}
		if (flag == 0)
			return;
		// (rows that are not sorted by column: Y holds part of the sums -- start again, the other way)
		HIP_CHECK(hipMemsetAsync(Y, 0, (size_t) N * m * sizeof(unsigned long long), stream));
	}
	if (w <= 0 && N <= 16 && nrows >= 4096) {
		// every row, few combinations: each row once, the combinations side by side (combine_all_rows_kernel); Y arrives zeroed
		unsigned long long *Yt = (unsigned long long *) big_alloc((size_t) m * 16 * sizeof(unsigned long long));
		HIP_CHECK(hipMemsetAsync(Yt, 0, (size_t) m * 16 * sizeof(unsigned long long), stream));
		if (colmap != nullptr)
			hipLaunchKernelGGL(combine_all_rows_kernel<true>, dim3((unsigned) std::min<int64_t>(nrows, 65536)), dim3(64), 0, stream, Ap, Aj, Ax, rows, nrows, N, salt, Yt,
			                   to_dev(M), colmap, base);
		else
			hipLaunchKernelGGL(combine_all_rows_kernel<false>, dim3((unsigned) std::min<int64_t>(nrows, 65536)), dim3(64), 0, stream, Ap, Aj, Ax, rows, nrows, N, salt, Yt,
			                   to_dev(M), colmap, base);
		hipLaunchKernelGGL(transpose_combinations_kernel, dim3((unsigned) (((int64_t) m * 16 + 255) / 256)), dim3(256), 0, stream, Yt, N, m, Y);
		HIP_CHECK(hipGetLastError());
		HIP_CHECK(hipStreamSynchronize(stream));
		big_free(Yt);
		return;
	}
	const int blocks = (int) (total < 65536 ? (total > 0 ? total : 1) : 65536);
	if (colmap != nullptr)
		hipLaunchKernelGGL(combine_rows_kernel<true>, dim3(blocks), dim3(64), 0, stream, Ap, Aj, Ax, rows, nrows, N, w, m, salt, Y,
		                   to_dev(M), colmap, base);
	else
		hipLaunchKernelGGL(combine_rows_kernel<false>, dim3(blocks), dim3(64), 0, stream, Ap, Aj, Ax, rows, nrows, N, w, m, salt, Y,
		                   to_dev(M), colmap, base);
	HIP_CHECK(hipGetLastError());
}

// do the rows only hold columns whose label is >= base (non-pivotal columns of the factor)?  *flag is raised when one does not
__global__ __launch_bounds__(256) void rows_nonpivotal_kernel(const int64_t *Ap, const int *Aj, const int *rows, int nrows, const uint32_t *lab, uint32_t base, int *flag)
{
	const int lane = threadIdx.x & 63;
	bool bad = false;
	for (int64_t t = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6); t < nrows; t += (int64_t) gridDim.x * 4) {
		const int i = rows[t];
		for (int64_t px = Ap[i] + lane; px < Ap[i + 1]; px += 64)
			bad = bad || lab[Aj[px]] < base;
	}
	if (bad)
		atomicOr(flag, 1);
}

bool rows_are_nonpivotal(const int64_t *Ap, const int *Aj, const int *rows, int nrows, const uint32_t *lab, uint32_t base, hipStream_t stream)
{
	int *d_flag = nullptr;
	HIP_CHECK(sh::malloc_or_trim((void **) &d_flag, sizeof(int)));
	HIP_CHECK(hipMemsetAsync(d_flag, 0, sizeof(int), stream));
	hipLaunchKernelGGL(rows_nonpivotal_kernel, dim3((unsigned) std::max(1, std::min(nrows / 4 + 1, 16384))), dim3(256), 0, stream, Ap, Aj, rows, nrows, lab, base, d_flag);
	HIP_CHECK(hipGetLastError());
	int flag = 1;
	HIP_CHECK(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	HIP_CHECK(hipFree(d_flag));
	return flag == 0;
}

// S[k][j] = Y[k][j] mod p: accumulators that span exactly the non-pivotal columns ARE the dense rows (nothing to reduce: the rows
// combined hold no pivotal column), no detour through CSR and the elimination kernels
__global__ __launch_bounds__(256) void dense_reduce_rows_kernel(const unsigned long long *Y, int N, int m, MontDev F, uint32_t *S, int64_t ldS)
{
	const int k = blockIdx.y;
	for (int j = blockIdx.x * 256 + threadIdx.x; j < m; j += gridDim.x * 256) {
		const unsigned long long v = Y[(int64_t) k * m + j];
		S[(int64_t) k * ldS + j] = (v == 0) ? 0u : reduce_sum(v, F);
	}
}

void launch_dense_reduce_rows(const unsigned long long *Y, int N, int m, const Mont &M, uint32_t *S, int64_t ldS, hipStream_t stream)
{
	if (N <= 0 || m <= 0)
		return;
	hipLaunchKernelGGL(dense_reduce_rows_kernel, dim3((unsigned) std::min(1024, (m + 255) / 256), (unsigned) N), dim3(256), 0, stream, Y, N, m, to_dev(M), S, ldS);
	HIP_CHECK(hipGetLastError());
}

void launch_dense_count(const unsigned long long *Y, int N, int m, uint32_t p, int *row_len, hipStream_t stream)
{
	hipLaunchKernelGGL(dense_count_kernel, dim3(N), dim3(256), 0, stream, Y, N, m, p, row_len);
	HIP_CHECK(hipGetLastError());
}

// A sparse row is reduced by one wavefront, entry after entry, and the reduction is linear in the row: a FEW very long rows
// (the handful of combinations of ALL the rows that ends a low-rank finish: 9 rows of 133,000 entries on mk13.b5, 28 ms
// on nine waves) are cut into `pieces` consecutive runs each -- the same entries under more row pointers --, every run is
// reduced as a row of its own and the dense results are added up.
__global__ __launch_bounds__(256) void split_rows_kernel(const int64_t *Sp, int N, int pieces, int64_t *out)
{
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx > N * pieces)
		return;
	if (idx == N * pieces) {
		out[idx] = Sp[N];
		return;
	}
	const int k = idx / pieces, s = idx % pieces;
	const int64_t lo = Sp[k], len = Sp[k + 1] - lo;
	out[idx] = lo + len * s / pieces;
}

__global__ __launch_bounds__(256) void sum_pieces_kernel(const uint32_t *parts, int64_t ldp, int pieces, int m, uint32_t p, uint32_t *out, int64_t ldo)
{
	const int k = blockIdx.y;
	for (int c = blockIdx.x * 256 + threadIdx.x; c < m; c += gridDim.x * 256) {
		unsigned long long acc = 0;
		for (int s = 0; s < pieces; s++)
			acc += parts[((int64_t) k * pieces + s) * ldp + c];
		out[(int64_t) k * ldo + c] = (uint32_t) (acc % p);
	}
}

void launch_split_rows(const int64_t *Sp, int N, int pieces, int64_t *out, hipStream_t stream)
{
	hipLaunchKernelGGL(split_rows_kernel, dim3((N * pieces + 1 + 255) / 256), dim3(256), 0, stream, Sp, N, pieces, out);
	HIP_CHECK(hipGetLastError());
}

void launch_sum_pieces(const uint32_t *parts, int64_t ldp, int N, int pieces, int m, uint32_t p, uint32_t *out, int64_t ldo, hipStream_t stream)
{
	hipLaunchKernelGGL(sum_pieces_kernel, dim3((m + 255) / 256, N), dim3(256), 0, stream, parts, ldp, pieces, m, p, out, ldo);
	HIP_CHECK(hipGetLastError());
}

void launch_dense_pack(const unsigned long long *Y, int N, int m, uint32_t p, const int64_t *Sp, int *Sj, int *Sx,
                       hipStream_t stream, const int *unmap)
{
	if (unmap != nullptr)
		hipLaunchKernelGGL(dense_pack_kernel<true>, dim3(N), dim3(256), 0, stream, Y, N, m, p, Sp, Sj, Sx, unmap);
	else
		hipLaunchKernelGGL(dense_pack_kernel<false>, dim3(N), dim3(256), 0, stream, Y, N, m, p, Sp, Sj, Sx, unmap);
	HIP_CHECK(hipGetLastError());
}

}  // namespace sh

// --------------------------------------------------------------------------
// Dense PLUQ mod p (replaces FFPACK::pPLUQ behind spasm_ffpack_LU, spasm_ffpack.cpp:57-86).
// Contract consumed by update_fact_after_LU (spasm_echelonize.c:228-297) and
// tests/dense_lu_ffpack.c: on return, with r = rank,
//   row i of the packed matrix corresponds to original row P[i], column j to original column Qinv[j];
//   M[i][j], j < min(i+1, r)  : L (lower trapezoid, pivots on its diagonal);
//   M[i][j], i < r, j > i     : U (unit diagonal implied);      A == L * U in original coordinates.
// Right-looking elimination, rows and columns physically swapped; a column with no non-zero below the
// current step is swapped to the end.  For p <= 65279: 64 pivots per round whenever the 64 x 64 block on
// the diagonal is non-singular (blocked steps below, trailing update on the matrix cores), one pivot per step
// otherwise.  This is the L-recording variant of the dense tail; the fast path is device_rref above.
// --------------------------------------------------------------------------
namespace sh {

// first row >= t of column t holding a non-zero, or -1; single workgroup
__global__ __launch_bounds__(1024) void lu_find_pivot(const uint32_t *A, int64_t ld, int n, int t, int *out)
{
	__shared__ int best;
	if (threadIdx.x == 0)
		best = 0x7FFFFFFF;
	__syncthreads();
	int mine = 0x7FFFFFFF;
	for (int i = t + threadIdx.x; i < n; i += 1024)
		if (A[(int64_t) i * ld + t] != 0) {
			mine = i;
			break;
		}
	if (mine != 0x7FFFFFFF)
		atomicMin(&best, mine);
	__syncthreads();
	if (threadIdx.x == 0)
		*out = (best == 0x7FFFFFFF) ? -1 : best;
}

__global__ void lu_swap_rows(uint32_t *A, int64_t ld, int m, int a, int b, int *P)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < m) {
		const uint32_t x = A[(int64_t) a * ld + j];
		A[(int64_t) a * ld + j] = A[(int64_t) b * ld + j];
		A[(int64_t) b * ld + j] = x;
	}
	if (j == 0) {
		const int x = P[a];
		P[a] = P[b];
		P[b] = x;
	}
}

__global__ void lu_swap_cols(uint32_t *A, int64_t ld, int n, int a, int b, int *Q)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		const uint32_t x = A[(int64_t) i * ld + a];
		A[(int64_t) i * ld + a] = A[(int64_t) i * ld + b];
		A[(int64_t) i * ld + b] = x;
	}
	if (i == 0) {
		const int x = Q[a];
		Q[a] = Q[b];
		Q[b] = x;
	}
}

// row t: U[t][j] = A[t][j] / pivot for j > t (in place)
__global__ void lu_scale_row(uint32_t *A, int64_t ld, int m, int t, MontDev F)
{
	const uint32_t inv = invmod(A[(int64_t) t * ld + t], F);
	const int j = t + 1 + blockIdx.x * blockDim.x + threadIdx.x;
	if (j < m)
		A[(int64_t) t * ld + j] = mulmod(A[(int64_t) t * ld + j], inv, F);
}

// A[i][j] -= A[i][t] * A[t][j]   for i > t, j > t
__global__ __launch_bounds__(256) void lu_rank1_update(uint32_t *A, int64_t ld, int n, int m, int t, MontDev F)
{
	const int j = t + 1 + blockIdx.x * 256 + threadIdx.x;
	const int i0 = t + 1 + blockIdx.y * 16;
	if (j >= m)
		return;
	const uint32_t u = A[(int64_t) t * ld + j];
	if (u == 0)
		return;
	for (int i = i0; i < i0 + 16 && i < n; i++) {
		const uint32_t l = A[(int64_t) i * ld + t];
		if (l != 0) {
			uint32_t *x = A + (int64_t) i * ld + j;
			*x = submod(*x, mulmod(l, u, F), F);
		}
	}
}

// ---- blocked steps (p <= 65279): 64 pivots per round, the trailing update on the matrix cores ----
// The 64 x 64 block on the diagonal is factored by one workgroup in LDS with row pivoting INSIDE the block (the usual
// case: it is non-singular); then L21 = A21 U11^-1 row by row, the block's row order applied to the other columns,
// U12 = L11^-1 A12 column by column (both emit the signed base-256 digit planes of their result), and
// A22 -= L21 U12 through rref_update_mfma_multi.  A block with a column without pivot writes nothing: the caller takes
// one step of the one-pivot-at-a-time code (any row may hold the pivot, a dead column moves to the end) and tries again.

// v * w mod p for p < 2^16 (plain residues)
__device__ __forceinline__ uint32_t lu_mul16(uint32_t v, uint32_t w, uint32_t p, uint32_t bm)
{
	const uint32_t t = __umul24(v, w);
	const uint32_t q = __umulhi(t, bm);
	uint32_t rem = t - __umul24(q, p);
	rem = (rem >= p) ? rem - p : rem;
	rem = (rem >= p) ? rem - p : rem;
	return rem;
}

// info[0] = 1: factored in place, perm[r] = row of the block (0 .. 63) that went to position r; P updated.  0: untouched.
__global__ __launch_bounds__(256) void lu_block_kernel(uint32_t *A, int64_t ld, int t, int *P, int *perm, int *info, MontDev F)
{
	__shared__ uint32_t B[NB][NB + 1];
	__shared__ int s_perm[NB], s_P[NB];
	__shared__ int s_pr;
	__shared__ uint32_t s_inv;
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const uint32_t p = F.p, bm = (uint32_t) (0x100000000ull / F.p);
	for (int e = tid; e < NB * NB; e += 256)
		B[e >> 6][e & 63] = A[(int64_t) (t + (e >> 6)) * ld + t + (e & 63)];
	if (tid < NB) {
		s_perm[tid] = tid;
		s_P[tid] = P[t + tid];
	}
	__syncthreads();
	for (int k = 0; k < NB; k++) {
		if (w == 0) {
			const unsigned long long nz = __ballot(lane >= k && B[lane][k] != 0);
			if (lane == 0) {
				s_pr = (nz != 0) ? (int) __builtin_ctzll(nz) : -1;
				if (nz != 0)
					s_inv = invmod(B[__builtin_ctzll(nz)][k], F);
			}
		}
		__syncthreads();
		const int pr = s_pr;
		if (pr < 0) {
			if (tid == 0)
				info[0] = 0;
			return;
		}
		if (pr != k && tid < NB) {          // swap rows k and pr of the block
			const uint32_t x = B[k][tid];
			B[k][tid] = B[pr][tid];
			B[pr][tid] = x;
			if (tid == 0) {
				const int a = s_perm[k], b = s_P[k];
				s_perm[k] = s_perm[pr];
				s_perm[pr] = a;
				s_P[k] = s_P[pr];
				s_P[pr] = b;
			}
		}
		__syncthreads();
		if (tid > k && tid < NB)             // the row of U: divided by the pivot (the pivot itself stays, on the diagonal of L)
			B[k][tid] = lu_mul16(B[k][tid], s_inv, p, bm);
		__syncthreads();
		// rows below, columns to the right
		for (int e = tid; e < (NB - 1 - k) * (NB - 1 - k); e += 256) {
			const int i = k + 1 + e / (NB - 1 - k), j = k + 1 + e % (NB - 1 - k);
			const uint32_t l = B[i][k];
			if (l != 0) {
				const uint32_t sub = lu_mul16(l, B[k][j], p, bm);
				B[i][j] = (B[i][j] >= sub) ? B[i][j] - sub : B[i][j] + p - sub;
			}
		}
		__syncthreads();
	}
	for (int e = tid; e < NB * NB; e += 256)
		A[(int64_t) (t + (e >> 6)) * ld + t + (e & 63)] = B[e >> 6][e & 63];
	if (tid < NB) {
		perm[tid] = s_perm[tid];
		P[t + tid] = s_P[tid];
	}
	if (tid == 0)
		info[0] = 1;
}

// the row order of the block for the columns outside it: row t + r <- old row t + perm[r]
__global__ __launch_bounds__(256) void lu_permute_rows(uint32_t *A, int64_t ld, int m, int t, const int *perm)
{
	__shared__ int sp[NB];
	if (threadIdx.x < NB)
		sp[threadIdx.x] = perm[threadIdx.x];
	__syncthreads();
	const int j = blockIdx.x * 256 + threadIdx.x;
	if (j >= m || (j >= t && j < t + NB))
		return;
	uint32_t v[NB];
#pragma unroll
	for (int r = 0; r < NB; r++)
		v[r] = A[(int64_t) (t + sp[r]) * ld + j];
#pragma unroll
	for (int r = 0; r < NB; r++)
		A[(int64_t) (t + r) * ld + j] = v[r];
}

// L21 = A21 U11^-1 (U11 unit upper triangular): one thread per row below the block, its 64 entries in LDS (entry-major:
// no bank conflicts, and plain loops -- the fully unrolled register version took minutes to compile); also the digit
// planes of -L21 for the trailing update (row i - (t + 64) of Mh / Ml)
__global__ __launch_bounds__(256) void lu_L21_kernel(uint32_t *A, int64_t ld, int n, int t, signed char *Mh, signed char *Ml, MontDev F)
{
	__shared__ uint32_t U[NB][NB];
	__shared__ uint32_t xs[NB][256];
	const uint32_t p = F.p, bm = (uint32_t) (0x100000000ull / F.p);
	const int tid = threadIdx.x;
	for (int e = tid; e < NB * NB; e += 256)
		U[e >> 6][e & 63] = A[(int64_t) (t + (e >> 6)) * ld + t + (e & 63)];
	const int i = t + NB + blockIdx.x * 256 + tid;
	uint32_t *row = A + (int64_t) (i < n ? i : t) * ld + t;
	if (i < n)
		for (int q = 0; q < NB / 4; q++) {
			const uint4 v = *reinterpret_cast<const uint4 *>(row + 4 * q);          // (t and ld are multiples of 4 on this path)
			xs[4 * q][tid] = v.x;
			xs[4 * q + 1][tid] = v.y;
			xs[4 * q + 2][tid] = v.z;
			xs[4 * q + 3][tid] = v.w;
		}
	__syncthreads();
	if (i >= n)
		return;
	for (int s = 0; s < NB; s++) {
		const uint32_t xv = xs[s][tid];
		if (xv == 0)
			continue;
		const uint32_t neg = p - xv;
#pragma unroll 4
		for (int j = s + 1; j < NB; j++) {
			const uint32_t v = xs[j][tid] + lu_mul16(neg, U[s][j], p, bm);
			xs[j][tid] = (v >= p) ? v - p : v;
		}
	}
	int4 *dh = reinterpret_cast<int4 *>(Mh + (int64_t) (i - t - NB) * 64), *dl = reinterpret_cast<int4 *>(Ml + (int64_t) (i - t - NB) * 64);
	for (int q4 = 0; q4 < 4; q4++) {
		unsigned int wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const int c = 16 * q4 + 4 * q;
			const uint4 v = make_uint4(xs[c][tid], xs[c + 1][tid], xs[c + 2][tid], xs[c + 3][tid]);
			*reinterpret_cast<uint4 *>(row + c) = v;
			const uint32_t e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int b = 0; b < 4; b++) {
				int hi, lo;
				split_digits((e4[b] == 0) ? 0u : p - e4[b], F, hi, lo);
				wh[q] |= (unsigned int) (hi & 255) << (8 * b);
				wl[q] |= (unsigned int) (lo & 255) << (8 * b);
			}
		}
		dh[q4] = make_int4((int) wh[0], (int) wh[1], (int) wh[2], (int) wh[3]);
		dl[q4] = make_int4((int) wl[0], (int) wl[1], (int) wl[2], (int) wl[3]);
	}
}

// U12 = L11^-1 A12 (L11 lower triangular with the pivots on its diagonal): one thread per column right of the block, its
// 64 entries in LDS; also its digit planes (column j - (t + 64) of Bh / Bl)
__global__ __launch_bounds__(256) void lu_U12_kernel(uint32_t *A, int64_t ld, int m, int t, signed char *Bh, signed char *Bl, MontDev F)
{
	__shared__ uint32_t L[NB][NB + 1];
	__shared__ uint32_t dinv[NB];
	__shared__ uint32_t us[NB][256];
	const uint32_t p = F.p, bm = (uint32_t) (0x100000000ull / F.p);
	const int tid = threadIdx.x;
	for (int e = tid; e < NB * NB; e += 256)
		L[e >> 6][e & 63] = A[(int64_t) (t + (e >> 6)) * ld + t + (e & 63)];
	__syncthreads();
	if (tid < NB)
		dinv[tid] = invmod(L[tid][tid], F);
	__syncthreads();
	const int j = t + NB + blockIdx.x * 256 + tid;
	if (j >= m)
		return;
	for (int s = 0; s < NB; s++)
		us[s][tid] = A[(int64_t) (t + s) * ld + j];
	for (int q = 0; q < NB; q++) {
		const uint32_t uq = lu_mul16(us[q][tid], dinv[q], p, bm);
		us[q][tid] = uq;
		if (uq == 0)
			continue;
		const uint32_t neg = p - uq;
#pragma unroll 4
		for (int s = q + 1; s < NB; s++) {
			const uint32_t v = us[s][tid] + lu_mul16(L[s][q], neg, p, bm);
			us[s][tid] = (v >= p) ? v - p : v;
		}
	}
	int4 *dh = reinterpret_cast<int4 *>(Bh + (int64_t) (j - t - NB) * 64), *dl = reinterpret_cast<int4 *>(Bl + (int64_t) (j - t - NB) * 64);
	for (int q4 = 0; q4 < 4; q4++) {
		unsigned int wh[4] = {0, 0, 0, 0}, wl[4] = {0, 0, 0, 0};
#pragma unroll
		for (int q = 0; q < 4; q++)
#pragma unroll
			for (int b = 0; b < 4; b++) {
				const int srow = 16 * q4 + 4 * q + b;
				const uint32_t v = us[srow][tid];
				A[(int64_t) (t + srow) * ld + j] = v;
				int hi, lo;
				split_digits(v, F, hi, lo);
				wh[q] |= (unsigned int) (hi & 255) << (8 * b);
				wl[q] |= (unsigned int) (lo & 255) << (8 * b);
			}
		dh[q4] = make_int4((int) wh[0], (int) wh[1], (int) wh[2], (int) wh[3]);
		dl[q4] = make_int4((int) wl[0], (int) wl[1], (int) wl[2], (int) wl[3]);
	}
}

int device_lu(int64_t prime, int n, int m, uint32_t *dA, int64_t ld, int *dP, int *dQ, hipStream_t stream)
{
	const Mont M = mont_setup(prime);
	const MontDev F = to_dev(M);
	int *d_piv = nullptr;
	HIP_CHECK(sh::malloc_or_trim((void **) &d_piv, 64));
	int t = 0;
	int mlast = m;                   // columns [mlast, m) are known to be zero below the diagonal block
	const int rmax = (n < m) ? n : m;
	// blocked steps: the matrix cores need two signed base-256 digits (p <= 65279), the row kernels 16-byte rows
	bool blocked = prime <= 65279 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(dA) % 16) == 0;
	if (const char *e = sh::env_get("SPASM_HIP_LU_BLOCKED"))
		blocked = blocked && std::atoi(e) != 0;
	signed char *lu_M8 = nullptr, *lu_B8 = nullptr;
	int *lu_perm = nullptr;
	if (blocked) {
		HIP_CHECK(sh::malloc_or_trim((void **) &lu_M8, (size_t) 2 * (size_t) n * 64));
		HIP_CHECK(sh::malloc_or_trim((void **) &lu_B8, (size_t) 2 * (size_t) m * 64));
		HIP_CHECK(sh::malloc_or_trim((void **) &lu_perm, NB * sizeof(int)));
	}
	int block_cooldown = 0;          // after a block that could not be factored: single steps before the next attempt
	while (t < rmax && t < mlast) {
		if (blocked && block_cooldown == 0 && t % 4 == 0 && t + NB <= n && t + NB <= mlast) {
			int ok = 0;
			hipLaunchKernelGGL(lu_block_kernel, dim3(1), dim3(256), 0, stream, dA, ld, t, dP, lu_perm, d_piv + 4, F);
			HIP_CHECK(hipMemcpyAsync(&ok, d_piv + 4, sizeof(int), hipMemcpyDeviceToHost, stream));
			HIP_CHECK(hipStreamSynchronize(stream));
			if (ok) {
				const int nb = n - t - NB, mb = m - t - NB;          // rows below, columns right of the block
				hipLaunchKernelGGL(lu_permute_rows, dim3((m + 255) / 256), dim3(256), 0, stream, dA, ld, m, t, lu_perm);
				if (nb > 0)
					hipLaunchKernelGGL(lu_L21_kernel, dim3((nb + 255) / 256), dim3(256), 0, stream, dA, ld, n, t, lu_M8, lu_M8 + (size_t) n * 64, F);
				if (mb > 0)
					hipLaunchKernelGGL(lu_U12_kernel, dim3((mb + 255) / 256), dim3(256), 0, stream, dA, ld, m, t, lu_B8, lu_B8 + (size_t) m * 64, F);
				if (nb > 0 && mb > 0) {
					UpdSets one{};
					one.nsets = 1;
					one.Mh[0] = lu_M8;
					one.Ml[0] = lu_M8 + (size_t) n * 64;
					one.Bh[0] = lu_B8;
					one.Bl[0] = lu_B8 + (size_t) m * 64;
					dim3 grid((mb + 63) / 64, (nb + 63) / 64);
					hipLaunchKernelGGL(rref_update_mfma_multi, grid, dim3(256), 0, stream, dA + (int64_t) (t + NB) * ld, ld, nb, t + NB, mb, one, F);
				}
				HIP_CHECK(hipGetLastError());
				t += NB;
				continue;
			}
			block_cooldown = 8;
		}
		if (block_cooldown > 0)
			block_cooldown -= 1;
		hipLaunchKernelGGL(lu_find_pivot, dim3(1), dim3(1024), 0, stream, dA, ld, n, t, d_piv);
		int row = -1;
		HIP_CHECK(hipMemcpyAsync(&row, d_piv, sizeof(int), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		if (row < 0) {               // dead column: move it out of the way and try the one that takes its place
			mlast -= 1;
			if (mlast > t)
				hipLaunchKernelGGL(lu_swap_cols, dim3((n + 255) / 256), dim3(256), 0, stream, dA, ld, n, t, mlast, dQ);
			continue;
		}
		if (row != t)
			hipLaunchKernelGGL(lu_swap_rows, dim3((m + 255) / 256), dim3(256), 0, stream, dA, ld, m, t, row, dP);
		if (t + 1 < m)
			hipLaunchKernelGGL(lu_scale_row, dim3((m - t - 1 + 255) / 256), dim3(256), 0, stream, dA, ld, m, t, F);
		if (t + 1 < m && t + 1 < n) {
			dim3 grid((m - t - 1 + 255) / 256, (n - t - 1 + 15) / 16);
			hipLaunchKernelGGL(lu_rank1_update, grid, dim3(256), 0, stream, dA, ld, n, m, t, F);
		}
		HIP_CHECK(hipGetLastError());
		t += 1;
	}
	HIP_CHECK(hipStreamSynchronize(stream));
	sh::big_free(d_piv);
	sh::big_free(lu_M8);
	sh::big_free(lu_B8);
	sh::big_free(lu_perm);
	return t;
}

}  // namespace sh
