// Sparse back-substituted factor image: S = A_n - A_p R with R = U_pp^-1 U_pn kept SPARSE.
//
// backsolve.hip stores R dense (r x Sm 16-bit entries): right when the Schur complement is going to be dense anyway
// (mk13.b5: 72 %), wasteful when it stays sparse -- mk14.b4: 273,000 x 42,000 entries = 23 GB built and gathered for a
// Schur complement that is 3.7 % dense, and no image at all beyond 131,072 non-pivotal columns.  The reference never
// forms R: it solves x = a U_pp^-1 row by row (spasm_schur.c:86-171 -> spasm_triangular.c:110-146 -> spasm_reach.c:22-135),
// and what that costs on a GPU is the random read-modify-write traffic of the row-group kernel.
// Here R is formed once per factor, like in backsolve.hip, but as sparse rows:
//
//   * the non-pivotal columns are cut into SEGMENTS of SP_SEG = 8,192 columns;
//   * a row of R is a set of FRAGMENTS, one per segment it has entries in: 4-byte entries (column inside the segment |
//     signed 16-bit value << 16, sorted by column) in a bump-allocated pool; frag[c * nseg + g] = where and how long;
//   * sp_build_kernel, one launch per elimination level from the last to the first (what spasm_reach's depth-first search
//     orders for one row, the level schedule orders for all of them): one WAVE per (row c, segment g) adds up
//     U_n[c] - sum_t u_ct R[t] in 16 KB of LDS (16-bit accumulators, one read-modify-write per entry of a fragment: the
//     entries of a fragment have distinct columns and a wave's LDS accesses are served in order, so no atomics), counts,
//     reserves its room and writes the fragment.  Work = sum over the pivotal entries of U of the fill of the rows they
//     point at -- it scales with nnz(R), not with r x Sm;
//   * sp_apply_kernel: one wave per (reduced row k, segment g), the same accumulation over the pivotal entries of the row
//     of A, fragments of S into a pool; a scan of the row lengths and sp_gather_kernel put the rows in their final place
//     (W->d_Sp / d_Sj / d_Sx, columns sorted).
//
// Arithmetic: signed 16-bit representatives with the fp32 reduction of sgn_dev.h (p <= 44,927; 42013 -- the reference's
// default -- qualifies); every multiply-add is reduced at once (|x| <= B, |c v| <= (p/2) B: the sum fits 31 bits).
// Exact mod p, so S is the matrix the other paths compute, bit for bit (tests/test_gpu_sparse_image.py).
#include <algorithm>
#include <cinttypes>
#include <vector>

#include "device_types.h"
#include "field_dev.h"
#include "sgn_dev.h"

namespace sh {

namespace {

constexpr int SEGW = SP_SEG / 2;                       // 32-bit words of a segment's accumulators
constexpr uint64_t LEN_MASK = (1ull << SP_LEN_BITS) - 1;
constexpr uint64_t OFF_MASK = (1ull << SP_OFF_BITS) - 1;
constexpr int SHARD_STRIDE = 16;                        // 64-bit words per shard: one 128-byte line

int env_sp(const char *name, int dflt)
{
	const char *e = std::getenv(name);
	return (e == nullptr || *e == 0) ? dflt : std::atoi(e);
}

template <typename T> T *dalloc(int64_t count)
{
	return static_cast<T *>(sh::big_alloc((size_t) (count > 0 ? count : 1) * sizeof(T)));
}

template <typename T> void upload(T *dst, const std::vector<T> &src, hipStream_t s)
{
	if (!src.empty())
		HIP_CHECK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

__device__ __forceinline__ const uint32_t *frag_ptr(const SpPools &P, uint64_t f)
{
	return P.base[(f >> (SP_LEN_BITS + SP_OFF_BITS)) & 15u] + ((f >> SP_LEN_BITS) & OFF_MASK);
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src)
{
	const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, src);
	const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), src);
	return ((uint64_t) hi << 32) | lo;
}

__device__ __forceinline__ void sp_zero(uint32_t *accw, int lane)
{
	uint4 *a4 = reinterpret_cast<uint4 *>(accw);
#pragma unroll
	for (int t = 0; t < SEGW / 4 / 64; t++)
		a4[t * 64 + lane] = uint4{0u, 0u, 0u, 0u};
}

// acc[column] += coef * value for one entry (column | value << 16); the sum is reduced at once
__device__ __forceinline__ void sp_entry(short *acc, uint32_t e, int coef, const SgnDev &G)
{
	short *a = acc + (e & 0xFFFFu);
	int t = (int) *a;
	asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(t) : "v"(e), "v"(coef));
	*a = (short) sgn_reduce(t, G);
}

// four entries of ONE fragment (distinct columns): the four reads are in flight together
__device__ __forceinline__ void sp_entry4(short *acc, uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, int coef, const SgnDev &G)
{
	short *a0 = acc + (e0 & 0xFFFFu), *a1 = acc + (e1 & 0xFFFFu), *a2 = acc + (e2 & 0xFFFFu), *a3 = acc + (e3 & 0xFFFFu);
	int t0 = (int) *a0, t1 = (int) *a1, t2 = (int) *a2, t3 = (int) *a3;
	asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(t0) : "v"(e0), "v"(coef));
	asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(t1) : "v"(e1), "v"(coef));
	asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(t2) : "v"(e2), "v"(coef));
	asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(t3) : "v"(e3), "v"(coef));
	*a0 = (short) sgn_reduce(t0, G);
	*a1 = (short) sgn_reduce(t1, G);
	*a2 = (short) sgn_reduce(t2, G);
	*a3 = (short) sgn_reduce(t3, G);
}

// acc += coef * fragment, for every lane whose fragment word f is not empty (wave-uniform loop over those lanes).  The
// first 64 entries of the next fragment are in flight while the current one is added.
__device__ __forceinline__ void sp_accumulate(short *acc, uint64_t f, int coef, const SpPools &pools, int lane, const SgnDev &G,
                                              unsigned long long &ops)
{
	uint64_t live = __ballot((f & LEN_MASK) != 0);
	if (live == 0)
		return;
	int s = __builtin_ctzll(live);
	live &= live - 1;
	uint64_t fc = readlane64(f, s);
	int cc = __builtin_amdgcn_readlane(coef, s);
	const uint32_t *src = frag_ptr(pools, fc);
	int len = (int) (fc & LEN_MASK);
	uint32_t e0 = (lane < len) ? src[lane] : 0u;
	for (;;) {
		// the next fragment's head
		const bool more = live != 0;
		uint64_t fn = 0;
		int cn = 0, lenn = 0;
		const uint32_t *srcn = src;
		uint32_t en = 0;
		if (more) {
			s = __builtin_ctzll(live);
			live &= live - 1;
			fn = readlane64(f, s);
			cn = __builtin_amdgcn_readlane(coef, s);
			srcn = frag_ptr(pools, fn);
			lenn = (int) (fn & LEN_MASK);
			en = (lane < lenn) ? srcn[lane] : 0u;
		}
		ops += (unsigned long long) len;
		if (lane < len)
			sp_entry(acc, e0, cc, G);
		int i = lane + 64;
		for (; i + 192 < len; i += 256) {
			const uint32_t a0 = src[i], a1 = src[i + 64], a2 = src[i + 128], a3 = src[i + 192];
			sp_entry4(acc, a0, a1, a2, a3, cc, G);
		}
		for (; i < len; i += 64)
			sp_entry(acc, src[i], cc, G);
		if (!more)
			break;
		src = srcn;
		len = lenn;
		cc = cn;
		e0 = en;
	}
}

// non-zero accumulators of the segment (the whole wave gets the sum)
__device__ __forceinline__ int sp_count(const uint32_t *accw, int lane)
{
	const uint4 *a4 = reinterpret_cast<const uint4 *>(accw);
	int cnt = 0;
#pragma unroll 4
	for (int t = 0; t < SEGW / 4 / 64; t++) {
		const uint4 w = a4[t * 64 + lane];
		cnt += ((w.x & 0xFFFFu) != 0) + ((w.x >> 16) != 0) + ((w.y & 0xFFFFu) != 0) + ((w.y >> 16) != 0);
		cnt += ((w.z & 0xFFFFu) != 0) + ((w.z >> 16) != 0) + ((w.w & 0xFFFFu) != 0) + ((w.w >> 16) != 0);
	}
	for (int sft = 32; sft >= 1; sft >>= 1)
		cnt += __shfl_xor(cnt, sft);
	return cnt;
}

// the non-zero accumulators as (column | value << 16) entries, sorted by column; CANON: values brought into [-p/2, p/2]
template <bool CANON> __device__ __forceinline__ void sp_emit(const uint32_t *accw, uint32_t *out, int lane, const SgnDev &G)
{
	uint32_t wpos = 0;
	for (int t0 = 0; t0 < SEGW; t0 += 64) {
		const uint32_t w = accw[t0 + lane];
		if (__ballot(w != 0) == 0)
			continue;
		int v0, v1;
		sgn_unpack(w, v0, v1);
		if (CANON) {
			v0 = sgn_canonical(v0, G);
			v1 = sgn_canonical(v1, G);
		}
		const uint64_t m0 = __ballot(v0 != 0), m1 = __ballot(v1 != 0);
		uint32_t dst = wpos;
		dst = __builtin_amdgcn_mbcnt_hi((uint32_t) (m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m0, dst));
		dst = __builtin_amdgcn_mbcnt_hi((uint32_t) (m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m1, dst));
		const uint32_t c0 = 2u * (uint32_t) (t0 + lane);
		if (v0 != 0) {
			out[dst] = c0 | ((uint32_t) v0 << 16);
			dst += 1;
		}
		if (v1 != 0)
			out[dst] = (c0 + 1u) | ((uint32_t) v1 << 16);
		wpos += (uint32_t) (__popcll(m0) + __popcll(m1));
	}
}

__device__ __forceinline__ uint32_t sp_hash(uint32_t c, uint32_t g)
{
	uint32_t h = c * 0x9E3779B1u + g * 0x85EBCA77u;
	h ^= h >> 15;
	h *= 0x2C1B3C6Du;
	return h >> 24;          // SP_SHARDS = 256
}

// ---------------------------------------------------------------------------------------------------
// build of R, one level per launch
// ---------------------------------------------------------------------------------------------------
struct SpBuildArgs {
	const uint64_t *dep_rp;
	const uint2 *dep;
	const uint64_t *np_rp;
	const uint2 *np;
	uint64_t *frag;
	int nseg;
	int row_lo, row_hi;           // compact rows of this level
	int level, chunk;
	SpPools pools;
	uint32_t *chunk_base;         // the chunk fragments are written to
	unsigned long long *shard;
	int *ovf_level;               // largest level in which a reservation failed (-1: none)
	SgnDev G;
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
	}
}

__global__ __launch_bounds__(64) void sp_build_kernel(SpBuildArgs b)
{
	__shared__ __attribute__((aligned(16))) uint32_t accw[SEGW];
	short *acc = reinterpret_cast<short *>(accw);
	const int lane = threadIdx.x;
	const int task = blockIdx.x;
	const int c = b.row_lo + task / b.nseg;
	const int g = task - (c - b.row_lo) * b.nseg;
	const uint32_t col0 = (uint32_t) g * SP_SEG;
	const SgnDev G = b.G;
	const uint64_t d0 = b.dep_rp[c], d1 = b.dep_rp[c + 1], n0 = b.np_rp[c], n1 = b.np_rp[c + 1];
	uint64_t *fout = b.frag + (uint64_t) c * b.nseg + g;
	unsigned long long ops = 0;
	bool zeroed = false;
	// the row's own non-pivotal entries that fall into this segment
	for (uint64_t e = n0; e < n1; e += 64) {
		uint32_t idx = 0xFFFFFFFFu;
		int val = 0;
		if (e + lane < n1) {
			const uint2 en = b.np[e + lane];
			idx = en.x - col0;
			val = (int) en.y;
		}
		const bool in = idx < (uint32_t) SP_SEG;
		if (__ballot(in) == 0)
			continue;
		if (!zeroed) {
			sp_zero(accw, lane);
			zeroed = true;
		}
		if (in)
			acc[idx] = (short) val;
	}
	// minus the rows of R its pivotal entries point at (coefficients are stored negated)
	for (uint64_t e = d0; e < d1; e += 64) {
		uint64_t f = 0;
		int coef = 0;
		if (e + lane < d1) {
			const uint2 de = b.dep[e + lane];
			f = b.frag[(uint64_t) de.x * b.nseg + g];
			coef = (int) de.y;
		}
		if (__ballot((f & LEN_MASK) != 0) == 0)
			continue;
		if (!zeroed) {
			sp_zero(accw, lane);
			zeroed = true;
		}
		sp_accumulate(acc, f, coef, b.pools, lane, G, ops);
	}
	if (!zeroed) {
		if (lane == 0)
			*fout = 0;
		return;
	}
	const int cnt = sp_count(accw, lane);
	const uint32_t sh = sp_hash((uint32_t) c, (uint32_t) g);
	unsigned long long *S = b.shard + (size_t) sh * SHARD_STRIDE;
	if (cnt == 0) {
		if (lane == 0) {
			*fout = 0;
			atomicAdd(&S[2], ops);
		}
		return;
	}
	unsigned long long off = 0;
	if (lane == 0)
		off = atomicAdd(&S[0], (unsigned long long) cnt);
	off = readlane64(off, 0);
	const unsigned long long limit = S[1];
	if (off + (unsigned long long) cnt > limit) {
		if (lane == 0) {
			atomicMax(b.ovf_level, b.level);
			*fout = 0;
		}
		return;
	}
	sp_emit<false>(accw, b.chunk_base + off, lane, G);
	if (lane == 0) {
		*fout = ((uint64_t) b.chunk << (SP_LEN_BITS + SP_OFF_BITS)) | ((uint64_t) off << SP_LEN_BITS) | (uint64_t) cnt;
		atomicAdd(&S[2], ops);
		atomicAdd(&S[3], (unsigned long long) cnt);
		atomicAdd(&S[4], 1ull);
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
__global__ __launch_bounds__(256) void sp_census_kernel(const uint64_t *frag, int64_t n, SpPools pools, unsigned long long *out)
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
		frags += (lane == 0);
		for (int i = lane; i < len; i += 64) {
			const uint32_t c = src[i] & 0xFFFFu;
			const uint32_t prev = (i > 0) ? (src[i - 1] & 0xFFFFu) : 0xFFFFFFFFu;
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
struct SpApplyArgs {
	SchurArgs a;
	const int *col;               // column -> compact id of its pivot row, or r + index among the non-pivotal columns
	int r, nseg;
	const uint64_t *frag;
	SpPools pools;
	SgnDev G;
	uint32_t *fpool;              // fragments of S (column inside the segment | value << 16, values in [-p/2, p/2])
	int64_t fcap;
	uint64_t *T;                  // nrows x nseg: offset << SP_LEN_BITS | length of the fragment of (row, segment)
	unsigned long long *block_sum;// sum of the lengths of every block of 1024 rows (zeroed before the launch)
	int arena;                    // entries a wave reserves from the pool at a time (0: every fragment on its own)
	uint32_t *dense_out;          // dense rows instead (values in [0, p)), leading dimension ldS
	int64_t ldS;
};

__global__ __launch_bounds__(64) void sp_apply_kernel(SpApplyArgs d)
{
	__shared__ __attribute__((aligned(16))) uint32_t accw[SEGW];
	short *acc = reinterpret_cast<short *>(accw);
	const SchurArgs &a = d.a;
	const int lane = threadIdx.x;
	const SgnDev G = d.G;
	const MontDev F = a.F;
	const int nrows = a.nrows;
	const long long ntasks = (long long) nrows * d.nseg;
	unsigned long long st_input = 0, ops = 0;
	int st_done = 0, st_piv = 0;
	long long ar_cur = 0, ar_end = 0;            // the wave's arena in the fragment pool
	bool pool_full = false;
	// segment-major: the waves of the chip work on the same segment of R at the same time (its fragments -- 1 / nseg of the
	// image -- are what the caches then hold)
	for (long long task = blockIdx.x; task < ntasks; task += gridDim.x) {
		const int g = (int) (task / nrows);
		const int k = (int) (task - (long long) g * nrows);
		const uint32_t col0 = (uint32_t) g * SP_SEG;
		const int i = a.rows[k];
		const int64_t lo = a.Ap[i], hi = a.Ap[i + 1];
		if (g == 0)
			st_input += (unsigned long long) (hi - lo);
		bool zeroed = false;
		for (int64_t base = lo; base < hi; base += 64) {
			uint64_t f = 0;
			int coef = 0, bal = 0;
			uint32_t idx = 0xFFFFFFFFu;
			if (base + lane < hi) {
				const uint32_t cid = (uint32_t) d.col[a.Aj[base + lane]];
				bal = sgn_from_residue(reduce_sum(from_balanced(a.Ax[base + lane], F), F), G);
				if (cid >= (uint32_t) d.r) {
					idx = cid - (uint32_t) d.r - col0;
				} else if (bal != 0) {
					f = d.frag[(uint64_t) cid * d.nseg + g];
					coef = -bal;
					st_piv += (g == 0) ? 1 : 0;
				}
			}
			const bool in = idx < (uint32_t) SP_SEG;
			if ((__ballot(in) | __ballot((f & LEN_MASK) != 0)) == 0)
				continue;
			if (!zeroed) {
				sp_zero(accw, lane);
				zeroed = true;
			}
			if (in)
				acc[idx] = (short) sgn_canonical((int) acc[idx] + bal, G);
			sp_accumulate(acc, f, coef, d.pools, lane, G, ops);
		}
		if (d.dense_out != nullptr) {
			uint32_t *out = d.dense_out + (int64_t) k * d.ldS + col0;
			const int ncols = min(SP_SEG, a.Sm - (int) col0);
			for (int t = lane; t < ncols; t += 64) {
				const int v = zeroed ? sgn_canonical((int) acc[t], G) : 0;
				out[t] = (uint32_t) (v < 0 ? v + G.p : v);
			}
			if (g == 0) {
				if (lane == 0)
					a.row_len[k] = a.Sm;
				st_done += 1;
			}
			continue;
		}
		st_done += (g == 0) ? 1 : 0;
		const int cnt = zeroed ? sp_count(accw, lane) : 0;
		if (cnt == 0) {
			if (lane == 0)
				d.T[(uint64_t) k * d.nseg + g] = 0;
			continue;
		}
		if (ar_cur + cnt > ar_end) {
			const long long want = (d.arena > cnt) ? d.arena : cnt;
			unsigned long long got = 0;
			if (lane == 0)
				got = atomicAdd(&a.ctr64[C64_POOL], (unsigned long long) want);
			got = readlane64(got, 0);
			ar_cur = (long long) got;
			ar_end = ar_cur + want;
			if (ar_end > d.fcap) {
				// the pool is exhausted: a smaller reservation may still fit for this fragment, the call fails anyway
				pool_full = true;
				ar_end = ar_cur;
			}
		}
		if (ar_cur + cnt > ar_end) {
			if (lane == 0)
				d.T[(uint64_t) k * d.nseg + g] = 0;
			continue;
		}
		sp_emit<true>(accw, d.fpool + ar_cur, lane, G);
		if (lane == 0) {
			d.T[(uint64_t) k * d.nseg + g] = ((uint64_t) ar_cur << SP_LEN_BITS) | (uint64_t) cnt;
			atomicAdd(&a.row_len[k], cnt);
			atomicAdd(&d.block_sum[k >> 10], (unsigned long long) cnt);
		}
		ar_cur += cnt;
	}
	if (lane == 0) {
		atomicAdd(&a.ctr64[C64_INPUT], st_input);
		atomicAdd(&a.ctr64[C64_STREAM], ops);
		atomicAdd(&a.ctr[a.done_ctr], st_done);
		if (pool_full)
			atomicOr(&a.ctr[CTR_STATUS], 1);
	}
	// (pivotal entries are counted per lane)
	for (int sft = 32; sft >= 1; sft >>= 1)
		st_piv += __shfl_xor(st_piv, sft);
	if (lane == 0)
		atomicAdd(&a.ctr64[C64_ELIM], (unsigned long long) st_piv);
}

struct SpGatherArgs {
	const uint64_t *T;
	const uint32_t *fpool;
	int nrows, nseg;
	const int64_t *Sp;
	int *Sj, *Sx;
	int64_t cap;
	const int *q;                 // index among the non-pivotal columns -> column
};

// the fragments of a row, segment after segment, as (column, value) pairs at the row's final place
__global__ __launch_bounds__(256) void sp_gather_kernel(SpGatherArgs e)
{
	const int lane = threadIdx.x & 63;
	const int wave = (int) ((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int) ((gridDim.x * blockDim.x) >> 6);
	for (int k = wave; k < e.nrows; k += nwaves) {
		const int64_t off = e.Sp[k], end = e.Sp[k + 1];
		if (end > e.cap || end == off)
			continue;
		int64_t w = off;
		for (int g0 = 0; g0 < e.nseg; g0 += 64) {
			const uint64_t t = (g0 + lane < e.nseg) ? e.T[(uint64_t) k * e.nseg + g0 + lane] : 0;
			uint64_t live = __ballot((t & LEN_MASK) != 0);
			while (live != 0) {
				const int s = __builtin_ctzll(live);
				live &= live - 1;
				const uint64_t tt = readlane64(t, s);
				const int len = (int) (tt & LEN_MASK);
				const uint32_t *src = e.fpool + (tt >> SP_LEN_BITS);
				const int *q = e.q + (int64_t) (g0 + s) * SP_SEG;
				int *oj = e.Sj + w, *ox = e.Sx + w;
				for (int i = lane; i < len; i += 64) {
					const uint32_t en = src[i];
					oj[i] = q[en & 0xFFFFu];
					ox[i] = (int) en >> 16;
				}
				w += len;
			}
		}
	}
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool sparse_image_possible(int64_t prime)
{
	return sgn_eligible(prime) && env_sp("SPASM_HIP_SPARSE_IMAGE", -1) != 0;
}

// dependency tables of the build: per compact row (level order) its pivotal entries (compact row, negated balanced
// coefficient) and its non-pivotal entries (index among the non-pivotal columns, balanced value)
void sparse_image_plan(const FactPlan &P, spasm_hip_dfact *F, hipStream_t stream)
{
	SpImage &S = F->sp;
	const int r = P.r, rpad = P.rpad, m = P.m;
	const int64_t prime = P.prime;
	S.r = r;
	S.Sm = m - r;
	S.nseg = (S.Sm + SP_SEG - 1) / SP_SEG;
	S.nlevels = P.nlevels;
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
	S.lvl_lo.assign((size_t) P.nlevels + 1, 0);
	for (int l = 0; l < P.nlevels; l++)
		S.lvl_lo[l + 1] = S.lvl_lo[l] + P.lvl_count[l];
	std::vector<int> colmap((size_t) (m > 0 ? m : 1), 0);
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
	std::vector<uint64_t> dep_rp((size_t) r + 1, 0), np_rp((size_t) r + 1, 0);
	std::vector<uint2> dep, np;
	dep.reserve(P.ent.size());
	np.reserve(P.ent.size() / 4 + 16);
	for (int n = 0; n < r; n++) {
		const int c = label_of[n];
		for (uint64_t e = P.rp[c]; e < P.rp[c + 1]; e++) {
			const uint2 en = P.ent[e];
			const int32_t v = balanced(en.y);
			if (en.x < (uint32_t) rpad)
				dep.push_back(uint2{(uint32_t) cid[en.x], (uint32_t) (-v)});
			else
				np.push_back(uint2{en.x - (uint32_t) rpad, (uint32_t) v});
		}
		dep_rp[n + 1] = dep.size();
		np_rp[n + 1] = np.size();
	}
	S.ndeps = (int64_t) dep.size();
	S.nnp = (int64_t) np.size();
	S.d_col = dalloc<int>(m);
	S.d_dep_rp = dalloc<uint64_t>((int64_t) r + 1);
	S.d_dep = dalloc<uint2>((int64_t) dep.size());
	S.d_np_rp = dalloc<uint64_t>((int64_t) r + 1);
	S.d_np = dalloc<uint2>((int64_t) np.size());
	upload(S.d_col, colmap, stream);
	upload(S.d_dep_rp, dep_rp, stream);
	upload(S.d_dep, dep, stream);
	upload(S.d_np_rp, np_rp, stream);
	upload(S.d_np, np, stream);
	HIP_CHECK(hipStreamSynchronize(stream));          // the host vectors die here
	S.planned = true;
	S.valid = false;
	S.failed = false;
}

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
	sparse_image_drop_chunks(S);
	sh::big_free(S.d_col);
	sh::big_free(S.d_dep_rp);
	sh::big_free(S.d_dep);
	sh::big_free(S.d_np_rp);
	sh::big_free(S.d_np);
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
	if (!S.planned)
		die("sparse_image_build: the factor has no plan for the sparse image");
	S.valid = false;
	if (S.ev0 == nullptr) {
		HIP_CHECK(hipEventCreate(&S.ev0));
		HIP_CHECK(hipEventCreate(&S.ev1));
	}
	if (S.d_frag == nullptr)
		S.d_frag = dalloc<uint64_t>((int64_t) S.r * S.nseg);
	if (S.d_shard == nullptr)
		S.d_shard = dalloc<unsigned long long>((int64_t) SP_SHARDS * SHARD_STRIDE + 16);
	int *d_ovf = reinterpret_cast<int *>(S.d_shard + (size_t) SP_SHARDS * SHARD_STRIDE);
	// Room: the pool grows by chunks.  A build that runs out of room in some level allocates the next chunk (twice the size)
	// and redoes the levels from that one on; what it may take in all is bounded -- an R that needs more than half the bytes
	// of its dense form is not sparse, and the other paths are the better ones for it.
	size_t free_b = 0, total_b = 0;
	HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
	int64_t held = 0;
	for (int k = 0; k < S.nchunks; k++)
		held += S.chunk_cap[k] * 4;
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
		                                 : std::max<int64_t>((int64_t) 16 << 20, 64 * (F->nnz + S.r));
		if (env_sp("SPASM_HIP_SPARSE_IMAGE_CHUNK", 0) > 0)          // (tests: pool extensions on small inputs)
			cap = env_sp("SPASM_HIP_SPARSE_IMAGE_CHUNK", 0);
		cap = std::min<int64_t>(cap, std::max<int64_t>(budget / 4, (int64_t) SP_SHARDS * 64));
		cap = (cap + SP_SHARDS - 1) / SP_SHARDS * SP_SHARDS;
		S.d_chunk[0] = dalloc<uint32_t>(cap);
		S.chunk_cap[0] = cap;
		S.nchunks = 1;
	}
	SpBuildArgs b{};
	b.dep_rp = S.d_dep_rp;
	b.dep = S.d_dep;
	b.np_rp = S.d_np_rp;
	b.np = S.d_np;
	b.frag = S.d_frag;
	b.nseg = S.nseg;
	b.shard = S.d_shard;
	b.ovf_level = d_ovf;
	b.G = sgn_setup(F->prime);
	HIP_CHECK(hipEventRecord(S.ev0, stream));
	int chunk = 0, from_level = S.nlevels - 1;
	S.launches = 0;
	bool first = true;
	for (;;) {
		for (int k = 0; k < SP_MAX_CHUNKS; k++)
			b.pools.base[k] = S.d_chunk[k < S.nchunks ? k : 0];
		b.chunk = chunk;
		b.chunk_base = S.d_chunk[chunk];
		hipLaunchKernelGGL(sp_reset_shards_kernel, dim3(SP_SHARDS / 64), dim3(64), 0, stream, S.d_shard,
		                   (unsigned long long) (S.chunk_cap[chunk] / SP_SHARDS), first ? 1 : 0);
		HIP_CHECK(hipMemsetAsync(d_ovf, 0xFF, sizeof(int), stream));
		first = false;
		for (int l = from_level; l >= 0; l--) {
			b.row_lo = S.lvl_lo[l];
			b.row_hi = S.lvl_lo[l + 1];
			b.level = l;
			const int64_t ntasks = (int64_t) (b.row_hi - b.row_lo) * S.nseg;
			if (ntasks <= 0)
				continue;
			if (ntasks > 0x7FFFFFFFll)
				die("sparse_image_build: level %d has %lld (row, segment) pairs", l, (long long) ntasks);
			hipLaunchKernelGGL(sp_build_kernel, dim3((unsigned) ntasks), dim3(64), 0, stream, b);
			S.launches += 1;
		}
		HIP_CHECK(hipGetLastError());
		int ovf = -1;
		HIP_CHECK(hipMemcpyAsync(&ovf, d_ovf, sizeof(int), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		if (ovf < 0)
			break;
		// out of room in level `ovf`: the next chunk, and again from there
		int64_t total = 0;
		for (int k = 0; k < S.nchunks; k++)
			total += S.chunk_cap[k] * 4;
		int64_t cap = S.chunk_cap[S.nchunks - 1] * 2;
		if (chunk + 1 >= SP_MAX_CHUNKS || total + cap * 4 > budget) {
			if (verbose() >= 2)
				logmsg("[sparse image] gave up in level %d of %d: %.2f GB of fragments would not hold R (budget %.2f GB) -- R is not sparse\n", ovf,
				       S.nlevels, 1e-9 * (double) total, 1e-9 * (double) budget);
			S.failed = true;
			HIP_CHECK(hipEventRecord(S.ev1, stream));
			return false;
		}
		if (chunk + 1 >= S.nchunks) {
			S.d_chunk[S.nchunks] = dalloc<uint32_t>(cap);
			S.chunk_cap[S.nchunks] = cap;
			S.nchunks += 1;
		}
		chunk += 1;
		from_level = ovf;
	}
	HIP_CHECK(hipEventRecord(S.ev1, stream));
	std::vector<unsigned long long> h((size_t) SP_SHARDS * SHARD_STRIDE);
	HIP_CHECK(hipMemcpyAsync(h.data(), S.d_shard, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	S.ops_build = 0;
	S.nnz = 0;
	for (int s = 0; s < SP_SHARDS; s++) {
		S.ops_build += (int64_t) h[(size_t) s * SHARD_STRIDE + 2];
		S.nnz += (int64_t) h[(size_t) s * SHARD_STRIDE + 3];
	}
	S.pool_used = S.nnz;          // (entries written, fragments of redone levels included: what a rebuild needs in one chunk)
	if (chunk > 0) {
		// levels were redone: their fragments were counted twice
		HIP_CHECK(hipMemsetAsync(S.d_shard, 0, sizeof(unsigned long long), stream));
		hipLaunchKernelGGL(sp_sum_frag_kernel, dim3(1024), dim3(256), 0, stream, S.d_frag, (int64_t) S.r * S.nseg, S.d_shard);
		unsigned long long exact = 0;
		HIP_CHECK(hipMemcpyAsync(&exact, S.d_shard, sizeof(exact), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		S.nnz = (int64_t) exact;
	}
	S.valid = true;
	S.failed = false;
	S.builds += 1;
	if (verbose() >= 2)
		logmsg("[sparse image] R: %d rows x %d columns in %d segments, %lld entries (%.3f %% of the dense form, %.1f per row), %lld multiply-adds, %d "
		       "launches, %d pool chunk(s) of %.2f GB in all\n",
		       S.r, S.Sm, S.nseg, (long long) S.nnz, 100.0 * (double) S.nnz / std::max(1.0, (double) S.r * (double) S.Sm),
		       (double) S.nnz / std::max(1, S.r), (long long) S.ops_build, S.launches, S.nchunks, [&] {
			       double t = 0;
			       for (int k = 0; k < S.nchunks; k++)
				       t += 4e-9 * (double) S.chunk_cap[k];
			       return t;
		       }());
	return true;
}

// S rows from the sparse image: sparse rows in W's final arrays (dense_out == nullptr) or dense rows.
//   fpool / fcap: room for the fragments of S (4-byte entries); T: nrows * nseg words; block_sum: (nrows + 1023) / 1024 words
void launch_sparse_image_apply(const SchurArgs &a, const spasm_hip_dfact *F, uint32_t *dense_out, int64_t ldS, uint32_t *fpool, int64_t fcap,
                               uint64_t *T, unsigned long long *block_sum, int64_t *Sp, int *Sj, int *Sx, int64_t cap, hipStream_t stream,
                               hipEvent_t ev_gather)
{
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
	for (int k = 0; k < SP_MAX_CHUNKS; k++)
		d.pools.base[k] = S.d_chunk[k < S.nchunks ? k : 0];
	d.G = sgn_setup(F->prime);
	d.fpool = fpool;
	d.fcap = fcap;
	d.T = T;
	d.block_sum = block_sum;
	d.dense_out = dense_out;
	d.ldS = ldS;
	const int64_t ntasks = (int64_t) a.nrows * S.nseg;
	if (ntasks <= 0)
		return;
	// nine waves per CU (16 KB of LDS each), every wave a workgroup of its own
	const int per_cu = std::max(1, std::min(16, env_sp("SPASM_HIP_SPARSE_IMAGE_WAVES", 9)));
	const int blocks = (int) std::min<int64_t>(ntasks, (int64_t) prop.multiProcessorCount * per_cu);
	// a wave reserves the room of its fragments 32,768 entries at a time when the pool is large enough for every wave to
	// strand one such arena; else fragment by fragment
	d.arena = (fcap >= (int64_t) blocks * 32768 * 8) ? 32768 : 0;
	if (dense_out == nullptr) {
		const int nblocks = (a.nrows + 1023) / 1024;
		HIP_CHECK(hipMemsetAsync(block_sum, 0, (size_t) nblocks * sizeof(unsigned long long), stream));
		HIP_CHECK(hipMemsetAsync(a.row_len, 0, (size_t) a.nrows * sizeof(int), stream));
		HIP_CHECK(hipMemsetAsync(Sp, 0, sizeof(int64_t), stream));
	}
	hipLaunchKernelGGL(sp_apply_kernel, dim3(blocks), dim3(64), 0, stream, d);
	HIP_CHECK(hipGetLastError());
	if (dense_out != nullptr)
		return;
	if (ev_gather != nullptr)
		HIP_CHECK(hipEventRecord(ev_gather, stream));
	launch_scan_lengths(a.row_len, a.nrows, block_sum, Sp, cap, a.ctr, stream);
	SpGatherArgs e{T, fpool, a.nrows, S.nseg, Sp, Sj, Sx, cap, a.q};
	const int gblocks = std::max(1, std::min((a.nrows + 3) / 4, prop.multiProcessorCount * 8));
	hipLaunchKernelGGL(sp_gather_kernel, dim3(gblocks), dim3(256), 0, stream, e);
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
	hipLaunchKernelGGL(sp_census_kernel, dim3(2048), dim3(256), 0, stream, S.d_frag, (int64_t) S.r * S.nseg, pools, d);
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
