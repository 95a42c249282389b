// Row-group Schur complement WITHOUT atomics: a left-looking ("pull") numeric pass.
//
// Same data layout as schur_group_kernel (schur_kernels.hip): a workgroup owns 64 consecutive rows of the batch,
// lane = row, and a slice X[label][64] of accumulator lines (256 bytes per label), all zero between groups.  That
// kernel pushes: for every pending pivot c and every entry (t, u) of U'[c] it issues one no-return atomic on the line
// of t, and it is bound by the chip's rate of such atomic requests.  Here the work is split in two sweeps over the
// elimination levels:
//   symbolic: which labels does the group touch?  One bit per label in LDS; a touched pivot marks the targets of its
//             row of U' (right-looking on bits only: LDS traffic, the entries of U' come out of the L2);
//   numeric:  for every touched label t of a level -- final once the earlier levels are done --
//             X[t] = a[t] - sum over the touched pivots c that hold t of X[c] * u_ct,
//             with plain coalesced loads of the (final) lines X[c] and ONE store.  Needs U' by target label (CSC).
// Lines are reduced residues (u32) whatever the prime: the sums live in 64-bit registers.
#include <algorithm>

#include "device_types.h"
#include "field_dev.h"

namespace sh {

namespace {

constexpr int PL_NW = 2;            // waves per workgroup
constexpr int PL_LIST = 1024;       // touched labels of a level handled per pass
constexpr int PL_B = 4;             // labels a wave works on at a time (their loads are in flight together)
constexpr int PL_P = 4;             // predecessors loaded per label and round

struct PullArgs {
	SchurArgs a;
	unsigned char *scratch;
	int64_t slot_bytes;
	const uint64_t *cp;           // r + Sm + 1 offsets into cent, by target label
	const uint2 *cent;            // (source label, value * 2^32 mod p)
	const int2 *lvl;              // [first label, last label + 1) of every level
	int nlev;
	int words;                    // bitmap words (r + Sm bits)
};

template <typename V> __device__ __forceinline__ V pl_ld(const V *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // from the L2 (stores are write-through)
}

__device__ __forceinline__ void pl_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int pl_exclusive_scan(int v, int lane, int &total)
{
	int x = v;
#pragma unroll
	for (int dd = 1; dd < 64; dd <<= 1) {
		const int y = __shfl_up(x, dd);
		if (lane >= dd)
			x += y;
	}
	total = __shfl(x, 63);
	return x - v;
}

__device__ __forceinline__ unsigned long long pl_wave_sum(unsigned long long v)
{
	for (int s = 32; s >= 1; s >>= 1) {
		const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) v, s);
		const uint32_t hi = (uint32_t) __shfl_xor((int) (uint32_t) (v >> 32), s);
		v += ((unsigned long long) hi << 32) | lo;
	}
	return v;
}

__global__ __launch_bounds__(64 * PL_NW) void schur_pull_kernel(PullArgs d)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t pl_lds[];
	__shared__ int s_group, s_count;
	__shared__ unsigned long long s_off;
	__shared__ uint32_t tlist[PL_LIST];
	__shared__ int cnt_w[PL_NW][64];
	const SchurArgs &a = d.a;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	constexpr int T = 64 * PL_NW;
	const uint32_t r = (uint32_t) a.r;
	const int nlab = a.r + a.Sm;
	const MontDev F = a.F;
	uint32_t *tbits = pl_lds;                  // touched labels
	uint32_t *ibits = pl_lds + d.words;        // labels that hold an input entry
	uint32_t *X = reinterpret_cast<uint32_t *>(d.scratch + (int64_t) blockIdx.x * d.slot_bytes);
	const int ngroups = (a.nrows + 63) / 64;
	unsigned long long st_input = 0, st_elim = 0, st_stream = 0, st_gp = 0;
	int st_done = 0;
	auto tbit = [&](uint32_t c) { return (tbits[c >> 5] >> (c & 31)) & 1u; };

	for (;;) {
		if (tid == 0)
			s_group = atomicAdd(&a.ctr[a.next_ctr], 1);
		__syncthreads();
		const int g = s_group;
		__syncthreads();
		if (g >= ngroups)
			break;
		const int k = g * 64 + lane;
		const bool valid = k < a.nrows;
		const int i = valid ? a.rows[k] : 0;
		const int64_t lo = valid ? a.Ap[i] : 0, hi = valid ? a.Ap[i + 1] : 0;
		for (int t = tid; t < 2 * d.words; t += T)
			pl_lds[t] = 0;
		__syncthreads();
		// ---- input rows into the (all-zero) slice; their labels are touched ----
		if (wave == 0)
			st_input += (unsigned long long) (hi - lo);
		for (int64_t px = lo + wave; px < hi; px += PL_NW) {
			const uint32_t c = a.lab[a.Aj[px]];
			const uint32_t v = reduce_sum(from_balanced(a.Ax[px], F), F);
			if (v != 0) {
				X[(int64_t) c * 64 + lane] = v;
				atomicOr(&tbits[c >> 5], 1u << (c & 31));
				atomicOr(&ibits[c >> 5], 1u << (c & 31));
			}
		}
		pl_drain();
		__syncthreads();
		// ---- symbolic sweep: touched pivots mark the targets of their rows ----
		for (int l = 0; l < d.nlev; l++) {
			const int2 lv = d.lvl[l];
			for (int c = lv.x + tid; c < lv.y; c += T) {
				if (!tbit((uint32_t) c))
					continue;
				for (uint64_t e = a.rp[c]; e < a.rp[c + 1]; e++) {
					const uint32_t t = a.ent[e].x;
					atomicOr(&tbits[t >> 5], 1u << (t & 31));
				}
			}
			__syncthreads();
		}
		// ---- numeric sweep: level by level, then the non-pivotal labels ----
		for (int l = 0; l <= d.nlev; l++) {
			const int2 lv = (l < d.nlev) ? d.lvl[l] : int2{(int) r, nlab};
			for (int base = lv.x; base < lv.y; base += PL_LIST) {
				// the touched labels of [base, base + PL_LIST) as a list
				if (tid == 0)
					s_count = 0;
				__syncthreads();
				const int top = min(lv.y, base + PL_LIST);
				for (int c0 = base; c0 < top; c0 += T) {
					const int c = c0 + tid;
					const bool on = c < top && tbit((uint32_t) c);
					const uint64_t mk = __ballot(on);
					int off = 0;
					if (lane == 0 && mk != 0)
						off = atomicAdd(&s_count, __popcll(mk));
					off = __builtin_amdgcn_readfirstlane(off);
					if (on)
						tlist[off + __popcll(mk & ((1ull << lane) - 1ull))] = (uint32_t) c;
				}
				__syncthreads();
				const int count = s_count;
				for (int q0 = wave * PL_B; q0 < count; q0 += PL_NW * PL_B) {
					uint32_t tt[PL_B];
					uint64_t e0[PL_B], e1[PL_B];
					unsigned long long acc[PL_B];
#pragma unroll
					for (int b = 0; b < PL_B; b++) {
						const bool have = q0 + b < count;
						tt[b] = have ? tlist[q0 + b] : 0u;
						e0[b] = have ? d.cp[tt[b]] : 0;
						e1[b] = have ? d.cp[tt[b] + 1] : 0;
						acc[b] = 0;
					}
					// own (input) values and the first PL_P predecessors of every label: all loads first
					uint32_t own[PL_B], pv[PL_B][PL_P], py[PL_B][PL_P];
#pragma unroll
					for (int b = 0; b < PL_B; b++) {
						const bool have = q0 + b < count;
						own[b] = (have && ((ibits[tt[b] >> 5] >> (tt[b] & 31)) & 1u)) ? pl_ld(&X[(int64_t) tt[b] * 64 + lane]) : 0u;
#pragma unroll
						for (int s = 0; s < PL_P; s++) {
							pv[b][s] = 0;
							py[b][s] = 0;
							if (e0[b] + s < e1[b]) {
								const uint2 en = d.cent[e0[b] + s];
								if (tbit(en.x)) {
									pv[b][s] = pl_ld(&X[(int64_t) en.x * 64 + lane]);
									py[b][s] = en.y;
								}
							}
						}
					}
#pragma unroll
					for (int b = 0; b < PL_B; b++) {
						acc[b] = own[b];
#pragma unroll
						for (int s = 0; s < PL_P; s++)
							if (pv[b][s] != 0)
								acc[b] += (unsigned long long) (F.p - montmul(pv[b][s], py[b][s], F));
						// the rest of a long predecessor list
						for (uint64_t e = e0[b] + PL_P; e < e1[b]; e++) {
							const uint2 en = d.cent[e];
							if (!tbit(en.x))
								continue;
							const uint32_t x = pl_ld(&X[(int64_t) en.x * 64 + lane]);
							if (x != 0)
								acc[b] += (unsigned long long) (F.p - montmul(x, en.y, F));
						}
						if (q0 + b < count) {
							const uint32_t val = reduce_sum(acc[b], F);
							X[(int64_t) tt[b] * 64 + lane] = val;
							if (tt[b] < r) {
								const uint64_t nzm = __ballot(val != 0);
								const unsigned long long nzc = (unsigned long long) __popcll(nzm);
								st_elim += nzc;
								st_stream += nzc * (unsigned long long) (a.rp[tt[b] + 1] - a.rp[tt[b]]);
								st_gp += (nzm != 0) ? 1ull : 0ull;
							}
						}
					}
				}
				pl_drain();               // the lines of this level are in the L2 before the next level reads them
				__syncthreads();
			}
		}
		// ---- output: touched non-pivotal lines, lane = row, waves take contiguous shares of the label range ----
		const int np_words0 = (int) (r >> 5), np_words1 = (nlab + 31) >> 5;
		const int span = (np_words1 - np_words0 + PL_NW - 1) / PL_NW;
		const int w_lo = np_words0 + wave * span, w_hi = min(np_words1, w_lo + span);
		int count = 0;
		for (int w = w_lo; w < w_hi; w++) {
			uint32_t bits = tbits[w];
			if (w == np_words0)
				bits &= ~((1u << (r & 31)) - 1u);          // (labels below r in the first word are pivotal)
			while (bits != 0) {
				const uint32_t t = (uint32_t) (w * 32 + __builtin_ctz(bits));
				bits &= bits - 1;
				count += (valid && pl_ld(&X[(int64_t) t * 64 + lane]) != 0) ? 1 : 0;
			}
		}
		cnt_w[wave][lane] = count;
		__syncthreads();
		int mine_before = 0, mine_total = 0;
#pragma unroll
		for (int w = 0; w < PL_NW; w++) {
			mine_before += (w < wave) ? cnt_w[w][lane] : 0;
			mine_total += cnt_w[w][lane];
		}
		int group_total;
		const int excl = pl_exclusive_scan(mine_total, lane, group_total);
		unsigned long long got = 0;
		if (lane == 0)
			got = (wave == 0) ? atomicAdd(&a.ctr64[C64_POOL], (unsigned long long) group_total) : 0ull;
		if (tid == 0)
			s_off = got;
		__syncthreads();
		const int64_t off_group = (int64_t) s_off;
		const bool fits = off_group + group_total <= a.pool_cap;
		int64_t wpos = off_group + excl + mine_before;
		for (int w = w_lo; w < w_hi; w++) {
			uint32_t bits = tbits[w];
			if (w == np_words0)
				bits &= ~((1u << (r & 31)) - 1u);
			while (bits != 0) {
				const uint32_t t = (uint32_t) (w * 32 + __builtin_ctz(bits));
				bits &= bits - 1;
				const uint32_t v = pl_ld(&X[(int64_t) t * 64 + lane]);
				if (valid && v != 0 && fits) {
					a.pool_j[wpos] = a.q[t - r];
					a.pool_x[wpos] = to_balanced(v, F);
					wpos += 1;
				}
			}
		}
		if (wave == 0 && valid) {
			if (fits) {
				a.row_off[k] = (off_group + excl) | (1LL << 62);       // sorted by column already
				a.row_len[k] = mine_total;
			} else {
				a.row_len[k] = -1;
			}
		}
		if (tid == 0 && !fits)
			atomicOr(&a.ctr[CTR_STATUS], 1);
		if (wave == 0)
			st_done += (valid && fits) ? 1 : 0;
		pl_drain();
		__syncthreads();
		// ---- every touched line back to zero (the slice is all zero between groups) ----
		for (int w = tid >> 6; w < d.words; w += PL_NW) {
			uint32_t bits = tbits[w];
			while (bits != 0) {
				const uint32_t t = (uint32_t) (w * 32 + __builtin_ctz(bits));
				bits &= bits - 1;
				X[(int64_t) t * 64 + lane] = 0;
			}
		}
		pl_drain();
		__syncthreads();
	}
	// statistics: eliminations / streamed entries / applied pivots are wave-uniform counts, the rest is per lane
	const unsigned long long inp = pl_wave_sum(st_input);
	int done = st_done;
	for (int sft = 32; sft >= 1; sft >>= 1)
		done += __shfl_xor(done, sft);
	if (lane == 0) {
		atomicAdd(&a.ctr64[C64_ELIM], st_elim);
		atomicAdd(&a.ctr64[C64_STREAM], st_stream);
		atomicAdd(&a.ctr64[C64_WAVEPIV], st_gp);
		atomicAdd(&a.ctr64[C64_INPUT], inp);
		atomicAdd(&a.ctr[a.done_ctr], done);
	}
}

}  // namespace

size_t pull_lds_bytes(int rpad, int Sm) { return (size_t) 2 * (size_t) ((rpad + Sm + 31) / 32 + 1) * 4; }

int64_t pull_slot_bytes(int rpad, int Sm) { return ((int64_t) rpad + Sm) * 256; }

void launch_schur_pull(const SchurArgs &a, unsigned char *scratch, int64_t slot_bytes, const uint64_t *cp, const uint2 *cent,
                       const int2 *lvl, int nlev, int blocks, hipStream_t stream)
{
	PullArgs d{};
	d.a = a;
	d.scratch = scratch;
	d.slot_bytes = slot_bytes;
	d.cp = cp;
	d.cent = cent;
	d.lvl = lvl;
	d.nlev = nlev;
	d.words = (a.r + a.Sm + 31) / 32 + 1;
	const size_t lds = pull_lds_bytes(a.r, a.Sm);
	static size_t configured = 0;
	if (lds > configured) {
		HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&schur_pull_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
		configured = lds;
	}
	hipLaunchKernelGGL(schur_pull_kernel, dim3(blocks), dim3(64 * PL_NW), lds, stream, d);
	HIP_CHECK(hipGetLastError());
}

}  // namespace sh
