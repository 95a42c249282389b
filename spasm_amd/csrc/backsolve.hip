// Back-substituted factor image and the Schur complement computed from it.
//
// The reference reduces every non-pivotal row a of A by a sparse triangular solve against U
// (spasm_schur.c:86-171 -> spasm_triangular.c:110-146): x = a U_pp^-1, then s = a_n - x U_pn.  The
// same product can be bracketed the other way round:
//
//       S = A_n - A_p (U_pp^-1 U_pn) = A_n - A_p R,
//
// where row c of R is the non-pivotal part of the fully reduced pivot row c (the rows spasm_rref would
// produce, spasm_rref.c:25).  R is dense, r x Sm (Sm = number of non-pivotal columns).  When Sm is small
// -- a Schur complement that is going to be dense anyway -- building R costs nnz(U') * Sm multiply-adds,
// streaming and free of atomics, and every reduced row is then a combination of the few rows of R its
// pivotal entries select.  mk13.b5: 130183 x 4952 (2.6 GB) against 3.97e9 (row, pivot) eliminations.
// Arithmetic mod p is exact, so the result is the same matrix, bit for bit.
//
// R[c] = U_n[c] - sum_{t pivotal in U'[c]} u_ct R[t], and t always lies in a later elimination level
// than c.  The columns of R are independent: one workgroup owns a slab of 16 columns and walks the rows
// from the last level to the first, with no communication between workgroups at all.  Inside a
// workgroup the chain of levels runs in LDS: rows are cut into chunks of <= RING consecutive rows;
//   phase A (throughput): every row of the chunk gathers what it needs from outside the chunk
//            (rows of R that are final, in HBM) into an LDS ring,
//   phase B (latency): level by level, rows pick up their dependencies inside the chunk from the ring
//            (one LDS round trip + one workgroup barrier per level),
//   phase C: the ring is written back to R.
#include <algorithm>
#include <cinttypes>
#include <vector>

#include "device_types.h"
#include "field_dev.h"

namespace sh {

namespace {

constexpr int BS_CW = 16;          // columns per slab (one 64-byte segment of a row of R)
constexpr int BS_NW = 8;           // waves per workgroup
constexpr int BS_RING = 768;       // rows per chunk (LDS ring: RING * 64 bytes)
constexpr int BS_NEARCAP = 1024;   // dependencies inside a chunk
constexpr int BS_STEPCAP = 128;    // levels with such dependencies inside a chunk
constexpr int BS_ROWS_PER_ITER = (64 / BS_CW) * BS_NW;          // rows handled by one instruction of every wave
constexpr int BS_ITERS = BS_RING / BS_ROWS_PER_ITER;            // 24
constexpr int BS_UNR = 8;                                      // rows in flight per lane in phase A
static_assert(BS_ITERS % BS_UNR == 0, "phase A is unrolled in parts of BS_UNR rows");
constexpr uint32_t BS_NONE = 0xFFFFFFFFu;

template <typename T> T *dalloc(int64_t count)
{
	T *p = nullptr;
	HIP_CHECK(hipMalloc((void **) &p, (size_t) (count > 0 ? count : 1) * sizeof(T)));
	return p;
}

template <typename T> void upload(T *dst, const std::vector<T> &src, hipStream_t s)
{
	if (!src.empty())
		HIP_CHECK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

struct BsArgs {
	uint32_t *R;
	int64_t ldR;
	int nchunks;
	const BsChunk *chunk;
	const int *chunk_extra;       // per chunk: 1 when some row has more than two dependencies outside the chunk
	const int2 *step;
	const uint2 *brow;
	const uint2 *near;
	const uint4 *far_head;
	const uint64_t *far_rp;
	const uint2 *far;
	const uint64_t *np_rp;
	const uint2 *np;
	int r;
	MontDev F;
};

// R <- U_n (values out of Montgomery form); R was zeroed before
__global__ __launch_bounds__(256) void bs_init_kernel(BsArgs b)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= b.r)
		return;
	uint32_t *row = b.R + (int64_t) c * b.ldR;
	for (uint64_t e = b.np_rp[c]; e < b.np_rp[c + 1]; e++) {
		const uint2 en = b.np[e];
		row[en.x] = montmul(en.y, 1u, b.F);
	}
}

__global__ __launch_bounds__(64 * BS_NW, 4) void backsolve_kernel(BsArgs b)
{
	__shared__ uint32_t ring[BS_RING * BS_CW];
	__shared__ uint2 near[BS_NEARCAP];
	__shared__ uint2 brow[BS_RING];
	__shared__ int2 step[BS_STEPCAP];
	const int tid = threadIdx.x;
	const int lane = tid & 63, wave = tid >> 6;
	const int rs = lane >> 4, col = lane & 15;
	const MontDev F = b.F;
	const int64_t ldR = b.ldR;
	uint32_t *Rs = b.R + (int64_t) blockIdx.x * BS_CW + col;
	const int slot0 = wave * (64 / BS_CW) + rs;          // this lane's row slot within an iteration

	for (int k = 0; k < b.nchunks; k++) {
		const BsChunk ch = b.chunk[k];
		const int nrows = ch.hi - ch.lo;
		// metadata of phase B into LDS (the same for every slab: served by the L2)
		for (int t = tid; t < ch.nnear; t += 64 * BS_NW)
			near[t] = b.near[ch.near0 + t];
		for (int t = tid; t < ch.nbrow; t += 64 * BS_NW)
			brow[t] = b.brow[ch.brow0 + t];
		for (int t = tid; t < ch.nsteps; t += 64 * BS_NW)
			step[t] = b.step[ch.step0 + t];

		// ---- phase A: own row + dependencies outside the chunk, BS_UNR rows in flight per lane ----
		for (int half = 0; half < BS_ITERS / BS_UNR; half++) {
			if (half * BS_UNR * BS_ROWS_PER_ITER >= nrows)
				break;
			uint4 hd[BS_UNR];
			uint32_t acc[BS_UNR];
#pragma unroll
			for (int u = 0; u < BS_UNR; u++) {
				const int s = (half * BS_UNR + u) * BS_ROWS_PER_ITER + slot0;
				const bool ok = s < nrows;
				const int c = ch.lo + (ok ? s : 0);
				hd[u] = b.far_head[c];
				acc[u] = Rs[(int64_t) c * ldR];
				if (!ok)
					hd[u].x = hd[u].z = BS_NONE;
			}
			uint32_t v0[BS_UNR], v1[BS_UNR];
#pragma unroll
			for (int u = 0; u < BS_UNR; u++) {
				v0[u] = (hd[u].x != BS_NONE) ? Rs[(int64_t) hd[u].x * ldR] : 0u;
				v1[u] = (hd[u].z != BS_NONE) ? Rs[(int64_t) hd[u].z * ldR] : 0u;
			}
#pragma unroll
			for (int u = 0; u < BS_UNR; u++) {
				const int s = (half * BS_UNR + u) * BS_ROWS_PER_ITER + slot0;
				uint32_t x = acc[u];
				if (hd[u].x != BS_NONE)
					x = submod(x, montmul(v0[u], hd[u].y, F), F);
				if (hd[u].z != BS_NONE)
					x = submod(x, montmul(v1[u], hd[u].w, F), F);
				if (s < nrows)
					ring[s * BS_CW + col] = x;
			}
		}
		__syncthreads();
		if (b.chunk_extra[k]) {
			// rows with more than two outside dependencies (long rows of U): the rest of their lists
			for (int s = slot0; s < nrows; s += BS_ROWS_PER_ITER) {
				const int c = ch.lo + s;
				const uint64_t e0 = b.far_rp[c], e1 = b.far_rp[c + 1];
				if (e0 == e1)
					continue;
				uint32_t x = ring[s * BS_CW + col];
				for (uint64_t e = e0; e < e1; e++) {
					const uint2 en = b.far[e];
					x = submod(x, montmul(Rs[(int64_t) en.x * ldR], en.y, F), F);
				}
				ring[s * BS_CW + col] = x;
			}
			__syncthreads();
		}

		// ---- phase B: the chain of levels, in LDS ----
		for (int st = 0; st < ch.nsteps; st++) {
			const int2 sp = step[st];
			for (int q = sp.x + slot0; q < sp.y; q += BS_ROWS_PER_ITER) {
				const uint2 br = brow[q];
				const int slot = (int) (br.x & 0xFFFFu), cnt = (int) (br.x >> 16);
				uint32_t x = ring[slot * BS_CW + col];
				for (int j = 0; j < cnt; j++) {
					const uint2 en = near[br.y + j];
					x = submod(x, montmul(ring[en.x * BS_CW + col], en.y, F), F);
				}
				ring[slot * BS_CW + col] = x;
			}
			__syncthreads();
		}

		// ---- phase C: write the chunk back ----
		for (int s = slot0; s < nrows; s += BS_ROWS_PER_ITER)
			Rs[(int64_t) (ch.lo + s) * ldR] = ring[s * BS_CW + col];
		__syncthreads();          // (workgroup-scope release/acquire: later chunks read these rows)
	}
}

// --------------------------------------------------------------------------
// S = A_n - A_p R, one wave per row.  The row of S is accumulated in LDS (one
// u32 per non-pivotal column), the pivotal entries of the input row are
// collected in a small LDS list and applied tile by tile.
// --------------------------------------------------------------------------
constexpr int AP_LIST = 256;       // pivotal entries applied per pass
constexpr int AP_TU = 4;           // 64-column tiles per inner step

struct ApplyArgs {
	SchurArgs a;
	const uint32_t *R;
	int64_t ldR;
	const int *col;               // column -> compact id
	int r;                        // rows of R
	int Smpad;                    // Sm rounded up to 64 * AP_TU
	int waves;                    // waves per workgroup
	uint32_t *dense_out;
	int64_t ldS;
};

__global__ __launch_bounds__(512) void bs_apply_kernel(ApplyArgs d)
{
	extern __shared__ unsigned char lds_raw[];
	const SchurArgs &a = d.a;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const MontDev F = a.F;
	const int Sm = a.Sm, Smpad = d.Smpad;
	uint32_t *xbuf = reinterpret_cast<uint32_t *>(lds_raw) + (size_t) wave * ((size_t) Smpad + 2 * AP_LIST);
	uint2 *plist = reinterpret_cast<uint2 *>(xbuf + Smpad);
	const int64_t ldR = d.ldR;
	unsigned long long st_input = 0, st_piv = 0;
	int st_done = 0;

	for (int k = blockIdx.x * d.waves + wave; k < a.nrows; k += gridDim.x * d.waves) {
		const int i = a.rows[k];
		const int64_t lo = a.Ap[i], hi = a.Ap[i + 1];
		st_input += (unsigned long long) (hi - lo);
		for (int t = lane; t < Smpad; t += 64)
			xbuf[t] = 0;
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
					uint32_t sum = xbuf[t] + v;
					if (sum < v || sum >= F.p)
						sum -= F.p;
					xbuf[t] = sum;
				} else {
					piv = v != 0;
				}
			}
			const uint64_t mk = __ballot(piv);
			if (piv)
				plist[npl + __popcll(mk & ((1ull << lane) - 1ull))] = uint2{cid, montmul(v, F.r2, F)};
			npl += __popcll(mk);
			st_piv += (unsigned long long) __popcll(mk);
			const bool last = base + 64 >= hi;
			if (npl > 0 && (last || npl + 64 > AP_LIST)) {
				// apply the queued pivotal entries: x[tile] -= sum_e a_e R[e][tile]
				for (int t0 = 0; t0 < Smpad; t0 += 64 * AP_TU) {
					uint32_t acc[AP_TU];
#pragma unroll
					for (int u = 0; u < AP_TU; u++)
						acc[u] = xbuf[t0 + u * 64 + lane];
					int e = 0;
					for (; e + 2 <= npl; e += 2) {
						const uint2 p0 = plist[e], p1 = plist[e + 1];
						const uint32_t *r0 = d.R + (int64_t) p0.x * ldR + t0 + lane;
						const uint32_t *r1 = d.R + (int64_t) p1.x * ldR + t0 + lane;
						uint32_t w0[AP_TU], w1[AP_TU];
#pragma unroll
						for (int u = 0; u < AP_TU; u++) {
							w0[u] = r0[u * 64];
							w1[u] = r1[u * 64];
						}
#pragma unroll
						for (int u = 0; u < AP_TU; u++) {
							acc[u] = submod(acc[u], montmul(w0[u], p0.y, F), F);
							acc[u] = submod(acc[u], montmul(w1[u], p1.y, F), F);
						}
					}
					if (e < npl) {
						const uint2 p0 = plist[e];
						const uint32_t *r0 = d.R + (int64_t) p0.x * ldR + t0 + lane;
#pragma unroll
						for (int u = 0; u < AP_TU; u++)
							acc[u] = submod(acc[u], montmul(r0[u * 64], p0.y, F), F);
					}
#pragma unroll
					for (int u = 0; u < AP_TU; u++)
						xbuf[t0 + u * 64 + lane] = acc[u];
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
				out[t] = xbuf[t];
			if (lane == 0)
				a.row_len[k] = Sm;
			st_done += 1;
			continue;
		}
		int count = 0;
		for (int t0 = 0; t0 < Smpad; t0 += 64)
			count += __popcll(__ballot(xbuf[t0 + lane] != 0));
		unsigned long long got = 0;
		if (lane == 0)
			got = atomicAdd(&a.ctr64[C64_POOL], (unsigned long long) count);
		const uint32_t g_lo = __builtin_amdgcn_readfirstlane((uint32_t) got);
		const uint32_t g_hi = __builtin_amdgcn_readfirstlane((uint32_t) (got >> 32));
		const int64_t off = (int64_t) (((uint64_t) g_hi << 32) | g_lo);
		const bool fits = off + count <= a.pool_cap;
		if (fits) {
			int64_t wpos = off;
			for (int t0 = 0; t0 < Smpad; t0 += 64) {
				const uint32_t v = xbuf[t0 + lane];
				const uint64_t mk = __ballot(v != 0);
				if (v != 0) {
					const int64_t dst = wpos + __popcll(mk & ((1ull << lane) - 1ull));
					a.pool_j[dst] = a.q[t0 + lane];
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
	if (lane == 0) {
		atomicAdd(&a.ctr64[C64_INPUT], st_input);
		atomicAdd(&a.ctr64[C64_ELIM], st_piv);          // rows of R combined (not the reference's count of eliminations)
		atomicAdd(&a.ctr[a.done_ctr], st_done);
	}
}

}  // namespace

// --------------------------------------------------------------------------
// host plan
// --------------------------------------------------------------------------
// Is the back-substituted image worth having for this factor?  Memory: r x Sm words.  Work: nnz(U') * Sm.
bool backsolve_eligible(int r, int Sm, int64_t nnz_u, int64_t *bytes)
{
	const int64_t ldR = ((int64_t) Sm + 255) / 256 * 256;
	*bytes = (int64_t) r * ldR * 4;
	if (r <= 0 || Sm <= 0)
		return false;
	if (Sm > 24576)                       // the apply kernel keeps one row of S in LDS (96 KB)
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
	B.ldR = ((int64_t) B.Sm + 255) / 256 * 256;          // whole tiles of the apply kernel (64 * AP_TU columns): the padding stays zero
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

	// split every row into pivotal dependencies (compact ids) and non-pivotal entries
	std::vector<uint64_t> dep_rp((size_t) r + 1, 0), np_rp((size_t) r + 1, 0);
	std::vector<uint2> dep, np;
	dep.reserve(P.ent.size());
	np.reserve(P.ent.size());
	for (int n = 0; n < r; n++) {
		const int c = label_of[n];
		for (uint64_t e = P.rp[c]; e < P.rp[c + 1]; e++) {
			const uint2 en = P.ent[e];
			if (en.x < (uint32_t) rpad)
				dep.push_back(uint2{(uint32_t) cid[en.x], en.y});
			else
				np.push_back(uint2{en.x - (uint32_t) rpad, en.y});
		}
		dep_rp[n + 1] = dep.size();
		np_rp[n + 1] = np.size();
	}

	// chunks, from the last row to the first
	std::vector<BsChunk> chunks;
	std::vector<int> chunk_extra;
	std::vector<int2> steps;
	std::vector<uint2> brow, near;
	std::vector<uint4> far_head((size_t) (r > 0 ? r : 1), uint4{BS_NONE, 0u, BS_NONE, 0u});
	std::vector<uint64_t> far_rp((size_t) r + 1, 0);
	std::vector<uint2> far;
	std::vector<uint64_t> far_cnt((size_t) (r > 0 ? r : 1), 0);
	// first pass: chunk boundaries (a chunk grows downwards while its rows, the dependencies they have inside the
	// chunk and the levels holding such rows fit the LDS arrays of the kernel)
	int hi = r;
	while (hi > 0) {
		BsChunk ch{};
		ch.hi = hi;
		int lo = hi, nnear = 0, nsteps = 0, nbrow = 0, last_level = -1;
		while (lo > 0 && hi - lo < BS_RING) {
			const int c = lo - 1;
			int nc = 0;
			for (uint64_t e = dep_rp[c]; e < dep_rp[c + 1]; e++)
				nc += dep[e].x < (uint32_t) hi;
			const bool new_step = nc > 0 && level[c] != last_level;
			if (nnear + nc > BS_NEARCAP || nc > 65535 || (new_step && nsteps + 1 > BS_STEPCAP))
				break;                           // (the first row of a chunk never has dependencies inside it)
			if (nc > 0) {
				nsteps += new_step ? 1 : 0;
				last_level = level[c];
				nbrow += 1;
				nnear += nc;
			}
			lo = c;
		}
		ch.lo = lo;
		ch.nsteps = nsteps;
		ch.nnear = nnear;
		ch.nbrow = nbrow;
		chunks.push_back(ch);
		chunk_extra.push_back(0);
		hi = lo;
	}
	// second pass: the lists of every chunk, slots relative to its first row
	{
		for (size_t k = 0; k < chunks.size(); k++) {
			BsChunk &ch = chunks[k];
			ch.step0 = (int) steps.size();
			ch.near0 = (int) near.size();
			ch.brow0 = (int) brow.size();
			int last_level = -1, extra = 0;
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
				if (level[c] != last_level) {
					if (last_level >= 0)
						steps.back().y = (int) brow.size() - ch.brow0;
					steps.push_back(int2{(int) brow.size() - ch.brow0, 0});
					last_level = level[c];
				}
				brow.push_back(uint2{(uint32_t) (c - ch.lo) | ((uint32_t) nc << 16), (uint32_t) ((int) near.size() - ch.near0)});
				for (uint64_t e = dep_rp[c]; e < dep_rp[c + 1]; e++)
					if (dep[e].x < (uint32_t) ch.hi)
						near.push_back(uint2{dep[e].x - (uint32_t) ch.lo, dep[e].y});
			}
			if (last_level >= 0)
				steps.back().y = (int) brow.size() - ch.brow0;
			if ((int) steps.size() - ch.step0 != ch.nsteps || (int) near.size() - ch.near0 != ch.nnear ||
			    (int) brow.size() - ch.brow0 != ch.nbrow)
				die("backsolve_plan: chunk %zu was counted differently on the second pass", k);
			chunk_extra[k] = extra;
		}
	}
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

	B.nchunks = (int) chunks.size();
	B.nnear = (int64_t) near.size();
	B.nfar = (int64_t) far_rp[r];
	B.nnp = (int64_t) np.size();
	B.ndeps = (int64_t) dep.size();
	B.d_col = dalloc<int>(m);
	B.d_chunk = dalloc<BsChunk>((int64_t) chunks.size());
	B.d_step = dalloc<int2>((int64_t) steps.size());
	B.d_brow = dalloc<uint2>((int64_t) brow.size());
	B.d_near = dalloc<uint2>((int64_t) near.size());
	B.d_far_head = dalloc<uint4>(r);
	B.d_far_rp = dalloc<uint64_t>((int64_t) r + 1);
	B.d_far = dalloc<uint2>((int64_t) far.size());
	B.d_np_rp = dalloc<uint64_t>((int64_t) r + 1);
	B.d_np = dalloc<uint2>((int64_t) np.size());
	B.d_chunk_extra = dalloc<int>((int64_t) chunk_extra.size());
	upload(B.d_col, colmap, stream);
	upload(B.d_chunk, chunks, stream);
	upload(B.d_chunk_extra, chunk_extra, stream);
	upload(B.d_step, steps, stream);
	upload(B.d_brow, brow, stream);
	upload(B.d_near, near, stream);
	upload(B.d_far_head, far_head, stream);
	upload(B.d_far_rp, far_rp, stream);
	upload(B.d_far, far, stream);
	upload(B.d_np_rp, np_rp, stream);
	upload(B.d_np, np, stream);
	HIP_CHECK(hipStreamSynchronize(stream));      // the host vectors die here
	B.planned = true;
	B.valid = false;
}

void backsolve_free(spasm_hip_dfact *F)
{
	BsImage &B = F->bs;
	(void) hipFree(B.d_R);
	(void) hipFree(B.d_col);
	(void) hipFree(B.d_chunk);
	(void) hipFree(B.d_chunk_extra);
	(void) hipFree(B.d_step);
	(void) hipFree(B.d_brow);
	(void) hipFree(B.d_near);
	(void) hipFree(B.d_far_head);
	(void) hipFree(B.d_far_rp);
	(void) hipFree(B.d_far);
	(void) hipFree(B.d_np_rp);
	(void) hipFree(B.d_np);
	if (B.ev0 != nullptr)
		(void) hipEventDestroy(B.ev0);
	if (B.ev1 != nullptr)
		(void) hipEventDestroy(B.ev1);
	B = BsImage{};
}

// (re)computes R on `stream`.  The caller synchronises before reading B.ms_build.
void backsolve_build(const spasm_hip_dfact *F, hipStream_t stream)
{
	BsImage &B = F->bs;
	if (!B.planned)
		die("backsolve_build: the factor has no back-substitution plan");
	if (B.d_R == nullptr)
		B.d_R = dalloc<uint32_t>((int64_t) B.r * B.ldR);
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
	b.step = B.d_step;
	b.brow = B.d_brow;
	b.near = B.d_near;
	b.far_head = B.d_far_head;
	b.far_rp = B.d_far_rp;
	b.far = B.d_far;
	b.np_rp = B.d_np_rp;
	b.np = B.d_np;
	b.r = B.r;
	b.F = to_dev(F->mont);
	HIP_CHECK(hipEventRecord(B.ev0, stream));
	HIP_CHECK(hipMemsetAsync(B.d_R, 0, (size_t) B.r * (size_t) B.ldR * 4, stream));
	hipLaunchKernelGGL(bs_init_kernel, dim3((B.r + 255) / 256), dim3(256), 0, stream, b);
	hipLaunchKernelGGL(backsolve_kernel, dim3((unsigned) ((B.Sm + BS_CW - 1) / BS_CW)), dim3(64 * BS_NW), 0, stream, b);     // (padding columns stay zero)
	HIP_CHECK(hipGetLastError());
	HIP_CHECK(hipEventRecord(B.ev1, stream));
	B.valid = true;
	B.builds += 1;
}

// S rows from R: sparse rows into the pool of `a` (dense_out == nullptr) or dense rows.
void launch_backsolve_apply(const SchurArgs &a, const spasm_hip_dfact *F, uint32_t *dense_out, int64_t ldS, hipStream_t stream)
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
	static_assert(64 * AP_TU == 256, "ldR is padded to whole tiles of the apply kernel");
	d.Smpad = (int) B.ldR;
	const size_t per_wave = ((size_t) d.Smpad + 2 * AP_LIST) * 4;
	if (per_wave > 150 * 1024)
		die("launch_backsolve_apply: %d non-pivotal columns do not fit the LDS row buffer", B.Sm);
	// as many waves as fit half of a CU's LDS (two workgroups per CU), at most 8, at least 1
	const int waves = (int) std::max<size_t>(1, std::min<size_t>(8, (size_t) (76 * 1024) / per_wave));
	d.waves = waves;
	d.dense_out = dense_out;
	d.ldS = ldS;
	const size_t lds = per_wave * (size_t) waves;
	static size_t configured = 0;
	if (lds > configured) {
		HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&bs_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
		                              (int) lds));
		configured = lds;
	}
	int dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	hipDeviceProp_t prop;
	HIP_CHECK(hipGetDeviceProperties(&prop, dev));
	const int blocks = std::max(1, std::min((a.nrows + waves - 1) / waves, prop.multiProcessorCount * 8));
	hipLaunchKernelGGL(bs_apply_kernel, dim3(blocks), dim3(64 * waves), lds, stream, d);
	HIP_CHECK(hipGetLastError());
}

}  // namespace sh
