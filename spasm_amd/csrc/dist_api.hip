// Multi-GPU layer of the C ABI: one process per GPU, RCCL over xGMI (include/spasm_hip.h, section (M)).
//
// What shards is what the reference hands to OpenMP threads: the rows of a Schur complement (spasm_schur.c:86-171).
// Every rank holds A and the factor, reduces a contiguous slice of the row list on its GPU, and the slices are
// reassembled ON THE DEVICES with an all-gatherv -- one ncclAllGather for the sizes, then one grouped round of direct
// sends and receives between every pair of ranks with the exact counts (no padding, no host staging).  The host pivot search runs
// once, on rank 0 (it is threaded and timing dependent), and its outcome is broadcast.  With a communicator installed
// (spasm_hip_set_comm) the ordinary entry points -- spasm_hip_pivots_extract_structural, spasm_hip_schur and with them
// spasm_hip_echelonize -- do all this by themselves; every rank must make the same calls in the same order.
#include <rccl/rccl.h>

#include <algorithm>
#include <cinttypes>
#include <cstring>
#include <vector>

#include "device_types.h"

#define NCCL_CHECK(expr)                                                                                     \
	do {                                                                                                 \
		ncclResult_t r_ = (expr);                                                                    \
		if (r_ != ncclSuccess)                                                                       \
			sh::die("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__);   \
	} while (0)

struct spasm_hip_comm {
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1;
	hipStream_t stream = nullptr;       // host-driven exchanges (broadcasts of pivot search results)
	int64_t *d_sizes = nullptr;         // 2 * world
	void *d_stage = nullptr;            // staging buffer of comm_bcast_host
	size_t stage_bytes = 0;
};

namespace sh {

static spasm_hip_comm *g_comm = nullptr;

spasm_hip_comm *current_comm() { return g_comm; }
int comm_rank(const spasm_hip_comm *c) { return c->rank; }
int comm_world(const spasm_hip_comm *c) { return c->world; }

__global__ void rebase_offsets_kernel(int64_t *Sp, int n, int64_t base)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < n)
		Sp[t] += base;
}

// ---- column split: the slabs of a Schur complement, stacked by the all-gatherv, become whole rows ----------------
// Rank k has reduced ALL n rows on its slab of the non-pivotal columns; the all-gatherv stacks the slabs (rows k n .. k n + n
// of the stack are rank k's), and row i of S is the concatenation over k of row k n + i -- the slabs are ranges of columns
// in increasing order, so the row comes out sorted.
__global__ __launch_bounds__(256) void stitch_lengths_kernel(const int64_t *gSp, int n, int parts, int *len, unsigned long long *block_sum)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	int64_t t = 0;
	for (int k = 0; k < parts; k++)
		t += gSp[(int64_t) k * n + i + 1] - gSp[(int64_t) k * n + i];
	len[i] = (int) t;
	atomicAdd(&block_sum[i >> 10], (unsigned long long) t);
}

__global__ __launch_bounds__(256) void stitch_rows_kernel(const int64_t *gSp, const int *gSj, const int *gSx, int n, int parts, const int64_t *Sp, int *Sj, int *Sx,
                                                          int64_t cap)
{
	const int lane = threadIdx.x & 63;
	const int wave = (int) ((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int) ((gridDim.x * blockDim.x) >> 6);
	for (int i = wave; i < n; i += nwaves) {
		int64_t w = Sp[i];
		if (Sp[i + 1] > cap)
			continue;          // (the scan has raised the overflow flag)
		for (int k = 0; k < parts; k++) {
			const int64_t lo = gSp[(int64_t) k * n + i], hi = gSp[(int64_t) k * n + i + 1];
			for (int64_t t = lo + lane; t < hi; t += 64) {
				Sj[w + (t - lo)] = gSj[t];
				Sx[w + (t - lo)] = gSx[t];
			}
			w += hi - lo;
		}
	}
}

// slab-local column numbers -> columns of the matrix
__global__ __launch_bounds__(256) void map_columns_kernel(int *Sj, int64_t nnz, const int *cols)
{
	for (int64_t t = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; t < nnz; t += (int64_t) gridDim.x * blockDim.x)
		Sj[t] = cols[Sj[t]];
}

void launch_map_columns(int *d_Sj, int64_t nnz, const int *d_cols, hipStream_t stream)
{
	if (nnz > 0)
		hipLaunchKernelGGL(map_columns_kernel, dim3((unsigned) std::min<int64_t>((nnz + 255) / 256, 65536)), dim3(256), 0, stream, d_Sj, nnz, d_cols);
}

// the stack of `parts` slabs of n rows each (gSp: parts * n + 1 offsets) -> S (Sp: n + 1; Sj, Sx: gSp[parts * n] entries);
// d_len: n ints, d_block_sum: (n + 1023) / 1024 words of scratch
void launch_stitch_slabs(const int64_t *gSp, const int *gSj, const int *gSx, int n, int parts, int64_t *Sp, int *Sj, int *Sx, int64_t cap, int *d_len,
                         unsigned long long *d_block_sum, int *d_ctr, hipStream_t stream)
{
	if (n <= 0)
		return;
	const int nblocks = (n + 1023) / 1024;
	HIP_CHECK(hipMemsetAsync(d_block_sum, 0, (size_t) nblocks * sizeof(unsigned long long), stream));
	HIP_CHECK(hipMemsetAsync(Sp, 0, sizeof(int64_t), stream));
	hipLaunchKernelGGL(stitch_lengths_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, gSp, n, parts, d_len, d_block_sum);
	launch_scan_lengths(d_len, n, d_block_sum, Sp, cap, d_ctr, stream);
	hipLaunchKernelGGL(stitch_rows_kernel, dim3((unsigned) std::min(n / 4 + 1, 256 * 8)), dim3(256), 0, stream, gSp, gSj, gSx, n, parts, Sp, Sj, Sx, cap);
	HIP_CHECK(hipGetLastError());
}

// broadcast of a host buffer from `root` (staged through device memory: RCCL moves device buffers)
void comm_bcast_host(spasm_hip_comm *c, void *buf, size_t bytes, int root)
{
	if (c == nullptr || c->world == 1 || bytes == 0)
		return;
	if (c->stage_bytes < bytes) {
		if (c->d_stage != nullptr)
			(void) hipFree(c->d_stage);
		HIP_CHECK(sh::malloc_or_trim(&c->d_stage, bytes));
		c->stage_bytes = bytes;
	}
	if (c->rank == root)
		HIP_CHECK(hipMemcpyAsync(c->d_stage, buf, bytes, hipMemcpyHostToDevice, c->stream));
	NCCL_CHECK(ncclBroadcast(c->d_stage, c->d_stage, bytes, ncclUint8, root, c->comm, c->stream));
	if (c->rank != root)
		HIP_CHECK(hipMemcpyAsync(buf, c->d_stage, bytes, hipMemcpyDeviceToHost, c->stream));
	HIP_CHECK(hipStreamSynchronize(c->stream));
}

}  // namespace sh

using namespace sh;

extern "C" {

int spasm_hip_comm_id_bytes(void) { return (int) sizeof(ncclUniqueId); }

void spasm_hip_comm_new_id(void *id)
{
	ncclUniqueId u;
	NCCL_CHECK(ncclGetUniqueId(&u));
	std::memcpy(id, &u, sizeof(u));
}

// collective over the `world` processes that hold the same id; uses the calling thread's current HIP device
spasm_hip_comm *spasm_hip_comm_create(const void *id, int rank, int world)
{
	if (spasm_hip_device_count() == 0)
		die("spasm_hip_comm_create: no HIP device (this library has no CPU path)");
	if (world < 1 || rank < 0 || rank >= world)
		die("spasm_hip_comm_create: rank %d of %d", rank, world);
	spasm_hip_comm *c = new spasm_hip_comm();
	c->rank = rank;
	c->world = world;
	ncclUniqueId u;
	std::memcpy(&u, id, sizeof(u));
	NCCL_CHECK(ncclCommInitRank(&c->comm, world, u, rank));
	HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
	HIP_CHECK(sh::malloc_or_trim((void **) &c->d_sizes, (size_t) 2 * world * sizeof(int64_t)));
	return c;
}

void spasm_hip_comm_destroy(spasm_hip_comm *c)
{
	if (c == nullptr)
		return;
	if (g_comm == c)
		g_comm = nullptr;
	(void) hipFree(c->d_sizes);
	if (c->d_stage != nullptr)
		(void) hipFree(c->d_stage);
	(void) hipStreamDestroy(c->stream);
	(void) ncclCommDestroy(c->comm);
	delete c;
}

int spasm_hip_comm_rank(const spasm_hip_comm *c) { return c->rank; }
int spasm_hip_comm_world(const spasm_hip_comm *c) { return c->world; }

// installs (or, with NULL, removes) the communicator the host-pointer entry points use
void spasm_hip_set_comm(spasm_hip_comm *c) { g_comm = c; }

// [lo, hi) of the contiguous slice of n rows owned by `rank` (sizes differ by at most one)
void spasm_hip_shard(int n, int rank, int world, int *lo, int *hi)
{
	const int base = n / world, extra = n % world;
	*lo = rank * base + std::min(rank, extra);
	*hi = *lo + base + (rank < extra ? 1 : 0);
}

// All-gatherv of the Schur complements the ranks left in their workspaces (last spasm_hip_dschur on W): the rows of
// rank 0, then rank 1, ... as one CSR in caller-provided device buffers (d_Sp: total rows + 1; d_Sj, d_Sx: cap entries).
// Returns 0, or 1 when cap is too small (nothing is moved then; *total_nnz says what is needed).  total_rows / total_nnz
// may be NULL.  Everything is enqueued on `stream`, which is synchronised once for the sizes.
int spasm_hip_dschur_allgatherv(spasm_hip_comm *c, const spasm_hip_dwork *W, i64 *d_Sp, int *d_Sj, spasm_ZZp *d_Sx, i64 cap,
                                int *total_rows, i64 *total_nnz, void *stream_)
{
	return sh::comm_allgatherv_csr(c, W->last_rows, W->last_nnz, W->d_Sp, W->d_Sj, W->d_Sx, d_Sp, d_Sj, d_Sx, cap, total_rows, total_nnz, (hipStream_t) stream_);
}

}  // extern "C"

namespace sh {

// element-wise reductions over the ranks, in place, of device buffers every rank holds: the minimum of int32 words (the leftmost
// column of every row of a Schur complement kept as column slabs) and the sum of uint32 words (dense rows formed slab by slab:
// the slabs are disjoint ranges of columns, so at most one rank holds a non-zero word anywhere and the sum is exact)
void comm_allreduce_min_i32(spasm_hip_comm *c, int *d_buf, int64_t count, hipStream_t stream)
{
	if (c == nullptr || c->world == 1 || count <= 0)
		return;
	NCCL_CHECK(ncclAllReduce(d_buf, d_buf, (size_t) count, ncclInt32, ncclMin, c->comm, stream));
}

void comm_allreduce_sum_u32(spasm_hip_comm *c, uint32_t *d_buf, int64_t count, hipStream_t stream)
{
	if (c == nullptr || c->world == 1 || count <= 0)
		return;
	// (in pieces of 2^30 words: a count is a size_t, but the ring works through it in one go)
	for (int64_t at = 0; at < count; at += (int64_t) 1 << 30)
		NCCL_CHECK(ncclAllReduce(d_buf + at, d_buf + at, (size_t) std::min<int64_t>((int64_t) 1 << 30, count - at), ncclUint32, ncclSum, c->comm, stream));
}

void comm_allreduce_sum_i32(spasm_hip_comm *c, int *d_buf, int64_t count, hipStream_t stream)
{
	if (c == nullptr || c->world == 1 || count <= 0)
		return;
	NCCL_CHECK(ncclAllReduce(d_buf, d_buf, (size_t) count, ncclInt32, ncclSum, c->comm, stream));
}

// the all-gatherv of spasm_hip_dschur_allgatherv on plain device arrays (my_rows rows, my_nnz entries: own_Sp / own_Sj / own_Sx)
int comm_allgatherv_csr(spasm_hip_comm *c, int my_rows, int64_t my_nnz, const int64_t *own_Sp, const int *own_Sj, const int *own_Sx, int64_t *d_Sp, int *d_Sj,
                        int *d_Sx, int64_t cap, int *total_rows, int64_t *total_nnz, hipStream_t stream)
{
	const int world = c->world;
	const int64_t mine[2] = {(int64_t) my_rows, my_nnz};
	HIP_CHECK(hipMemcpyAsync(c->d_sizes + 2 * c->rank, mine, sizeof(mine), hipMemcpyHostToDevice, stream));
	NCCL_CHECK(ncclAllGather(c->d_sizes + 2 * c->rank, c->d_sizes, 2, ncclInt64, c->comm, stream));
	std::vector<int64_t> sizes((size_t) 2 * world);
	HIP_CHECK(hipMemcpyAsync(sizes.data(), c->d_sizes, sizes.size() * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	// who sends what to whom, in which order, at which offsets: spasm_hip_allgatherv_plan (host_dist.cpp, a pure host
	// function that the CPU tests check for worlds of 2 to 8) -- this loop only executes its list
	std::vector<int64_t> row_base((size_t) world + 1, 0), nz_base((size_t) world + 1, 0);
	const int nsteps = spasm_hip_allgatherv_plan(world, c->rank, sizes.data(), nullptr, 0, row_base.data(), nz_base.data());
	if (total_rows != nullptr)
		*total_rows = (int) row_base[world];
	if (total_nnz != nullptr)
		*total_nnz = nz_base[world];
	if (cap < 0 || nz_base[world] > cap)
		return 1;
	std::vector<spasm_hip_xfer> plan((size_t) (nsteps > 0 ? nsteps : 1));
	(void) spasm_hip_allgatherv_plan(world, c->rank, sizes.data(), plan.data(), nsteps, nullptr, nullptr);
	// exact counts, no padding.  xGMI is point-to-point (every GPU has a link to every other one): each rank SENDS its
	// slice straight to every peer and receives theirs, all in one group, so that all the links carry payload at once
	// (a ring all-gather or a broadcast tree would push the whole of S through single links).  The local slice is a
	// device-to-device copy.  SPASM_HIP_ALLGATHERV=bcast: one ncclBroadcast per rank and array instead.
	const char *how = sh::env_get("SPASM_HIP_ALLGATHERV");
	const bool use_bcast = how != nullptr && std::strcmp(how, "bcast") == 0;
	const void *own[3] = {own_Sp, own_Sj, own_Sx};
	void *all[3] = {d_Sp, d_Sj, d_Sx};
	const size_t width[3] = {sizeof(int64_t), sizeof(int), sizeof(int)};
	const ncclDataType_t type[3] = {ncclInt64, ncclInt32, ncclInt32};
	NCCL_CHECK(ncclGroupStart());
	if (use_bcast) {
		for (int r = 0; r < world; r++) {
			const int64_t nr = sizes[2 * r], nz = sizes[2 * r + 1];
			if (nr > 0)
				NCCL_CHECK(ncclBroadcast(own_Sp, d_Sp + row_base[r], (size_t) nr, ncclInt64, r, c->comm, stream));
			if (nz > 0) {
				NCCL_CHECK(ncclBroadcast(own_Sj, d_Sj + nz_base[r], (size_t) nz, ncclInt32, r, c->comm, stream));
				NCCL_CHECK(ncclBroadcast(own_Sx, d_Sx + nz_base[r], (size_t) nz, ncclInt32, r, c->comm, stream));
			}
		}
	} else {
		for (const spasm_hip_xfer &x : plan) {
			if (x.kind == SPASM_HIP_XFER_SEND)
				NCCL_CHECK(ncclSend((const char *) own[x.array] + (size_t) x.src * width[x.array], (size_t) x.count, type[x.array], x.peer, c->comm, stream));
			else if (x.kind == SPASM_HIP_XFER_RECV)
				NCCL_CHECK(ncclRecv((char *) all[x.array] + (size_t) x.dst * width[x.array], (size_t) x.count, type[x.array], x.peer, c->comm, stream));
		}
	}
	NCCL_CHECK(ncclGroupEnd());
	if (!use_bcast)
		for (const spasm_hip_xfer &x : plan)
			if (x.kind == SPASM_HIP_XFER_COPY)
				HIP_CHECK(hipMemcpyAsync((char *) all[x.array] + (size_t) x.dst * width[x.array], (const char *) own[x.array] + (size_t) x.src * width[x.array],
				                         (size_t) x.count * width[x.array], hipMemcpyDeviceToDevice, stream));
	// the row pointers of a slice start at 0: shift them to where the slice went
	for (int r = 0; r < world; r++) {
		const int nr = (int) sizes[2 * r];
		if (nr > 0 && nz_base[r] != 0)
			hipLaunchKernelGGL(rebase_offsets_kernel, dim3((nr + 255) / 256), dim3(256), 0, stream, d_Sp + row_base[r], nr, nz_base[r]);
	}
	HIP_CHECK(hipMemcpyAsync(d_Sp + row_base[world], &nz_base[world], sizeof(int64_t), hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipStreamSynchronize(stream));        // (nz_base dies here)
	return 0;
}

}  // namespace sh

extern "C" {

// Test hook (and building block of the column split): `parts` slabs of n rows each, stacked in device arrays (gSp: parts * n + 1
// offsets into gSj / gSx), stitched into whole rows -- row i = slab 0's row i, then slab 1's, ... -- in d_Sp (n + 1) / d_Sj /
// d_Sx (cap entries).  Returns 0, or 1 when cap is too small.  Everything on `stream`, synchronised before returning.
int spasm_hip_dstitch_slabs(const int64_t *d_gSp, const int *d_gSj, const spasm_ZZp *d_gSx, int n, int parts, int64_t *d_Sp, int *d_Sj, spasm_ZZp *d_Sx,
                            int64_t cap, void *stream_)
{
	hipStream_t stream = (hipStream_t) stream_;
	int *d_len = static_cast<int *>(sh::big_alloc((size_t) (n > 0 ? n : 1) * sizeof(int)));
	unsigned long long *d_bs = static_cast<unsigned long long *>(sh::big_alloc((size_t) ((n + 1023) / 1024 + 1) * sizeof(unsigned long long)));
	int *d_ctr = static_cast<int *>(sh::big_alloc((size_t) sh::CTR_COUNT * sizeof(int)));
	HIP_CHECK(hipMemsetAsync(d_ctr, 0, (size_t) sh::CTR_COUNT * sizeof(int), stream));
	sh::launch_stitch_slabs(d_gSp, d_gSj, d_gSx, n, parts, d_Sp, d_Sj, d_Sx, cap, d_len, d_bs, d_ctr, stream);
	int status = 0;
	HIP_CHECK(hipMemcpyAsync(&status, d_ctr + sh::CTR_STATUS, sizeof(int), hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	sh::big_free(d_len);
	sh::big_free(d_bs);
	sh::big_free(d_ctr);
	return status & 1;
}

// spasm_echelonize with the Schur complements of every round sharded over the ranks of `c` (one process per GPU).
// Collective: every rank passes the same A and options and gets the same factorization: the pivots come from rank 0 (broadcast),
// the Schur complements are gathered on every rank, and the random combinations of the dense finish are drawn from a
// generator keyed by what is combined (shape of the block, rows left, round: dense_api.hip) -- not by how many calls the
// process has made --, so that the replicated finish takes the same pivots on every rank.
struct spasm_lu *spasm_hip_echelonize_dist(const struct spasm_csr *A, struct echelonize_opts *opts, spasm_hip_comm *c)
{
	spasm_hip_comm *saved = g_comm;
	g_comm = c;
	struct spasm_lu *fact = spasm_hip_echelonize(A, opts);
	g_comm = saved;
	return fact;
}

}  // extern "C"
