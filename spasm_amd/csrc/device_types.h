// Device-side data structures shared by the kernels and their launchers.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <vector>

#include "common.h"

namespace sh {

#define HIP_CHECK(expr)                                                                                  \
	do {                                                                                             \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess)                                                                    \
			sh::die("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

struct MontDev {           // copy of sh::Mont for kernels
	uint32_t p, pinv, r1, r2, half;
};

inline MontDev to_dev(const Mont &M) { return MontDev{M.p, M.pinv, M.r1, M.r2, M.half}; }

// counters living in one small device array (spasm_hip_dwork::d_ctr)
enum Ctr {
	CTR_ROW_NEXT = 0,      // next row batch to hand out (tier 0)
	CTR_ROW_NEXT2,         // ... tier 1 (large LDS table)
	CTR_ROW_NEXT3,         // ... tier 2 (dense accumulator)
	CTR_ROW_NEXT_G,        // ... row-group kernel (groups)
	CTR_OVF1,              // rows that overflowed the small table
	CTR_OVF2,              // rows that overflowed the large table
	CTR_STATUS,            // bit 0: pool exhausted
	CTR_DONE0, CTR_DONE1, CTR_DONE2,   // rows finished per tier
	CTR_GROUP_ABORT,       // row-group kernel gave up (poor lane efficiency)
	CTR_GROUPS_DONE,
	CTR_COUNT = 16
};

enum Ctr64 {
	C64_POOL = 0,          // pool cursor (entries)
	C64_ELIM,              // pivot rows applied
	C64_STREAM,            // entries of U' streamed
	C64_INPUT,             // entries of input rows
	C64_WAVEPIV,           // row-group kernel: pivots applied per group (wave-level count)
	C64_LPOOL,             // cursor of the L pool
	C64_PROF0 = 8,         // -DSPASM_GROUP_PROFILE builds: cycles per phase of the row-group kernel (8 slots)
	C64_COUNT = 16
};

struct SchurArgs {
	// rows of A to reduce
	const int64_t *Ap;
	const int *Aj;
	const int *Ax;
	const int *rows;          // nrows row indices into A
	int nrows;
	// optional explicit work list (positions into rows[]); count read on device
	const int *list;
	const int *list_count;
	// factor
	const uint32_t *lab;      // column -> label
	const int *q;             // label - r -> column
	const uint64_t *rp;       // r + 1 offsets into ent
	const uint2 *ent;         // (label, value * R mod p)
	const uint2 *head;        // label * 4: first four entries of the row (0xFFFFFFFF-padded)
	const uint32_t *lvl_end;  // label -> end of its level
	const uint32_t *lvl_end_w;// 32-label word -> first word of the next level
	int r;                    // size of the (padded) pivot label space; labels >= r are non-pivotal
	int Sm;                   // number of non-pivotal columns
	int m;
	MontDev F;
	// output pool
	int *pool_j;
	int *pool_x;
	int64_t pool_cap;
	int64_t *row_off;
	int *row_len;             // -1: not produced (pool full)
	int *ovf_list;            // rows handed to the next tier
	int *ctr;
	unsigned long long *ctr64;
	int next_ctr;             // which CTR_ROW_NEXT* this launch uses
	int ovf_ctr;              // which CTR_OVF* receives overflowing rows
	int done_ctr;
	int skip_done;            // tier 0 only: leave rows alone whose row_len is no longer -1
	int skip_ctr;             // ... and do nothing at all unless ctr[skip_ctr] says the row-group kernel gave up
	const uint32_t *comp;     // label -> component of the pivot graph (rows of different components share nothing)
	const int *order;         // row-group kernel: list positions in processing order (null: 0, 1, 2, ...)
	// optional record of the elimination coefficients (the L factor): triplets (row, pivot index, value)
	int *L_i;                 // null: not recorded.  Pool pre-filled with -1; unused slots stay -1
	int *L_j;
	int *L_x;
	int64_t L_cap;
	const int *kof;           // label -> row of U
	const int *row_orig;      // output row k -> row index to record (p_in[p[k]] or p[k])
	int avg_row_entries;      // host hint: entries of A per reduced row (0: unknown)
};

// device pools receiving the elimination coefficients (triplets) of a Schur call
struct LOut {
	const int *row_orig = nullptr;
	int *Li = nullptr, *Lj = nullptr, *Lx = nullptr;
	int64_t cap = 0;
	int64_t used = 0;         // out: pool cursor after the call (slots handed out, some may be unused)
};

// Host part of the factor image (built by plan_factor, schur_api.hip).
struct FactPlan {
	int m = 0, r = 0, nlevels = 0;
	int rpad = 0;                 // size of the label space of the pivots: every level starts on a multiple of 32
	int maxdeg = 0;               // largest number of rows of U' that hold one given label
	int64_t ndeps = 0;            // entries of U' on pivotal columns (the dependencies between the rows)
	int ncomp = 0, comp_largest = 0;   // connected components of the pivot graph (labels that hold a row), size of the largest
	int64_t prime = 0;
	std::vector<uint32_t> lab, lvl_end;     // lvl_end is indexed by (padded) label
	std::vector<uint32_t> lvl_end_w;        // per 32-label word: first word of the next level
	std::vector<int> q, kof, label_of_row;  // kof: label -> row of U, -1 for padding labels
	std::vector<int> lvl_count;             // rows per level
	std::vector<uint64_t> rp;               // rpad + 1
	std::vector<uint2> ent;
	std::vector<uint2> head;                // 4 entries per label: the first entries of the row, 0xFFFFFFFF-padded
	std::vector<uint32_t> comp;             // per label: smallest label of its connected component in the pivot graph
	std::vector<int> lvl_first;             // first label of every level (levels of >= 32 rows start on a multiple of 32)
};

// ---- back-substituted factor ("R image", backsolve.hip) ------------------------------------------
// R = U_pp^-1 U_pn: row c = the non-pivotal part of the fully reduced pivot row c (what spasm_rref would
// give).  With it the Schur complement of a row a is a_n - a_p R: no elimination order, no atomics.
// Rows are numbered by COMPACT id = rank of their label among the labels that hold a row (so level order).
struct BsChunk {              // a run of consecutive compact rows solved inside LDS by one workgroup per column slab
	int lo, hi;               // compact rows [lo, hi)
	int pass0, npass;         // phase-B passes of 32 rows (rows of one level that have dependencies inside the chunk)
	int near0, nnear;         // dependencies of the rows that have more than two of them inside the chunk
	int np0;                  // first non-pivotal entry (index into d_np) of the rows of the chunk ...
	int npn;                  // ... their number (bits 0-30); bit 31: some row has more than two dependencies outside the chunk
};

struct BsImage {
	bool planned = false;     // the host plan below was built and uploaded (the factor is eligible)
	bool valid = false;       // d_R holds R
	int r = 0, Sm = 0, nchunks = 0;
	int64_t ldR = 0;          // Sm rounded up to 256
	int64_t nfar = 0, nnear = 0, nnp = 0, ndeps = 0;
	void *d_R = nullptr;              // r x ldR entries of elem_bytes each (uint16_t when p < 2^16, else uint32_t)
	int elem_bytes = 4;
	bool plain = false;               // coefficients of the plan are plain residues (p < 2^16) instead of Montgomery form
	bool sgn = false;                 // ... negated balanced residues, and R holds signed 16-bit entries (small p: backsolve.hip, SgnDev)
	int *d_col = nullptr;             // column -> compact row id of its pivot, or r + index among the non-pivotal columns
	BsChunk *d_chunk = nullptr;
	uint4 *d_ptab = nullptr;          // 32 entries per pass: (slot | count << 16, dep0 | dep1 << 16, coefficient 0, coefficient 1); a row
	                                  // with more than two dependencies keeps the first inline, (offset into d_near - near0) in .w, the others there
	uint2 *d_near = nullptr;          // (slot of the dependency, coefficient)
	uint4 *d_far_head = nullptr;      // per compact row: its first two dependencies outside the chunk (t0, y0, t1, y1); t = ~0: none
	uint64_t *d_far_rp = nullptr;     // per compact row: the others, [r + 1] offsets into d_far
	uint2 *d_far = nullptr;
	uint64_t *d_np_rp = nullptr;      // per compact row: non-pivotal entries (index among the non-pivotal columns, value * 2^32 mod p)
	uint2 *d_np = nullptr;
	int *d_np_row = nullptr;          // compact row of every entry of d_np
	int *d_chunk_extra = nullptr;     // per chunk: some row has more than two dependencies outside the chunk
	hipEvent_t ev0 = nullptr, ev1 = nullptr;     // around the last build (memset + init + backsolve kernel)
	int builds = 0;
	double density_hint = -1.0;       // expected density of the Schur complements of this factor (< 0: unknown)
	double elim_hint = -1.0;          // (row, pivot) eliminations per reduced row, as the row-by-row kernels measured them on a sample (< 0: unknown)
	char kernel_build[64] = "backsolve_kernel";      // variant launched by the last build, as rocprofv3 prints it
	int shape = 2;                    // workgroup shape of the build kernel the plan was cut for (backsolve_plan)
	int ring = 768, passrows = 32, passcap = 80;      // rows per chunk, rows per phase-B pass, passes per chunk of that plan
};

// ---- sparse back-substituted factor ("sparse image", sparse_image.hip) ----------------------------
// The same R = U_pp^-1 U_pn as BsImage, for factors whose R is mostly zeros (a Schur complement that stays sparse): rows
// of R are kept as FRAGMENTS -- the entries a row has in one segment of SP_SEG non-pivotal columns, 4 bytes each
// (column inside the segment | signed 16-bit value << 16) -- in a bump-allocated pool; frag[c * nseg + g] says where the
// fragment of row c (compact id) in segment g lies and how long it is.  Bytes and work scale with the fill of R, not with
// r x Sm, and there is no limit on Sm.
#ifndef SPASM_SP_SEG
#define SPASM_SP_SEG 4096
#endif
constexpr int SP_SEG = SPASM_SP_SEG;    // columns per segment: 16 KB of 16-bit accumulators per wave (4,096: 8 KB)
constexpr int SP_LEN_BITS = 14;         // a fragment holds at most SP_SEG entries
constexpr int SP_OFF_BITS = 40;         // offset inside a pool chunk (entries)
constexpr int SP_MAX_CHUNKS = 12;
constexpr int SP_SHARDS = 256;          // allocation cursors of the build (one 128-byte line each)

struct SpPools {
	const uint32_t *base[SP_MAX_CHUNKS];
};

struct SpPending;          // (sparse_image.hip: the tables while a host thread is making them)
struct SpImage {
	std::shared_ptr<SpPending> pending;          // tables on their way (sparse_image_plan_start); sparse_image_planned() waits for them
	int pending_m = 0;
	bool planned = false;     // the dependency tables below are on the device
	bool valid = false;       // the fragments hold R
	bool failed = false;      // the last build gave up (R is not sparse: the pool budget ran out)
	bool wide = false;        // 8-byte entries (column, plain residue) and 32-bit accumulators: primes beyond the signed 16-bit arithmetic
	int r = 0, Sm = 0, nseg = 0, nlevels = 0;
	int64_t ndeps = 0, nnp = 0;
	std::vector<int> lvl_lo;          // first compact row of every level (nlevels + 1 entries)
	int *d_col = nullptr;             // column -> compact row id of its pivot, or r + index among the non-pivotal columns
	uint64_t *d_dep_rp = nullptr;     // per compact row: its pivotal entries ...
	uint2 *d_dep = nullptr;           // ... (compact row of the pivot, NEGATED balanced coefficient; wide: Montgomery form of the negated coefficient)
	uint64_t *d_np_rp = nullptr;      // per compact row: its non-pivotal entries ...
	uint2 *d_np = nullptr;            // ... (index among the non-pivotal columns, balanced value; wide: plain residue)
	uint64_t *d_segmask = nullptr;    // per compact row: the segments in which its row of R CAN have entries (its own non-pivotal entries
	                                  // and those of the rows it depends on: structural, an upper bound) -- nullptr beyond 64 segments
	uint2 *d_head = nullptr;          // per compact row, 8 words: both lists of a row of at most seven entries in one 64-byte line
	uint64_t *d_rowmask = nullptr;    // per compact row, (nseg + 63) / 64 words: the segments in which its row of R HAS entries (from the
	                                  // fragment words, after every build): sp_apply_kernel visits no other segment of a row
	uint64_t *d_frag = nullptr;       // r * nseg words: chunk << 54 | offset << 14 | length
	uint32_t *d_chunk[SP_MAX_CHUNKS] = {};
	int64_t chunk_cap[SP_MAX_CHUNKS] = {};        // entries (4 bytes each; wide: 8)
	int nchunks = 0;
	unsigned long long *d_shard = nullptr;        // SP_SHARDS x 16 words: [0] cursor, [1] limit (entries inside the current chunk)
	unsigned long long *d_stat = nullptr;         // 8 counters of the build (see sparse_image.hip)
	hipEvent_t ev0 = nullptr, ev1 = nullptr;      // around the last build
	int builds = 0;
	int64_t nnz = 0;                  // entries of R (last build)
	int64_t ops_build = 0;            // (entry of a fragment, row that uses it) pairs of the last build
	int64_t pool_used = 0;            // entries handed out by the cursors
	int launches = 0;                 // kernels of the last build (levels, + those redone after a pool extension)
};

struct SchurArgs;
struct FactPlan;
void launch_map_columns(int *d_Sj, int64_t nnz, const int *d_cols, hipStream_t stream);
// dist_api.hip: collectives on plain device arrays (no-ops in a world of one)
void comm_allreduce_min_i32(struct ::spasm_hip_comm *c, int *d_buf, int64_t count, hipStream_t stream);
void comm_allreduce_sum_i32(struct ::spasm_hip_comm *c, int *d_buf, int64_t count, hipStream_t stream);
void comm_allreduce_sum_u32(struct ::spasm_hip_comm *c, uint32_t *d_buf, int64_t count, hipStream_t stream);
int comm_allgatherv_csr(struct ::spasm_hip_comm *c, int my_rows, int64_t my_nnz, const int64_t *own_Sp, const int *own_Sj, const int *own_Sx, int64_t *d_Sp, int *d_Sj,
                        int *d_Sx, int64_t cap, int *total_rows, int64_t *total_nnz, hipStream_t stream);
void launch_stitch_slabs(const int64_t *gSp, const int *gSj, const int *gSx, int n, int parts, int64_t *Sp, int *Sj, int *Sx, int64_t cap, int *d_len,
                         unsigned long long *d_block_sum, int *d_ctr, hipStream_t stream);
void launch_scan_lengths(const int *len, int n, const unsigned long long *block_sum, int64_t *Sp, int64_t cap, int *ctr, hipStream_t stream);

// device copy of a host CSR matrix for the duration of a call (schur_api.hip: resident between the calls of a driver)
struct DeviceMatrix {
	int64_t *p = nullptr;
	int *j = nullptr, *x = nullptr;
	int64_t nnz = 0;
	bool owned = false;
	// slab_will_do: a matrix that is resident as column slabs (schur_api.hip, ResidentEntry) is handed out as it is -- this rank's
	// slab; nnz = its entries -- instead of being gathered into whole rows first
	DeviceMatrix(const struct spasm_csr *A, hipStream_t stream, bool slab_will_do = false);
	~DeviceMatrix();
	DeviceMatrix(const DeviceMatrix &) = delete;
	DeviceMatrix &operator=(const DeviceMatrix &) = delete;
};
// Large device buffers of the host-level entry points (row pools, dense accumulators, stacks of echelon rows) come from a
// small cache of freed blocks: hipMalloc of tens of GB costs up to 40 ms per GB on these boxes (tools/probe_alloc.py) and
// the driver asks for the same sizes round after round.  big_free takes any device pointer (blocks it does not know go to
// hipFree); big_trim releases everything cached (the driver calls it on exit).
void *big_alloc(size_t bytes);
void big_free(void *ptr);
void big_trim(size_t keep_bytes = 0);
// host -> device through pinned staging buffers of the library (schur_api.hip): the source is consumed when the call returns
void h2d(void *dst, const void *src, size_t bytes, hipStream_t stream);
// device -> host the same way; returns when dst holds the bytes (everything queued on the stream before it has run)
void d2h(void *dst, const void *src, size_t bytes, hipStream_t stream);
void big_age(int max_idle);          // cached blocks unused through more than max_idle driver calls go back to the device
hipError_t malloc_or_trim(void **ptr, size_t bytes);          // hipMalloc; on failure the cache is emptied and it is tried again
void mem_info(size_t *free_b, size_t *total_b);                // hipMemGetInfo + what the cache parks (given back on demand)
void resident_adopt_slab(const struct spasm_csr *host, int64_t *dp, int *dj, int *dx, int64_t local_nnz, struct ::spasm_hip_comm *comm);
struct ::spasm_hip_comm *resident_slab_comm(const struct spasm_csr *A);
void resident_begin();
void resident_end();
void resident_forget(const struct spasm_csr *A);
bool resident_enabled();
void resident_prefetch(const struct spasm_csr *A);
void ensure_row_tables(const struct ::spasm_hip_dfact *F, hipStream_t stream, bool components = true);
void resident_counters(int64_t *uploads, int64_t *hits);
void resident_lazy_downloads(bool on);
void resident_materialize(const struct spasm_csr *A);
int resident_fl_census(const struct spasm_csr *A);

// where bs_apply_kernel writes a sparse result directly in its final place (rows in order, offsets by look-back)
struct BsDirectOut {
	unsigned long long *status;   // nrows words, zeroed before the launch
	int *ticket;                  // ticket counters (backsolve.hip: next_ticket), zeroed before the launch
	int64_t *Sp;                  // nrows + 1
	int *Sj, *Sx;
	int64_t cap;                  // capacity of Sj / Sx
	// staged output (backsolve.hip): room for stage_rows rows of R's width; nullptr: look-back output
	uint32_t *stage = nullptr;
	int64_t stage_rows = 0;
	hipEvent_t ev_expand = nullptr;   // recorded before the (last) expansion kernel
	bool staged = false;              // set by the launch: the staged output ran
	int slices = 0;
};

}  // namespace sh

// opaque handles of the C ABI
struct spasm_hip_dfact {
	int m = 0, r = 0, Sm = 0, nlevels = 0;
	int rpad = 0;
	// (filled by ensure_row_tables when a row-by-row path first runs)
	mutable bool row_tables = false;
	mutable int maxdeg = 0;
	mutable int ncomp = 0, comp_largest = 0;   // connected components of the pivot graph, rows of the largest one
	int64_t nnz = 0;
	int64_t prime = 0;
	sh::Mont mont{};
	uint32_t *d_lab = nullptr;
	int *d_q = nullptr;
	uint64_t *d_rp = nullptr;
	uint2 *d_ent = nullptr;
	uint2 *d_head = nullptr;
	mutable uint32_t *d_comp = nullptr;     // label -> smallest label of its connected component in the pivot graph (ensure_row_tables)
	uint32_t *d_lvl_end = nullptr;
	uint32_t *d_lvl_end_w = nullptr;
	int *d_kof = nullptr;          // label -> row of U (-1: padding label)
	uint64_t *d_cp = nullptr;      // U' by TARGET label (schur_pull.hip): rpad + Sm + 1 offsets into d_cent
	uint2 *d_cent = nullptr;       // (source label, value * 2^32 mod p)
	int2 *d_lvl = nullptr;         // [first label, last label + 1) of every level
	bool has_pull = false;         // d_cp / d_cent / d_lvl were filled (SPASM_HIP_PULL=1 when the image was created)
	std::vector<int> h_q;          // host copy of q
	std::vector<int> h_kof;
	mutable sh::BsImage bs;        // back-substituted image, built on first use when the factor is eligible
	mutable sh::SpImage sp;        // sparse back-substituted image (sparse_image.hip)
	// a factor that has the tables of the sparse image plans its dense image only when a batch asks for it (the plan is 40 % of
	// the image's cost and such factors seldom take the dense image): the host part of the image waits here until then
	mutable bool bs_deferred = false;
	mutable std::unique_ptr<sh::FactPlan> host_plan;
};

struct spasm_hip_dwork {
	int max_rows = 0, m = 0;
	int64_t pool_cap = 0;
	int *d_pool_j = nullptr, *d_pool_x = nullptr;
	int64_t *d_row_off = nullptr;
	int *d_row_len = nullptr;
	int *d_ovf1 = nullptr, *d_ovf2 = nullptr;
	int *d_ctr = nullptr;
	unsigned long long *d_ctr64 = nullptr;
	int64_t *d_Sp = nullptr;
	int64_t *d_blocksum = nullptr;
	int *d_order = nullptr, *d_sortbuf = nullptr;   // rows regrouped by connected component of the pivot graph
	int64_t sortbuf_ints = 0;
	int *d_Sj = nullptr, *d_Sx = nullptr;
	uint32_t *d_stage = nullptr;                  // packed rows of the staged sparse output (backsolve.hip)
	int64_t stage_bytes = 0;
	uint64_t *d_spT = nullptr;                    // sparse image: where the fragment of every (row, segment) of S lies (sparse_image.hip)
	int64_t spT_words = 0;
	unsigned long long *d_lb_status = nullptr;   // look-back words of the direct sparse output (one per row, then the ticket counters: schur_api.hip)
	unsigned char *d_scratch = nullptr;   // per-wave dense accumulators (all zero between calls)
	int64_t scratch_bytes = 0;
	int64_t scratch_budget = 0;           // 0: up to half of the free HBM; else a cap in bytes
	int scratch_slots = 0;
	int64_t slot_bytes = 0, off_bm = 0, off_xn = 0;
	hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	int last_rows = 0;
	int64_t last_nnz = 0;
};
