/*
 * spasm_hip.h -- C ABI of the MI355X (gfx950) echelonization hot path.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no C++ or
 * torch types.  Two layers are exported by spasm_amd/csrc/libspasm_hip.so:
 *
 *  (H) host-pointer entry points with the reference's own signatures.  Each
 *      one replaces the reference function named in its comment (file:line in
 *      cbouilla/spasm) and takes/returns the reference's structs
 *      (struct spasm_csr, struct spasm_lu, ...), allocated with malloc so the
 *      reference's spasm_csr_free()/spasm_lu_free() can release them.
 *
 *  (D) device-pointer entry points (spasm_hip_d*) for callers that keep the
 *      matrices resident in HBM between rounds (bench.py, the multi-GPU
 *      driver).  They take device pointers + a hipStream_t passed as void*.
 *
 * If the reference's spasm.h was included first (_SPASM_H defined) its struct
 * definitions are used; otherwise layout-identical ones are declared here.
 *
 * Error behaviour: like the reference (spasm_util.c:62-83, err()/errx()), a
 * fatal condition -- no usable GPU, out of device memory, malformed input --
 * prints "[spasm-hip] ..." on stderr and exits.  There is NO CPU fallback.
 *
 * Environment.  These switches are supported (read at every call -- SPASM_HIP_SPARSE_IMAGE and SPASM_HIP_BACKSOLVE also decide what
 * the image of a factor plans when it is created: a cached image keeps the setting it was created under):
 *   SPASM_HIP_VERBOSE=0..3        progress messages on stderr (default 1: what the reference prints)
 *   SPASM_HIP_THREADS=n           host threads of the pivot search and the planning (default: the CPU quota of the cgroup; 1: a
 *                                 sequential search on the host -- a pivot set that does not depend on timing)
 *   SPASM_HIP_PIVOT_SEARCH=host|device   where the greedy cycle-free search runs (default: the device when there is one)
 *   SPASM_HIP_PIVOT_LABELS=0|1    greedy search without / with depth labels (default 1; 0 with SPASM_HIP_THREADS=1: the reference's
 *                                 single-thread pivot set, whatever the size of the matrix)
 *   SPASM_HIP_SPARSE_IMAGE=0|1    never / always take the sparse image R = U_pp^-1 U_pn for a Schur complement (default: cost model)
 *   SPASM_HIP_BACKSOLVE=0|1       never / always take the dense image (default: cost model); both 0: row-by-row kernels
 *   SPASM_HIP_DEVICE_FINISH=0|1|2 dense / low-rank finish in the host loops / on the device / on the device only with the dense image
 *   SPASM_HIP_KEEP_GB=n           device memory the block cache keeps between calls (default: a third of the HBM, at most 96 GB; 0: none;
 *                                 a cached block that two driver calls in a row did not use goes back whatever the total)
 *   SPASM_HIP_SCRATCH_GB=n        accumulator slices of the row-by-row kernels (default: up to half of the free HBM)
 *   SPASM_HIP_STAGE_GB=n          staging buffer of the dense image's output (default 8)
 *   SPASM_HIP_SPARSE_IMAGE_GB=n   cap of the fragment pool of the sparse image (default: a third of the free HBM, at most half
 *                                 the bytes of the dense form)
 * Every other SPASM_HIP_* name in the sources picks a kernel variant, a debugging aid or a code path kept for A/B runs and
 * tests; they are ignored (with one warning) unless SPASM_HIP_EXPERIMENT=1 is set as well.
 */
#ifndef SPASM_HIP_H
#define SPASM_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef _SPASM_H
/* ---- layout-compatible restatement of the reference's public types ---- */
typedef uint8_t u8;
typedef int64_t i64;
typedef uint64_t u64;
typedef uint32_t u32;
typedef int32_t i32;

typedef i32 spasm_ZZp;                 /* balanced representative mod p (spasm.h:27) */

struct spasm_field_struct {            /* spasm.h:29-35 */
	i64 p;
	i64 halfp;
	i64 mhalfp;
	double dinvp;
};
typedef struct spasm_field_struct spasm_field[1];

struct spasm_csr {                     /* spasm.h:37-50; n rows, m columns */
	i64 nzmax;
	int n;
	int m;
	i64 *p;
	int *j;
	spasm_ZZp *x;
	spasm_field field;
};

struct spasm_triplet {                 /* spasm.h:52-61 */
	i64 nzmax;
	i64 nz;
	int n;
	int m;
	int *i;
	int *j;
	spasm_ZZp *x;
	spasm_field field;
};

struct spasm_lu {                      /* spasm.h:63-71 */
	int r;
	bool complete;
	struct spasm_csr *L;
	struct spasm_csr *U;
	int *qinv;
	int *p;
	struct spasm_triplet *Ltmp;
};

struct echelonize_opts {               /* spasm.h:84-108 */
	bool enable_greedy_pivot_search;
	bool enable_tall_and_skinny;
	bool enable_dense;
	bool enable_GPLU;
	bool L;
	bool complete;
	double min_pivot_proportion;
	int max_round;
	double sparsity_threshold;
	int dense_block_size;
	double low_rank_ratio;
	double tall_and_skinny_ratio;
	double low_rank_start_weight;
};

typedef enum {SPASM_DOUBLE, SPASM_FLOAT, SPASM_I64} spasm_datatype;   /* spasm.h:139 */
#endif /* _SPASM_H */

/* ======================================================================
 * (H) host-pointer entry points, reference signatures
 * ====================================================================== */

/* library / device probe.  Returns the number of usable gfx950 devices (0 if none). */
int spasm_hip_device_count(void);
/* CPUs this process may really use: hardware threads cut down to the CPU quota of its control group (the pivot search
 * takes that many threads; a box of this pool reports 256 hardware threads and grants 16 CPUs) */
int spasm_hip_usable_cpus(void);
/* Between host-level calls the library parks device memory it would otherwise allocate again -- the accumulator scratch of
 * the row-by-row kernels, and the blocks of its buffer cache up to SPASM_HIP_KEEP_GB (default: a third of the device memory, and nothing that two calls in a row did not use; 0 = none:
 * multi-GB blocks take 0.1 to 1 s apiece to free and allocate again, erratically).  This gives all of it back. */
void spasm_hip_release_cached_memory(void);
const char *spasm_hip_version(void);

/* --- containers and field (replace spasm_util.c:85-191, spasm_ZZp.c) ---
 * Moduli: the host-side field functions and containers take any 2 <= p <= 0xfffffffb, as the reference does
 * (spasm_ZZp.c:5-15).  The GPU entry points (everything that eliminates: spasm_hip_schur*, _echelonize, _rref, _kernel,
 * _ffpack_*, the spasm_hip_d* layer) take ODD p only and die with "modulus ... unsupported on the GPU path" otherwise:
 * their arithmetic is Montgomery form mod 2^32 (and signed 16-bit / Barrett forms for small p), which needs p coprime
 * to 2.  p = 2 -- which the reference accepts -- is therefore REFUSED, not emulated. */
void spasm_hip_field_init(i64 p, spasm_field F);                                   /* spasm_ZZp.c:5-15 */
spasm_ZZp spasm_hip_ZZp_init(const spasm_field F, i64 x);                          /* spasm_ZZp.c:26 */
spasm_ZZp spasm_hip_ZZp_add(const spasm_field F, spasm_ZZp a, spasm_ZZp b);        /* spasm_ZZp.c:32 */
spasm_ZZp spasm_hip_ZZp_sub(const spasm_field F, spasm_ZZp a, spasm_ZZp b);        /* spasm_ZZp.c:37 */
spasm_ZZp spasm_hip_ZZp_mul(const spasm_field F, spasm_ZZp a, spasm_ZZp b);        /* spasm_ZZp.c:42 */
spasm_ZZp spasm_hip_ZZp_inverse(const spasm_field F, spasm_ZZp a);                 /* spasm_ZZp.c:67 */
spasm_ZZp spasm_hip_ZZp_axpy(const spasm_field F, spasm_ZZp a, spasm_ZZp x, spasm_ZZp y);  /* spasm_ZZp.c:76 */

void *spasm_hip_malloc(i64 size);                                                  /* spasm_util.c:65 */
void *spasm_hip_calloc(i64 count, i64 size);                                       /* spasm_util.c:73 */
void *spasm_hip_realloc(void *ptr, i64 size);                                      /* spasm_util.c:81 */
i64 spasm_hip_nnz(const struct spasm_csr *A);                                      /* spasm_util.c:16 */
double spasm_hip_wtime(void);                                                      /* spasm_util.c:9 */
struct spasm_csr *spasm_hip_csr_alloc(int n, int m, i64 nzmax, i64 prime, bool with_values);   /* spasm_util.c:86 */
void spasm_hip_csr_realloc(struct spasm_csr *A, i64 nzmax);                        /* spasm_util.c:122 */
void spasm_hip_csr_resize(struct spasm_csr *A, int n, int m);                      /* spasm_util.c:177 */
void spasm_hip_csr_free(struct spasm_csr *A);                                      /* spasm_util.c:155 */
struct spasm_triplet *spasm_hip_triplet_alloc(int n, int m, i64 nzmax, i64 prime, bool with_values); /* spasm_util.c:102 */
void spasm_hip_triplet_realloc(struct spasm_triplet *T, i64 nzmax);                /* spasm_util.c:139 */
void spasm_hip_triplet_free(struct spasm_triplet *T);                              /* spasm_util.c:165 */
void spasm_hip_lu_free(struct spasm_lu *N);                                        /* spasm_util.c:208 */
void spasm_hip_add_entry(struct spasm_triplet *T, int i, int j, i64 x);            /* spasm_triplet.c:7 */
void spasm_hip_triplet_transpose(struct spasm_triplet *T);                         /* spasm_triplet.c:25 */
struct spasm_csr *spasm_hip_compress(const struct spasm_triplet *T);               /* spasm_triplet.c:97 */
struct spasm_csr *spasm_hip_transpose(const struct spasm_csr *C, int keep_values); /* spasm_transpose.c:5 */

/* --- SMS / MatrixMarket I/O (replace spasm_io.c:60-192) --- */
struct spasm_triplet *spasm_hip_triplet_load(FILE *f, i64 prime, u8 *hash);        /* spasm_io.c:60 */
void spasm_hip_triplet_save(const struct spasm_triplet *A, FILE *f);               /* spasm_io.c:183 */
void spasm_hip_csr_save(const struct spasm_csr *A, FILE *f);                       /* spasm_io.c:163 */

/* --- structural pivot search (replaces spasm_pivots.c:369) ---
 * The Faugere-Lachartre steps, the topological order and the rows of U are host work.  The greedy cycle-free search
 * (spasm_pivots.c:147-305) of a matrix with at least 20,000 rows runs ON THE DEVICE when the process has one
 * (spasm_amd/csrc/pivots_device.hip: one wavefront per candidate row; up to 2^25 columns; SPASM_HIP_PIVOT_SEARCH=host
 * keeps it on the host threads, =device refuses to fall back); without a device -- the CPU tests, hosts that only plan --
 * the host search of host_pivots.cpp runs: same transactions, same guarantees (a cycle-free set; which one depends on
 * timing, as it does in the reference under OpenMP; one thread = the reference's sequential outcome). */
int spasm_hip_pivots_extract_structural(const struct spasm_csr *A, const int *p_in, struct spasm_lu *fact,
                                        int *p, struct echelonize_opts *opts);     /* spasm_pivots.c:369 */

/* --- the hot path: Schur complement on the GPU --- */

/* replaces spasm_schur (spasm_schur.c:61-193).  Row k of the result is the
 * reduction of row p[k] of A (the reference emits rows in thread-arrival
 * order and records the mapping in p_out; here the order is always p's).
 * Entries of a row are sorted by column.  L != NULL: the elimination coefficients are appended to L
 * as (p_in[row] or row, index of the pivot row in U, coefficient), like the reference. */
struct spasm_csr *spasm_hip_schur(const struct spasm_csr *A, const int *p, int n, const struct spasm_lu *fact,
                                  double est_density, struct spasm_triplet *L, const int *p_in, int *p_out);

/* replaces spasm_schur_estimate_density (spasm_schur.c:11-48) */
double spasm_hip_schur_estimate_density(const struct spasm_csr *A, const int *p, int n,
                                        const struct spasm_csr *U, const int *qinv, int R);

/* replaces spasm_schur_dense (spasm_schur.c:257-343) */
void spasm_hip_schur_dense(const struct spasm_csr *A, const int *p, int n, const int *p_in,
                           struct spasm_lu *fact, void *S, spasm_datatype datatype, int *q, int *p_out);

/* replaces spasm_schur_dense_randomized (spasm_schur.c:346-425) */
void spasm_hip_schur_dense_randomized(const struct spasm_csr *A, const int *p, int n, const struct spasm_csr *U,
                                      const int *qinv, void *S, spasm_datatype datatype, int *q, int N, int w);

/* replace spasm_ffpack_rref / spasm_ffpack_LU and the datatype helpers
 * (spasm_ffpack.cpp:88-148): dense (reduced) echelon form mod p on the GPU */
int spasm_hip_ffpack_rref(i64 prime, int n, int m, void *A, int ldA, spasm_datatype datatype, size_t *qinv);
int spasm_hip_ffpack_LU(i64 prime, int n, int m, void *A, int ldA, spasm_datatype datatype, size_t *p, size_t *qinv);
spasm_ZZp spasm_hip_datatype_read(const void *A, size_t i, spasm_datatype datatype);
void spasm_hip_datatype_write(void *A, size_t i, spasm_datatype datatype, spasm_ZZp value);
size_t spasm_hip_datatype_size(spasm_datatype datatype);
spasm_datatype spasm_hip_datatype_choose(i64 prime);
const char *spasm_hip_datatype_name(spasm_datatype datatype);

/* --- drivers (replace spasm_echelonize.c:9-28, :478-616; spasm_rref.c:25; spasm_kernel.c:9) --- */
void spasm_hip_echelonize_init_opts(struct echelonize_opts *opts);
/* Threads: the driver keeps process-wide state between its stages (device-resident matrices, the cached factor image, the
 * profile below); concurrent calls are taken one at a time (a mutex inside).  The other host-pointer entry points share the
 * one-entry factor cache (serialised too); the spasm_hip_d* layer keeps its state in the handles it is given. */
struct spasm_lu *spasm_hip_echelonize(const struct spasm_csr *A, struct echelonize_opts *opts);
/* seconds the last spasm_hip_echelonize call spent in: [0] the whole call, [1] the host pivot search, [2] density
 * estimates, [3] sparse Schur complements, [4] the dense / low-rank finish, [5] number of sparse Schur rounds,
 * [6] the structural-rounds finish that replaces GPLU, [7] host matrices uploaded to the device by this process so far
 * (inside the driver a matrix goes up at most once and Schur complements stay where they were computed).  8 doubles. */
void spasm_hip_echelonize_profile(double *out);
/* events since the last spasm_hip_echelonize call started, that its time split does not show -- out[k] for k < count:
 * [0] spasm_hip_schur calls redone because the pool of S was too small, [1] pools sized from a sampled run of the sparse
 * image, [2] chunks added to the pool of the sparse image R during a build, [3] persistent builds of R aborted and redone
 * level by level, [4] device blocks the block cache did not have and [5] their bytes, [6] factor images planned on the
 * host, [7] pivot rows visited by the device pivot searches, [8] of them by searches that ended with a pivot, [9] items of
 * the label cascades, [10] rows that got / [11] did not get a pivot in the greedy search, [12] pivots accepted on their
 * labels alone, [13] rows the labelled search deferred to the ticket search.  Returns how many there are (14). */
int spasm_hip_echelonize_counters(long long *out, int count);
struct spasm_csr *spasm_hip_rref(const struct spasm_lu *fact, int *Rqinv);
struct spasm_csr *spasm_hip_kernel(const struct spasm_lu *fact);

/* ======================================================================
 * (D) device-resident entry points
 * ====================================================================== */

/* Streams.  Every entry point below enqueues its work on the stream it is given and may return before it is done (the
 * ones that return a count -- ranks, sizes -- synchronise that stream first).  The objects (spasm_hip_dfact,
 * spasm_hip_dwork) and their buffers come from a cache of device blocks that is NOT stream-aware: destroying an object
 * while work that uses it is still queued, or using one object from two streams without an event between them, hands a
 * block that is still being written to the next allocation.  One stream per object, synchronised before
 * spasm_hip_d*_destroy: that is the contract (the host-level entry points use the null stream throughout). */

/* A CSR matrix whose arrays live in HBM.  Same layout and value convention
 * (balanced int32) as struct spasm_csr: p has n+1 int64, j/x have p[n] int32. */
typedef struct {
	int n;
	int m;
	i64 nnz;
	const i64 *p;
	const int *j;
	const spasm_ZZp *x;
} spasm_hip_dcsr;

/* Device image of the factor (U, qinv): rows of U sorted by elimination
 * level, columns relabelled so that the pivot of row c is column c, values in
 * Montgomery form.  Built on the host from the reference structs, then
 * uploaded.  See DESIGN.md "Data layout in HBM". */
typedef struct spasm_hip_dfact spasm_hip_dfact;
typedef struct spasm_hip_comm spasm_hip_comm;
spasm_hip_dfact *spasm_hip_dfact_create(const struct spasm_csr *U, const int *qinv, void *stream);
void spasm_hip_dfact_destroy(spasm_hip_dfact *F);
/* forgets derived state cached in the image (the back-substituted rows R): the next Schur call rebuilds it.
 * bench.py calls this before every step so that a step is the whole of spasm_schur. */
void spasm_hip_dfact_forget(spasm_hip_dfact *F);
/* Hints for the choice between the two elimination paths (a cost model): the expected density
 * (entries / (rows * non-pivotal columns)) of the Schur complements computed with this factor, and the number of (row,
 * pivot) eliminations a reduced row takes (what the row-by-row kernels count; spasm_hip_schur records it from the driver's
 * density sample by itself).  < 0: unknown (5 % density, 5 % of the pivots per row are assumed). */
void spasm_hip_dfact_hint_density(spasm_hip_dfact *F, double density);
void spasm_hip_dfact_hint_eliminations(spasm_hip_dfact *F, double per_row);
/* fill of R = U_pp^-1 U_pn as the sparse image holds it: out[0] entries, out[1] occupied 64-column tiles, out[2] non-empty
 * fragments, out[3] (row, segment) pairs.  Returns 1 when the factor holds a valid sparse image (else 0, out zeroed). */
int spasm_hip_dfact_sparse_image_census(const spasm_hip_dfact *F, i64 *out, void *stream);
int spasm_hip_dfact_rank(const spasm_hip_dfact *F);
int spasm_hip_dfact_levels(const spasm_hip_dfact *F);
i64 spasm_hip_dfact_nnz(const spasm_hip_dfact *F);

/* Scratch + output pool for the device Schur complement.  pool_entries is the
 * capacity of the row pool (one entry = one (column, value) pair). */
typedef struct spasm_hip_dwork spasm_hip_dwork;
spasm_hip_dwork *spasm_hip_dwork_create(int max_rows, int m, i64 pool_entries);
void spasm_hip_dwork_destroy(spasm_hip_dwork *W);

/* statistics of the last spasm_hip_dschur call on this workspace */
typedef struct {
	i64 nnz;                /* entries of S */
	i64 eliminations;       /* pivot rows applied (sum over rows) */
	i64 entries_streamed;   /* entries of U' read by those eliminations */
	i64 input_entries;      /* entries of the reduced rows of A */
	i64 group_pivots;       /* row-group kernel: (group, pivot) pairs applied; eliminations / (64 * this) = lane efficiency */
	int rows;               /* rows reduced */
	int rows_lds;           /* rows finished by the LDS hash kernel, small table */
	int rows_lds_big;       /* ... by the large-table variant */
	int rows_dense;         /* rows finished by the dense-accumulator kernel */
	int status;             /* 0 = ok, 1 = pool too small (call again with a larger pool) */
	int used_group_kernel;  /* 1: the row-group kernel ran (its rows are counted in rows_dense) */
	int group_aborted;      /* 1: it stopped early (poor lane efficiency), the per-row tiers finished the batch */
	float ms_eliminate;     /* device time of the elimination kernels, all tiers (HIP events on the call's stream) */
	float ms_group;         /* ... row-group kernel (schur_group_kernel), 0 when not used */
	float ms_tier0;         /* ... small-table LDS kernel (schur_lds_kernel<1024>) */
	float ms_tier1;         /* ... large-table LDS kernel (0 when skipped) */
	float ms_tier2;         /* ... dense-accumulator kernel (schur_wave_dense_kernel) */
	float ms_finalize;      /* row pointers + sorted gather */
	float ms_total;         /* device time of the whole call */
	int used_backsolve;     /* 1: the rows were computed from the back-substituted factor image (S = A_n - A_p R) */
	int backsolve_built;    /* 1: ... and R was (re)built by this call (ms_backsolve, bytes_backsolve say at what cost) */
	float ms_backsolve;     /* device time of building R = U_pp^-1 U_pn (memset + init + backsolve_kernel) */
	float ms_apply;         /* device time of the apply kernel (S rows from R): bs_apply_kernel, or bs_apply_s16_kernel */
	i64 bytes_backsolve;    /* algorithmic bytes of that build */
	i64 bytes_apply;        /* ... of the apply kernel */
	char kernel[64];        /* name of the dominant elimination kernel this call launched, as rocprofv3 shows it */
	char kernel_other[64];  /* back-substituted path: the other of its two kernels (build of R / apply) */
	float ms_expand;        /* staged output (the default): device time of bs_expand_kernel, not part of ms_apply; else 0 */
	float ms_pad;
	i64 bytes_expand;       /* ... its algorithmic bytes (the entries of S written); they are then not part of bytes_apply */
	i64 bytes_staged;       /* ... bytes of the packed rows in between (written by the apply kernel, read by the expansion) */
	char kernel_expand[64]; /* ... its name, or "" */
	/* row-group kernel: accumulator slices (= groups in flight) the launch ran with, the number that would fill the chip at
	 * its wave count (8 waves per CU), waves per group, bytes of one slice ((rpad + Sm) * 256 B) */
	int group_slots, group_slots_wanted, group_waves;
	i64 group_slot_bytes;
	/* sparse image (S = A_n - A_p R with R kept as sparse fragments: spasm_amd/csrc/sparse_image.hip) */
	int used_sparse_image;      /* 1: the rows were computed from the sparse image */
	int sparse_image_built;     /* 1: ... and the image was (re)built by this call */
	float ms_sparse_build;      /* device time of that build (sp_build_kernel: one cooperative launch, rows handed out by tickets) */
	float ms_sparse_apply;      /* ... of sp_apply_kernel (fragments of S) */
	float ms_sparse_gather;     /* ... of the scan of the row lengths + sp_gather_kernel (rows in their final place) */
	int sparse_image_levels;    /* elimination levels of the factor */
	int sparse_image_launches;  /* launches of the build (1; + one per pool extension; the level-by-level fall-back: one per level) */
	int sparse_image_pad;
	i64 sparse_image_nnz;       /* entries of R */
	i64 sparse_image_ops_build; /* multiply-adds of the build (entries of the fragments added up) */
	i64 sparse_image_ops_apply; /* ... of the rows of S */
	i64 bytes_sparse_build;     /* algorithmic bytes of the three kernels */
	i64 bytes_sparse_apply;
	i64 bytes_sparse_gather;
} spasm_hip_schur_stats;

/* S = Schur complement of rows d_rows[0..nrows) of A w.r.t. F, left in the
 * workspace (device).  Everything is enqueued on `stream`; the call returns
 * after synchronising that stream (it has to read the total size back).
 * Returns 0 on success, 1 if the pool was too small. */
int spasm_hip_dschur(const spasm_hip_dcsr *A, const int *d_rows, int nrows, const spasm_hip_dfact *F,
                     spasm_hip_dwork *W, void *stream, spasm_hip_schur_stats *stats);

/* copies the result of the last spasm_hip_dschur into caller-provided device
 * buffers (Sp: nrows+1 int64; Sj, Sx: stats.nnz int32 each). */
void spasm_hip_dschur_fetch(const spasm_hip_dwork *W, i64 *d_Sp, int *d_Sj, spasm_ZZp *d_Sx, void *stream);
/* ... only its row pointers (rows + 1 int64; enqueued on `stream`, not synchronised) */
void spasm_hip_dschur_row_pointers(const spasm_hip_dwork *W, i64 *d_Sp, void *stream);

/* dense rows of the Schur complement, left on the device: d_S is nrows x Sm
 * (Sm = m - rank), row-major with leading dimension ldS, values in [0, p). */
/* test hook: N pseudo-random combinations of rows of A (w > 0: of w random rows each; w <= 0: of all nrows rows), as the dense /
 * low-rank finish forms them on the device, N x m residues in [0, p) written to the host array `out` */
void spasm_hip_debug_combine(const struct spasm_csr *A, const int *rows, int nrows, int N, int w, uint64_t salt, u32 *out);

int spasm_hip_dschur_dense(const spasm_hip_dcsr *A, const int *d_rows, int nrows, const spasm_hip_dfact *F,
                           spasm_hip_dwork *W, u32 *d_S, i64 ldS, void *stream);

/* in-place reduced row echelon form of a dense n x m matrix mod p resident in
 * HBM (values in [0,p), leading dimension ld).  On return d_pivcol[0..rank)
 * holds the pivot column of each echelon row.  Returns the rank. */
int spasm_hip_drref(i64 prime, int n, int m, u32 *d_A, i64 ld, int *d_pivcol, void *stream);

/* Extends k reduced echelon rows (rows [0, k) of d_M, identity on their pivot columns d_piv[0..k)) by the Sn rows below
 * them: these are reduced by the echelon rows and by each other, the non-zero ones join the echelon rows (which stay
 * reduced).  Pivots are taken where the rows have their leftmost entries, 64 rows at a time -- an echelon basis of the row
 * space, not the column rank profile spasm_hip_drref returns: made for wide stacks of low rank (the dense / low-rank finish
 * on tens of thousands of columns).  Values in [0, p), p <= 65279.  Returns the new number of echelon rows k'; d_piv[0..k')
 * are their pivot columns.  Rows beyond k' are left undefined. */
int spasm_hip_dechelon_extend(i64 prime, int m, u32 *d_M, i64 ld, int k, int Sn, int *d_piv, void *stream);

/* ======================================================================
 * (M) multi-GPU: one process per GPU, RCCL over xGMI (spasm_amd/csrc/dist_api.hip)
 *
 * EXPERIMENTAL: this section has never run on more than one GPU (the test pool grants one per call).  What can be
 * checked without a second GPU is: the exchange plan and the column split are pure host functions tested for every world
 * size (tests/test_dist_cpu.py), the RCCL code runs in a world of one (tests/test_gpu_dist.py).
 *
 * What shards is what the reference hands to its OpenMP threads: the rows of a Schur complement
 * (spasm_schur.c:86-171).  Every rank holds A and the factor and reduces a contiguous slice of the row list; the
 * slices are reassembled on the devices with an all-gatherv.  With a communicator installed, the entry points of
 * section (H) shard by themselves and stay in step: every rank makes the same calls with the same arguments.
 * ====================================================================== */
int spasm_hip_comm_id_bytes(void);                      /* size of the opaque id below (an ncclUniqueId) */
void spasm_hip_comm_new_id(void *id);                   /* one process draws an id and hands it to the others (any channel) */
spasm_hip_comm *spasm_hip_comm_create(const void *id, int rank, int world);     /* collective; the thread's current device */
void spasm_hip_comm_destroy(spasm_hip_comm *c);
int spasm_hip_comm_rank(const spasm_hip_comm *c);
int spasm_hip_comm_world(const spasm_hip_comm *c);
void spasm_hip_set_comm(spasm_hip_comm *c);             /* NULL: back to one GPU */
void spasm_hip_shard(int n, int rank, int world, int *lo, int *hi);             /* slice [lo, hi) of n rows owned by rank */

/* all-gatherv of what the last spasm_hip_dschur left in every rank's workspace: rows of rank 0, then rank 1, ... as one
 * CSR in device buffers (d_Sp: total rows + 1, d_Sj / d_Sx: cap entries).  Returns 1 (nothing moved, *total_nnz set) when
 * cap is too small -- call it with cap = -1 to learn the sizes. */
int spasm_hip_dschur_allgatherv(spasm_hip_comm *c, const spasm_hip_dwork *W, i64 *d_Sp, int *d_Sj, spasm_ZZp *d_Sx, i64 cap,
                                int *total_rows, i64 *total_nnz, void *stream);

/* Column split (the image paths: the columns of R and of S never meet).  With a communicator installed spasm_hip_schur gives
 * rank k the slab k of the non-pivotal columns (spasm_hip_column_slab below): it builds ITS columns of the image only, reduces
 * ALL rows on them, and the slabs are stacked by the all-gatherv and stitched into whole rows on every device.  The stitching
 * alone, for `parts` slabs of n rows stacked in device arrays (d_gSp: parts * n + 1 offsets): row i of the result is slab 0's
 * row i, then slab 1's, ...  Returns 0, or 1 when cap (entries of d_Sj / d_Sx) is too small. */
int spasm_hip_dstitch_slabs(const i64 *d_gSp, const int *d_gSj, const spasm_ZZp *d_gSx, int n, int parts, i64 *d_Sp, int *d_Sj, spasm_ZZp *d_Sx, i64 cap,
                            void *stream);

/* The exchange plan of spasm_hip_dschur_allgatherv as a pure host function (spasm_amd/csrc/host_dist.cpp; no GPU, no
 * RCCL): the steps rank `me` issues, in order, for slices of sizes[2r] rows and sizes[2r + 1] entries.  The collective
 * executes exactly this list; tests/test_dist_cpu.py checks it for worlds of 2 to 8 (every pair's send sequence equals the
 * peer's receive sequence, receives + local copy tile the output). */
enum {SPASM_HIP_XFER_SEND = 0, SPASM_HIP_XFER_RECV = 1, SPASM_HIP_XFER_COPY = 2};
typedef struct {
	int kind;       /* SPASM_HIP_XFER_* */
	int peer;       /* the other rank (this rank for a local copy) */
	int array;      /* 0: row pointers (int64), 1: column indices, 2: values (int32) */
	i64 src;        /* send, copy: offset in this rank's own array */
	i64 dst;        /* receive, copy: offset in the gathered array */
	i64 count;      /* elements */
} spasm_hip_xfer;
int spasm_hip_allgatherv_plan(int world, int me, const i64 *sizes, spasm_hip_xfer *out, int cap, i64 *row_base, i64 *nz_base);

/* The split that fits the back-substituted path: rank `part` of `parts` owns a contiguous range of the non-pivotal columns
 * (in increasing column order) and works on (A, U) with the other non-pivotal columns deleted -- no replicated image, no
 * exchange until somebody needs whole rows.  cols (room for A->m ints) maps the columns of the slab problem back; returns
 * their number.  Pure host function (spasm_amd/csrc/host_dist.cpp).  Reference loop being split: spasm_schur.c:86-171. */
int spasm_hip_column_slab(const struct spasm_csr *A, const struct spasm_lu *fact, int part, int parts, struct spasm_csr **A_slab,
                          struct spasm_lu **fact_slab, int *cols);

/* For callers that TIME the product's path of a sparse round (bench.py --gpus N): spasm_hip_schur (spasm_schur, spasm_schur.c:61) as
 * spasm_hip_echelonize calls it between two rounds -- residency on, the entries of S left on the device, the installed communicator
 * in force: with several ranks this is the column (or row) split with its all-gatherv and its stitching.  Returns nnz(S); S is
 * dropped.  spasm_hip_forget_cached_images: the cached factor images forget R, so that the next call pays for all of spasm_schur. */
i64 spasm_hip_schur_resident(const struct spasm_csr *A, const int *p, int n, const struct spasm_lu *fact, double est_density);
void spasm_hip_forget_cached_images(void);

/* spasm_echelonize (spasm_echelonize.c:473) with every round's Schur complement sharded over the ranks of c; the pivot
 * search runs on rank 0 and is broadcast.  Collective; same rank of the matrix on every rank. */
struct spasm_lu *spasm_hip_echelonize_dist(const struct spasm_csr *A, struct echelonize_opts *opts, spasm_hip_comm *c);

#ifdef __cplusplus
}
#endif
#endif
