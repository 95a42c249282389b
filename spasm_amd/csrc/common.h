// Internal helpers shared by the host-side sources of libspasm_hip.so.
#pragma once

#include <cstdarg>
#include <cstdint>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/spasm_hip.h"

namespace sh {

// Fatal error: same contract as the reference's err()/errx() use -- message on
// stderr, then exit(1).  The product never falls back to a CPU path.
[[noreturn]] inline void die(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	std::fprintf(stderr, "[spasm-hip] fatal: ");
	std::vfprintf(stderr, fmt, ap);
	std::fprintf(stderr, "\n");
	va_end(ap);
	std::exit(1);
}

inline void *xmalloc(int64_t bytes)
{
	void *q = std::malloc(bytes > 0 ? (size_t) bytes : 1);
	if (q == nullptr)
		die("malloc failed (%lld bytes)", (long long) bytes);
	return q;
}

inline void *xrealloc(void *old, int64_t bytes)
{
	void *q = std::realloc(old, bytes > 0 ? (size_t) bytes : 1);
	if (q == nullptr)
		die("realloc failed (%lld bytes)", (long long) bytes);
	return q;
}

double wtime();

// Environment switches.  The SUPPORTED ones (listed in include/spasm_hip.h) are read as they are; every other SPASM_HIP_* name
// selects an experiment, a debugging aid or a code path kept for A/B runs and tests, and is only honoured when
// SPASM_HIP_EXPERIMENT=1 is set as well (tests/conftest.py sets it).  Returns the value or nullptr.
const char *env_get(const char *name);
// fn(0) .. fn(ntasks - 1) on the library's worker threads and the caller (host_util.cpp); returns when all are done
void pool_run(int ntasks, const std::function<void(int)> &fn);

// verbosity of the progress messages on stderr (SPASM_HIP_VERBOSE=0 silences them)
int verbose();
void logmsg(const char *fmt, ...);

// Events of a driver call that its time split does not show (include/spasm_hip.h: spasm_hip_echelonize_counters): a stage that
// took 2 s instead of 0.07 s in one call of five is a retry, an extension or memory that had to be fetched -- counted here,
// reset when spasm_hip_echelonize starts, added to by whoever causes them (also outside the driver).
enum {
	CNT_POOL_RETRIES = 0,        // spasm_hip_schur calls redone because the row pool of S was too small
	CNT_POOL_RESIZED,            // ... pools sized from a sampled run of the sparse image instead of the driver's 100-row estimate
	CNT_SP_CHUNK_EXTENSIONS,     // chunks added to the pool of the sparse image R during a build
	CNT_SP_BUILD_ABORTS,         // persistent builds of R that were aborted (watchdog / residency) and redone level by level
	CNT_BIG_ALLOC_MISSES,        // device blocks >= 1 MB that the block cache did not have (hipMalloc)
	CNT_BIG_ALLOC_MISS_BYTES,
	CNT_FACTOR_PLANS,            // factor images planned on the host (level schedule + upload)
	CNT_PIVOT_VISITS,            // pivot rows visited by the device searches (walks of both passes)
	CNT_PIVOT_VISITS_WON,        // ... of them by searches that ended with a pivot
	CNT_PIVOT_CASCADE_ITEMS,     // (column, label) items of the label cascades
	CNT_PIVOT_ROWS_WON,
	CNT_PIVOT_ROWS_LOST,
	CNT_PIVOT_FREE_ACCEPTS,      // pivots accepted on their labels alone (no walk)
	CNT_PIVOT_DEFERRED,          // rows the labelled search handed to the ticket search
	CNT_SLABS_KEPT,              // Schur complements left as column slabs on the devices (column split, between two rounds)
	CNT_SLABS_GATHERED,          // ... of them gathered into whole rows after all (another sparse round, a download)
	CNT_COUNT
};
long long *counters();

// ---- GF(p), balanced representatives, exact 64-bit arithmetic ----
inline spasm_ZZp zp_balance(int64_t p, int64_t r)
{
	const int64_t hi = p / 2, lo = p / 2 - p + 1;
	if (r < lo)
		r += p;
	else if (r > hi)
		r -= p;
	return (spasm_ZZp) r;
}
inline spasm_ZZp zp_init(int64_t p, int64_t x) { return zp_balance(p, x % p); }
inline spasm_ZZp zp_mul(int64_t p, spasm_ZZp a, spasm_ZZp b) { return zp_balance(p, ((int64_t) a * b) % p); }
inline spasm_ZZp zp_axpy(int64_t p, spasm_ZZp a, spasm_ZZp x, spasm_ZZp y)
{
	return zp_balance(p, ((int64_t) a * x + y) % p);
}
spasm_ZZp zp_inverse(int64_t p, spasm_ZZp a);

// [0,p) representative of a balanced one
inline uint32_t zp_unsigned(int64_t p, spasm_ZZp a) { return (uint32_t) (a < 0 ? (int64_t) a + p : (int64_t) a); }

// ---- Montgomery constants for an odd modulus p < 2^32 (R = 2^32) ----
struct Mont {
	uint32_t p;
	uint32_t pinv;     // p^-1 mod 2^32
	uint32_t r1;       // R mod p
	uint32_t r2;       // R^2 mod p
	uint32_t half;     // p / 2
};
Mont mont_setup(int64_t prime);

}  // namespace sh
