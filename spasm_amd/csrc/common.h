// Internal helpers shared by the host-side sources of libspasm_hip.so.
#pragma once

#include <cstdarg>
#include <cstdint>
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

// verbosity of the progress messages on stderr (SPASM_HIP_VERBOSE=0 silences them)
int verbose();
void logmsg(const char *fmt, ...);

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
