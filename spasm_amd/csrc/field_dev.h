// Device-side GF(p) arithmetic shared by the kernels: 32-bit Montgomery (R = 2^32), any odd p < 2^32.
#pragma once

#include "device_types.h"

namespace sh {

// a * b * 2^-32 mod p, for a * b < p * 2^32; result in [0, p)
__device__ __forceinline__ uint32_t montmul(uint32_t a, uint32_t b, const MontDev &F)
{
	uint64_t t = (uint64_t) a * b;
	uint32_t mq = (uint32_t) t * F.pinv;
	uint32_t q = __umulhi(mq, F.p);
	uint32_t th = (uint32_t) (t >> 32);
	uint32_t r = th - q;
	return (th < q) ? r + F.p : r;
}

// same product without the final correction: a value in (0, 2p) congruent to a * b * 2^-32; two instructions
// shorter, for sums that are reduced later anyway (needs p < 2^31)
__device__ __forceinline__ uint32_t montmul_lazy(uint32_t a, uint32_t b, const MontDev &F)
{
	uint64_t t = (uint64_t) a * b;
	uint32_t mq = (uint32_t) t * F.pinv;
	uint32_t q = __umulhi(mq, F.p);
	return (uint32_t) (t >> 32) - q + F.p;
}

// v mod p for any 32-bit v
__device__ __forceinline__ uint32_t reduce_sum(uint32_t v, const MontDev &F) { return montmul(v, F.r1, F); }

// v mod p for any 64-bit v: (hi * 2^32 + lo) mod p with two Montgomery products
__device__ __forceinline__ uint32_t reduce_sum(unsigned long long v, const MontDev &F)
{
	uint32_t a = montmul((uint32_t) (v >> 32), F.r2, F);
	uint32_t b = montmul((uint32_t) v, F.r1, F);
	uint32_t s = a + b;
	if (s < a || s >= F.p)
		s -= F.p;
	return s;
}

// a * b mod p for a, b in [0, p)
__device__ __forceinline__ uint32_t mulmod(uint32_t a, uint32_t b, const MontDev &F)
{
	return montmul(montmul(a, b, F), F.r2, F);
}

__device__ __forceinline__ uint32_t submod(uint32_t a, uint32_t b, const MontDev &F)
{
	return (a >= b) ? a - b : a + (F.p - b);
}

// balanced representative (spasm_ZZp) <-> canonical [0, p)
__device__ __forceinline__ int to_balanced(uint32_t v, const MontDev &F)
{
	return (v > F.half) ? (int) (v - F.p) : (int) v;
}

__device__ __forceinline__ uint32_t from_balanced(int a, const MontDev &F)
{
	return (a < 0) ? (uint32_t) a + F.p : (uint32_t) a;
}

}  // namespace sh
