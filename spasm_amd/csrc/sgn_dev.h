// Signed 16-bit representatives mod p with deferred reduction: shared by the dense (backsolve.hip) and the sparse
// (sparse_image.hip) back-substituted images.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace sh {

// ---- signed 16-bit entries with deferred reduction (SGN) -------------------------------------------
// For small primes an entry of R is a signed 16-bit representative v, |v| <= B = p/2 + p/64 + 1, and a coefficient c
// is the NEGATED balanced residue, |c| <= p/2.  x - sum c_i v_i then costs ONE v_mad_i32_i16 per term and component
// (op_sel picks the half of the packed word: nothing to unpack), and up to four terms are added to an entry before it
// is reduced: 4 B^2 + B < 2^31.  The reduction goes through fp32: q = rint(float(t) / p) is the nearest multiple up to an
// error of 3 * 2^-24 |t| / p < 1/64 (t < 2^31, p < 2^16), so |t - q p| <= p/2 + p/64: the slack B allows.  Zero is
// exact (|r| < p), and whoever hands values out brings them into [-p/2, p/2].  p <= 44934 qualifies (42013 does).
struct SgnDev {
	int p, negp, half;
	float invp;
};

inline bool sgn_eligible(int64_t prime)
{
	const int64_t B = prime / 2 + prime / 64 + 1;
	return prime >= 3 && B <= 32767 && 4 * B * B + B <= 0x7FFFFFFFll;
}

inline SgnDev sgn_setup(int64_t prime)
{
	return SgnDev{(int) prime, -(int) prime, (int) (prime / 2), 1.0f / (float) prime};
}

__device__ __forceinline__ void sgn_unpack(uint32_t w, int &lo, int &hi)
{
	lo = (int) (short) (w & 0xFFFFu);
	hi = (int) w >> 16;
}

// lo += low half of w * c, hi += high half of w * c (16-bit signed factors, 32-bit sums)
__device__ __forceinline__ void sgn_mad(uint32_t w, int c, int &lo, int &hi)
{
	asm("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(lo) : "v"(w), "v"(c));
	asm("v_mad_i32_i16 %0, %1, %2, %0 op_sel:[1,0,0,0]" : "+v"(hi) : "v"(w), "v"(c));
}

__device__ __forceinline__ int sgn_reduce(int t, const SgnDev &G)
{
	// q = round(t / p) out of the mantissa: fl(t) * (1/p) + 1.5 * 2^23 in one fused operation leaves q + 2^22 in the low 23 bits
	// (|q| < 2^17), rounded once to nearest -- two instructions where multiply, round and convert were three; the error of
	// q against t / p is at most 1/2 + |q| 2^-23, as before
	const int q = __float_as_int(__builtin_fmaf((float) t, G.invp, 12582912.0f)) - 0x4B400000;
	// q * (-p) + t in ONE full-rate instruction (|q| <= |t| / p + 1 < 2^17, p < 2^16: both fit 24 bits).  Written out: left to
	// itself the compiler took v_mad_u64_u32 for `t + __mul24(q, negp)` in sparse_image.hip -- a quarter-rate instruction.
	int r;
	// (-p as a scalar operand: G is wave-uniform everywhere; through a "v" constraint it cost a v_mov per reduction)
	asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(q), "s"(G.negp), "v"(t));
	return r;
}

__device__ __forceinline__ uint32_t sgn_pack(int lo, int hi)
{
	return __builtin_amdgcn_perm((uint32_t) hi, (uint32_t) lo, 0x05040100u);       // (lo & 0xFFFF) | (hi << 16)
}

__device__ __forceinline__ int sgn_canonical(int v, const SgnDev &G)          // into [-p/2, p/2]
{
	v = (v > G.half) ? v - G.p : v;
	return (v < -G.half) ? v + G.p : v;
}

__device__ __forceinline__ int sgn_from_residue(uint32_t v, const SgnDev &G)   // 0 <= v < p
{
	return ((int) v > G.half) ? (int) v - G.p : (int) v;
}

}  // namespace sh
