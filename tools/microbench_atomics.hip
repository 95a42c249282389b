// Scattered 4-byte read-modify-write into a large per-wave private region: which primitive is fast
// on gfx950?  (a) fire-and-forget atomic add, wavefront scope; (b) same, agent scope;
// (c) sc1 load + add + plain store; (d) returning atomic add.  One wave per block, each wave owns a
// 512 KiB slice and issues ROUNDS x 64 random updates.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE> __global__ __launch_bounds__(64) void k(uint32_t *buf, int words_per_wave, int rounds, uint32_t *sink)
{
	uint32_t *mine = buf + (size_t) blockIdx.x * words_per_wave;
	uint32_t s = blockIdx.x * 64u + threadIdx.x + 12345u;
	uint32_t acc = 0;
	for (int r = 0; r < rounds; r++) {
		s = s * 1664525u + 1013904223u;
		uint32_t idx = (s >> 8) % (uint32_t) words_per_wave;
		if (MODE == 0)
			(void) __hip_atomic_fetch_add(&mine[idx], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		else if (MODE == 1)
			(void) __hip_atomic_fetch_add(&mine[idx], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else if (MODE == 2) {
			uint32_t v = __hip_atomic_load(&mine[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			mine[idx] = v + 3u;
		} else if (MODE == 3) {
			acc += __hip_atomic_fetch_add(&mine[idx], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
		} else if (MODE == 4) {
			uint32_t v = mine[idx];
			mine[idx] = v + 3u;
		} else if (MODE == 5) {
			mine[idx] = s;            // store only
		} else if (MODE == 6) {
			acc += __hip_atomic_load(&mine[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // load only
		}
	}
	if (acc == 0xdeadbeef)
		sink[0] = acc;
}

// Line shape of the row-group kernel: a wave updates random 256-byte lines of a large slice, lane l -> word l of
// the line, with only the first ACTIVE lanes taking part.  MODE 0: no-return atomic add; 1: load + add + store
// (loads issued 8 lines ahead); 2: store only.  Reports wave-instructions per second.
template <int MODE, int ACTIVE> __global__ __launch_bounds__(64) void kline(uint32_t *buf, int lines_per_wave, int rounds, uint32_t *sink)
{
	uint32_t *mine = buf + (size_t) blockIdx.x * lines_per_wave * 64;
	uint32_t s = blockIdx.x * 2654435761u + 12345u;
	const int lane = threadIdx.x;
	uint32_t acc = 0;
	for (int r = 0; r < rounds; r += 8) {
		uint32_t idx[8], v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) {
			s = s * 1664525u + 1013904223u;
			idx[u] = ((s >> 8) % (uint32_t) lines_per_wave) * 64 + lane;
		}
		if (lane < ACTIVE) {
			if (MODE == 1) {
#pragma unroll
				for (int u = 0; u < 8; u++)
					v[u] = __hip_atomic_load(&mine[idx[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
				for (int u = 0; u < 8; u++)
					mine[idx[u]] = v[u] + 3u;
			} else if (MODE == 0) {
#pragma unroll
				for (int u = 0; u < 8; u++)
					(void) __hip_atomic_fetch_add(&mine[idx[u]], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
			} else {
#pragma unroll
				for (int u = 0; u < 8; u++)
					mine[idx[u]] = s + u;
			}
		}
	}
	if (acc == 0xdeadbeef)
		sink[0] = acc;
}

template <int MODE, int ACTIVE> void run_line(const char *name, uint32_t *buf, int blocks, int lpw, int rounds, uint32_t *sink)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	hipLaunchKernelGGL((kline<MODE, ACTIVE>), dim3(blocks), dim3(64), 0, 0, buf, lpw, rounds / 4, sink);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a, 0));
	hipLaunchKernelGGL((kline<MODE, ACTIVE>), dim3(blocks), dim3(64), 0, 0, buf, lpw, rounds, sink);
	CK(hipEventRecord(b, 0));
	CK(hipDeviceSynchronize());
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	double ops = (double) blocks * rounds;
	printf("%-52s %2d lanes %8.2f ms  %7.2f G wave-instr/s  %7.2f G 64-B requests/s\n", name, ACTIVE, ms, ops / ms / 1e6,
	       ops * ((ACTIVE + 15) / 16) / ms / 1e6);
}

template <int MODE> void run(const char *name, uint32_t *buf, int blocks, int wpw, int rounds, uint32_t *sink)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, buf, wpw, rounds / 4, sink);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a, 0));
	hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, buf, wpw, rounds, sink);
	CK(hipEventRecord(b, 0));
	CK(hipDeviceSynchronize());
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	double ops = (double) blocks * 64.0 * rounds;
	printf("%-44s %8.2f ms  %8.2f G updates/s\n", name, ms, ops / ms / 1e6);
}

int main(int argc, char **argv)
{
	int blocks = argc > 1 ? atoi(argv[1]) : 8192;
	int wpw = 128 * 1024;                 // 512 KiB per wave
	int rounds = 4096;
	uint32_t *buf, *sink;
	CK(hipMalloc(&buf, (size_t) blocks * wpw * 4));
	CK(hipMalloc(&sink, 64));
	CK(hipMemset(buf, 0, (size_t) blocks * wpw * 4));
	printf("blocks=%d waves, %d KiB each, %d x 64 scattered updates per wave\n", blocks, wpw * 4 / 1024, rounds);
	run<0>("atomic add, no return, wavefront scope", buf, blocks, wpw, rounds, sink);
	run<1>("atomic add, no return, agent scope", buf, blocks, wpw, rounds, sink);
	run<3>("atomic add, returning, wavefront scope", buf, blocks, wpw, rounds, sink);
	run<2>("sc1 load + plain store", buf, blocks, wpw, rounds, sink);
	run<4>("plain load + plain store", buf, blocks, wpw, rounds, sink);
	run<5>("plain store only", buf, blocks, wpw, rounds, sink);
	run<6>("sc1 load only", buf, blocks, wpw, rounds, sink);
	printf("\nwhole lines (256 B, lane = word), %d lines per wave, random:\n", wpw / 64);
	run_line<0, 64>("atomic add, no return", buf, blocks, wpw / 64, rounds, sink);
	run_line<0, 32>("atomic add, no return", buf, blocks, wpw / 64, rounds, sink);
	run_line<0, 16>("atomic add, no return", buf, blocks, wpw / 64, rounds, sink);
	run_line<1, 64>("sc1 load + add + plain store, 8 lines in flight", buf, blocks, wpw / 64, rounds, sink);
	run_line<1, 16>("sc1 load + add + plain store, 8 lines in flight", buf, blocks, wpw / 64, rounds, sink);
	run_line<2, 64>("plain store only", buf, blocks, wpw / 64, rounds, sink);
	return 0;
}
