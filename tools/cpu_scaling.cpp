// how many threads of the box really run in parallel?  T threads, each a fixed amount of (a) register arithmetic and
// (b) random reads in a private 8 MB array; prints the wall time per T.  (The container may be limited to a CPU quota
// below the number of hardware threads it reports.)
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
	for (int T : {1, 8, 16, 32, 64, 128, 256}) {
		std::vector<uint64_t> out((size_t) T);
		std::vector<std::thread> pool;
		const double t0 = now();
		for (int t = 0; t < T; t++)
			pool.emplace_back([&, t]() {
				uint64_t x = 88172645463325252ull + t, acc = 0;
				for (long k = 0; k < 200000000L; k++) {
					x ^= x << 13; x ^= x >> 7; x ^= x << 17;
					acc += x;
				}
				out[t] = acc;
			});
		for (auto &th : pool) th.join();
		const double t1 = now();
		pool.clear();
		for (int t = 0; t < T; t++)
			pool.emplace_back([&, t]() {
				std::vector<uint32_t> a(2 << 20);
				for (size_t k = 0; k < a.size(); k++) a[k] = (uint32_t) ((k * 2654435761u + 12345u) & (a.size() - 1));
				uint32_t p = t;
				for (long k = 0; k < 20000000L; k++) p = a[p] ^ (uint32_t) (k & 1023);
				out[t] += p;
			});
		for (auto &th : pool) th.join();
		const double t2 = now();
		printf("%3d threads: arithmetic %.3f s, random reads %.3f s (checksum %llu)\n", T, t1 - t0, t2 - t1, (unsigned long long) out[0]);
		fflush(stdout);
	}
	return 0;
}
