// Structural pivot search (replaces spasm_pivots.c).
//   1. Faugere-Lachartre: leftmost entry of each row, sparsest row wins;
//   2. the same idea on columns not touched by a pivotal row;
//   3. greedy search for pivots that keep the pivot graph acyclic (PASCO'17): on the device when there is one
//      (pivots_device.hip), else here -- with threads and optimistic transactions, or in row order (one thread: the
//      outcome of the reference with one thread).
// Then the pivotal rows are ordered topologically and appended to U, scaled so
// that every pivot is 1 and stored first in its row.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <algorithm>
#include <vector>

#include "common.h"

using namespace sh;

namespace sh {
// multi-GPU layer (dist_api.hip)
spasm_hip_comm *current_comm();
int comm_rank(const spasm_hip_comm *c);
int comm_world(const spasm_hip_comm *c);
void comm_bcast_host(spasm_hip_comm *c, void *buf, size_t bytes, int root);
// pivots_device.hip: the greedy search on the device; -1 = does not apply here (no device, too many columns, switched off)
int device_acyclic_greedy(const struct spasm_csr *A, int *pinv, int *qinv, std::vector<int> *col_label);
bool resident_prefetch_possible();                              // schur_api.hip
int resident_current_device();
void resident_prefetch_matrix(const struct spasm_csr *A, int dev);
void level_hint_set(const struct spasm_csr *U, int rows, std::vector<int> &&height);          // schur_api.hip
}  // namespace sh

namespace sh {
// CPUs this process may really use: the hardware threads, cut down to the CPU quota of its control group when there is
// one (cgroup v2 cpu.max, v1 cfs_quota_us).  The MI355X boxes of this pool report 256 hardware threads and grant 16 CPUs
// (cpu.max = "1600000 100000"; tools/cpu_scaling.cpp: 16 threads run in the time of one, 32 take twice as long) --
// which is why the threaded pivot search stopped scaling at 16 threads in round 2, not its commit lock.
static int usable_cpus_uncached();

int usable_cpus()
{
	static const int cached = usable_cpus_uncached();          // (two files of /sys/fs/cgroup are read: once per process)
	return cached;
}

static int usable_cpus_uncached()
{
	int hw = (int) std::thread::hardware_concurrency();
	if (hw <= 0)
		hw = 1;
	double quota = 0.0;
	if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
		char a[64] = {0};
		long long period = 0;
		if (std::fscanf(f, "%63s %lld", a, &period) == 2 && std::strcmp(a, "max") != 0 && period > 0)
			quota = std::atof(a) / (double) period;
		std::fclose(f);
	} else {
		long long q = -1, period = 0;
		if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
			if (std::fscanf(g, "%lld", &q) != 1)
				q = -1;
			std::fclose(g);
		}
		if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
			if (std::fscanf(g, "%lld", &period) != 1)
				period = 0;
			std::fclose(g);
		}
		if (q > 0 && period > 0)
			quota = (double) q / (double) period;
	}
	if (quota >= 1.0 && quota < (double) hw)
		hw = (int) (quota + 0.5);
	return hw;
}
}  // namespace sh

namespace {

struct Search {
	const struct spasm_csr *A;
	std::vector<int> pinv;   // row -> pivot column or -1
	std::vector<int> qinv;   // column -> pivot row or -1

	int weight(int i) const { return (int) (A->p[i + 1] - A->p[i]); }

	// returns 1 when neither the row nor the column had a pivot before
	int take(int i, int j)
	{
		int fresh = 1;
		if (pinv[i] != -1) {
			qinv[pinv[i]] = -1;
			fresh = 0;
		}
		if (qinv[j] != -1) {
			pinv[qinv[j]] = -1;
			fresh = 0;
		}
		pinv[i] = j;
		qinv[j] = i;
		return fresh;
	}

	// the same on large inputs, by threads: the pivot of a column is the row of smallest (weight, index) among the rows whose
	// leftmost entry it is -- what the loop below arrives at, row after row -- found with one atomic minimum per row
	int leftmost_entries_threads(int T)
	{
		const int n = A->n, m = A->m;
		std::vector<std::atomic<uint64_t>> best((size_t) (m > 0 ? m : 1));
		for (int j = 0; j < m; j++)
			best[(size_t) j].store(~0ull, std::memory_order_relaxed);
		auto scan = [&](int lo, int hi) {
			for (int i = lo; i < hi; i++) {
				int left = m + 1;
				for (i64 px = A->p[i]; px < A->p[i + 1]; px++)
					if (A->j[px] < left)
						left = A->j[px];
				if (left > m)
					continue;
				const uint64_t key = ((uint64_t) (uint32_t) weight(i) << 32) | (uint32_t) i;
				uint64_t cur = best[(size_t) left].load(std::memory_order_relaxed);
				while (key < cur && !best[(size_t) left].compare_exchange_weak(cur, key, std::memory_order_relaxed)) {
				}
			}
		};
		std::vector<std::thread> pool;
		for (int t = 1; t < T; t++)
			pool.emplace_back(scan, (int) ((i64) n * t / T), (int) ((i64) n * (t + 1) / T));
		scan(0, (int) ((i64) n / T));
		for (auto &th : pool)
			th.join();
		int found = 0;
		for (int j = 0; j < m; j++) {
			const uint64_t key = best[(size_t) j].load(std::memory_order_relaxed);
			if (key != ~0ull)
				found += take((int) (uint32_t) key, j);
		}
		return found;
	}

	int leftmost_entries()
	{
		if (A->n >= 200000 && usable_cpus() > 1)
			return leftmost_entries_threads(std::min(16, usable_cpus()));
		int found = 0;
		for (int i = 0; i < A->n; i++) {
			int best = A->m + 1;
			for (i64 px = A->p[i]; px < A->p[i + 1]; px++)
				if (A->j[px] < best)
					best = A->j[px];
			if (best > A->m)
				continue;
			if (qinv[best] == -1 || weight(i) < weight(qinv[best]))
				found += take(i, best);
		}
		return found;
	}

	int free_columns()
	{
		std::vector<char> open((size_t) (A->m > 0 ? A->m : 1), 1);
		// (large inputs, round 5: the columns of the pivotal rows are closed by threads -- everybody writes the same zero --, and
		//  so are the rows looked at that have no open column left at that point: the loop below only ever closes columns, such a
		//  row cannot take one.  On the generated families that is every row: 8 ms of two serial passes over mk15.b4's 14 M entries)
		const int T = (A->n >= 200000) ? std::max(1, std::min(16, usable_cpus())) : 1;
		std::vector<char> hopeless;
		if (T > 1) {
			hopeless.assign((size_t) A->n, 0);
			auto in_threads = [&](auto &&body) {
				std::vector<std::thread> pool;
				for (int t = 1; t < T; t++)
					pool.emplace_back(body, (int) ((i64) A->n * t / T), (int) ((i64) A->n * (t + 1) / T));
				body(0, (int) ((i64) A->n / T));
				for (auto &th : pool)
					th.join();
			};
			char *op = open.data();
			in_threads([&](int lo, int hi) {
				for (int i = lo; i < hi; i++)
					if (pinv[i] >= 0)
						for (i64 px = A->p[i]; px < A->p[i + 1]; px++)
							__atomic_store_n(&op[A->j[px]], (char) 0, __ATOMIC_RELAXED);          // (several threads may close the same column)
			});
			in_threads([&](int lo, int hi) {
				for (int i = lo; i < hi; i++) {
					bool any = false;
					if (pinv[i] < 0)
						for (i64 px = A->p[i]; px < A->p[i + 1] && !any; px++)
							any = op[A->j[px]] != 0 && qinv[A->j[px]] < 0;
					hopeless[(size_t) i] = !any;
				}
			});
		} else {
			for (int i = 0; i < A->n; i++)
				if (pinv[i] >= 0)
					for (i64 px = A->p[i]; px < A->p[i + 1]; px++)
						open[A->j[px]] = 0;
		}
		int found = 0;
		for (int i = 0; i < A->n; i++) {
			if (pinv[i] >= 0 || (T > 1 && hopeless[(size_t) i]))
				continue;
			for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
				int j = A->j[px];
				if (!open[j] || qinv[j] >= 0)
					continue;
				found += take(i, j);
				for (i64 py = A->p[i]; py < A->p[i + 1]; py++)
					open[A->j[py]] = 0;
				break;
			}
		}
		return found;
	}

	// mark[j]: 1 = candidate entry of the current row, -1 = reached / pivotal, 0 = untouched
	int acyclic_greedy()
	{
		const int n = A->n, m = A->m;
		std::vector<signed char> mark((size_t) (m > 0 ? m : 1), 0);
		std::vector<int> fifo((size_t) (m > 0 ? m : 1));
		int found = 0;
		for (int i = 0; i < n; i++) {
			if (pinv[i] >= 0)
				continue;
			int head = 0, tail = 0, candidates = 0;
			for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
				int j = A->j[px];
				if (qinv[j] < 0) {
					mark[j] = 1;
					candidates += 1;
				} else {
					fifo[tail++] = j;
					candidates -= mark[j];
					mark[j] = -1;
				}
			}
			while (head < tail && candidates > 0) {
				int row = qinv[fifo[head++]];
				if (row == -1)
					continue;
				for (i64 px = A->p[row]; px < A->p[row + 1]; px++) {
					int j = A->j[px];
					if (mark[j] >= 0) {
						fifo[tail++] = j;
						candidates -= mark[j];
						mark[j] = -1;
					}
				}
			}
			if (candidates > 0) {
				int chosen = -1;
				for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
					chosen = A->j[px];
					if (mark[chosen] == 1)
						break;
				}
				found += take(i, chosen);
			}
			for (i64 px = A->p[i]; px < A->p[i + 1]; px++)
				mark[A->j[px]] = 0;
			for (int t = 0; t < tail; t++)
				mark[fifo[t]] = 0;
		}
		return found;
	}

	// The sequential search with DEPTH LABELS (round 5; the device version and the invariant are described in pivots_device.hip):
	// every column carries a label D with D[e] > D[c] for every other entry e of the pivot row of a pivotal column c.  A
	// candidate whose label does not exceed the label of any pivotal entry of its row cannot be reached (accepted without a
	// walk); a walk only expands pivot rows whose label is below the largest label of a candidate that is still unreached; a
	// new pivot restores the invariant by raising what hangs below its row (the cascade).  Deterministic -- one row after the
	// other, in row order -- and 10-20x fewer visits than acyclic_greedy() on the generated families (mk14.b4: 1.04e9 visited
	// pivot rows -> 2.3e7 + 4.6e7 raised labels; 71 s -> 5 s), so it is what a process without a device, or SPASM_HIP_THREADS=1 on a
	// large matrix (the fixed pivot sets of bench.py), runs.  The candidate it takes is the unreached one with the smallest
	// label, not the first in the row: a valid cycle-free set, not the reference's.
	int acyclic_greedy_labels()
	{
		const int n = A->n, m = A->m;
		std::vector<int> D((size_t) (m > 0 ? m : 1), 0), pend((size_t) (m > 0 ? m : 1), 0);
		std::vector<std::pair<int, int>> items;
		// The cascade of the device search: a breadth-first list of (column, new label) items -- children = parent + 1 --, built
		// without touching the labels and applied only when it is complete.  Returns 0 when row i may take the pivot j (labels
		// raised), 1 when the cascade comes back to j itself (j IS reachable from the row, through pivots the walk did not look
		// at: see `lag` below), 2 when it outgrows `cap` items; in both cases nothing was changed.
		auto cascade = [&](int i, int j, size_t cap) -> int {          // row i gets the pivot j
			items.clear();
			const int dj = D[j];
			int outcome = 0;
			for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
				const int x = A->j[px];
				if (x != j && D[x] <= dj && pend[x] <= dj) {
					items.push_back({x, dj + 1});
					pend[x] = dj + 1;
				}
			}
			for (size_t h = 0; h < items.size() && outcome == 0; h++) {
				const int x = items[h].first, v = items[h].second;
				if (pend[x] > v || qinv[x] < 0)
					continue;
				const int r = qinv[x];
				for (i64 px = A->p[r]; px < A->p[r + 1]; px++) {
					const int e = A->j[px];
					if (e == x)
						continue;
					if (e == j) {
						outcome = 1;
						break;
					}
					if (D[e] <= v && pend[e] <= v) {
						items.push_back({e, v + 1});
						pend[e] = v + 1;
					}
				}
				if (items.size() > cap)
					outcome = 2;
			}
			for (size_t h = items.size(); h-- > 0;) {
				if (outcome == 0)
					D[items[h].first] = std::max(D[items[h].first], items[h].second);
				pend[items[h].first] = 0;
			}
			return outcome;
		};
		for (int i = 0; i < n; i++)
			if (pinv[i] >= 0)
				cascade(i, pinv[i], (size_t) -1);
		// The device runs 2,048 searches at a time: a walk does not see the pivots committed since its row was picked up, and the
		// pivots of rows that are in flight together do not chain off each other -- which is why the pivot graphs of the device
		// are a fifth as deep as those of a strictly sequential search (mk15.b4: 450-650 elimination levels against 2,500-3,100)
		// and the Schur complements that follow a quarter of the size.  To give the same kind of pivot set, deterministically, a
		// walk here only expands pivots that are at least `lag` rows old (the cascade -- which sees everything -- rejects a
		// candidate that the younger ones make reachable).
		const int lag = 8192;
		std::vector<int> born((size_t) (m > 0 ? m : 1), -0x40000000);          // row at which the pivot of a column was taken
		std::vector<signed char> mark((size_t) (m > 0 ? m : 1), 0);
		std::vector<int> fifo((size_t) (m > 0 ? m : 1));
		int found = 0;
		unsigned long long visits = 0;
		// As on the device, a candidate that sits more than `gap` labels above the lowest pivotal entry of its row is put off:
		// such pivots hang long chains under the rows that hold them (mk15.b4: 3,125 elimination levels and a Schur complement
		// of 4.2e9 entries when they are taken as they come, 500-600 levels and 0.7-1.5e9 entries when they wait), and most
		// of the rows put off find their column taken, or reached, when they are looked at again at the end.
		int gap = 64;
		if (const char *e = sh::env_get("SPASM_HIP_PIVOT_GAP"))
			gap = std::max(0, std::atoi(e));
		// ... and, as on the device, a cascade that outgrows 8,192 items puts its row off as well (these are the pivots that
		// make the graph deep: mk14.b4 has 1,900-2,150 elimination levels when every cascade is run as it comes, a few hundred
		// when the long ones wait); the rows put off are taken again with cascades of up to 65,536 items, and what is left
		// after that without a limit.
		std::vector<int> put_off, later;
		for (int pass = 0; pass < 3; pass++) {
		const size_t cap = (pass == 0) ? 8192 : (pass == 1) ? 65536 : (size_t) -1;
		if (pass == 2)
			put_off.swap(later);
		for (int i0 = 0; i0 < (pass == 0 ? n : (int) put_off.size()); i0++) {
			const int i = (pass == 0) ? i0 : put_off[(size_t) i0];
			if (pinv[i] >= 0)
				continue;
			int head = 0, tail = 0, candidates = 0, lowest = 0x7fffffff;
			for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
				const int j = A->j[px];
				if (qinv[j] < 0) {
					mark[j] = 1;
					candidates += 1;
				} else {
					fifo[tail++] = j;
					candidates -= mark[j];
					mark[j] = -1;
					lowest = std::min(lowest, D[j]);
				}
			}
			int chosen = -1;
			for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {          // on the labels alone?
				const int j = A->j[px];
				if (mark[j] == 1 && D[j] <= lowest && (chosen < 0 || D[j] < D[chosen]))
					chosen = j;
			}
			if (chosen < 0) {
				int reach = -1;
				auto largest_live_label = [&]() {
					reach = -1;
					for (i64 px = A->p[i]; px < A->p[i + 1]; px++)
						if (mark[A->j[px]] == 1)
							reach = std::max(reach, D[A->j[px]]);
				};
				largest_live_label();
				while (head < tail && candidates > 0) {
					const int c = fifo[head++];
					const int row = qinv[c];
					if (row == -1 || D[c] >= reach || (pass == 0 && born[c] + lag > i0))
						continue;
					visits += 1;
					const int before = candidates;
					for (i64 px = A->p[row]; px < A->p[row + 1]; px++) {
						const int j = A->j[px];
						if (mark[j] >= 0) {
							fifo[tail++] = j;
							candidates -= mark[j];
							mark[j] = -1;
						}
					}
					if (candidates != before)
						largest_live_label();
				}
				if (candidates > 0)
					for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
						const int j = A->j[px];
						if (mark[j] == 1 && (chosen < 0 || D[j] < D[chosen]))
							chosen = j;
					}
			}
			if (chosen >= 0 && pass == 0 && lowest != 0x7fffffff && D[chosen] - lowest > gap) {
				put_off.push_back(i);
			} else if (chosen >= 0) {
				const int outcome = cascade(i, chosen, cap);
				if (outcome == 0) {
					found += take(i, chosen);
					born[chosen] = (pass == 0) ? i0 : -0x40000000;
				} else if (pass == 0) {
					put_off.push_back(i);          // (reachable through young pivots, or a long cascade: looked at again later, with everything in sight)
				} else if (pass == 1 && outcome == 2) {
					later.push_back(i);
				}
			}
			for (i64 px = A->p[i]; px < A->p[i + 1]; px++)
				mark[A->j[px]] = 0;
			for (int t = 0; t < tail; t++)
				mark[fifo[t]] = 0;
		}
		}
		if (sh::env_get("SPASM_HIP_PIVOT_STATS"))
			logmsg("[pivots] sequential search with labels: %llu pivot rows visited for %d pivots\n", visits, found);
		return found;
	}

	// The same search with T threads and optimistic transactions (the scheme of the reference,
	// spasm_pivots.c:147-305): a thread explores its row against the pivots it can see, then commits
	// under a lock if no pivot appeared meanwhile; otherwise it replays the journal of new pivots on
	// its marks and goes on.  The set of pivots depends on timing (as in the reference); it is always
	// cycle-free.
	int acyclic_greedy_threads(int T)
	{
		const int n = A->n, m = A->m;
		std::vector<std::atomic<int>> q((size_t) (m > 0 ? m : 1));
		for (int j = 0; j < m; j++)
			q[j].store(qinv[j], std::memory_order_relaxed);
		// What a search visits is "the other columns of the pivot row of column c".  Going through q[c] -> A->p[row] ->
		// A->j[...] is three dependent cache misses per visit; the rows of these matrices are short (a boundary matrix has
		// K + 1 entries per row), so every pivotal column keeps its row -- minus itself -- in a 32-byte record: one miss.
		// A column becomes pivotal once and stays so: the writer fills the entries, then publishes the length (release);
		// len 0 = no pivot, len < 0 = row too long for a record (its index is in ent[0]).
		struct alignas(32) PivRec {
			std::atomic<int> len;
			int ent[7];
		};
		std::vector<PivRec> rec((size_t) (m > 0 ? m : 1));
		auto publish = [&](int col, int row) {
			PivRec &R = rec[col];
			const i64 lo = A->p[row], hi = A->p[row + 1];
			if (hi - lo - 1 > 7) {
				R.ent[0] = row;
				R.len.store(-1, std::memory_order_release);
				return;
			}
			int k = 0;
			for (i64 px = lo; px < hi; px++)
				if (A->j[px] != col)
					R.ent[k++] = A->j[px];
			if (k == 0) {                    // (a row with nothing but its pivot: nothing to visit, but the column IS pivotal)
				R.ent[0] = row;
				R.len.store(-1, std::memory_order_release);
				return;
			}
			R.len.store(k, std::memory_order_release);
		};
		for (int j = 0; j < m; j++) {
			rec[j].len.store(0, std::memory_order_relaxed);
			if (qinv[j] >= 0)
				publish(j, qinv[j]);
		}
		std::vector<int> journal((size_t) (n > 0 ? n : 1));
		std::atomic<int> npiv{0};
		std::atomic<int> next_row{0};
		std::mutex commit;
		std::atomic<unsigned long long> total_visits{0}, total_retries{0};
		auto worker = [&]() {
			unsigned long long visits = 0, retries = 0;
			std::vector<signed char> mark((size_t) (m > 0 ? m : 1), 0);
			std::vector<int> fifo((size_t) (m > 0 ? m : 1));
			for (;;) {
				const int begin = next_row.fetch_add(256);
				if (begin >= n)
					break;
				const int end = std::min(n, begin + 256);
				for (int i = begin; i < end; i++) {
					if (pinv[i] >= 0)
						continue;
					int seen = npiv.load(std::memory_order_acquire);
					int head = 0, tail = 0, candidates = 0;
					auto push = [&](int j) {
						fifo[tail++] = j;
						candidates -= mark[j];
						mark[j] = -1;
					};
					for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
						const int j = A->j[px];
						if (rec[j].len.load(std::memory_order_acquire) == 0) {
							mark[j] = 1;
							candidates += 1;
						} else {
							push(j);
						}
					}
					// the other columns of the pivot row of column c become reached
					auto expand = [&](int c) {
						const PivRec &R = rec[c];
						const int len = R.len.load(std::memory_order_acquire);
						if (len == 0)
							return;                  // not a pivot (a reached non-pivotal column)
						visits += 1;
						if (len > 0) {
							for (int e = 0; e < len; e++) {
								const int j = R.ent[e];
								if (mark[j] >= 0)
									push(j);
							}
						} else {
							const int row = R.ent[0];
							for (i64 px = A->p[row]; px < A->p[row + 1]; px++) {
								const int j = A->j[px];
								if (mark[j] >= 0)
									push(j);
							}
						}
					};
					// a pivot that appeared behind our back only matters when its column is one we marked: a candidate of
					// ours that became pivotal, or a column of a row we reached (whose new row must then be explored).
					// Pivots on untouched columns cannot be reached from our row and leave the search as it stands.
					auto absorb = [&](int jn) {
						if (mark[jn] == 0)
							return false;
						if (mark[jn] == 1)
							push(jn);
						else
							expand(jn);
						return true;
					};
					for (;;) {
						while (head < tail && candidates > 0) {
							if (head + 4 < tail)
								__builtin_prefetch(&rec[fifo[head + 4]]);
							expand(fifo[head++]);
						}
						if (candidates == 0)
							break;
						retries += 1;
						// catch up with the journal WITHOUT the lock; go back to the search if any of it touched us
						bool touched = false;
						for (const int target = npiv.load(std::memory_order_acquire); seen < target; seen++)
							touched |= absorb(journal[seen]);
						if (touched)
							continue;
						int chosen = -1;
						for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
							chosen = A->j[px];
							if (mark[chosen] == 1)
								break;
						}
						// commit: under the lock only the few pivots that arrived since the catch-up are looked at, and the
						// commit goes through unless one of THEM touches this search (the reference retries whenever
						// anything at all was committed meanwhile, spasm_pivots.c:262-290)
						bool done = false;
						{
							std::lock_guard<std::mutex> lock(commit);
							const int now = npiv.load(std::memory_order_relaxed);
							bool clean = true;
							for (int t = seen; t < now; t++)
								if (mark[journal[t]] != 0) {
									clean = false;
									break;
								}
							if (clean) {
								q[chosen].store(i, std::memory_order_relaxed);
								pinv[i] = chosen;
								publish(chosen, i);
								journal[now] = chosen;
								npiv.store(now + 1, std::memory_order_release);
								done = true;
							}
						}
						if (done)
							break;
					}
					for (i64 px = A->p[i]; px < A->p[i + 1]; px++)
						mark[A->j[px]] = 0;
					for (int t = 0; t < tail; t++)
						mark[fifo[t]] = 0;
				}
			}
			total_visits += visits;
			total_retries += retries;
		};
		std::vector<std::thread> pool;
		for (int t = 0; t < T; t++)
			pool.emplace_back(worker);
		for (auto &th : pool)
			th.join();
		if (sh::env_get("SPASM_HIP_PIVOT_STATS"))
			logmsg("[pivots] %d threads: %llu pivot rows visited, %llu commit attempts for %d pivots\n", T,
			       total_visits.load(), total_retries.load(), npiv.load());
		for (int j = 0; j < m; j++)
			qinv[j] = q[j].load(std::memory_order_relaxed);
		return npiv.load();
	}

	// is the order p of the pivotal rows triangular: does every pivotal row only touch pivot columns of rows after it?
	bool triangular(int npiv, const int *p) const
	{
		std::vector<int> when((size_t) (A->n > 0 ? A->n : 1), -1);
		for (int t = 0; t < npiv; t++)
			when[p[t]] = t;
		std::atomic<bool> fine{true};
		auto check = [&](int t_lo, int t_hi) {
			for (int t = t_lo; t < t_hi && fine.load(std::memory_order_relaxed); t++) {
				const int i = p[t];
				bool ok = pinv[i] >= 0;
				for (i64 px = A->p[i]; ok && px < A->p[i + 1]; px++) {
					const int j = A->j[px];
					if (j == pinv[i] || qinv[j] < 0)
						continue;
					ok = when[qinv[j]] > t;
				}
				if (!ok)
					fine.store(false, std::memory_order_relaxed);
			}
		};
		const int T = (npiv < 20000) ? 1 : std::max(1, std::min(8, usable_cpus()));
		std::vector<std::thread> pool;
		for (int t = 1; t < T; t++)
			pool.emplace_back(check, (int) ((i64) npiv * t / T), (int) ((i64) npiv * (t + 1) / T));
		check(0, (int) ((i64) npiv / T));
		for (auto &th : pool)
			th.join();
		return fine.load();
	}

	// reverse post-order of the pivot graph: a pivotal column precedes all
	// the pivotal columns its row touches
	void topological_rows(int npiv, int *p) const
	{
		const int n = A->n, m = A->m;
		std::vector<int> order;
		order.reserve((size_t) m);
		std::vector<char> seen((size_t) (m > 0 ? m : 1), 0);
		std::vector<int> stack_col, stack_pos;
		// The walk goes column -> its pivot row -> the columns of that row: qinv[j], A->p[i], A->j[...] are three dependent cache
		// misses per node.  As in the search (PivRec), a pivotal column keeps the columns of its row -- all of them, in row order,
		// the walk skips what it has seen -- in a 32-byte record when they are at most seven: one miss.  cnt -1: not pivotal;
		// -2: the row is longer (its index is in ent[0]).  The records are filled row by row, by a few threads on large inputs.
		struct alignas(32) Rec {
			int cnt;
			int ent[7];
		};
		std::vector<Rec> rec((size_t) (m > 0 ? m : 1));
		for (int j = 0; j < m; j++)
			rec[j].cnt = -1;
		auto fill = [&](int i_lo, int i_hi) {
			for (int i = i_lo; i < i_hi; i++) {
				const int j = pinv[i];
				if (j < 0)
					continue;
				Rec &R = rec[j];
				const int w = weight(i);
				if (w > 7) {
					R.cnt = -2;
					R.ent[0] = i;
				} else {
					for (int k = 0; k < w; k++)
						R.ent[k] = A->j[A->p[i] + k];
					R.cnt = w;
				}
			}
		};
		{
			const int T = (npiv < 20000) ? 1 : std::max(1, std::min(8, usable_cpus()));
			std::vector<std::thread> pool;
			for (int t = 1; t < T; t++)
				pool.emplace_back(fill, (int) ((i64) n * t / T), (int) ((i64) n * (t + 1) / T));
			fill(0, (int) ((i64) n / T));
			for (auto &th : pool)
				th.join();
		}
		for (int j0 = 0; j0 < m; j0++) {
			if (rec[j0].cnt == -1 || seen[j0])
				continue;
			stack_col.assign(1, j0);
			stack_pos.assign(1, 0);
			seen[j0] = 1;
			while (!stack_col.empty()) {
				const int j = stack_col.back();
				const Rec &R = rec[j];
				bool down = false;
				if (R.cnt != -1) {
					const int *ents = R.ent;
					int w = R.cnt;
					if (w == -2) {
						const int i = R.ent[0];
						ents = A->j + A->p[i];
						w = weight(i);
					}
					for (int k = stack_pos.back(); k < w; k++) {
						const int jj = ents[k];
						if (seen[jj])
							continue;
						stack_pos.back() = k + 1;
						seen[jj] = 1;
						stack_col.push_back(jj);
						stack_pos.push_back(0);
						down = true;
						break;
					}
				}
				if (!down) {
					order.push_back(j);
					stack_col.pop_back();
					stack_pos.pop_back();
				}
			}
		}
		// the reference fills its output from the end: later finishes come first
		// within one tree, later trees come before earlier ones
		int k = 0;
		for (size_t t = order.size(); t-- > 0;) {
			int i = qinv[order[t]];
			if (i != -1)
				p[k++] = i;
		}
		if (k != npiv)
			die("pivot reordering lost pivots (%d != %d)", k, npiv);
		for (int i = 0; i < n; i++)
			if (pinv[i] == -1)
				p[k++] = i;
	}
};

}  // namespace

extern "C" int spasm_hip_usable_cpus(void) { return usable_cpus(); }

extern "C" int spasm_hip_pivots_extract_structural(const struct spasm_csr *A, const int *p_in, struct spasm_lu *fact,
                                                   int *p, struct echelonize_opts *opts)
{
	const int n = A->n, m = A->m;
	const i64 prime = A->field->p;
	double t0 = wtime();
	Search S;
	S.A = A;
	S.pinv.assign((size_t) (n > 0 ? n : 1), -1);
	S.qinv.assign((size_t) (m > 0 ? m : 1), -1);
	// one process per GPU (dist_api.hip): the search is threaded and its outcome depends on timing, so rank 0 searches
	// and every rank gets its result -- the pivots (pinv) and the row order (p); the rows of U follow from them
	spasm_hip_comm *comm = current_comm();
	const bool dist = comm != nullptr && comm_world(comm) > 1;
	int npiv = 0;
	std::vector<int> hint_height;          // heights of the new rows of U from the labels of the device search (see level_hint_set)
	if (!dist || comm_rank(comm) == 0) {
		bool ordered = false;
		double t_fl = 0.0, t_greedy = 0.0, t_device = 0.0;
		// (round 5, late) the matrix starts for the device now, beside the Faugere-Lachartre steps: the greedy search on the device is
		// the first stage that wants it there, and the copy from pageable host memory took 1-36 ms of its time
		std::thread upload;
		{
			const char *where = sh::env_get("SPASM_HIP_PIVOT_SEARCH");
			int threads = 0;
			if (const char *e = sh::env_get("SPASM_HIP_THREADS"))
				threads = std::atoi(e);
			if ((opts == nullptr || opts->enable_greedy_pivot_search) && A->n >= 20000 && m <= (1 << 25) && threads != 1 && (where == nullptr || std::strcmp(where, "host") != 0) &&
			    resident_prefetch_possible() && usable_cpus() > 1)
			{
				const int dev = resident_current_device();
				upload = std::thread([A, dev]() { resident_prefetch_matrix(A, dev); });
			}
		}
		struct Joiner {
			std::thread &t;
			~Joiner()
			{
				if (t.joinable())
					t.join();
			}
		} joiner{upload};
		npiv = S.leftmost_entries();
		logmsg("[pivots] Faugere-Lachartre: %d pivots found [%.1fs]\n", npiv, wtime() - t0);
		double t1 = wtime();
		int extra = S.free_columns();
		npiv += extra;
		logmsg("[pivots] Faugere-Lachartre on columns: %d pivots found [%.1fs]\n", extra, wtime() - t1);
		t_fl = wtime() - t0;
		if (opts == nullptr || opts->enable_greedy_pivot_search) {
			t1 = wtime();
			int threads = 0;
			if (const char *e = sh::env_get("SPASM_HIP_THREADS"))
				threads = std::atoi(e);
			if (threads <= 0)
				threads = usable_cpus();
			if (A->n < 20000)
				threads = 1;                  // small inputs: the sequential search (deterministic) is as fast
			// on the device when there is one (pivots_device.hip); what comes back is checked -- the order below must be
			// triangular -- before anything is built on it
			std::vector<int> pinv0, qinv0;
			if (threads > 1) {
				pinv0 = S.pinv;
				qinv0 = S.qinv;
			}
			std::vector<int> col_label;
			if (upload.joinable())
				upload.join();
			extra = (threads > 1) ? device_acyclic_greedy(A, S.pinv.data(), S.qinv.data(), &col_label) : -1;
			t_device = wtime() - t1;
			bool by_labels = extra >= 0 && !col_label.empty();
			if (by_labels) {
				// The device hands the depth label of every column: a pivotal row only touches pivot columns with LARGER labels than
				// its own pivot's (that the sweeps which computed them ended is the proof that the set is cycle-free), so the pivotal
				// rows sorted by the label of their pivot are in triangular order -- a counting sort instead of the depth-first
				// search and the check below (35 + 8 ms on mk15.b4).  SPASM_HIP_PIVOT_CHECK=1: checked on the host all the same.
				const double ta = wtime();
				int top = 0;
				for (int j = 0; j < m; j++)
					if (S.qinv[j] >= 0)
						top = std::max(top, col_label[(size_t) j]);
				std::vector<int> start((size_t) top + 2, 0);
				for (int j = 0; j < m; j++)
					if (S.qinv[j] >= 0)
						start[(size_t) col_label[(size_t) j] + 1] += 1;
				for (int l = 0; l <= top; l++)
					start[(size_t) l + 1] += start[(size_t) l];
				int placed = 0;
				for (int j = 0; j < m; j++)
					if (S.qinv[j] >= 0) {
						p[start[(size_t) col_label[(size_t) j]]++] = S.qinv[j];
						placed += 1;
					}
				if (placed != npiv + extra)
					die("pivot ordering by labels lost pivots (%d != %d)", placed, npiv + extra);
				hint_height.resize((size_t) placed);
				for (int t = 0; t < placed; t++)
					hint_height[(size_t) t] = top - col_label[(size_t) S.pinv[p[t]]];
				for (int i = 0; i < n; i++)
					if (S.pinv[i] == -1)
						p[placed++] = i;
				ordered = true;
				const double tb = wtime();
				// always, and cheap (a threaded pass over the pivotal rows): the property the counting sort rests on -- every other
				// pivotal entry of a pivot row carries a larger label than its pivot.  A search that reported `settled` short of the
				// fixpoint would have handed a mis-ordered U to everything downstream without a word.
				bool fine = true;
				{
					std::atomic<int> bad{0};
					const int T = std::max(1, std::min(16, usable_cpus()));
					sh::pool_run(T, [&](int t) {
						const int lo = (int) ((i64) n * t / T), hi = (int) ((i64) n * (t + 1) / T);
						for (int i = lo; i < hi && bad.load(std::memory_order_relaxed) == 0; i++) {
							const int c = S.pinv[i];
							if (c < 0)
								continue;
							const int lc = col_label[(size_t) c];
							for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
								const int e = A->j[px];
								if (e != c && S.qinv[e] >= 0 && col_label[(size_t) e] <= lc) {
									bad.store(1, std::memory_order_relaxed);
									break;
								}
							}
						}
					});
					fine = bad.load() == 0;
				}
				if (fine && sh::env_get("SPASM_HIP_PIVOT_CHECK"))
					fine = S.triangular(npiv + extra, p);
				if (verbose() >= 3)
					logmsg("[pivots] order by labels %.1f ms (%d levels)%s\n", 1e3 * (tb - ta), top + 1, sh::env_get("SPASM_HIP_PIVOT_CHECK") ? (fine ? ", checked on the host" : ", NOT triangular") : "");
				if (!fine) {
					std::fprintf(stderr, "[pivots] the depth labels of the device search do not order its pivots (a bug: please report); ordering them by a search on the host\n");
					by_labels = false;
					ordered = false;
					hint_height.clear();
				}
			}
			if (extra >= 0 && !by_labels) {
				const double ta = wtime();
				S.topological_rows(npiv + extra, p);
				const double tb = wtime();
				const bool fine = S.triangular(npiv + extra, p);
				if (verbose() >= 3)
					logmsg("[pivots] order %.1f ms, check %.1f ms\n", 1e3 * (tb - ta), 1e3 * (wtime() - tb));
				if (!fine) {
					std::fprintf(stderr, "[pivots] the pivots of the device search are NOT cycle-free (a bug: please report); discarded, searching on the host\n");
					S.pinv = pinv0;
					S.qinv = qinv0;
					extra = -1;
				} else {
					ordered = true;
				}
			}
			if (extra < 0) {
				const char *where = sh::env_get("SPASM_HIP_PIVOT_SEARCH");
				if (threads > 1 && where != nullptr && std::strcmp(where, "device") == 0)
					die("SPASM_HIP_PIVOT_SEARCH=device: the search on the device does not apply to this %d x %d matrix here", n, m);
				// (one thread on a large matrix -- SPASM_HIP_THREADS=1: a pivot set that does not depend on timing -- takes the labelled
				//  sequential search; small inputs keep the row-order search whose outcome is the reference's with one thread)
				const char *lab = sh::env_get("SPASM_HIP_PIVOT_LABELS");
				const bool labelled = A->n >= 20000 && !(lab != nullptr && std::atoi(lab) == 0);
				extra = (threads > 1) ? S.acyclic_greedy_threads(threads) : labelled ? S.acyclic_greedy_labels() : S.acyclic_greedy();
			}
			npiv += extra;
			t_greedy = wtime() - t1;
			logmsg("[pivots] greedy alternating cycle-free search: %d pivots found [%.1fs]\n", extra, wtime() - t1);
		}
		logmsg("[pivots] %d pivots found\n", npiv);
		const double t_order = wtime();
		if (!ordered)
			S.topological_rows(npiv, p);
		if (verbose() >= 2)
			logmsg("[pivots] Faugere-Lachartre steps %.1f ms, greedy search %.1f ms (%.1f on the device, then order + check), order %.1f ms\n", 1e3 * t_fl, 1e3 * t_greedy, 1e3 * t_device,
			       1e3 * (wtime() - t_order));

	}
	if (dist) {
		comm_bcast_host(comm, &npiv, sizeof(int), 0);
		comm_bcast_host(comm, p, (size_t) n * sizeof(int), 0);
		comm_bcast_host(comm, S.pinv.data(), (size_t) n * sizeof(int), 0);
	}

	const double t_searched = wtime();
	const uint64_t pu64 = (uint64_t) prime, barrett = ~0ull / pu64;          // x mod p = x - floor(x * floor(2^64 / p) / 2^64) * p, or p more
	struct spasm_csr *U = fact->U;
	struct spasm_triplet *L = fact->Ltmp;
	const i64 unz = U->p[U->n];
	// where every new row goes, then the rows themselves -- by a few threads on large inputs: a row is a gather from A (its
	// extent, its columns, its values: three cache misses, asked for a few rows ahead), and so are the lengths (round 5: the
	// two serial passes over the pivots before the threaded one -- 600,000 random looks at A->p on mk15.b4 -- took as long as it)
	const int n0 = U->n;
	const int T_rows = (npiv < 20000) ? 1 : std::max(1, std::min(16, usable_cpus()));
	auto in_threads = [&](auto &&body) {
		std::vector<std::thread> pool;
		for (int t = 1; t < T_rows; t++)
			pool.emplace_back(body, (int) ((i64) npiv * t / T_rows), (int) ((i64) npiv * (t + 1) / T_rows));
		body(0, (int) ((i64) npiv / T_rows));
		for (auto &th : pool)
			th.join();
	};
	in_threads([&](int t_lo, int t_hi) {
		for (int t = t_lo; t < t_hi; t++) {
			if (t + 8 < t_hi)
				__builtin_prefetch(&A->p[p[t + 8]]);
			const int i = p[t];
			fact->qinv[S.pinv[i]] = n0 + t;
			U->p[n0 + t + 1] = S.weight(i);          // (lengths for now)
		}
	});
	i64 need = 0;
	for (int t = 0; t < npiv; t++)
		need += U->p[n0 + t + 1];
	if (unz + need > U->nzmax)
		spasm_hip_csr_realloc(U, unz + need);
	for (int t = 0; t < npiv; t++)
		U->p[n0 + t + 1] += U->p[n0 + t];
	auto pivot_of = [&](int i, int j) -> spasm_ZZp {
		for (i64 px = A->p[i]; px < A->p[i + 1]; px++)
			if (A->j[px] == j && A->x[px] != 0)
				return A->x[px];
		die("structural pivot (%d, %d) has no value", i, j);
		return 0;
	};
	if (L != nullptr)
		for (int t = 0; t < npiv; t++) {
			const int i = p[t];
			const int i_out = (p_in != nullptr) ? p_in[i] : i;
			spasm_hip_add_entry(L, i_out, n0 + t, pivot_of(i, S.pinv[i]));
			fact->p[n0 + t] = i_out;
		}
	auto fill_rows = [&](int t_lo, int t_hi) {
		for (int t = t_lo; t < t_hi; t++) {
			if (t + 16 < t_hi)
				__builtin_prefetch(&A->p[p[t + 16]]);
			if (t + 8 < t_hi) {
				const i64 pf = A->p[p[t + 8]];
				__builtin_prefetch(&A->j[pf]);
				__builtin_prefetch(&A->x[pf]);
			}
			const int i = p[t];
			const int j = S.pinv[i];
			const spasm_ZZp pivot = pivot_of(i, j);
			// (boundary matrices have pivots +-1: no inverse, no product; otherwise one reduction per entry without a division)
			const spasm_ZZp scale = (pivot == 1) ? 1 : (pivot == -1) ? -1 : zp_inverse(prime, pivot);
			i64 w = U->p[n0 + t];
			U->j[w] = j;
			U->x[w] = 1;
			w += 1;
			for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
				if (A->j[px] == j)
					continue;
				U->j[w] = A->j[px];
				const spasm_ZZp a = A->x[px];
				spasm_ZZp v;
				if (scale == 1) {
					v = a;
				} else if (scale == -1 && a != -a) {
					v = -a;
				} else {
					const int64_t prod = (int64_t) scale * a;
					const uint64_t u = (uint64_t) (prod < 0 ? -prod : prod);
					uint64_t rem = u - (uint64_t) (((unsigned __int128) u * barrett) >> 64) * pu64;
					while (rem >= pu64)
						rem -= pu64;
					if (prod < 0 && rem != 0)
						rem = pu64 - rem;
					v = zp_balance(prime, (int64_t) rem);
				}
				U->x[w] = v;
				w += 1;
			}
		}
	};
	in_threads(fill_rows);
	U->n = n0 + npiv;
	if (n0 == 0 && !dist && (int) hint_height.size() == npiv && npiv > 0)
		level_hint_set(U, npiv, std::move(hint_height));
	if (verbose() >= 2)
		logmsg("[pivots] search and order %.1f ms, rows of U %.1f ms\n", 1e3 * (t_searched - t0), 1e3 * (wtime() - t_searched));
	return npiv;
}
