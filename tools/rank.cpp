// rank: drop-in for the reference's tools/rank (tools/rank.c + tools/common.c): same options,
// reads an SMS / MatrixMarket matrix (stdin or --matrix), prints "rank = N" on stderr like the
// reference.  Everything heavy runs on the GPU through libspasm_hip.so.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/time.h>

#include "spasm_hip.h"

static double now()
{
	struct timeval tv;
	gettimeofday(&tv, nullptr);
	return tv.tv_sec + 1e-6 * tv.tv_usec;
}

int main(int argc, char **argv)
{
	struct echelonize_opts opts;
	spasm_hip_echelonize_init_opts(&opts);
	const char *filename = nullptr;
	i64 prime = 42013;
	bool allow_transpose = true;
	enum { NO_LOW_RANK = 1000, NO_DENSE, NO_GPLU, MAX_ITER, DENSE_THR, MIN_PIV, DENSE_BLK, MIN_RANK, MAX_ASPECT, NO_GREEDY };
	static struct option longopts[] = {
		{"matrix", required_argument, nullptr, 'm'},
		{"modulus", required_argument, nullptr, 'p'},
		{"no-transpose", no_argument, nullptr, 't'},
		{"no-low-rank-mode", no_argument, nullptr, NO_LOW_RANK},
		{"no-dense-mode", no_argument, nullptr, NO_DENSE},
		{"no-GPLU", no_argument, nullptr, NO_GPLU},
		{"no-greedy-pivot-search", no_argument, nullptr, NO_GREEDY},
		{"max-iterations", required_argument, nullptr, MAX_ITER},
		{"dense-threshold", required_argument, nullptr, DENSE_THR},
		{"min-pivot-proportion", required_argument, nullptr, MIN_PIV},
		{"dense-block-size", required_argument, nullptr, DENSE_BLK},
		{"min-rank-ratio", required_argument, nullptr, MIN_RANK},
		{"max-aspect-ratio", required_argument, nullptr, MAX_ASPECT},
		{nullptr, 0, nullptr, 0}};
	int ch;
	while ((ch = getopt_long(argc, argv, "m:p:tc", longopts, nullptr)) != -1) {
		switch (ch) {
		case 'm': filename = optarg; break;
		case 'p': prime = atoll(optarg); break;
		case 't': allow_transpose = false; break;
		case 'c':
			fprintf(stderr, "rank certificates are built by the reference's own code (spasm_certificate.c) on top of the L and U computed "
			                "here:\nlink the reference's tools/rank.c against libspasm_hip_facade.so (INTEGRATION.md, option B) and run "
			                "that with --certificate\n");
			return 1;
		case NO_LOW_RANK: opts.enable_tall_and_skinny = 0; break;
		case NO_DENSE: opts.enable_dense = 0; break;
		case NO_GPLU: opts.enable_GPLU = 0; break;
		case NO_GREEDY: opts.enable_greedy_pivot_search = 0; break;
		case MAX_ITER: opts.max_round = atoi(optarg); break;
		case DENSE_THR: opts.sparsity_threshold = atof(optarg); break;
		case MIN_PIV: opts.min_pivot_proportion = atof(optarg); break;
		case DENSE_BLK: opts.dense_block_size = atoi(optarg); break;
		case MIN_RANK: opts.low_rank_ratio = atof(optarg); break;
		case MAX_ASPECT: opts.tall_and_skinny_ratio = atof(optarg); break;
		default: fprintf(stderr, "unknown option\n"); return 1;
		}
	}
	FILE *f = stdin;
	if (filename != nullptr) {
		f = fopen(filename, "r");
		if (f == nullptr) {
			perror(filename);
			return 1;
		}
	}
	u8 hash[32];
	struct spasm_triplet *T = spasm_hip_triplet_load(f, prime, hash);
	if (f != stdin)
		fclose(f);
	if (allow_transpose && T->n < T->m) {
		fprintf(stderr, "[rank] transposing matrix\n");
		spasm_hip_triplet_transpose(T);
	}
	struct spasm_csr *A = spasm_hip_compress(T);
	spasm_hip_triplet_free(T);
	fprintf(stderr, "start. A is %d x %d (%lld nnz)\n", A->n, A->m, (long long) A->p[A->n]);
	double t0 = now();
	struct spasm_lu *fact = spasm_hip_echelonize(A, &opts);
	fprintf(stderr, "done in %.3f s rank = %d\n", now() - t0, fact->U->n);
	printf("%d\n", fact->U->n);
	spasm_hip_lu_free(fact);
	spasm_hip_csr_free(A);
	return 0;
}
