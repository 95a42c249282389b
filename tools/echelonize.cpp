// echelonize: drop-in for the reference's tools/echelonize (tools/echelonize.c): reads an SMS
// matrix on stdin, writes the echelon form U (or, with --rref, the RREF of A*Q) in SMS format.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>

#include "spasm_hip.h"

int main(int argc, char **argv)
{
	struct echelonize_opts opts;
	spasm_hip_echelonize_init_opts(&opts);
	i64 prime = 42013;
	bool want_rref = false;
	static struct option longopts[] = {
		{"modulus", required_argument, nullptr, 'p'},
		{"rref", no_argument, nullptr, 'r'},
		{"no-greedy-pivot-search", no_argument, nullptr, 'g'},
		{"no-low-rank-mode", no_argument, nullptr, 'l'},
		{"dense-block-size", required_argument, nullptr, 'd'},
		{"low-rank-start-weight", required_argument, nullptr, 'w'},
		{"sparsity-threshold", required_argument, nullptr, 's'},
		{nullptr, 0, nullptr, 0}};
	int ch;
	while ((ch = getopt_long(argc, argv, "", longopts, nullptr)) != -1) {
		switch (ch) {
		case 'p': prime = atoll(optarg); break;
		case 'r': want_rref = true; break;
		case 'g': opts.enable_greedy_pivot_search = 0; break;
		case 'l': opts.enable_tall_and_skinny = 0; break;
		case 'd': opts.dense_block_size = atoi(optarg); break;
		case 'w': opts.low_rank_start_weight = atoi(optarg); break;
		case 's': opts.sparsity_threshold = atof(optarg); break;
		default: fprintf(stderr, "Unknown option\n"); return 1;
		}
	}
	struct spasm_triplet *T = spasm_hip_triplet_load(stdin, prime, nullptr);
	struct spasm_csr *A = spasm_hip_compress(T);
	spasm_hip_triplet_free(T);
	const int m = A->m;
	struct spasm_lu *fact = spasm_hip_echelonize(A, &opts);
	spasm_hip_csr_free(A);
	if (want_rref) {
		int *Rqinv = (int *) malloc((size_t) (m > 0 ? m : 1) * sizeof(int));
		struct spasm_csr *R = spasm_hip_rref(fact, Rqinv);
		spasm_hip_csr_save(R, stdout);
		spasm_hip_csr_free(R);
		free(Rqinv);
	} else {
		spasm_hip_csr_save(fact->U, stdout);
	}
	spasm_hip_lu_free(fact);
	return 0;
}
