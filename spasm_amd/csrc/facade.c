/*
 * libspasm_hip_facade.so -- the hot path under the reference's OWN symbol names.
 *
 * An unmodified object file of cbouilla/spasm (tools/rank.o, tools/kernel.o, ...) that is linked against this
 * library before the reference's libspasm gets spasm_echelonize, spasm_schur*, spasm_ffpack_*, spasm_rref and
 * spasm_kernel from the GPU; everything else (certificates, solvers, Dulmage-Mendelsohn, I/O) stays the
 * reference's.  Signatures: src/spasm.h (cited per function in include/spasm_hip.h).
 */
#include "../../include/spasm_hip.h"

void spasm_echelonize_init_opts(struct echelonize_opts *opts) { spasm_hip_echelonize_init_opts(opts); }
struct spasm_lu *spasm_echelonize(const struct spasm_csr *A, struct echelonize_opts *opts) { return spasm_hip_echelonize(A, opts); }

struct spasm_csr *spasm_schur(const struct spasm_csr *A, const int *p, int n, const struct spasm_lu *fact, double est_density,
                              struct spasm_triplet *L, const int *p_in, int *p_out)
{
	return spasm_hip_schur(A, p, n, fact, est_density, L, p_in, p_out);
}

double spasm_schur_estimate_density(const struct spasm_csr *A, const int *p, int n, const struct spasm_csr *U, const int *qinv, int R)
{
	return spasm_hip_schur_estimate_density(A, p, n, U, qinv, R);
}

void spasm_schur_dense(const struct spasm_csr *A, const int *p, int n, const int *p_in, struct spasm_lu *fact, void *S,
                       spasm_datatype datatype, int *q, int *p_out)
{
	spasm_hip_schur_dense(A, p, n, p_in, fact, S, datatype, q, p_out);
}

void spasm_schur_dense_randomized(const struct spasm_csr *A, const int *p, int n, const struct spasm_csr *U, const int *qinv,
                                  void *S, spasm_datatype datatype, int *q, int N, int w)
{
	spasm_hip_schur_dense_randomized(A, p, n, U, qinv, S, datatype, q, N, w);
}

int spasm_ffpack_rref(i64 prime, int n, int m, void *A, int ldA, spasm_datatype datatype, size_t *qinv)
{
	return spasm_hip_ffpack_rref(prime, n, m, A, ldA, datatype, qinv);
}

int spasm_ffpack_LU(i64 prime, int n, int m, void *A, int ldA, spasm_datatype datatype, size_t *p, size_t *qinv)
{
	return spasm_hip_ffpack_LU(prime, n, m, A, ldA, datatype, p, qinv);
}

spasm_ZZp spasm_datatype_read(const void *A, size_t i, spasm_datatype datatype) { return spasm_hip_datatype_read(A, i, datatype); }
void spasm_datatype_write(void *A, size_t i, spasm_datatype datatype, spasm_ZZp value) { spasm_hip_datatype_write(A, i, datatype, value); }
size_t spasm_datatype_size(spasm_datatype datatype) { return spasm_hip_datatype_size(datatype); }
spasm_datatype spasm_datatype_choose(i64 prime) { return spasm_hip_datatype_choose(prime); }
const char *spasm_datatype_name(spasm_datatype datatype) { return spasm_hip_datatype_name(datatype); }

struct spasm_csr *spasm_rref(const struct spasm_lu *fact, int *Rqinv) { return spasm_hip_rref(fact, Rqinv); }
struct spasm_csr *spasm_kernel(const struct spasm_lu *fact) { return spasm_hip_kernel(fact); }
