// Entry points of include/spasm_hip.h that are not implemented yet.  They
// exist so that the ABI is complete and fail loudly (no silent fallback).
#include "common.h"

using namespace sh;

#define NOT_YET(name) die(name ": not implemented yet on the GPU path")

extern "C" {

void spasm_hip_schur_dense_randomized(const struct spasm_csr *, const int *, int, const struct spasm_csr *,
                                      const int *, void *, spasm_datatype, int *, int, int)
{
	NOT_YET("spasm_hip_schur_dense_randomized");
}

int spasm_hip_ffpack_LU(i64, int, int, void *, int, spasm_datatype, size_t *, size_t *) { NOT_YET("spasm_hip_ffpack_LU"); }

struct spasm_csr *spasm_hip_kernel(const struct spasm_lu *) { NOT_YET("spasm_hip_kernel"); }

double spasm_hip_schur_estimate_density(const struct spasm_csr *, const int *, int, const struct spasm_csr *, const int *, int)
{
	NOT_YET("spasm_hip_schur_estimate_density");
}



struct spasm_lu *spasm_hip_echelonize(const struct spasm_csr *, struct echelonize_opts *) { NOT_YET("spasm_hip_echelonize"); }

struct spasm_csr *spasm_hip_rref(const struct spasm_lu *, int *) { NOT_YET("spasm_hip_rref"); }



}  // extern "C"
