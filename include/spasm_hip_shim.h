/*
 * spasm_hip_shim.h -- compile an UNMODIFIED program of cbouilla/spasm against libspasm_hip.so.
 *
 *     cc -include <this repo>/include/spasm_hip_shim.h -I<reference>/src tools/echelonize.c \
 *        -L<this repo>/spasm_amd/csrc -lspasm_hip
 *
 * The reference's own spasm.h is included first, so ITS struct definitions and prototypes are the ones the program
 * is compiled against; every function on the echelonization path (and the containers / I/O a tool needs around it)
 * is then renamed to its spasm_hip_ twin, whose prototype in spasm_hip.h has to agree with the reference's -- a
 * mismatch is a compile error here.  tests/test_host.py builds the reference's tools/echelonize.c this way;
 * tests/test_gpu_tools.py runs the result on the GPU.
 */
#ifndef SPASM_HIP_SHIM_H
#define SPASM_HIP_SHIM_H

#include "spasm.h"          /* the reference's (-I<reference>/src) */
#include "spasm_hip.h"

/* the hot path (spasm_echelonize.c, spasm_schur.c, spasm_ffpack.cpp, spasm_rref.c, spasm_kernel.c) */
#define spasm_echelonize_init_opts    spasm_hip_echelonize_init_opts
#define spasm_echelonize              spasm_hip_echelonize
#define spasm_schur                   spasm_hip_schur
#define spasm_schur_estimate_density  spasm_hip_schur_estimate_density
#define spasm_schur_dense             spasm_hip_schur_dense
#define spasm_schur_dense_randomized  spasm_hip_schur_dense_randomized
#define spasm_ffpack_rref             spasm_hip_ffpack_rref
#define spasm_ffpack_LU               spasm_hip_ffpack_LU
#define spasm_datatype_read           spasm_hip_datatype_read
#define spasm_datatype_write          spasm_hip_datatype_write
#define spasm_datatype_size           spasm_hip_datatype_size
#define spasm_datatype_choose         spasm_hip_datatype_choose
#define spasm_datatype_name           spasm_hip_datatype_name
#define spasm_rref                    spasm_hip_rref
#define spasm_kernel                  spasm_hip_kernel
#define spasm_pivots_extract_structural spasm_hip_pivots_extract_structural

/* containers, field, I/O (spasm_util.c, spasm_ZZp.c, spasm_triplet.c, spasm_transpose.c, spasm_io.c) */
#define spasm_malloc                  spasm_hip_malloc
#define spasm_calloc                  spasm_hip_calloc
#define spasm_realloc                 spasm_hip_realloc
#define spasm_nnz                     spasm_hip_nnz
#define spasm_wtime                   spasm_hip_wtime
#define spasm_field_init              spasm_hip_field_init
#define spasm_ZZp_init                spasm_hip_ZZp_init
#define spasm_ZZp_add                 spasm_hip_ZZp_add
#define spasm_ZZp_sub                 spasm_hip_ZZp_sub
#define spasm_ZZp_mul                 spasm_hip_ZZp_mul
#define spasm_ZZp_inverse             spasm_hip_ZZp_inverse
#define spasm_ZZp_axpy                spasm_hip_ZZp_axpy
#define spasm_csr_alloc               spasm_hip_csr_alloc
#define spasm_csr_realloc             spasm_hip_csr_realloc
#define spasm_csr_resize              spasm_hip_csr_resize
#define spasm_csr_free                spasm_hip_csr_free
#define spasm_triplet_alloc           spasm_hip_triplet_alloc
#define spasm_triplet_realloc         spasm_hip_triplet_realloc
#define spasm_triplet_free            spasm_hip_triplet_free
#define spasm_lu_free                 spasm_hip_lu_free
#define spasm_add_entry               spasm_hip_add_entry
#define spasm_triplet_transpose       spasm_hip_triplet_transpose
#define spasm_compress                spasm_hip_compress
#define spasm_transpose               spasm_hip_transpose
#define spasm_triplet_load            spasm_hip_triplet_load
#define spasm_triplet_save            spasm_hip_triplet_save
#define spasm_csr_save                spasm_hip_csr_save

#endif
