// Entry points of include/spasm_hip.h that are not implemented yet.  They
// exist so that the ABI is complete and fail loudly (no silent fallback).
#include "common.h"

using namespace sh;

extern "C" {

// PLUQ with an explicit L (spasm_ffpack.cpp:57-86) is only needed when opts->L is set, which the
// GPU driver refuses for now.
int spasm_hip_ffpack_LU(i64, int, int, void *, int, spasm_datatype, size_t *, size_t *)
{
	die("spasm_hip_ffpack_LU: not implemented yet on the GPU path (needed only with opts->L)");
}

}  // extern "C"
