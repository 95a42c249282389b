// Datatype plumbing of the dense tail (replaces the helpers at the end of
// spasm_ffpack.cpp:99-148) and the default options of the driver.
#include "common.h"

using namespace sh;

extern "C" {

spasm_ZZp spasm_hip_datatype_read(const void *A, size_t i, spasm_datatype datatype)
{
	switch (datatype) {
	case SPASM_DOUBLE: return (spasm_ZZp) ((const double *) A)[i];
	case SPASM_FLOAT: return (spasm_ZZp) ((const float *) A)[i];
	case SPASM_I64: return (spasm_ZZp) ((const i64 *) A)[i];
	}
	die("unknown datatype %d", (int) datatype);
}

void spasm_hip_datatype_write(void *A, size_t i, spasm_datatype datatype, spasm_ZZp value)
{
	switch (datatype) {
	case SPASM_DOUBLE: ((double *) A)[i] = value; return;
	case SPASM_FLOAT: ((float *) A)[i] = (float) value; return;
	case SPASM_I64: ((i64 *) A)[i] = value; return;
	}
	die("unknown datatype %d", (int) datatype);
}

size_t spasm_hip_datatype_size(spasm_datatype datatype)
{
	switch (datatype) {
	case SPASM_DOUBLE: return sizeof(double);
	case SPASM_FLOAT: return sizeof(float);
	case SPASM_I64: return sizeof(i64);
	}
	die("unknown datatype %d", (int) datatype);
}

// same thresholds as the reference (they come from what FFLAS can hold exactly
// in a float / a double); on the GPU the arithmetic is integer whatever the
// host-side container is.
spasm_datatype spasm_hip_datatype_choose(i64 prime)
{
	if (prime <= 8191)
		return SPASM_FLOAT;
	if (prime <= 189812531)
		return SPASM_DOUBLE;
	return SPASM_I64;
}

const char *spasm_hip_datatype_name(spasm_datatype datatype)
{
	switch (datatype) {
	case SPASM_DOUBLE: return "double";
	case SPASM_FLOAT: return "float";
	case SPASM_I64: return "i64";
	}
	die("unknown datatype %d", (int) datatype);
}

void spasm_hip_echelonize_init_opts(struct echelonize_opts *opts)
{
	opts->enable_greedy_pivot_search = 1;
	opts->enable_tall_and_skinny = 1;
	opts->enable_dense = 1;
	opts->enable_GPLU = 1;
	opts->L = 0;
	opts->complete = 0;
	opts->min_pivot_proportion = 0.1;
	opts->max_round = 3;
	opts->sparsity_threshold = 0.05;
	opts->tall_and_skinny_ratio = 5;
	opts->dense_block_size = 1000;
	opts->low_rank_ratio = 0.5;
	opts->low_rank_start_weight = -1;
}

}  // extern "C"
