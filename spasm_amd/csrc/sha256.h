// SHA-256 (FIPS 180-4).  Used for the input-matrix digest (spasm_io.c) and as
// the counter-mode generator behind the random linear combinations of the
// low-rank mode (spasm_prng.c).
#pragma once
#include <cstddef>
#include <cstdint>

namespace sh {

struct Sha256 {
	uint32_t h[8];
	uint64_t total;
	uint8_t buf[64];
	size_t fill;
	void reset();
	void update(const void *data, size_t len);
	void finish(uint8_t out[32]);
private:
	void block(const uint8_t *b);
};

}  // namespace sh
