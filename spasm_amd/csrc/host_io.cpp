// SMS / MatrixMarket reader and SMS writer (replaces spasm_io.c:60-192), plus
// the SHA-256 digest of the input stream that the reference reports.
#include <cctype>
#include <cinttypes>
#include <string>

#include "common.h"
#include "sha256.h"

using namespace sh;

namespace sh {

static const uint32_t K256[64] = {
	0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
	0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
	0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
	0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
	0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
	0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
	0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
	0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

static inline uint32_t rotr(uint32_t x, int k) { return (x >> k) | (x << (32 - k)); }

void Sha256::reset()
{
	static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
	                               0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
	for (int i = 0; i < 8; i++)
		h[i] = iv[i];
	total = 0;
	fill = 0;
}

void Sha256::block(const uint8_t *b)
{
	uint32_t w[64];
	for (int t = 0; t < 16; t++)
		w[t] = ((uint32_t) b[4 * t] << 24) | ((uint32_t) b[4 * t + 1] << 16) | ((uint32_t) b[4 * t + 2] << 8) | b[4 * t + 3];
	for (int t = 16; t < 64; t++) {
		uint32_t s0 = rotr(w[t - 15], 7) ^ rotr(w[t - 15], 18) ^ (w[t - 15] >> 3);
		uint32_t s1 = rotr(w[t - 2], 17) ^ rotr(w[t - 2], 19) ^ (w[t - 2] >> 10);
		w[t] = w[t - 16] + s0 + w[t - 7] + s1;
	}
	uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
	for (int t = 0; t < 64; t++) {
		uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
		uint32_t ch = (e & f) ^ (~e & g);
		uint32_t t1 = hh + S1 + ch + K256[t] + w[t];
		uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
		uint32_t mj = (a & bb) ^ (a & c) ^ (bb & c);
		uint32_t t2 = S0 + mj;
		hh = g; g = f; f = e; e = d + t1;
		d = c; c = bb; bb = a; a = t1 + t2;
	}
	h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

void Sha256::update(const void *data, size_t len)
{
	const uint8_t *src = (const uint8_t *) data;
	total += len;
	while (len > 0) {
		size_t take = 64 - fill;
		if (take > len)
			take = len;
		std::memcpy(buf + fill, src, take);
		fill += take;
		src += take;
		len -= take;
		if (fill == 64) {
			block(buf);
			fill = 0;
		}
	}
}

void Sha256::finish(uint8_t out[32])
{
	uint64_t bits = total * 8;
	uint8_t pad = 0x80;
	update(&pad, 1);
	uint8_t zero = 0;
	while (fill != 56)
		update(&zero, 1);
	uint8_t lenb[8];
	for (int i = 0; i < 8; i++)
		lenb[i] = (uint8_t) (bits >> (56 - 8 * i));
	update(lenb, 8);
	for (int i = 0; i < 8; i++) {
		out[4 * i] = (uint8_t) (h[i] >> 24);
		out[4 * i + 1] = (uint8_t) (h[i] >> 16);
		out[4 * i + 2] = (uint8_t) (h[i] >> 8);
		out[4 * i + 3] = (uint8_t) h[i];
	}
}

}  // namespace sh

// one text line; returns false at end of file
static bool next_line(FILE *f, char *buf, int cap, i64 lineno, Sha256 *ctx)
{
	if (std::fgets(buf, cap, f) == nullptr) {
		if (std::feof(f))
			return false;
		die("cannot read line %" PRId64 " of the input matrix", lineno);
	}
	size_t l = std::strlen(buf);
	if (l == 0)
		die("empty line %" PRId64 " in the input matrix", lineno);
	if (buf[l - 1] != '\n' && !std::feof(f))
		die("line %" PRId64 " of the input matrix is longer than %d characters", lineno, cap);
	if (ctx != nullptr)
		ctx->update(buf, l);
	return true;
}

static std::string lower(const char *s)
{
	std::string r(s);
	for (auto &c : r)
		c = (char) std::tolower((unsigned char) c);
	return r;
}

extern "C" {

struct spasm_triplet *spasm_hip_triplet_load(FILE *f, i64 prime, u8 *hash)
{
	if (f == nullptr)
		die("spasm_hip_triplet_load: null stream");
	double t0 = wtime();
	Sha256 sha;
	sha.reset();
	Sha256 *ctx = (hash != nullptr) ? &sha : nullptr;
	char buf[1024];
	i64 lineno = 0;
	if (!next_line(f, buf, sizeof(buf), lineno, ctx))
		die("empty matrix file");
	int n = 0, m = 0;
	i64 announced = 1;
	bool mm = false;
	if (std::strncmp(buf, "%%MatrixMarket", 14) == 0) {
		mm = true;
		char a[1024], b[1024], c[1024], d[1024];
		if (std::sscanf(buf, "%%%%MatrixMarket %1023s %1023s %1023s %1023s", a, b, c, d) != 4)
			die("incomplete MatrixMarket header");
		if (lower(a) != "matrix")
			die("unsupported MatrixMarket object type %s (only ``matrix'')", a);
		if (lower(b) != "coordinate")
			die("unsupported MatrixMarket format %s (only ``coordinate'')", b);
		if (lower(c) != "integer")
			die("unsupported MatrixMarket data type %s (only ``integer'')", c);
		if (lower(d) != "general")
			die("unsupported MatrixMarket storage scheme %s (only ``general'')", d);
		for (;;) {
			lineno += 1;
			if (!next_line(f, buf, sizeof(buf), lineno, ctx))
				die("premature end of file on line %" PRId64 " (expected the dimensions)", lineno);
			if (buf[0] != '%')
				break;
		}
		if (std::sscanf(buf, "%d %d %" SCNd64, &n, &m, &announced) != 3)
			die("bad MatrixMarket dimensions (line %" PRId64 ")", lineno);
		logmsg("[IO] loading %d x %d MatrixMarket matrix modulo %" PRId64 " with %" PRId64 " non-zero... ", n, m, prime, announced);
	} else {
		char type;
		if (std::sscanf(buf, "%d %d %c", &n, &m, &type) != 3)
			die("bad SMS file (header)");
		if (prime != -1 && type != 'M')
			die("only ``Modular'' SMS files are supported");
		logmsg("[IO] loading %d x %d SMS matrix modulo %" PRId64 "... ", n, m, prime);
	}
	struct spasm_triplet *T = spasm_hip_triplet_alloc(n, m, announced, prime, prime != -1);
	bool done = false;
	i64 entries = 0;
	for (;;) {
		lineno += 1;
		bool got = next_line(f, buf, sizeof(buf), lineno, ctx);
		if (done && !got)
			break;
		if (done && got) {
			std::fprintf(stderr, "[spasm-hip] warning: garbage after the end of the matrix\n");
			continue;
		}
		if (!got)
			die("premature end of file (line %" PRId64 ", %" PRId64 " entries read)", lineno, entries);
		int i, j;
		i64 x;
		if (std::sscanf(buf, "%d %d %" SCNd64, &i, &j, &x) != 3)
			die("parse error on line %" PRId64, lineno);
		if (i == 0 && j == 0 && x == 0) {
			if (mm)
				die("SMS end marker in a MatrixMarket file");
			done = true;
		}
		if (!done) {
			spasm_hip_add_entry(T, i - 1, j - 1, x);
			entries += 1;
		}
		if (mm && entries == announced)
			done = true;
	}
	if (!mm)
		spasm_hip_triplet_realloc(T, -1);
	logmsg("%" PRId64 " non-zero [%.1fs]\n", T->nz, wtime() - t0);
	if (hash != nullptr)
		sha.finish(hash);
	return T;
}

void spasm_hip_csr_save(const struct spasm_csr *A, FILE *f)
{
	std::fprintf(f, "%d %d M\n", A->n, A->m);
	for (int i = 0; i < A->n; i++)
		for (i64 px = A->p[i]; px < A->p[i + 1]; px++) {
			i64 x = (A->x != nullptr) ? A->x[px] : 1;
			std::fprintf(f, "%d %d %" PRId64 "\n", i + 1, A->j[px] + 1, x);
		}
	std::fprintf(f, "0 0 0\n");
}

void spasm_hip_triplet_save(const struct spasm_triplet *A, FILE *f)
{
	std::fprintf(f, "%d %d M\n", A->n, A->m);
	for (i64 px = 0; px < A->nz; px++)
		std::fprintf(f, "%d %d %d\n", A->i[px] + 1, A->j[px] + 1, (A->x != nullptr) ? A->x[px] : 1);
	std::fprintf(f, "0 0 0\n");
}

}  // extern "C"
