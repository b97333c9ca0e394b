/*
 * orc_core.c -- oracle (TEST INFRASTRUCTURE ONLY, see mts_oracle.h):
 * Random, keyed stream, low-discrepancy code, warps, Fresnel, deterministic
 * elementary functions, triangle clipping, TriAccel.
 *
 * Every function cites the reference file:line it restates (/root/reference).
 * Float = IEEE binary32, compiled with -ffp-contract=off, no fast-math.
 */
#include "orc_internal.h"

/* ========================================================================== */
/* Random -- MT19937-64 (src/libcore/random.cpp:99-227, random.h:82-148)      */
/* ========================================================================== */
#define MT_N 312
#define MT_M 156
#define MT_MATRIX_A   0xB5026F5AA96619E9ULL
#define MT_UPPER_MASK 0xFFFFFFFF80000000ULL
#define MT_LOWER_MASK 0x7FFFFFFFULL

/* random.cpp:99-103 */
void orc_random_seed(orc_random *r, uint64_t s) {
	r->mt[0] = s;
	for (r->mti = 1; r->mti < MT_N; r->mti++)
		r->mt[r->mti] = 6364136223846793005ULL * (r->mt[r->mti-1] ^ (r->mt[r->mti-1] >> 62)) + (uint64_t) r->mti;
}

/* random.cpp:118-140 (init_by_array64) */
void orc_random_seed_array(orc_random *r, const uint64_t *init_key, uint64_t key_length) {
	uint64_t i, j, k;
	uint64_t *mt = r->mt;
	orc_random_seed(r, 19650218ULL);
	i = 1; j = 0;
	k = (MT_N > key_length ? MT_N : key_length);
	for (; k; k--) {
		mt[i] = (mt[i] ^ ((mt[i-1] ^ (mt[i-1] >> 62)) * 3935559000370003845ULL)) + init_key[j] + j;
		i++; j++;
		if (i >= MT_N) { mt[0] = mt[MT_N-1]; i = 1; }
		if (j >= key_length) j = 0;
	}
	for (k = MT_N-1; k; k--) {
		mt[i] = (mt[i] ^ ((mt[i-1] ^ (mt[i-1] >> 62)) * 2862933555777941757ULL)) - i;
		i++;
		if (i >= MT_N) { mt[0] = mt[MT_N-1]; i = 1; }
	}
	mt[0] = 1ULL << 63;
}

/* random.cpp:105-110: 312 draws from the parent, then init_by_array */
void orc_random_seed_from(orc_random *r, orc_random *parent) {
	uint64_t buf[MT_N];
	for (int i = 0; i < MT_N; ++i)
		buf[i] = orc_random_next_ulong(parent);
	orc_random_seed_array(r, buf, MT_N);
}

/* random.cpp:143-178 */
uint64_t orc_random_next_ulong(orc_random *r) {
	static const uint64_t mag01[2] = { 0ULL, MT_MATRIX_A };
	uint64_t *mt = r->mt;
	uint64_t x;
	int i;
	if (r->mti >= MT_N) {
		if (r->mti == MT_N+1)
			orc_random_seed(r, 5489ULL);
		for (i = 0; i < MT_N-MT_M; i++) {
			x = (mt[i] & MT_UPPER_MASK) | (mt[i+1] & MT_LOWER_MASK);
			mt[i] = mt[i+MT_M] ^ (x >> 1) ^ mag01[(int) (x & 1ULL)];
		}
		for (; i < MT_N-1; i++) {
			x = (mt[i] & MT_UPPER_MASK) | (mt[i+1] & MT_LOWER_MASK);
			mt[i] = mt[i+(MT_M-MT_N)] ^ (x >> 1) ^ mag01[(int) (x & 1ULL)];
		}
		x = (mt[MT_N-1] & MT_UPPER_MASK) | (mt[0] & MT_LOWER_MASK);
		mt[MT_N-1] = mt[MT_M-1] ^ (x >> 1) ^ mag01[(int) (x & 1ULL)];
		r->mti = 0;
	}
	x = mt[r->mti++];
	x ^= (x >> 29) & 0x5555555555555555ULL;
	x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
	x ^= (x << 37) & 0xFFF7EEE000000000ULL;
	x ^= (x >> 43);
	return x;
}

/* random.cpp:218-227 (single precision branch) */
float orc_ulong_to_float(uint64_t v) {
	union { uint32_t u; float f; } x;
	x.u = (uint32_t) ((v & 0xFFFFFFFFULL) >> 9) | 0x3f800000UL;
	return x.f - 1.0f;
}

float orc_random_next_float(orc_random *r) {
	return orc_ulong_to_float(orc_random_next_ulong(r));
}

/* random.cpp:196-215: bit mask + rejection */
uint64_t orc_size_bitmask(uint64_t n) {
	uint64_t bitmask = n;
	bitmask |= bitmask >> 1;
	bitmask |= bitmask >> 2;
	bitmask |= bitmask >> 4;
	bitmask |= bitmask >> 8;
	bitmask |= bitmask >> 16;
	bitmask |= bitmask >> 32;
	return bitmask;
}

uint64_t orc_random_next_size(orc_random *r, uint64_t n) {
	uint64_t result, bitmask = orc_size_bitmask(n);
	while ((result = (orc_random_next_ulong(r) & bitmask)) >= n)
		;
	return result;
}

/* random.h:145-148: for (it = end-1; it > begin; --it) swap(it, begin + nextSize(it-begin)) */
void orc_random_shuffle_u32(orc_random *r, uint32_t *a, size_t n) {
	if (n < 2) return;
	for (size_t it = n - 1; it > 0; --it) {
		size_t other = (size_t) orc_random_next_size(r, (uint64_t) it);
		uint32_t tmp = a[it]; a[it] = a[other]; a[other] = tmp;
	}
}

/* ========================================================================== */
/* Keyed stream: SplitMix64 keyed by (seed, a, b).  Stands in for Random in   */
/* the order-independent samplers; not in the reference.*/
/* ========================================================================== */
static inline uint64_t sm64_mix(uint64_t z) {
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

uint64_t orc_keyed_init(uint64_t seed, uint64_t a, uint64_t b) {
	uint64_t s = sm64_mix(seed + 0x9E3779B97F4A7C15ULL * (a + 1));
	s = sm64_mix(s + 0xD1342543DE82EF95ULL * (b + 1));
	return s;
}

uint64_t orc_keyed_next(uint64_t *state) {
	*state += 0x9E3779B97F4A7C15ULL;
	return sm64_mix(*state);
}

static uint64_t keyed_next_size(uint64_t *st, uint64_t n) {
	uint64_t result, bitmask = orc_size_bitmask(n);
	while ((result = (orc_keyed_next(st) & bitmask)) >= n)
		;
	return result;
}

/* ========================================================================== */
/* (0,2)-sequence (src/samplers/ldsampler.cpp:104-141)                        */
/* ========================================================================== */
/* ldsampler.cpp:104-112 without the final divide */
uint32_t orc_vdc_bits(uint32_t n, uint32_t scramble) {
	n = (n << 16) | (n >> 16);
	n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
	n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
	n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
	n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
	n ^= scramble;
	return n;
}

/* ldsampler.cpp:114-118 without the final divide */
uint32_t orc_sobol2_bits(uint32_t n, uint32_t scramble) {
	for (uint32_t v = 1U << 31; n != 0; n >>= 1, v ^= v >> 1)
		if (n & 0x1) scramble ^= v;
	return scramble;
}

/* (Float) n / (Float) 0x100000000LL  (ldsampler.cpp:111,117); may return 1.0f */
float orc_u32_to_unit(uint32_t n) {
	return (float) n / (float) 0x100000000LL;
}

/* generate1D/generate2D/generate (ldsampler.cpp:125-158).  The shuffle permutes
 * the value arrays; we permute an index array and evaluate afterwards, which is
 * the same thing (values[j] = f(index[j])). */
void orc_ld_generate_mt(orc_random *r, uint32_t spp, int depth, float *out1d, float *out2d) {
	uint32_t *perm = (uint32_t *) malloc(sizeof(uint32_t) * spp);
	for (int i = 0; i < depth; ++i) {
		/* generate1D */
		uint32_t scramble = (uint32_t) (orc_random_next_ulong(r) & 0xFFFFFFFFULL);
		for (uint32_t k = 0; k < spp; ++k) perm[k] = k;
		orc_random_shuffle_u32(r, perm, spp);
		for (uint32_t k = 0; k < spp; ++k)
			out1d[(size_t) i * spp + k] = orc_u32_to_unit(orc_vdc_bits(perm[k], scramble));
		/* generate2D: union { uint64_t qword; uint32_t dword[2]; } on little endian */
		uint64_t q = orc_random_next_ulong(r);
		uint32_t s0 = (uint32_t) (q & 0xFFFFFFFFULL), s1 = (uint32_t) (q >> 32);
		for (uint32_t k = 0; k < spp; ++k) perm[k] = k;
		orc_random_shuffle_u32(r, perm, spp);
		for (uint32_t k = 0; k < spp; ++k) {
			out2d[((size_t) i * spp + k) * 2 + 0] = orc_u32_to_unit(orc_vdc_bits(perm[k], s0));
			out2d[((size_t) i * spp + k) * 2 + 1] = orc_u32_to_unit(orc_sobol2_bits(perm[k], s1));
		}
	}
	free(perm);
}

static void keyed_shuffle_u32(uint64_t *st, uint32_t *a, size_t n) {
	if (n < 2) return;
	for (size_t it = n - 1; it > 0; --it) {
		size_t other = (size_t) keyed_next_size(st, (uint64_t) it);
		uint32_t tmp = a[it]; a[it] = a[other]; a[other] = tmp;
	}
}

/* Index form used by the renderer: scrambles [depth][3] (1D, 2D.x, 2D.y) and
 * permutations [depth][2][spp] */
uint64_t orc_ld_generate_keyed_tables(uint64_t seed, uint32_t pixel_key, uint32_t spp, int depth,
                                      uint32_t *scr, uint32_t *perm) {
	uint64_t st = orc_keyed_init(seed, pixel_key, 0);
	for (int i = 0; i < depth; ++i) {
		uint32_t *p1 = perm + ((size_t) i * 2 + 0) * spp, *p2 = perm + ((size_t) i * 2 + 1) * spp;
		scr[i*3+0] = (uint32_t) (orc_keyed_next(&st) & 0xFFFFFFFFULL);
		for (uint32_t k = 0; k < spp; ++k) p1[k] = k;
		keyed_shuffle_u32(&st, p1, spp);
		uint64_t q = orc_keyed_next(&st);
		scr[i*3+1] = (uint32_t) (q & 0xFFFFFFFFULL);
		scr[i*3+2] = (uint32_t) (q >> 32);
		for (uint32_t k = 0; k < spp; ++k) p2[k] = k;
		keyed_shuffle_u32(&st, p2, spp);
	}
	return st;          /* the stream goes on with the requested sample arrays (ldsampler.cpp:149-153) */
}

/* StratifiedSampler::generate (src/samplers/stratified.cpp:121-141) with the keyed stream: the stratum permutations
 * [depth][2][spp] (1D, 2D) */
uint64_t orc_strat_generate_keyed_tables(uint64_t seed, uint32_t pixel_key, uint32_t spp, int depth, uint32_t *perm) {
	uint64_t st = orc_keyed_init(seed, pixel_key, 0);
	for (int i = 0; i < depth; ++i) {
		uint32_t *p1 = perm + ((size_t) i * 2 + 0) * spp, *p2 = perm + ((size_t) i * 2 + 1) * spp;
		for (uint32_t k = 0; k < spp; ++k) p1[k] = k;
		keyed_shuffle_u32(&st, p1, spp);
		for (uint32_t k = 0; k < spp; ++k) p2[k] = k;
		keyed_shuffle_u32(&st, p2, spp);
	}
	return st;
}

/* Sampler::request2DArray / next2DArray (src/librender/sampler.cpp:71-87): one array of `size` points per camera
 * sample, all spp * size points of a pixel filled by generate().  The keyed samplers draw them from the pixel's
 * generate() stream (seed, pixel, 0), which *st continues.
 * kind 0: IndependentSampler::generate (independent.cpp:63-66): x then y per point */
void orc_independent_generate_array(uint64_t *st, size_t n, float *out) {
	for (size_t j = 0; j < n; ++j) {
		out[2*j+0] = orc_ulong_to_float(orc_keyed_next(st));
		out[2*j+1] = orc_ulong_to_float(orc_keyed_next(st));
	}
}

/* LowDiscrepancySampler::generate2D (ldsampler.cpp:129-141): one (0,2)-sequence of n points, scrambled by the two
 * halves of one 64-bit draw, then shuffled as points */
void orc_ld_generate_array(uint64_t *st, size_t n, float *out) {
	uint64_t q = orc_keyed_next(st);
	uint32_t lo = (uint32_t) (q & 0xFFFFFFFFULL), hi = (uint32_t) (q >> 32);
	uint32_t *perm = (uint32_t *) malloc(sizeof(uint32_t) * (n ? n : 1));
	for (size_t k = 0; k < n; ++k) perm[k] = (uint32_t) k;
	keyed_shuffle_u32(st, perm, n);
	for (size_t k = 0; k < n; ++k) {
		out[2*k+0] = orc_u32_to_unit(orc_vdc_bits(perm[k], lo));
		out[2*k+1] = orc_u32_to_unit(orc_sobol2_bits(perm[k], hi));
	}
	free(perm);
}

/* latinHypercube(random, dest, nSamples, nDim = 2) (src/libcore/util.cpp:529-540), as StratifiedSampler::generate
 * fills its 2D arrays (stratified.cpp:136-138) */
void orc_latin_hypercube_array(uint64_t *st, size_t n, float *out) {
	float delta = 1 / (float) n;
	for (size_t i = 0; i < n; ++i)
		for (size_t j = 0; j < 2; ++j)
			out[2*i+j] = (i + orc_ulong_to_float(orc_keyed_next(st))) * delta;
	for (size_t i = 0; i < 2; ++i) {
		for (size_t j = 0; j < n; ++j) {
			size_t other = (size_t) keyed_next_size(st, (uint64_t) n);
			float tmp = out[2*j+i]; out[2*j+i] = out[2*other+i]; out[2*other+i] = tmp;
		}
	}
}

void orc_ld_generate_keyed(uint64_t seed, uint32_t pixel_key, uint32_t spp, int depth,
                           float *out1d, float *out2d) {
	uint32_t *scr = (uint32_t *) malloc(sizeof(uint32_t) * 3 * (size_t) depth);
	uint32_t *perm = (uint32_t *) malloc(sizeof(uint32_t) * 2 * (size_t) depth * spp);
	orc_ld_generate_keyed_tables(seed, pixel_key, spp, depth, scr, perm);
	for (int i = 0; i < depth; ++i) {
		const uint32_t *p1 = perm + ((size_t) i * 2 + 0) * spp, *p2 = perm + ((size_t) i * 2 + 1) * spp;
		for (uint32_t k = 0; k < spp; ++k) {
			out1d[(size_t) i * spp + k] = orc_u32_to_unit(orc_vdc_bits(p1[k], scr[i*3+0]));
			out2d[((size_t) i * spp + k) * 2 + 0] = orc_u32_to_unit(orc_vdc_bits(p2[k], scr[i*3+1]));
			out2d[((size_t) i * spp + k) * 2 + 1] = orc_u32_to_unit(orc_sobol2_bits(p2[k], scr[i*3+2]));
		}
	}
	free(scr); free(perm);
}

/* util.cpp:738-750 */
float orc_radical_inverse(int b, uint64_t i) {
	float invB = (float) 1 / (float) b;
	float x = 0.0f, f = invB;
	while (i) {
		x += f * (float) (i % (uint64_t) b);
		i /= (uint64_t) b;
		f *= invB;
	}
	return x;
}

/* util.cpp:752-768 */
float orc_radical_inverse_incremental(int b, float x) {
	float invB = (float) 1 / (float) b;
	float h, hh, r = 1.0f - x - (float) 1e-10;
	if (invB < r) {
		x += invB;
	} else {
		h = invB;
		do {
			hh = h;
			h *= invB;
		} while (h >= r);
		x += hh + h - 1.0f;
	}
	return x;
}

/* ========================================================================== */
/* Deterministic elementary functions.                                        */
/* The reference calls libm (std::sin/cos/exp/log/atan/pow).  libm and the    */
/* device math library differ in the last bit, and one flipped bit changes a  */
/* path, so oracle and kernels both implement this specification instead:     */
/* evaluate in binary64 with a fixed operation order, round once to binary32. */
/* Results are faithfully rounded (checked against libm in tests/).           */
/* ========================================================================== */
static inline double u64_as_double(uint64_t u) { union { uint64_t u; double d; } x; x.u = u; return x.d; }
static inline uint64_t double_as_u64(double d) { union { uint64_t u; double d; } x; x.d = d; return x.u; }

static inline double dm_sin_poly(double r) {
	double r2 = r * r;
	double p = -1.0 / 1307674368000.0;
	p = p * r2 + 1.0 / 6227020800.0;
	p = p * r2 - 1.0 / 39916800.0;
	p = p * r2 + 1.0 / 362880.0;
	p = p * r2 - 1.0 / 5040.0;
	p = p * r2 + 1.0 / 120.0;
	p = p * r2 - 1.0 / 6.0;
	return r + r * (r2 * p);
}

static inline double dm_cos_poly(double r) {
	double r2 = r * r;
	double p = -1.0 / 87178291200.0;
	p = p * r2 + 1.0 / 479001600.0;
	p = p * r2 - 1.0 / 3628800.0;
	p = p * r2 + 1.0 / 40320.0;
	p = p * r2 - 1.0 / 720.0;
	p = p * r2 + 1.0 / 24.0;
	p = p * r2 - 0.5;
	return 1.0 + r2 * p;
}

/* quadrant reduction; valid for |x| up to ~1e5 (the path only needs [0, 2pi]) */
static inline void dm_sincos(double x, double *s, double *c) {
	const double TWO_OVER_PI = 0.63661977236758134308;
	const double PIO2_HI = 1.57079632679489655800e+00;
	const double PIO2_LO = 6.12323399573676603587e-17;
	double kd = x * TWO_OVER_PI;
	long long k = (long long) (kd + (kd >= 0 ? 0.5 : -0.5));
	double kf = (double) k;
	double r = (x - kf * PIO2_HI) - kf * PIO2_LO;
	double sr = dm_sin_poly(r), cr = dm_cos_poly(r);
	switch ((int) (k & 3)) {
		case 0: *s = sr;  *c = cr;  break;
		case 1: *s = cr;  *c = -sr; break;
		case 2: *s = -sr; *c = -cr; break;
		default: *s = -cr; *c = sr; break;
	}
}

float orc_sinf(float x) { double s, c; dm_sincos((double) x, &s, &c); return (float) s; }
float orc_cosf(float x) { double s, c; dm_sincos((double) x, &s, &c); return (float) c; }

float orc_expf(float x) {
	const double LOG2E = 1.44269504088896338700e+00;
	const double LN2_HI = 6.93147180369123816490e-01;
	const double LN2_LO = 1.90821492927058770002e-10;
	double xd = (double) x;
	if (x != x) return x;
	if (xd > 89.0) return INFINITY;
	if (xd < -104.0) return 0.0f;
	double kd = xd * LOG2E;
	long long k = (long long) (kd + (kd >= 0 ? 0.5 : -0.5));
	double kf = (double) k;
	double r = (xd - kf * LN2_HI) - kf * LN2_LO;
	double p = 1.0 / 6227020800.0;
	p = p * r + 1.0 / 479001600.0;
	p = p * r + 1.0 / 39916800.0;
	p = p * r + 1.0 / 3628800.0;
	p = p * r + 1.0 / 362880.0;
	p = p * r + 1.0 / 40320.0;
	p = p * r + 1.0 / 5040.0;
	p = p * r + 1.0 / 720.0;
	p = p * r + 1.0 / 120.0;
	p = p * r + 1.0 / 24.0;
	p = p * r + 1.0 / 6.0;
	p = p * r + 0.5;
	p = p * r + 1.0;
	p = p * r + 1.0;
	double scale = u64_as_double((uint64_t) (k + 1023) << 52);
	return (float) (p * scale);
}

float orc_logf(float x) {
	const double LN2 = 6.93147180559945286227e-01;
	const double SQRT2 = 1.41421356237309514547e+00;
	if (x != x || x < 0.0f) return NAN;
	if (x == 0.0f) return -INFINITY;
	if (x == INFINITY) return INFINITY;
	double xd = (double) x;
	uint64_t bits = double_as_u64(xd);
	long long e = (long long) ((bits >> 52) & 0x7ff) - 1023;
	double m = u64_as_double((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
	if (m > SQRT2) { m = m * 0.5; e = e + 1; }
	double s = (m - 1.0) / (m + 1.0);
	double s2 = s * s;
	double p = 1.0 / 19.0;
	p = p * s2 + 1.0 / 17.0;
	p = p * s2 + 1.0 / 15.0;
	p = p * s2 + 1.0 / 13.0;
	p = p * s2 + 1.0 / 11.0;
	p = p * s2 + 1.0 / 9.0;
	p = p * s2 + 1.0 / 7.0;
	p = p * s2 + 1.0 / 5.0;
	p = p * s2 + 1.0 / 3.0;
	p = p * s2 + 1.0;
	double logm = 2.0 * s * p;
	return (float) ((double) e * LN2 + logm);
}

static double dm_atan_d(double xin) {
	const double PIO2 = 1.57079632679489655800e+00;
	const double PIO4 = 7.85398163397448278999e-01;
	const double TAN_PIO8 = 0.41421356237309503;
	double xd = xin;
	int neg = xd < 0.0;
	if (neg) xd = -xd;
	int inv = xd > 1.0;
	if (inv) xd = 1.0 / xd;
	double base = 0.0, y = xd;
	if (xd > TAN_PIO8) { y = (xd - 1.0) / (xd + 1.0); base = PIO4; }
	double y2 = y * y;
	double p = 1.0 / 35.0;
	p = -p * y2 + 1.0 / 33.0;
	p = -p * y2 + 1.0 / 31.0;
	p = -p * y2 + 1.0 / 29.0;
	p = -p * y2 + 1.0 / 27.0;
	p = -p * y2 + 1.0 / 25.0;
	p = -p * y2 + 1.0 / 23.0;
	p = -p * y2 + 1.0 / 21.0;
	p = -p * y2 + 1.0 / 19.0;
	p = -p * y2 + 1.0 / 17.0;
	p = -p * y2 + 1.0 / 15.0;
	p = -p * y2 + 1.0 / 13.0;
	p = -p * y2 + 1.0 / 11.0;
	p = -p * y2 + 1.0 / 9.0;
	p = -p * y2 + 1.0 / 7.0;
	p = -p * y2 + 1.0 / 5.0;
	p = -p * y2 + 1.0 / 3.0;
	p = -p * y2 + 1.0;
	double a = base + y * p;
	if (inv) a = PIO2 - a;
	if (neg) a = -a;
	return a;
}

float orc_atanf(float x) {
	if (x != x) return x;
	return (float) dm_atan_d((double) x);
}

/* std::atan2 (envmap.cpp:149,185): quadrant logic around the binary64 arctangent of y/x */
float orc_atan2f(float y, float x) {
	const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00;
	if (x != x || y != y) return NAN;
	const double yd = (double) y, xd = (double) x;
	double r;
	if (xd == 0.0) {
		if (yd == 0.0) r = signbit(x) ? PI : 0.0;
		else return (float) (yd > 0.0 ? PIO2 : -PIO2);
		return (float) (signbit(y) ? -r : r);
	}
	if (isinf(xd) && isinf(yd)) {
		r = xd > 0.0 ? 0.25 * PI : 0.75 * PI;
		return (float) (yd > 0.0 ? r : -r);
	}
	r = dm_atan_d(yd / xd);
	if (xd < 0.0) r += signbit(y) ? -PI : PI;
	return (float) r;
}

/* std::pow(x, 4.0f) */
float orc_pow4f(float x) {
	double d = (double) x * (double) x;
	return (float) (d * d);
}

/* binary64 exp / log with the same reductions and polynomials as orc_expf / orc_logf */
static double dm_exp_d(double xd) {
	const double LOG2E = 1.44269504088896338700e+00;
	const double LN2_HI = 6.93147180369123816490e-01;
	const double LN2_LO = 1.90821492927058770002e-10;
	if (xd != xd) return xd;
	if (xd > 700.0) return INFINITY;
	if (xd < -700.0) return 0.0;
	double kd = xd * LOG2E;
	long long k = (long long) (kd + (kd >= 0 ? 0.5 : -0.5));
	double kf = (double) k;
	double r = (xd - kf * LN2_HI) - kf * LN2_LO;
	double p = 1.0 / 6227020800.0;
	p = p * r + 1.0 / 479001600.0;
	p = p * r + 1.0 / 39916800.0;
	p = p * r + 1.0 / 3628800.0;
	p = p * r + 1.0 / 362880.0;
	p = p * r + 1.0 / 40320.0;
	p = p * r + 1.0 / 5040.0;
	p = p * r + 1.0 / 720.0;
	p = p * r + 1.0 / 120.0;
	p = p * r + 1.0 / 24.0;
	p = p * r + 1.0 / 6.0;
	p = p * r + 0.5;
	p = p * r + 1.0;
	p = p * r + 1.0;
	return p * u64_as_double((uint64_t) (k + 1023) << 52);
}

static double dm_log_d(double xd) {      /* xd > 0, finite, normal */
	const double LN2 = 6.93147180559945286227e-01;
	const double SQRT2 = 1.41421356237309514547e+00;
	uint64_t bits = double_as_u64(xd);
	long long e = (long long) ((bits >> 52) & 0x7ff) - 1023;
	double m = u64_as_double((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
	if (m > SQRT2) { m = m * 0.5; e = e + 1; }
	double s = (m - 1.0) / (m + 1.0);
	double s2 = s * s;
	double p = 1.0 / 19.0;
	p = p * s2 + 1.0 / 17.0;
	p = p * s2 + 1.0 / 15.0;
	p = p * s2 + 1.0 / 13.0;
	p = p * s2 + 1.0 / 11.0;
	p = p * s2 + 1.0 / 9.0;
	p = p * s2 + 1.0 / 7.0;
	p = p * s2 + 1.0 / 5.0;
	p = p * s2 + 1.0 / 3.0;
	p = p * s2 + 1.0;
	return (double) e * LN2 + 2.0 * s * p;
}

/* std::pow(x, y) for x >= 0 (phong.cpp:120,133,163-164): exp(y * log(x)) in binary64 */
float orc_powf(float x, float y) {
	if (x != x || y != y) return NAN;
	if (y == 0.0f) return 1.0f;
	if (x < 0.0f) return NAN;
	if (x == 0.0f) return y > 0.0f ? 0.0f : INFINITY;
	if (x == 1.0f) return 1.0f;
	if (x == INFINITY) return y > 0.0f ? INFINITY : 0.0f;
	return (float) dm_exp_d((double) y * dm_log_d((double) x));
}

/* std::acos (spot.cpp:98): 2 atan(sqrt((1-x)/(1+x))) in binary64 */
float orc_acosf(float x) {
	const double PI = 3.14159265358979311600e+00;
	if (x != x || x > 1.0f || x < -1.0f) return NAN;
	if (x == -1.0f) return (float) PI;
	double xd = (double) x;
	double q = sqrt((1.0 - xd) / (1.0 + xd));
	return (float) (2.0 * dm_atan_d(q));
}

/* ========================================================================== */
/* Warps, coordinate system, Fresnel (src/libcore/util.cpp)                   */
/* ========================================================================== */
/* util.cpp:553-559 */
void orc_square_to_sphere(const float s[2], float out[3]) {
	float z = 1.0f - 2.0f * s[1];
	float r = 1.0f - z * z;
	r = sqrtf(fmaxf_((float) 0, r));
	float phi = 2.0f * ORC_PI * s[0];
	out[0] = r * orc_cosf(phi); out[1] = r * orc_sinf(phi); out[2] = z;
}

/* util.cpp:572-588 */
void orc_square_to_hemisphere_psa(const float s[2], float out[3]) {
	float r = sqrtf(s[0]);
	float phi = 2.0f * ORC_PI * s[1];
	float dirX = r * orc_cosf(phi);
	float dirY = r * orc_sinf(phi);
	float z = sqrtf(1 - fminf_((float) 1, dirX*dirX + dirY*dirY));
	if (z == 0) {
		/* normalize(Vector(dirX, dirY, Epsilon)) : v / length -> v * (1/length) */
		float len = sqrtf(dirX*dirX + dirY*dirY + ORC_EPS*ORC_EPS);
		float inv = 1.0f / len;
		out[0] = dirX * inv; out[1] = dirY * inv; out[2] = ORC_EPS * inv;
		return;
	}
	out[0] = dirX; out[1] = dirY; out[2] = z;
}

/* util.cpp:613-616 */
void orc_square_to_triangle(const float s[2], float out[2]) {
	float a = sqrtf(1.0f - s[0]);
	out[0] = 1 - a; out[1] = a * s[1];
}

/* util.cpp:629-651 */
void orc_square_to_disk_concentric(const float s[2], float out[2]) {
	float r1 = 2.0f*s[0] - 1.0f;
	float r2 = 2.0f*s[1] - 1.0f;
	float cx, cy;
	if (r1 == 0 && r2 == 0) {
		cx = 0; cy = 0;
	} else if (r1 > -r2) {
		if (r1 > r2) { cx = r1; cy = (ORC_PI/4.0f) * r2/r1; }
		else { cx = r2; cy = (ORC_PI/4.0f) * (2.0f - r1/r2); }
	} else {
		if (r1 < r2) { cx = -r1; cy = (ORC_PI/4.0f) * (4.0f + r2/r1); }
		else { cx = -r2; cy = (ORC_PI/4.0f) * (6.0f - r1/r2); }
	}
	out[0] = cx * orc_cosf(cy);
	out[1] = cx * orc_sinf(cy);
}

/* util.cpp:602-611 */
void orc_coordinate_system(const float a[3], float b[3], float c[3]) {
	if (fabsf(a[0]) > fabsf(a[1])) {
		float invLen = 1.0f / sqrtf(a[0]*a[0] + a[2]*a[2]);
		b[0] = -a[2] * invLen; b[1] = 0.0f; b[2] = a[0] * invLen;
	} else {
		float invLen = 1.0f / sqrtf(a[1]*a[1] + a[2]*a[2]);
		b[0] = 0.0f; b[1] = -a[2] * invLen; b[2] = a[1] * invLen;
	}
	v3_cross(c, a, b);
}

/* util.cpp:680-688 */
float orc_fresnel_dielectric(float cosTheta1, float cosTheta2, float etaI, float etaT) {
	float Rs = (etaI * cosTheta1 - etaT * cosTheta2) / (etaI * cosTheta1 + etaT * cosTheta2);
	float Rp = (etaT * cosTheta1 - etaI * cosTheta2) / (etaT * cosTheta1 + etaI * cosTheta2);
	return (Rs * Rs + Rp * Rp) / 2.0f;
}

/* util.cpp:704-727 */
float orc_fresnel(float cosThetaI, float etaExt, float etaInt) {
	float etaI = etaExt, etaT = etaInt;
	if (cosThetaI < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
	float sinThetaT = etaI / etaT * sqrtf(fmaxf_((float) 0.0f, 1.0f - cosThetaI*cosThetaI));
	if (sinThetaT > 1.0f)
		return 1.0f;
	float cosThetaT = sqrtf(1.0f - sinThetaT*sinThetaT);
	return orc_fresnel_dielectric(fabsf(cosThetaI), cosThetaT, etaI, etaT);
}

/* util.cpp:690-702; Spectrum ops are per channel, Spectrum/2.0f = * (1/2.0f) (spectrum.h:228-240) */
void orc_fresnel_conductor(float cosTheta, const float eta[3], const float k[3], float out[3]) {
	for (int i = 0; i < 3; ++i) {
		float tmp = (eta[i]*eta[i] + k[i]*k[i]) * (cosTheta * cosTheta);
		float rParl2 = (tmp - (eta[i] * (2.0f * cosTheta)) + 1.0f)
		             / (tmp + (eta[i] * (2.0f * cosTheta)) + 1.0f);
		float tmpF = eta[i]*eta[i] + k[i]*k[i];
		float rPerp2 = (tmpF - (eta[i] * (2.0f * cosTheta)) + (cosTheta*cosTheta))
		             / (tmpF + (eta[i] * (2.0f * cosTheta)) + (cosTheta*cosTheta));
		out[i] = (rParl2 + rPerp2) * (1.0f / 2.0f);
	}
}

/* ========================================================================== */
/* Triangle::getClippedAABB (src/libcore/triangle.cpp:59-158)                 */
/* ========================================================================== */
#define MAX_VERTS 10

/* triangle.cpp:61-106 */
static int sutherland_hodgman(double (*input)[3], int inCount, double (*output)[3], int axis,
                              double splitPos, int isMinimum) {
	if (inCount < 3)
		return 0;
	double cur[3] = { input[0][0], input[0][1], input[0][2] };
	double sign = isMinimum ? 1.0f : -1.0f;
	double distance = sign * (cur[axis] - splitPos);
	int curIsInside = (distance >= 0);
	int outCount = 0;
	for (int i = 0; i < inCount; ++i) {
		int nextIdx = i+1;
		if (nextIdx == inCount)
			nextIdx = 0;
		double next[3] = { input[nextIdx][0], input[nextIdx][1], input[nextIdx][2] };
		distance = sign * (next[axis] - splitPos);
		int nextIsInside = (distance >= 0);
		if (curIsInside && nextIsInside) {
			for (int c = 0; c < 3; ++c) output[outCount][c] = next[c];
			outCount++;
		} else if (curIsInside && !nextIsInside) {
			double t = (splitPos - cur[axis]) / (next[axis] - cur[axis]);
			for (int c = 0; c < 3; ++c) output[outCount][c] = cur[c] + (next[c] - cur[c]) * t;
			output[outCount][axis] = splitPos;
			outCount++;
		} else if (!curIsInside && nextIsInside) {
			double t = (splitPos - cur[axis]) / (next[axis] - cur[axis]);
			for (int c = 0; c < 3; ++c) output[outCount][c] = cur[c] + (next[c] - cur[c]) * t;
			output[outCount][axis] = splitPos;
			outCount++;
			for (int c = 0; c < 3; ++c) output[outCount][c] = next[c];
			outCount++;
		}
		for (int c = 0; c < 3; ++c) cur[c] = next[c];
		curIsInside = nextIsInside;
	}
	return outCount;
}

/* triangle.cpp:108-158 */
int orc_clipped_aabb(const float p0[3], const float p1[3], const float p2[3],
                     const float bmin[3], const float bmax[3], float omin[3], float omax[3]) {
	double vertices1[MAX_VERTS][3], vertices2[MAX_VERTS][3];
	int nVertices = 3;
	for (int c = 0; c < 3; ++c) {
		vertices1[0][c] = (double) p0[c];
		vertices1[1][c] = (double) p1[c];
		vertices1[2][c] = (double) p2[c];
	}
	for (int axis = 0; axis < 3; ++axis) {
		nVertices = sutherland_hodgman(vertices1, nVertices, vertices2, axis, (double) bmin[axis], 1);
		nVertices = sutherland_hodgman(vertices2, nVertices, vertices1, axis, (double) bmax[axis], 0);
	}
	for (int c = 0; c < 3; ++c) { omin[c] = INFINITY; omax[c] = -INFINITY; }
	for (int i = 0; i < nVertices; ++i) {
		for (int j = 0; j < 3; ++j) {
			double pos_d = vertices1[i][j];
			float pos_f = (float) pos_d;
			float pos_roundedDown, pos_roundedUp;
			if (pos_f < pos_d) {
				pos_roundedDown = pos_f;
				pos_roundedUp = nextafterf(pos_f, INFINITY);
			} else if (pos_f > pos_d) {
				pos_roundedUp = pos_f;
				pos_roundedDown = nextafterf(pos_f, -INFINITY);
			} else {
				pos_roundedDown = pos_roundedUp = pos_f;
			}
			omin[j] = fminf_(omin[j], pos_roundedDown);
			omax[j] = fmaxf_(omax[j], pos_roundedUp);
		}
	}
	/* result.clip(aabb)  (aabb.h:82-87) */
	for (int c = 0; c < 3; ++c) {
		omin[c] = fmaxf_(omin[c], bmin[c]);
		omax[c] = fminf_(omax[c], bmax[c]);
	}
	/* AABB::isValid (aabb.h:214-219) */
	for (int c = 0; c < 3; ++c)
		if (omax[c] < omin[c])
			return 0;
	return 1;
}

/* ========================================================================== */
/* TriAccel (include/mitsuba/render/triaccel.h)                               */
/* ========================================================================== */
/* triaccel.h:63-95 */
int orc_triaccel_load(const float A[3], const float B[3], const float C[3], uint32_t out[12]) {
	static const int waldModulo[4] = { 1, 2, 0, 1 };
	float b[3], c[3], N[3];
	v3_sub(b, C, A); v3_sub(c, B, A); v3_cross(N, c, b);
	uint32_t k = 0;
	for (int j = 0; j < 3; j++)
		if (fabsf(N[j]) > fabsf(N[k]))
			k = (uint32_t) j;
	uint32_t u = (uint32_t) waldModulo[k], v = (uint32_t) waldModulo[k+1];
	const float n_k = N[k], denom = b[u]*c[v] - b[v]*c[u];
	float *f = (float *) out;
	if (denom == 0) {
		out[0] = 3;
		return 1;
	}
	out[0] = k;
	f[1] = N[u] / n_k;                 /* n_u  */
	f[2] = N[v] / n_k;                 /* n_v  */
	f[3] = v3_dot(A, N) / n_k;         /* n_d  */
	f[4] = A[u];                       /* a_u  */
	f[5] = A[v];                       /* a_v  */
	f[6] = b[u] / denom;               /* b_nu */
	f[7] = -b[v] / denom;              /* b_nv */
	f[8] = c[v] / denom;               /* c_nu */
	f[9] = -c[u] / denom;              /* c_nv */
	return 0;
}

/* triaccel.h:98-159 */
int orc_triaccel_intersect(const uint32_t ta[12], const float o[3], const float d[3],
                           float mint, float maxt, float *u, float *v, float *t) {
	const float *f = (const float *) ta;
	float o_u, o_v, o_k, d_u, d_v, d_k;
	switch (ta[0]) {
		case 0: o_u = o[1]; o_v = o[2]; o_k = o[0]; d_u = d[1]; d_v = d[2]; d_k = d[0]; break;
		case 1: o_u = o[2]; o_v = o[0]; o_k = o[1]; d_u = d[2]; d_v = d[0]; d_k = d[1]; break;
		case 2: o_u = o[0]; o_v = o[1]; o_k = o[2]; d_u = d[0]; d_v = d[1]; d_k = d[2]; break;
		default: return 0;
	}
	const float n_u = f[1], n_v = f[2], n_d = f[3], a_u = f[4], a_v = f[5],
	            b_nu = f[6], b_nv = f[7], c_nu = f[8], c_nv = f[9];
	float recip = 1.0f / (d_u * n_u + d_v * n_v + d_k);
	*t = (n_d - o_u*n_u - o_v*n_v - o_k) * recip;
	if (*t < mint || *t > maxt)
		return 0;
	const float hu = o_u + *t * d_u - a_u;
	const float hv = o_v + *t * d_v - a_v;
	*u = hv * b_nu + hu * b_nv;
	*v = hu * c_nu + hv * c_nv;
	return *u >= 0 && *v >= 0 && *u + *v <= 1.0f;
}
