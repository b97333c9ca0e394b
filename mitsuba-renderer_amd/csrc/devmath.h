// devmath.h -- float math shared by the gfx950 kernels and the host-side
// flattening code of libmtsgpu (compiled by hipcc for both sides).
//
// Everything here follows the reference's operation order (Mitsuba 0.2.1,
// citations per function) in IEEE binary32 with -ffp-contract=off: one flipped
// bit changes a path, so no reassociation, no FMA, no fast-math.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define HD __host__ __device__ __forceinline__

namespace mg {

// include/mitsuba/core/constants.h:31-50
constexpr float kEpsilon = 1e-4f;
constexpr float kShadowEpsilon = 1e-3f;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
#define MG_INF __builtin_huge_valf()

// std::max / std::min: the second operand wins only on a strict compare
HD float smax(float a, float b) { return (a < b) ? b : a; }
HD float smin(float a, float b) { return (b < a) ? b : a; }

struct V3 {
	float x, y, z;
	HD V3() {}
	HD V3(float a, float b, float c) : x(a), y(b), z(c) {}
	HD float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
HD V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
HD V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
HD V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
HD V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
HD V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
// include/mitsuba/core/vector.h:386-401
HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HD V3 cross(V3 a, V3 b) {
	return V3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x));
}
HD float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
// vector.h:312-330: division by a scalar multiplies by the reciprocal
HD V3 divs(V3 a, float f) { float r = 1.0f / f; return V3(a.x * r, a.y * r, a.z * r); }
HD V3 normalize(V3 a) { return divs(a, length(a)); }
HD bool isZero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }

// ---------------------------------------------------------------------------
// Deterministic elementary functions (specification: DESIGN.md section 5).
// binary64 evaluation in a fixed order, one rounding to binary32 at the end.
// ---------------------------------------------------------------------------
HD double bits2d(uint64_t u) { union { uint64_t u; double d; } x; x.u = u; return x.d; }
HD uint64_t d2bits(double d) { union { uint64_t u; double d; } x; x.d = d; return x.u; }

HD void sincos_d(double x, double &s, double &c) {
	const double TWO_OVER_PI = 0.63661977236758134308;
	const double PIO2_HI = 1.57079632679489655800e+00;
	const double PIO2_LO = 6.12323399573676603587e-17;
	double kd = x * TWO_OVER_PI;
	long long k = (long long) (kd + (kd >= 0 ? 0.5 : -0.5));
	double kf = (double) k;
	double r = (x - kf * PIO2_HI) - kf * PIO2_LO;
	double r2 = r * r;
	double ps = -1.0 / 1307674368000.0;
	ps = ps * r2 + 1.0 / 6227020800.0;
	ps = ps * r2 - 1.0 / 39916800.0;
	ps = ps * r2 + 1.0 / 362880.0;
	ps = ps * r2 - 1.0 / 5040.0;
	ps = ps * r2 + 1.0 / 120.0;
	ps = ps * r2 - 1.0 / 6.0;
	double sr = r + r * (r2 * ps);
	double pc = -1.0 / 87178291200.0;
	pc = pc * r2 + 1.0 / 479001600.0;
	pc = pc * r2 - 1.0 / 3628800.0;
	pc = pc * r2 + 1.0 / 40320.0;
	pc = pc * r2 - 1.0 / 720.0;
	pc = pc * r2 + 1.0 / 24.0;
	pc = pc * r2 - 0.5;
	double cr = 1.0 + r2 * pc;
	int q = (int) (k & 3);
	if (q == 0) { s = sr; c = cr; }
	else if (q == 1) { s = cr; c = -sr; }
	else if (q == 2) { s = -sr; c = -cr; }
	else { s = -cr; c = sr; }
}
HD void dsincos(float x, float &s, float &c) { double sd, cd; sincos_d((double) x, sd, cd); s = (float) sd; c = (float) cd; }
HD float dsin(float x) { float s, c; dsincos(x, s, c); return s; }
HD float dcos(float x) { float s, c; dsincos(x, s, c); return c; }

HD float dexp(float x) {
	const double LOG2E = 1.44269504088896338700e+00;
	const double LN2_HI = 6.93147180369123816490e-01;
	const double LN2_LO = 1.90821492927058770002e-10;
	double xd = (double) x;
	if (x != x) return x;
	if (xd > 89.0) return MG_INF;
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
	double scale = bits2d((uint64_t) (k + 1023) << 52);
	return (float) (p * scale);
}

HD float dlog(float x) {
	const double LN2 = 6.93147180559945286227e-01;
	const double SQRT2 = 1.41421356237309514547e+00;
	if (x != x || x < 0.0f) return __builtin_nanf("");
	if (x == 0.0f) return -MG_INF;
	if (x == MG_INF) return MG_INF;
	double xd = (double) x;
	uint64_t bits = d2bits(xd);
	long long e = (long long) ((bits >> 52) & 0x7ff) - 1023;
	double m = bits2d((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
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

HD double atan_d(double xin) {
	const double PIO2 = 1.57079632679489655800e+00;
	const double PIO4 = 7.85398163397448278999e-01;
	const double TAN_PIO8 = 0.41421356237309503;
	double xd = xin;
	bool neg = xd < 0.0;
	if (neg) xd = -xd;
	bool inv = xd > 1.0;
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
HD float datan(float x) { if (x != x) return x; return (float) atan_d((double) x); }
// std::acos: 2 atan(sqrt((1 - x) / (1 + x)))
HD float dacos(float x) {
	const double PI = 3.14159265358979311600e+00;
	if (x != x || x > 1.0f || x < -1.0f) return __builtin_nanf("");
	if (x == -1.0f) return (float) PI;
	const double xd = (double) x;
	return (float) (2.0 * atan_d(sqrt((1.0 - xd) / (1.0 + xd))));
}

// std::atan2 (envmap.cpp:149,185): quadrant logic around the binary64 arctangent of y / x
HD float datan2(float y, float x) {
	const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00;
	if (x != x || y != y) return __builtin_nanf("");
	const double yd = (double) y, xd = (double) x;
	const bool ysign = (__builtin_bit_cast(uint32_t, y) >> 31) != 0u, xsign = (__builtin_bit_cast(uint32_t, x) >> 31) != 0u;
	double r;
	if (xd == 0.0) {
		if (yd == 0.0) r = xsign ? PI : 0.0;
		else return (float) (yd > 0.0 ? PIO2 : -PIO2);
		return (float) (ysign ? -r : r);
	}
	if ((xd == (double) MG_INF || xd == -(double) MG_INF) && (yd == (double) MG_INF || yd == -(double) MG_INF)) {
		r = xd > 0.0 ? 0.25 * PI : 0.75 * PI;
		return (float) (yd > 0.0 ? r : -r);
	}
	r = atan_d(yd / xd);
	if (xd < 0.0) r += ysign ? -PI : PI;
	return (float) r;
}

HD float dpow4(float x) { double d = (double) x * (double) x; return (float) (d * d); }

// binary64 exp / log (same reductions and polynomials as dexp / dlog) for pow
HD double exp_d(double xd) {
	const double LOG2E = 1.44269504088896338700e+00;
	const double LN2_HI = 6.93147180369123816490e-01;
	const double LN2_LO = 1.90821492927058770002e-10;
	if (xd != xd) return xd;
	if (xd > 700.0) return (double) MG_INF;
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
	return p * bits2d((uint64_t) (k + 1023) << 52);
}
HD double log_d(double xd) {
	const double LN2 = 6.93147180559945286227e-01;
	const double SQRT2 = 1.41421356237309514547e+00;
	uint64_t bits = d2bits(xd);
	long long e = (long long) ((bits >> 52) & 0x7ff) - 1023;
	double m = bits2d((bits & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
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
// std::pow(x, y), x >= 0
HD float dpow(float x, float y) {
	if (x != x || y != y) return __builtin_nanf("");
	if (y == 0.0f) return 1.0f;
	if (x < 0.0f) return __builtin_nanf("");
	if (x == 0.0f) return y > 0.0f ? 0.0f : MG_INF;
	if (x == 1.0f) return 1.0f;
	if (x == MG_INF) return y > 0.0f ? MG_INF : 0.0f;
	return (float) exp_d((double) y * log_d((double) x));
}

// ---------------------------------------------------------------------------
// Keyed stream + Random's derived draws (src/libcore/random.cpp:196-227)
// ---------------------------------------------------------------------------
HD uint64_t sm64mix(uint64_t z) {
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
HD uint64_t keyedInit(uint64_t seed, uint64_t a, uint64_t b) {
	uint64_t s = sm64mix(seed + 0x9E3779B97F4A7C15ULL * (a + 1));
	return sm64mix(s + 0xD1342543DE82EF95ULL * (b + 1));
}
HD uint64_t keyedNext(uint64_t &state) {
	state += 0x9E3779B97F4A7C15ULL;
	return sm64mix(state);
}
// Random::nextFloat, single precision branch (random.cpp:218-227)
HD float ulongToFloat(uint64_t v) {
	union { uint32_t u; float f; } x;
	x.u = (uint32_t) ((v & 0xFFFFFFFFULL) >> 9) | 0x3f800000u;
	return x.f - 1.0f;
}
// Random::nextSize (random.cpp:196-215)
HD uint64_t keyedNextSize(uint64_t &state, uint64_t n) {
	uint64_t bitmask = n;
	bitmask |= bitmask >> 1; bitmask |= bitmask >> 2; bitmask |= bitmask >> 4;
	bitmask |= bitmask >> 8; bitmask |= bitmask >> 16; bitmask |= bitmask >> 32;
	uint64_t result;
	do { result = keyedNext(state) & bitmask; } while (result >= n);
	return result;
}

// ---------------------------------------------------------------------------
// (0,2)-sequence (src/samplers/ldsampler.cpp:104-118), integer parts
// ---------------------------------------------------------------------------
HD uint32_t vdcBits(uint32_t n, uint32_t scramble) {
	n = (n << 16) | (n >> 16);
	n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
	n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
	n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
	n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
	return n ^ scramble;
}
HD uint32_t sobol2Bits(uint32_t n, uint32_t scramble) {
	for (uint32_t v = 1u << 31; n != 0; n >>= 1, v ^= v >> 1)
		if (n & 1u) scramble ^= v;
	return scramble;
}
// (Float) n / (Float) 0x100000000LL -- can be exactly 1.0f
HD float u32ToUnit(uint32_t n) { return (float) n / 4294967296.0f; }

// radicalInverse (src/libcore/util.cpp:740-750)
HD float radicalInverse(int b, uint64_t i) {
	const float invB = (float) 1 / (float) b;
	float x = 0.0f, f = invB;
	while (i) {
		x += f * (float) (i % (uint64_t) b);
		i /= (uint64_t) b;
		f *= invB;
	}
	return x;
}

// ---------------------------------------------------------------------------
// Warps / frames / Fresnel (src/libcore/util.cpp:543-738)
// ---------------------------------------------------------------------------
// util.cpp:553-559
HD V3 squareToSphere(float sx, float sy) {
	float z = 1.0f - 2.0f * sy;
	float r = 1.0f - z * z;
	r = sqrtf(smax(0.0f, r));
	float phi = 2.0f * kPi * sx;
	float s, c; dsincos(phi, s, c);
	return V3(r * c, r * s, z);
}
// util.cpp:572-588
HD V3 squareToHemispherePSA(float sx, float sy) {
	float r = sqrtf(sx);
	float phi = 2.0f * kPi * sy;
	float s, c; dsincos(phi, s, c);
	float dirX = r * c, dirY = r * s;
	float z = sqrtf(1 - smin(1.0f, dirX * dirX + dirY * dirY));
	if (z == 0)
		return normalize(V3(dirX, dirY, kEpsilon));
	return V3(dirX, dirY, z);
}
// util.cpp:613-616
HD void squareToTriangle(float sx, float sy, float &bx, float &by) {
	float a = sqrtf(1.0f - sx);
	bx = 1 - a; by = a * sy;
}
// util.cpp:629-651
HD void squareToDiskConcentric(float sx, float sy, float &ox, float &oy) {
	const float r1 = 2.0f * sx - 1.0f, r2 = 2.0f * sy - 1.0f;
	float cx, cy;
	if (r1 == 0 && r2 == 0) { cx = 0; cy = 0; }
	else if (r1 > -r2) {
		if (r1 > r2) { cx = r1; cy = (kPi / 4.0f) * r2 / r1; }
		else { cx = r2; cy = (kPi / 4.0f) * (2.0f - r1 / r2); }
	} else {
		if (r1 < r2) { cx = -r1; cy = (kPi / 4.0f) * (4.0f + r2 / r1); }
		else { cx = -r2; cy = (kPi / 4.0f) * (6.0f - r1 / r2); }
	}
	float s, c; dsincos(cy, s, c);
	ox = cx * c; oy = cx * s;
}
// util.cpp:602-611
HD void coordinateSystem(V3 a, V3 &b, V3 &c) {
	if (fabsf(a.x) > fabsf(a.y)) {
		float invLen = 1.0f / sqrtf(a.x * a.x + a.z * a.z);
		b = V3(-a.z * invLen, 0.0f, a.x * invLen);
	} else {
		float invLen = 1.0f / sqrtf(a.y * a.y + a.z * a.z);
		b = V3(0.0f, -a.z * invLen, a.y * invLen);
	}
	c = cross(a, b);
}
// util.cpp:680-688
HD float fresnelDielectric(float cosTheta1, float cosTheta2, float etaI, float etaT) {
	float Rs = (etaI * cosTheta1 - etaT * cosTheta2) / (etaI * cosTheta1 + etaT * cosTheta2);
	float Rp = (etaT * cosTheta1 - etaI * cosTheta2) / (etaT * cosTheta1 + etaI * cosTheta2);
	return (Rs * Rs + Rp * Rp) / 2.0f;
}
// util.cpp:704-727
HD float fresnel(float cosThetaI, float etaExt, float etaInt) {
	float etaI = etaExt, etaT = etaInt;
	if (cosThetaI < 0.0f) { float t = etaI; etaI = etaT; etaT = t; }
	float sinThetaT = etaI / etaT * sqrtf(smax(0.0f, 1.0f - cosThetaI * cosThetaI));
	if (sinThetaT > 1.0f)
		return 1.0f;
	float cosThetaT = sqrtf(1.0f - sinThetaT * sinThetaT);
	return fresnelDielectric(fabsf(cosThetaI), cosThetaT, etaI, etaT);
}
// util.cpp:690-702 (one spectral channel)
HD float fresnelConductor1(float cosTheta, float eta, float k) {
	float tmp = (eta * eta + k * k) * (cosTheta * cosTheta);
	float rParl2 = (tmp - (eta * (2.0f * cosTheta)) + 1.0f) / (tmp + (eta * (2.0f * cosTheta)) + 1.0f);
	float tmpF = eta * eta + k * k;
	float rPerp2 = (tmpF - (eta * (2.0f * cosTheta)) + (cosTheta * cosTheta))
	             / (tmpF + (eta * (2.0f * cosTheta)) + (cosTheta * cosTheta));
	return (rParl2 + rPerp2) * (1.0f / 2.0f);
}

// ---------------------------------------------------------------------------
// TriAccel (include/mitsuba/render/triaccel.h)
// ---------------------------------------------------------------------------
// TriAccel::load, triaccel.h:63-95; out = 12 dwords (k, n_u, n_v, n_d, a_u, a_v, b_nu, b_nv, c_nu, c_nv, shape, prim)
HD int triAccelLoad(V3 A, V3 B, V3 C, uint32_t *out) {
	V3 b = C - A, c = B - A, N = cross(c, b);
	uint32_t k = 0;
	for (int j = 0; j < 3; j++)
		if (fabsf(N[j]) > fabsf(N[(int) k]))
			k = (uint32_t) j;
	const uint32_t u = (k + 1) % 3, v = (k + 2) % 3;   // waldModulo
	const float n_k = N[(int) k], denom = b[(int) u] * c[(int) v] - b[(int) v] * c[(int) u];
	float *f = reinterpret_cast<float *>(out);
	if (denom == 0) { out[0] = 3; return 1; }
	out[0] = k;
	f[1] = N[(int) u] / n_k;
	f[2] = N[(int) v] / n_k;
	f[3] = dot(A, N) / n_k;
	f[4] = A[(int) u];
	f[5] = A[(int) v];
	f[6] = b[(int) u] / denom;
	f[7] = -b[(int) v] / denom;
	f[8] = c[(int) v] / denom;
	f[9] = -c[(int) u] / denom;
	return 0;
}

} // namespace mg
