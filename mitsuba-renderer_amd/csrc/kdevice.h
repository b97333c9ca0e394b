// kdevice.h -- device-side helpers every translation unit of the kernels shares: lane index, wave-aggregated append,
// streaming (non-temporal) loads and stores, the sphere shape's ray tests.
#pragma once
#include "kernels.h"
#include "devmath.h"
#include <algorithm>

namespace mg {

// ===========================================================================
// small helpers
// ===========================================================================
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// Wave-aggregated append: every lane with pred gets a unique slot; one atomic per wave.
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t *counter) {
	const unsigned long long mask = __ballot(pred);
	if (mask == 0ull)
		return 0u;
	const uint32_t lane = lane_id();
	const uint32_t rank = (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
	const int leader = __ffsll((long long) mask) - 1;
	uint32_t base = 0;
	if ((int) lane == leader)
		base = atomicAdd(counter, (uint32_t) __popcll(mask));
	base = __shfl(base, leader);
	return base + rank;
}

__device__ __forceinline__ float sel3(float x, float y, float z, int axis) {
	return axis == 0 ? x : (axis == 1 ? y : z);
}

// MG_NT: non-temporal hints on data that is touched once per launch -- bit 0: path records, ray
// and id queues in k_trace; bit 1: leaf records in k_trace; bit 2: records and queues in k_shade -- so that they do not
// push the tree out of the L1 / L2
#ifndef MG_NT
#define MG_NT 4      // measured (64-spp C3 frame): bit 0 +12 ms, bit 1 +120 ms (the leaf records live in the L2), bit 2 -3.6 ms
#endif
typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
template <int BIT, typename T> __device__ __forceinline__ T ld_stream(const T *p) {
	if (MG_NT & BIT) {
		static_assert(sizeof(T) == 16 || sizeof(T) == 4, "16-byte or 4-byte objects");
		T out;
		if (sizeof(T) == 16) { const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4 *>(p)); __builtin_memcpy(&out, &v, 16); }
		else { const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p)); __builtin_memcpy(&out, &v, 4); }
		return out;
	}
	return *p;
}
template <int BIT, typename T> __device__ __forceinline__ void st_stream(T *p, const T &v) {
	if (MG_NT & BIT) {
		static_assert(sizeof(T) == 16 || sizeof(T) == 4, "16-byte or 4-byte objects");
		if (sizeof(T) == 16) { nt_u4 x; __builtin_memcpy(&x, &v, 16); __builtin_nontemporal_store(x, reinterpret_cast<nt_u4 *>(p)); }
		else { uint32_t x; __builtin_memcpy(&x, &v, 4); __builtin_nontemporal_store(x, reinterpret_cast<uint32_t *>(p)); }
	} else {
		*p = v;
	}
}

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned) ((n + bs - 1) / bs); }

// ===========================================================================
// Sphere shape (src/shapes/sphere.cpp).  SP = shape parameter block:
// [0..2] centre [3] radius [4] inverted [5..13] objectToWorld 3x3 [14..22] worldToObject 3x3 [23] 1/area
// ===========================================================================
// solveQuadratic (src/libcore/util.cpp:450-488)
__device__ __forceinline__ bool solve_quadratic(float a, float b, float c, float &x0, float &x1) {
	if (a == 0) {
		if (b != 0) { x0 = x1 = -c / b; return true; }
		return false;
	}
	const float discrim = b * b - 4.0f * a * c;
	if (discrim < 0)
		return false;
	const float sqrtDiscrim = sqrtf(discrim);
	float temp;
	if (b < 0) temp = -0.5f * (b - sqrtDiscrim);
	else       temp = -0.5f * (b + sqrtDiscrim);
	x0 = temp / a;
	x1 = c / temp;
	if (x0 > x1) { const float t = x0; x0 = x1; x1 = t; }
	return true;
}
// the quadratic of Sphere::rayIntersect (sphere.cpp:94-101)
__device__ __forceinline__ bool sphere_roots(V3 center, float radius, V3 ro, V3 rd, float &nearT, float &farT) {
	const V3 o = ro - center;
	const float A = rd.x * rd.x + rd.y * rd.y + rd.z * rd.z;
	const float B = 2 * (rd.x * o.x + rd.y * o.y + rd.z * o.z);
	const float C = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
	return solve_quadratic(A, B, C, nearT, farT);
}
// Sphere::rayIntersect(ray, mint, maxt, t, tmp) (sphere.cpp:94-116)
__device__ __forceinline__ bool sphere_intersect(V3 center, float radius, V3 ro, V3 rd, float mint, float maxt, float &t) {
	float nearT, farT;
	if (!sphere_roots(center, radius, ro, rd, nearT, farT))
		return false;
	if (nearT > maxt || farT < mint)
		return false;
	if (nearT < mint) {
		if (farT > maxt)
			return false;
		t = farT;
	} else {
		t = nearT;
	}
	return true;
}
// Sphere::rayIntersect(ray, mint, maxt) (sphere.cpp:118-134)
__device__ __forceinline__ bool sphere_occludes(V3 center, float radius, V3 ro, V3 rd, float mint, float maxt) {
	float nearT, farT;
	if (!sphere_roots(center, radius, ro, rd, nearT, farT))
		return false;
	if (nearT > maxt || farT < mint)
		return false;
	if (nearT < mint && farT > maxt)
		return false;
	return true;
}

} // namespace mg
