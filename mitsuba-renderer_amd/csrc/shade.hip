// shade.hip -- K3+K5: intersection record, luminaires, BSDFs and one iteration of MIPathTracer::Li per material queue.
#include "sampler.h"

namespace mg {

// ===========================================================================
// Intersection record, luminaires, BSDFs
// ===========================================================================
struct Its {
	V3 p, geoN, shS, shT, shN, wi;
	uint32_t shape;
};

// fillIntersectionRecord<true> (include/mitsuba/render/skdtree.h:352-432)
// t0, t1, t2: the first three chunks of the primitive's gather record (sc.tri_pos), fetched by the caller
__device__ __forceinline__ void fill_its(const DScene &sc, V3 rayO, V3 rayD, float t, uint32_t prim, float u, float v,
                                         const float4 t0, const float4 t1, const float4 t2, Its &its) {
	V3 sS, sT;
	if (__float_as_uint(t2.w) & 0x80000000u) {
		// Sphere::fillIntersectionRecord (src/shapes/sphere.cpp:136-178): its.p = ray(t), frame from dpdu / dpdv
		its.shape = __float_as_uint(t2.z);
		const float *SP = sc.shape_params + 24 * (size_t) its.shape;
		const float *O2W = SP + 5, *W2O = SP + 14;
		const V3 center(SP[0], SP[1], SP[2]);
		const float radius = SP[3];
		its.p = V3(rayO.x + t * rayD.x, rayO.y + t * rayD.y, rayO.z + t * rayD.z);
		const V3 pc = its.p - center;
		const V3 local(W2O[0] * pc.x + W2O[1] * pc.y + W2O[2] * pc.z, W2O[3] * pc.x + W2O[4] * pc.y + W2O[5] * pc.z,
		               W2O[6] * pc.x + W2O[7] * pc.y + W2O[8] * pc.z);
		const float theta = dacos(smin(smax(local.z / radius, -1.0f), 1.0f));
		const V3 du(-local.y * (2 * kPi), local.x * (2 * kPi), 0 * (2 * kPi));
		const V3 dpdu(O2W[0] * du.x + O2W[1] * du.y + O2W[2] * du.z, O2W[3] * du.x + O2W[4] * du.y + O2W[5] * du.z,
		              O2W[6] * du.x + O2W[7] * du.y + O2W[8] * du.z);
		V3 n = normalize(pc);
		const float zrad = sqrtf(local.x * local.x + local.y * local.y);
		if (zrad > 0) {
			const float invZRad = 1.0f / zrad, cosPhi = local.x * invZRad, sinPhi = local.y * invZRad;
			float st, ct;
			dsincos(theta, st, ct);
			const V3 dv((local.z * cosPhi) * kPi, (local.z * sinPhi) * kPi, (-st * radius) * kPi);
			const V3 dpdv(O2W[0] * dv.x + O2W[1] * dv.y + O2W[2] * dv.z, O2W[3] * dv.x + O2W[4] * dv.y + O2W[5] * dv.z,
			              O2W[6] * dv.x + O2W[7] * dv.y + O2W[8] * dv.z);
			sS = normalize(dpdu);
			sT = normalize(dpdv);
		} else {
			coordinateSystem(n, sS, sT);
		}
		if (SP[4] != 0.0f)
			n = V3(n.x * -1, n.y * -1, n.z * -1);
		its.geoN = n; its.shN = n;
	} else {
	const V3 p0(t0.x, t0.y, t0.z), p1(t0.w, t1.x, t1.y), p2(t1.z, t1.w, t2.x);
	const float bx = 1 - u - v, by = u, bz = v;
	its.p = V3(p0.x * bx + p1.x * by + p2.x * bz, p0.y * bx + p1.y * by + p2.y * bz, p0.z * bx + p1.z * by + p2.z * bz);
	V3 faceNormal = cross(p1 - p0, p2 - p0);
	const float len = length(faceNormal);
	if (!isZero(faceNormal))
		faceNormal = divs(faceNormal, len);
	its.geoN = faceNormal;
	its.shape = __float_as_uint(t2.z);
	if (__float_as_uint(t2.w) & 1u) {
		const float4 *TN = sc.tri_nrm + kTriStride * (size_t) prim;
		const float4 m0 = TN[0], m1 = TN[1], m2 = TN[2];
		const V3 n0(m0.x, m0.y, m0.z), n1(m0.w, m1.x, m1.y), n2(m1.z, m1.w, m2.x);
		its.shN = normalize(V3(n0.x * bx + n1.x * by + n2.x * bz, n0.y * bx + n1.y * by + n2.y * bz, n0.z * bx + n1.z * by + n2.z * bz));
	} else {
		its.shN = its.geoN;
	}
	coordinateSystem(its.shN, sS, sT);
	}
	its.shS = sS; its.shT = sT;
	const V3 md = -rayD;
	its.wi = V3(dot(md, its.shS), dot(md, its.shT), dot(md, its.shN));
}

struct LRec { V3 p, n, d, value; float pdf; int lum; };

// DiscretePDF::sample / sampleReuse (include/mitsuba/core/pdf.h:102-133)
__device__ __forceinline__ int dpdf_sample_reuse(const float *cdf, uint32_t n, float &sampleValue) {
	uint32_t lo = 0, count = n + 1;       // std::lower_bound over n + 1 knots
	while (count > 0) {
		const uint32_t step = count / 2, it = lo + step;
		if (cdf[it] < sampleValue) { lo = it + 1; count -= step + 1; }
		else count = step;
	}
	int index = (int) lo - 1;
	if (index < 0) index = 0;
	if (index > (int) n - 1) index = (int) n - 1;
	sampleValue = (sampleValue - cdf[index]) / (cdf[index + 1] - cdf[index]);
	return index;
}

// BSphere::rayIntersect (include/mitsuba/core/bsphere.h:85-118)
__device__ __forceinline__ bool bsphere_ray_intersect(V3 center, float radius, V3 o, V3 d, float &nearHit, float &farHit) {
	const V3 originToCenter = center - o;
	const float distToRayClosest = dot(originToCenter, d);
	const float tmp1 = dot(originToCenter, originToCenter) - radius * radius;
	if (tmp1 <= 0.0f) {
		nearHit = farHit = sqrtf(distToRayClosest * distToRayClosest - tmp1) + distToRayClosest;
		return true;
	}
	if (distToRayClosest < 0.0f)
		return false;
	const float sqrOriginToCenterLength = dot(originToCenter, originToCenter);
	const float sqrHalfChordDist = radius * radius - sqrOriginToCenterLength + distToRayClosest * distToRayClosest;
	if (sqrHalfChordDist < 0)
		return false;
	const float hitDistance = sqrtf(sqrHalfChordDist);
	nearHit = distToRayClosest - hitDistance;
	farHit = distToRayClosest + hitDistance;
	if (nearHit == 0)
		nearHit = farHit;
	return true;
}

// ---- EnvMapLuminaire (src/luminaires/envmap.cpp) ----
// MIPMap::triangle(0, x, y) with ERepeat (mipmap.cpp:226-243, getTexel :203-224)
__device__ __forceinline__ V3 env_triangle(const DScene &sc, float x, float y) {
	const int W = (int) sc.env_width, H = (int) sc.env_height;
	x = x * W - 0.5f;
	y = y * H - 0.5f;
	const int xPos = (int) floorf(x), yPos = (int) floorf(y);
	const float dx = x - xPos, dy = y - yPos;
	V3 acc(0, 0, 0);
	#pragma unroll
	for (int k = 0; k < 4; ++k) {
		int tx = xPos + (k >> 1), ty = yPos + (k & 1);
		if (tx <= 0 || ty < 0 || tx >= W || ty >= H) {
			int r = tx - (tx / W) * W; tx = (r < 0) ? r + W : r;               // modulo (util.cpp:424-427)
			r = ty - (ty / H) * H; ty = (r < 0) ? r + H : r;
		}
		const float *t = sc.env_pixels + 3 * ((size_t) tx + (size_t) W * ty);
		const float a = (k < 2) ? (1.0f - dx) : dx, b = (k & 1) ? dy : (1.0f - dy);
		const V3 term(t[0] * a * b, t[1] * a * b, t[2] * a * b);
		acc = (k == 0) ? term : V3(acc.x + term.x, acc.y + term.y, acc.z + term.z);
	}
	return acc;
}
// Le(direction) (envmap.cpp:147-153); LP = luminaire parameter block
__device__ __forceinline__ V3 env_le(const DScene &sc, const float *LP, V3 dir) {
	const float *M = LP + 7;
	const V3 d(M[0] * dir.x + M[1] * dir.y + M[2] * dir.z, M[3] * dir.x + M[4] * dir.y + M[5] * dir.z, M[6] * dir.x + M[7] * dir.y + M[8] * dir.z);
	const float u = .5f * (1 + datan2(d.x, -d.z) / kPi);
	const float v = dacos(smax(-1.0f, smin(1.0f, d.y))) / kPi;
	const V3 t = env_triangle(sc, u, v);
	return V3(t.x * LP[0], t.y * LP[0], t.z * LP[0]);
}
// pdf(p, lRec, delta) (envmap.cpp:176-193); ld = lRec.d
__device__ __forceinline__ float env_pdf(const DScene &sc, const float *LP, V3 ld) {
	const float *M = LP + 7;
	const V3 nd = -ld;
	const V3 d(M[0] * nd.x + M[1] * nd.y + M[2] * nd.z, M[3] * nd.x + M[4] * nd.y + M[5] * nd.z, M[6] * nd.x + M[7] * nd.y + M[8] * nd.z);
	const int rx = (int) sc.env_pdf_width, ry = (int) sc.env_pdf_height;
	const float x = .5f * (1 + datan2(d.x, -d.z) / kPi) * rx;
	const float y = dacos(smax(-1.0f, smin(1.0f, d.y))) / kPi * ry;
	int xPos = (int) floorf(x); xPos = xPos < 0 ? 0 : (xPos > rx - 1 ? rx - 1 : xPos);
	int yPos = (int) floorf(y); yPos = yPos < 0 ? 0 : (yPos > ry - 1 ? ry - 1 : yPos);
	const float pdf = sc.env_pdf[xPos + yPos * rx];
	const float sinTheta = sqrtf(smax(kEpsilon, 1 - d.y * d.y));
	const float psx = 2 * kPi / rx, psy = kPi / ry;
	return pdf / (psx * psy * sinTheta);
}

// Scene::sampleLuminaire without the visibility test (scene.cpp:396-415):
// returns true when a shadow ray has to be traced; value is already divided by pdf.
__device__ __forceinline__ bool sample_luminaire(const DScene &sc, V3 p, float s0, float s1, LRec &lRec) {
	float sx = s0, sy = s1;
	const int l = dpdf_sample_reuse(sc.lum_sel_cdf, sc.n_lums, sx);
	const float lumPdf = sc.lum_sel_pdf[l];
	const float *LP = sc.lum_params + kLumStride * (size_t) l;
	if (sc.lum_type[l] == 0u && sc.shape_type[sc.lum_shape[l]] == 1u) {
		// AreaLuminaire::sample (area.cpp:68-79) -> Sphere::sampleSolidAngle (src/shapes/sphere.cpp:196-237)
		const float *SP = sc.shape_params + 24 * (size_t) sc.lum_shape[l];
		const V3 center(SP[0], SP[1], SP[2]);
		const float radius = SP[3];
		const V3 w = center - p;
		const float invDistW = 1 / length(w);
		const float squareTerm = fabsf(radius * invDistW);
		if (squareTerm >= 1 - kEpsilon) {
			// inside the sphere: uniform sampling
			const V3 d = squareToSphere(sx, sy);
			lRec.p = V3(center.x + d.x * radius, center.y + d.y * radius, center.z + d.z * radius);
			lRec.n = d;
			const V3 lumToPoint = p - lRec.p;
			const float distSquared = dot(lumToPoint, lumToPoint), dp = dot(lumToPoint, lRec.n);
			lRec.pdf = (dp > 0) ? (SP[23] * distSquared * sqrtf(distSquared) / dp) : 0.0f;
		} else {
			const float cosThetaMax = sqrtf(smax(0.0f, 1 - squareTerm * squareTerm));
			// squareToCone (util.cpp:656-662)
			const float cosTheta = (1 - sx) + sx * cosThetaMax;
			const float sinTheta = sqrtf(1 - cosTheta * cosTheta);
			const float phi = sy * (2 * kPi);
			float sphi, cphi;
			dsincos(phi, sphi, cphi);
			const V3 cone(cphi * sinTheta, sphi * sinTheta, cosTheta);
			// Frame(w * invDistW).toWorld(cone)
			const V3 fn = w * invDistW;
			V3 fs, ft;
			coordinateSystem(fn, fs, ft);
			const V3 d(fs.x * cone.x + ft.x * cone.y + fn.x * cone.z, fs.y * cone.x + ft.y * cone.y + fn.y * cone.z,
			           fs.z * cone.x + ft.z * cone.y + fn.z * cone.z);
			float t;
			if (!sphere_intersect(center, radius, p, d, 0.0f, MG_INF, t)) {
				lRec.pdf = 0.0f;         // roundoff: no sample
			} else {
				lRec.p = V3(p.x + t * d.x, p.y + t * d.y, p.z + t * d.z);
				lRec.n = normalize(lRec.p - center);
				lRec.pdf = 1 / ((2 * kPi) * (1 - cosThetaMax));
			}
		}
		lRec.d = p - lRec.p;
		if (lRec.pdf > 0 && dot(lRec.d, lRec.n) > 0) {
			lRec.value = V3(LP[0], LP[1], LP[2]);
			lRec.d = normalize(lRec.d);
		} else {
			lRec.pdf = 0;
		}
	} else if (sc.lum_type[l] == 0u) {
		// AreaLuminaire::sample (area.cpp:68-79) -> Shape::sampleSolidAngle (shape.cpp:65-75)
		// -> TriMesh::sampleArea (trimesh.cpp:297-302) -> Triangle::sample (triangle.cpp:23-47)
		const uint32_t s = (uint32_t) sc.lum_shape[l];
		const uint32_t t0 = sc.shape_tri_offset[s], nT = sc.shape_tri_offset[s + 1] - t0;
		const int index = dpdf_sample_reuse(sc.lum_tri_cdf + sc.lum_cdf_offset[l], nT, sy);
		const size_t tri = (size_t) t0 + (uint32_t) index;
		const float4 *TP = sc.tri_pos + kTriStride * tri;
		const float4 q0 = TP[0], q1 = TP[1], q2 = TP[2];
		const V3 p0(q0.x, q0.y, q0.z), p1(q0.w, q1.x, q1.y), p2(q1.z, q1.w, q2.x);
		float bx, by;
		squareToTriangle(sx, sy, bx, by);
		const V3 sideA = p1 - p0, sideB = p2 - p0;
		lRec.p = V3(p0.x + (sideA.x * bx) + (sideB.x * by), p0.y + (sideA.y * bx) + (sideB.y * by), p0.z + (sideA.z * bx) + (sideB.z * by));
		if (__float_as_uint(q2.w) & 1u) {
			const float4 *TN = sc.tri_nrm + kTriStride * tri;
			const float4 m0 = TN[0], m1 = TN[1], m2 = TN[2];
			const V3 n0(m0.x, m0.y, m0.z), n1(m0.w, m1.x, m1.y), n2(m1.z, m1.w, m2.x);
			const float b0 = 1.0f - bx - by;
			lRec.n = normalize(V3(n0.x * b0 + n1.x * bx + n2.x * by, n0.y * b0 + n1.y * bx + n2.y * by, n0.z * b0 + n1.z * bx + n2.z * by));
		} else {
			lRec.n = normalize(cross(sideA, sideB));
		}
		const float pdfArea = sc.lum_inv_area[l];
		const V3 lumToPoint = p - lRec.p;
		const float distSquared = dot(lumToPoint, lumToPoint), dp = dot(lumToPoint, lRec.n);
		lRec.pdf = (dp > 0) ? (pdfArea * distSquared * sqrtf(distSquared) / dp) : 0.0f;
		lRec.d = p - lRec.p;
		if (lRec.pdf > 0 && dot(lRec.d, lRec.n) > 0) {
			lRec.value = V3(LP[0], LP[1], LP[2]);
			lRec.d = normalize(lRec.d);
		} else {
			lRec.pdf = 0;
		}
	} else if (sc.lum_type[l] == 2u || sc.lum_type[l] == 4u) {
		// PointLuminaire::sample (point.cpp:55-63) / SpotLuminaire::sample (spot.cpp:110-118)
		const V3 pos(LP[3], LP[4], LP[5]);
		const V3 lumToP = p - pos;
		const float invDist = 1.0f / length(lumToP);
		lRec.p = pos;
		lRec.d = lumToP * invDist;
		lRec.n = V3(0, 0, 0);
		lRec.pdf = 1.0f;
		V3 result(LP[0], LP[1], LP[2]);
		if (sc.lum_type[l] == 4u) {
			// falloffCurve (spot.cpp:84-103), constant texture; cosTheta = m_worldToLuminaire(d).z
			const float cosTheta = LP[16] * lRec.d.x + LP[17] * lRec.d.y + LP[18] * lRec.d.z;
			if (cosTheta <= LP[7]) result = V3(0, 0, 0);
			else if (!(cosTheta >= LP[6])) result = result * ((LP[8] - dacos(cosTheta)) * LP[9]);
		}
		lRec.value = result * (invDist * invDist);
	} else if (sc.lum_type[l] == 5u) {
		// EnvMapLuminaire::sampleDirection + sample (envmap.cpp:123-145, :159-172)
		const int rx = (int) sc.env_pdf_width, ry = (int) sc.env_pdf_height;
		const int idx = dpdf_sample_reuse(sc.env_cdf, (uint32_t) (rx * ry), sx);
		float pdf = sc.env_pdf[idx];
		const int row = idx / rx, col = idx - rx * row;
		const float x = col + sx, y = row + sy;
		const V3 tv = env_triangle(sc, x * (1.0f / rx), y * (1.0f / ry));
		const float psx = 2 * kPi / rx, psy = kPi / ry;
		const float theta = psy * y, phi = psx * x - kPi;
		float sinTheta, cosTheta, sinPhi, cosPhi;
		dsincos(theta, sinTheta, cosTheta); dsincos(phi, sinPhi, cosPhi);
		pdf = pdf / (psx * psy * sinTheta);
		const float *L2W = LP + 16;
		const V3 v(-sinTheta * sinPhi, -cosTheta, sinTheta * cosPhi);
		const V3 d(L2W[0] * v.x + L2W[1] * v.y + L2W[2] * v.z, L2W[3] * v.x + L2W[4] * v.y + L2W[5] * v.z, L2W[6] * v.x + L2W[7] * v.y + L2W[8] * v.z);
		lRec.pdf = pdf;
		lRec.value = V3(tv.x * LP[0], tv.y * LP[0], tv.z * LP[0]);
		const V3 center(LP[3], LP[4], LP[5]);
		const float radius = LP[6];
		float nearHit, farHit;
		if (length(p - center) <= radius && bsphere_ray_intersect(center, radius, p, -d, nearHit, farHit)) {
			lRec.p = V3(p.x - d.x * nearHit, p.y - d.y * nearHit, p.z - d.z * nearHit);
			lRec.n = normalize(center - lRec.p);
			lRec.d = d;
		} else {
			lRec.pdf = 0.0f;
		}
	} else if (sc.lum_type[l] == 6u) {
		// CollimatedBeamLuminaire::sample (src/luminaires/collimated.cpp:62-76)
		const float *Wm = LP + 4, *Lm = LP + 16;
		const V3 local(Wm[0] * p.x + Wm[1] * p.y + Wm[2] * p.z + Wm[3], Wm[4] * p.x + Wm[5] * p.y + Wm[6] * p.z + Wm[7],
		               Wm[8] * p.x + Wm[9] * p.y + Wm[10] * p.z + Wm[11]);
		if (sqrtf(local.x * local.x + local.y * local.y) > LP[3] || local.z < 0) {
			lRec.pdf = 0.0f;
		} else {
			lRec.p = V3(Lm[0] * local.x + Lm[1] * local.y + Lm[2] * 0.0f + Lm[3], Lm[4] * local.x + Lm[5] * local.y + Lm[6] * 0.0f + Lm[7],
			            Lm[8] * local.x + Lm[9] * local.y + Lm[10] * 0.0f + Lm[11]);
			lRec.d = V3(Lm[0] * 0.0f + Lm[1] * 0.0f + Lm[2] * 1.0f, Lm[4] * 0.0f + Lm[5] * 0.0f + Lm[6] * 1.0f, Lm[8] * 0.0f + Lm[9] * 0.0f + Lm[10] * 1.0f);
			lRec.n = V3(0, 0, 0);
			lRec.pdf = 1.0f;
			lRec.value = V3(LP[0], LP[1], LP[2]);
		}
	} else if (sc.lum_type[l] == 3u) {
		// DirectionalLuminaire::sample (directional.cpp:84-91)
		const V3 dir(LP[3], LP[4], LP[5]);
		const float k = 2 * LP[6];
		lRec.p = V3(p.x - dir.x * k, p.y - dir.y * k, p.z - dir.z * k);
		lRec.d = dir;
		lRec.n = V3(0, 0, 0);
		lRec.pdf = 1.0f;
		lRec.value = V3(LP[0], LP[1], LP[2]);
	} else {
		// ConstantLuminaire::sample (constant.cpp:73-87)
		const V3 d = squareToSphere(sx, sy);
		const V3 center(LP[3], LP[4], LP[5]);
		const float radius = LP[6];
		float nearHit, farHit;
		if (length(p - center) <= radius && bsphere_ray_intersect(center, radius, p, d, nearHit, farHit)) {
			lRec.p = V3(p.x + d.x * nearHit, p.y + d.y * nearHit, p.z + d.z * nearHit);
			lRec.pdf = 1.0f / (4 * kPi);
			lRec.n = normalize(center - lRec.p);
			lRec.d = -d;
			lRec.value = V3(LP[0], LP[1], LP[2]);
		} else {
			lRec.pdf = 0.0f;
		}
	}
	if (lRec.pdf != 0) {
		lRec.pdf *= lumPdf;
		const float recip = 1.0f / lRec.pdf;
		lRec.value = lRec.value * recip;
		lRec.lum = l;
		return true;
	}
	return false;
}

// Scene::pdfLuminaire (scene.cpp:381-394); Shape::pdfSolidAngle (shape.cpp:77-83); constant.cpp:89-91
__device__ __forceinline__ float pdf_luminaire(const DScene &sc, V3 p, int lum, V3 lp, V3 ln, V3 ld) {
	const float fraction = 1.0f / sc.lum_sel_sum;
	float pdf;
	if (sc.lum_type[lum] == 0u && sc.shape_type[sc.lum_shape[lum]] == 1u) {
		// Sphere::pdfSolidAngle (sphere.cpp:239-255)
		const float *SP = sc.shape_params + 24 * (size_t) sc.lum_shape[lum];
		const V3 w = p - V3(SP[0], SP[1], SP[2]);
		const float invDistW = 1 / length(w);
		const float squareTerm = fabsf(SP[3] * invDistW);
		if (squareTerm >= 1 - kEpsilon) {
			const V3 lumToPoint = p - lp;
			const float distSquared = dot(lumToPoint, lumToPoint), dp = dot(lumToPoint, ln);
			pdf = (dp > 0) ? (SP[23] * distSquared * sqrtf(distSquared) / dp) : 0.0f;
		} else {
			const float cosThetaMax = sqrtf(smax(0.0f, 1 - squareTerm * squareTerm));
			pdf = 1 / (2 * kPi * (1 - cosThetaMax));          // squareToConePdf (util.cpp:652-654)
		}
	} else if (sc.lum_type[lum] == 0u) {
		const V3 lumToPoint = p - lp;
		const float distSquared = dot(lumToPoint, lumToPoint);
		const float invDP = smax(0.0f, sqrtf(distSquared) / dot(lumToPoint, ln));
		pdf = sc.lum_inv_area[lum] * distSquared * invDP;
	} else if (sc.lum_type[lum] == 5u) {
		pdf = env_pdf(sc, sc.lum_params + kLumStride * (size_t) lum, ld);
	} else {
		pdf = 1.0f / (4 * kPi);
	}
	return pdf * fraction;
}

// --- BSDF building blocks (roughmetal.cpp:75-117 == microfacet.cpp:95-136) ---
enum : uint32_t { T_DIFFUSE_REFL = 0x1, T_DIFFUSE_TRANS = 0x2, T_DELTA_REFL = 0x4, T_DELTA_TRANS = 0x8, T_GLOSSY_REFL = 0x10, T_GLOSSY_TRANS = 0x20,
                  T_DELTA = 0xC, T_TRANSMISSION = 0x2A };

__device__ __forceinline__ float frame_tan_theta(V3 v) {      // frame.h:98-103
	const float temp = 1 - v.z * v.z;
	if (temp <= 0.0f) return 0.0f;
	return sqrtf(temp) / v.z;
}
__device__ __forceinline__ float beckmann_d(float alphaB, V3 m) {
	const float ex = frame_tan_theta(m) / alphaB;
	return dexp(-(ex * ex)) / (kPi * alphaB * alphaB * dpow4(m.z));
}
__device__ __forceinline__ V3 sample_beckmann_d(float alphaB, float sx, float sy) {
	const float thetaM = datan(sqrtf(-alphaB * alphaB * dlog(1.0f - sx)));
	const float phiM = (2.0f * kPi) * sy;
	float st, ct, sp, cp;
	dsincos(thetaM, st, ct); dsincos(phiM, sp, cp);
	return V3(st * cp, st * sp, ct);                           // sphericalDirection (util.cpp:543-550)
}
__device__ __forceinline__ float smith_g1(float alphaB, V3 v, V3 m) {
	if (dot(v, m) * v.z <= 0) return 0.0f;
	const float tanTheta = frame_tan_theta(v);
	if (tanTheta == 0.0f) return 1.0f;
	const float a = 1.0f / (alphaB * tanTheta);
	const float aSqr = a * a;
	if (a >= 1.6f) return 1.0f;
	return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
}
__device__ __forceinline__ V3 mf_reflect(V3 wi, V3 n) {
	const float s = 2.0f * dot(n, wi);
	return V3(n.x * s - wi.x, n.y * s - wi.y, n.z * s - wi.z);
}

template <int BT> struct Bsdf;

// Lambertian (src/bsdfs/lambertian.cpp:95-126)
template <> struct Bsdf<0> {
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return V3(0, 0, 0);
		return V3(P[0] * kInvPi, P[1] * kInvPi, P[2] * kInvPi);
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return 0.0f;
		return wo.z * kInvPi;
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdf, uint32_t &st) {
		pdf = 0; st = 0; wo = V3(0, 0, 0);
		if (wi.z <= 0) return V3(0, 0, 0);
		wo = squareToHemispherePSA(sx, sy);
		st = T_DIFFUSE_REFL;
		pdf = wo.z * kInvPi;
		return V3(P[0] * kInvPi, P[1] * kInvPi, P[2] * kInvPi);
	}
};

// Dielectric (src/bsdfs/dielectric.cpp:101-107, :200-261): f = pdf = 0, delta sampling
template <> struct Bsdf<1> {
	static __device__ __forceinline__ V3 f(const float *, V3, V3) { return V3(0, 0, 0); }
	static __device__ __forceinline__ float pdf(const float *, V3, V3) { return 0.0f; }
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdf, uint32_t &st) {
		const float cosThetaI = wi.z;
		float etaI = P[1], etaT = P[0];
		const bool entering = cosThetaI > 0.0f;
		if (!entering) { const float t = etaI; etaI = etaT; etaT = t; }
		const float eta = etaI / etaT, sinThetaTSqr = eta * eta * (1.0f - wi.z * wi.z);
		float Fr, cosThetaT = 0;
		if (sinThetaTSqr >= 1.0f) {
			Fr = 1.0f;
		} else {
			cosThetaT = sqrtf(1.0f - sinThetaTSqr);
			Fr = fresnelDielectric(fabsf(cosThetaI), cosThetaT, etaI, etaT);
			if (entering) cosThetaT = -cosThetaT;
		}
		if (sx <= Fr) {
			st = T_DELTA_REFL;
			wo = V3(-wi.x, -wi.y, wi.z);
			pdf = Fr * fabsf(wo.z);
			return V3(P[2] * Fr, P[3] * Fr, P[4] * Fr);
		} else {
			st = T_DELTA_TRANS;
			wo = V3(-eta * wi.x, -eta * wi.y, cosThetaT);
			pdf = (1 - Fr) * fabsf(wo.z);
			return V3(P[5] * (1 - Fr) * (eta * eta), P[6] * (1 - Fr) * (eta * eta), P[7] * (1 - Fr) * (eta * eta));
		}
	}
};

// RoughMetal (src/bsdfs/roughmetal.cpp:119-167) through BSDF::sample(bRec, pdf, s) (bsdf.cpp:37-48)
template <> struct Bsdf<2> {
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return V3(0, 0, 0);
		const V3 Hr = normalize(wi + wo);
		const float c = dot(wi, Hr);
		const V3 F(fresnelConductor1(c, P[1], P[4]), fresnelConductor1(c, P[2], P[5]), fresnelConductor1(c, P[3], P[6]));
		const float D = beckmann_d(P[0], Hr);
		const float G = smith_g1(P[0], wi, Hr) * smith_g1(P[0], wo, Hr);
		const float k = D * G / (4.0f * wi.z * wo.z);
		return V3(P[7] * (F.x * k), P[8] * (F.y * k), P[9] * (F.z * k));
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return 0.0f;
		const V3 Hr = normalize(wi + wo);
		const float dwhr_dwo = 1.0f / (4.0f * fabsf(dot(wo, Hr)));
		return beckmann_d(P[0], Hr) * Hr.z * dwhr_dwo;
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		pdfv = 0; st = 0; wo = V3(0, 0, 0);
		if (wi.z <= 0) return V3(0, 0, 0);
		const V3 m = sample_beckmann_d(P[0], sx, sy);
		wo = mf_reflect(wi, m);
		st = T_GLOSSY_REFL;
		if (wo.z <= 0) return V3(0, 0, 0);
		const V3 fv = f(P, wi, wo);
		const float p = pdf(P, wi, wo);
		const V3 qv = fv * (1.0f / p);          // sample() = f / pdf; zero -> pdf 0, value 0
		if (isZero(qv)) return V3(0, 0, 0);
		pdfv = p;
		return fv;
	}
};

// Microfacet (src/bsdfs/microfacet.cpp:151-269) through BSDF::sample(bRec, pdf, s)
template <> struct Bsdf<3> {
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return V3(0, 0, 0);
		const float alphaB = P[0], kd = P[1], ks = P[2], intIOR = P[3], extIOR = P[4];
		const V3 Hr = normalize(wi + wo);
		const float F = fresnel(dot(wi, Hr), extIOR, intIOR);
		const float D = beckmann_d(alphaB, Hr);
		const float G = smith_g1(alphaB, wi, Hr) * smith_g1(alphaB, wo, Hr);
		const float specRef = D * G / (4.0f * wi.z * wo.z);
		const float fk = F * ks;
		V3 r(0.0f + (P[8] * specRef) * fk, 0.0f + (P[9] * specRef) * fk, 0.0f + (P[10] * specRef) * fk);
		const float dk = kInvPi * (1 - F) * kd;
		r.x += P[5] * dk; r.y += P[6] * dk; r.z += P[7] * dk;
		return r;
	}
	static __device__ __forceinline__ float pdf_spec(const float *P, V3 wi, V3 wo) {
		const V3 Hr = normalize(wi + wo);
		return beckmann_d(P[0], Hr) * Hr.z / (4.0f * fabsf(dot(wo, Hr)));
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return 0.0f;
		const float kd = P[1], ks = P[2], intIOR = P[3], extIOR = P[4];
		float fr = fresnel(wi.z, extIOR, intIOR);
		fr = smin(smax(fr, 0.05f), 0.95f);
		const float diffuseSamplingWeight = (1 - fr) * kd;
		const float specularSamplingWeight = fr * ks;
		const float normalization = 1 / (diffuseSamplingWeight + specularSamplingWeight);
		return (specularSamplingWeight * pdf_spec(P, wi, wo) + diffuseSamplingWeight * (wo.z * kInvPi)) * normalization;
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		pdfv = 0; st = 0; wo = V3(0, 0, 0);
		if (wi.z <= 0) return V3(0, 0, 0);
		const float kd = P[1], ks = P[2], intIOR = P[3], extIOR = P[4];
		float fr = fresnel(wi.z, extIOR, intIOR);
		fr = smin(smax(fr, 0.05f), 0.95f);
		float diffuseSamplingWeight = (1 - fr) * kd;
		float specularSamplingWeight = fr * ks;
		const float normalization = 1 / (diffuseSamplingWeight + specularSamplingWeight);
		specularSamplingWeight *= normalization;
		diffuseSamplingWeight *= normalization;
		V3 qv(0, 0, 0);
		if (sx < specularSamplingWeight) {
			sx /= specularSamplingWeight;
			const V3 m = sample_beckmann_d(P[0], sx, sy);      // sampleSpecular (:203-218)
			wo = mf_reflect(wi, m);
			st = T_GLOSSY_REFL;
			if (wo.z <= 0) return V3(0, 0, 0);
			const float pdfValue = pdf(P, wi, wo);
			if (pdfValue == 0) return V3(0, 0, 0);
			qv = f(P, wi, wo) * (1.0f / pdfValue);
		} else {
			sx = (sx - specularSamplingWeight) / diffuseSamplingWeight;
			wo = squareToHemispherePSA(sx, sy);                // sampleLambertian (:224-229)
			st = T_DIFFUSE_REFL;
			qv = f(P, wi, wo) * (1.0f / pdf(P, wi, wo));
		}
		if (isZero(qv)) return V3(0, 0, 0);
		pdfv = pdf(P, wi, wo);
		return f(P, wi, wo);
	}
};

// Mirror (src/bsdfs/mirror.cpp:60-86): f = pdf = 0, delta reflection
template <> struct Bsdf<4> {
	static __device__ __forceinline__ V3 f(const float *, V3, V3) { return V3(0, 0, 0); }
	static __device__ __forceinline__ float pdf(const float *, V3, V3) { return 0.0f; }
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float, float, V3 &wo, float &pdf, uint32_t &st) {
		wo = V3(-wi.x, -wi.y, wi.z);
		st = T_DELTA_REFL;
		pdf = fabsf(wo.z);
		return V3(P[0], P[1], P[2]);
	}
};

// Phong (src/bsdfs/phong.cpp:104-212), parameters after Phong::configure, through BSDF::sample(bRec, pdf, s)
template <> struct Bsdf<5> {
	static constexpr float kInvTwoPi = 0.15915494309189533577f;
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return V3(0, 0, 0);
		const V3 R(-wi.x, -wi.y, wi.z);
		const float alpha = dot(R, wo);
		float specRef;
		if (alpha <= 0.0f) specRef = 0.0f;
		else specRef = (P[0] + 2) * kInvTwoPi * dpow(alpha, P[0]) * P[2];
		V3 r(0.0f + P[8] * specRef, 0.0f + P[9] * specRef, 0.0f + P[10] * specRef);
		const float dk = kInvPi * P[1];
		r.x += P[5] * dk; r.y += P[6] * dk; r.z += P[7] * dk;
		return r;
	}
	static __device__ __forceinline__ float pdf_spec(const float *P, V3 wi, V3 wo) {
		const V3 R(-wi.x, -wi.y, wi.z);
		const float alpha = dot(R, wo);
		float specPdf = dpow(alpha, P[0]) * (P[0] + 1.0f) / (2.0f * kPi);
		if (alpha <= 0) specPdf = 0;
		return specPdf;
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		if (wo.z <= 0 || wi.z <= 0) return 0.0f;
		return P[3] * pdf_spec(P, wi, wo) + P[4] * (wo.z * kInvPi);
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		pdfv = 0; st = 0; wo = V3(0, 0, 0);
		if (wi.z <= 0) return V3(0, 0, 0);
		V3 qv(0, 0, 0);
		if (sx <= P[3]) {
			sx /= P[3];
			const V3 R(-wi.x, -wi.y, wi.z);                     // sampleSpecular (:157-182)
			const float sinAlpha = sqrtf(1 - dpow(sy, 2 / (P[0] + 1)));
			const float cosAlpha = dpow(sy, 1 / (P[0] + 1));
			const float phi = (2.0f * kPi) * sx;
			float sp, cp; dsincos(phi, sp, cp);
			const V3 l(sinAlpha * cp, sinAlpha * sp, cosAlpha);
			V3 fs, ft;
			coordinateSystem(R, fs, ft);                         // Frame(R).toWorld(localDir)
			wo = V3(fs.x * l.x + ft.x * l.y + R.x * l.z, fs.y * l.x + ft.y * l.y + R.y * l.z, fs.z * l.x + ft.z * l.y + R.z * l.z);
			st = T_GLOSSY_REFL;
			if (wo.z <= 0) return V3(0, 0, 0);
			const float pdfVal = pdf(P, wi, wo);
			if (pdfVal == 0) return V3(0, 0, 0);
			qv = f(P, wi, wo) * (1.0f / pdfVal);
		} else {
			sx = (sx - P[3]) / P[4];
			wo = squareToHemispherePSA(sx, sy);                  // sampleDiffuse (:188-193)
			st = T_DIFFUSE_REFL;
			qv = f(P, wi, wo) * (1.0f / pdf(P, wi, wo));
		}
		if (isZero(qv)) return V3(0, 0, 0);
		pdfv = pdf(P, wi, wo);
		return f(P, wi, wo);
	}
};

// RoughGlass (src/bsdfs/roughglass.cpp) through BSDF::sample(bRec, pdf, s) (bsdf.cpp:37-48): the plugin's own
// 3-argument sample() takes its pdf by value (roughglass.cpp:619) and therefore does not override the virtual.
// path.cpp leaves bRec.sampler NULL (clamped Fresnel term of the surface normal), quantity = ERadiance.
// P: [0] distribution (0 beckmann, 1 phong, 2 ggx) [1] alpha [2] intIOR [3] extIOR [4..6] specRefl [7..9] specTrans
template <> struct Bsdf<6> {
	static constexpr float kInvTwoPi = 0.15915494309189533577f;
	static __device__ __forceinline__ float signum(float v) { return (v < 0) ? -1.0f : 1.0f; }
	// evalD (roughglass.cpp:213-257)
	static __device__ __forceinline__ float evalD(int distr, V3 m, float alpha) {
		if (m.z <= 0) return 0.0f;
		float result;
		if (distr == 0) {
			const float ex = frame_tan_theta(m) / alpha;
			result = dexp(-(ex * ex)) / (kPi * alpha * alpha * dpow4(m.z));
		} else if (distr == 1) {
			result = (alpha + 2) * kInvTwoPi * dpow(m.z, alpha);
		} else {
			const float tanTheta = frame_tan_theta(m), cosTheta = m.z;
			const float root = alpha / (cosTheta * cosTheta * (alpha * alpha + tanTheta * tanTheta));
			result = kInvPi * (root * root);
		}
		if ((double) result < 1e-40) result = 0;
		return result;
	}
	// sampleD (roughglass.cpp:266-293) + sphericalDirection (util.cpp:543-550)
	static __device__ __forceinline__ V3 sampleD(int distr, float sx, float sy, float alpha) {
		const float phiM = (2.0f * kPi) * sy;
		float thetaM;
		if (distr == 0) thetaM = datan(sqrtf(-alpha * alpha * dlog(1.0f - sx)));
		else if (distr == 1) thetaM = dacos(dpow(sx, (float) 1 / (alpha + 2)));
		else thetaM = datan(alpha * sqrtf(sx) / sqrtf(1.0f - sx));
		float st, ct, sp, cp;
		dsincos(thetaM, st, ct); dsincos(phiM, sp, cp);
		return V3(st * cp, st * sp, ct);
	}
	// smithG1 (roughglass.cpp:303-343)
	static __device__ __forceinline__ float smithG1(int distr, V3 v, V3 m, float alpha) {
		const float tanTheta = fabsf(frame_tan_theta(v));
		if (tanTheta == 0.0f) return 1.0f;
		if (dot(v, m) * v.z <= 0) return 0.0f;
		if (distr == 2) {
			const float root = alpha * tanTheta;
			return 2.0f / (1.0f + sqrtf(1.0f + root * root));
		}
		if (distr == 1) alpha = sqrtf(0.5f * alpha + 1) / tanTheta;     // falls through to the Beckmann case
		const float a = 1.0f / (alpha * tanTheta);
		const float aSqr = a * a;
		if (a >= 1.6f) return 1.0f;
		return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
	}
	// the half-vector of f() and pdf() (roughglass.cpp:355-377 == :417-446)
	static __device__ __forceinline__ V3 halfVector(const float *P, V3 wi, V3 wo, bool reflect, float etaI, float etaT) {
		if (reflect)
			return normalize(wo + wi) * signum(wo.z);
		const V3 n = normalize(V3(wi.x * etaI + wo.x * etaT, wi.y * etaI + wo.y * etaT, wi.z * etaI + wo.z * etaT));
		const float sgn = (P[3] > P[2]) ? 1.0f : -1.0f;
		return V3(sgn * n.x, sgn * n.y, sgn * n.z);
	}
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		const int distr = (int) P[0];
		const bool reflect = wi.z * wo.z > 0;
		float etaI = P[3], etaT = P[2];
		if (wi.z < 0) { const float t = etaI; etaI = etaT; etaT = t; }
		const V3 H = halfVector(P, wi, wo, reflect, etaI, etaT);
		const float alpha = P[1];
		const float D = evalD(distr, H, alpha);
		if (D == 0) return V3(0, 0, 0);
		const float F = fresnel(dot(wi, H), P[3], P[2]);
		const float G = smithG1(distr, wi, H, alpha) * smithG1(distr, wo, H, alpha);
		if (reflect) {
			const float value = F * D * G / (4.0f * wi.z * wo.z);
			return V3(P[4] * value, P[5] * value, P[6] * value);
		}
		const float sqrtDenom = etaI * dot(wi, H) + etaT * dot(wo, H);
		float value = ((1 - F) * D * G * etaT * etaT * dot(wi, H) * dot(wo, H)) / (wi.z * wo.z * sqrtDenom * sqrtDenom);
		value *= (etaI * etaI) / (etaT * etaT);                     // bRec.quantity == ERadiance
		const float av = fabsf(value);
		return V3(P[7] * av, P[8] * av, P[9] * av);
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		const int distr = (int) P[0];
		const bool reflect = wi.z * wo.z > 0;
		float etaI = P[3], etaT = P[2];
		if (wi.z < 0) { const float t = etaI; etaI = etaT; etaT = t; }
		const V3 H = halfVector(P, wi, wo, reflect, etaI, etaT);
		float dwh_dwo;
		if (reflect) {
			dwh_dwo = 1.0f / (4.0f * dot(wo, H));
		} else {
			const float sqrtDenom = etaI * dot(wi, H) + etaT * dot(wo, H);
			dwh_dwo = (etaT * etaT * dot(wo, H)) / (sqrtDenom * sqrtDenom);
		}
		float alpha = P[1];
		alpha = alpha * (1.2f - 0.2f * sqrtf(fabsf(wi.z)));
		float prob = evalD(distr, H, alpha);
		const float F = smin(0.9f, smax(0.1f, fresnel(wi.z, P[3], P[2])));
		prob *= reflect ? F : (1 - F);
		return fabsf(prob * H.z * dwh_dwo);
	}
	// sample(bRec, sample) (roughglass.cpp:487-617), then pdf() and f() as BSDF::sample(bRec, pdf, s) does
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		pdfv = 0; st = 0; wo = V3(0, 0, 0);
		const int distr = (int) P[0];
		bool choseReflection = true;
		float sampleF = smin(0.9f, smax(0.1f, fresnel(wi.z, P[3], P[2])));
		if (sx < sampleF) {
			sx /= sampleF;
		} else {
			sx = (sx - sampleF) / (1 - sampleF);
			choseReflection = false;
		}
		const float alpha = P[1];
		const float sampleAlpha = alpha * (1.2f - 0.2f * sqrtf(fabsf(wi.z)));
		const V3 m = sampleD(distr, sx, sy, sampleAlpha);
		V3 result;
		if (choseReflection) {
			const float k = 2 * dot(wi, m);                          // reflect (roughglass.cpp:180-182)
			wo = V3(k * m.x - wi.x, k * m.y - wi.y, k * m.z - wi.z);
			st = T_GLOSSY_REFL;
			if (wi.z * wo.z <= 0) return V3(0, 0, 0);
			result = V3(P[4], P[5], P[6]);
		} else {
			float etaI = P[3], etaT = P[2];
			if (wi.z < 0) { const float t = etaI; etaI = etaT; etaT = t; }
			const float eta = etaI / etaT, c = dot(wi, m);           // refract (roughglass.cpp:185-201)
			const float cosThetaTSqr = 1 + eta * eta * (c * c - 1);
			if (cosThetaTSqr < 0) return V3(0, 0, 0);
			const float k = eta * c - signum(wi.z) * sqrtf(cosThetaTSqr);
			wo = V3(m.x * k - wi.x * eta, m.y * k - wi.y * eta, m.z * k - wi.z * eta);
			st = T_GLOSSY_TRANS;
			if (wi.z * wo.z >= 0) return V3(0, 0, 0);
			const float scale = (etaI * etaI) / (etaT * etaT);
			result = V3(P[7] * scale, P[8] * scale, P[9] * scale);
		}
		float numerator = evalD(distr, m, alpha) * smithG1(distr, wi, m, alpha) * smithG1(distr, wo, m, alpha) * dot(wi, m);
		float denominator = evalD(distr, m, sampleAlpha) * m.z * wi.z * wo.z;
		float F = fresnel(dot(wi, m), P[3], P[2]);
		if (!choseReflection) {
			sampleF = 1 - sampleF;
			F = 1 - F;
		}
		numerator *= F;
		denominator *= sampleF;
		const float w = fabsf(numerator / denominator);
		const V3 qv(result.x * w, result.y * w, result.z * w);
		if (isZero(qv)) return V3(0, 0, 0);
		pdfv = pdf(P, wi, wo);
		return f(P, wi, wo);
	}
};

// DiffuseTransmitter (src/bsdfs/difftrans.cpp:92-131)
template <> struct Bsdf<7> {
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z * wo.z >= 0) return V3(0, 0, 0);
		return V3(P[0] * kInvPi, P[1] * kInvPi, P[2] * kInvPi);
	}
	static __device__ __forceinline__ float pdf(const float *, V3 wi, V3 wo) {
		if (wi.z * wo.z >= 0) return 0.0f;
		return fabsf(wo.z) * kInvPi;
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		wo = squareToHemispherePSA(sx, sy);
		if (wi.z > 0) wo.z *= -1;
		st = T_DIFFUSE_TRANS;
		pdfv = fabsf(wo.z) * kInvPi;
		if (wo.z == 0) return V3(0, 0, 0);
		return V3(P[0] * kInvPi, P[1] * kInvPi, P[2] * kInvPi);
	}
};

// "terminal" bin: never evaluated
template <> struct Bsdf<kNumBsdfTypes> {
	static __device__ __forceinline__ V3 f(const float *, V3, V3) { return V3(0, 0, 0); }
	static __device__ __forceinline__ float pdf(const float *, V3, V3) { return 0.0f; }
	static __device__ __forceinline__ V3 sample(const float *, V3, float, float, V3 &wo, float &pdf, uint32_t &st) {
		wo = V3(0, 0, 0); pdf = 0; st = 0; return V3(0, 0, 0);
	}
};

// TwoSidedBRDF adapter (src/bsdfs/twosided.cpp:80-130) around any BSDF whose type carries MTSGPU_BSDF_TWOSIDED
template <int BT> struct Bsdf2 {
	static __device__ __forceinline__ V3 f(bool two, const float *P, V3 wi, V3 wo) {
		if (two && wi.z < 0) { wi.z *= -1; wo.z *= -1; }
		return Bsdf<BT>::f(P, wi, wo);
	}
	static __device__ __forceinline__ float pdf(bool two, const float *P, V3 wi, V3 wo) {
		if (two && wi.z < 0) { wi.z *= -1; wo.z *= -1; }
		return Bsdf<BT>::pdf(P, wi, wo);
	}
	static __device__ __forceinline__ V3 sample(bool two, const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdf, uint32_t &st) {
		bool flipped = false;
		if (two && wi.z < 0) { wi.z *= -1; flipped = true; }
		const V3 result = Bsdf<BT>::sample(P, wi, sx, sy, wo, pdf, st);
		if (flipped && !isZero(result) && pdf != 0) wo.z *= -1;
		return result;
	}
};

// BSDF::f / BSDF::pdf / BSDF::sample(bRec, pdf, sample) read out for tests (mtsgpu_bsdf_eval): the chi-square procedure of
// src/tests/test_chisquare.cpp:299-420 runs against exactly the code k_shade runs.  One query record per thread:
// wi = q[i][0..2]; op 0 / 1: wo = q[i][3..5]; op 2: sample = q[i][3..4].
template <int BT>
__device__ __forceinline__ void bsdf_eval_one(bool two, const float *P, int op, const float *q, float *o) {
	const V3 wi(q[0], q[1], q[2]);
	if (op == 0) {
		const V3 f = Bsdf2<BT>::f(two, P, wi, V3(q[3], q[4], q[5]));
		o[0] = f.x; o[1] = f.y; o[2] = f.z;
	} else if (op == 1) {
		o[0] = Bsdf2<BT>::pdf(two, P, wi, V3(q[3], q[4], q[5]));
	} else {
		V3 wo; float pdf; uint32_t st;
		const V3 f = Bsdf2<BT>::sample(two, P, wi, q[3], q[4], wo, pdf, st);
		o[0] = wo.x; o[1] = wo.y; o[2] = wo.z; o[3] = pdf; o[4] = f.x; o[5] = f.y; o[6] = f.z; o[7] = __uint_as_float(st);
	}
}
struct BsdfParams { float v[kBsdfNParams]; };
__global__ void k_bsdf_eval(uint32_t type, BsdfParams params, int op, uint32_t n, const float *queries, float *out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const bool two = (type & 0x100u) != 0;
	const float *P = params.v, *q = queries + 6 * (size_t) i;
	float o[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	switch (type & 0xFFu) {
		case 0: bsdf_eval_one<0>(two, P, op, q, o); break;
		case 1: bsdf_eval_one<1>(two, P, op, q, o); break;
		case 2: bsdf_eval_one<2>(two, P, op, q, o); break;
		case 3: bsdf_eval_one<3>(two, P, op, q, o); break;
		case 4: bsdf_eval_one<4>(two, P, op, q, o); break;
		case 5: bsdf_eval_one<5>(two, P, op, q, o); break;
		case 6: bsdf_eval_one<6>(two, P, op, q, o); break;
		default: bsdf_eval_one<7>(two, P, op, q, o); break;
	}
	#pragma unroll
	for (int k = 0; k < 8; ++k) out[8 * (size_t) i + k] = o[k];
}

__device__ __forceinline__ float mi_weight(float pdfA, float pdfB) {     // path.cpp:218-222
	pdfA *= pdfA;
	pdfB *= pdfB;
	return pdfA / (pdfA + pdfB);
}

// ===========================================================================
// K3+K5: one iteration of the loop of MIPathTracer::Li (path.cpp:61-209) for
// all paths whose current hit has BSDF type BT.  The tail of the previous
// iteration (emitter hit by the BSDF sample, Russian roulette, throughput
// update; path.cpp:171-208) runs first because it needs the new hit.
// ===========================================================================
// ROUNDS: the instantiation the rounds of MIDirectIntegrator use (DConfig::dr_mode != 0); the path tracer and the
// one-sample direct integrator run the one without that code
// The iteration for ONE path (id): what it leaves behind in registers is whether the path continues and its pending
// direct-light term with the shadow ray that guards it.
// The path's 128-byte record is staged in LDS by k_shade (`row`, slot k = record slot k): ro / rd / h / T4 / L4 were
// read from it already and, for a valid hit, slots 0, 1, 3 now hold the primitive's position chunks (see k_shade).
// What changes is written back to `row`: ray_o, ray_d, bsdf when the path continues; thr, Li, misc always.
// A staged path record in LDS: slot k of lane l lives at column k ^ (l & 7) of the lane's 8-slot row when the rows
// are packed (MG_SHADE_PACKED: 128 B per lane, what 5 waves per SIMD can afford; two lanes share a bank group), or at
// column k of a 9-slot row (144 B per lane, conflict-free)
#ifndef MG_SHADE_PACKED
#define MG_SHADE_PACKED 0
#endif
constexpr int kRowStride = MG_SHADE_PACKED ? kPathSlots : kPathSlots + 1;
struct ShadeRow {
	float4 *base; uint32_t x;
	__device__ __forceinline__ float4 &operator[](int k) const { return base[MG_SHADE_PACKED ? ((uint32_t) k ^ x) : (uint32_t) k]; }
};
__device__ __forceinline__ uint32_t shade_row_index(uint32_t lane, uint32_t k) { return lane * kRowStride + (MG_SHADE_PACKED ? (k ^ (lane & 7u)) : k); }

template <int BT, bool ROUNDS>
__device__ __forceinline__ void shade_path(const DScene &sc, const DPaths &ps, const DConfig &cfg, const uint32_t id,
                                           const float4 ro, const float4 rd, const uint4 h, const float4 T4, const float4 L4,
                                           const ShadeRow row, bool &continues, bool &wantShadow, V3 &neeV, V3 &shO, V3 &shD) {
	{
		// rounds of MIDirectIntegrator (DConfig::dr_mode): later BSDF samples start again from the camera hit
		const int mode = ROUNDS ? cfg.dr_mode : 0;
		const bool skipToNee = ROUNDS && mode == 1 && cfg.dr_index > 0, skipToBsdf = ROUNDS && mode == 2;
		const V3 rayO(ro.x, ro.y, ro.z), rayD(rd.x, rd.y, rd.z);
		const bool valid = h.w != kNoPrim;
		V3 thr(T4.x, T4.y, T4.z), Li(L4.x, L4.y, L4.z);
		int depth = __float_as_int(T4.w);
		uint32_t flags = __float_as_uint(L4.w);
		PathSampler smp;
		uint2 misc_zw;
		{
			const uint4 r = reinterpret_cast<const uint4 &>(row[6]);
			smp.stream = (uint64_t) r.x | ((uint64_t) r.y << 32);
			smp.slot = cfg.slot_per_path ? id : (id / cfg.spp);
			smp.j = r.z;
			misc_zw = make_uint2(r.z, r.w);
			smp.d1 = (flags >> F_D1_SHIFT) & 0xFFu; smp.d2 = (flags >> F_D2_SHIFT) & 0xFFu;
		}
		const bool direct = cfg.integrator == 1;
		Its its;
		if (valid)
			fill_its(sc, rayO, rayD, __uint_as_float(h.x), h.w, __uint_as_float(h.y), __uint_as_float(h.z), row[0], row[1], row[3], its);
		const int shapeLum = valid ? sc.shape_lum[its.shape] : -1;

		do {
			if (skipToNee || skipToBsdf) {
				// nothing before the sampling loops runs again
			} else if (flags & F_FIRST) {
				// rRec.rayIntersect (records.inl:89-105): alpha = 1 on a hit
				flags &= ~F_FIRST;
				if (valid) flags |= F_ALPHA;
				// while (rRec.depth <= m_maxDepth || m_maxDepth < 0) with depth == 1 (path.cpp:61): maxDepth == 0 never
				// enters the loop, the sample is black with the alpha of the camera ray
				if (!direct && !(depth <= cfg.max_depth || cfg.max_depth < 0))
					break;
			} else {
				// ---- tail of the previous iteration (path.cpp:147-208) ----
				const float4 B4 = row[5];
				const V3 bsdfVal(B4.x, B4.y, B4.z);
				const float bsdfPdf = B4.w;
				const uint32_t sampledType = flags >> F_ST_SHIFT;
				bool hitLuminaire = false;
				V3 lvalue(0, 0, 0), lp(0, 0, 0), ln(0, 0, 0);
				int llum = -1;
				if (valid) {
					if (shapeLum >= 0) {
						// LuminaireSamplingRecord(its, -ray.d); value = its.Le(-ray.d) (area.cpp:62-66)
						const float *LP = sc.lum_params + kLumStride * (size_t) shapeLum;
						lp = its.p; ln = its.geoN; llum = shapeLum;
						lvalue = (dot(-rayD, its.geoN) <= 0) ? V3(0, 0, 0) : V3(LP[0], LP[1], LP[2]);
						hitLuminaire = true;
					}
				} else {
					if (sc.background_lum >= 0) {
						const float *LP = sc.lum_params + kLumStride * (size_t) sc.background_lum;
						llum = sc.background_lum;
						lvalue = (sc.lum_type[llum] == 5u) ? env_le(sc, LP, normalize(rayD)) : V3(LP[0], LP[1], LP[2]);
						hitLuminaire = true;
					} else {
						if (!direct) depth++;
						break;
					}
				}
				if (hitLuminaire) {
					const float lumPdf = (!(sampledType & T_DELTA)) ? pdf_luminaire(sc, rayO, llum, lp, ln, -rayD) : 0.0f;
					// direct.cpp:189-191 weighs the two strategies by their sample counts
					const float weight = direct ? mi_weight(bsdfPdf * cfg.frac_bsdf, lumPdf * cfg.frac_lum) * cfg.weight_bsdf
					                            : mi_weight(bsdfPdf, lumPdf);
					Li.x += thr.x * lvalue.x * bsdfVal.x * weight;
					Li.y += thr.y * lvalue.y * bsdfVal.y * weight;
					Li.z += thr.z * lvalue.z * bsdfVal.z * weight;
				}
				if (!valid || direct)
					break;                                  // MIDirectIntegrator stops after its BSDF sample (direct.cpp:193)
				flags &= ~F_EMITTED;                       // rRec.type = ERadianceNoEmission
				if (depth >= cfg.rr_depth && !(sampledType & T_TRANSMISSION)) {
					const float approxAlbedo = smin(0.9f, smax(smax(bsdfVal.x, bsdfVal.y), bsdfVal.z));
					if (sampler_next1d(cfg, smp) > approxAlbedo)
						break;
					thr = thr * (1.0f / approxAlbedo);
				}
				thr = thr * bsdfVal;
				depth++;
				if (!(depth <= cfg.max_depth || cfg.max_depth < 0))
					break;
			}

			// ---- head of the iteration (path.cpp:62-98) ----
			if (!valid) {
				if (skipToNee || skipToBsdf) break;
				if ((flags & F_EMITTED) && sc.background_lum >= 0) {
					const float *LP = sc.lum_params + kLumStride * (size_t) sc.background_lum;
					const V3 le = (sc.lum_type[sc.background_lum] == 5u) ? env_le(sc, LP, normalize(rayD)) : V3(LP[0], LP[1], LP[2]);
					Li.x += thr.x * le.x; Li.y += thr.y * le.y; Li.z += thr.z * le.z;
				}
				break;
			}
			if (BT == kNumBsdfTypes)
				break;                                      // bsdf == NULL (path.cpp:72-77)
			const int bsdfIdx = sc.shape_bsdf[its.shape];
			const float *BP = sc.bsdf_params + 16 * (size_t) bsdfIdx;
			const bool twoSided = (sc.bsdf_type[bsdfIdx] & 0x100u) != 0;
			if (shapeLum >= 0 && (flags & F_EMITTED) && !(skipToNee || skipToBsdf)) {
				// Li += pathThroughput * its.Le(-ray.d) (path.cpp:80-81, area.cpp:62-66)
				const float *LP = sc.lum_params + kLumStride * (size_t) shapeLum;
				const V3 le = (dot(-rayD, its.geoN) <= 0) ? V3(0.0f, 0.0f, 0.0f) : V3(LP[0], LP[1], LP[2]);
				Li.x += thr.x * le.x; Li.y += thr.y * le.y; Li.z += thr.z * le.z;
			}
			if (!direct) {      // MonteCarloIntegrator properties; the direct integrator has neither (direct.cpp:33-41)
				if (cfg.max_depth > 0 && depth >= cfg.max_depth)
					break;
				const float wiDotGeoN = -dot(its.geoN, rayD), wiDotShN = its.wi.z;
				if (wiDotGeoN * wiDotShN < 0 && cfg.strict_normals)
					break;
			}
			const bool strict = cfg.strict_normals && !direct;

			// ---- luminaire sampling (path.cpp:100-126) ----
			if (!skipToBsdf) {
				float s0, s1;
				if (ROUNDS && direct && cfg.n_lum > 1) sampler_array2d(cfg, smp, misc_zw.y, 0, (uint32_t) cfg.dr_index, s0, s1);   // direct.cpp:122-123
				else sampler_next2d(cfg, smp, s0, s1);
				LRec lRec;
				if ((!direct || cfg.n_lum > 0) && sample_luminaire(sc, its.p, s0, s1, lRec)) {
					const V3 wo = -lRec.d;
					const V3 woL(dot(wo, its.shS), dot(wo, its.shT), dot(wo, its.shN));
					V3 bsdfVal = Bsdf2<BT>::f(twoSided, BP, its.wi, woL) * fabsf(woL.z);
					const float woDotGeoN = dot(its.geoN, wo);
					if (!isZero(bsdfVal) && (!strict || woDotGeoN * woL.z > 0)) {
						// isIntersectable() || isBackgroundLuminaire() (path.cpp:118-120): 0 for delta luminaires
						const uint32_t lt = sc.lum_type[lRec.lum];      // area, constant and envmap luminaires can be hit by BSDF samples
						const float bsdfPdf = (lt <= 1u || lt == 5u) ? Bsdf2<BT>::pdf(twoSided, BP, its.wi, woL) : 0.0f;
						const float weight = direct ? mi_weight(lRec.pdf * cfg.frac_lum, bsdfPdf * cfg.frac_bsdf) * cfg.weight_lum
						                            : mi_weight(lRec.pdf, bsdfPdf);          // direct.cpp:143-145
						// added to Li by k_trace<shadow> iff the segment is unoccluded
						// (kept in registers until the shadow-queue slot of this path is known, see the end of the kernel)
						neeV = V3(thr.x * lRec.value.x * bsdfVal.x * weight,
						          thr.y * lRec.value.y * bsdfVal.y * weight,
						          thr.z * lRec.value.z * bsdfVal.z * weight);
						shO = its.p;
						shD = lRec.p - its.p;               // Ray(p1, p2 - p1) (scene.h:241-246)
						wantShadow = true;
					}
				}
			}

			if (mode == 1)
				break;                                      // a luminaire round ends here

			// ---- BSDF sampling (path.cpp:128-146) ----
			float s0, s1;
			if (ROUNDS && direct && cfg.n_bsdf > 1) sampler_array2d(cfg, smp, misc_zw.y, cfg.n_lum > 1 ? 1 : 0, (uint32_t) cfg.dr_index, s0, s1);   // direct.cpp:156-157
			else sampler_next2d(cfg, smp, s0, s1);
			if (direct && cfg.n_bsdf <= 0)
				break;                                      // the sample is drawn even when it is not used (direct.cpp:156-161)
			V3 woL; float bsdfPdf; uint32_t sampledType;
			V3 bsdfVal = Bsdf2<BT>::sample(twoSided, BP, its.wi, s0, s1, woL, bsdfPdf, sampledType);
			if (!isZero(bsdfVal))
				bsdfVal = bsdfVal * fabsf(woL.z);          // sampleCos (bsdf.h:273-279)
			if (isZero(bsdfVal))
				break;
			bsdfVal = bsdfVal * (1.0f / bsdfPdf);
			const V3 wo(its.shS.x * woL.x + its.shT.x * woL.y + its.shN.x * woL.z,
			            its.shS.y * woL.x + its.shT.y * woL.y + its.shN.y * woL.z,
			            its.shS.z * woL.x + its.shT.z * woL.y + its.shN.z * woL.z);
			const float woDotGeoN = dot(its.geoN, wo);
			if (woDotGeoN * woL.z <= 0 && strict)
				break;
			// ray = Ray(its.p, wo, time): mint = Epsilon, maxt = inf
			row[0] = make_float4(its.p.x, its.p.y, its.p.z, kEpsilon);
			row[1] = make_float4(wo.x, wo.y, wo.z, MG_INF);
			row[5] = make_float4(bsdfVal.x, bsdfVal.y, bsdfVal.z, bsdfPdf);
			flags = (flags & 0x00FFFFFFu) | (sampledType << F_ST_SHIFT);
			continues = true;
		} while (false);
		if (!continues) { row[0] = ro; row[1] = rd; }      // a path that ends keeps its last ray (the slots held triangle data)

		flags = (flags & ~((0xFFu << F_D1_SHIFT) | (0xFFu << F_D2_SHIFT))) | (smp.d1 << F_D1_SHIFT) | (smp.d2 << F_D2_SHIFT);
		row[3] = make_float4(thr.x, thr.y, thr.z, __int_as_float(depth));
		row[4] = make_float4(Li.x, Li.y, Li.z, __uint_as_float(flags));
		reinterpret_cast<uint4 &>(row[6]) = make_uint4((uint32_t) (smp.stream & 0xFFFFFFFFull), (uint32_t) (smp.stream >> 32), misc_zw.x, misc_zw.y);
	}

}

#ifndef MG_SHADE_ALL_SLOTS
#define MG_SHADE_ALL_SLOTS 1      // whole 128-byte lines in both directions; 0 (only the slots needed: 7 read, 6 written, 3 of the
                                  // triangle) was measured at 66 ms instead of 44 ms per frame: partial lines cost a read-modify-write
#endif

#ifndef MG_SHADE_WAVES
#define MG_SHADE_WAVES 0
#endif
#if MG_SHADE_WAVES
#define MG_SHADE_BOUNDS __launch_bounds__(kShadeBlock, MG_SHADE_WAVES)
#else
#define MG_SHADE_BOUNDS __launch_bounds__(kShadeBlock)
#endif
// the workgroup's LDS: per-wave counts and the two queue offsets of the stream compaction, the staged path records
struct ShadeShared {
	uint32_t cnt[2][kShadeBlock / 64];
	uint32_t base[2];
	float4 rows[kShadeBlock / 64][64 * kRowStride];
};
// One workgroup of k_shade: the paths block * kShadeBlock .. of the material queue whose segment sizes are `prefix`
// (prefix[kBinShards] entries in kBinShards segments of bin_ids)
template <int BT, bool ROUNDS>
__device__ __forceinline__ void shade_block(const DScene &sc, const DPaths &ps, const DConfig &cfg, const DQueues &q, const uint32_t *prefix,
                                            const uint32_t *bin_ids, const uint32_t block, ShadeShared &sh) {
	uint32_t (&s_cnt)[2][kShadeBlock / 64] = sh.cnt;
	uint32_t (&s_base)[2] = sh.base;
	float4 (&s_rows)[kShadeBlock / 64][64 * kRowStride] = sh.rows;
	const uint32_t gtid = block * kShadeBlock + threadIdx.x;
	const uint32_t total = prefix[kBinShards];
	if (block * kShadeBlock >= total)
		return;                            // (uniform) a grid sized for the worst case
	const bool active = gtid < total;
	uint32_t id = 0u;
	uint4 binHit = make_uint4(0u, 0u, 0u, kNoPrim); bool haveBinHit = false;
	if (active) {
		int seg = 0;
		#pragma unroll
		for (int k = 1; k < kBinShards; ++k)
			if (gtid >= prefix[k]) seg = k;
		const size_t at = (size_t) seg * q.bin_seg_cap + (gtid - prefix[seg]);
		id = bin_ids[at];
		// the hit came with the id when the closest-hit kernel filled this bin (DQueues::bin_hits)
		if (q.bin_hits && bin_ids >= q.bins_base && bin_ids < q.bins_base + (size_t) kNumBins * q.bin_stride) {
			binHit = q.bin_hits[(size_t) (bin_ids - q.bins_base) + at]; haveBinHit = true;
		}
	}
	// ---- the path records of the wave, staged through LDS ----
	// The ids come from a material-sorted queue, so every lane owns a different 128-byte line.  Read field by field
	// that is seven 16-byte gathers per lane which each occupy the texture-address unit for 64 lines and -- the L1
	// holds 32 KB, the CU's waves hold far more lines -- mostly go to the L2 again.  Instead eight lanes fetch one
	// record together (one fully used line per request, eight records per instruction), rows of 9 float4 keep the
	// LDS accesses free of bank conflicts, and the rows are written back the same way: whole lines, coalesced.
	// All LDS traffic is private to the wave (program order suffices, no barrier).
	float4 *rows = s_rows[threadIdx.x >> 6];
	const ShadeRow row{ rows + lane_id() * kRowStride, lane_id() & 7u };
	const uint32_t sub = lane_id() & 7u, grp = lane_id() >> 3;
	const uint64_t actMask = __ballot(active);
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t src = grp + 8u * r;
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) src);
		// slot 7 (the raster position) is only read by the film kernels
		if (((actMask >> src) & 1ull) && (MG_SHADE_ALL_SLOTS || sub != 7u)) rows[shade_row_index(src, sub)] = ld_stream<4>(&ps.base[(size_t) sid * kPathSlots + sub]);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	bool continues = false, wantShadow = false;
	V3 neeV(0, 0, 0), shO(0, 0, 0), shD(0, 0, 0);      // pending direct-light term and its shadow ray
	float4 ro = make_float4(0, 0, 0, 0), rd = ro, T4 = ro, L4 = ro;
	uint4 h = make_uint4(0u, 0u, 0u, kNoPrim);
	if (active) {
		// a direct-light term the any-hit kernel parked in slot 2 (DQueues::nee_parked) is added before anything else of this
		// Li iteration, where the sequential loop adds it (path.cpp:124)
		const float4 slot2 = row[2];
		L4 = settled_Li(row[4], slot2);
		h = haveBinHit ? binHit : reinterpret_cast<const uint4 &>(slot2);
		// what the write-back below leaves in slot 2: nothing while terms are parked there, otherwise the hit
		if (q.nee_parked) row[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		else reinterpret_cast<uint4 &>(row[2]) = h;
		ro = row[0]; rd = row[1]; T4 = row[3];
		if (ROUNDS && cfg.dr_mode == 2) {
			// rounds of MIDirectIntegrator: later BSDF samples start again from the camera hit (kept in ps.prim)
			if (cfg.dr_index > 0) {
				ro = ps.prim[3 * (size_t) id]; rd = ps.prim[3 * (size_t) id + 1];
				h = reinterpret_cast<const uint4 &>(ps.prim[3 * (size_t) id + 2]);
			} else if (cfg.n_bsdf > 1) {
				ps.prim[3 * (size_t) id] = ro; ps.prim[3 * (size_t) id + 1] = rd;
				ps.prim[3 * (size_t) id + 2] = reinterpret_cast<const float4 &>(h);
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	// the first 64 bytes of the hit primitives' gather records (three position chunks + one of the normals), four lanes
	// per record, into row slots 0, 1, 3, 4 -- whose contents sit in registers now
	{
		const uint32_t prim = h.w;
		const uint64_t validMask = __ballot(active && prim != kNoPrim);
		const uint32_t sub4 = lane_id() & 3u, grp4 = lane_id() >> 2;
		const uint32_t slotOf = sub4 < 2u ? sub4 : sub4 + 1u;      // chunks 0, 1, 2, 3 -> slots 0, 1, 3, 4
		#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t src = grp4 + 16u * r;
			const uint32_t sprim = (uint32_t) __shfl((int) prim, (int) src);
			if (((validMask >> src) & 1ull) && (MG_SHADE_ALL_SLOTS || sub4 != 3u)) rows[shade_row_index(src, slotOf)] = sc.tri_pos[(size_t) sprim * kTriStride + sub4];
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	if (active)
		shade_path<BT, ROUNDS>(sc, ps, cfg, id, ro, rd, h, T4, L4, row, continues, wantShadow, neeV, shO, shD);
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t src = grp + 8u * r;
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) src);
		// the raster position (slot 7) does not change here; slot 2 does when direct-light terms are parked in it (the term this
		// shading added has to go: every later reader would add it again), otherwise it holds the hit, unchanged
		if (((actMask >> src) & 1ull) && (MG_SHADE_ALL_SLOTS || ((sub != 2u || q.nee_parked) && sub != 7u))) st_stream<4>(&ps.base[(size_t) sid * kPathSlots + sub], rows[shade_row_index(src, sub)]);
	}

	// stream compaction: survivors -> next closest-hit queue, shadow rays -> shadow queue.
	// ballot + prefix popcount inside each wave, an LDS scan across the waves, ONE atomic per workgroup and queue
	// (a queue counter is a single word: every atomic on it serialises, which is why the workgroups are as large as
	// they can be: 1024 threads, 43.6 -> 41.9 ms per 64-spp frame against 512).  Measured and rejected: both queues
	// reserved with one 64-bit atomic on a shared word (43.4 ms); the reservation issued before the records are written
	// back so that its round trip hides under those stores (45 ms: the extra barrier delays the stores of every wave)
	const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
	const unsigned long long mN = __ballot(continues), mS = __ballot(wantShadow);
	if (lane == 0) { s_cnt[0][wave] = (uint32_t) __popcll(mN); s_cnt[1][wave] = (uint32_t) __popcll(mS); }
	__syncthreads();
	if (threadIdx.x < 2) {
		uint32_t total = 0;
		for (int w = 0; w < kShadeBlock / 64; ++w) total += s_cnt[threadIdx.x][w];
		s_base[threadIdx.x] = total ? atomicAdd(&q.counters[threadIdx.x == 0 ? kNextWord : kShadowWord], total) : 0u;
	}
	__syncthreads();
	uint32_t offN = s_base[0], offS = s_base[1];
	for (uint32_t w = 0; w < wave; ++w) { offN += s_cnt[0][w]; offS += s_cnt[1][w]; }
	const unsigned long long below = (1ull << lane) - 1ull;
	if (continues) {
		const uint32_t pos = offN + (uint32_t) __popcll(mN & below);
		q.next[pos] = id;
		if (ps.rqn_o) {       // the new ray once more, in the order of the queue it was appended to
			st_stream<4>(&ps.rqn_o[pos], rows[shade_row_index(lane, 0)]);
			st_stream<4>(&ps.rqn_d[pos], rows[shade_row_index(lane, 1)]);
		}
	}
	if (wantShadow) {
		// the shadow ray lives in queue order (coalesced for both kernels); the path id rides in nee.w
		const uint32_t pos = offS + (uint32_t) __popcll(mS & below);
		st_stream<4>(&ps.shq_o[pos], make_float4(shO.x, shO.y, shO.z, 0.0f));
		st_stream<4>(&ps.shq_d[pos], make_float4(shD.x, shD.y, shD.z, 0.0f));
		st_stream<4>(&ps.shq_nee[pos], make_float4(neeV.x, neeV.y, neeV.z, __uint_as_float(id)));
	}
}

template <int BT, bool ROUNDS>
__global__ MG_SHADE_BOUNDS void k_shade(DScene sc, DPaths ps, DConfig cfg, DQueues q, BinView view_host,
                                                       const BinView *views_dev, const uint32_t *bin_ids) {
	__shared__ ShadeShared sh;
	// the bin's segment sizes: a kernel argument when the host read the counters back, otherwise what k_prep wrote
	shade_block<BT, ROUNDS>(sc, ps, cfg, q, views_dev ? views_dev[BT].prefix : view_host.prefix, bin_ids, blockIdx.x, sh);
}

// All material queues of a bounce in ONE launch (device-driven bounces): the workgroups are dealt to the bins in bin order,
// ceil(size / kShadeBlock) each, the sizes read from what k_prep left in device memory.  A frame of few paths is a chain of
// short launches, and a launch of k_shade -- 1024 threads and 148 KB of LDS per workgroup -- costs 10-20 us even when
// nearly all of its worst-case grid exits at once: one launch per bounce instead of one per BSDF type present.
__global__ MG_SHADE_BOUNDS void k_shade_all(DScene sc, DPaths ps, DConfig cfg, DQueues q, const BinView *views_dev, uint32_t bin_mask) {
	__shared__ ShadeShared sh;
	uint32_t block = blockIdx.x;
	int bin = -1;
	for (int b = 0; b < kNumBins; ++b) {
		if (!((bin_mask >> b) & 1u)) continue;
		const uint32_t nb = (views_dev[b].prefix[kBinShards] + kShadeBlock - 1u) / kShadeBlock;
		if (block < nb) { bin = b; break; }
		block -= nb;
	}
	if (bin < 0) return;
	const uint32_t *prefix = views_dev[bin].prefix, *ids = q.bin(bin);
	switch (bin) {
		case 0: shade_block<0, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 1: shade_block<1, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 2: shade_block<2, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 3: shade_block<3, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 4: shade_block<4, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 5: shade_block<5, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 6: shade_block<6, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 7: shade_block<7, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		default: shade_block<kNumBsdfTypes, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
	}
}

void launch_bsdf_eval(hipStream_t s, uint32_t type, const float *params, int op, uint32_t n, const float *queries, float *out) {
	BsdfParams p;
	for (int k = 0; k < kBsdfNParams; ++k) p.v[k] = params[k];
	if (n) hipLaunchKernelGGL(k_bsdf_eval, dim3(blocks_for(n, 256)), dim3(256), 0, s, type, p, op, n, queries, out);
}

void launch_shade(hipStream_t s, int bin, const DScene &sc, const DPaths &ps, const DConfig &cfg,
                  const DQueues &q, const BinView &view, const BinView *views_dev, uint32_t n_bound, const uint32_t *bin_ids) {
	const uint32_t n = views_dev ? n_bound : view.prefix[kBinShards];
	if (!n) return;
	if (!bin_ids) bin_ids = q.bin(bin);
	const dim3 g(blocks_for(n, kShadeBlock)), b(kShadeBlock);
	#define MG_SHADE(BT) do { if (cfg.dr_mode != 0) hipLaunchKernelGGL((k_shade<BT, true>), g, b, 0, s, sc, ps, cfg, q, view, views_dev, bin_ids); \
	                          else hipLaunchKernelGGL((k_shade<BT, false>), g, b, 0, s, sc, ps, cfg, q, view, views_dev, bin_ids); } while (0)
	switch (bin) {
		case 0: MG_SHADE(0); break;
		case 1: MG_SHADE(1); break;
		case 2: MG_SHADE(2); break;
		case 3: MG_SHADE(3); break;
		case 4: MG_SHADE(4); break;
		case 5: MG_SHADE(5); break;
		case 6: MG_SHADE(6); break;
		case 7: MG_SHADE(7); break;
		default: MG_SHADE(kNumBsdfTypes); break;
	}
	#undef MG_SHADE
}

void launch_shade_all(hipStream_t s, const DScene &sc, const DPaths &ps, const DConfig &cfg, const DQueues &q,
                      const BinView *views_dev, uint32_t bin_mask, uint32_t n_bound) {
	if (!n_bound || !bin_mask) return;
	// every bin rounds its size up to whole workgroups
	const unsigned blocks = blocks_for(n_bound, kShadeBlock) + (unsigned) __builtin_popcount(bin_mask);
	hipLaunchKernelGGL(k_shade_all, dim3(blocks), dim3(kShadeBlock), 0, s, sc, ps, cfg, q, views_dev, bin_mask);
}

} // namespace mg
