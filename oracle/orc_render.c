/*
 * orc_render.c -- oracle (TEST INFRASTRUCTURE ONLY): kd-tree traversal,
 * intersection records, luminaire sampling, BSDFs, MIPathTracer::Li,
 * SampleIntegrator::renderBlock and ImageBlock::putSample, restated in plain C.
 */
#include "orc_internal.h"
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float o[3], d[3], dRcp[3], mint, maxt; } ray_t;

typedef struct {
	float t, p[3];
	float geoS[3], geoT[3], geoN[3];
	float shS[3], shT[3], shN[3];
	float wi[3];
	uint32_t shape, prim;   /* prim = global primitive id */
} its_t;

/* Ray(o, d, time): mint = Epsilon, maxt = inf, dRcp = 1/d (include/mitsuba/core/ray.h:63-74) */
static void ray_init(ray_t *r, const float o[3], const float d[3]) {
	for (int i = 0; i < 3; ++i) { r->o[i] = o[i]; r->d[i] = d[i]; r->dRcp[i] = (float) 1.0f / d[i]; }
	r->mint = ORC_EPS; r->maxt = INFINITY;
}

/* AABB::rayIntersect (include/mitsuba/core/aabb.h:349-382) */
static int aabb_ray_intersect(const float bmin[3], const float bmax[3], const ray_t *ray, float *nearT, float *farT) {
	*nearT = -INFINITY; *farT = INFINITY;
	for (int i = 0; i < 3; i++) {
		const float direction = ray->d[i], origin = ray->o[i];
		const float minVal = bmin[i], maxVal = bmax[i];
		if (direction == 0) {
			if (origin < minVal || origin > maxVal)
				return 0;
		} else {
			float t1 = (minVal - origin) * ray->dRcp[i];
			float t2 = (maxVal - origin) * ray->dRcp[i];
			if (t1 > t2) { float tmp = t1; t1 = t2; t2 = tmp; }
			*nearT = fmaxf_(*nearT, t1);
			*farT = fminf_(*farT, t2);
			if (*nearT > *farT)
				return 0;
		}
	}
	return 1;
}

typedef struct { uint32_t node; float t; uint32_t prev; float p[3]; } kdstack_t;
#define KD_NULL 0xFFFFFFFFu
#define KD_MAXDEPTH 48
#define MAILBOX_SIZE 8   /* MTS_KD_MAILBOX_SIZE (sahkdtree3.h:28-29) */
#define MAILBOX_MASK 7

typedef struct { uint32_t shapeIndex, primIndex, prim; float u, v; } icache_t;

/* rayIntersectHavran<shadowRay> (include/mitsuba/render/sahkdtree3.h:170-300) with
 * ShapeKDTree::intersect (include/mitsuba/render/skdtree.h:244-336) inlined */
/* solveQuadratic (src/libcore/util.cpp:450-488) */
static int solve_quadratic(float a, float b, float c, float *x0, float *x1) {
	if (a == 0) {
		if (b != 0) { *x0 = *x1 = -c / b; return 1; }
		return 0;
	}
	float discrim = b*b - 4.0f*a*c;
	if (discrim < 0)
		return 0;
	float temp, sqrtDiscrim = sqrtf(discrim);
	if (b < 0)
		temp = -0.5f * (b - sqrtDiscrim);
	else
		temp = -0.5f * (b + sqrtDiscrim);
	*x0 = temp / a;
	*x1 = c / temp;
	if (*x0 > *x1) { float tmp = *x0; *x0 = *x1; *x1 = tmp; }
	return 1;
}

/* Sphere::rayIntersect(ray, mint, maxt, t, tmp) (src/shapes/sphere.cpp:94-116); P = shape parameter block */
static int sphere_ray_intersect(const float *P, const float ro[3], const float rd[3], float mint, float maxt, float *t) {
	float o[3];
	v3_sub(o, ro, P);
	float A = rd[0]*rd[0] + rd[1]*rd[1] + rd[2]*rd[2];
	float B = 2 * (rd[0]*o[0] + rd[1]*o[1] + rd[2]*o[2]);
	float C = o[0]*o[0] + o[1]*o[1] + o[2]*o[2] - P[3]*P[3];
	float nearT, farT;
	if (!solve_quadratic(A, B, C, &nearT, &farT))
		return 0;
	if (nearT > maxt || farT < mint)
		return 0;
	if (nearT < mint) {
		if (farT > maxt)
			return 0;
		*t = farT;
	} else {
		*t = nearT;
	}
	return 1;
}

/* Sphere::rayIntersect(ray, mint, maxt) (sphere.cpp:118-134) */
static int sphere_ray_intersect_shadow(const float *P, const float ro[3], const float rd[3], float mint, float maxt) {
	float o[3];
	v3_sub(o, ro, P);
	float A = rd[0]*rd[0] + rd[1]*rd[1] + rd[2]*rd[2];
	float B = 2 * (rd[0]*o[0] + rd[1]*o[1] + rd[2]*o[2]);
	float C = o[0]*o[0] + o[1]*o[1] + o[2]*o[2] - P[3]*P[3];
	float nearT, farT;
	if (!solve_quadratic(A, B, C, &nearT, &farT))
		return 0;
	if (nearT > maxt || farT < mint)
		return 0;
	if (nearT < mint && farT > maxt)
		return 0;
	return 1;
}

static int havran(const mtsgpu_scene *sc, const ray_t *ray, float mint, float maxt, float *t,
                  icache_t *cache, int shadowRay, orc_trace_counts *cnt) {
	kdstack_t stack[KD_MAXDEPTH + 2];
	uint32_t mailbox[MAILBOX_SIZE];
	memset(mailbox, 0xFF, sizeof(mailbox));
	uint32_t enPt = 0;
	stack[enPt].t = mint;
	for (int i = 0; i < 3; ++i) stack[enPt].p[i] = ray->o[i] + mint * ray->d[i];
	uint32_t exPt = 1;
	stack[exPt].t = maxt;
	for (int i = 0; i < 3; ++i) stack[exPt].p[i] = ray->o[i] + maxt * ray->d[i];
	stack[exPt].node = KD_NULL;

	int foundIntersection = 0;
	uint32_t currNode = 0;
	while (currNode != KD_NULL) {
		const uint32_t *n = sc->kd_nodes + 2 * (size_t) currNode;
		while (!(n[0] & 0x80000000u)) {
			float splitVal; memcpy(&splitVal, &n[1], 4);
			const int axis = (int) (n[0] & 0x3);
			const uint32_t left = currNode + ((n[0] & ~(0x3u + 0x40000000u)) >> 2);
			uint32_t farChild;
			if (cnt) cnt->n_inner++;
			if (stack[enPt].p[axis] <= splitVal) {
				if (stack[exPt].p[axis] <= splitVal) {
					currNode = left; n = sc->kd_nodes + 2 * (size_t) currNode;
					continue;
				}
				if (stack[enPt].p[axis] == splitVal) {
					currNode = left + 1; n = sc->kd_nodes + 2 * (size_t) currNode;
					continue;
				}
				currNode = left;
				farChild = left + 1;
			} else {
				if (splitVal < stack[exPt].p[axis]) {
					currNode = left + 1; n = sc->kd_nodes + 2 * (size_t) currNode;
					continue;
				}
				farChild = left;
				currNode = left + 1;
			}
			n = sc->kd_nodes + 2 * (size_t) currNode;
			float distToSplit = (splitVal - ray->o[axis]) * ray->dRcp[axis];
			const uint32_t tmp = exPt++;
			if (exPt == enPt)
				++exPt;
			stack[exPt].prev = tmp;
			stack[exPt].t = distToSplit;
			stack[exPt].node = farChild;
			for (int i = 0; i < 3; ++i) stack[exPt].p[i] = ray->o[i] + distToSplit * ray->d[i];
			stack[exPt].p[axis] = splitVal;
		}

		if (cnt) cnt->n_leaf++;
		for (uint32_t entry = n[0] & 0x7FFFFFFFu, last = n[1]; entry != last; entry++) {
			const uint32_t primIdx = sc->kd_indices[entry];
			if (cnt) cnt->n_idx++;
			if (mailbox[primIdx & MAILBOX_MASK] == primIdx)
				continue;
			if (cnt) cnt->n_tri_tested++;
			const uint32_t *ta = sc->triaccel + 12 * (size_t) primIdx;
			float tempU, tempV, tempT;
			int result;
			if (ta[0] == MTSGPU_KNOTRIANGLE) {
				/* non-triangle shape (skdtree.h:287-296 / :328-332) */
				const float *SP = sc->shape_params + MTSGPU_SHAPE_NPARAMS * (size_t) ta[10];
				if (!shadowRay) {
					result = sphere_ray_intersect(SP, ray->o, ray->d, mint, maxt, t);
					if (result) {
						cache->shapeIndex = ta[10]; cache->primIndex = MTSGPU_KNOTRIANGLE; cache->prim = primIdx;
						cache->u = cache->v = 0.0f;
					}
				} else {
					result = sc->shape_bsdf[ta[10]] >= 0 && sphere_ray_intersect_shadow(SP, ray->o, ray->d, mint, maxt);
				}
			} else if (!shadowRay) {
				result = orc_triaccel_intersect(ta, ray->o, ray->d, mint, maxt, &tempU, &tempV, &tempT);
				if (result) {
					*t = tempT;
					cache->shapeIndex = ta[10]; cache->primIndex = ta[11]; cache->prim = primIdx;
					cache->u = tempU; cache->v = tempV;
				}
			} else {
				/* shape->isOccluder() == has a BSDF (shape.h:324, shape.cpp:88-90) */
				result = sc->shape_bsdf[ta[10]] >= 0 &&
					orc_triaccel_intersect(ta, ray->o, ray->d, mint, maxt, &tempU, &tempV, &tempT);
			}
			if (result) {
				if (shadowRay)
					return 1;
				maxt = *t;
				foundIntersection = 1;
			}
			mailbox[primIdx & MAILBOX_MASK] = primIdx;
		}

		if (stack[exPt].t > maxt)
			break;
		enPt = exPt;
		currNode = stack[exPt].node;
		exPt = stack[enPt].prev;
	}
	return foundIntersection;
}

/* the common prologue of ShapeKDTree::rayIntersect (skdtree.cpp:108-123 / :180-192) */
static int kd_clip(const mtsgpu_scene *sc, const ray_t *ray, int closest, float *mint, float *maxt) {
	if (!aabb_ray_intersect(sc->aabb_min, sc->aabb_max, ray, mint, maxt))
		return 0;
	float rayMinT = ray->mint;
	if (rayMinT == ORC_EPS) {
		float m = fmaxf_(fmaxf_(fabsf(ray->o[0]), fabsf(ray->o[1])), fabsf(ray->o[2]));
		if (closest)
			m = fmaxf_(m, ORC_EPS);   /* only the (ray, its) variant has the inner max */
		rayMinT *= m;
	}
	if (rayMinT > *mint) *mint = rayMinT;
	if (ray->maxt < *maxt) *maxt = ray->maxt;
	return *maxt > *mint;
}

/* fillIntersectionRecord<true> (include/mitsuba/render/skdtree.h:352-432) */
/* Sphere::fillIntersectionRecord (src/shapes/sphere.cpp:136-178); uv / dpdu / dpdv only matter through the frame */
static void sphere_fill_its(const float *P, const ray_t *ray, its_t *its) {
	const float *O2W = P + 5, *W2O = P + 14;
	const float radius = P[3];
	for (int i = 0; i < 3; ++i) its->p[i] = ray->o[i] + its->t * ray->d[i];
	float pc[3], local[3];
	v3_sub(pc, its->p, P);
	for (int i = 0; i < 3; ++i) local[i] = W2O[3*i] * pc[0] + W2O[3*i+1] * pc[1] + W2O[3*i+2] * pc[2];   /* transform.h:163-170 */
	const float theta = orc_acosf(fminf_(fmaxf_(local[2] / radius, -(float) 1), (float) 1));
	/* its.dpdu = m_objectToWorld(Vector(-local.y, local.x, 0) * (2*M_PI)) */
	float du[3] = { -local[1] * (2*ORC_PI), local[0] * (2*ORC_PI), 0 * (2*ORC_PI) }, dpdu[3], dpdv[3];
	for (int i = 0; i < 3; ++i) dpdu[i] = O2W[3*i] * du[0] + O2W[3*i+1] * du[1] + O2W[3*i+2] * du[2];
	v3_normalize(its->geoN, pc);
	const float zrad = sqrtf(local[0]*local[0] + local[1]*local[1]);
	if (zrad > 0) {
		const float invZRad = 1.0f / zrad, cosPhi = local[0] * invZRad, sinPhi = local[1] * invZRad;
		float dv[3] = { (local[2] * cosPhi) * ORC_PI, (local[2] * sinPhi) * ORC_PI, (-orc_sinf(theta) * radius) * ORC_PI };
		for (int i = 0; i < 3; ++i) dpdv[i] = O2W[3*i] * dv[0] + O2W[3*i+1] * dv[1] + O2W[3*i+2] * dv[2];
		v3_normalize(its->geoS, dpdu);
		v3_normalize(its->geoT, dpdv);
	} else {
		orc_coordinate_system(its->geoN, its->geoS, its->geoT);
	}
	if (P[4] != 0.0f)
		for (int i = 0; i < 3; ++i) its->geoN[i] *= -1;
	for (int i = 0; i < 3; ++i) { its->shN[i] = its->geoN[i]; its->shS[i] = its->geoS[i]; its->shT[i] = its->geoT[i]; }
	float md[3] = { -ray->d[0], -ray->d[1], -ray->d[2] };
	its->wi[0] = v3_dot(md, its->shS); its->wi[1] = v3_dot(md, its->shT); its->wi[2] = v3_dot(md, its->shN);
}

static void fill_its(const mtsgpu_scene *sc, const ray_t *ray, const icache_t *cache, its_t *its) {
	if (cache->primIndex == MTSGPU_KNOTRIANGLE) {
		sphere_fill_its(sc->shape_params + MTSGPU_SHAPE_NPARAMS * (size_t) cache->shapeIndex, ray, its);
		its->shape = cache->shapeIndex;
		its->prim = cache->prim;
		return;
	}
	const uint32_t *tri = sc->tri_idx + 3 * (size_t) cache->prim;
	const float b[3] = { 1 - cache->u - cache->v, cache->u, cache->v };
	const float *p0 = sc->vtx_pos + 3 * (size_t) tri[0];
	const float *p1 = sc->vtx_pos + 3 * (size_t) tri[1];
	const float *p2 = sc->vtx_pos + 3 * (size_t) tri[2];
	for (int i = 0; i < 3; ++i)
		its->p[i] = p0[i] * b[0] + p1[i] * b[1] + p2[i] * b[2];
	float e1[3], e2[3], faceNormal[3];
	v3_sub(e1, p1, p0); v3_sub(e2, p2, p0);
	v3_cross(faceNormal, e1, e2);
	float length = v3_length(faceNormal);
	if (!(faceNormal[0] == 0 && faceNormal[1] == 0 && faceNormal[2] == 0))
		v3_div(faceNormal, faceNormal, length);
	for (int i = 0; i < 3; ++i) its->geoN[i] = faceNormal[i];
	orc_coordinate_system(its->geoN, its->geoS, its->geoT);
	if (sc->shape_flags[cache->shapeIndex] & MTSGPU_SHAPE_HAS_NORMALS) {
		const float *n0 = sc->vtx_nrm + 3 * (size_t) tri[0];
		const float *n1 = sc->vtx_nrm + 3 * (size_t) tri[1];
		const float *n2 = sc->vtx_nrm + 3 * (size_t) tri[2];
		float n[3];
		for (int i = 0; i < 3; ++i)
			n[i] = n0[i] * b[0] + n1[i] * b[1] + n2[i] * b[2];
		v3_normalize(its->shN, n);
		orc_coordinate_system(its->shN, its->shS, its->shT);
	} else {
		for (int i = 0; i < 3; ++i) { its->shN[i] = its->geoN[i]; its->shS[i] = its->geoS[i]; its->shT[i] = its->geoT[i]; }
	}
	float md[3] = { -ray->d[0], -ray->d[1], -ray->d[2] };
	its->wi[0] = v3_dot(md, its->shS); its->wi[1] = v3_dot(md, its->shT); its->wi[2] = v3_dot(md, its->shN);
	its->shape = cache->shapeIndex;
	its->prim = cache->prim;
}

/* ShapeKDTree::rayIntersect(ray, its) (skdtree.cpp:108-132) */
static int scene_ray_intersect(const mtsgpu_scene *sc, const ray_t *ray, its_t *its, mtsgpu_stats *st) {
	icache_t cache;
	float mint, maxt;
	its->t = INFINITY;
	if (st) st->rays_closest++;
	if (kd_clip(sc, ray, 1, &mint, &maxt)) {
		if (havran(sc, ray, mint, maxt, &its->t, &cache, 0, NULL)) {
			fill_its(sc, ray, &cache, its);
			return 1;
		}
	}
	return 0;
}

/* Scene::isOccluded (include/mitsuba/render/scene.h:241-246) -> ShapeKDTree::rayIntersect(ray) (skdtree.cpp:180-199) */
static int scene_is_occluded(const mtsgpu_scene *sc, const float p1[3], const float p2[3], mtsgpu_stats *st) {
	ray_t ray; float d[3], mint, maxt, t = INFINITY;
	v3_sub(d, p2, p1);
	ray_init(&ray, p1, d);
	ray.mint = ORC_SHADOW_EPS;
	ray.maxt = 1 - ORC_SHADOW_EPS;
	if (st) st->rays_shadow++;
	if (kd_clip(sc, &ray, 0, &mint, &maxt))
		if (havran(sc, &ray, mint, maxt, &t, NULL, 1, NULL))
			return 1;
	return 0;
}

/* The ray generator of the reference's own traversal benchmark (src/tests/test_kd.cpp:96-116): chords between two
 * uniform points of a sphere, drawn from a default-seeded Random.  rays: [n][8] = o, mint, d, maxt with the
 * defaults of Ray(o, d, time) (mint = Epsilon, maxt = inf).  Point2(nextFloat(), nextFloat()) leaves the order of
 * its two draws to the compiler; x first is used. */
void orc_chord_rays(const float center[3], float radius, uint32_t n, float *rays) {
	orc_random rnd;
	orc_random_seed(&rnd, 5489ULL);                     /* Random::Random() (random.cpp:61-76) */
	for (uint32_t i = 0; i < n; ++i) {
		float s1[2], s2[2], d1[3], d2[3], p1[3], p2[3], dir[3];
		s1[0] = orc_random_next_float(&rnd); s1[1] = orc_random_next_float(&rnd);
		s2[0] = orc_random_next_float(&rnd); s2[1] = orc_random_next_float(&rnd);
		orc_square_to_sphere(s1, d1); orc_square_to_sphere(s2, d2);
		for (int k = 0; k < 3; ++k) { p1[k] = center[k] + d1[k] * radius; p2[k] = center[k] + d2[k] * radius; }
		v3_sub(dir, p2, p1);
		v3_normalize(dir, dir);
		float *r = rays + 8 * (size_t) i;
		r[0] = p1[0]; r[1] = p1[1]; r[2] = p1[2]; r[3] = ORC_EPS;
		r[4] = dir[0]; r[5] = dir[1]; r[6] = dir[2]; r[7] = INFINITY;
	}
}

void orc_trace_rays(const mtsgpu_scene *sc, const float *rays, uint32_t n, int shadow,
                    uint32_t *hits, orc_trace_counts *counts) {
	if (counts) memset(counts, 0, sizeof(*counts));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4096) if (!counts && n > 100000)
#endif
	for (uint32_t i = 0; i < n; ++i) {
		const float *r = rays + 8 * (size_t) i;
		ray_t ray;
		ray_init(&ray, r, r + 4);
		ray.mint = r[3]; ray.maxt = r[7];
		uint32_t *h = hits + 4 * (size_t) i;
		float mint, maxt, t = INFINITY;
		icache_t cache; memset(&cache, 0, sizeof(cache));
		int found = 0;
		if (kd_clip(sc, &ray, !shadow, &mint, &maxt))
			found = havran(sc, &ray, mint, maxt, &t, &cache, shadow, counts);
		if (shadow) {
			h[0] = h[1] = h[2] = 0; h[3] = found ? 1u : 0u;
		} else if (found) {
			memcpy(&h[0], &t, 4); memcpy(&h[1], &cache.u, 4); memcpy(&h[2], &cache.v, 4); h[3] = cache.prim;
		} else {
			float inf = INFINITY;
			memcpy(&h[0], &inf, 4); h[1] = h[2] = 0; h[3] = 0xFFFFFFFFu;
		}
	}
}

/* ========================================================================== */
/* DiscretePDF sampling (include/mitsuba/core/pdf.h:102-133)                  */
/* ========================================================================== */
static int dpdf_sample(const float *cdf, uint32_t n, float sampleValue) {
	/* std::lower_bound over n+1 entries: first element >= sampleValue */
	uint32_t lo = 0, count = n + 1;
	while (count > 0) {
		uint32_t step = count / 2, it = lo + step;
		if (cdf[it] < sampleValue) { lo = it + 1; count -= step + 1; }
		else count = step;
	}
	int index = (int) lo - 1;
	if (index < 0) index = 0;
	if (index > (int) n - 1) index = (int) n - 1;
	return index;
}
static int dpdf_sample_reuse(const float *cdf, uint32_t n, float *sampleValue) {
	int index = dpdf_sample(cdf, n, *sampleValue);
	*sampleValue = (*sampleValue - cdf[index]) / (cdf[index + 1] - cdf[index]);
	return index;
}

/* ========================================================================== */
/* Luminaires                                                                 */
/* ========================================================================== */
typedef struct { float p[3], n[3], d[3], pdf, value[3]; int lum; } lrec_t;

/* BSphere::rayIntersect (include/mitsuba/core/bsphere.h:85-118) */
static int bsphere_ray_intersect(const float center[3], float radius, const float o[3], const float d[3],
                                 float *nearHit, float *farHit) {
	float originToCenter[3];
	v3_sub(originToCenter, center, o);
	float distToRayClosest = v3_dot(originToCenter, d);
	float tmp1 = v3_dot(originToCenter, originToCenter) - radius*radius;
	if (tmp1 <= 0.0f) {
		*nearHit = *farHit = sqrtf(distToRayClosest * distToRayClosest - tmp1) + distToRayClosest;
		return 1;
	}
	if (distToRayClosest < 0.0f)
		return 0;
	float sqrOriginToCenterLength = v3_dot(originToCenter, originToCenter);
	float sqrHalfChordDist = radius * radius - sqrOriginToCenterLength + distToRayClosest * distToRayClosest;
	if (sqrHalfChordDist < 0)
		return 0;
	float hitDistance = sqrtf(sqrHalfChordDist);
	*nearHit = distToRayClosest - hitDistance;
	*farHit = distToRayClosest + hitDistance;
	if (*nearHit == 0)
		*nearHit = *farHit;
	return 1;
}

/* ---- EnvMapLuminaire (src/luminaires/envmap.cpp) ---- */
/* MIPMap::triangle(0, x, y) with EEWA/ERepeat (mipmap.cpp:226-243, getTexel :203-224) */
static void env_triangle(const mtsgpu_scene *sc, float x, float y, float out[3]) {
	const int W = (int) sc->env_width, H = (int) sc->env_height;
	x = x * W - 0.5f;
	y = y * H - 0.5f;
	const int xPos = (int) floorf(x), yPos = (int) floorf(y);
	const float dx = x - xPos, dy = y - yPos;
	const int ox[4] = { 0, 0, 1, 1 }, oy[4] = { 0, 1, 0, 1 };
	float acc[3] = { 0, 0, 0 };
	for (int k = 0; k < 4; ++k) {
		int tx = xPos + ox[k], ty = yPos + oy[k];
		if (tx <= 0 || ty < 0 || tx >= W || ty >= H) {
			int r = tx - (int) (tx / W) * W; tx = (r < 0) ? r + W : r;         /* modulo (util.cpp:424-427) */
			r = ty - (int) (ty / H) * H; ty = (r < 0) ? r + H : r;
		}
		const float *t = sc->env_pixels + 3 * ((size_t) tx + (size_t) W * ty);
		/* getTexel(..) * a * b: the Spectrum is scaled by the two factors one after the other */
		const float a = (k < 2) ? (1.0f - dx) : dx, b = (k & 1) ? dy : (1.0f - dy);
		if (k == 0) for (int c = 0; c < 3; ++c) acc[c] = t[c] * a * b;
		else for (int c = 0; c < 3; ++c) acc[c] = acc[c] + t[c] * a * b;
	}
	out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
}

/* Le(direction) (envmap.cpp:147-153) */
static void env_le(const mtsgpu_scene *sc, const float *P, const float direction[3], float out[3]) {
	const float *M = P + 7;
	const float d[3] = { M[0]*direction[0] + M[1]*direction[1] + M[2]*direction[2],
	                     M[3]*direction[0] + M[4]*direction[1] + M[5]*direction[2],
	                     M[6]*direction[0] + M[7]*direction[1] + M[8]*direction[2] };
	const float u = .5f * (1 + orc_atan2f(d[0], -d[2]) / ORC_PI);
	const float v = orc_acosf(fmaxf_((float) -1.0f, fminf_((float) 1.0f, d[1]))) / ORC_PI;
	env_triangle(sc, u, v, out);
	for (int c = 0; c < 3; ++c) out[c] *= P[0];
}

/* Le(ray) = Le(normalize(ray.d)) (envmap.cpp:155-157) */
static void env_le_ray(const mtsgpu_scene *sc, const float *P, const float rd[3], float out[3]) {
	float n[3];
	v3_normalize(n, rd);
	env_le(sc, P, n, out);
}

/* pdf(p, lRec, delta) (envmap.cpp:176-193) */
static float env_pdf(const mtsgpu_scene *sc, const float *P, const float lrec_d[3]) {
	const float *M = P + 7;
	const float nd[3] = { -lrec_d[0], -lrec_d[1], -lrec_d[2] };
	const float d[3] = { M[0]*nd[0] + M[1]*nd[1] + M[2]*nd[2], M[3]*nd[0] + M[4]*nd[1] + M[5]*nd[2], M[6]*nd[0] + M[7]*nd[1] + M[8]*nd[2] };
	const int rx = (int) sc->env_pdf_width, ry = (int) sc->env_pdf_height;
	const float x = .5f * (1 + orc_atan2f(d[0], -d[2]) / ORC_PI) * rx;
	const float y = orc_acosf(fmaxf_((float) -1.0f, fminf_((float) 1.0f, d[1]))) / ORC_PI * ry;
	int xPos = (int) floorf(x); if (xPos < 0) xPos = 0; if (xPos > rx - 1) xPos = rx - 1;
	int yPos = (int) floorf(y); if (yPos < 0) yPos = 0; if (yPos > ry - 1) yPos = ry - 1;
	const float pdf = sc->env_pdf[xPos + yPos * rx];
	const float sinTheta = sqrtf(fmaxf_((float) ORC_EPS, 1 - d[1]*d[1]));
	const float psx = 2 * ORC_PI / rx, psy = ORC_PI / ry;             /* m_pdfPixelSize (envmap.cpp:108) */
	return pdf / (psx * psy * sinTheta);
}

/* Luminaire::sample for the two plugins (src/luminaires/area.cpp:68-79, constant.cpp:73-87) */
static void luminaire_sample(const mtsgpu_scene *sc, int l, const float p[3], lrec_t *lRec, const float sample[2]) {
	const float *P = sc->lum_params + MTSGPU_LUM_NPARAMS * (size_t) l;
	if (sc->lum_type[l] == MTSGPU_LUM_AREA && sc->shape_type && sc->shape_type[sc->lum_shape[l]] == MTSGPU_SHAPE_SPHERE) {
		/* Sphere::sampleSolidAngle (src/shapes/sphere.cpp:196-237), then AreaLuminaire::sample (area.cpp:68-79) */
		const float *SP = sc->shape_params + MTSGPU_SHAPE_NPARAMS * (size_t) sc->lum_shape[l];
		const float radius = SP[3];
		float w[3];
		v3_sub(w, SP, p);
		const float invDistW = 1 / v3_length(w);
		const float squareTerm = fabsf(radius * invDistW);
		if (squareTerm >= 1 - ORC_EPS) {
			/* inside the sphere: uniform sampling */
			float d[3];
			orc_square_to_sphere(sample, d);
			for (int i = 0; i < 3; ++i) { lRec->p[i] = SP[i] + d[i] * radius; lRec->n[i] = d[i]; }
			float lumToPoint[3];
			v3_sub(lumToPoint, p, lRec->p);
			float distSquared = v3_dot(lumToPoint, lumToPoint), dp = v3_dot(lumToPoint, lRec->n);
			if (dp > 0)
				lRec->pdf = SP[23] * distSquared * sqrtf(distSquared) / dp;
			else
				lRec->pdf = 0;
		} else {
			const float cosThetaMax = sqrtf(fmaxf_((float) 0, 1 - squareTerm*squareTerm));
			/* squareToCone (util.cpp:656-662) */
			const float cosTheta = (1 - sample[0]) + sample[0] * cosThetaMax;
			const float sinTheta = sqrtf(1 - cosTheta * cosTheta);
			const float phi = sample[1] * (2 * ORC_PI);
			const float cone[3] = { orc_cosf(phi) * sinTheta, orc_sinf(phi) * sinTheta, cosTheta };
			/* Frame(w*invDistW).toWorld(cone) (frame.h:44-60) */
			float fn[3], fs[3], ft[3], d[3];
			v3_scale(fn, w, invDistW);
			orc_coordinate_system(fn, fs, ft);
			for (int i = 0; i < 3; ++i) d[i] = fs[i] * cone[0] + ft[i] * cone[1] + fn[i] * cone[2];
			float t;
			if (!sphere_ray_intersect(SP, p, d, 0, INFINITY, &t)) {
				lRec->pdf = 0;       /* roundoff: no sample */
			} else {
				for (int i = 0; i < 3; ++i) lRec->p[i] = p[i] + t * d[i];
				float pc[3];
				v3_sub(pc, lRec->p, SP);
				v3_normalize(lRec->n, pc);
				lRec->pdf = 1 / ((2 * ORC_PI) * (1 - cosThetaMax));
			}
		}
		v3_sub(lRec->d, p, lRec->p);
		if (lRec->pdf > 0 && v3_dot(lRec->d, lRec->n) > 0) {
			lRec->value[0] = P[0]; lRec->value[1] = P[1]; lRec->value[2] = P[2];
			v3_normalize(lRec->d, lRec->d);
		} else {
			lRec->pdf = 0;
		}
	} else if (sc->lum_type[l] == MTSGPU_LUM_AREA) {
		/* Shape::sampleSolidAngle (shape.cpp:65-75) -> TriMesh::sampleArea (trimesh.cpp:297-302) */
		const uint32_t s = (uint32_t) sc->lum_shape[l];
		const uint32_t t0 = sc->shape_tri_offset[s], nT = sc->shape_tri_offset[s+1] - t0;
		float newSeed[2] = { sample[0], sample[1] };
		int index = dpdf_sample_reuse(sc->lum_tri_cdf + sc->lum_cdf_offset[l], nT, &newSeed[1]);
		/* Triangle::sample (src/libcore/triangle.cpp:23-47) */
		const uint32_t *tri = sc->tri_idx + 3 * ((size_t) t0 + (uint32_t) index);
		const float *p0 = sc->vtx_pos + 3 * (size_t) tri[0];
		const float *p1 = sc->vtx_pos + 3 * (size_t) tri[1];
		const float *p2 = sc->vtx_pos + 3 * (size_t) tri[2];
		float bary[2], sideA[3], sideB[3];
		orc_square_to_triangle(newSeed, bary);
		v3_sub(sideA, p1, p0); v3_sub(sideB, p2, p0);
		for (int i = 0; i < 3; ++i)
			lRec->p[i] = p0[i] + (sideA[i] * bary[0]) + (sideB[i] * bary[1]);
		if (sc->shape_flags[s] & MTSGPU_SHAPE_HAS_NORMALS) {
			const float *n0 = sc->vtx_nrm + 3 * (size_t) tri[0];
			const float *n1 = sc->vtx_nrm + 3 * (size_t) tri[1];
			const float *n2 = sc->vtx_nrm + 3 * (size_t) tri[2];
			float n[3];
			for (int i = 0; i < 3; ++i)
				n[i] = n0[i] * (1.0f - bary[0] - bary[1]) + n1[i] * bary[0] + n2[i] * bary[1];
			v3_normalize(lRec->n, n);
		} else {
			float n[3];
			v3_cross(n, sideA, sideB);
			v3_normalize(lRec->n, n);
		}
		float pdfArea = sc->lum_inv_area[l];
		float lumToPoint[3];
		v3_sub(lumToPoint, p, lRec->p);
		float distSquared = v3_dot(lumToPoint, lumToPoint), dp = v3_dot(lumToPoint, lRec->n);
		if (dp > 0)
			lRec->pdf = pdfArea * distSquared * sqrtf(distSquared) / dp;
		else
			lRec->pdf = 0.0f;
		/* AreaLuminaire::sample (area.cpp:68-79) */
		v3_sub(lRec->d, p, lRec->p);
		if (lRec->pdf > 0 && v3_dot(lRec->d, lRec->n) > 0) {
			lRec->value[0] = P[0]; lRec->value[1] = P[1]; lRec->value[2] = P[2];
			v3_normalize(lRec->d, lRec->d);
		} else {
			lRec->pdf = 0;
		}
	} else if (sc->lum_type[l] == MTSGPU_LUM_POINT || sc->lum_type[l] == MTSGPU_LUM_SPOT) {
		/* PointLuminaire::sample (point.cpp:55-63) / SpotLuminaire::sample (spot.cpp:110-118) */
		float lumToP[3];
		v3_sub(lumToP, p, P + 3);
		float invDist = 1.0f / v3_length(lumToP);
		lRec->p[0] = P[3]; lRec->p[1] = P[4]; lRec->p[2] = P[5];
		v3_scale(lRec->d, lumToP, invDist);
		lRec->n[0] = lRec->n[1] = lRec->n[2] = 0.0f;
		lRec->pdf = 1.0f;
		float result[3] = { P[0], P[1], P[2] };
		if (sc->lum_type[l] == MTSGPU_LUM_SPOT) {
			/* falloffCurve (spot.cpp:84-103), constant texture */
			const float *M = P + 10;
			const float *d = lRec->d;
			const float cosTheta = M[6] * d[0] + M[7] * d[1] + M[8] * d[2];     /* m_worldToLuminaire(d).z */
			if (cosTheta <= P[7]) {
				result[0] = result[1] = result[2] = 0.0f;
			} else if (!(cosTheta >= P[6])) {
				float f = (P[8] - orc_acosf(cosTheta)) * P[9];
				result[0] *= f; result[1] *= f; result[2] *= f;
			}
		}
		float i2 = invDist*invDist;
		lRec->value[0] = result[0] * i2; lRec->value[1] = result[1] * i2; lRec->value[2] = result[2] * i2;
	} else if (sc->lum_type[l] == MTSGPU_LUM_ENVMAP) {
		/* EnvMapLuminaire::sampleDirection + sample (envmap.cpp:123-145, :159-172) */
		const int rx = (int) sc->env_pdf_width, ry = (int) sc->env_pdf_height;
		float sx = sample[0], pdf;
		const int idx = dpdf_sample_reuse(sc->env_cdf, (uint32_t) (rx * ry), &sx);
		pdf = sc->env_pdf[idx];
		const int row = idx / rx, col = idx - rx * row;
		const float x = col + sx, y = row + sample[1];
		float value[3];
		env_triangle(sc, x * (1.0f / rx), y * (1.0f / ry), value);
		for (int c = 0; c < 3; ++c) value[c] *= P[0];
		const float psx = 2 * ORC_PI / rx, psy = ORC_PI / ry;
		const float theta = psy * y, phi = psx * x - ORC_PI;
		const float sinTheta = orc_sinf(theta), cosTheta = orc_cosf(theta);
		const float sinPhi = orc_sinf(phi), cosPhi = orc_cosf(phi);
		pdf = pdf / (psx * psy * sinTheta);
		const float *L2W = P + 16;
		const float v[3] = { -sinTheta * sinPhi, -cosTheta, sinTheta*cosPhi };
		float d[3];
		for (int i = 0; i < 3; ++i) d[i] = L2W[3*i] * v[0] + L2W[3*i+1] * v[1] + L2W[3*i+2] * v[2];
		lRec->pdf = pdf;
		lRec->value[0] = value[0]; lRec->value[1] = value[1]; lRec->value[2] = value[2];
		const float *center = P + 3; const float radius = P[6];
		float dv[3], md[3] = { -d[0], -d[1], -d[2] }, nearHit, farHit;
		v3_sub(dv, p, center);
		if (v3_length(dv) <= radius && bsphere_ray_intersect(center, radius, p, md, &nearHit, &farHit)) {
			for (int i = 0; i < 3; ++i) lRec->p[i] = p[i] - d[i] * nearHit;
			float cn[3];
			v3_sub(cn, center, lRec->p);
			v3_normalize(lRec->n, cn);
			lRec->d[0] = d[0]; lRec->d[1] = d[1]; lRec->d[2] = d[2];
		} else {
			lRec->pdf = 0.0f;
		}
	} else if (sc->lum_type[l] == MTSGPU_LUM_COLLIMATED) {
		/* CollimatedBeamLuminaire::sample (src/luminaires/collimated.cpp:62-76) */
		const float *W = P + 4, *L = P + 16;
		const float local[3] = { W[0]*p[0] + W[1]*p[1] + W[2]*p[2] + W[3], W[4]*p[0] + W[5]*p[1] + W[6]*p[2] + W[7],
		                         W[8]*p[0] + W[9]*p[1] + W[10]*p[2] + W[11] };
		if (sqrtf(local[0]*local[0] + local[1]*local[1]) > P[3] || local[2] < 0) {
			lRec->pdf = 0.0f;
		} else {
			for (int i = 0; i < 3; ++i) lRec->p[i] = L[4*i] * local[0] + L[4*i+1] * local[1] + L[4*i+2] * 0.0f + L[4*i+3];
			/* m_direction = m_luminaireToWorld(Vector(0, 0, 1)) (collimated.cpp:36) */
			for (int i = 0; i < 3; ++i) lRec->d[i] = L[4*i] * 0.0f + L[4*i+1] * 0.0f + L[4*i+2] * 1.0f;
			lRec->n[0] = lRec->n[1] = lRec->n[2] = 0.0f;
			lRec->pdf = 1.0f;
			lRec->value[0] = P[0]; lRec->value[1] = P[1]; lRec->value[2] = P[2];
		}
	} else if (sc->lum_type[l] == MTSGPU_LUM_DIRECTIONAL) {
		/* DirectionalLuminaire::sample (directional.cpp:84-91) */
		const float k = 2 * P[6];
		lRec->p[0] = p[0] - P[3] * k; lRec->p[1] = p[1] - P[4] * k; lRec->p[2] = p[2] - P[5] * k;
		lRec->d[0] = P[3]; lRec->d[1] = P[4]; lRec->d[2] = P[5];
		lRec->n[0] = lRec->n[1] = lRec->n[2] = 0.0f;
		lRec->pdf = 1.0f;
		lRec->value[0] = P[0]; lRec->value[1] = P[1]; lRec->value[2] = P[2];
	} else {
		/* ConstantLuminaire::sample (constant.cpp:73-87) */
		float d[3], nearHit, farHit, dv[3];
		orc_square_to_sphere(sample, d);
		const float *center = P + 3; const float radius = P[6];
		v3_sub(dv, p, center);
		if (v3_length(dv) <= radius && bsphere_ray_intersect(center, radius, p, d, &nearHit, &farHit)) {
			for (int i = 0; i < 3; ++i) lRec->p[i] = p[i] + d[i] * nearHit;
			lRec->pdf = 1.0f / (4*ORC_PI);
			float cn[3];
			v3_sub(cn, center, lRec->p);
			v3_normalize(lRec->n, cn);
			lRec->d[0] = -d[0]; lRec->d[1] = -d[1]; lRec->d[2] = -d[2];
			lRec->value[0] = P[0]; lRec->value[1] = P[1]; lRec->value[2] = P[2];
		} else {
			lRec->pdf = 0.0f;
		}
	}
}

/* Scene::sampleLuminaire (src/librender/scene.cpp:396-415) */
static int scene_sample_luminaire(const mtsgpu_scene *sc, const float p[3], lrec_t *lRec, const float s[2], mtsgpu_stats *st) {
	float sample[2] = { s[0], s[1] };
	int index = dpdf_sample_reuse(sc->lum_sel_cdf, sc->n_lums, &sample[0]);
	float lumPdf = sc->lum_sel_pdf[index];
	luminaire_sample(sc, index, p, lRec, sample);
	if (lRec->pdf != 0) {
		if (scene_is_occluded(sc, p, lRec->p, st))
			return 0;
		lRec->pdf *= lumPdf;
		float recip = 1.0f / lRec->pdf;
		lRec->value[0] *= recip; lRec->value[1] *= recip; lRec->value[2] *= recip;
		lRec->lum = index;
		return 1;
	}
	return 0;
}

/* Scene::pdfLuminaire (scene.cpp:381-394) with Luminaire::pdf (area.cpp:81-83 ->
 * Shape::pdfSolidAngle shape.cpp:77-83; constant.cpp:89-91) */
static float scene_pdf_luminaire(const mtsgpu_scene *sc, const float p[3], const lrec_t *lRec) {
	const float luminance = 1.0f;   /* getSamplingWeight(), luminaire.cpp:33 */
	const float fraction = luminance / sc->lum_sel_sum;
	float pdf;
	if (sc->lum_type[lRec->lum] == MTSGPU_LUM_AREA && sc->shape_type && sc->shape_type[sc->lum_shape[lRec->lum]] == MTSGPU_SHAPE_SPHERE) {
		/* Sphere::pdfSolidAngle (sphere.cpp:239-255) */
		const float *SP = sc->shape_params + MTSGPU_SHAPE_NPARAMS * (size_t) sc->lum_shape[lRec->lum];
		float w[3];
		v3_sub(w, p, SP);
		const float invDistW = 1 / v3_length(w);
		const float squareTerm = fabsf(SP[3] * invDistW);
		if (squareTerm >= 1 - ORC_EPS) {
			float lumToPoint[3];
			v3_sub(lumToPoint, p, lRec->p);
			float distSquared = v3_dot(lumToPoint, lumToPoint), dp = v3_dot(lumToPoint, lRec->n);
			if (dp > 0)
				pdf = SP[23] * distSquared * sqrtf(distSquared) / dp;
			else
				pdf = 0;
		} else {
			const float cosThetaMax = sqrtf(fmaxf_((float) 0, 1 - squareTerm*squareTerm));
			pdf = 1 / (2 * ORC_PI * (1 - cosThetaMax));       /* squareToConePdf (util.cpp:652-654) */
		}
	} else if (sc->lum_type[lRec->lum] == MTSGPU_LUM_AREA) {
		float lumToPoint[3];
		v3_sub(lumToPoint, p, lRec->p);
		float distSquared = v3_dot(lumToPoint, lumToPoint);
		float invDP = fmaxf_((float) 0, sqrtf(distSquared) / v3_dot(lumToPoint, lRec->n));
		pdf = sc->lum_inv_area[lRec->lum] * distSquared * invDP;
	} else if (sc->lum_type[lRec->lum] == MTSGPU_LUM_ENVMAP) {
		pdf = env_pdf(sc, sc->lum_params + MTSGPU_LUM_NPARAMS * (size_t) lRec->lum, lRec->d);
	} else {
		pdf = 1.0f / (4*ORC_PI);
	}
	return pdf * fraction;
}

/* test hooks: Luminaire::sample (no visibility test) and Scene::pdfLuminaire for one luminaire.
 * out[13] = p, n, d, value, pdf */
void orc_luminaire_sample(const mtsgpu_scene *sc, int l, const float p[3], const float sample[2], float out[13]) {
	lrec_t r; memset(&r, 0, sizeof(r));
	luminaire_sample(sc, l, p, &r, sample);
	for (int i = 0; i < 3; ++i) { out[i] = r.p[i]; out[3+i] = r.n[i]; out[6+i] = r.d[i]; out[9+i] = r.value[i]; }
	out[12] = r.pdf;
}
float orc_luminaire_pdf(const mtsgpu_scene *sc, int l, const float p[3], const float lp[3], const float ln[3], const float ld[3]) {
	lrec_t r; memset(&r, 0, sizeof(r));
	for (int i = 0; i < 3; ++i) { r.p[i] = lp[i]; r.n[i] = ln[i]; r.d[i] = ld[i]; }
	r.lum = l;
	return scene_pdf_luminaire(sc, p, &r);
}

/* AreaLuminaire::Le (area.cpp:62-66) */
static void area_le(const mtsgpu_scene *sc, int l, const float n[3], const float d[3], float out[3]) {
	const float *P = sc->lum_params + MTSGPU_LUM_NPARAMS * (size_t) l;
	if (v3_dot(d, n) <= 0) { out[0] = out[1] = out[2] = 0.0f; return; }
	out[0] = P[0]; out[1] = P[1]; out[2] = P[2];
}

/* ========================================================================== */
/* BSDFs (local shading frame)                                                */
/* ========================================================================== */
enum { T_DIFFUSE_REFL = 0x1, T_DIFFUSE_TRANS = 0x2, T_DELTA_REFL = 0x4, T_DELTA_TRANS = 0x8, T_GLOSSY_REFL = 0x10, T_GLOSSY_TRANS = 0x20,
       T_DELTA = 0x4 | 0x8, T_TRANSMISSION = 0x2 | 0x8 | 0x20 };

static inline int spec_is_zero(const float s[3]) { return !(s[0] != 0.0f) && !(s[1] != 0.0f) && !(s[2] != 0.0f); }

/* Frame::tanTheta (include/mitsuba/core/frame.h:98-103) */
static inline float frame_tan_theta(const float v[3]) {
	float temp = 1 - v[2]*v[2];
	if (temp <= 0.0f)
		return 0.0f;
	return sqrtf(temp) / v[2];
}

/* beckmannD (roughmetal.cpp:75-79 == microfacet.cpp:95-99) */
static float beckmann_d(float alphaB, const float m[3]) {
	float ex = frame_tan_theta(m) / alphaB;
	return orc_expf(-(ex*ex)) / (ORC_PI * alphaB*alphaB * orc_pow4f(m[2]));
}

/* sampleBeckmannD (roughmetal.cpp:85-90) + sphericalDirection (util.cpp:543-550) */
static void sample_beckmann_d(float alphaB, const float sample[2], float m[3]) {
	float thetaM = orc_atanf(sqrtf(-alphaB*alphaB * orc_logf(1.0f - sample[0])));
	float phiM = (2.0f * ORC_PI) * sample[1];
	float sinTheta = orc_sinf(thetaM);
	m[0] = sinTheta * orc_cosf(phiM);
	m[1] = sinTheta * orc_sinf(phiM);
	m[2] = orc_cosf(thetaM);
}

/* smithBeckmannG1 (roughmetal.cpp:97-113) */
static float smith_beckmann_g1(float alphaB, const float v[3], const float m[3]) {
	if (v3_dot(v, m) * v[2] <= 0)
		return 0.0;
	const float tanTheta = frame_tan_theta(v);
	if (tanTheta == 0.0f)
		return 1.0f;
	const float a = 1.0f / (alphaB * tanTheta);
	const float aSqr = a * a;
	if (a >= 1.6f)
		return 1.0f;
	return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
}

/* reflect(wi, n) = n*(2*dot(n, wi)) - wi  (roughmetal.cpp:115-117) */
static void mf_reflect(const float wi[3], const float n[3], float wo[3]) {
	float s = 2.0f * v3_dot(n, wi);
	wo[0] = n[0]*s - wi[0]; wo[1] = n[1]*s - wi[1]; wo[2] = n[2]*s - wi[2];
}

/* ---- RoughGlass (src/bsdfs/roughglass.cpp) ----
 * P: [0] distribution [1] alpha [2] intIOR [3] extIOR [4..6] specularReflectance [7..9] specularTransmittance.
 * path.cpp leaves bRec.sampler NULL, component -1, typeMask all, quantity ERadiance.  The 3-argument
 * sample() of the plugin takes its pdf BY VALUE (roughglass.cpp:619), so it does not override the virtual and
 * BSDF::sample(bRec, pdf&, sample) (bsdf.cpp:37-48) runs: sample(bRec, s), then pdf(bRec) and f(bRec). */
#define ORC_INV_TWOPI 0.15915494309189533577f
static inline float rg_signum(float value) { return (value < 0) ? -1.0f : 1.0f; }

/* evalD (roughglass.cpp:213-257) */
static float rg_eval_d(int distr, const float m[3], float alpha) {
	if (m[2] <= 0)
		return 0.0f;
	float result;
	if (distr == 0) {
		const float ex = frame_tan_theta(m) / alpha;
		result = orc_expf(-(ex*ex)) / (ORC_PI * alpha*alpha * orc_pow4f(m[2]));
	} else if (distr == 1) {
		result = (alpha + 2) * ORC_INV_TWOPI * orc_powf(m[2], alpha);
	} else {
		const float tanTheta = frame_tan_theta(m), cosTheta = m[2];
		const float root = alpha / (cosTheta*cosTheta * (alpha*alpha + tanTheta*tanTheta));
		result = ORC_INV_PI * (root * root);
	}
	if (result < 1e-40)
		result = 0;
	return result;
}

/* sampleD (roughglass.cpp:266-293) + sphericalDirection (util.cpp:543-550) */
static void rg_sample_d(int distr, const float sample[2], float alpha, float m[3]) {
	float phiM = (2.0f * ORC_PI) * sample[1], thetaM = 0.0f;
	if (distr == 0)
		thetaM = orc_atanf(sqrtf(-alpha*alpha * orc_logf(1.0f - sample[0])));
	else if (distr == 1)
		thetaM = orc_acosf(orc_powf(sample[0], (float) 1 / (alpha + 2)));
	else
		thetaM = orc_atanf(alpha * sqrtf(sample[0]) / sqrtf(1.0f - sample[0]));
	float sinTheta = orc_sinf(thetaM);
	m[0] = sinTheta * orc_cosf(phiM);
	m[1] = sinTheta * orc_sinf(phiM);
	m[2] = orc_cosf(thetaM);
}

/* smithG1 (roughglass.cpp:303-343) */
static float rg_smith_g1(int distr, const float v[3], const float m[3], float alpha) {
	const float tanTheta = fabsf(frame_tan_theta(v));
	if (tanTheta == 0.0f)
		return 1.0f;
	if (v3_dot(v, m) * v[2] <= 0)
		return 0.0f;
	if (distr == 2) {
		const float root = alpha * tanTheta;
		return 2.0f / (1.0f + sqrtf(1.0f + root*root));
	}
	if (distr == 1)
		alpha = sqrtf(0.5f * alpha + 1) / tanTheta;      /* falls through to the Beckmann case */
	const float a = 1.0f / (alpha * tanTheta);
	const float aSqr = a * a;
	if (a >= 1.6f)
		return 1.0f;
	return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
}

/* the half-vector of f() and pdf() (roughglass.cpp:355-377 == :417-446) */
static void rg_half_vector(const float *P, const float wi[3], const float wo[3], int reflect, float etaI, float etaT, float H[3]) {
	if (reflect) {
		float h[3], n[3];
		v3_add(h, wo, wi); v3_normalize(n, h);
		v3_scale(H, n, rg_signum(wo[2]));
	} else {
		float h[3], n[3];
		for (int i = 0; i < 3; ++i) h[i] = wi[i]*etaI + wo[i]*etaT;
		v3_normalize(n, h);
		const float sgn = (P[3] > P[2] ? (float) 1 : (float) -1);
		for (int i = 0; i < 3; ++i) H[i] = sgn * n[i];
	}
}

/* f (roughglass.cpp:345-411) */
static void roughglass_f(const float *P, const float wi[3], const float wo[3], float out[3]) {
	const int distr = (int) P[0];
	const int reflect = wi[2] * wo[2] > 0;
	float etaI = P[3], etaT = P[2];
	if (wi[2] < 0) { float t = etaI; etaI = etaT; etaT = t; }
	float H[3];
	rg_half_vector(P, wi, wo, reflect, etaI, etaT, H);
	const float alpha = P[1];
	out[0] = out[1] = out[2] = 0.0f;
	const float D = rg_eval_d(distr, H, alpha);
	if (D == 0)
		return;
	const float F = orc_fresnel(v3_dot(wi, H), P[3], P[2]);
	const float G = rg_smith_g1(distr, wi, H, alpha) * rg_smith_g1(distr, wo, H, alpha);
	if (reflect) {
		float value = F * D * G / (4.0f * wi[2] * wo[2]);
		for (int i = 0; i < 3; ++i) out[i] = P[4+i] * value;
	} else {
		float sqrtDenom = etaI * v3_dot(wi, H) + etaT * v3_dot(wo, H);
		float value = ((1 - F) * D * G * etaT * etaT * v3_dot(wi, H)*v3_dot(wo, H)) /
			(wi[2] * wo[2] * sqrtDenom * sqrtDenom);
		value *= (etaI*etaI) / (etaT*etaT);                    /* bRec.quantity == ERadiance */
		for (int i = 0; i < 3; ++i) out[i] = P[7+i] * fabsf(value);
	}
}

/* pdf (roughglass.cpp:413-485), no sampler: the clamped Fresnel term of the surface normal */
static float roughglass_pdf(const float *P, const float wi[3], const float wo[3]) {
	const int distr = (int) P[0];
	const int reflect = wi[2] * wo[2] > 0;
	float etaI = P[3], etaT = P[2];
	if (wi[2] < 0) { float t = etaI; etaI = etaT; etaT = t; }
	float H[3], dwh_dwo;
	rg_half_vector(P, wi, wo, reflect, etaI, etaT, H);
	if (reflect) {
		dwh_dwo = 1.0f / (4.0f * v3_dot(wo, H));
	} else {
		float sqrtDenom = etaI * v3_dot(wi, H) + etaT * v3_dot(wo, H);
		dwh_dwo = (etaT*etaT * v3_dot(wo, H)) / (sqrtDenom*sqrtDenom);
	}
	float alpha = P[1];
	alpha = alpha * (1.2f - 0.2f * sqrtf(fabsf(wi[2])));
	float prob = rg_eval_d(distr, H, alpha);
	const float F = fminf_((float) 0.9f, fmaxf_((float) 0.1f, orc_fresnel(wi[2], P[3], P[2])));
	prob *= reflect ? F : (1-F);
	return fabsf(prob * H[2] * dwh_dwo);
}

/* sample(bRec, sample) (roughglass.cpp:487-617), no sampler */
static void roughglass_sample(const float *P, const float wi[3], const float _sample[2], float wo[3], uint32_t *stype, float out[3]) {
	const int distr = (int) P[0];
	float sample[2] = { _sample[0], _sample[1] };
	int choseReflection = 1;
	out[0] = out[1] = out[2] = 0.0f;
	float sampleF = fminf_((float) 0.9f, fmaxf_((float) 0.1f, orc_fresnel(wi[2], P[3], P[2])));
	if (sample[0] < sampleF) {
		sample[0] /= sampleF;
	} else {
		sample[0] = (sample[0] - sampleF) / (1 - sampleF);
		choseReflection = 0;
	}
	const float alpha = P[1];
	const float sampleAlpha = alpha * (1.2f - 0.2f * sqrtf(fabsf(wi[2])));
	float m[3];
	rg_sample_d(distr, sample, sampleAlpha, m);
	float result[3];
	if (choseReflection) {
		/* reflect(wi, m) = 2 * dot(wi, m) * Vector(m) - wi (roughglass.cpp:180-182) */
		const float k = 2 * v3_dot(wi, m);
		for (int i = 0; i < 3; ++i) wo[i] = k * m[i] - wi[i];
		*stype = T_GLOSSY_REFL;
		if (wi[2] * wo[2] <= 0)
			return;
		for (int i = 0; i < 3; ++i) result[i] = P[4+i];
	} else {
		float etaI = P[3], etaT = P[2];
		if (wi[2] < 0) { float t = etaI; etaI = etaT; etaT = t; }
		/* refract (roughglass.cpp:185-201) */
		const float eta = etaI / etaT, c = v3_dot(wi, m);
		const float cosThetaTSqr = 1 + eta * eta * (c*c-1);
		if (cosThetaTSqr < 0)
			return;
		const float k = eta*c - rg_signum(wi[2]) * sqrtf(cosThetaTSqr);
		for (int i = 0; i < 3; ++i) wo[i] = m[i] * k - wi[i] * eta;
		*stype = T_GLOSSY_TRANS;
		if (wi[2] * wo[2] >= 0)
			return;
		const float scale = (etaI*etaI) / (etaT*etaT);
		for (int i = 0; i < 3; ++i) result[i] = P[7+i] * scale;
	}
	float numerator = rg_eval_d(distr, m, alpha) * rg_smith_g1(distr, wi, m, alpha) * rg_smith_g1(distr, wo, m, alpha) * v3_dot(wi, m);
	float denominator = rg_eval_d(distr, m, sampleAlpha) * m[2] * wi[2] * wo[2];
	float F = orc_fresnel(v3_dot(wi, m), P[3], P[2]);
	if (!choseReflection) {
		sampleF = 1-sampleF;
		F = 1-F;
	}
	numerator *= F;
	denominator *= sampleF;
	const float w = fabsf(numerator / denominator);
	for (int i = 0; i < 3; ++i) out[i] = result[i] * w;
}

/* ---- Lambertian (src/bsdfs/lambertian.cpp:95-126) ---- */
static void lambertian_f(const float *P, const float wi[3], const float wo[3], float out[3]) {
	if (wi[2] <= 0 || wo[2] <= 0) { out[0] = out[1] = out[2] = 0.0f; return; }
	out[0] = P[0] * ORC_INV_PI; out[1] = P[1] * ORC_INV_PI; out[2] = P[2] * ORC_INV_PI;
}
static float lambertian_pdf(const float wi[3], const float wo[3]) {
	if (wi[2] <= 0 || wo[2] <= 0)
		return 0.0f;
	return wo[2] * ORC_INV_PI;
}

/* ---- RoughMetal (src/bsdfs/roughmetal.cpp:119-167) ---- */
static void roughmetal_f(const float *P, const float wi[3], const float wo[3], float out[3]) {
	if (wi[2] <= 0 || wo[2] <= 0) { out[0] = out[1] = out[2] = 0.0f; return; }
	float h[3], Hr[3], F[3];
	v3_add(h, wi, wo); v3_normalize(Hr, h);
	orc_fresnel_conductor(v3_dot(wi, Hr), P + 1, P + 4, F);
	float D = beckmann_d(P[0], Hr);
	float G = smith_beckmann_g1(P[0], wi, Hr) * smith_beckmann_g1(P[0], wo, Hr);
	float k = D * G / (4.0f * wi[2] * wo[2]);
	for (int i = 0; i < 3; ++i) out[i] = P[7+i] * (F[i] * k);
}
static float roughmetal_pdf(const float *P, const float wi[3], const float wo[3]) {
	if (wi[2] <= 0 || wo[2] <= 0)
		return 0.0f;
	float h[3], Hr[3];
	v3_add(h, wi, wo); v3_normalize(Hr, h);
	float dwhr_dwo = 1.0f / (4.0f * fabsf(v3_dot(wo, Hr)));
	return beckmann_d(P[0], Hr) * Hr[2] * dwhr_dwo;
}

/* ---- Microfacet (src/bsdfs/microfacet.cpp:151-269) ---- */
static void microfacet_f(const float *P, const float wi[3], const float wo[3], float out[3]) {
	out[0] = out[1] = out[2] = 0.0f;
	if (wi[2] <= 0 || wo[2] <= 0)
		return;
	const float alphaB = P[0], kd = P[1], ks = P[2], intIOR = P[3], extIOR = P[4];
	float h[3], Hr[3];
	v3_add(h, wi, wo); v3_normalize(Hr, h);
	float F = orc_fresnel(v3_dot(wi, Hr), extIOR, intIOR);
	/* fSpec (:151-161) * (F * m_ks) */
	float D = beckmann_d(alphaB, Hr);
	float G = smith_beckmann_g1(alphaB, wi, Hr) * smith_beckmann_g1(alphaB, wo, Hr);
	float specRef = D * G / (4.0f * wi[2] * wo[2]);
	float fk = F * ks;
	for (int i = 0; i < 3; ++i) out[i] += (P[8+i] * specRef) * fk;
	float dk = ORC_INV_PI * (1-F) * kd;
	for (int i = 0; i < 3; ++i) out[i] += P[5+i] * dk;
}
static float microfacet_pdf_spec(const float *P, const float wi[3], const float wo[3]) {
	float h[3], Hr[3];
	v3_add(h, wi, wo); v3_normalize(Hr, h);
	return beckmann_d(P[0], Hr) * Hr[2] / (4.0f * fabsf(v3_dot(wo, Hr)));
}
static float microfacet_pdf(const float *P, const float wi[3], const float wo[3]) {
	if (wi[2] <= 0 || wo[2] <= 0)
		return 0.0f;
	const float kd = P[1], ks = P[2], intIOR = P[3], extIOR = P[4];
	float fr = orc_fresnel(wi[2], extIOR, intIOR);
	fr = fminf_(fmaxf_(fr, (float) 0.05f), (float) 0.95f);
	float diffuseSamplingWeight = (1-fr) * kd;
	float specularSamplingWeight = fr * ks;
	float normalization = 1 / (diffuseSamplingWeight + specularSamplingWeight);
	return (specularSamplingWeight * microfacet_pdf_spec(P, wi, wo)
	      + diffuseSamplingWeight * (wo[2] * ORC_INV_PI)) * normalization;
}
/* Microfacet::sample(bRec, sample) (:233-262): returns f/pdf */
static void microfacet_sample(const float *P, const float wi[3], const float _sample[2],
                              float wo[3], uint32_t *stype, float out[3]) {
	float sample[2] = { _sample[0], _sample[1] };
	out[0] = out[1] = out[2] = 0.0f;
	if (wi[2] <= 0)
		return;
	const float kd = P[1], ks = P[2], intIOR = P[3], extIOR = P[4];
	float fr = orc_fresnel(wi[2], extIOR, intIOR);
	fr = fminf_(fmaxf_(fr, (float) 0.05f), (float) 0.95f);
	float diffuseSamplingWeight = (1-fr) * kd;
	float specularSamplingWeight = fr * ks;
	float normalization = 1 / (diffuseSamplingWeight + specularSamplingWeight);
	specularSamplingWeight *= normalization;
	diffuseSamplingWeight *= normalization;
	if (sample[0] < specularSamplingWeight) {
		sample[0] /= specularSamplingWeight;
		/* sampleSpecular (:203-218) */
		float m[3];
		sample_beckmann_d(P[0], sample, m);
		mf_reflect(wi, m, wo);
		*stype = T_GLOSSY_REFL;
		if (wo[2] <= 0)
			return;
		float pdfValue = microfacet_pdf(P, wi, wo);
		if (pdfValue == 0)
			return;
		float f[3]; microfacet_f(P, wi, wo, f);
		float recip = 1.0f / pdfValue;
		out[0] = f[0] * recip; out[1] = f[1] * recip; out[2] = f[2] * recip;
	} else {
		sample[0] = (sample[0] - specularSamplingWeight) / diffuseSamplingWeight;
		/* sampleLambertian (:224-229) */
		orc_square_to_hemisphere_psa(sample, wo);
		*stype = T_DIFFUSE_REFL;
		float f[3]; microfacet_f(P, wi, wo, f);
		float recip = 1.0f / microfacet_pdf(P, wi, wo);
		out[0] = f[0] * recip; out[1] = f[1] * recip; out[2] = f[2] * recip;
	}
}

/* ---- Phong (src/bsdfs/phong.cpp:104-212); params after configure() ---- */
#define ORC_INV_TWOPI 0.15915494309189533577f
static void phong_f(const float *P, const float wi[3], const float wo[3], float out[3]) {
	out[0] = out[1] = out[2] = 0.0f;
	if (wi[2] <= 0 || wo[2] <= 0)
		return;
	const float R[3] = { -wi[0], -wi[1], wi[2] };
	float alpha = v3_dot(R, wo);
	float specRef;
	if (alpha <= 0.0f)
		specRef = 0.0f;
	else
		specRef = (P[0] + 2) * ORC_INV_TWOPI * orc_powf(alpha, P[0]) * P[2];
	for (int i = 0; i < 3; ++i) out[i] += P[8+i] * specRef;
	const float dk = ORC_INV_PI * P[1];
	for (int i = 0; i < 3; ++i) out[i] += P[5+i] * dk;
}
static float phong_pdf_spec(const float *P, const float wi[3], const float wo[3]) {
	const float R[3] = { -wi[0], -wi[1], wi[2] };
	float alpha = v3_dot(R, wo);
	float specPdf = orc_powf(alpha, P[0]) * (P[0] + 1.0f) / (2.0f * ORC_PI);
	if (alpha <= 0)
		specPdf = 0;
	return specPdf;
}
static float phong_pdf(const float *P, const float wi[3], const float wo[3]) {
	if (wo[2] <= 0 || wi[2] <= 0)
		return 0.0f;
	return P[3] * phong_pdf_spec(P, wi, wo) + P[4] * (wo[2] * ORC_INV_PI);
}
/* Phong::sample(bRec, sample): returns f / pdf */
static void phong_sample(const float *P, const float wi[3], const float _sample[2], float wo[3], uint32_t *stype, float out[3]) {
	float sample[2] = { _sample[0], _sample[1] };
	out[0] = out[1] = out[2] = 0.0f;
	if (wi[2] <= 0)
		return;
	if (sample[0] <= P[3]) {
		sample[0] /= P[3];
		/* sampleSpecular (:157-182) */
		const float R[3] = { -wi[0], -wi[1], wi[2] };
		float sinAlpha = sqrtf(1 - orc_powf(sample[1], 2 / (P[0] + 1)));
		float cosAlpha = orc_powf(sample[1], 1 / (P[0] + 1));
		float phi = (2.0f * ORC_PI) * sample[0];
		float localDir[3] = { sinAlpha * orc_cosf(phi), sinAlpha * orc_sinf(phi), cosAlpha };
		float fs[3], ft[3];
		orc_coordinate_system(R, fs, ft);                         /* Frame(R).toWorld(localDir) */
		for (int i = 0; i < 3; ++i) wo[i] = fs[i] * localDir[0] + ft[i] * localDir[1] + R[i] * localDir[2];
		*stype = T_GLOSSY_REFL;
		if (wo[2] <= 0)
			return;
		float pdfVal = phong_pdf(P, wi, wo);
		if (pdfVal == 0)
			return;
		float f[3]; phong_f(P, wi, wo, f);
		float recip = 1.0f / pdfVal;
		out[0] = f[0] * recip; out[1] = f[1] * recip; out[2] = f[2] * recip;
	} else {
		sample[0] = (sample[0] - P[3]) / P[4];
		orc_square_to_hemisphere_psa(sample, wo);                 /* sampleDiffuse (:188-193) */
		*stype = T_DIFFUSE_REFL;
		float f[3]; phong_f(P, wi, wo, f);
		float recip = 1.0f / phong_pdf(P, wi, wo);
		out[0] = f[0] * recip; out[1] = f[1] * recip; out[2] = f[2] * recip;
	}
}

static void bsdf_f_base(uint32_t type, const float *P, const float wi[3], const float wo[3], float out[3]) {
	switch (type) {
		case MTSGPU_BSDF_LAMBERTIAN: lambertian_f(P, wi, wo, out); break;
		case MTSGPU_BSDF_ROUGHMETAL: roughmetal_f(P, wi, wo, out); break;
		case MTSGPU_BSDF_MICROFACET: microfacet_f(P, wi, wo, out); break;
		case MTSGPU_BSDF_PHONG: phong_f(P, wi, wo, out); break;
		case MTSGPU_BSDF_ROUGHGLASS: roughglass_f(P, wi, wo, out); break;
		case MTSGPU_BSDF_DIFFTRANS:          /* difftrans.cpp:92-98 */
			if (wi[2]*wo[2] >= 0) { out[0] = out[1] = out[2] = 0.0f; }
			else { out[0] = P[0] * ORC_INV_PI; out[1] = P[1] * ORC_INV_PI; out[2] = P[2] * ORC_INV_PI; }
			break;
		default: out[0] = out[1] = out[2] = 0.0f; break;   /* dielectric.cpp:101-103, mirror.cpp:60-62 */
	}
}

static float bsdf_pdf_base(uint32_t type, const float *P, const float wi[3], const float wo[3]) {
	switch (type) {
		case MTSGPU_BSDF_LAMBERTIAN: return lambertian_pdf(wi, wo);
		case MTSGPU_BSDF_ROUGHMETAL: return roughmetal_pdf(P, wi, wo);
		case MTSGPU_BSDF_MICROFACET: return microfacet_pdf(P, wi, wo);
		case MTSGPU_BSDF_PHONG: return phong_pdf(P, wi, wo);
		case MTSGPU_BSDF_ROUGHGLASS: return roughglass_pdf(P, wi, wo);
		case MTSGPU_BSDF_DIFFTRANS: return (wi[2]*wo[2] >= 0) ? 0.0f : fabsf(wo[2]) * ORC_INV_PI;     /* difftrans.cpp:100-104 */
		default: return 0.0f;                              /* dielectric.cpp:105-107, mirror.cpp:64-66 */
	}
}

static void bsdf_sample_base(uint32_t type, const float *P, const float wi[3], const float s[2],
                             float wo[3], float *pdf, uint32_t *stype, float out[3]);

/* TwoSidedBRDF (src/bsdfs/twosided.cpp:80-130) wraps any of the above when MTSGPU_BSDF_TWOSIDED is set */
void orc_bsdf_f(uint32_t type, const float *P, const float wi[3], const float wo[3], float out[3]) {
	float a[3] = { wi[0], wi[1], wi[2] }, b[3] = { wo[0], wo[1], wo[2] };
	if ((type & MTSGPU_BSDF_TWOSIDED) && a[2] < 0) { a[2] *= -1; b[2] *= -1; }
	bsdf_f_base(type & 0xFFu, P, a, b, out);
}

float orc_bsdf_pdf(uint32_t type, const float *P, const float wi[3], const float wo[3]) {
	float a[3] = { wi[0], wi[1], wi[2] }, b[3] = { wo[0], wo[1], wo[2] };
	if ((type & MTSGPU_BSDF_TWOSIDED) && a[2] < 0) { a[2] *= -1; b[2] *= -1; }
	return bsdf_pdf_base(type & 0xFFu, P, a, b);
}

void orc_bsdf_sample(uint32_t type, const float *P, const float wi[3], const float s[2],
                     float wo[3], float *pdf, uint32_t *stype, float out[3]) {
	float a[3] = { wi[0], wi[1], wi[2] };
	int flipped = 0;
	if ((type & MTSGPU_BSDF_TWOSIDED) && a[2] < 0) { a[2] *= -1; flipped = 1; }
	bsdf_sample_base(type & 0xFFu, P, a, s, wo, pdf, stype, out);
	if (flipped && !spec_is_zero(out) && *pdf != 0)
		wo[2] *= -1;
}

/* n query records of one parameter block, the layout of mtsgpu_bsdf_eval (include/mtsgpu.h): queries [n][6], out [n][8];
 * op 0 f, 1 pdf, 2 sample(bRec, pdf, sample) -> wo, pdf, f, sampledType */
void orc_bsdf_eval(uint32_t type, const float *P, int op, uint32_t n, const float *queries, float *out) {
	#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < (int64_t) n; ++i) {
		const float *q = queries + 6 * i;
		float *o = out + 8 * i;
		for (int k = 0; k < 8; ++k) o[k] = 0.0f;
		if (op == 0) {
			orc_bsdf_f(type, P, q, q + 3, o);
		} else if (op == 1) {
			o[0] = orc_bsdf_pdf(type, P, q, q + 3);
		} else {
			uint32_t st = 0;
			orc_bsdf_sample(type, P, q, q + 3, o, o + 3, &st, o + 4);
			memcpy(o + 7, &st, 4);
		}
	}
}

/* BSDF::sample(bRec, pdf, sample): value NOT divided by pdf */
static void bsdf_sample_base(uint32_t type, const float *P, const float wi[3], const float s[2],
                             float wo[3], float *pdf, uint32_t *stype, float out[3]) {
	out[0] = out[1] = out[2] = 0.0f; *pdf = 0.0f; *stype = 0;
	wo[0] = wo[1] = wo[2] = 0.0f;
	switch (type) {
	case MTSGPU_BSDF_LAMBERTIAN: {
		/* lambertian.cpp:118-126 */
		if (wi[2] <= 0)
			return;
		orc_square_to_hemisphere_psa(s, wo);
		*stype = T_DIFFUSE_REFL;
		*pdf = wo[2] * ORC_INV_PI;
		out[0] = P[0] * ORC_INV_PI; out[1] = P[1] * ORC_INV_PI; out[2] = P[2] * ORC_INV_PI;
		return;
	}
	case MTSGPU_BSDF_DIELECTRIC: {
		/* dielectric.cpp:200-261 (both components sampled, quantity == ERadiance) */
		float cosThetaI = wi[2], etaI = P[1], etaT = P[0];
		int entering = cosThetaI > 0.0f;
		if (!entering) { float t = etaI; etaI = etaT; etaT = t; }
		float eta = etaI / etaT, sinThetaTSqr = eta*eta * (1.0f - wi[2] * wi[2]);
		float Fr, cosThetaT = 0;
		if (sinThetaTSqr >= 1.0f) {
			Fr = 1.0f;
		} else {
			cosThetaT = sqrtf(1.0f - sinThetaTSqr);
			Fr = orc_fresnel_dielectric(fabsf(cosThetaI), cosThetaT, etaI, etaT);
			if (entering)
				cosThetaT = -cosThetaT;
		}
		if (s[0] <= Fr) {
			*stype = T_DELTA_REFL;
			wo[0] = -wi[0]; wo[1] = -wi[1]; wo[2] = wi[2];
			*pdf = Fr * fabsf(wo[2]);
			out[0] = P[2] * Fr; out[1] = P[3] * Fr; out[2] = P[4] * Fr;
		} else {
			*stype = T_DELTA_TRANS;
			wo[0] = -eta*wi[0]; wo[1] = -eta*wi[1]; wo[2] = cosThetaT;
			*pdf = (1-Fr) * fabsf(wo[2]);
			for (int i = 0; i < 3; ++i) out[i] = P[5+i] * (1-Fr) * (eta*eta);
		}
		return;
	}
	case MTSGPU_BSDF_ROUGHMETAL: {
		/* BSDF::sample(bRec, pdf, sample) fallback (bsdf.cpp:37-48) over roughmetal.cpp:148-167 */
		if (wi[2] <= 0)
			return;
		float m[3];
		sample_beckmann_d(P[0], s, m);
		mf_reflect(wi, m, wo);
		*stype = T_GLOSSY_REFL;
		if (wo[2] <= 0)
			return;
		float f[3]; roughmetal_f(P, wi, wo, f);
		float p = roughmetal_pdf(P, wi, wo);
		float recip = 1.0f / p;
		float q[3] = { f[0] * recip, f[1] * recip, f[2] * recip };
		if (spec_is_zero(q))
			return;
		*pdf = p;
		out[0] = f[0]; out[1] = f[1]; out[2] = f[2];
		return;
	}
	case MTSGPU_BSDF_MICROFACET: {
		float q[3];
		microfacet_sample(P, wi, s, wo, stype, q);
		if (spec_is_zero(q))
			return;
		*pdf = microfacet_pdf(P, wi, wo);
		microfacet_f(P, wi, wo, out);
		return;
	}
	case MTSGPU_BSDF_MIRROR: {
		/* mirror.cpp:77-86 */
		wo[0] = -wi[0]; wo[1] = -wi[1]; wo[2] = wi[2];
		*stype = T_DELTA_REFL;
		*pdf = fabsf(wo[2]);
		out[0] = P[0]; out[1] = P[1]; out[2] = P[2];
		return;
	}
	case MTSGPU_BSDF_PHONG: {
		/* BSDF::sample(bRec, pdf, sample) fallback (bsdf.cpp:37-48) over phong.cpp:195-212 */
		float q[3];
		phong_sample(P, wi, s, wo, stype, q);
		if (spec_is_zero(q))
			return;
		*pdf = phong_pdf(P, wi, wo);
		phong_f(P, wi, wo, out);
		return;
	}
	case MTSGPU_BSDF_ROUGHGLASS: {
		/* BSDF::sample(bRec, pdf, sample) fallback (bsdf.cpp:37-48) over roughglass.cpp:487-617 */
		float q[3];
		roughglass_sample(P, wi, s, wo, stype, q);
		if (spec_is_zero(q))
			return;
		*pdf = roughglass_pdf(P, wi, wo);
		roughglass_f(P, wi, wo, out);
		return;
	}
	case MTSGPU_BSDF_DIFFTRANS: {
		/* difftrans.cpp:119-131 */
		orc_square_to_hemisphere_psa(s, wo);
		if (wi[2] > 0)
			wo[2] *= -1;
		*stype = T_DIFFUSE_TRANS;
		*pdf = fabsf(wo[2]) * ORC_INV_PI;
		if (wo[2] == 0)
			return;
		out[0] = P[0] * ORC_INV_PI; out[1] = P[1] * ORC_INV_PI; out[2] = P[2] * ORC_INV_PI;
		return;
	}
	default: return;
	}
}

/* ========================================================================== */
/* Samplers                                                                   */
/* ========================================================================== */
typedef struct {
	int kind;            /* 0 keyed independent, 1 keyed LD, 2 MT independent, 3 MT LD, 4 halton, 5 hammersley, 6 keyed stratified */
	int resolution;      /* StratifiedSampler::m_resolution */
	int qdepth;          /* m_sampleDepth of the QMC samplers */
	uint64_t stream;     /* keyed overflow / independent stream */
	orc_random *mt;
	/* LD state */
	int depth, d1, d2; uint32_t spp, index;
	const uint32_t *scr, *perm;      /* keyed tables */
	const float *t1d, *t2d;          /* MT tables */
	/* Sampler::m_sampleArrays2D / m_req2D / m_sampleDepth2DArray (sampler.h) */
	const float *arr[2]; uint32_t arr_size[2]; int n_arr, adepth;
} sampler_t;

/* Sampler::next2DArray (src/librender/sampler.cpp:76-87) */
static const float *sampler_next2d_array(sampler_t *s, uint32_t size) {
	if (s->adepth >= s->n_arr || s->arr_size[s->adepth] != size) {
		fprintf(stderr, "oracle: a size-%u 2D array was not requested\n", size);
		abort();
	}
	const float *a = s->arr[s->adepth] + 2 * (size_t) s->index * size;
	s->adepth++;
	return a;
}

/* primeTable (src/libcore/util.cpp:64-122): the first 1000 primes */
static int orc_prime(int i) {
	static int table[1000], ready = 0;
	if (!ready) {
		int n = 0;
		for (int c = 2; n < 1000; ++c) {
			int isPrime = 1;
			for (int d = 2; d * d <= c; ++d) if (c % d == 0) { isPrime = 0; break; }
			if (isPrime) table[n++] = c;
		}
		ready = 1;
	}
	return table[i];
}
void orc_prime_table_init(void) { (void) orc_prime(0); }

/* HaltonSequence::nextValue (halton.cpp:73-75) / HammersleySequence::nextValue (hammersley.cpp:75-82) */
static float qmc_next_value(sampler_t *s) {
	if (s->kind == 5) {
		if (s->qdepth == 0) {
			s->qdepth++;
			return s->index * (1.0f / s->spp);          /* m_sampleIndex * m_invSamplesPerPixel (hammersley.cpp:40) */
		}
		return orc_radical_inverse(orc_prime((s->qdepth++) - 1), s->index);
	}
	return orc_radical_inverse(orc_prime(s->qdepth++), s->index);
}

static float sampler_next_float(sampler_t *s) {
	if (s->mt) return orc_random_next_float(s->mt);
	return orc_ulong_to_float(orc_keyed_next(&s->stream));
}

/* next1D (independent.cpp:72-74, ldsampler.cpp:172-178) */
static float sampler_next1d(sampler_t *s) {
	if (s->kind == 6) {
		/* StratifiedSampler::next1D (stratified.cpp:155-163) */
		if (s->d1 < s->depth) {
			int k = (int) s->perm[((size_t) (s->d1++) * 2 + 0) * s->spp + s->index];
			return (k + sampler_next_float(s)) * (1 / (float) s->spp);
		}
		return sampler_next_float(s);
	}
	if (s->kind >= 4)
		return qmc_next_value(s);
	if ((s->kind == 1 || s->kind == 3) && s->d1 < s->depth) {
		int i = s->d1++;
		if (s->kind == 3) return s->t1d[(size_t) i * s->spp + s->index];
		return orc_u32_to_unit(orc_vdc_bits(s->perm[((size_t) i * 2 + 0) * s->spp + s->index], s->scr[i*3+0]));
	}
	return sampler_next_float(s);
}

/* next2D (independent.cpp:76-81, ldsampler.cpp:180-186).  The overflow branch of
 * ldsampler.cpp:185 leaves the evaluation order of its two nextFloat() calls to
 * the compiler; x-then-y is used here, as independent.cpp enforces. */
static void sampler_next2d(sampler_t *s, float out[2]) {
	if (s->kind == 6) {
		/* StratifiedSampler::next2D (stratified.cpp:165-181); x is drawn first */
		if (s->d2 < s->depth) {
			int k = (int) s->perm[((size_t) (s->d2++) * 2 + 1) * s->spp + s->index];
			int x = k % s->resolution, y = k / s->resolution;
			const float invResolution = 1 / (float) s->resolution;
			float jx = sampler_next_float(s), jy = sampler_next_float(s);
			out[0] = (x + jx) * invResolution; out[1] = (y + jy) * invResolution;
			return;
		}
		float v1 = sampler_next_float(s), v2 = sampler_next_float(s);
		out[0] = v1; out[1] = v2;
		return;
	}
	if (s->kind >= 4) {
		out[0] = qmc_next_value(s);
		out[1] = qmc_next_value(s);
		return;
	}
	if ((s->kind == 1 || s->kind == 3) && s->d2 < s->depth) {
		int i = s->d2++;
		if (s->kind == 3) {
			out[0] = s->t2d[((size_t) i * s->spp + s->index) * 2 + 0];
			out[1] = s->t2d[((size_t) i * s->spp + s->index) * 2 + 1];
			return;
		}
		uint32_t k = s->perm[((size_t) i * 2 + 1) * s->spp + s->index];
		out[0] = orc_u32_to_unit(orc_vdc_bits(k, s->scr[i*3+1]));
		out[1] = orc_u32_to_unit(orc_sobol2_bits(k, s->scr[i*3+2]));
		return;
	}
	float value1 = sampler_next_float(s);
	float value2 = sampler_next_float(s);
	out[0] = value1; out[1] = value2;
}

/* ========================================================================== */
/* Camera (src/cameras/perspective.cpp:77-112)                                */
/* ========================================================================== */
static void camera_generate_ray(const mtsgpu_camera *cam, const float dirSample[2], const float lensSample[2], ray_t *ray) {
	const float (*m)[4] = (const float (*)[4]) cam->raster_to_camera;
	const float (*w)[4] = (const float (*)[4]) cam->camera_to_world;
	/* Transform::operator()(Point, Point&) (transform.h:133-149) on (x, y, 0) */
	float px = dirSample[0], py = dirSample[1], pz = 0;
	float ic[3];
	ic[0] = m[0][0] * px + m[0][1] * py + m[0][2] * pz + m[0][3];
	ic[1] = m[1][0] * px + m[1][1] * py + m[1][2] * pz + m[1][3];
	ic[2] = m[2][0] * px + m[2][1] * py + m[2][2] * pz + m[2][3];
	float wv = m[3][0] * px + m[3][1] * py + m[3][2] * pz + m[3][3];
	if (wv != 1.0f)
		v3_div(ic, ic, wv);
	if (cam->kind == 1) {
		/* OrthographicCamera::generateRay (src/cameras/orthographic.cpp:104-118) */
		const float ldir[3] = { 0, 0, 1 };
		float o[3], d[3];
		for (int i = 0; i < 3; ++i) {
			o[i] = w[i][0] * ic[0] + w[i][1] * ic[1] + w[i][2] * ic[2] + w[i][3];
			d[i] = w[i][0] * ldir[0] + w[i][1] * ldir[1] + w[i][2] * ldir[2];
		}
		float wo_ = w[3][0] * ic[0] + w[3][1] * ic[1] + w[3][2] * ic[2] + w[3][3];
		if (wo_ != 1.0f)
			v3_div(o, o, wo_);
		ray_init(ray, o, d);
		ray->mint = 0; ray->maxt = cam->far_clip - cam->near_clip;
		return;
	}
	float lo[3] = { 0, 0, 0 };
	if (cam->aperture_radius > 0.0f) {
		/* perspective.cpp:90-103: sample the aperture, aim at the focal plane */
		float lensPos[2];
		orc_square_to_disk_concentric(lensSample, lensPos);
		lensPos[0] *= cam->aperture_radius; lensPos[1] *= cam->aperture_radius;
		float tf = cam->focus_depth / ic[2];
		float itsFocal[3] = { 0.0f + tf * ic[0], 0.0f + tf * ic[1], 0.0f + tf * ic[2] };
		lo[0] += lensPos[0];
		lo[1] += lensPos[1];
		v3_sub(ic, itsFocal, lo);
	}
	float ld[3];
	v3_normalize(ld, ic);
	float invZ = 1.0f / ld[2];
	float mint = cam->near_clip * invZ, maxt = cam->far_clip * invZ;
	/* Transform::operator()(Ray, Ray&) (transform.h:219-235): o as point, d as vector */
	float o[3], d[3];
	for (int i = 0; i < 3; ++i) {
		o[i] = w[i][0] * lo[0] + w[i][1] * lo[1] + w[i][2] * lo[2] + w[i][3];
		d[i] = w[i][0] * ld[0] + w[i][1] * ld[1] + w[i][2] * ld[2];
	}
	float wv2 = w[3][0] * lo[0] + w[3][1] * lo[1] + w[3][2] * lo[2] + w[3][3];
	if (wv2 != 1.0f)
		v3_div(o, o, wv2);
	ray_init(ray, o, d);
	ray->mint = mint; ray->maxt = maxt;
}

/* ========================================================================== */
/* MIPathTracer::Li (src/integrators/path/path.cpp:47-216)                    */
/* ========================================================================== */
static inline float mi_weight(float pdfA, float pdfB) {
	pdfA *= pdfA;
	pdfB *= pdfB;
	return pdfA / (pdfA + pdfB);
}

typedef struct { float Li[3], alpha; int depth; } li_result;

static void path_li(const mtsgpu_scene *sc, const orc_render_params *prm, const ray_t *r, sampler_t *smp,
                    li_result *res, mtsgpu_stats *st) {
	const int maxDepth = prm->max_depth, rrDepth = prm->rr_depth, strictNormals = prm->strict_normals;
	its_t its;
	ray_t ray = *r;
	float Li[3] = { 0.0f, 0.0f, 0.0f };
	int depth = 1;                 /* RadianceQueryRecord::newQuery (integrator.h:186-191) */
	int emitted = 1;               /* type & EEmittedRadiance; ECameraRay has it */

	/* rRec.rayIntersect(ray) (records.inl:89-105): alpha = 1 on hit, 0 on miss (no medium) */
	int valid = scene_ray_intersect(sc, &ray, &its, st);
	res->alpha = valid ? 1.0f : 0.0f;
	ray.mint = ORC_EPS;

	float pathThroughput[3] = { 1.0f, 1.0f, 1.0f };

	while (depth <= maxDepth || maxDepth < 0) {
		if (!valid) {
			/* scene->LeBackground(ray) (scene.h:403-405): ConstantLuminaire::Le = intensity */
			if (emitted && sc->background_lum >= 0) {
				const float *P = sc->lum_params + MTSGPU_LUM_NPARAMS * (size_t) sc->background_lum;
				float le[3] = { P[0], P[1], P[2] };
				if (sc->lum_type[sc->background_lum] == MTSGPU_LUM_ENVMAP) env_le_ray(sc, P, ray.d, le);
				for (int i = 0; i < 3; ++i) Li[i] += pathThroughput[i] * le[i];
			}
			break;
		}
		const int bsdfIdx = sc->shape_bsdf[its.shape];
		if (bsdfIdx < 0)
			break;
		const uint32_t btype = sc->bsdf_type[bsdfIdx];
		const float *BP = sc->bsdf_params + MTSGPU_BSDF_NPARAMS * (size_t) bsdfIdx;
		const int shapeLum = sc->shape_lum[its.shape];
		float md[3] = { -ray.d[0], -ray.d[1], -ray.d[2] };

		if (shapeLum >= 0 && emitted) {
			float le[3];
			area_le(sc, shapeLum, its.geoN, md, le);
			for (int i = 0; i < 3; ++i) Li[i] += pathThroughput[i] * le[i];
		}

		if (maxDepth > 0 && depth >= maxDepth)
			break;

		/* ---- luminaire sampling ---- */
		float wiDotGeoN = -v3_dot(its.geoN, ray.d), wiDotShN = its.wi[2];
		if (wiDotGeoN * wiDotShN < 0 && strictNormals)
			break;

		lrec_t lRec; memset(&lRec, 0, sizeof(lRec));
		float s2[2];
		sampler_next2d(smp, s2);
		if (scene_sample_luminaire(sc, its.p, &lRec, s2, st)) {
			const float wo[3] = { -lRec.d[0], -lRec.d[1], -lRec.d[2] };
			float woL[3] = { v3_dot(wo, its.shS), v3_dot(wo, its.shT), v3_dot(wo, its.shN) };
			float bsdfVal[3];
			orc_bsdf_f(btype, BP, its.wi, woL, bsdfVal);
			float ac = fabsf(woL[2]);
			bsdfVal[0] *= ac; bsdfVal[1] *= ac; bsdfVal[2] *= ac;
			float woDotGeoN = v3_dot(its.geoN, wo);
			if (!spec_is_zero(bsdfVal) && (!strictNormals || woDotGeoN * woL[2] > 0)) {
				/* Luminaire::isIntersectable() || isBackgroundLuminaire() (path.cpp:118-120): false for delta lights */
				const uint32_t lt = sc->lum_type[lRec.lum];
				float bsdfPdf = (lt == MTSGPU_LUM_AREA || lt == MTSGPU_LUM_CONSTANT || lt == MTSGPU_LUM_ENVMAP) ? orc_bsdf_pdf(btype, BP, its.wi, woL) : 0;
				const float weight = mi_weight(lRec.pdf, bsdfPdf);
				for (int i = 0; i < 3; ++i)
					Li[i] += pathThroughput[i] * lRec.value[i] * bsdfVal[i] * weight;
			}
		}

		/* ---- BSDF sampling ---- */
		float woL[3], bsdfPdf, bsdfVal[3];
		uint32_t sampledType;
		sampler_next2d(smp, s2);
		orc_bsdf_sample(btype, BP, its.wi, s2, woL, &bsdfPdf, &sampledType, bsdfVal);
		if (!spec_is_zero(bsdfVal)) {      /* sampleCos: * |cosTheta(wo)| (bsdf.h:273-279) */
			float ac = fabsf(woL[2]);
			bsdfVal[0] *= ac; bsdfVal[1] *= ac; bsdfVal[2] *= ac;
		}
		if (spec_is_zero(bsdfVal))
			break;
		{
			float recip = 1.0f / bsdfPdf;
			bsdfVal[0] *= recip; bsdfVal[1] *= recip; bsdfVal[2] *= recip;
		}
		float wo[3];
		for (int i = 0; i < 3; ++i)
			wo[i] = its.shS[i] * woL[0] + its.shT[i] * woL[1] + its.shN[i] * woL[2];
		float woDotGeoN = v3_dot(its.geoN, wo);
		if (woDotGeoN * woL[2] <= 0 && strictNormals)
			break;

		ray_init(&ray, its.p, wo);
		int hitLuminaire = 0;
		valid = scene_ray_intersect(sc, &ray, &its, st);
		if (valid) {
			int l = sc->shape_lum[its.shape];
			if (l >= 0) {
				/* LuminaireSamplingRecord(its, -ray.d) (records.inl:82-87) */
				float nd[3] = { -ray.d[0], -ray.d[1], -ray.d[2] };
				for (int i = 0; i < 3; ++i) { lRec.p[i] = its.p[i]; lRec.n[i] = its.geoN[i]; lRec.d[i] = nd[i]; }
				lRec.lum = l;
				area_le(sc, l, its.geoN, nd, lRec.value);
				hitLuminaire = 1;
			}
		} else {
			if (sc->background_lum >= 0) {
				const float *P = sc->lum_params + MTSGPU_LUM_NPARAMS * (size_t) sc->background_lum;
				lRec.lum = sc->background_lum;
				lRec.value[0] = P[0]; lRec.value[1] = P[1]; lRec.value[2] = P[2];
				if (sc->lum_type[sc->background_lum] == MTSGPU_LUM_ENVMAP) env_le_ray(sc, P, ray.d, lRec.value);
				lRec.d[0] = -ray.d[0]; lRec.d[1] = -ray.d[1]; lRec.d[2] = -ray.d[2];
				hitLuminaire = 1;
			} else {
				depth++;
				break;
			}
		}

		if (hitLuminaire) {
			const float lumPdf = (!(sampledType & T_DELTA)) ? scene_pdf_luminaire(sc, ray.o, &lRec) : 0;
			const float weight = mi_weight(bsdfPdf, lumPdf);
			for (int i = 0; i < 3; ++i)
				Li[i] += pathThroughput[i] * lRec.value[i] * bsdfVal[i] * weight;
		}

		/* ---- indirect illumination ---- */
		if (!valid)
			break;
		emitted = 0;                 /* rRec.type = ERadianceNoEmission */

		if (depth >= rrDepth && !(sampledType & T_TRANSMISSION)) {
			float mx = bsdfVal[0];
			mx = fmaxf_(mx, bsdfVal[1]); mx = fmaxf_(mx, bsdfVal[2]);
			float approxAlbedo = fminf_((float) 0.9f, mx);
			if (sampler_next1d(smp) > approxAlbedo) {
				break;
			} else {
				float recip = 1.0f / approxAlbedo;
				pathThroughput[0] *= recip; pathThroughput[1] *= recip; pathThroughput[2] *= recip;
			}
		}
		pathThroughput[0] *= bsdfVal[0]; pathThroughput[1] *= bsdfVal[1]; pathThroughput[2] *= bsdfVal[2];
		depth++;
	}
	res->Li[0] = Li[0]; res->Li[1] = Li[1]; res->Li[2] = Li[2];
	res->depth = depth;
}

/* MIDirectIntegrator::Li (src/integrators/direct/direct.cpp:64-198) for a camera ray (rRec.depth == 1); sample counts
 * above one draw their points from Sampler::next2DArray (direct.cpp:122-127,156-161) */
static void direct_li(const mtsgpu_scene *sc, const orc_render_params *prm, const ray_t *r, sampler_t *smp,
                      li_result *res, mtsgpu_stats *st) {
	its_t its, bsdfIts;
	ray_t ray = *r;
	float Li[3] = { 0.0f, 0.0f, 0.0f };
	/* configure() (direct.cpp:51-56) */
	const int numLuminaireSamples = prm->luminaire_samples, numBSDFSamples = prm->bsdf_samples;
	const float weightBSDF = 1 / (float) numBSDFSamples, weightLum = 1 / (float) numLuminaireSamples;
	const float fracBSDF = numBSDFSamples / (float) (numLuminaireSamples + numBSDFSamples);
	const float fracLum = numLuminaireSamples / (float) (numLuminaireSamples + numBSDFSamples);
	res->depth = 1;

	int valid = scene_ray_intersect(sc, &ray, &its, st);
	res->alpha = valid ? 1.0f : 0.0f;
	if (!valid) {
		if (sc->background_lum >= 0) {
			const float *P = sc->lum_params + MTSGPU_LUM_NPARAMS * (size_t) sc->background_lum;
			float le[3] = { P[0], P[1], P[2] };
			if (sc->lum_type[sc->background_lum] == MTSGPU_LUM_ENVMAP) env_le_ray(sc, P, ray.d, le);
			for (int i = 0; i < 3; ++i) Li[i] = le[i];
		}
		goto done;
	}
	{
		const int shapeLum = sc->shape_lum[its.shape];
		float md[3] = { -ray.d[0], -ray.d[1], -ray.d[2] };
		if (shapeLum >= 0) {
			float le[3];
			area_le(sc, shapeLum, its.geoN, md, le);
			for (int i = 0; i < 3; ++i) Li[i] += le[i];
		}
		const int bsdfIdx = sc->shape_bsdf[its.shape];
		if (bsdfIdx < 0)
			goto done;
		const uint32_t btype = sc->bsdf_type[bsdfIdx];
		const float *BP = sc->bsdf_params + MTSGPU_BSDF_NPARAMS * (size_t) bsdfIdx;

		/* ---- luminaire sampling (direct.cpp:122-150): the sample is drawn even when no luminaire sample is taken ---- */
		lrec_t lRec; memset(&lRec, 0, sizeof(lRec));
		float sample[2];
		const float *sampleArray = sample;
		if (numLuminaireSamples > 1) sampleArray = sampler_next2d_array(smp, (uint32_t) numLuminaireSamples);
		else sampler_next2d(smp, sample);
		for (int k = 0; k < numLuminaireSamples; ++k) {
			if (scene_sample_luminaire(sc, its.p, &lRec, sampleArray + 2 * k, st)) {
				const float wo[3] = { -lRec.d[0], -lRec.d[1], -lRec.d[2] };
				float woL[3] = { v3_dot(wo, its.shS), v3_dot(wo, its.shT), v3_dot(wo, its.shN) };
				float bsdfVal[3];
				orc_bsdf_f(btype, BP, its.wi, woL, bsdfVal);
				float ac = fabsf(woL[2]);
				bsdfVal[0] *= ac; bsdfVal[1] *= ac; bsdfVal[2] *= ac;
				if (!spec_is_zero(bsdfVal)) {
					const uint32_t lt = sc->lum_type[lRec.lum];
					float bsdfPdf = (lt == MTSGPU_LUM_AREA || lt == MTSGPU_LUM_CONSTANT || lt == MTSGPU_LUM_ENVMAP) ? orc_bsdf_pdf(btype, BP, its.wi, woL) : 0;
					const float weight = mi_weight(lRec.pdf * fracLum, bsdfPdf * fracBSDF) * weightLum;
					for (int i = 0; i < 3; ++i)
						Li[i] += lRec.value[i] * bsdfVal[i] * weight;
				}
			}
		}

		/* ---- BSDF sampling (direct.cpp:152-195) ---- */
		sampleArray = sample;
		if (numBSDFSamples > 1) sampleArray = sampler_next2d_array(smp, (uint32_t) numBSDFSamples);
		else sampler_next2d(smp, sample);
		for (int k = 0; k < numBSDFSamples; ++k) {
			float woL[3], bsdfPdf, bsdfVal[3];
			uint32_t sampledType;
			orc_bsdf_sample(btype, BP, its.wi, sampleArray + 2 * k, woL, &bsdfPdf, &sampledType, bsdfVal);
			if (!spec_is_zero(bsdfVal)) {
				float ac = fabsf(woL[2]);
				bsdfVal[0] *= ac; bsdfVal[1] *= ac; bsdfVal[2] *= ac;
			}
			if (spec_is_zero(bsdfVal))
				continue;
			{
				float recip = 1.0f / bsdfPdf;
				bsdfVal[0] *= recip; bsdfVal[1] *= recip; bsdfVal[2] *= recip;
			}
			float wo[3];
			for (int i = 0; i < 3; ++i)
				wo[i] = its.shS[i] * woL[0] + its.shT[i] * woL[1] + its.shN[i] * woL[2];
			ray_t bsdfRay;
			ray_init(&bsdfRay, its.p, wo);
			if (scene_ray_intersect(sc, &bsdfRay, &bsdfIts, st)) {
				int l = sc->shape_lum[bsdfIts.shape];
				if (l < 0)
					continue;
				float nd[3] = { -bsdfRay.d[0], -bsdfRay.d[1], -bsdfRay.d[2] };
				for (int i = 0; i < 3; ++i) { lRec.p[i] = bsdfIts.p[i]; lRec.n[i] = bsdfIts.geoN[i]; lRec.d[i] = nd[i]; }
				lRec.lum = l;
				area_le(sc, l, bsdfIts.geoN, nd, lRec.value);
			} else {
				if (sc->background_lum < 0)
					continue;
				const float *P = sc->lum_params + MTSGPU_LUM_NPARAMS * (size_t) sc->background_lum;
				lRec.lum = sc->background_lum;
				lRec.value[0] = P[0]; lRec.value[1] = P[1]; lRec.value[2] = P[2];
				if (sc->lum_type[sc->background_lum] == MTSGPU_LUM_ENVMAP) env_le_ray(sc, P, bsdfRay.d, lRec.value);
				lRec.d[0] = -bsdfRay.d[0]; lRec.d[1] = -bsdfRay.d[1]; lRec.d[2] = -bsdfRay.d[2];
			}
			const float lumPdf = (!(sampledType & T_DELTA)) ? scene_pdf_luminaire(sc, its.p, &lRec) : 0;
			const float weight = mi_weight(bsdfPdf * fracBSDF, lumPdf * fracLum) * weightBSDF;
			for (int i = 0; i < 3; ++i)
				Li[i] += lRec.value[i] * bsdfVal[i] * weight;
		}
	}
done:
	res->Li[0] = Li[0]; res->Li[1] = Li[1]; res->Li[2] = Li[2];
}

/* Integrator::Li of the configured integrator plugin */
static void integrator_li(const mtsgpu_scene *sc, const orc_render_params *prm, const ray_t *r, sampler_t *smp,
                          li_result *res, mtsgpu_stats *st) {
	if (prm->integrator == 1) direct_li(sc, prm, r, smp, res, st);
	else path_li(sc, prm, r, smp, res, st);
}

/* ========================================================================== */
/* ImageBlock::putSample with the tabulated box filter                        */
/* (include/mitsuba/render/imageblock.h:80-138, src/librender/rfilter.cpp:40-69,*/
/*  src/rfilters/box.cpp).  border = 0, block offset = 0: the film is the     */
/*  union of the blocks (Film::putImageBlock sums them, mfilm.cpp:118-143).   */
/* ========================================================================== */
#define FILTER_RESOLUTION 15
typedef struct { float sizeX, sizeY, factorX, factorY, values[FILTER_RESOLUTION+1][FILTER_RESOLUTION+1]; } tabfilter_t;

static void tabfilter_box(tabfilter_t *f) {
	f->sizeX = f->sizeY = 0.5f;
	f->factorX = FILTER_RESOLUTION / f->sizeX; f->factorY = FILTER_RESOLUTION / f->sizeY;
	float sum = 0;
	for (int y = 0; y < FILTER_RESOLUTION+1; ++y)
		for (int x = 0; x < FILTER_RESOLUTION+1; ++x) {
			if (x == FILTER_RESOLUTION || y == FILTER_RESOLUTION) f->values[y][x] = 0;
			else f->values[y][x] = 1.0f;
			sum += f->values[y][x];
		}
	sum *= 4*f->sizeX*f->sizeY / (FILTER_RESOLUTION*FILTER_RESOLUTION);
	for (int y = 0; y < FILTER_RESOLUTION+1; ++y)
		for (int x = 0; x < FILTER_RESOLUTION+1; ++x)
			f->values[y][x] /= sum;
}

static int put_sample(float *film, int W, int H, const tabfilter_t *filter, float sx, float sy,
                      const float spec[3], float alphaValue) {
	/* Spectrum::isValid (spectrum.h:285-290) */
	for (int i = 0; i < 3; ++i)
		if (spec[i] != spec[i] || spec[i] < 0.0f)
			return 0;
	sx = sx - 0.5f - 0; sy = sy - 0.5f - 0;
	int xStart = (int) ceilf(sx - filter->sizeX), xEnd = (int) floorf(sx + filter->sizeX);
	int yStart = (int) ceilf(sy - filter->sizeY), yEnd = (int) floorf(sy + filter->sizeY);
	if (xStart < 0) xStart = 0;
	if (yStart < 0) yStart = 0;
	if (xEnd > W-1) xEnd = W-1;
	if (yEnd > H-1) yEnd = H-1;
	for (int y = yStart; y <= yEnd; ++y) {
		const float trafoY = filter->factorY * fabsf(y - sy);
		int iy = (int) trafoY; if (iy > FILTER_RESOLUTION) iy = FILTER_RESOLUTION;
		for (int x = xStart; x <= xEnd; ++x) {
			const float trafoX = filter->factorX * fabsf(x - sx);
			int ix = (int) trafoX; if (ix > FILTER_RESOLUTION) ix = FILTER_RESOLUTION;
			float weight = filter->values[iy][ix];
			/* the reference adds spec*0 here; for the finite, non-negative spec that
			 * passed isValid() that is a no-op, and skipping it keeps the multi-threaded
			 * oracle free of writes outside the sample's own pixel */
			if (weight == 0.0f)
				continue;
			float *px = film + 5 * ((size_t) y * W + x);
			px[0] += spec[0] * weight; px[1] += spec[1] * weight; px[2] += spec[2] * weight;
			px[3] += alphaValue * weight;
			px[4] += weight;
		}
	}
	return 1;
}

/* ========================================================================== */
/* SampleIntegrator::renderBlock (src/librender/integrator.cpp:131-170)       */
/* ========================================================================== */
static uint32_t round_to_pow2(uint32_t v) { uint32_t r = 1; while (r < v) r <<= 1; return r; }

/* the arrays MIDirectIntegrator::configureSampler requests (direct.cpp:58-63): luminaire samples first */
typedef struct { int n; uint32_t size[2]; float *data[2]; } sample_arrays;

static int arrays_init(sample_arrays *A, const orc_render_params *prm, uint32_t spp) {
	memset(A, 0, sizeof(*A));
	if (prm->integrator != 1) return 0;
	if (prm->luminaire_samples > 1) A->size[A->n++] = (uint32_t) prm->luminaire_samples;
	if (prm->bsdf_samples > 1) A->size[A->n++] = (uint32_t) prm->bsdf_samples;
	for (int i = 0; i < A->n; ++i) A->data[i] = (float *) malloc(sizeof(float) * 2 * (size_t) spp * A->size[i]);
	return A->n;
}
static void arrays_free(sample_arrays *A) { for (int i = 0; i < A->n; ++i) free(A->data[i]); }
static void sampler_bind_arrays(sampler_t *s, const sample_arrays *A) {
	s->n_arr = A->n; s->adepth = 0;
	for (int i = 0; i < A->n; ++i) { s->arr[i] = A->data[i]; s->arr_size[i] = A->size[i]; }
}

/* Sampler::generate() of the keyed samplers for one pixel: the per-pixel tables of the table-based ones, then the
 * requested sample arrays (independent.cpp:59-70, ldsampler.cpp:143-158, stratified.cpp:121-141) */
static void sampler_generate_tables(const orc_render_params *prm, uint32_t pixelKey, uint32_t spp, int depth, uint32_t *scr, uint32_t *perm,
                                    sample_arrays *A) {
	uint64_t st;
	if (prm->sampler_kind == MTSGPU_SAMPLER_STRATIFIED_KEYED) st = orc_strat_generate_keyed_tables(prm->seed, pixelKey, spp, depth, perm);
	else if (prm->sampler_kind == MTSGPU_SAMPLER_LD_KEYED) st = orc_ld_generate_keyed_tables(prm->seed, pixelKey, spp, depth, scr, perm);
	else st = orc_keyed_init(prm->seed, pixelKey, 0);
	if (!A || A->n == 0) return;
	if (prm->sampler_kind == MTSGPU_SAMPLER_HALTON || prm->sampler_kind == MTSGPU_SAMPLER_HAMMERSLEY) {
		fprintf(stderr, "oracle: request2DArray() is not supported by QMC samplers! (halton.cpp:102-104)\n");
		abort();
	}
	for (int i = 0; i < A->n; ++i) {
		const size_t n = (size_t) spp * A->size[i];
		if (prm->sampler_kind == MTSGPU_SAMPLER_STRATIFIED_KEYED) orc_latin_hypercube_array(&st, n, A->data[i]);
		else if (prm->sampler_kind == MTSGPU_SAMPLER_LD_KEYED) orc_ld_generate_array(&st, n, A->data[i]);
		else orc_independent_generate_array(&st, n, A->data[i]);
	}
}

static int isqrt_u32(uint32_t v) { uint32_t i = 1; while (i * i < v) ++i; return (int) i; }

static int sampler_kind_of(const orc_render_params *p) {
	switch (p->sampler_kind) {
		case MTSGPU_SAMPLER_LD_KEYED: return 1;
		case MTSGPU_SAMPLER_HALTON: return 4;
		case MTSGPU_SAMPLER_HAMMERSLEY: return 5;
		case MTSGPU_SAMPLER_STRATIFIED_KEYED: return 6;
		default: return 0;
	}
}

static uint32_t effective_spp(const orc_render_params *p) {
	/* ldsampler.cpp:52-57: rounded up to a power of two */
	if (p->sampler_kind == MTSGPU_SAMPLER_LD_KEYED) return round_to_pow2(p->spp);
	if (p->sampler_kind == MTSGPU_SAMPLER_STRATIFIED_KEYED) {      /* stratified.cpp:36-44: the next perfect square */
		uint32_t i = 1;
		while (i * i < p->spp) ++i;
		return i * i;
	}
	return p->spp;
}

void orc_render_rect(const mtsgpu_scene *sc, const mtsgpu_camera *cam, const orc_render_params *prm,
                     int x0, int y0, int x1, int y1, float *film, mtsgpu_stats *stats) {
	(void) orc_prime(0);            /* fill the prime table before any thread needs it */
	const uint32_t spp = effective_spp(prm);
	const int W = cam->width, H = cam->height;
	tabfilter_t filter; tabfilter_box(&filter);
	const int isLD = prm->sampler_kind == MTSGPU_SAMPLER_LD_KEYED || prm->sampler_kind == MTSGPU_SAMPLER_STRATIFIED_KEYED;     /* per-pixel tables */
	const int depth = prm->ld_depth > 0 ? prm->ld_depth : 3;
	uint64_t nClosest = 0, nShadow = 0, nDepth = 0;
#ifdef _OPENMP
	int nthreads = prm->n_threads > 0 ? prm->n_threads : omp_get_max_threads();
#pragma omp parallel num_threads(nthreads) reduction(+:nClosest,nShadow,nDepth)
#endif
	{
		uint32_t *scr = isLD ? (uint32_t *) malloc(sizeof(uint32_t) * 3 * (size_t) depth) : NULL;
		uint32_t *perm = isLD ? (uint32_t *) malloc(sizeof(uint32_t) * 2 * (size_t) depth * spp) : NULL;
		sample_arrays arrays; const int hasArrays = arrays_init(&arrays, prm, spp);
		mtsgpu_stats st; memset(&st, 0, sizeof(st));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1) collapse(2)
#endif
		for (int y = y0; y < y1; ++y) {
			for (int x = x0; x < x1; ++x) {
				const uint32_t pixelKey = (uint32_t) y * (uint32_t) W + (uint32_t) x;
				if (isLD || hasArrays)
					sampler_generate_tables(prm, pixelKey, spp, depth, scr, perm, &arrays);   /* sampler->generate() */
				for (uint32_t j = 0; j < spp; ++j) {
					sampler_t smp; memset(&smp, 0, sizeof(smp));
					sampler_bind_arrays(&smp, &arrays);
					smp.kind = sampler_kind_of(prm);
					smp.stream = orc_keyed_init(prm->seed, pixelKey, 1 + (uint64_t) j);
					smp.depth = depth; smp.spp = spp; smp.index = j; smp.scr = scr; smp.perm = perm; smp.resolution = isqrt_u32(spp);
					float sample[2], lens[2] = { 0, 0 };
					if (cam->aperture_radius > 0.0f && cam->kind == 0) sampler_next2d(&smp, lens);     /* needsLensSample (integrator.cpp:156-157) */
					sampler_next2d(&smp, sample);
					sample[0] += x; sample[1] += y;
					ray_t eyeRay;
					camera_generate_ray(cam, sample, lens, &eyeRay);
					li_result res;
					integrator_li(sc, prm, &eyeRay, &smp, &res, &st);
					nDepth += (uint64_t) res.depth;          /* avgPathLength += rRec.depth (path.cpp:212-213) */
					put_sample(film, W, H, &filter, sample[0], sample[1], res.Li, res.alpha);
				}
			}
		}
		nClosest += st.rays_closest; nShadow += st.rays_shadow;
		free(scr); free(perm); arrays_free(&arrays);
	}
	if (stats) {
		stats->camera_samples += (uint64_t) (x1 - x0) * (uint64_t) (y1 - y0) * spp;
		stats->rays_closest += nClosest; stats->rays_shadow += nShadow;
		stats->path_length_sum += nDepth;
	}
}

/* What a Sampler hands out for camera sample `j` of a pixel: generate() for the pixel (keyed stream where the
 * sampler draws random numbers), then n calls of next1D() (two_d == 0: out[n]) or next2D() (out[2n]).  For halton /
 * hammersley these are the reference's own values (src/tests/test_samplers.cpp:33-78). */
void orc_sampler_values(const orc_render_params *prm, uint32_t pixelKey, uint32_t j, uint32_t n, int two_d, float *out) {
	(void) orc_prime(0);
	const uint32_t spp = effective_spp(prm);
	const int isLD = prm->sampler_kind == MTSGPU_SAMPLER_LD_KEYED || prm->sampler_kind == MTSGPU_SAMPLER_STRATIFIED_KEYED;
	const int depth = prm->ld_depth > 0 ? prm->ld_depth : 3;
	uint32_t *scr = isLD ? (uint32_t *) malloc(sizeof(uint32_t) * 3 * (size_t) depth) : NULL;
	uint32_t *perm = isLD ? (uint32_t *) malloc(sizeof(uint32_t) * 2 * (size_t) depth * spp) : NULL;
	sample_arrays arrays; const int hasArrays = arrays_init(&arrays, prm, spp);
	if (isLD || hasArrays) sampler_generate_tables(prm, pixelKey, spp, depth, scr, perm, &arrays);
	sampler_t smp; memset(&smp, 0, sizeof(smp));
	sampler_bind_arrays(&smp, &arrays);
	smp.kind = sampler_kind_of(prm);
	smp.stream = orc_keyed_init(prm->seed, pixelKey, 1 + (uint64_t) j);
	smp.depth = depth; smp.spp = spp; smp.index = j; smp.scr = scr; smp.perm = perm; smp.resolution = isqrt_u32(spp);
	for (uint32_t i = 0; i < n; ++i) {
		if (two_d) sampler_next2d(&smp, out + 2 * (size_t) i);
		else out[i] = sampler_next1d(&smp);
	}
	free(scr); free(perm); arrays_free(&arrays);
}

void orc_li_samples(const mtsgpu_scene *sc, const mtsgpu_camera *cam, const orc_render_params *prm,
                    const uint32_t *pix_samples, uint32_t n, float *out) {
	(void) orc_prime(0);            /* fill the prime table before any thread needs it */
	const uint32_t spp = effective_spp(prm);
	const int isLD = prm->sampler_kind == MTSGPU_SAMPLER_LD_KEYED || prm->sampler_kind == MTSGPU_SAMPLER_STRATIFIED_KEYED;     /* per-pixel tables */
	const int depth = prm->ld_depth > 0 ? prm->ld_depth : 3;
	uint32_t *scr = isLD ? (uint32_t *) malloc(sizeof(uint32_t) * 3 * (size_t) depth) : NULL;
	uint32_t *perm = isLD ? (uint32_t *) malloc(sizeof(uint32_t) * 2 * (size_t) depth * spp) : NULL;
	uint32_t lastKey = 0xFFFFFFFFu;
	sample_arrays arrays; const int hasArrays = arrays_init(&arrays, prm, spp);
	for (uint32_t i = 0; i < n; ++i) {
		const uint32_t x = pix_samples[3*(size_t)i], y = pix_samples[3*(size_t)i+1], j = pix_samples[3*(size_t)i+2];
		const uint32_t pixelKey = y * (uint32_t) cam->width + x;
		if ((isLD || hasArrays) && pixelKey != lastKey) {
			sampler_generate_tables(prm, pixelKey, spp, depth, scr, perm, &arrays);
			lastKey = pixelKey;
		}
		sampler_t smp; memset(&smp, 0, sizeof(smp));
		sampler_bind_arrays(&smp, &arrays);
		smp.kind = sampler_kind_of(prm);
		smp.stream = orc_keyed_init(prm->seed, pixelKey, 1 + (uint64_t) j);
		smp.depth = depth; smp.spp = spp; smp.index = j; smp.scr = scr; smp.perm = perm; smp.resolution = isqrt_u32(spp);
		float sample[2], lens[2] = { 0, 0 };
		if (cam->aperture_radius > 0.0f && cam->kind == 0) sampler_next2d(&smp, lens);
		sampler_next2d(&smp, sample);
		sample[0] += x; sample[1] += y;
		ray_t eyeRay;
		camera_generate_ray(cam, sample, lens, &eyeRay);
		li_result res;
		integrator_li(sc, prm, &eyeRay, &smp, &res, NULL);
		float *o = out + 8 * (size_t) i;
		o[0] = res.Li[0]; o[1] = res.Li[1]; o[2] = res.Li[2]; o[3] = res.alpha;
		o[4] = sample[0]; o[5] = sample[1]; o[6] = (float) res.depth; o[7] = 0.0f;
	}
	free(scr); free(perm); arrays_free(&arrays);
}

/* The reference's own sequential sampling: one Random (default seed 5489, the
 * un-cloned sampler), scanline order inside the rectangle (integrator.cpp:204-226) */
void orc_render_rect_mt(const mtsgpu_scene *sc, const mtsgpu_camera *cam, const orc_render_params *prm,
                        int kind, int x0, int y0, int x1, int y1, float *film) {
	(void) orc_prime(0);            /* fill the prime table before any thread needs it */
	const int W = cam->width, H = cam->height;
	const int depth = prm->ld_depth > 0 ? prm->ld_depth : 3;
	const uint32_t spp = kind == 1 ? round_to_pow2(prm->spp) : prm->spp;
	tabfilter_t filter; tabfilter_box(&filter);
	orc_random rnd; rnd.mti = ORC_MT_N + 1;
	orc_random_seed(&rnd, prm->seed ? prm->seed : 5489ULL);
	float *t1d = kind == 1 ? (float *) malloc(sizeof(float) * (size_t) depth * spp) : NULL;
	float *t2d = kind == 1 ? (float *) malloc(sizeof(float) * 2 * (size_t) depth * spp) : NULL;
	for (int y = y0; y < y1; ++y) {
		for (int x = x0; x < x1; ++x) {
			if (kind == 1)
				orc_ld_generate_mt(&rnd, spp, depth, t1d, t2d);
			for (uint32_t j = 0; j < spp; ++j) {
				sampler_t smp; memset(&smp, 0, sizeof(smp));
				smp.kind = kind == 1 ? 3 : 2;
				smp.mt = &rnd;
				smp.depth = depth; smp.spp = spp; smp.index = j; smp.t1d = t1d; smp.t2d = t2d;
				float sample[2], lens[2] = { 0, 0 };
				if (cam->aperture_radius > 0.0f && cam->kind == 0) sampler_next2d(&smp, lens);
				sampler_next2d(&smp, sample);
				sample[0] += x; sample[1] += y;
				ray_t eyeRay;
				camera_generate_ray(cam, sample, lens, &eyeRay);
				li_result res;
				integrator_li(sc, prm, &eyeRay, &smp, &res, NULL);
				put_sample(film, W, H, &filter, sample[0], sample[1], res.Li, res.alpha);
			}
		}
	}
	free(t1d); free(t2d);
}

/* ========================================================================== */
/* General reconstruction filters: tiles with borders                         */
/* ========================================================================== */
/* TabulatedFilter::TabulatedFilter (rfilter.cpp:40-69) over BoxFilter::evaluate (box.cpp:42-44)
 * or GaussianFilter::evaluate (gaussian.cpp:62-65, ctor :30-42).  std::exp here is the host libm,
 * as in the reference's configure step (not on the per-sample path). */
/* mitchellNetravali (src/rfilters/mitchell.cpp:60-74 == catmullrom.cpp) */
static float mitchell_netravali(float x, float B, float C) {
	x = fabsf(x);
	float xSquared = x*x, xCubed = xSquared*x;
	if (x < 1) {
		return 1.0f/6.0f * ((12-9*B-6*C)*xCubed + (-18+12*B+6*C) * xSquared + (6-2*B));
	} else if (x < 2) {
		return 1.0f/6.0f * ((-B-6*C)*xCubed + (6*B+30*C) * xSquared + (-12*B-48*C) * x + (8*B + 24*C));
	} else {
		return 0.0f;
	}
}

/* lanczosSinc (src/libcore/util.cpp:664-674); host libm as in the reference */
static float lanczos_sinc_(float t, float tau) {
	t = fabsf(t);
	if (t < ORC_EPS)
		return 1.0f;
	else if (t > 1.0f)
		return 0.0f;
	t *= ORC_PI;
	float sincTerm = sinf(t*tau)/(t*tau);
	float windowTerm = sinf(t)/t;
	return sincTerm * windowTerm;
}

/* ReconstructionFilter::evaluate of the five plugins: box (box.cpp), gaussian (gaussian.cpp:62-65),
 * mitchell (mitchell.cpp:55-58), catmullrom (catmullrom.cpp), wsinc (wsinc.cpp:53-56) */
static float rfilter_evaluate(int kind, float x, float y, float sx, float sy, float p0, float p1, float alpha, float cst) {
	switch (kind) {
		case 1: return fmaxf_((float) 0.0f, expf(-alpha * x * x) - cst) * fmaxf_((float) 0.0f, expf(-alpha * y * y) - cst);
		case 2: case 3: return mitchell_netravali(2.0f * x / sx, p0, p1) * mitchell_netravali(2.0f * y / sy, p0, p1);
		case 4: return lanczos_sinc_(x / sx, p0) * lanczos_sinc_(y / sy, p0);
		default: return 1.0f;
	}
}

/* TabulatedFilter::TabulatedFilter (src/librender/rfilter.cpp:40-69).  kind: 0 box, 1 gaussian (p0 = stddev),
 * 2 mitchell (p0 = B, p1 = C), 3 catmullrom, 4 wsinc (p0 = cycles); half_size <= 0 selects the plugin default */
void orc_tabulate_filter(int kind, float half_size, float p0, float p1, orc_tabfilter *out) {
	float alpha = 0, cst = 0;
	if (kind == 1) {
		if (half_size <= 0) half_size = 2.0f;
		if (p0 <= 0) p0 = 0.5f;
		alpha = 1 / (2*p0*p0);
		out->size_x = out->size_y = half_size;
		cst = expf(-alpha * out->size_x * out->size_x);
	} else if (kind == 2 || kind == 3) {
		if (half_size <= 0) half_size = 2.0f;
		if (kind == 3) { p0 = 0.0f; p1 = 0.5f; }
		else { if (p0 < 0) p0 = 1.0f / 3.0f; if (p1 < 0) p1 = 1.0f / 3.0f; }
		out->size_x = out->size_y = half_size;
	} else if (kind == 4) {
		if (half_size <= 0) half_size = 3.0f;
		if (p0 <= 0) p0 = 3.0f;
		out->size_x = out->size_y = half_size;
	} else {
		out->size_x = out->size_y = 0.5f;
	}
	float sum = 0;
	for (int y = 0; y < FILTER_RESOLUTION+1; ++y) {
		float yPos = (y + 0.5f) / FILTER_RESOLUTION * out->size_y;
		for (int x = 0; x < FILTER_RESOLUTION+1; ++x) {
			if (x == FILTER_RESOLUTION || y == FILTER_RESOLUTION) {
				out->values[y][x] = 0;
			} else {
				float xPos = (x + 0.5f) / FILTER_RESOLUTION * out->size_x;
				out->values[y][x] = rfilter_evaluate(kind, xPos, yPos, out->size_x, out->size_y, p0, p1, alpha, cst);
			}
			sum += out->values[y][x];
		}
	}
	sum *= 4*out->size_x*out->size_y / (FILTER_RESOLUTION*FILTER_RESOLUTION);
	for (int y = 0; y < FILTER_RESOLUTION+1; ++y)
		for (int x = 0; x < FILTER_RESOLUTION+1; ++x)
			out->values[y][x] /= sum;
}

typedef struct { float L[3], alpha, sx, sy; int valid; } tsample_t;

/* which part renders tile (tx, ty): bits of tx and ty interleaved, tx lowest (mtsgpu_set_tiles, include/mtsgpu.h) */
static uint32_t tile_morton(uint32_t tx, uint32_t ty) {
	uint32_t m = 0;
	for (int b = 0; b < 16; ++b)
		m |= ((tx >> b) & 1u) << (2 * b) | ((ty >> b) & 1u) << (2 * b + 1);
	return m;
}

void orc_render_tiles(const mtsgpu_scene *sc, const mtsgpu_camera *cam, const orc_render_params *prm,
                      const orc_tabfilter *filter, int bs, int part, int n_parts, int hq_edges,
                      float *film, mtsgpu_stats *stats) {
	(void) orc_prime(0);            /* fill the prime table before any thread needs it */
	const uint32_t spp = effective_spp(prm);
	const int W = cam->width, H = cam->height;
	const int isLD = prm->sampler_kind == MTSGPU_SAMPLER_LD_KEYED || prm->sampler_kind == MTSGPU_SAMPLER_STRATIFIED_KEYED;     /* per-pixel tables */
	const int depth = prm->ld_depth > 0 ? prm->ld_depth : 3;
	/* m_borderSize (renderproc.cpp:143-144) */
	const int border = (int) ceilf(fmaxf_(filter->size_x, filter->size_y) - (float) 0.5);
	/* Film::hasHighQualityEdges: the rendered rectangle grows by the border on every side
	   (renderproc.cpp:146-153); samples outside the film still reach the pixels inside */
	const int off = hq_edges ? -border : 0;
	const int RW = W - 2 * off, RH = H - 2 * off;      /* size of the rendered rectangle */
	const int tx = (RW + bs - 1) / bs, ty = (RH + bs - 1) / bs, nTiles = tx * ty;
	const int full = bs + 2 * border;
	const float factorX = FILTER_RESOLUTION / filter->size_x, factorY = FILTER_RESOLUTION / filter->size_y;
	const int RX = (int) ceilf(filter->size_x + 0.5f), RY = (int) ceilf(filter->size_y + 0.5f);
	float **blocks = (float **) calloc((size_t) nTiles, sizeof(float *));
	uint64_t nClosest = 0, nShadow = 0, nSamples = 0;
#ifdef _OPENMP
	int nthreads = prm->n_threads > 0 ? prm->n_threads : omp_get_max_threads();
#pragma omp parallel num_threads(nthreads) reduction(+:nClosest,nShadow,nSamples)
#endif
	{
		uint32_t *scr = isLD ? (uint32_t *) malloc(sizeof(uint32_t) * 3 * (size_t) depth) : NULL;
		uint32_t *perm = isLD ? (uint32_t *) malloc(sizeof(uint32_t) * 2 * (size_t) depth * spp) : NULL;
		tsample_t *smp = (tsample_t *) malloc(sizeof(tsample_t) * (size_t) bs * bs * spp);
		sample_arrays arrays; const int hasArrays = arrays_init(&arrays, prm, spp);
		mtsgpu_stats st; memset(&st, 0, sizeof(st));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
		for (int t = 0; t < nTiles; ++t) {
			if (tile_morton((uint32_t) (t % tx), (uint32_t) (t / tx)) % (uint32_t) n_parts != (uint32_t) part) continue;   /* mtsgpu_set_tiles */
			const int x0 = off + (t % tx) * bs, y0 = off + (t / tx) * bs;
			const int w = (x0 + bs <= off + RW ? bs : off + RW - x0), h = (y0 + bs <= off + RH ? bs : off + RH - y0);
			/* 1. the camera samples of the tile (integrator.cpp:150-169) */
			for (int py = 0; py < h; ++py) for (int px = 0; px < w; ++px) {
				/* sampler key = index of the pixel inside the rendered rectangle */
				const uint32_t pixelKey = (uint32_t) (y0 + py - off) * (uint32_t) RW + (uint32_t) (x0 + px - off);
				if (isLD || hasArrays) sampler_generate_tables(prm, pixelKey, spp, depth, scr, perm, &arrays);
				for (uint32_t j = 0; j < spp; ++j) {
					sampler_t s; memset(&s, 0, sizeof(s));
					sampler_bind_arrays(&s, &arrays);
					s.kind = sampler_kind_of(prm);
					s.stream = orc_keyed_init(prm->seed, pixelKey, 1 + (uint64_t) j);
					s.depth = depth; s.spp = spp; s.index = j; s.scr = scr; s.perm = perm; s.resolution = isqrt_u32(spp);
					float sample[2], lens[2] = { 0, 0 };
					if (cam->aperture_radius > 0.0f && cam->kind == 0) sampler_next2d(&s, lens);
					sampler_next2d(&s, sample);
					sample[0] += x0 + px; sample[1] += y0 + py;
					ray_t eyeRay;
					camera_generate_ray(cam, sample, lens, &eyeRay);
					li_result res;
					integrator_li(sc, prm, &eyeRay, &s, &res, &st);
					tsample_t *o = &smp[((size_t) py * w + px) * spp + j];
					o->L[0] = res.Li[0]; o->L[1] = res.Li[1]; o->L[2] = res.Li[2]; o->alpha = res.alpha;
					o->sx = sample[0]; o->sy = sample[1];
					o->valid = 1;
					for (int c = 0; c < 3; ++c) if (res.Li[c] != res.Li[c] || res.Li[c] < 0.0f) o->valid = 0;
				}
			}
			nSamples += (uint64_t) w * h * spp;
			/* 2. ImageBlock::putSample for every block pixel, gathered in a fixed order */
			float *blk = (float *) calloc((size_t) full * full * 5, sizeof(float));
			const int fullW = w + 2 * border, fullH = h + 2 * border;
			const float offX = (float) (x0 - border), offY = (float) (y0 - border);
			for (int yl = 0; yl < fullH; ++yl) for (int xl = 0; xl < fullW; ++xl) {
				const int X = x0 - border + xl, Y = y0 - border + yl;
				if (X < 0 || X >= W || Y < 0 || Y >= H) continue;     /* dropped by Film::putImageBlock */
				float acc[5] = { 0, 0, 0, 0, 0 };
				for (int py = (Y - RY > y0 ? Y - RY : y0); py <= (Y + RY < y0 + h - 1 ? Y + RY : y0 + h - 1); ++py)
				for (int px = (X - RX > x0 ? X - RX : x0); px <= (X + RX < x0 + w - 1 ? X + RX : x0 + w - 1); ++px) {
					const tsample_t *ps = &smp[((size_t) (py - y0) * w + (px - x0)) * spp];
					for (uint32_t j = 0; j < spp; ++j) {
						const tsample_t *q = &ps[j];
						if (!q->valid) continue;
						const float slx = q->sx - 0.5f - offX, sly = q->sy - 0.5f - offY;
						int xStart = (int) ceilf(slx - filter->size_x), xEnd = (int) floorf(slx + filter->size_x);
						int yStart = (int) ceilf(sly - filter->size_y), yEnd = (int) floorf(sly + filter->size_y);
						if (xStart < 0) xStart = 0;
						if (yStart < 0) yStart = 0;
						if (xEnd > fullW-1) xEnd = fullW-1;
						if (yEnd > fullH-1) yEnd = fullH-1;
						if (xl < xStart || xl > xEnd || yl < yStart || yl > yEnd) continue;
						int ix = (int) (factorX * fabsf(xl - slx)); if (ix > FILTER_RESOLUTION) ix = FILTER_RESOLUTION;
						int iy = (int) (factorY * fabsf(yl - sly)); if (iy > FILTER_RESOLUTION) iy = FILTER_RESOLUTION;
						const float weight = filter->values[iy][ix];
						if (weight == 0.0f) continue;
						acc[0] += q->L[0] * weight; acc[1] += q->L[1] * weight; acc[2] += q->L[2] * weight;
						acc[3] += q->alpha * weight; acc[4] += weight;
					}
				}
				float *o = blk + 5 * ((size_t) yl * full + xl);
				for (int c = 0; c < 5; ++c) o[c] = acc[c];
			}
			blocks[t] = blk;
		}
		nClosest += st.rays_closest; nShadow += st.rays_shadow;
		free(scr); free(perm); free(smp); arrays_free(&arrays);
	}
	/* 3. Film::putImageBlock in colour order */
	for (int colour = 0; colour < 4; ++colour)
		for (int t = 0; t < nTiles; ++t) {
			if (!blocks[t] || (((t % tx) & 1) + 2 * ((t / tx) & 1)) != colour) continue;
			const int x0 = off + (t % tx) * bs, y0 = off + (t / tx) * bs;
			const int w = (x0 + bs <= off + RW ? bs : off + RW - x0), h = (y0 + bs <= off + RH ? bs : off + RH - y0);
			for (int yl = 0; yl < h + 2 * border; ++yl) for (int xl = 0; xl < w + 2 * border; ++xl) {
				const int X = x0 - border + xl, Y = y0 - border + yl;
				if (X < 0 || X >= W || Y < 0 || Y >= H) continue;
				const float *b = blocks[t] + 5 * ((size_t) yl * full + xl);
				float *o = film + 5 * ((size_t) Y * W + X);
				for (int c = 0; c < 5; ++c) o[c] += b[c];
			}
		}
	for (int t = 0; t < nTiles; ++t) free(blocks[t]);
	free(blocks);
	if (stats) { stats->camera_samples += nSamples; stats->rays_closest += nClosest; stats->rays_shadow += nShadow; }
}
