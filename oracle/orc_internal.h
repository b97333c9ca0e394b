/* orc_internal.h -- private helpers of the oracle (TEST INFRASTRUCTURE ONLY). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "mts_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* include/mitsuba/core/constants.h:31-50 (single precision) */
#define ORC_EPS        1e-4f
#define ORC_SHADOW_EPS 1e-3f
#define ORC_PI         3.14159265358979323846f
#define ORC_INV_PI     0.31830988618379067154f

/* std::max / std::min semantics (second argument wins only on strict compare) */
static inline float fmaxf_(float a, float b) { return (a < b) ? b : a; }
static inline float fminf_(float a, float b) { return (b < a) ? b : a; }

static inline void v3_sub(float r[3], const float a[3], const float b[3]) {
	r[0] = a[0]-b[0]; r[1] = a[1]-b[1]; r[2] = a[2]-b[2];
}
static inline void v3_add(float r[3], const float a[3], const float b[3]) {
	r[0] = a[0]+b[0]; r[1] = a[1]+b[1]; r[2] = a[2]+b[2];
}
static inline void v3_scale(float r[3], const float a[3], float s) {
	r[0] = a[0]*s; r[1] = a[1]*s; r[2] = a[2]*s;
}
/* include/mitsuba/core/vector.h dot: x*x' + y*y' + z*z' left to right */
static inline float v3_dot(const float a[3], const float b[3]) {
	return a[0]*b[0] + a[1]*b[1] + a[2]*b[2];
}
/* vector.h cross */
static inline void v3_cross(float r[3], const float a[3], const float b[3]) {
	float x = (a[1] * b[2]) - (a[2] * b[1]);
	float y = (a[2] * b[0]) - (a[0] * b[2]);
	float z = (a[0] * b[1]) - (a[1] * b[0]);
	r[0] = x; r[1] = y; r[2] = z;
}
static inline float v3_length(const float a[3]) {
	return sqrtf(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]);
}
/* vector.h:312-330: v / f == v * (1/f);  normalize(v) = v / v.length() (:403-404) */
static inline void v3_div(float r[3], const float a[3], float f) {
	float recip = 1.0f / f;
	r[0] = a[0]*recip; r[1] = a[1]*recip; r[2] = a[2]*recip;
}
static inline void v3_normalize(float r[3], const float a[3]) {
	v3_div(r, a, v3_length(a));
}

float    orc_ulong_to_float(uint64_t v);
uint64_t orc_size_bitmask(uint64_t n);
uint64_t orc_ld_generate_keyed_tables(uint64_t seed, uint32_t pixel_key, uint32_t spp, int depth,
                                      uint32_t *scr, uint32_t *perm);
uint64_t orc_strat_generate_keyed_tables(uint64_t seed, uint32_t pixel_key, uint32_t spp, int depth, uint32_t *perm);
void     orc_independent_generate_array(uint64_t *st, size_t n, float *out);
void     orc_ld_generate_array(uint64_t *st, size_t n, float *out);
void     orc_latin_hypercube_array(uint64_t *st, size_t n, float *out);

/* kd-tree builder (orc_kdtree.c) */
typedef struct orc_kdtree {
	uint32_t n_nodes, n_indices;
	uint32_t *nodes;    /* [n_nodes][2] */
	uint32_t *indices;
	float aabb_min[3], aabb_max[3];         /* enlarged */
	float tight_min[3], tight_max[3];
	double stats[6];
} orc_kdtree;

int  orc_kd_build(const float *vtx_pos, const uint32_t *tri_idx, uint32_t n_tris, const float *gen_aabb,
                  const mtsgpu_kd_params *params, orc_kdtree *out);
void orc_kd_free(orc_kdtree *t);

#endif
