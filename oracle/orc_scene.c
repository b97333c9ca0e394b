/*
 * orc_scene.c -- oracle (TEST INFRASTRUCTURE ONLY): what Scene::initialize,
 * TriMesh::configure and PerspectiveCameraImpl::configure compute on the host,
 * restated in plain C and written into the flat mtsgpu_scene layout.
 */
#include "orc_internal.h"

struct orc_flat_scene {
	mtsgpu_scene sc;
	orc_kdtree kd;
	float *vtx_pos, *vtx_nrm;
	uint32_t *tri_idx, *shape_tri_offset, *shape_flags, *triaccel;
	int32_t *shape_bsdf, *shape_lum;
	uint32_t *shape_type; float *shape_params;
	uint32_t *bsdf_type; float *bsdf_params;
	uint32_t *lum_type; float *lum_params; int32_t *lum_shape; float *lum_inv_area;
	uint32_t *lum_cdf_offset; float *lum_tri_cdf, *lum_sel_cdf, *lum_sel_pdf;
	float *env_pixels, *env_pdf, *env_cdf;
};

/* unitAngle (include/mitsuba/core/util.h:343-348) */
static float unit_angle(const float u[3], const float v[3]) {
	float t[3];
	if (v3_dot(u, v) < 0) {
		v3_add(t, v, u);
		return ORC_PI - 2 * asinf(v3_length(t) / 2);
	} else {
		v3_sub(t, v, u);
		return 2 * asinf(v3_length(t) / 2);
	}
}

/* TriMesh::computeNormals, smooth branch (src/librender/trimesh.cpp:497-539) */
static void compute_normals(const float *pos, uint32_t nVerts, const uint32_t *tris, uint32_t nTris, float *nrm) {
	memset(nrm, 0, sizeof(float) * 3 * (size_t) nVerts);
	for (uint32_t i = 0; i < nTris; i++) {
		const uint32_t *tri = tris + 3 * (size_t) i;
		float n[3] = { 0.0f, 0.0f, 0.0f };
		for (int k = 0; k < 3; ++k) {
			const float *v0 = pos + 3 * (size_t) tri[k];
			const float *v1 = pos + 3 * (size_t) tri[(k+1)%3];
			const float *v2 = pos + 3 * (size_t) tri[(k+2)%3];
			float sideA[3], sideB[3];
			v3_sub(sideA, v1, v0); v3_sub(sideB, v2, v0);
			if (k == 0) {
				v3_cross(n, sideA, sideB);
				float length = v3_length(n);
				if (length == 0)
					break;
				v3_div(n, n, length);
			}
			float a[3], b[3];
			v3_normalize(a, sideA); v3_normalize(b, sideB);
			float angle = unit_angle(a, b);
			float *dst = nrm + 3 * (size_t) tri[k];
			dst[0] += n[0] * angle; dst[1] += n[1] * angle; dst[2] += n[2] * angle;
		}
	}
	for (uint32_t i = 0; i < nVerts; i++) {
		float *n = nrm + 3 * (size_t) i;
		float length = v3_length(n);
		if (length != 0) {
			v3_div(n, n, length);
		} else {
			n[0] = 1; n[1] = 0; n[2] = 0;
		}
	}
}

/* Triangle::surfaceArea (src/libcore/triangle.cpp:49-55) */
static float tri_surface_area(const float *pos, const uint32_t *tri) {
	float sideA[3], sideB[3], c[3];
	v3_sub(sideA, pos + 3 * (size_t) tri[1], pos + 3 * (size_t) tri[0]);
	v3_sub(sideB, pos + 3 * (size_t) tri[2], pos + 3 * (size_t) tri[0]);
	v3_cross(c, sideA, sideB);
	return 0.5f * v3_length(c);
}

/* DiscretePDF::build (include/mitsuba/core/pdf.h:82-95); pdf may be NULL */
static float discrete_pdf_build(const float *values, uint32_t n, float *cdf, float *pdf) {
	cdf[0] = 0.0f;
	for (uint32_t i = 1; i < n + 1; ++i)
		cdf[i] = cdf[i-1] + values[i-1];
	float originalSum = cdf[n];
	for (uint32_t i = 0; i < n; ++i) {
		cdf[i] /= originalSum;
		if (pdf) pdf[i] = values[i] / originalSum;
	}
	cdf[n] = 1.0f;
	return originalSum;
}

static int build_envmap(orc_flat_scene *fs, const mtsgpu_scene_desc *d, float *P);

int orc_flatten(const mtsgpu_scene_desc *d, const mtsgpu_kd_params *kdp, orc_flat_scene **out) {
	orc_flat_scene *fs = (orc_flat_scene *) calloc(1, sizeof(orc_flat_scene));
	uint32_t nShapes = d->n_meshes, nVerts = 0, nTris = 0;
	/* m_shapeMap (skdtree.cpp:43-60): a TriMesh contributes its triangles, any other shape ONE primitive */
	for (uint32_t s = 0; s < nShapes; ++s) {
		const int sphere = d->meshes[s].shape_type == MTSGPU_SHAPE_SPHERE;
		nVerts += sphere ? 0 : d->meshes[s].n_verts; nTris += sphere ? 1 : d->meshes[s].n_tris;
	}
	float *genAABB = (float *) calloc(6 * ((size_t) nTris + 1), sizeof(float));
	fs->shape_type = (uint32_t *) calloc(nShapes + 1, sizeof(uint32_t));
	fs->shape_params = (float *) calloc(MTSGPU_SHAPE_NPARAMS * ((size_t) nShapes + 1), sizeof(float));
	fs->vtx_pos = (float *) malloc(sizeof(float) * 3 * ((size_t) nVerts + 1));
	fs->vtx_nrm = (float *) calloc(3 * ((size_t) nVerts + 1), sizeof(float));
	fs->tri_idx = (uint32_t *) malloc(sizeof(uint32_t) * 3 * ((size_t) nTris + 1));
	fs->shape_tri_offset = (uint32_t *) malloc(sizeof(uint32_t) * (nShapes + 1));
	fs->shape_flags = (uint32_t *) calloc(nShapes + 1, sizeof(uint32_t));
	fs->shape_bsdf = (int32_t *) malloc(sizeof(int32_t) * (nShapes + 1));
	fs->shape_lum = (int32_t *) malloc(sizeof(int32_t) * (nShapes + 1));
	fs->triaccel = (uint32_t *) calloc(12 * ((size_t) nTris + 1), sizeof(uint32_t));

	fs->bsdf_type = (uint32_t *) malloc(sizeof(uint32_t) * (d->n_bsdfs + 1));
	fs->bsdf_params = (float *) malloc(sizeof(float) * MTSGPU_BSDF_NPARAMS * (d->n_bsdfs + 1));
	memcpy(fs->bsdf_type, d->bsdf_type, sizeof(uint32_t) * d->n_bsdfs);
	memcpy(fs->bsdf_params, d->bsdf_params, sizeof(float) * MTSGPU_BSDF_NPARAMS * d->n_bsdfs);

	uint32_t nLums = d->n_lums;
	fs->lum_type = (uint32_t *) malloc(sizeof(uint32_t) * (nLums + 1));
	fs->lum_params = (float *) calloc(MTSGPU_LUM_NPARAMS * (nLums + 1), sizeof(float));
	fs->lum_shape = (int32_t *) malloc(sizeof(int32_t) * (nLums + 1));
	fs->lum_inv_area = (float *) calloc(nLums + 1, sizeof(float));
	fs->lum_cdf_offset = (uint32_t *) calloc(nLums + 2, sizeof(uint32_t));
	fs->lum_sel_cdf = (float *) calloc(nLums + 2, sizeof(float));
	fs->lum_sel_pdf = (float *) calloc(nLums + 1, sizeof(float));
	memcpy(fs->lum_type, d->lum_type, sizeof(uint32_t) * nLums);
	memcpy(fs->lum_params, d->lum_params, sizeof(float) * MTSGPU_LUM_NPARAMS * nLums);
	for (uint32_t l = 0; l < nLums; ++l) fs->lum_shape[l] = -1;

	/* geometry: ShapeKDTree::addShape order, m_shapeMap prefix sums (skdtree.cpp:43-65) */
	uint32_t vbase = 0, tbase = 0;
	for (uint32_t s = 0; s < nShapes; ++s) {
		const mtsgpu_mesh *m = &d->meshes[s];
		fs->shape_tri_offset[s] = tbase;
		fs->shape_bsdf[s] = m->bsdf;
		fs->shape_lum[s] = m->lum;
		if (m->lum >= 0 && (uint32_t) m->lum < nLums)
			fs->lum_shape[m->lum] = (int32_t) s;
		if (m->shape_type == MTSGPU_SHAPE_SPHERE) {
			/* Sphere::Sphere with `center` + `radius` (src/shapes/sphere.cpp:44-60): objectToWorld is a translation */
			float *P = fs->shape_params + MTSGPU_SHAPE_NPARAMS * (size_t) s;
			fs->shape_type[s] = MTSGPU_SHAPE_SPHERE;
			const float r = m->sphere_radius;
			for (int i = 0; i < 3; ++i) P[i] = 0.0f * 0.0f + 0.0f * 0.0f + 0.0f * 0.0f + m->sphere_center[i];    /* m_objectToWorld(Point(0,0,0)) */
			P[3] = r; P[4] = m->sphere_inverted ? 1.0f : 0.0f;
			P[5] = P[9] = P[13] = 1.0f;  P[14] = P[18] = P[22] = 1.0f;
			P[23] = 1 / (4 * ORC_PI * r * r);
			/* Sphere::getAABB (sphere.cpp:82-88) */
			const float absRadius = fabsf(r);
			for (int i = 0; i < 3; ++i) { genAABB[6 * (size_t) tbase + i] = P[i] - absRadius; genAABB[6 * (size_t) tbase + 3 + i] = P[i] + absRadius; }
			for (int k = 0; k < 3; ++k) fs->tri_idx[3 * (size_t) tbase + k] = MTSGPU_KNOTRIANGLE;
			tbase += 1;
			continue;
		}
		memcpy(fs->vtx_pos + 3 * (size_t) vbase, m->positions, sizeof(float) * 3 * (size_t) m->n_verts);
		if (!m->face_normals) {
			fs->shape_flags[s] |= MTSGPU_SHAPE_HAS_NORMALS;
			if (m->normals)
				memcpy(fs->vtx_nrm + 3 * (size_t) vbase, m->normals, sizeof(float) * 3 * (size_t) m->n_verts);
			else
				compute_normals(m->positions, m->n_verts, m->triangles, m->n_tris, fs->vtx_nrm + 3 * (size_t) vbase);
		}
		for (uint32_t t = 0; t < m->n_tris; ++t)
			for (int k = 0; k < 3; ++k)
				fs->tri_idx[3 * ((size_t) tbase + t) + k] = m->triangles[3 * (size_t) t + k] + vbase;
		vbase += m->n_verts; tbase += m->n_tris;
	}
	fs->shape_tri_offset[nShapes] = tbase;

	/* kd-tree over all triangles + TriAccel table (skdtree.cpp:62-101) */
	orc_kd_build(fs->vtx_pos, fs->tri_idx, nTris, genAABB, kdp, &fs->kd);
	free(genAABB);
	for (uint32_t s = 0; s < nShapes; ++s) {
		for (uint32_t t = fs->shape_tri_offset[s]; t < fs->shape_tri_offset[s+1]; ++t) {
			const uint32_t *tri = fs->tri_idx + 3 * (size_t) t;
			uint32_t *ta = fs->triaccel + 12 * (size_t) t;
			if (fs->shape_type[s] != MTSGPU_SHAPE_TRIMESH) {
				/* a 'fake' triangle which redirects to the Shape (skdtree.cpp:92-96) */
				memset(ta, 0, 48);
				ta[10] = s; ta[0] = MTSGPU_KNOTRIANGLE;
				continue;
			}
			orc_triaccel_load(fs->vtx_pos + 3 * (size_t) tri[0], fs->vtx_pos + 3 * (size_t) tri[1],
			                  fs->vtx_pos + 3 * (size_t) tri[2], ta);
			ta[10] = s;
			ta[11] = t - fs->shape_tri_offset[s];
		}
	}

	/* scene bounding sphere: AABB::getBSphere of the enlarged box (aabb.cpp:44-47) */
	float center[3], tmp[3];
	v3_add(tmp, fs->kd.aabb_max, fs->kd.aabb_min);
	v3_scale(center, tmp, 0.5f);
	v3_sub(tmp, center, fs->kd.aabb_max);
	float radius = v3_length(tmp);

	/* luminaires */
	uint32_t cdfTotal = 0;
	for (uint32_t l = 0; l < nLums; ++l) {
		fs->lum_cdf_offset[l] = cdfTotal;
		if (fs->lum_type[l] == MTSGPU_LUM_AREA && fs->lum_shape[l] >= 0 && fs->shape_type[fs->lum_shape[l]] == MTSGPU_SHAPE_TRIMESH) {
			uint32_t s = (uint32_t) fs->lum_shape[l];
			cdfTotal += fs->shape_tri_offset[s+1] - fs->shape_tri_offset[s] + 1;
		}
	}
	fs->lum_cdf_offset[nLums] = cdfTotal;
	fs->lum_tri_cdf = (float *) calloc((size_t) cdfTotal + 1, sizeof(float));
	int32_t background = -1;
	for (uint32_t l = 0; l < nLums; ++l) {
		float *P = fs->lum_params + MTSGPU_LUM_NPARAMS * (size_t) l;
		if (fs->lum_type[l] == MTSGPU_LUM_AREA) {
			if (fs->lum_shape[l] < 0) { orc_flat_scene_free(fs); return MTSGPU_EINVAL; }
			uint32_t s = (uint32_t) fs->lum_shape[l];
			if (fs->shape_type[s] == MTSGPU_SHAPE_SPHERE) {
				fs->lum_inv_area[l] = fs->shape_params[MTSGPU_SHAPE_NPARAMS * (size_t) s + 23];    /* m_invSurfaceArea */
				continue;
			}
			/* TriMesh::configure (trimesh.cpp:279-283) */
			uint32_t t0 = fs->shape_tri_offset[s], n = fs->shape_tri_offset[s+1] - t0;
			float *areas = (float *) malloc(sizeof(float) * ((size_t) n + 1));
			for (uint32_t t = 0; t < n; ++t)
				areas[t] = tri_surface_area(fs->vtx_pos, fs->tri_idx + 3 * ((size_t) t0 + t));
			float surfaceArea = discrete_pdf_build(areas, n, fs->lum_tri_cdf + fs->lum_cdf_offset[l], NULL);
			fs->lum_inv_area[l] = 1.0f / surfaceArea;
			free(areas);
		} else if (fs->lum_type[l] == MTSGPU_LUM_CONSTANT) {
			/* ConstantLuminaire::preprocess (src/luminaires/constant.cpp:49-63) */
			float bc[3] = { center[0], center[1], center[2] }, br = radius;
			br *= 1.01f;
			if (d->has_camera) {
				float oldr = br, dv[3];
				v3_sub(dv, d->camera_pos, bc);
				br = fmaxf_(br, v3_length(dv));
				if (oldr != br)
					br *= 1.01f;
			}
			P[3] = bc[0]; P[4] = bc[1]; P[5] = bc[2]; P[6] = br;
			background = (int32_t) l;
		} else if (fs->lum_type[l] == MTSGPU_LUM_ENVMAP) {
			/* EnvMapLuminaire::preprocess (src/luminaires/envmap.cpp:112-126): same bounding sphere logic */
			float bc[3] = { center[0], center[1], center[2] }, br = radius;
			br *= 1.01f;
			if (d->has_camera) {
				float oldr = br, dv[3];
				v3_sub(dv, d->camera_pos, bc);
				br = fmaxf_(br, v3_length(dv));
				if (oldr != br)
					br *= 1.01f;
			}
			P[3] = bc[0]; P[4] = bc[1]; P[5] = bc[2]; P[6] = br;
			if (background >= 0 || build_envmap(fs, d, P) != 0) { orc_flat_scene_free(fs); return MTSGPU_EINVAL; }
			background = (int32_t) l;
		} else if (fs->lum_type[l] == MTSGPU_LUM_DIRECTIONAL) {
			/* DirectionalLuminaire::preprocess (directional.cpp:65-72): m_diskRadius = scene bsphere radius */
			P[6] = radius;
		} else if (fs->lum_type[l] == MTSGPU_LUM_SPOT) {
			/* SpotLuminaire::configure (spot.cpp:56-62): host libm, as the reference */
			P[6] = cosf(P[19]);
			P[7] = cosf(P[8]);
			P[9] = 1.0f / (P[8] - P[19]);
		}
	}
	/* Scene::initialize luminaire PDF (scene.cpp:320-330): weight 1.0 each */
	if (nLums > 0) {
		float *w = (float *) malloc(sizeof(float) * nLums);
		for (uint32_t l = 0; l < nLums; ++l) w[l] = 1.0f;
		fs->sc.lum_sel_sum = discrete_pdf_build(w, nLums, fs->lum_sel_cdf, fs->lum_sel_pdf);
		free(w);
	}

	mtsgpu_scene *sc = &fs->sc;
	sc->abi_version = MTSGPU_ABI_VERSION;
	sc->n_shapes = nShapes; sc->n_tris = nTris; sc->n_verts = nVerts;
	sc->vtx_pos = fs->vtx_pos; sc->vtx_nrm = fs->vtx_nrm; sc->tri_idx = fs->tri_idx;
	sc->shape_tri_offset = fs->shape_tri_offset; sc->shape_bsdf = fs->shape_bsdf;
	sc->shape_lum = fs->shape_lum; sc->shape_flags = fs->shape_flags;
	sc->shape_type = fs->shape_type; sc->shape_params = fs->shape_params;
	sc->n_nodes = fs->kd.n_nodes; sc->n_indices = fs->kd.n_indices;
	sc->kd_nodes = fs->kd.nodes; sc->kd_indices = fs->kd.indices; sc->triaccel = fs->triaccel;
	for (int a = 0; a < 3; ++a) { sc->aabb_min[a] = fs->kd.aabb_min[a]; sc->aabb_max[a] = fs->kd.aabb_max[a]; }
	sc->n_bsdfs = d->n_bsdfs; sc->bsdf_type = fs->bsdf_type; sc->bsdf_params = fs->bsdf_params;
	sc->n_lums = nLums; sc->lum_type = fs->lum_type; sc->lum_params = fs->lum_params;
	sc->lum_shape = fs->lum_shape; sc->lum_inv_area = fs->lum_inv_area;
	sc->lum_cdf_offset = fs->lum_cdf_offset; sc->lum_tri_cdf = fs->lum_tri_cdf;
	sc->lum_sel_cdf = fs->lum_sel_cdf; sc->lum_sel_pdf = fs->lum_sel_pdf;
	sc->background_lum = background;
	sc->env_pixels = fs->env_pixels; sc->env_pdf = fs->env_pdf; sc->env_cdf = fs->env_cdf;     /* sizes set by build_envmap */
	*out = fs;
	return 0;
}

const mtsgpu_scene *orc_flat_scene_get(const orc_flat_scene *fs) { return &fs->sc; }

int orc_flat_scene_kdstats(const orc_flat_scene *fs, double *out6) {
	for (int i = 0; i < 6; ++i) out6[i] = fs->kd.stats[i];
	return 0;
}

void orc_flat_scene_free(orc_flat_scene *fs) {
	if (!fs) return;
	orc_kd_free(&fs->kd);
	free(fs->vtx_pos); free(fs->vtx_nrm); free(fs->tri_idx); free(fs->shape_tri_offset);
	free(fs->shape_flags); free(fs->shape_bsdf); free(fs->shape_lum); free(fs->triaccel);
	free(fs->shape_type); free(fs->shape_params);
	free(fs->env_pixels); free(fs->env_pdf); free(fs->env_cdf);
	free(fs->bsdf_type); free(fs->bsdf_params); free(fs->lum_type); free(fs->lum_params);
	free(fs->lum_shape); free(fs->lum_inv_area); free(fs->lum_cdf_offset); free(fs->lum_tri_cdf);
	free(fs->lum_sel_cdf); free(fs->lum_sel_pdf);
	free(fs);
}

/* ========================================================================== */
/* Camera: Transform algebra (src/libcore/transform.cpp, matrix.inl:140-190)  */
/* ========================================================================== */
typedef struct { float m[4][4], inv[4][4]; } xform_t;

static void mat_mul(float r[4][4], const float a[4][4], const float b[4][4]) {
	float t[4][4];
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 4; ++j) {
			float sum = 0;
			for (int k = 0; k < 4; ++k)
				sum += a[i][k] * b[k][j];
			t[i][j] = sum;
		}
	memcpy(r, t, sizeof(t));
}

/* Matrix::invert, Gauss-Jordan with full pivoting (matrix.inl:140-190) */
static int mat_invert(const float src[4][4], float target[4][4]) {
	int indxc[4], indxr[4], ipiv[4] = { 0, 0, 0, 0 };
	memcpy(target, src, sizeof(float) * 16);
	for (int i = 0; i < 4; i++) {
		int irow = -1, icol = -1;
		float big = 0;
		for (int j = 0; j < 4; j++) {
			if (ipiv[j] != 1) {
				for (int k = 0; k < 4; k++) {
					if (ipiv[k] == 0) {
						if (fabsf(target[j][k]) >= big) {
							big = fabsf(target[j][k]);
							irow = j; icol = k;
						}
					} else if (ipiv[k] > 1) {
						return 0;
					}
				}
			}
		}
		++ipiv[icol];
		if (irow != icol)
			for (int k = 0; k < 4; ++k) { float t = target[irow][k]; target[irow][k] = target[icol][k]; target[icol][k] = t; }
		indxr[i] = irow; indxc[i] = icol;
		if (target[icol][icol] == 0)
			return 0;
		float pivinv = 1.f / target[icol][icol];
		target[icol][icol] = 1.f;
		for (int j = 0; j < 4; j++)
			target[icol][j] *= pivinv;
		for (int j = 0; j < 4; j++) {
			if (j != icol) {
				float save = target[j][icol];
				target[j][icol] = 0;
				for (int k = 0; k < 4; k++)
					target[j][k] -= target[icol][k]*save;
			}
		}
	}
	for (int j = 3; j >= 0; j--) {
		if (indxr[j] != indxc[j])
			for (int k = 0; k < 4; k++) { float t = target[k][indxr[j]]; target[k][indxr[j]] = target[k][indxc[j]]; target[k][indxc[j]] = t; }
	}
	return 1;
}

static void xf_from_matrix(xform_t *x, const float m[4][4]) { memcpy(x->m, m, sizeof(x->m)); mat_invert(m, x->inv); }
/* Transform::operator* (transform.cpp:28-31) */
static void xf_mul(xform_t *r, const xform_t *a, const xform_t *b) {
	xform_t t;
	mat_mul(t.m, a->m, b->m);
	mat_mul(t.inv, b->inv, a->inv);
	*r = t;
}
static void xf_inverse(xform_t *r, const xform_t *a) { xform_t t; memcpy(t.m, a->inv, sizeof(t.m)); memcpy(t.inv, a->m, sizeof(t.m)); *r = t; }
/* transform.cpp:33-63 */
static void xf_translate(xform_t *x, float vx, float vy, float vz) {
	float m[4][4] = { {1,0,0,vx}, {0,1,0,vy}, {0,0,1,vz}, {0,0,0,1} };
	float i[4][4] = { {1,0,0,-vx}, {0,1,0,-vy}, {0,0,1,-vz}, {0,0,0,1} };
	memcpy(x->m, m, sizeof(m)); memcpy(x->inv, i, sizeof(i));
}
static void xf_scale(xform_t *x, float vx, float vy, float vz) {
	float m[4][4] = { {vx,0,0,0}, {0,vy,0,0}, {0,0,vz,0}, {0,0,0,1} };
	float i[4][4] = { {1.0f/vx,0,0,0}, {0,1.0f/vy,0,0}, {0,0,1.0f/vz,0}, {0,0,0,1} };
	memcpy(x->m, m, sizeof(m)); memcpy(x->inv, i, sizeof(i));
}

/* ========================================================================== */
/* Environment map: MIPMap::fromBitmap + EnvMapLuminaire::configure            */
/* ========================================================================== */
static int is_pow2_(uint32_t v) { return v && !(v & (v - 1)); }
static uint32_t round_to_pow2_(uint32_t i) { i--; i |= i >> 1; i |= i >> 2; i |= i >> 4; i |= i >> 8; i |= i >> 16; return i + 1; }
static int log2i_u32(uint32_t value) { int r = 0; while ((value >> r) != 0) r++; return r - 1; }      /* util.cpp:410-415 */
static int modulo_(int a, int b) { int result = a - (int) (a / b) * b; return (result < 0) ? result + b : result; }   /* util.cpp:424-427 */

/* lanczosSinc (util.cpp:664-674), tau = 2; host libm like the reference */
static float lanczos_sinc(float t, float tau) {
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

typedef struct { int firstTexel; float weight[4]; } resample_weight_t;

/* MIPMap::resampleWeights (mipmap.cpp:183-201) */
static resample_weight_t *resample_weights(int oldRes, int newRes) {
	float filterWidth = 2.0f;
	resample_weight_t *weights = (resample_weight_t *) malloc(sizeof(resample_weight_t) * (size_t) newRes);
	for (int i = 0; i < newRes; i++) {
		float center = (i + .5f) * oldRes / newRes;
		weights[i].firstTexel = (int) floorf(center - filterWidth + (float) 0.5f);
		float weightSum = 0;
		for (int j = 0; j < 4; j++) {
			float pos = weights[i].firstTexel + j + .5f;
			float weight = lanczos_sinc((pos - center) / filterWidth, 2);
			weightSum += weight;
			weights[i].weight[j] = weight;
		}
		float invWeights = 1.0f / weightSum;
		for (int j = 0; j < 4; j++)
			weights[i].weight[j] *= invWeights;
	}
	return weights;
}

/* MIPMap::getTexel with ERepeat (mipmap.cpp:203-224) */
static const float *mip_texel(const float *img, int w, int h, int x, int y) {
	if (x <= 0 || y < 0 || x >= w || y >= h) {
		x = modulo_(x, w);
		y = modulo_(y, h);
	}
	return img + 3 * ((size_t) x + (size_t) w * y);
}

/* MIPMap::fromBitmap (mipmap.cpp:161-181) -> MIPMap::MIPMap (EEWA, ERepeat; :30-92), pyramid up to the level
 * EnvMapLuminaire::configure needs (envmap.cpp:95-110); fills env_pixels / env_pdf / env_cdf and the sizes */
static int build_envmap(orc_flat_scene *fs, const mtsgpu_scene_desc *d, float *P) {
	const int width = (int) d->env_width, height = (int) d->env_height;
	if (!d->env_bitmap || width <= 0 || height <= 0 || width > 16384 || height > 16384)
		return -1;
	float *pixels = (float *) malloc(sizeof(float) * 3 * (size_t) width * height);
	for (size_t i = 0; i < 3 * (size_t) width * height; ++i)
		pixels[i] = fmaxf_((float) 0.0f, d->env_bitmap[i]);             /* fromLinearRGB + clampNegative */
	int m_width = width, m_height = height;
	float *texture = pixels;
	if (!is_pow2_((uint32_t) width) || !is_pow2_((uint32_t) height)) {
		m_width = (int) round_to_pow2_((uint32_t) width);
		m_height = (int) round_to_pow2_((uint32_t) height);
		float *texture1 = (float *) calloc(3 * (size_t) m_width * height, sizeof(float));
		resample_weight_t *weights = resample_weights(width, m_width);
		for (int y = 0; y < height; y++)
			for (int x = 0; x < m_width; x++) {
				float *dst = texture1 + 3 * ((size_t) x + (size_t) m_width * y);
				dst[0] = dst[1] = dst[2] = 0.0f;
				for (int j = 0; j < 4; j++) {
					int pos = weights[x].firstTexel + j;
					if (pos < 0 || pos >= height)                        /* sic: tested against the height (mipmap.cpp:48) */
						pos = modulo_(pos, width);
					if (pos >= 0 && pos < width)
						for (int c = 0; c < 3; ++c) dst[c] += pixels[3 * ((size_t) pos + (size_t) y * width) + c] * weights[x].weight[j];
				}
			}
		free(weights);
		free(pixels);
		texture = (float *) calloc(3 * (size_t) m_width * m_height, sizeof(float));
		weights = resample_weights(height, m_height);
		for (int x = 0; x < m_width; x++)
			for (int y = 0; y < m_height; y++)
				for (int j = 0; j < 4; j++) {
					int pos = weights[y].firstTexel + j;
					if (pos < 0 || pos >= height)
						pos = modulo_(pos, height);
					if (pos >= 0 && pos < height)
						for (int c = 0; c < 3; ++c)
							texture[3 * ((size_t) x + (size_t) m_width * y) + c] += texture1[3 * ((size_t) x + (size_t) pos * m_width) + c] * weights[y].weight[j];
				}
		for (size_t i = 0; i < 3 * (size_t) m_width * m_height; ++i)
			texture[i] = fmaxf_((float) 0.0f, texture[i]);
		free(weights);
		free(texture1);
	}
	const int levels = 1 + log2i_u32((uint32_t) (width > height ? width : height));    /* the ORIGINAL size (mipmap.cpp:81) */
	const int mipMapLevel = (3 < levels - 1) ? 3 : levels - 1;
	/* pyramid levels 1..mipMapLevel (mipmap.cpp:93-108) */
	float *cur = texture; int cw = m_width, ch = m_height;
	for (int i = 1; i <= mipMapLevel; ++i) {
		const int nw = (cw / 2 > 1) ? cw / 2 : 1, nh = (ch / 2 > 1) ? ch / 2 : 1;
		float *next = (float *) malloc(sizeof(float) * 3 * (size_t) nw * nh);
		for (int y = 0; y < nh; y++)
			for (int x = 0; x < nw; x++)
				for (int c = 0; c < 3; ++c)
					next[3 * ((size_t) x + (size_t) y * nw) + c] =
						(mip_texel(cur, cw, ch, 2*x, 2*y)[c] + mip_texel(cur, cw, ch, 2*x+1, 2*y)[c] +
						 mip_texel(cur, cw, ch, 2*x, 2*y+1)[c] + mip_texel(cur, cw, ch, 2*x+1, 2*y+1)[c]) * 0.25f;
		if (cur != texture) free(cur);
		cur = next; cw = nw; ch = nh;
	}
	/* EnvMapLuminaire::configure (envmap.cpp:95-110) */
	const int rx = cw, ry = ch;
	float *values = (float *) malloc(sizeof(float) * (size_t) rx * ry);
	int index = 0;
	for (int y = 0; y < ry; ++y) {
		float sinFactor = sinf(ORC_PI * (y + .5f) / ry);
		for (int x = 0; x < rx; ++x) {
			const float *s = cur + 3 * ((size_t) x + (size_t) y * rx);
			values[index++] = (s[0] * 0.212671f + s[1] * 0.715160f + s[2] * 0.072169f) * sinFactor;    /* getLuminance, spectrum.h:387-389 */
		}
	}
	fs->env_pdf = (float *) malloc(sizeof(float) * (size_t) rx * ry);
	fs->env_cdf = (float *) malloc(sizeof(float) * ((size_t) rx * ry + 1));
	discrete_pdf_build(values, (uint32_t) (rx * ry), fs->env_cdf, fs->env_pdf);
	free(values);
	if (cur != texture) free(cur);
	fs->env_pixels = texture;
	fs->sc.env_width = (uint32_t) m_width; fs->sc.env_height = (uint32_t) m_height;
	fs->sc.env_pdf_width = (uint32_t) rx; fs->sc.env_pdf_height = (uint32_t) ry;
	/* m_worldToLuminaire = m_luminaireToWorld.inverse() (luminaire.cpp:26-33): Matrix4x4 inverse of the rotation */
	float m[4][4] = { { P[16], P[17], P[18], 0 }, { P[19], P[20], P[21], 0 }, { P[22], P[23], P[24], 0 }, { 0, 0, 0, 1 } }, inv[4][4];
	if (!mat_invert(m, inv))
		return -1;
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) P[7 + 3*i + j] = inv[i][j];
	return 0;
}

int orc_make_camera(const float origin[3], const float target[3], const float up[3],
                    float fov_deg, int width, int height, mtsgpu_camera *out) {
	/* Transform::lookAt (transform.cpp:174-190) */
	float dirct[3], right[3], newUp[3], t[3];
	v3_sub(t, target, origin); v3_normalize(dirct, t);
	v3_cross(t, dirct, up); v3_normalize(right, t);
	v3_cross(newUp, right, dirct);
	float c2w[4][4] = {
		{ right[0], newUp[0], dirct[0], origin[0] },
		{ right[1], newUp[1], dirct[1], origin[1] },
		{ right[2], newUp[2], dirct[2], origin[2] },
		{ 0, 0, 0, 1 } };
	/* ProjectiveCamera defaults (camera.cpp:121-123) */
	const float nearClip = 1e-2f, farClip = 1e4f;
	float aspect = (float) width / (float) height;
	/* PerspectiveCameraImpl::configure (perspective.cpp:43-71), mapSmallerSide = true */
	xform_t s1, s2, tr, screenToRaster, tmp;
	if (aspect >= 1.0f) {
		xf_scale(&s1, (float) width, (float) height, 1.0f);
		xf_scale(&s2, 1/(2*aspect), -0.5f, 1.0f);
		xf_translate(&tr, aspect, -1.0f, 0);
	} else {
		xf_scale(&s1, (float) width, (float) height, 1.0f);
		xf_scale(&s2, 0.5f, -0.5f * aspect, 1.0f);
		xf_translate(&tr, 1.0f, -1 / aspect, 0);
	}
	xf_mul(&tmp, &s1, &s2);
	xf_mul(&screenToRaster, &tmp, &tr);
	/* Transform::perspective (transform.cpp:100-124); degToRad = v * (M_PI / 180.0f) */
	float recip = 1.0f / (farClip - nearClip);
	float trafo[4][4] = { {1,0,0,0}, {0,1,0,0}, {0,0,farClip * recip, -nearClip * farClip * recip}, {0,0,1,0} };
	float cot = 1.0f / tanf((fov_deg / 2.0f) * (ORC_PI / 180.0f));
	xform_t persp, sc, cameraToScreen, a, b, rasterToCamera;
	xf_from_matrix(&persp, trafo);
	xf_scale(&sc, cot, cot, 1.0f);
	xf_mul(&cameraToScreen, &sc, &persp);
	xf_inverse(&a, &cameraToScreen);
	xf_inverse(&b, &screenToRaster);
	xf_mul(&rasterToCamera, &a, &b);
	memcpy(out->raster_to_camera, rasterToCamera.m, sizeof(float) * 16);
	memcpy(out->camera_to_world, c2w, sizeof(float) * 16);
	out->near_clip = nearClip; out->far_clip = farClip;
	out->width = width; out->height = height;
	out->aperture_radius = 0.0f; out->focus_depth = farClip;    /* camera.cpp:164-166 defaults */
	out->kind = 0;
	return 0;
}

/* OrthographicCamera::configure (src/cameras/orthographic.cpp:46-82) with toWorld = lookAt * scale(sx, sy, 1) */
int orc_make_camera_ortho(const float origin[3], const float target[3], const float up[3],
                          float scale_x, float scale_y, int width, int height, mtsgpu_camera *out) {
	float dirct[3], right[3], newUp[3], t[3];
	v3_sub(t, target, origin); v3_normalize(dirct, t);
	v3_cross(t, dirct, up); v3_normalize(right, t);
	v3_cross(newUp, right, dirct);
	float look[4][4] = {
		{ right[0], newUp[0], dirct[0], origin[0] },
		{ right[1], newUp[1], dirct[1], origin[1] },
		{ right[2], newUp[2], dirct[2], origin[2] },
		{ 0, 0, 0, 1 } };
	xform_t lookAt, sc, cameraToWorld;
	xf_from_matrix(&lookAt, look);
	xf_scale(&sc, scale_x, scale_y, 1.0f);
	xf_mul(&cameraToWorld, &lookAt, &sc);
	const float nearClip = 1e-2f, farClip = 1e4f;
	float aspect = (float) width / (float) height;
	xform_t s1, s2, tr, screenToRaster, tmp;
	if (aspect >= 1.0f) {                        /* mapSmallerSide = true -> mapYToNDC01 (orthographic.cpp:57-59) */
		xf_scale(&s1, (float) width, (float) height, 1.0f);
		xf_scale(&s2, 1/(2*aspect), -0.5f, 1.0f);
		xf_translate(&tr, aspect, -1.0f, 0);
	} else {
		xf_scale(&s1, (float) width, (float) height, 1.0f);
		xf_scale(&s2, 0.5f, -0.5f * aspect, 1.0f);
		xf_translate(&tr, 1.0f, -1 / aspect, 0);
	}
	xf_mul(&tmp, &s1, &s2);
	xf_mul(&screenToRaster, &tmp, &tr);
	/* Transform::orthographic (transform.cpp:155-158) */
	xform_t osc, otr, cameraToScreen, a, b, rasterToCamera;
	xf_scale(&osc, 1.0f, 1.0f, 1.0f / (farClip - nearClip));
	xf_translate(&otr, 0.0f, 0.0f, -nearClip);
	xf_mul(&cameraToScreen, &osc, &otr);
	xf_inverse(&a, &cameraToScreen);
	xf_inverse(&b, &screenToRaster);
	xf_mul(&rasterToCamera, &a, &b);
	memcpy(out->raster_to_camera, rasterToCamera.m, sizeof(float) * 16);
	memcpy(out->camera_to_world, cameraToWorld.m, sizeof(float) * 16);
	out->near_clip = nearClip; out->far_clip = farClip;
	out->width = width; out->height = height;
	out->aperture_radius = 0.0f; out->focus_depth = farClip;
	out->kind = 1;
	return 0;
}
