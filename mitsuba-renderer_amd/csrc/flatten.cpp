// flatten.cpp -- host side of the boundary: turns plain meshes + property blocks into
// the HBM layout of mtsgpu_scene, i.e. what Scene::initialize / TriMesh::configure /
// ShapeKDTree::build / PerspectiveCameraImpl::configure leave behind in the reference.
#include "host.h"
#include "devmath.h"
#include <cmath>
#include <cstring>

namespace mg {
namespace {

inline V3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
inline void st3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// unitAngle (include/mitsuba/core/util.h:343-348)
float unitAngle(V3 u, V3 v) {
	if (dot(u, v) < 0)
		return kPi - 2 * std::asin(length(v + u) / 2);
	return 2 * std::asin(length(v - u) / 2);
}

// TriMesh::computeNormals, angle-weighted branch (src/librender/trimesh.cpp:497-539)
void computeVertexNormals(const float *pos, uint32_t nVerts, const uint32_t *tris, uint32_t nTris, float *nrm) {
	std::memset(nrm, 0, sizeof(float) * 3 * (size_t) nVerts);
	for (uint32_t t = 0; t < nTris; ++t) {
		const uint32_t *idx = tris + 3 * (size_t) t;
		V3 n(0.0f, 0.0f, 0.0f);
		for (int i = 0; i < 3; ++i) {
			const V3 v0 = ld3(pos + 3 * (size_t) idx[i]);
			const V3 v1 = ld3(pos + 3 * (size_t) idx[(i + 1) % 3]);
			const V3 v2 = ld3(pos + 3 * (size_t) idx[(i + 2) % 3]);
			const V3 sideA = v1 - v0, sideB = v2 - v0;
			if (i == 0) {
				n = cross(sideA, sideB);
				const float len = length(n);
				if (len == 0)
					break;
				n = divs(n, len);
			}
			const float angle = unitAngle(normalize(sideA), normalize(sideB));
			float *dst = nrm + 3 * (size_t) idx[i];
			dst[0] += n.x * angle; dst[1] += n.y * angle; dst[2] += n.z * angle;
		}
	}
	for (uint32_t v = 0; v < nVerts; ++v) {
		float *p = nrm + 3 * (size_t) v;
		const V3 n = ld3(p);
		const float len = length(n);
		if (len != 0) st3(p, divs(n, len));
		else st3(p, V3(1, 0, 0));
	}
}

// DiscretePDF::build (include/mitsuba/core/pdf.h:82-95)
float buildCdf(const std::vector<float> &values, float *cdf, float *pdf) {
	const size_t n = values.size();
	cdf[0] = 0.0f;
	for (size_t i = 1; i <= n; ++i)
		cdf[i] = cdf[i - 1] + values[i - 1];
	const float originalSum = cdf[n];
	for (size_t i = 0; i < n; ++i) {
		cdf[i] /= originalSum;
		if (pdf) pdf[i] = values[i] / originalSum;
	}
	cdf[n] = 1.0f;
	return originalSum;
}

} // namespace

void buildEnvMap(const mtsgpu_scene_desc &d, float *P, FlatScene &fs);     // defined below (needs the matrix inverse)

void flattenScene(const mtsgpu_scene_desc &d, const mtsgpu_kd_params *kp, FlatScene &fs) {
	const uint32_t nShapes = d.n_meshes, nLums = d.n_lums;
	size_t nVerts = 0, nTris = 0;
	// m_shapeMap (skdtree.cpp:43-60): a TriMesh contributes its triangles, any other shape ONE primitive
	for (uint32_t s = 0; s < nShapes; ++s) {
		const bool sphere = d.meshes[s].shape_type == MTSGPU_SHAPE_SPHERE;
		if (!sphere && d.meshes[s].shape_type != MTSGPU_SHAPE_TRIMESH)
			throw std::runtime_error("flatten: unknown shape type");
		nVerts += sphere ? 0 : d.meshes[s].n_verts; nTris += sphere ? 1 : d.meshes[s].n_tris;
	}
	std::vector<float> genBox(6 * nTris + 6, 0.0f);
	fs.shapeType.assign(nShapes + 1, (uint32_t) MTSGPU_SHAPE_TRIMESH);
	fs.shapeParams.assign((size_t) MTSGPU_SHAPE_NPARAMS * (nShapes + 1), 0.0f);
	if (nTris >= 0x7FFFFFFFull || nVerts >= 0xFFFFFFFFull)
		throw std::runtime_error("flatten: too many primitives");
	fs.vtxPos.assign(3 * nVerts + 3, 0.0f);
	fs.vtxNrm.assign(3 * nVerts + 3, 0.0f);
	fs.triIdx.assign(3 * nTris + 3, 0u);
	fs.shapeTriOffset.assign(nShapes + 1, 0u);
	fs.shapeFlags.assign(nShapes + 1, 0u);
	fs.shapeBsdf.assign(nShapes + 1, -1);
	fs.shapeLum.assign(nShapes + 1, -1);
	fs.triaccel.assign(12 * nTris + 12, 0u);
	fs.bsdfType.assign(d.bsdf_type, d.bsdf_type + d.n_bsdfs); fs.bsdfType.push_back(0);
	fs.bsdfParams.assign(d.bsdf_params, d.bsdf_params + (size_t) MTSGPU_BSDF_NPARAMS * d.n_bsdfs);
	fs.bsdfParams.resize(fs.bsdfParams.size() + MTSGPU_BSDF_NPARAMS, 0.0f);
	fs.lumType.assign(d.lum_type, d.lum_type + nLums); fs.lumType.push_back(0);
	fs.lumParams.assign(d.lum_params, d.lum_params + (size_t) MTSGPU_LUM_NPARAMS * nLums);
	fs.lumParams.resize(fs.lumParams.size() + MTSGPU_LUM_NPARAMS, 0.0f);
	fs.lumShape.assign(nLums + 1, -1);
	fs.lumInvArea.assign(nLums + 1, 0.0f);
	fs.lumCdfOffset.assign(nLums + 2, 0u);
	fs.lumSelCdf.assign(nLums + 2, 0.0f);
	fs.lumSelPdf.assign(nLums + 1, 0.0f);

	// ShapeKDTree::addShape order; primitive ids = prefix sums of triangle counts (skdtree.cpp:43-65)
	uint32_t vbase = 0, tbase = 0;
	for (uint32_t s = 0; s < nShapes; ++s) {
		const mtsgpu_mesh &m = d.meshes[s];
		if (m.bsdf >= (int32_t) d.n_bsdfs || m.lum >= (int32_t) nLums)
			throw std::runtime_error("flatten: mesh references a missing BSDF/luminaire");
		fs.shapeTriOffset[s] = tbase;
		fs.shapeBsdf[s] = m.bsdf;
		fs.shapeLum[s] = m.lum;
		if (m.lum >= 0) {
			if (fs.lumShape[m.lum] >= 0 || fs.lumType[m.lum] != MTSGPU_LUM_AREA)
				throw std::runtime_error("flatten: area luminaire must be attached to exactly one mesh");
			fs.lumShape[m.lum] = (int32_t) s;
		}
		if (m.shape_type == MTSGPU_SHAPE_SPHERE) {
			// Sphere::Sphere with `center` + `radius` (src/shapes/sphere.cpp:44-60): objectToWorld is a translation
			float *P = &fs.shapeParams[(size_t) MTSGPU_SHAPE_NPARAMS * s];
			fs.shapeType[s] = MTSGPU_SHAPE_SPHERE;
			const float r = m.sphere_radius;
			if (!(r != 0.0f) || !std::isfinite(r))
				throw std::runtime_error("flatten: sphere radius must be finite and non-zero");
			for (int i = 0; i < 3; ++i) P[i] = m.sphere_center[i];          // m_objectToWorld(Point(0,0,0))
			P[3] = r; P[4] = m.sphere_inverted ? 1.0f : 0.0f;
			P[5] = P[9] = P[13] = 1.0f; P[14] = P[18] = P[22] = 1.0f;
			P[23] = 1 / (4 * kPi * r * r);                                  // m_invSurfaceArea
			const float absRadius = std::fabs(r);                           // Sphere::getAABB (sphere.cpp:82-88)
			for (int i = 0; i < 3; ++i) { genBox[6 * (size_t) tbase + i] = P[i] - absRadius; genBox[6 * (size_t) tbase + 3 + i] = P[i] + absRadius; }
			for (int k = 0; k < 3; ++k) fs.triIdx[3 * (size_t) tbase + k] = MTSGPU_KNOTRIANGLE;
			tbase += 1;
			continue;
		}
		std::memcpy(&fs.vtxPos[3 * (size_t) vbase], m.positions, sizeof(float) * 3 * (size_t) m.n_verts);
		if (!m.face_normals) {
			fs.shapeFlags[s] |= MTSGPU_SHAPE_HAS_NORMALS;
			if (m.normals)
				std::memcpy(&fs.vtxNrm[3 * (size_t) vbase], m.normals, sizeof(float) * 3 * (size_t) m.n_verts);
			else
				computeVertexNormals(m.positions, m.n_verts, m.triangles, m.n_tris, &fs.vtxNrm[3 * (size_t) vbase]);
		}
		for (size_t k = 0; k < 3 * (size_t) m.n_tris; ++k) {
			if (m.triangles[k] >= m.n_verts)
				throw std::runtime_error("flatten: triangle index out of range");
			fs.triIdx[3 * (size_t) tbase + k] = m.triangles[k] + vbase;
		}
		vbase += m.n_verts; tbase += m.n_tris;
	}
	fs.shapeTriOffset[nShapes] = tbase;

	// kd-tree + TriAccel table (ShapeKDTree::build, skdtree.cpp:62-101)
	buildKdTree(fs.vtxPos.data(), fs.triIdx.data(), tbase, genBox.data(), kp, fs.kd);
	for (uint32_t s = 0; s < nShapes; ++s)
		for (uint32_t t = fs.shapeTriOffset[s]; t < fs.shapeTriOffset[s + 1]; ++t) {
			const uint32_t *tri = &fs.triIdx[3 * (size_t) t];
			uint32_t *ta = &fs.triaccel[12 * (size_t) t];
			if (fs.shapeType[s] != MTSGPU_SHAPE_TRIMESH) {
				// a 'fake' triangle which redirects to the Shape (skdtree.cpp:92-96)
				std::memset(ta, 0, 48);
				ta[0] = MTSGPU_KNOTRIANGLE; ta[10] = s;
				continue;
			}
			triAccelLoad(ld3(&fs.vtxPos[3 * (size_t) tri[0]]), ld3(&fs.vtxPos[3 * (size_t) tri[1]]), ld3(&fs.vtxPos[3 * (size_t) tri[2]]), ta);
			ta[10] = s;
			ta[11] = t - fs.shapeTriOffset[s];
		}

	// AABB::getBSphere of the enlarged tree box (aabb.cpp:44-47)
	const V3 bmin = ld3(fs.kd.aabbMin), bmax = ld3(fs.kd.aabbMax);
	const V3 center = (bmax + bmin) * 0.5f;
	const float radius = length(center - bmax);

	// per-emitter triangle CDFs (TriMesh::configure, trimesh.cpp:279-283)
	uint32_t cdfTotal = 0;
	for (uint32_t l = 0; l < nLums; ++l) {
		fs.lumCdfOffset[l] = cdfTotal;
		if (fs.lumType[l] == MTSGPU_LUM_AREA) {
			if (fs.lumShape[l] < 0)
				throw std::runtime_error("flatten: area luminaire without a mesh");
			const uint32_t s = (uint32_t) fs.lumShape[l];
			if (fs.shapeType[s] == MTSGPU_SHAPE_TRIMESH)
				cdfTotal += fs.shapeTriOffset[s + 1] - fs.shapeTriOffset[s] + 1;
		}
	}
	fs.lumCdfOffset[nLums] = cdfTotal;
	fs.lumTriCdf.assign((size_t) cdfTotal + 1, 0.0f);
	int32_t background = -1;
	for (uint32_t l = 0; l < nLums; ++l) {
		float *P = &fs.lumParams[(size_t) MTSGPU_LUM_NPARAMS * l];
		if (fs.lumType[l] == MTSGPU_LUM_AREA) {
			const uint32_t s = (uint32_t) fs.lumShape[l];
			if (fs.shapeType[s] == MTSGPU_SHAPE_SPHERE) {
				fs.lumInvArea[l] = fs.shapeParams[(size_t) MTSGPU_SHAPE_NPARAMS * s + 23];
				continue;
			}
			const uint32_t t0 = fs.shapeTriOffset[s], n = fs.shapeTriOffset[s + 1] - t0;
			std::vector<float> areas(n);
			for (uint32_t t = 0; t < n; ++t) {
				// Triangle::surfaceArea (triangle.cpp:49-55)
				const uint32_t *tri = &fs.triIdx[3 * ((size_t) t0 + t)];
				const V3 p0 = ld3(&fs.vtxPos[3 * (size_t) tri[0]]);
				const V3 sideA = ld3(&fs.vtxPos[3 * (size_t) tri[1]]) - p0, sideB = ld3(&fs.vtxPos[3 * (size_t) tri[2]]) - p0;
				areas[t] = 0.5f * length(cross(sideA, sideB));
			}
			const float surfaceArea = buildCdf(areas, &fs.lumTriCdf[fs.lumCdfOffset[l]], nullptr);
			fs.lumInvArea[l] = 1.0f / surfaceArea;
		} else if (fs.lumType[l] == MTSGPU_LUM_CONSTANT) {
			// ConstantLuminaire::preprocess (src/luminaires/constant.cpp:49-63)
			float br = radius;
			br *= 1.01f;
			if (d.has_camera) {
				const float old = br;
				br = std::max(br, length(ld3(d.camera_pos) - center));
				if (old != br)
					br *= 1.01f;
			}
			P[3] = center.x; P[4] = center.y; P[5] = center.z; P[6] = br;
			background = (int32_t) l;
		} else if (fs.lumType[l] == MTSGPU_LUM_ENVMAP) {
			// EnvMapLuminaire::preprocess (src/luminaires/envmap.cpp:112-126): the scene's bounding sphere, enlarged
			float br = radius;
			br *= 1.01f;
			if (d.has_camera) {
				const float old = br;
				br = std::max(br, length(ld3(d.camera_pos) - center));
				if (old != br)
					br *= 1.01f;
			}
			P[3] = center.x; P[4] = center.y; P[5] = center.z; P[6] = br;
			if (background >= 0)
				throw std::runtime_error("flatten: more than one background luminaire");
			buildEnvMap(d, P, fs);
			background = (int32_t) l;
		} else if (fs.lumType[l] == MTSGPU_LUM_POINT || fs.lumType[l] == MTSGPU_LUM_COLLIMATED) {
			// nothing to derive (point.cpp:28-33)
		} else if (fs.lumType[l] == MTSGPU_LUM_DIRECTIONAL) {
			P[6] = radius;                          // DirectionalLuminaire::preprocess (directional.cpp:65-72)
		} else if (fs.lumType[l] == MTSGPU_LUM_SPOT) {
			P[6] = std::cos(P[19]);                 // SpotLuminaire::configure (spot.cpp:56-62)
			P[7] = std::cos(P[8]);
			P[9] = 1.0f / (P[8] - P[19]);
		} else {
			throw std::runtime_error("flatten: unknown luminaire type");
		}
	}
	// Scene::initialize: luminaire selection PDF, weight getSamplingWeight() = 1 (scene.cpp:320-330)
	float selSum = 0.0f;
	if (nLums > 0) {
		std::vector<float> w(nLums, 1.0f);
		selSum = buildCdf(w, fs.lumSelCdf.data(), fs.lumSelPdf.data());
	}

	mtsgpu_scene &sc = fs.sc;
	const uint32_t envW = sc.env_width, envH = sc.env_height, envPW = sc.env_pdf_width, envPH = sc.env_pdf_height;   // set by buildEnvMap
	std::memset(&sc, 0, sizeof(sc));
	sc.env_width = envW; sc.env_height = envH; sc.env_pdf_width = envPW; sc.env_pdf_height = envPH;
	sc.env_pixels = fs.envPixels.data(); sc.env_pdf = fs.envPdf.data(); sc.env_cdf = fs.envCdf.data();
	sc.abi_version = MTSGPU_ABI_VERSION;
	sc.n_shapes = nShapes; sc.n_tris = tbase; sc.n_verts = vbase;
	sc.vtx_pos = fs.vtxPos.data(); sc.vtx_nrm = fs.vtxNrm.data(); sc.tri_idx = fs.triIdx.data();
	sc.shape_tri_offset = fs.shapeTriOffset.data(); sc.shape_bsdf = fs.shapeBsdf.data();
	sc.shape_lum = fs.shapeLum.data(); sc.shape_flags = fs.shapeFlags.data();
	sc.shape_type = fs.shapeType.data(); sc.shape_params = fs.shapeParams.data();
	sc.n_nodes = (uint32_t) (fs.kd.nodes.size() / 2); sc.n_indices = (uint32_t) fs.kd.indices.size();
	if (fs.kd.indices.empty()) fs.kd.indices.push_back(0);
	sc.kd_nodes = fs.kd.nodes.data(); sc.kd_indices = fs.kd.indices.data(); sc.triaccel = fs.triaccel.data();
	for (int a = 0; a < 3; ++a) { sc.aabb_min[a] = fs.kd.aabbMin[a]; sc.aabb_max[a] = fs.kd.aabbMax[a]; }
	sc.n_bsdfs = d.n_bsdfs; sc.bsdf_type = fs.bsdfType.data(); sc.bsdf_params = fs.bsdfParams.data();
	sc.n_lums = nLums; sc.lum_type = fs.lumType.data(); sc.lum_params = fs.lumParams.data();
	sc.lum_shape = fs.lumShape.data(); sc.lum_inv_area = fs.lumInvArea.data();
	sc.lum_cdf_offset = fs.lumCdfOffset.data(); sc.lum_tri_cdf = fs.lumTriCdf.data();
	sc.lum_sel_cdf = fs.lumSelCdf.data(); sc.lum_sel_pdf = fs.lumSelPdf.data();
	sc.lum_sel_sum = selSum;
	sc.background_lum = background;
}

// TabulatedFilter::TabulatedFilter (src/librender/rfilter.cpp:40-69) over BoxFilter::evaluate
// (src/rfilters/box.cpp:42-44) or GaussianFilter (src/rfilters/gaussian.cpp:30-42,62-65).
// Host-side configure step; std::exp is the same libm call the reference makes.
namespace {

// mitchellNetravali (src/rfilters/mitchell.cpp:60-74, catmullrom.cpp with B = 0, C = 1/2)
float mitchellNetravali(float x, float B, float C) {
	x = std::fabs(x);
	const float xSquared = x * x, xCubed = xSquared * x;
	if (x < 1)
		return 1.0f / 6.0f * ((12 - 9 * B - 6 * C) * xCubed + (-18 + 12 * B + 6 * C) * xSquared + (6 - 2 * B));
	if (x < 2)
		return 1.0f / 6.0f * ((-B - 6 * C) * xCubed + (6 * B + 30 * C) * xSquared + (-12 * B - 48 * C) * x + (8 * B + 24 * C));
	return 0.0f;
}

// lanczosSinc (src/libcore/util.cpp:664-674)
float windowedSinc(float t, float tau) {
	t = std::fabs(t);
	if (t < kEpsilon) return 1.0f;
	if (t > 1.0f) return 0.0f;
	t *= kPi;
	const float sincTerm = std::sin(t * tau) / (t * tau);
	const float windowTerm = std::sin(t) / t;
	return sincTerm * windowTerm;
}

} // namespace

// TabulatedFilter::TabulatedFilter (src/librender/rfilter.cpp:40-69) over ReconstructionFilter::evaluate of
// box (0), gaussian (1; p0 = stddev), mitchell (2; p0 = B, p1 = C), catmullrom (3), wsinc (4; p0 = cycles)
void tabulateFilter(int kind, float halfSize, float p0, float p1, float *sizeXY, float *values) {
	constexpr int R = 15;                                    // FILTER_RESOLUTION
	float alpha = 0, cst = 0, sx = 0.5f, sy = 0.5f;
	if (kind == 1) {
		if (halfSize <= 0) halfSize = 2.0f;
		if (p0 <= 0) p0 = 0.5f;
		alpha = 1 / (2 * p0 * p0);
		sx = sy = halfSize;
		cst = std::exp(-alpha * sx * sx);
	} else if (kind == 2 || kind == 3) {
		if (halfSize <= 0) halfSize = 2.0f;
		if (kind == 3) { p0 = 0.0f; p1 = 0.5f; }
		else { if (p0 < 0) p0 = 1.0f / 3.0f; if (p1 < 0) p1 = 1.0f / 3.0f; }
		sx = sy = halfSize;
	} else if (kind == 4) {
		if (halfSize <= 0) halfSize = 3.0f;
		if (p0 <= 0) p0 = 3.0f;
		sx = sy = halfSize;
	}
	auto evaluate = [&](float x, float y) -> float {
		switch (kind) {
			case 1: return std::max(0.0f, std::exp(-alpha * x * x) - cst) * std::max(0.0f, std::exp(-alpha * y * y) - cst);
			case 2: case 3: return mitchellNetravali(2.0f * x / sx, p0, p1) * mitchellNetravali(2.0f * y / sy, p0, p1);
			case 4: return windowedSinc(x / sx, p0) * windowedSinc(y / sy, p0);
			default: return 1.0f;
		}
	};
	float sum = 0;
	for (int y = 0; y < R + 1; ++y) {
		const float yPos = (y + 0.5f) / R * sy;
		for (int x = 0; x < R + 1; ++x) {
			float v = 0;
			if (x != R && y != R)
				v = evaluate((x + 0.5f) / R * sx, yPos);
			values[y * 16 + x] = v;
			sum += v;
		}
	}
	sum *= 4 * sx * sy / (R * R);
	for (int i = 0; i < 256; ++i) values[i] /= sum;
	sizeXY[0] = sx; sizeXY[1] = sy;
}

// ---------------------------------------------------------------------------
// Camera: Transform algebra of src/libcore/transform.cpp with the generic
// Gauss-Jordan Matrix::invert (include/mitsuba/core/matrix.inl:140-190)
// ---------------------------------------------------------------------------
namespace {

struct M4 { float m[4][4]; };
struct Xf { M4 fwd, inv; };

M4 mul(const M4 &a, const M4 &b) {
	M4 r;
	for (int i = 0; i < 4; ++i)
		for (int j = 0; j < 4; ++j) {
			float sum = 0;
			for (int k = 0; k < 4; ++k)
				sum += a.m[i][k] * b.m[k][j];
			r.m[i][j] = sum;
		}
	return r;
}

bool invert(const M4 &src, M4 &t) {
	int indxc[4], indxr[4], ipiv[4] = { 0, 0, 0, 0 };
	t = src;
	for (int i = 0; i < 4; i++) {
		int irow = -1, icol = -1;
		float big = 0;
		for (int j = 0; j < 4; j++) {
			if (ipiv[j] == 1) continue;
			for (int k = 0; k < 4; k++) {
				if (ipiv[k] == 0) {
					if (std::abs(t.m[j][k]) >= big) { big = std::abs(t.m[j][k]); irow = j; icol = k; }
				} else if (ipiv[k] > 1) {
					return false;
				}
			}
		}
		++ipiv[icol];
		if (irow != icol)
			for (int k = 0; k < 4; ++k) std::swap(t.m[irow][k], t.m[icol][k]);
		indxr[i] = irow; indxc[i] = icol;
		if (t.m[icol][icol] == 0)
			return false;
		const float pivinv = 1.f / t.m[icol][icol];
		t.m[icol][icol] = 1.f;
		for (int j = 0; j < 4; j++) t.m[icol][j] *= pivinv;
		for (int j = 0; j < 4; j++) {
			if (j == icol) continue;
			const float save = t.m[j][icol];
			t.m[j][icol] = 0;
			for (int k = 0; k < 4; k++) t.m[j][k] -= t.m[icol][k] * save;
		}
	}
	for (int j = 3; j >= 0; j--)
		if (indxr[j] != indxc[j])
			for (int k = 0; k < 4; k++) std::swap(t.m[k][indxr[j]], t.m[k][indxc[j]]);
	return true;
}

Xf operator*(const Xf &a, const Xf &b) { return Xf{ mul(a.fwd, b.fwd), mul(b.inv, a.inv) }; }   // transform.cpp:28-31
Xf inverse(const Xf &a) { return Xf{ a.inv, a.fwd }; }
Xf translate(float x, float y, float z) {                                                       // transform.cpp:33-47
	return Xf{ M4{ { { 1, 0, 0, x }, { 0, 1, 0, y }, { 0, 0, 1, z }, { 0, 0, 0, 1 } } },
	           M4{ { { 1, 0, 0, -x }, { 0, 1, 0, -y }, { 0, 0, 1, -z }, { 0, 0, 0, 1 } } } };
}
Xf scale(float x, float y, float z) {                                                           // transform.cpp:49-63
	return Xf{ M4{ { { x, 0, 0, 0 }, { 0, y, 0, 0 }, { 0, 0, z, 0 }, { 0, 0, 0, 1 } } },
	           M4{ { { 1.0f / x, 0, 0, 0 }, { 0, 1.0f / y, 0, 0 }, { 0, 0, 1.0f / z, 0 }, { 0, 0, 0, 1 } } } };
}

} // namespace

// ---------------------------------------------------------------------------
// Environment map: MIPMap::fromBitmap (src/librender/mipmap.cpp:161-181) -> MIPMap::MIPMap with the
// defaults EEWA / ERepeat (:30-108), and the sampling density of EnvMapLuminaire::configure
// (src/luminaires/envmap.cpp:95-110)
// ---------------------------------------------------------------------------
namespace {

struct Rgb { float c[3]; };
struct Image {
	int w = 0, h = 0;
	std::vector<Rgb> px;
	Image(int w_, int h_) : w(w_), h(h_), px((size_t) w_ * h_, Rgb{ { 0, 0, 0 } }) {}
	Rgb &at(int x, int y) { return px[(size_t) x + (size_t) w * y]; }
	// MIPMap::getTexel with ERepeat (mipmap.cpp:203-224); modulo (util.cpp:424-427)
	const Rgb &texel(int x, int y) const {
		if (x <= 0 || y < 0 || x >= w || y >= h) { x = wrap(x, w); y = wrap(y, h); }
		return px[(size_t) x + (size_t) w * y];
	}
	static int wrap(int a, int b) { const int r = a - (a / b) * b; return r < 0 ? r + b : r; }
};

bool isPow2(uint32_t v) { return v && !(v & (v - 1)); }
uint32_t roundToPow2(uint32_t i) { i--; i |= i >> 1; i |= i >> 2; i |= i >> 4; i |= i >> 8; i |= i >> 16; return i + 1; }
int log2iU32(uint32_t value) { int r = 0; while ((value >> r) != 0) r++; return r - 1; }        // util.cpp:410-415

// lanczosSinc (util.cpp:664-674) with tau = 2; host libm like the reference
float lanczosSinc(float t) {
	const float tau = 2;
	t = std::fabs(t);
	if (t < kEpsilon) return 1.0f;
	if (t > 1.0f) return 0.0f;
	t *= kPi;
	const float sincTerm = std::sin(t * tau) / (t * tau);
	const float windowTerm = std::sin(t) / t;
	return sincTerm * windowTerm;
}

struct ResampleWeight { int firstTexel; float weight[4]; };

// MIPMap::resampleWeights (mipmap.cpp:183-201)
std::vector<ResampleWeight> resampleWeights(int oldRes, int newRes) {
	std::vector<ResampleWeight> w((size_t) newRes);
	const float filterWidth = 2.0f;
	for (int i = 0; i < newRes; ++i) {
		const float center = (i + .5f) * oldRes / newRes;
		w[i].firstTexel = (int) std::floor(center - filterWidth + 0.5f);
		float weightSum = 0;
		for (int j = 0; j < 4; ++j) {
			const float pos = w[i].firstTexel + j + .5f;
			const float weight = lanczosSinc((pos - center) / filterWidth);
			weightSum += weight;
			w[i].weight[j] = weight;
		}
		const float invWeights = 1.0f / weightSum;
		for (int j = 0; j < 4; ++j) w[i].weight[j] *= invWeights;
	}
	return w;
}

} // namespace

void buildEnvMap(const mtsgpu_scene_desc &d, float *P, FlatScene &fs) {
	const int width = (int) d.env_width, height = (int) d.env_height;
	if (!d.env_bitmap || width <= 0 || height <= 0 || width > 16384 || height > 16384)
		throw std::runtime_error("flatten: the envmap luminaire needs a bitmap of at most 16384 x 16384 pixels");
	Image src(width, height);
	for (size_t i = 0; i < (size_t) width * height; ++i)
		for (int c = 0; c < 3; ++c)
			src.px[i].c[c] = std::max(0.0f, d.env_bitmap[3 * i + c]);          // fromLinearRGB + clampNegative
	Image level0 = src;
	if (!isPow2((uint32_t) width) || !isPow2((uint32_t) height)) {
		// up-sampling to powers of two, x then y (mipmap.cpp:35-78)
		const int W = (int) roundToPow2((uint32_t) width), H = (int) roundToPow2((uint32_t) height);
		Image tmp(W, height);
		const std::vector<ResampleWeight> wx = resampleWeights(width, W);
		for (int y = 0; y < height; ++y)
			for (int x = 0; x < W; ++x)
				for (int j = 0; j < 4; ++j) {
					int pos = wx[x].firstTexel + j;
					if (pos < 0 || pos >= height)                                // sic: compared with the height (mipmap.cpp:48)
						pos = Image::wrap(pos, width);
					if (pos >= 0 && pos < width)
						for (int c = 0; c < 3; ++c) tmp.at(x, y).c[c] += src.at(pos, y).c[c] * wx[x].weight[j];
				}
		level0 = Image(W, H);
		const std::vector<ResampleWeight> wy = resampleWeights(height, H);
		for (int x = 0; x < W; ++x)
			for (int y = 0; y < H; ++y)
				for (int j = 0; j < 4; ++j) {
					int pos = wy[y].firstTexel + j;
					if (pos < 0 || pos >= height)
						pos = Image::wrap(pos, height);
					if (pos >= 0 && pos < height)
						for (int c = 0; c < 3; ++c) level0.at(x, y).c[c] += tmp.at(x, pos).c[c] * wy[y].weight[j];
				}
		for (Rgb &p : level0.px)
			for (int c = 0; c < 3; ++c) p.c[c] = std::max(0.0f, p.c[c]);
	}
	// the number of levels follows the ORIGINAL size (mipmap.cpp:81); the density uses level min(3, levels - 1)
	const int levels = 1 + log2iU32((uint32_t) std::max(width, height));
	const int pdfLevel = std::min(3, levels - 1);
	Image cur = level0;
	for (int i = 1; i <= pdfLevel; ++i) {
		Image next(std::max(1, cur.w / 2), std::max(1, cur.h / 2));
		for (int y = 0; y < next.h; ++y)
			for (int x = 0; x < next.w; ++x)
				for (int c = 0; c < 3; ++c)
					next.at(x, y).c[c] = (cur.texel(2 * x, 2 * y).c[c] + cur.texel(2 * x + 1, 2 * y).c[c]
					                      + cur.texel(2 * x, 2 * y + 1).c[c] + cur.texel(2 * x + 1, 2 * y + 1).c[c]) * 0.25f;
		cur = next;
	}
	std::vector<float> values((size_t) cur.w * cur.h);
	size_t index = 0;
	for (int y = 0; y < cur.h; ++y) {
		const float sinFactor = std::sin(kPi * (y + .5f) / cur.h);
		for (int x = 0; x < cur.w; ++x) {
			const Rgb &s = cur.at(x, y);
			values[index++] = (s.c[0] * 0.212671f + s.c[1] * 0.715160f + s.c[2] * 0.072169f) * sinFactor;   // getLuminance
		}
	}
	fs.envPdf.assign(values.size(), 0.0f);
	fs.envCdf.assign(values.size() + 1, 0.0f);
	buildCdf(values, fs.envCdf.data(), fs.envPdf.data());
	fs.envPixels.resize(3 * level0.px.size());
	for (size_t i = 0; i < level0.px.size(); ++i)
		for (int c = 0; c < 3; ++c) fs.envPixels[3 * i + c] = level0.px[i].c[c];
	fs.sc.env_width = (uint32_t) level0.w; fs.sc.env_height = (uint32_t) level0.h;
	fs.sc.env_pdf_width = (uint32_t) cur.w; fs.sc.env_pdf_height = (uint32_t) cur.h;
	// m_worldToLuminaire = m_luminaireToWorld.inverse() (luminaire.cpp:30-37)
	M4 m{}, inv{};
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.m[i][j] = P[16 + 3 * i + j];
	m.m[3][3] = 1.0f;
	if (!invert(m, inv))
		throw std::runtime_error("flatten: the envmap's toWorld rotation is singular");
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) P[7 + 3 * i + j] = inv.m[i][j];
}

void makeCamera(const float origin[3], const float target[3], const float up[3], float fovDeg, int width, int height,
                mtsgpu_camera &out) {
	// Transform::lookAt (transform.cpp:174-190): columns right, newUp, dir, p
	const V3 p = ld3(origin);
	const V3 dir = normalize(ld3(target) - p);
	const V3 right = normalize(cross(dir, ld3(up)));
	const V3 newUp = cross(right, dir);
	const float c2w[16] = { right.x, newUp.x, dir.x, p.x,  right.y, newUp.y, dir.y, p.y,
	                        right.z, newUp.z, dir.z, p.z,  0, 0, 0, 1 };
	const float nearClip = 1e-2f, farClip = 1e4f;                                               // camera.cpp:121-123
	const float aspect = (float) width / (float) height;
	// PerspectiveCameraImpl::configure (perspective.cpp:43-71), mapSmallerSide = true
	Xf screenToRaster;
	if (aspect >= 1.0f)
		screenToRaster = scale((float) width, (float) height, 1.0f) * scale(1 / (2 * aspect), -0.5f, 1.0f) * translate(aspect, -1.0f, 0);
	else
		screenToRaster = scale((float) width, (float) height, 1.0f) * scale(0.5f, -0.5f * aspect, 1.0f) * translate(1.0f, -1 / aspect, 0);
	// Transform::perspective (transform.cpp:100-124)
	const float recip = 1.0f / (farClip - nearClip);
	Xf persp;
	persp.fwd = M4{ { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, farClip * recip, -nearClip * farClip * recip }, { 0, 0, 1, 0 } } };
	invert(persp.fwd, persp.inv);
	const float cot = 1.0f / std::tan((fovDeg / 2.0f) * (kPi / 180.0f));
	const Xf cameraToScreen = scale(cot, cot, 1.0f) * persp;
	const Xf rasterToCamera = inverse(cameraToScreen) * inverse(screenToRaster);
	std::memcpy(out.raster_to_camera, rasterToCamera.fwd.m, sizeof(float) * 16);
	std::memcpy(out.camera_to_world, c2w, sizeof(float) * 16);
	out.near_clip = nearClip; out.far_clip = farClip;
	out.width = width; out.height = height;
	out.aperture_radius = 0.0f; out.focus_depth = farClip;     // camera.cpp:164-166 defaults
	out.kind = 0;
	out.crop_offset_x = out.crop_offset_y = 0; out.film_width = out.film_height = 0;
}

// OrthographicCamera::configure (src/cameras/orthographic.cpp:46-82) with toWorld = lookAt * scale(sx, sy, 1)
void makeCameraOrtho(const float origin[3], const float target[3], const float up[3], float scaleX, float scaleY,
                     int width, int height, mtsgpu_camera &out) {
	const V3 p = ld3(origin);
	const V3 dir = normalize(ld3(target) - p);
	const V3 right = normalize(cross(dir, ld3(up)));
	const V3 newUp = cross(right, dir);
	Xf lookAt;
	lookAt.fwd = M4{ { { right.x, newUp.x, dir.x, p.x }, { right.y, newUp.y, dir.y, p.y }, { right.z, newUp.z, dir.z, p.z }, { 0, 0, 0, 1 } } };
	invert(lookAt.fwd, lookAt.inv);
	const Xf cameraToWorld = lookAt * scale(scaleX, scaleY, 1.0f);
	const float nearClip = 1e-2f, farClip = 1e4f;
	const float aspect = (float) width / (float) height;
	Xf screenToRaster;
	if (aspect >= 1.0f)
		screenToRaster = scale((float) width, (float) height, 1.0f) * scale(1 / (2 * aspect), -0.5f, 1.0f) * translate(aspect, -1.0f, 0);
	else
		screenToRaster = scale((float) width, (float) height, 1.0f) * scale(0.5f, -0.5f * aspect, 1.0f) * translate(1.0f, -1 / aspect, 0);
	const Xf cameraToScreen = scale(1.0f, 1.0f, 1.0f / (farClip - nearClip)) * translate(0.0f, 0.0f, -nearClip);   // transform.cpp:155-158
	const Xf rasterToCamera = inverse(cameraToScreen) * inverse(screenToRaster);
	std::memcpy(out.raster_to_camera, rasterToCamera.fwd.m, sizeof(float) * 16);
	std::memcpy(out.camera_to_world, cameraToWorld.fwd.m, sizeof(float) * 16);
	out.near_clip = nearClip; out.far_clip = farClip;
	out.width = width; out.height = height;
	out.aperture_radius = 0.0f; out.focus_depth = farClip;
	out.kind = 1;
	out.crop_offset_x = out.crop_offset_y = 0; out.film_width = out.film_height = 0;
}

} // namespace mg
