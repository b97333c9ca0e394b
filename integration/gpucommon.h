/*
 * gpucommon.h -- shared by the two reference-side bindings of libmtsgpu's C ABI (include/mtsgpu.h): the Mitsuba 0.2.1
 * integrator plugins `gpupath` (integration/gpupath.cpp, in place of src/integrators/path) and `gpudirect`
 * (integration/gpudirect.cpp, in place of src/integrators/direct).  A Mitsuba plugin exports ONE class
 * (MTS_EXPORT_PLUGIN, cobject.h:79-87), so each integrator is its own .cpp / .so and what they share lives here:
 *
 *   FlatScene        Scene / ShapeKDTree / TriMesh / Sphere / BSDFs / luminaires  ->  mtsgpu_scene
 *   GPURenderDriver  Integrator::render: upload, camera, sampler, film filter, the device group's render, and the film
 *                    handed back as ImageBlocks (Film::putImageBlock + RenderQueue::signalWorkEnd,
 *                    src/librender/renderproc.cpp:123-130)
 *
 * NOT compiled in this repository's image (Mitsuba's headers need Boost, include/mitsuba/core/util.h:22); every call
 * into Mitsuba is annotated with the file:line of the declaration it relies on.
 *
 * What of a Mitsuba scene reaches the GPU (everything else stops with an error message that names the class):
 *   shapes      every TriMesh (obj, ply, serialized, ...: TriMesh accessors), `sphere`
 *   BSDFs       lambertian, dielectric, roughmetal, microfacet, mirror, phong, roughglass, difftrans, twosided(any of them);
 *               constant reflectances only (textures need uv partials and a MIPMap lookup per hit: out of scope)
 *   luminaires  area (on meshes and spheres), constant, point, spot (no projection texture), directional, collimated, envmap
 *   cameras     perspective (pinhole and thin lens), orthographic
 *   samplers    independent, ldsampler, stratified (keyed forms), halton, hammersley
 *   films       any (the result goes back as ImageBlocks); reconstruction filter = the film's own TabulatedFilter
 * Private members that block more: Texture2D bitmaps (BitmapTexture::m_mipmap has no accessor), `ward` and `composite`
 * BSDFs and every shape other than TriMesh / Sphere (not implemented by the library, not blocked by access), media and
 * subsurface integrators (path.cpp ignores media, :31-34).
 */
#ifndef MTSGPU_GPUCOMMON_H
#define MTSGPU_GPUCOMMON_H

#include <mitsuba/render/scene.h>
#include <mitsuba/render/renderproc.h>
#include <mitsuba/render/renderjob.h>
#include <mitsuba/render/imageblock.h>
#include <mitsuba/render/triaccel.h>
#include <mitsuba/render/texture.h>
#include <mitsuba/render/mipmap.h>
#include <mitsuba/core/bitmap.h>
#include <mitsuba/core/mstream.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/sched.h>
#include <mtsgpu.h>
#include "streamparse.h"
#include <sstream>
#include <cstdlib>

MTS_NAMESPACE_BEGIN

namespace {

/* ImageBlock keeps its alpha channel protected and has no setter (include/mitsuba/render/imageblock.h:277-297) */
class FilmBlock : public ImageBlock {
public:
	FilmBlock(const Vector2i &maxSize) : ImageBlock(maxSize, 0, true, true, false, false) { }   /* imageblock.h:64-66 */
	inline void setAlpha(size_t idx, Float a) { alpha[idx] = a; }
};

inline void rgbOf(const Spectrum &s, float *out) {
	Float r, g, b;
	s.toLinearRGB(r, g, b);                      /* SPECTRUM_SAMPLES == 3: the identity (spectrum.h) */
	out[0] = (float) r; out[1] = (float) g; out[2] = (float) b;
}

inline void copyMatrix(float *dst, const Matrix4x4 &m) {
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) dst[4 * i + j] = (float) m.m[i][j];     /* matrix layout: transform.h */
}

/* The private parameters of BSDFs, delta / environment luminaires and spheres are read back from what the object's own
 * serialize() writes: the object goes through InstanceManager::serialize into memory (wire format,
 * src/libcore/serialization.cpp:72-85: a new object is [id][class name][its serialize()], a known one just [id], NULL is [0])
 * and the bytes are taken apart by integration/streamparse.h, which follows the field order of each class's serialize() and is
 * tested on bytes without Mitsuba (tests/test_stream_parsers.py).  Nothing is re-instantiated.
 *
 * BSDFs are serialized as they stand: they have no parent (BSDF::setParent is empty, bsdf.cpp:55-57) and their only children
 * are textures. */
inline ref<MemoryStream> serializedBSDF(const BSDF *bsdf) {
	ref<MemoryStream> stream = new MemoryStream();                            /* include/mitsuba/core/mstream.h:40 */
	ref<InstanceManager> writer = new InstanceManager();
	writer->serialize(stream, bsdf);
	return stream;
}

/* A scene-level object (a delta luminaire, the environment map, a sphere): these plugins live in their .cpp files and keep
 * every parameter private.  ConfigurableObject::serialize (properties.cpp:358-363) starts with the PARENT, which for such an
 * object is the Scene: serializing it as it stands would drag the whole scene along.  The parent pointer is public API
 * (getParent / setParent, cobject.h:40-46, properties.cpp:351-353: a plain store), so it is taken off for the duration of the
 * call and put back; render() runs on the RenderJob thread before any worker touches the scene. */
inline ref<MemoryStream> serializedDetached(ConfigurableObject *obj) {
	ConfigurableObject *parent = obj->getParent();
	obj->setParent(NULL);
	ref<MemoryStream> stream = new MemoryStream();
	ref<InstanceManager> writer = new InstanceManager();
	try {
		writer->serialize(stream, obj);                                        /* serialization.cpp:72-85: [id][class name][serialize()] */
	} catch (...) {
		obj->setParent(parent);
		throw;
	}
	obj->setParent(parent);
	return stream;
}

inline void copy3x3(float *dst, const Matrix4x4 &m) {
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dst[3 * i + j] = (float) m.m[i][j];
}
inline void copy3x4(float *dst, const Matrix4x4 &m) {
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) dst[4 * i + j] = (float) m.m[i][j];
}

/* ------------------------------------------------------------------------------------------------------------
 * FlatScene: Scene + ShapeKDTree + TriMesh + BSDF / luminaire parameter blocks -> mtsgpu_scene
 * ---------------------------------------------------------------------------------------------------------- */
struct FlatScene {
	mtsgpu_scene sc;
	std::vector<float> vtxPos, vtxNrm, shapeParams, bsdfParams, lumParams, lumInvArea, lumTriCdf, lumSelCdf, lumSelPdf;
	std::vector<float> envPixels, envPdf, envCdf;
	std::vector<uint32_t> triIdx, shapeTriOffset, shapeFlags, shapeType, kdNodes, kdIndices, triaccel, bsdfType, lumType, lumCdfOffset;
	std::vector<int32_t> shapeBsdf, shapeLum, lumShape;
	std::map<const BSDF *, int> bsdfIndex;

	FlatScene(const Scene *scene) {
		memset(&sc, 0, sizeof(sc));
		sc.abi_version = MTSGPU_ABI_VERSION;
		const ShapeKDTree *kd = scene->getKDTree();                                   /* scene.h:498 */
		const std::vector<const Shape *> &shapes = kd->getShapes();                   /* skdtree.h:83: m_shapes order */
		const std::vector<Luminaire *> &lums = scene->getLuminaires();                /* scene.h:519 */

		/* --- luminaires first (shapes refer to them by index), Scene::m_luminaires order --- */
		std::map<const Luminaire *, int> lumIndex;
		lumParams.assign((size_t) MTSGPU_LUM_NPARAMS * lums.size(), 0.0f);
		lumShape.assign(lums.size(), -1); lumInvArea.assign(lums.size(), 0.0f);
		sc.background_lum = -1;
		for (size_t l = 0; l < lums.size(); ++l) {
			Luminaire *lum = lums[l];                                                 /* non-const: serializedDetached takes the parent off and puts it back */
			lumIndex[lum] = (int) l;
			float *P = &lumParams[(size_t) MTSGPU_LUM_NPARAMS * l];
			const std::string cls = lum->getClass()->getName();
			std::string err;
			/* area, constant and point luminaires hand their stored values out through the public evaluation interface
			 * unchanged; the others keep them private and are read back from serialize() (serializedDetached + streamparse.h) */
			if (cls == "AreaLuminaire") {
				lumType.push_back(MTSGPU_LUM_AREA);
				ShapeSamplingRecord sRec; sRec.n = Normal(0, 0, 1);
				rgbOf(lum->Le(sRec, Vector(0, 0, 1)), P);                           /* area.cpp:62-66: m_intensity when dot(d, n) > 0 */
			} else if (cls == "ConstantLuminaire") {
				lumType.push_back(MTSGPU_LUM_CONSTANT);
				rgbOf(lum->Le(Ray(Point(0, 0, 0), Vector(0, 0, 1), 0.0f)), P);      /* constant.cpp: Le(ray) = m_intensity */
				/* m_bsphere as ConstantLuminaire::preprocess derives it (constant.cpp:49-63) */
				BSphere bs = scene->getBSphere();                                   /* scene.h:273 */
				bs.radius *= 1.01f;
				if (scene->getCamera()) {
					const BSphere old = bs;
					bs.expandBy(scene->getCamera()->getPosition());                 /* bsphere.h; camera.h:83 */
					if (old != bs) bs.radius *= 1.01f;
				}
				P[3] = (float) bs.center.x; P[4] = (float) bs.center.y; P[5] = (float) bs.center.z; P[6] = (float) bs.radius;
				sc.background_lum = (int32_t) l;
			} else if (cls == "PointLuminaire") {
				lumType.push_back(MTSGPU_LUM_POINT);
				/* the public emission interface returns both stored values as they are (point.cpp:71-77) */
				EmissionRecord eRec;
				lum->sampleEmission(eRec, Point2(0.5f, 0.5f), Point2(0.5f, 0.5f));
				rgbOf(eRec.value, P);
				P[3] = (float) eRec.sRec.p.x; P[4] = (float) eRec.sRec.p.y; P[5] = (float) eRec.sRec.p.z;
			} else if (cls == "DirectionalLuminaire") {
				lumType.push_back(MTSGPU_LUM_DIRECTIONAL);
				ref<MemoryStream> st = serializedDetached(lum);                     /* directional.cpp:57-63; m_diskRadius after preprocess (:65-72) */
				if (!mtsgpu_stream::parseDirectional<Float>(st->getData(), st->getSize(), P, &err)) SLog(EError, "gpupath: %s", err.c_str());
			} else if (cls == "SpotLuminaire") {
				lumType.push_back(MTSGPU_LUM_SPOT);
				ref<MemoryStream> st = serializedDetached(lum);                     /* spot.cpp:63-70, configure() :56-62 */
				if (!mtsgpu_stream::parseSpot<Float>(st->getData(), st->getSize(), P, &err)) SLog(EError, "gpupath: %s", err.c_str());
			} else if (cls == "CollimatedBeamLuminaire") {
				lumType.push_back(MTSGPU_LUM_COLLIMATED);
				ref<MemoryStream> st = serializedDetached(lum);                     /* collimated.cpp:47-51 */
				if (!mtsgpu_stream::parseCollimated<Float>(st->getData(), st->getSize(), P, &err)) SLog(EError, "gpupath: %s", err.c_str());
			} else if (cls == "EnvMapLuminaire") {
				lumType.push_back(MTSGPU_LUM_ENVMAP);
				if (sc.background_lum >= 0) SLog(EError, "gpupath: more than one background luminaire");
				sc.background_lum = (int32_t) l;
				readEnvMap(lum, P);
			} else {
				SLog(EError, "gpupath: luminaire class %s is not on this path (area, constant, point, spot, directional, collimated "
					"and envmap are)", cls.c_str());
			}
		}

		/* --- shapes: primitive index space = concatenation in m_shapes order (skdtree.cpp:43-65) --- */
		shapeTriOffset.push_back(0);
		for (size_t s = 0; s < shapes.size(); ++s) {
			const Shape *shape = shapes[s];
			const bool isMesh = shape->getClass()->derivesFrom(MTS_CLASS(TriMesh));   /* skdtree.cpp:46-57 */
			shapeBsdf.push_back(bsdfOf(shape->getBSDF()));                            /* shape.h:385; NULL = not an occluder */
			shapeLum.push_back(shape->isLuminaire() ? lumIndex[shape->getLuminaire()] : -1);   /* shape.h:329-340 */
			shapeParams.insert(shapeParams.end(), MTSGPU_SHAPE_NPARAMS, 0.0f);
			if (isMesh) {
				const TriMesh *mesh = static_cast<const TriMesh *>(shape);
				const uint32_t base = (uint32_t) (vtxPos.size() / 3);
				const Point *pos = mesh->getVertexPositions(); const Normal *nrm = mesh->getVertexNormals();   /* trimesh.h:110,115 */
				for (size_t v = 0; v < mesh->getVertexCount(); ++v) {
					vtxPos.push_back((float) pos[v].x); vtxPos.push_back((float) pos[v].y); vtxPos.push_back((float) pos[v].z);
					vtxNrm.push_back(nrm ? (float) nrm[v].x : 0.0f); vtxNrm.push_back(nrm ? (float) nrm[v].y : 0.0f); vtxNrm.push_back(nrm ? (float) nrm[v].z : 0.0f);
				}
				const Triangle *tris = mesh->getTriangles();                          /* trimesh.h:105; Triangle::idx[3], triangle.h */
				for (size_t t = 0; t < mesh->getTriangleCount(); ++t) {
					for (int k = 0; k < 3; ++k) triIdx.push_back(base + tris[t].idx[k]);
					/* TriAccel rebuilt exactly as ShapeKDTree::build does (skdtree.cpp:77-91) with the public load() */
					TriAccel ta;
					ta.load(pos[tris[t].idx[0]], pos[tris[t].idx[1]], pos[tris[t].idx[2]]);      /* triaccel.h:63 */
					ta.shapeIndex = (uint32_t) s; ta.primIndex = (uint32_t) t;
					const uint32_t *w = reinterpret_cast<const uint32_t *>(&ta);                  /* 12 dwords, triaccel.h:34-48 */
					triaccel.insert(triaccel.end(), w, w + 12);
				}
				shapeTriOffset.push_back(shapeTriOffset.back() + (uint32_t) mesh->getTriangleCount());
				shapeFlags.push_back(nrm ? MTSGPU_SHAPE_HAS_NORMALS : 0u);
				shapeType.push_back(MTSGPU_SHAPE_TRIMESH);
				if (shape->isLuminaire()) {
					/* triangle-area CDF of the emitter (trimesh.cpp:279-283) */
					const int l = lumIndex[shape->getLuminaire()];
					lumShape[l] = (int32_t) s;
					lumInvArea[l] = (float) (1 / mesh->getSurfaceArea());             /* shape.h; TriMesh::m_invSurfaceArea */
				}
			} else if (shape->getClass()->getName() == "Sphere") {
				/* one kd-tree primitive known by its AABB (skdtree.cpp:54-57,92-96).  src/shapes/sphere.cpp keeps centre,
				 * radius and transform private: read back from serialize() (:72-78) with the parent taken off.  The shape's
				 * BSDF and luminaire are nested in that stream (Shape::serialize, shape.cpp:130-138); they are skipped by
				 * seeking from the END, where the sphere's own fields have a fixed size. */
				ref<MemoryStream> st = serializedDetached(const_cast<Shape *>(shape));
				float *SP = &shapeParams[(size_t) MTSGPU_SHAPE_NPARAMS * s];
				std::string err;
				if (!mtsgpu_stream::parseSphere<Float>(st->getData(), st->getSize(), SP, &err)) SLog(EError, "gpupath: %s", err.c_str());
				/* its primitive: a TriAccel row with k = KNoTriangleFlag (skdtree.cpp:92-96), no vertices */
				TriAccel ta; memset(&ta, 0, sizeof(ta));
				ta.k = KNoTriangleFlag; ta.shapeIndex = (uint32_t) s; ta.primIndex = 0;       /* triaccel.h:34-48,28 */
				const uint32_t *w = reinterpret_cast<const uint32_t *>(&ta);
				triaccel.insert(triaccel.end(), w, w + 12);
				for (int k = 0; k < 3; ++k) triIdx.push_back(0xFFFFFFFFu);
				shapeTriOffset.push_back(shapeTriOffset.back() + 1);
				shapeFlags.push_back(0u);
				shapeType.push_back(MTSGPU_SHAPE_SPHERE);
				if (shape->isLuminaire()) {
					const int l = lumIndex[shape->getLuminaire()];
					lumShape[l] = (int32_t) s;
					lumInvArea[l] = SP[23];
				}
			} else {
				SLog(EError, "gpupath: shape class %s is not on this path (triangle meshes of any loader and spheres are)",
					shape->getClass()->getName().c_str());
			}
		}

		/* --- per-emitter triangle CDFs (DiscretePDF over triangle areas, trimesh.cpp:279-283) and the
		 *     luminaire selection CDF (weight = getSamplingWeight(), scene.cpp:320-330) --- */
		lumCdfOffset.push_back(0);
		for (size_t l = 0; l < lums.size(); ++l) {
			const int s = lumShape[l];
			if (s >= 0 && shapeType[s] == MTSGPU_SHAPE_TRIMESH) {
				const TriMesh *mesh = static_cast<const TriMesh *>(shapes[s]);
				/* DiscretePDF::build (pdf.h:82-95) restated: m_cdf is private */
				const Point *pos = mesh->getVertexPositions(); const Triangle *tris = mesh->getTriangles();
				const size_t n = mesh->getTriangleCount(), base = lumTriCdf.size();
				lumTriCdf.push_back(0.0f);
				for (size_t t = 0; t < n; ++t) lumTriCdf.push_back(lumTriCdf.back() + (float) tris[t].surfaceArea(pos));   /* triangle.h:65 */
				const float sum = lumTriCdf.back();
				for (size_t k = 0; k < n; ++k) lumTriCdf[base + k] /= sum;
				lumTriCdf[base + n] = 1.0f;
			}
			lumCdfOffset.push_back((uint32_t) lumTriCdf.size());
		}
		{
			/* Scene::m_luminairePDF (scene.cpp:320-330): weight = getSamplingWeight() (luminaire.h:180), DiscretePDF::build */
			lumSelCdf.push_back(0.0f);
			for (size_t l = 0; l < lums.size(); ++l) lumSelCdf.push_back(lumSelCdf.back() + (float) lums[l]->getSamplingWeight());
			sc.lum_sel_sum = lumSelCdf.back();
			for (size_t l = 0; l < lums.size(); ++l) {
				lumSelCdf[l] /= sc.lum_sel_sum;
				lumSelPdf.push_back((float) lums[l]->getSamplingWeight() / sc.lum_sel_sum);
			}
			lumSelCdf[lums.size()] = 1.0f;
		}

		/* --- the SAH kd-tree as Scene::initialize built it: m_nodes / m_indices are public through
		 *     sahkdtree3.h:106-108 (`using Parent::m_nodes; using Parent::m_indices;`).  The counts are protected
		 *     (gkdtree.h:2630-2631), so the node array is walked: children are adjacent and follow their parent
		 *     (gkdtree.h:1068-1138), the root is m_nodes[0]. --- */
		{
			typedef ShapeKDTree::KDNode KDNode;
			const KDNode *nodes = kd->m_nodes;
			uint32_t nNodes = 1, nIdx = 0;
			std::vector<uint32_t> stack(1, 0u);
			while (!stack.empty()) {
				const uint32_t i = stack.back(); stack.pop_back();
				const KDNode &n = nodes[i];
				if (n.isLeaf()) { nIdx = std::max(nIdx, (uint32_t) n.getPrimEnd()); continue; }          /* gkdtree.h:520-535 */
				const uint32_t left = (uint32_t) (n.getLeft() - nodes);                                    /* gkdtree.h:541-544 */
				nNodes = std::max(nNodes, left + 2);
				stack.push_back(left); stack.push_back(left + 1);
			}
			kdNodes.resize(2 * (size_t) nNodes);
			memcpy(&kdNodes[0], nodes, sizeof(KDNode) * nNodes);                     /* 8 bytes per node, relative offsets kept */
			kdIndices.assign(kd->m_indices, kd->m_indices + nIdx);
			sc.n_nodes = nNodes; sc.n_indices = nIdx;
			const AABB &box = kd->getAABB();                                         /* already enlarged, gkdtree.h:1170-1176 */
			for (int a = 0; a < 3; ++a) { sc.aabb_min[a] = (float) box.min[a]; sc.aabb_max[a] = (float) box.max[a]; }
		}

		sc.n_shapes = (uint32_t) shapes.size(); sc.n_tris = shapeTriOffset.back(); sc.n_verts = (uint32_t) (vtxPos.size() / 3);
		sc.vtx_pos = ptr(vtxPos); sc.vtx_nrm = ptr(vtxNrm); sc.tri_idx = ptr(triIdx);
		sc.shape_tri_offset = ptr(shapeTriOffset); sc.shape_bsdf = ptr(shapeBsdf); sc.shape_lum = ptr(shapeLum);
		sc.shape_flags = ptr(shapeFlags); sc.shape_type = ptr(shapeType); sc.shape_params = ptr(shapeParams);
		sc.kd_nodes = ptr(kdNodes); sc.kd_indices = ptr(kdIndices); sc.triaccel = ptr(triaccel);
		sc.n_bsdfs = (uint32_t) bsdfType.size(); sc.bsdf_type = ptr(bsdfType); sc.bsdf_params = ptr(bsdfParams);
		sc.n_lums = (uint32_t) lums.size(); sc.lum_type = ptr(lumType); sc.lum_params = ptr(lumParams); sc.lum_shape = ptr(lumShape);
		sc.lum_inv_area = ptr(lumInvArea); sc.lum_cdf_offset = ptr(lumCdfOffset); sc.lum_tri_cdf = ptr(lumTriCdf);
		sc.lum_sel_cdf = ptr(lumSelCdf); sc.lum_sel_pdf = ptr(lumSelPdf);
		sc.env_pixels = ptr(envPixels); sc.env_pdf = ptr(envPdf); sc.env_cdf = ptr(envCdf);
	}

private:
	template <typename T> static const T *ptr(const std::vector<T> &v) { return v.empty() ? NULL : &v[0]; }

	/* EnvMapLuminaire (src/luminaires/envmap.cpp): what its unserialization constructor does (:52-77) -- decode the EXR
	 * bytes serialize() carries with Mitsuba's own Bitmap, build the MIPMap with Mitsuba's own MIPMap::fromBitmap -- and
	 * what configure() does (:95-110): the luminance x sin(theta) density over level min(3, levels - 1) */
	void readEnvMap(Luminaire *lum, float *P) {
		ref<MemoryStream> st = serializedDetached(lum);
		size_t exrOffset = 0; uint32_t size = 0; std::string err;
		if (!mtsgpu_stream::parseEnvMapHeader<Float>(st->getData(), st->getSize(), P, &exrOffset, &size, &err)) SLog(EError, "gpupath: %s", err.c_str());
		ref<MemoryStream> exr = new MemoryStream(size);
		exr->write(st->getData() + exrOffset, size);                                  /* the EXR file's bytes (envmap.cpp:85-92) */
		exr->setPos(0);
		ref<Bitmap> bitmap = new Bitmap(Bitmap::EEXR, exr);                           /* bitmap.h */
		ref<MIPMap> mip = MIPMap::fromBitmap(bitmap);                                 /* mipmap.h:57 */
		sc.env_width = (uint32_t) mip->getWidth(); sc.env_height = (uint32_t) mip->getHeight();
		const Spectrum *px = mip->getImageData();                                     /* level 0, mipmap.h:78 */
		envPixels.resize((size_t) 3 * sc.env_width * sc.env_height);
		for (size_t i = 0; i < (size_t) sc.env_width * sc.env_height; ++i) rgbOf(px[i], &envPixels[3 * i]);
		const int level = std::min(3, mip->getLevels() - 1);
		const Vector2i res = mip->getLevelResolution(level);                          /* mipmap.h:84 */
		const Spectrum *coarse = mip->getImageData(level);
		sc.env_pdf_width = (uint32_t) res.x; sc.env_pdf_height = (uint32_t) res.y;
		const size_t n = (size_t) res.x * res.y;
		envPdf.resize(n); envCdf.assign(n + 1, 0.0f);
		for (int y = 0, index = 0; y < res.y; ++y) {
			const float sinFactor = std::sin(M_PI * (y + .5f) / res.y);
			for (int x = 0; x < res.x; ++x, ++index)
				envPdf[index] = (float) coarse[x + y * res.x].getLuminance() * sinFactor;
		}
		/* DiscretePDF::build (pdf.h:82-95): prefix sums, then pdf and cdf divided by the sum, last knot forced to 1 */
		for (size_t i = 0; i < n; ++i) envCdf[i + 1] = envCdf[i] + envPdf[i];
		const float sum = envCdf[n];
		for (size_t i = 0; i < n; ++i) { envPdf[i] /= sum; envCdf[i + 1] /= sum; }
		envCdf[n] = 1.0f;
	}

	/* BSDF parameter block of one BSDF instance (shared instances are stored once) */
	int bsdfOf(const BSDF *bsdf) {
		if (!bsdf) return -1;
		std::map<const BSDF *, int>::const_iterator it = bsdfIndex.find(bsdf);
		if (it != bsdfIndex.end()) return it->second;
		const int index = (int) bsdfType.size();
		bsdfIndex[bsdf] = index;
		bsdfParams.insert(bsdfParams.end(), MTSGPU_BSDF_NPARAMS, 0.0f);
		bsdfType.push_back(0);
		ref<MemoryStream> st = serializedBSDF(bsdf);
		std::string err;
		uint32_t type = 0;
		if (!mtsgpu_stream::parseBSDF<Float>(st->getData(), st->getSize(), &type, &bsdfParams[(size_t) MTSGPU_BSDF_NPARAMS * index], &err))
			SLog(EError, "gpupath: %s", err.c_str());
		bsdfType[index] = type;
		return index;
	}
};


/* What both plugins (gpupath, gpudirect) do in Integrator::render: flatten, upload, configure, render, feed the film. */
struct GPURenderDriver {
	mtsgpu_group *group;
	volatile int cancelFlag;
	std::string reduceNoteLogged;        /* the last fall-back reason that went to the log */
	std::string devices;
	uint64_t seed;

	GPURenderDriver() : group(NULL), cancelFlag(0), devices("0"), seed(0x5EED) { }
	~GPURenderDriver() { if (group) mtsgpu_group_destroy(group); }

	/* a call on the whole group reports through the group, a call on one member through that member */
	void check(int rc, mtsgpu_ctx *member = NULL) const {
		if (rc != MTSGPU_OK)
			SLog(EError, "libmtsgpu: %s", member ? mtsgpu_last_error(member) : mtsgpu_group_last_error(group));
	}

	void createGroup() {
		if (group) return;
		std::vector<int> devs;
		std::istringstream is(devices);
		for (std::string tok; std::getline(is, tok, ','); ) {
			char *end = NULL;
			const long d = strtol(tok.c_str(), &end, 10);
			if (tok.empty() || end == tok.c_str() || *end != '\0' || d < 0)
				SLog(EError, "libmtsgpu: the `devices` property must be a comma separated list of HIP device indices, got \"%s\"", devices.c_str());
			devs.push_back((int) d);
		}
		if (devs.empty()) SLog(EError, "libmtsgpu: the `devices` property names no device");
		if (mtsgpu_create_multi((int) devs.size(), &devs[0], &group) != MTSGPU_OK)
			SLog(EError, "libmtsgpu: %s", mtsgpu_last_error(NULL));              /* throws; RenderJob::run cancels the job, renderjob.cpp:126-130 */
	}

	/* direct: luminaireSamples / bsdfSamples of MIDirectIntegrator, or -1 / -1 for the path tracer */
	bool render(Scene *scene, RenderQueue *queue, const RenderJob *job, int cameraResID, int samplerResID,
			int maxDepth, int rrDepth, bool strictNormals, int luminaireSamples, int bsdfSamples) {
		ref<Scheduler> sched = Scheduler::getInstance();
		ref<Camera> camera = static_cast<Camera *>(sched->getResource(cameraResID));        /* integrator.cpp:91-96 */
		ref<Film> film = camera->getFilm();
		const Sampler *sampler = static_cast<const Sampler *>(sched->getResource(samplerResID, 0));

		createGroup();

		/* --- scene (re-uploaded per render call: the GUI edits scenes between renders) --- */
		{
			FlatScene flat(scene);
			check(mtsgpu_group_upload_scene(group, &flat.sc));
		}

		/* --- camera: raster space of the FULL film, crop window as offset + size (perspective.cpp:43-71, film.cpp:33-41) --- */
		mtsgpu_camera cam;
		memset(&cam, 0, sizeof(cam));
		const std::string camClass = camera->getClass()->getName();
		if (camClass != "PerspectiveCameraImpl" && camClass != "OrthographicCamera")
			SLog(EError, "libmtsgpu: camera class %s is not supported", camClass.c_str());
		const ProjectiveCamera *pc = static_cast<const ProjectiveCamera *>(camera.get());
		const Vector2i filmSize = film->getSize(), cropSize = film->getCropSize();              /* film.h:62-68 */
		const Point2i cropOffset = film->getCropOffset();
		{
			/* m_rasterToCamera is protected (perspective.cpp:69): rebuilt from the public projection transform the same
			 * way PerspectiveCameraImpl::configure builds it */
			const Float aspect = (Float) filmSize.x / (Float) filmSize.y;
			Transform screenToRaster;
			if (aspect >= 1.0f)                                                                  /* mapSmallerSide = true (camera.cpp:189) */
				screenToRaster = Transform::scale(Vector((Float) filmSize.x, (Float) filmSize.y, 1.0f))
					* Transform::scale(Vector(1 / (2 * aspect), -0.5f, 1.0f)) * Transform::translate(Vector(aspect, -1.0f, 0));
			else
				screenToRaster = Transform::scale(Vector((Float) filmSize.x, (Float) filmSize.y, 1.0f))
					* Transform::scale(Vector(0.5f, -0.5f * aspect, 1.0f)) * Transform::translate(Vector(1.0f, -1 / aspect, 0));
			const Transform rasterToCamera = pc->getProjectionTransform().inverse() * screenToRaster.inverse();   /* camera.h:204 */
			copyMatrix(cam.raster_to_camera, rasterToCamera.getMatrix());
			copyMatrix(cam.camera_to_world, camera->getInverseViewTransform().getMatrix());      /* camera.h:92 */
		}
		cam.near_clip = (float) pc->getNearClip(); cam.far_clip = (float) pc->getFarClip();      /* camera.h:220-223 */
		cam.kind = camClass == "OrthographicCamera" ? 1 : 0;
		if (cam.kind == 0) {
			const PerspectiveCamera *persp = static_cast<const PerspectiveCamera *>(camera.get());
			cam.aperture_radius = (float) persp->getApertureRadius(); cam.focus_depth = (float) persp->getFocusDepth();   /* camera.h:282-288 */
		}
		cam.width = cropSize.x; cam.height = cropSize.y;
		cam.crop_offset_x = cropOffset.x; cam.crop_offset_y = cropOffset.y;
		cam.film_width = filmSize.x; cam.film_height = filmSize.y;
		check(mtsgpu_group_set_camera(group, &cam));
		if (luminaireSamples >= 0) {
			for (int i = 0; i < mtsgpu_group_size(group); ++i)                                       /* direct.cpp:32-39 */
				check(mtsgpu_set_direct_integrator(mtsgpu_group_ctx(group, i), luminaireSamples, bsdfSamples), mtsgpu_group_ctx(group, i));
		} else {
			check(mtsgpu_group_set_integrator(group, maxDepth, rrDepth, strictNormals ? 1 : 0));     /* integrator.h:419-421 */
		}

		/* --- sampler: class, sampleCount, depth (ldsampler.cpp:45-57) --- */
		const std::string sname = sampler->getClass()->getName();
		const int skind = sname == "LowDiscrepancySampler" ? MTSGPU_SAMPLER_LD_KEYED
			: sname == "StratifiedSampler" ? MTSGPU_SAMPLER_STRATIFIED_KEYED
			: sname == "HaltonSequence" ? MTSGPU_SAMPLER_HALTON
			: sname == "HammersleySequence" ? MTSGPU_SAMPLER_HAMMERSLEY
			: MTSGPU_SAMPLER_INDEPENDENT_KEYED;
		const int depth = sampler->getProperties().getInteger("depth", 3);
		check(mtsgpu_group_set_sampler(group, skind, (uint32_t) sampler->getSampleCount(), depth, seed));

		/* --- the film's reconstruction filter: the 16x16 table Film::getTabulatedFilter() holds (rfilter.h:65-102) --- */
		const TabulatedFilter *tf = film->getTabulatedFilter();                                   /* film.h:78 */
		float table[256];
		for (int y = 0; y < 16; ++y) for (int x = 0; x < 16; ++x) table[16 * y + x] = (float) tf->lookup(x, y);
		const bool box = tf->getName() == "BoxFilter";                                            /* border-free fast path */
		check(mtsgpu_group_set_rfilter(group, (float) tf->getFilterSize().x, (float) tf->getFilterSize().y, box ? NULL : table));
		for (int i = 0; i < mtsgpu_group_size(group); ++i)
			check(mtsgpu_set_film_edges(mtsgpu_group_ctx(group, i), film->hasHighQualityEdges() ? 1 : 0), mtsgpu_group_ctx(group, i));   /* film.h:75 */

		/* --- render: tiles sharded over the GPUs, films summed on GPU 0 (renderproc.cpp:123-130 in one collective) --- */
		cancelFlag = 0;
		const int bs = scene->getBlockSize();                                                     /* scene.h:543 */
		const int rc = mtsgpu_group_render(group, bs, /* ordered_reduce = */ box ? 0 : 1, &cancelFlag);
		if (rc == MTSGPU_ECANCEL) return false;
		if (rc != MTSGPU_OK) SLog(EError, "libmtsgpu: %s", mtsgpu_group_last_error(group));
		/* a collective that could not be used is not an error (the films were added up in member order instead), but
		 * the user should learn why the xGMI reduce did not run */
		const char *note = mtsgpu_group_reduce_note(group);
		if (note && note[0] && reduceNoteLogged != note) {          /* once per reason, not once per frame */
			SLog(EWarn, "libmtsgpu: film reduce fell back to the ordered sum: %s", note);
			reduceNoteLogged = note;
		}

		/* --- hand the film back as ImageBlocks: every Film plugin (exrfilm, pngfilm, mfilm) and the GUI keep working.
		 *     The sums of the crop window go out as border-less blocks: the filter has been applied already. --- */
		std::vector<float> rgbaw((size_t) cam.width * cam.height * 5);
		check(mtsgpu_read_film(mtsgpu_group_ctx(group, 0), &rgbaw[0]), mtsgpu_group_ctx(group, 0));
		for (int y0 = 0; y0 < cam.height; y0 += bs) for (int x0 = 0; x0 < cam.width; x0 += bs) {
			ref<FilmBlock> block = new FilmBlock(Vector2i(bs, bs));
			block->setOffset(Point2i(x0 + cropOffset.x, y0 + cropOffset.y));                     /* imageblock.h:261 */
			block->setSize(Vector2i(std::min(bs, cam.width - x0), std::min(bs, cam.height - y0)));
			block->clear();
			size_t idx = 0;
			for (int y = 0; y < block->getSize().y; ++y) for (int x = 0; x < block->getSize().x; ++x, ++idx) {
				const float *p = &rgbaw[5 * ((size_t) (y0 + y) * cam.width + (x0 + x))];
				Spectrum s; s.fromLinearRGB(p[0], p[1], p[2]);
				block->setPixel(idx, s); block->setAlpha(idx, p[3]); block->setWeight(idx, p[4]);   /* spectrum, alpha, weight sums */
			}
			film->putImageBlock(block);                                                           /* renderproc.cpp:126 */
			queue->signalWorkEnd(job, block);                                                     /* renderproc.cpp:128 */
		}
		return true;
	}
};

} /* namespace */

MTS_NAMESPACE_END
#endif /* MTSGPU_GPUCOMMON_H */
