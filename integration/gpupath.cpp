/*
 * gpupath.cpp -- the reference-side binding: Mitsuba 0.2.1 integrator plugin `gpupath` over libmtsgpu's C ABI
 * (include/mtsgpu.h).  Drop it into src/integrators/gpupath/ of the reference tree and add to
 * src/integrators/SConscript:
 *
 *     plugins += env.SharedLibrary('#plugins/gpupath', ['gpupath/gpupath.cpp'],
 *                                  CPPPATH = env['CPPPATH'] + ['<repo>/include'],
 *                                  LIBPATH = env['LIBPATH'] + ['<repo>/mitsuba-renderer_amd'], LIBS = env['LIBS'] + ['mtsgpu'])
 *
 * A scene then says <integrator type="gpupath"> where it said <integrator type="path">; properties maxDepth,
 * rrDepth, strictNormals as before (MonteCarloIntegrator, src/librender/integrator.cpp:272-292), plus
 *     devices  (string, default "0")   comma separated HIP device indices; more than one -> mtsgpu_create_multi
 *     seed     (integer, default 0x5EED)  key of the per-(pixel, sample) sampler streams
 *
 * This file is NOT compiled in this repository's image (Mitsuba's headers need Boost, include/mitsuba/core/util.h:22);
 * every call into Mitsuba is annotated with the file:line of the declaration it relies on.  What it does:
 *
 *   preprocess()  creates the nested CPU `path` integrator (m_cpu) that serves Li() when another integrator
 *                 (irrcache, errctrl: src/integrators/misc) nests this one, and forwards preprocess to it.
 *   render()      flattens Scene / ShapeKDTree / TriMesh / BSDFs / luminaires into a mtsgpu_scene (FlatScene below),
 *                 uploads it to every GPU of the group, renders the crop window of the film with the tiles sharded over
 *                 the GPUs, has the films summed on GPU 0 (RCCL), and feeds the result to the host Film as
 *                 ImageBlocks through Film::putImageBlock + RenderQueue::signalWorkEnd, as
 *                 BlockedRenderProcess::processResult does (src/librender/renderproc.cpp:123-130).
 *   cancel()      sets the flag the library polls between wavefront stages.
 */
#include <mitsuba/render/scene.h>
#include <mitsuba/render/renderproc.h>
#include <mitsuba/render/renderjob.h>
#include <mitsuba/render/imageblock.h>
#include <mitsuba/render/triaccel.h>
#include <mitsuba/render/texture.h>
#include <mitsuba/core/mstream.h>
#include <mitsuba/core/plugin.h>
#include <mitsuba/core/sched.h>
#include <mtsgpu.h>
#include <sstream>

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

inline void copyMatrix(float *dst, const Matrix4x4 *m) {
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) dst[4 * i + j] = (float) m->m[i][j];     /* matrix layout: transform.h */
}

/* Reads back what <BSDF class>::serialize wrote -- the only access to the BSDF plugins' private parameters.  The
 * object goes through InstanceManager::serialize into memory (wire format, src/libcore/serialization.cpp:72-85: a new
 * object is [id][class name][its serialize()], a known one just [id], NULL is [0]) and the fields are read back in the
 * order the plugin's serialize() documents.  Nothing is re-instantiated.  This works for BSDFs because they have no
 * parent (BSDF::setParent is empty, bsdf.cpp:55-57) and their only children are textures; shapes and luminaires point
 * back at the scene and cannot be read this way -- their values come from the public evaluation API instead. */
class BSDFParamReader {
public:
	BSDFParamReader(const BSDF *bsdf) {
		m_stream = new MemoryStream();                                 /* include/mitsuba/core/mstream.h:40 */
		ref<InstanceManager> writer = new InstanceManager();
		writer->serialize(m_stream, bsdf);
		m_stream->setPos(0);
		if (!openObject(m_className))
			Log(EError, "gpupath: cannot read back the parameters of a BSDF");
		/* BSDF::serialize (bsdf.cpp:50-53): ConfigurableObject::serialize = the parent (none), then m_name */
		skipReference(); m_stream->readString();
	}
	const std::string &className() const { return m_className; }
	Float readFloat() { return m_stream->readFloat(); }
	Spectrum readSpectrum() { return Spectrum(m_stream); }
	/* a texture child: [id][class name][Texture::serialize = parent reference][ConstantSpectrumTexture: the value]
	 * (src/librender/texture.cpp:39-41,89-93).  Anything but a constant needs computePartials + MIPMap: out of scope. */
	Spectrum readConstantTexture(const char *what) {
		std::string cls;
		const unsigned int id = m_stream->readUInt();
		if (id != 0 && m_values.count(id)) return m_values[id];         /* one texture shared by two slots */
		if (id == 0) Log(EError, "gpupath: %s of %s is missing", what, m_className.c_str());
		m_seen.insert(id);
		cls = m_stream->readString();
		if (cls != "ConstantSpectrumTexture")                               /* consttexture.h:28-60 */
			Log(EError, "gpupath: %s of %s is a %s; only constant reflectances are supported", what, m_className.c_str(), cls.c_str());
		skipReference();
		return m_values[id] = Spectrum(m_stream);
	}
	/* the nested BRDF of a `twosided` adapter: positions the reader on its fields */
	void enterNestedBSDF() {
		if (!openObject(m_className)) Log(EError, "gpupath: twosided BRDF without a nested BRDF");
		skipReference(); m_stream->readString();
	}
private:
	bool openObject(std::string &cls) {
		const unsigned int id = m_stream->readUInt();
		if (id == 0 || m_seen.count(id)) return false;
		m_seen.insert(id);
		cls = m_stream->readString();
		return true;
	}
	void skipReference() {
		const unsigned int id = m_stream->readUInt();
		if (id != 0 && !m_seen.count(id))
			Log(EError, "gpupath: unexpected nested object in the serialized form of %s", m_className.c_str());
	}
	ref<MemoryStream> m_stream;
	std::set<unsigned int> m_seen;
	std::map<unsigned int, Spectrum> m_values;
	std::string m_className;
};

/* ------------------------------------------------------------------------------------------------------------
 * FlatScene: Scene + ShapeKDTree + TriMesh + BSDF / luminaire parameter blocks -> mtsgpu_scene
 * ---------------------------------------------------------------------------------------------------------- */
struct FlatScene {
	mtsgpu_scene sc;
	std::vector<float> vtxPos, vtxNrm, shapeParams, bsdfParams, lumParams, lumInvArea, lumTriCdf, lumSelCdf, lumSelPdf;
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
			const Luminaire *lum = lums[l];
			lumIndex[lum] = (int) l;
			float *P = &lumParams[(size_t) MTSGPU_LUM_NPARAMS * l];
			const std::string cls = lum->getClass()->getName();
			/* luminaires point back at their shape / the scene, so serialize() would drag the whole scene along: their
			 * (few) parameters are read through the public evaluation interface, which returns the stored values as is */
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
			} else {
				Log(EError, "gpupath: luminaire class %s is not mapped by this plugin (area and constant are; libmtsgpu itself "
					"also implements point, spot, directional, collimated and envmap through mtsgpu_scene)", cls.c_str());
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
			} else {
				/* libmtsgpu renders spheres (MTSGPU_SHAPE_SPHERE), but src/shapes/sphere.cpp keeps centre, radius and
				 * transform private and a Shape's serialize() drags its parent scene along: mapping it needs three
				 * accessors added to that plugin, which a drop-in does not do */
				Log(EError, "gpupath: shape class %s is not supported by this plugin (triangle meshes of any loader are)",
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
	}

private:
	template <typename T> static const T *ptr(const std::vector<T> &v) { return v.empty() ? NULL : &v[0]; }

	/* BSDF parameter block of one BSDF instance (shared instances are stored once) */
	int bsdfOf(const BSDF *bsdf) {
		if (!bsdf) return -1;
		std::map<const BSDF *, int>::const_iterator it = bsdfIndex.find(bsdf);
		if (it != bsdfIndex.end()) return it->second;
		const int index = (int) bsdfType.size();
		bsdfIndex[bsdf] = index;
		bsdfParams.insert(bsdfParams.end(), MTSGPU_BSDF_NPARAMS, 0.0f);
		bsdfType.push_back(0);
		uint32_t flags = 0;
		BSDFParamReader rd(bsdf);
		float *P = &bsdfParams[(size_t) MTSGPU_BSDF_NPARAMS * index];
		if (rd.className() == "TwoSidedBRDF") {                                  /* twosided.cpp: the nested BRDF follows */
			flags |= MTSGPU_BSDF_TWOSIDED;
			rd.enterNestedBSDF();
		}
		const std::string cls = rd.className();
		if (cls == "Lambertian") {                                               /* lambertian.cpp: reflectance texture */
			bsdfType[index] = MTSGPU_BSDF_LAMBERTIAN | flags;
			rgbOf(rd.readConstantTexture("reflectance"), P);
		} else if (cls == "Dielectric") {                                        /* dielectric.cpp:88-95 */
			bsdfType[index] = MTSGPU_BSDF_DIELECTRIC | flags;
			P[0] = (float) rd.readFloat(); P[1] = (float) rd.readFloat();
			rgbOf(rd.readConstantTexture("specularReflectance"), P + 2);
			rgbOf(rd.readConstantTexture("specularTransmittance"), P + 5);
		} else if (cls == "RoughMetal") {                                        /* roughmetal.cpp:169-176 */
			bsdfType[index] = MTSGPU_BSDF_ROUGHMETAL | flags;
			rgbOf(rd.readConstantTexture("specularReflectance"), P + 7);
			P[0] = (float) rd.readFloat();
			rgbOf(rd.readSpectrum(), P + 1); rgbOf(rd.readSpectrum(), P + 4);
		} else if (cls == "Microfacet") {                                        /* microfacet.cpp:283-293 */
			bsdfType[index] = MTSGPU_BSDF_MICROFACET | flags;
			rgbOf(rd.readConstantTexture("diffuseReflectance"), P + 5);
			rgbOf(rd.readConstantTexture("specularReflectance"), P + 8);
			for (int k = 0; k < 5; ++k) P[k] = (float) rd.readFloat();         /* alphaB, kd, ks, intIOR, extIOR */
		} else if (cls == "Mirror") {                                            /* mirror.cpp:51-55 */
			bsdfType[index] = MTSGPU_BSDF_MIRROR | flags;
			rgbOf(rd.readSpectrum(), P);
		} else {
			Log(EError, "gpupath: BSDF class %s is not mapped by this plugin (lambertian, dielectric, roughmetal, microfacet, mirror "
				"and the twosided adapter are; libmtsgpu itself also implements phong, roughglass and difftrans)", cls.c_str());
		}
		return index;
	}
};

} /* namespace */

class GPUPathTracer : public MonteCarloIntegrator {
public:
	GPUPathTracer(const Properties &props) : MonteCarloIntegrator(props), m_group(NULL), m_cancel(0) {
		m_devices = props.getString("devices", "0");
		m_seed = (uint64_t) props.getLong("seed", 0x5EED);
	}
	/* unserialization ctor: MTS_IMPLEMENT_CLASS_S needs it (class.h:173-180); mtssrv nodes get the plugin this way */
	GPUPathTracer(Stream *stream, InstanceManager *manager)
		: MonteCarloIntegrator(stream, manager), m_group(NULL), m_cancel(0) {
		m_devices = stream->readString();
		m_seed = stream->readULong();
	}
	virtual ~GPUPathTracer() { if (m_group) mtsgpu_group_destroy(m_group); }

	void serialize(Stream *stream, InstanceManager *manager) const {
		MonteCarloIntegrator::serialize(stream, manager);
		stream->writeString(m_devices);
		stream->writeULong(m_seed);
	}

	/* Scene::configure -> integrator->configureSampler(sampler) (scene.cpp:251): the path tracer requests nothing */
	void configureSampler(Sampler *sampler) { MonteCarloIntegrator::configureSampler(sampler); }

	/* Scene::preprocess -> Integrator::preprocess (scene.cpp:343), after Scene::initialize() built the kd-tree */
	bool preprocess(const Scene *scene, RenderQueue *queue, const RenderJob *job,
			int sceneResID, int cameraResID, int samplerResID) {
		if (!MonteCarloIntegrator::preprocess(scene, queue, job, sceneResID, cameraResID, samplerResID))
			return false;
		/* nested use: irrcache / errctrl call Li() per ray from the scheduler's worker threads (integrator.h:287,
		 * src/integrators/misc): the scalar CPU implementation is the reference's own `path` plugin */
		Properties props("path");
		props.setInteger("maxDepth", m_maxDepth); props.setInteger("rrDepth", m_rrDepth);
		props.setBoolean("strictNormals", m_strictNormals);
		m_cpu = static_cast<SampleIntegrator *>(PluginManager::getInstance()->createObject(MTS_CLASS(Integrator), props));   /* plugin.cpp:142-209 */
		m_cpu->configure();
		return m_cpu->preprocess(scene, queue, job, sceneResID, cameraResID, samplerResID);
	}

	Spectrum Li(const RayDifferential &r, RadianceQueryRecord &rRec) const { return m_cpu->Li(r, rRec); }

	/* Scene::render -> Integrator::render (scene.cpp:356-359), on the RenderJob thread */
	bool render(Scene *scene, RenderQueue *queue, const RenderJob *job,
			int sceneResID, int cameraResID, int samplerResID) {
		ref<Scheduler> sched = Scheduler::getInstance();
		ref<Camera> camera = static_cast<Camera *>(sched->getResource(cameraResID));        /* integrator.cpp:91-96 */
		ref<Film> film = camera->getFilm();
		const Sampler *sampler = static_cast<const Sampler *>(sched->getResource(samplerResID, 0));

		/* --- the GPUs --- */
		if (!m_group) {
			std::vector<int> devs;
			std::istringstream is(m_devices);
			for (std::string tok; std::getline(is, tok, ','); ) devs.push_back(atoi(tok.c_str()));
			if (mtsgpu_create_multi((int) devs.size(), &devs[0], &m_group) != MTSGPU_OK)
				Log(EError, "gpupath: %s", mtsgpu_last_error(NULL));                          /* throws; RenderJob::run cancels the job, renderjob.cpp:126-130 */
		}

		/* --- scene (re-uploaded per render call: the GUI edits scenes between renders) --- */
		{
			FlatScene flat(scene);
			check(mtsgpu_group_upload_scene(m_group, &flat.sc));
		}

		/* --- camera: raster space of the FULL film, crop window as offset + size (perspective.cpp:43-71, film.cpp:33-41) --- */
		mtsgpu_camera cam;
		memset(&cam, 0, sizeof(cam));
		const std::string camClass = camera->getClass()->getName();
		if (camClass != "PerspectiveCameraImpl" && camClass != "OrthographicCamera")
			Log(EError, "gpupath: camera class %s is not supported", camClass.c_str());
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
		check(mtsgpu_group_set_camera(m_group, &cam));
		check(mtsgpu_group_set_integrator(m_group, m_maxDepth, m_rrDepth, m_strictNormals ? 1 : 0));   /* integrator.h:419-421 */

		/* --- sampler: class, sampleCount, depth (ldsampler.cpp:45-57) --- */
		const std::string sname = sampler->getClass()->getName();
		const int skind = sname == "LowDiscrepancySampler" ? MTSGPU_SAMPLER_LD_KEYED
			: sname == "StratifiedSampler" ? MTSGPU_SAMPLER_STRATIFIED_KEYED
			: sname == "HaltonSequence" ? MTSGPU_SAMPLER_HALTON
			: sname == "HammersleySequence" ? MTSGPU_SAMPLER_HAMMERSLEY
			: MTSGPU_SAMPLER_INDEPENDENT_KEYED;
		const int depth = sampler->getProperties().getInteger("depth", 3);
		check(mtsgpu_group_set_sampler(m_group, skind, (uint32_t) sampler->getSampleCount(), depth, m_seed));

		/* --- the film's reconstruction filter: the 16x16 table Film::getTabulatedFilter() holds (rfilter.h:65-102) --- */
		const TabulatedFilter *tf = film->getTabulatedFilter();                                   /* film.h:78 */
		float table[256];
		for (int y = 0; y < 16; ++y) for (int x = 0; x < 16; ++x) table[16 * y + x] = (float) tf->lookup(x, y);
		const bool box = tf->getName() == "BoxFilter";                                            /* border-free fast path */
		check(mtsgpu_group_set_rfilter(m_group, (float) tf->getFilterSize().x, (float) tf->getFilterSize().y, box ? NULL : table));
		for (int i = 0; i < mtsgpu_group_size(m_group); ++i)
			check(mtsgpu_set_film_edges(mtsgpu_group_ctx(m_group, i), film->hasHighQualityEdges() ? 1 : 0));   /* film.h:75 */

		/* --- render: tiles sharded over the GPUs, films summed on GPU 0 (renderproc.cpp:123-130 in one collective) --- */
		m_cancel = 0;
		const int bs = scene->getBlockSize();                                                     /* scene.h:543 */
		const int rc = mtsgpu_group_render(m_group, bs, /* ordered_reduce = */ box ? 0 : 1, &m_cancel);
		if (rc == MTSGPU_ECANCEL) return false;
		if (rc != MTSGPU_OK) Log(EError, "gpupath: %s", mtsgpu_group_last_error(m_group));

		/* --- hand the film back as ImageBlocks: every Film plugin (exrfilm, pngfilm, mfilm) and the GUI keep working.
		 *     The sums of the crop window go out as border-less blocks: the filter has been applied already. --- */
		std::vector<float> rgbaw((size_t) cam.width * cam.height * 5);
		check(mtsgpu_read_film(mtsgpu_group_ctx(m_group, 0), &rgbaw[0]));
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

	/* Integrator::cancel arrives from another thread (scene.cpp:363-368) */
	void cancel() { m_cancel = 1; }

	std::string toString() const {
		std::ostringstream oss;
		oss << "GPUPathTracer[maxDepth=" << m_maxDepth << ", rrDepth=" << m_rrDepth << ", devices=\"" << m_devices << "\"]";
		return oss.str();
	}

	MTS_DECLARE_CLASS()
private:
	void check(int rc) const { if (rc != MTSGPU_OK) Log(EError, "gpupath: %s", mtsgpu_group_last_error(m_group)); }

	mtsgpu_group *m_group;
	volatile int m_cancel;
	std::string m_devices;
	uint64_t m_seed;
	ref<SampleIntegrator> m_cpu;
};

MTS_IMPLEMENT_CLASS_S(GPUPathTracer, false, MonteCarloIntegrator)
MTS_EXPORT_PLUGIN(GPUPathTracer, "MI355X wavefront path tracer (libmtsgpu)");        /* cobject.h:79-87 */
MTS_NAMESPACE_END
