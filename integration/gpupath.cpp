/*
 * gpupath.cpp -- the reference-side binding for `path`: Mitsuba 0.2.1 integrator plugin `gpupath` over libmtsgpu's C ABI
 * (include/mtsgpu.h).  Drop this directory's files into src/integrators/gpu/ of the reference tree and add to
 * src/integrators/SConscript:
 *
 *     gpuEnv = dict(CPPPATH = env['CPPPATH'] + ['<repo>/include'],
 *                   LIBPATH = env['LIBPATH'] + ['<repo>/mitsuba-renderer_amd'], LIBS = env['LIBS'] + ['mtsgpu'])
 *     plugins += env.SharedLibrary('#plugins/gpupath',   ['gpu/gpupath.cpp'],   **gpuEnv)
 *     plugins += env.SharedLibrary('#plugins/gpudirect', ['gpu/gpudirect.cpp'], **gpuEnv)
 *
 * A scene then says <integrator type="gpupath"> where it said <integrator type="path">; properties maxDepth,
 * rrDepth, strictNormals as before (MonteCarloIntegrator, src/librender/integrator.cpp:272-292), plus
 *     devices  (string, default "0")      comma separated HIP device indices; more than one -> mtsgpu_create_multi
 *     seed     (integer, default 0x5EED)  key of the per-(pixel, sample) sampler streams
 *
 * NOT compiled in this repository's image (Mitsuba's headers need Boost, include/mitsuba/core/util.h:22); every call
 * into Mitsuba is annotated with the file:line of the declaration it relies on.  What the class does:
 *
 *   preprocess()  creates the nested CPU `path` integrator (m_cpu) that serves Li() when another integrator
 *                 (irrcache, errctrl: src/integrators/misc) nests this one, and forwards preprocess to it.
 *   render()      GPURenderDriver::render (gpucommon.h): flattens Scene / ShapeKDTree / shapes / BSDFs / luminaires into
 *                 a mtsgpu_scene, uploads it to every GPU of the group, renders the crop window of the film with the tiles
 *                 sharded over the GPUs, has the films summed on GPU 0 (RCCL), and feeds the result to the host Film as
 *                 ImageBlocks, as BlockedRenderProcess::processResult does (src/librender/renderproc.cpp:123-130).
 *   cancel()      sets the flag the library polls between wavefront stages.
 */
#include "gpucommon.h"

MTS_NAMESPACE_BEGIN

class GPUPathTracer : public MonteCarloIntegrator {
public:
	GPUPathTracer(const Properties &props) : MonteCarloIntegrator(props) {
		m_gpu.devices = props.getString("devices", "0");
		m_gpu.seed = (uint64_t) props.getLong("seed", 0x5EED);
	}
	/* unserialization ctor: MTS_IMPLEMENT_CLASS_S needs it (class.h:173-180); mtssrv nodes get the plugin this way */
	GPUPathTracer(Stream *stream, InstanceManager *manager) : MonteCarloIntegrator(stream, manager) {
		m_gpu.devices = stream->readString();
		m_gpu.seed = stream->readULong();
	}

	void serialize(Stream *stream, InstanceManager *manager) const {
		MonteCarloIntegrator::serialize(stream, manager);
		stream->writeString(m_gpu.devices);
		stream->writeULong(m_gpu.seed);
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
		return m_gpu.render(scene, queue, job, cameraResID, samplerResID, m_maxDepth, m_rrDepth, m_strictNormals, -1, -1);
	}

	/* Integrator::cancel arrives from another thread (scene.cpp:363-368) */
	void cancel() { m_gpu.cancelFlag = 1; }

	std::string toString() const {
		std::ostringstream oss;
		oss << "GPUPathTracer[maxDepth=" << m_maxDepth << ", rrDepth=" << m_rrDepth << ", devices=\"" << m_gpu.devices << "\"]";
		return oss.str();
	}

	MTS_DECLARE_CLASS()
private:
	GPURenderDriver m_gpu;
	ref<SampleIntegrator> m_cpu;
};

MTS_IMPLEMENT_CLASS_S(GPUPathTracer, false, MonteCarloIntegrator)
MTS_EXPORT_PLUGIN(GPUPathTracer, "MI355X wavefront path tracer (libmtsgpu)");        /* cobject.h:79-87 */
MTS_NAMESPACE_END
