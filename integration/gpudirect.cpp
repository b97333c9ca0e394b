/*
 * gpudirect.cpp -- the reference-side binding for `direct`: Mitsuba 0.2.1 integrator plugin `gpudirect` over libmtsgpu's
 * C ABI (mtsgpu_set_direct_integrator, include/mtsgpu.h).  Built like gpupath.cpp (see the SConscript lines there).
 *
 * A scene says <integrator type="gpudirect"> where it said <integrator type="direct">; properties as in
 * MIDirectIntegrator (src/integrators/direct/direct.cpp:32-39)
 *     luminaireSamples (integer, default 1)   samples of the luminaire sampling technique
 *     bsdfSamples      (integer, default 1)   samples of the BSDF sampling technique        (their sum must be positive)
 * plus `devices` and `seed` as in gpupath.  Counts above one draw from Sampler::next2DArray: configureSampler() requests
 * the arrays from the host sampler exactly as the reference does (direct.cpp:58-63) so that nested CPU use sees the same
 * sampler state, and the library generates its own arrays on the device for the independent, ldsampler and stratified
 * samplers; with halton / hammersley the render call fails like the reference's (halton.cpp:102-104).
 *
 * NOT compiled in this repository's image (Mitsuba's headers need Boost).
 */
#include "gpucommon.h"

MTS_NAMESPACE_BEGIN

class GPUDirectIntegrator : public SampleIntegrator {
public:
	GPUDirectIntegrator(const Properties &props) : SampleIntegrator(props) {
		m_luminaireSamples = props.getInteger("luminaireSamples", 1);              /* direct.cpp:34-36 */
		m_bsdfSamples = props.getInteger("bsdfSamples", 1);
		Assert(m_luminaireSamples >= 0 && m_bsdfSamples >= 0 && m_luminaireSamples + m_bsdfSamples > 0);   /* :38 */
		m_gpu.devices = props.getString("devices", "0");
		m_gpu.seed = (uint64_t) props.getLong("seed", 0x5EED);
	}
	GPUDirectIntegrator(Stream *stream, InstanceManager *manager) : SampleIntegrator(stream, manager) {
		m_luminaireSamples = stream->readInt();                                    /* direct.cpp:44-45 */
		m_bsdfSamples = stream->readInt();
		m_gpu.devices = stream->readString();
		m_gpu.seed = stream->readULong();
	}

	void serialize(Stream *stream, InstanceManager *manager) const {
		SampleIntegrator::serialize(stream, manager);
		stream->writeInt(m_luminaireSamples);
		stream->writeInt(m_bsdfSamples);
		stream->writeString(m_gpu.devices);
		stream->writeULong(m_gpu.seed);
	}

	/* Scene::configure -> integrator->configureSampler(sampler) (scene.cpp:251): direct.cpp:58-63 */
	void configureSampler(Sampler *sampler) {
		if (m_luminaireSamples > 1) sampler->request2DArray(m_luminaireSamples);   /* sampler.cpp:71-74 */
		if (m_bsdfSamples > 1) sampler->request2DArray(m_bsdfSamples);
	}

	bool preprocess(const Scene *scene, RenderQueue *queue, const RenderJob *job,
			int sceneResID, int cameraResID, int samplerResID) {
		if (!SampleIntegrator::preprocess(scene, queue, job, sceneResID, cameraResID, samplerResID))
			return false;
		/* nested use (irrcache, errctrl): Li() per ray from worker threads = the reference's own `direct` plugin */
		Properties props("direct");
		props.setInteger("luminaireSamples", m_luminaireSamples); props.setInteger("bsdfSamples", m_bsdfSamples);
		m_cpu = static_cast<SampleIntegrator *>(PluginManager::getInstance()->createObject(MTS_CLASS(Integrator), props));
		m_cpu->configure();
		return m_cpu->preprocess(scene, queue, job, sceneResID, cameraResID, samplerResID);
	}

	Spectrum Li(const RayDifferential &r, RadianceQueryRecord &rRec) const { return m_cpu->Li(r, rRec); }

	/* Scene::render -> Integrator::render (scene.cpp:356-359), on the RenderJob thread */
	bool render(Scene *scene, RenderQueue *queue, const RenderJob *job,
			int sceneResID, int cameraResID, int samplerResID) {
		/* MIDirectIntegrator has no maxDepth / rrDepth / strictNormals (direct.cpp:64-198) */
		return m_gpu.render(scene, queue, job, cameraResID, samplerResID, -1, 0, false, m_luminaireSamples, m_bsdfSamples);
	}

	void cancel() { m_gpu.cancelFlag = 1; }

	std::string toString() const {
		std::ostringstream oss;
		oss << "GPUDirectIntegrator[luminaireSamples=" << m_luminaireSamples << ", bsdfSamples=" << m_bsdfSamples
			<< ", devices=\"" << m_gpu.devices << "\"]";
		return oss.str();
	}

	MTS_DECLARE_CLASS()
private:
	GPURenderDriver m_gpu;
	int m_luminaireSamples, m_bsdfSamples;
	ref<SampleIntegrator> m_cpu;
};

MTS_IMPLEMENT_CLASS_S(GPUDirectIntegrator, false, SampleIntegrator)
MTS_EXPORT_PLUGIN(GPUDirectIntegrator, "MI355X direct illumination integrator (libmtsgpu)");
MTS_NAMESPACE_END
