// ctx.h -- the context behind the C ABI (include/mtsgpu.h), shared by api.cpp and group.cpp.
#pragma once
#include "host.h"
#include "kernels.h"
#include <map>
#include <string>
#include <utility>
#include <vector>

struct mtsgpu_ctx {
	int device = 0;
	uint32_t nCUs = 256;                   // hipDeviceProp_t::multiProcessorCount
	hipStream_t stream = nullptr;
	bool ownStream = false;
	// second stream: the shadow rays of bounce b are traced on it while `stream` already traces the closest hits of
	// bounce b + 1 (both only depend on the shading of bounce b); evShade / evShadow order the two
	hipStream_t stream2 = nullptr;
	hipEvent_t evShade[2] = { nullptr, nullptr }, evShadow[2] = { nullptr, nullptr }, evCount = nullptr;
	std::string error;

	// scene
	bool haveScene = false;
	mg::DScene dsc{};
	std::vector<void *> sceneAllocs;
	uint32_t nTris = 0;

	// configuration
	bool haveCamera = false;
	mtsgpu_camera cam{};
	int maxDepth = -1, rrDepth = 10, strictNormals = 0;
	int samplerKind = MTSGPU_SAMPLER_INDEPENDENT_KEYED;
	uint32_t spp = 4; int ldDepth = 3; uint64_t seed = 0;
	int blockSize = 32, part = 0, nParts = 1;
	uint64_t maxPaths = 0; bool countTraversal = false, timeKernels = false;
	std::map<std::string, long> tuning;      // mtsgpu_set_tuning

	// film
	float *film = nullptr; bool ownFilm = false; size_t filmPixels = 0;
	float filtSizeX = 0.5f, filtSizeY = 0.5f; int filtBorder = 0;
	bool hqEdges = false;
	int integrator = 0, nLumSamples = 1, nBsdfSamples = 1;
	float *filtValues = nullptr;           // device [16][16]
	mg::TileMeta *tileMeta = nullptr; size_t tileMetaCap = 0;
	float *blocks = nullptr; size_t blocksCap = 0;

	// per-pass buffers
	size_t pathCap = 0;
	mg::DPaths paths{};
	mg::DQueues q{};
	uint32_t *queueA = nullptr, *queueB = nullptr;
	float4 *rayqA[2] = { nullptr, nullptr }, *rayqB[2] = { nullptr, nullptr };      // the rays of queueA / queueB in queue order (o, d)
	uint4 *binHits = nullptr;               // DQueues::bin_hits: the hits of binned paths, next to their ids
	uint32_t *pixelList = nullptr; size_t pixelListCap = 0;
	// The work units of a frame (pixel keys in tile order + tile rectangles) depend on the film geometry and the tile
	// sharding only: they are kept from frame to frame (a 1-spp frame spent 0.55 of its 10.6 ms rebuilding and uploading
	// them).  renderKey is what they were built for; the test hooks that borrow pixelList clear renderListValid.
	std::vector<uint32_t> renderPixels; std::vector<mg::TileMeta> renderTiles;
	std::vector<long long> renderKey; bool renderListValid = false;
	uint32_t *ldScr = nullptr; uint16_t *ldPerm = nullptr; size_t ldScrCap = 0, ldPermCap = 0;
	uint16_t *ldScratch = nullptr; size_t ldScratchCap = 0;      // the tables of a pass while they are shuffled (kernels.h)
	// Sampler::request2DArray arrays of the direct integrator (per pass, like the tables above)
	unsigned long long *ldState = nullptr; size_t ldStateCap = 0;
	uint32_t *arrScr = nullptr; uint16_t *arrPerm = nullptr; float2 *arrPts = nullptr; size_t arrScrCap = 0, arrPermCap = 0, arrPtsCap = 0;
	float4 *primSave = nullptr; size_t primSaveCap = 0;
	uint16_t *primes = nullptr;        // primeTable (util.cpp:64-122) on the device
	uint32_t *explicitSamples = nullptr; size_t explicitCap = 0;
	uint32_t *hostCounters = nullptr;       // pinned
	uint32_t *counterSets = nullptr;        // kCounterSets x kNumCounters lines, used by alternate bounces
	uint32_t *spillClosest = nullptr, *spillShadow = nullptr;   // the two traversal kernels run concurrently
	mg::BinView *viewsDev = nullptr;        // device-driven bounces: per-bin views written by k_prep
	unsigned long long *devStats = nullptr; // kStat* counters of a device-driven frame
	bool devStatsUsed = false;              // a device-driven pass ran since the statistics were cleared
	uint32_t binMask = 0x1FFu;              // bins that can be non-empty with the uploaded scene (BSDF types present + terminal)
	unsigned long long *pathLen = nullptr;  // device: sum of the final path depths of a render (avgPathLength)
	std::vector<void *> pathAllocs;

	// stats
	mtsgpu_stats stats{};
	std::vector<std::pair<hipEvent_t, hipEvent_t>> traceEvents, shadeEvents;
	size_t traceEvUsed = 0, shadeEvUsed = 0;
	std::vector<unsigned char> traceEvClass;      // per traversal launch: 0 closest-hit, 1 closest-hit of a pass's first bounce, 2 any-hit
	// what mtsgpu_replay_roof needs to know about the pass rendered last: how to generate its camera rays again, and
	// how many slots of the shadow queue hold rays (the largest shadow launch of the pass; 0 after device-driven bounces)
	struct LastPass { bool valid = false; mg::DConfig cfg{}; size_t base = 0; uint32_t nSlots = 0, nPaths = 0, shadowMax = 0; } lastPass;
};


namespace mg {
extern thread_local std::string g_lastError;
// records the message (thread-local and in the context) and returns `code`
int fail(mtsgpu_ctx *ctx, int code, const char *fmt, ...);
}

#define HIPCHK(ctx, expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	return mg::fail(ctx, MTSGPU_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
