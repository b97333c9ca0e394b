// api.cpp -- the C ABI of libmtsgpu (include/mtsgpu.h): context, HBM residency of the
// flattened scene, and the host loop that drives the wavefront stages per bounce.
// There is no CPU fallback: every compute entry point needs a gfx950 device.
#include "host.h"
#include "kernels.h"
#include "devmath.h"
#include "ctx.h"
#include <algorithm>
#include <functional>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>

using namespace mg;

namespace mg {
thread_local std::string g_lastError;

int fail(mtsgpu_ctx *ctx, int code, const char *fmt, ...) {
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
	g_lastError = buf;
	if (ctx) ctx->error = buf;
	return code;
}
}

namespace {

template <typename T> int devAlloc(mtsgpu_ctx *ctx, T **p, size_t count, std::vector<void *> &owner) {
	void *raw = nullptr;
	HIPCHK(ctx, hipMalloc(&raw, std::max<size_t>(count, 1) * sizeof(T)));
	owner.push_back(raw);
	*p = static_cast<T *>(raw);
	return 0;
}

template <typename T> int upload(mtsgpu_ctx *ctx, const T **dst, const T *src, size_t count) {
	T *p = nullptr;
	int rc = devAlloc(ctx, &p, count, ctx->sceneAllocs);
	if (rc) return rc;
	if (count) HIPCHK(ctx, hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
	*dst = p;
	return 0;
}

void freeAll(std::vector<void *> &v) { for (void *p : v) (void) hipFree(p); v.clear(); }

// The events that order the two streams of a context against each other (shading of bounce b -> its shadow rays on the
// second stream -> shading of bounce b + 1): both streams run on one device, so a device-scope release is all they need
#ifndef MG_EV_DEVICE
#define MG_EV_DEVICE 0
#endif
constexpr unsigned kOrderEventFlags = hipEventDisableTiming | (MG_EV_DEVICE ? hipEventReleaseToDevice : 0u);

uint32_t roundToPow2(uint32_t v) { uint32_t r = 1; while (r < v) r <<= 1; return r; }

uint32_t squareAtLeast(uint32_t v) { uint32_t i = 1; while ((uint64_t) i * i < v) ++i; return i * i; }

uint32_t effectiveSpp(const mtsgpu_ctx *c) {
	if (c->samplerKind == MTSGPU_SAMPLER_LD_KEYED) return roundToPow2(c->spp);                 // ldsampler.cpp:52-57
	if (c->samplerKind == MTSGPU_SAMPLER_STRATIFIED_KEYED) return squareAtLeast(c->spp);       // stratified.cpp:36-44
	return c->spp;
}

// scheduling parameters of k_trace: the defaults, or what mtsgpu_set_tuning asked for
void applyTuning(mtsgpu_ctx *c) {
	auto get = [&](const char *key, long dflt) { auto it = c->tuning.find(key); return it == c->tuning.end() ? dflt : it->second; };
	c->q.refill_min = (uint32_t) get("refill_min", 32);
	c->q.desc_min = (uint32_t) get("desc_min", 8);
	c->q.leaf_min = (uint32_t) get("leaf_min", 8);
	c->q.tune_refill = c->tuning.count("refill_min") ? 1u : 0u;
	c->q.tune_batch = (uint32_t) get("batch", 0);
	c->q.tune_dyn_div = (uint32_t) get("dyn_div", 0);
	c->q.tune_blocks_per_cu = (uint32_t) get("blocks_per_cu", 0);
	c->q.tune_plain_below = (uint32_t) get("plain_below", 0);
	c->q.tune_dyn_min_rounds = (uint32_t) get("dyn_min_rounds", 0);
	c->q.bin_hits = c->binHits;
}

// size of the full film the crop window lies in (film.cpp:33-41); without a crop window the film itself
int filmWidth(const mtsgpu_ctx *c) { return c->cam.film_width > 0 ? c->cam.film_width : c->cam.width; }
int filmHeight(const mtsgpu_ctx *c) { return c->cam.film_height > 0 ? c->cam.film_height : c->cam.height; }

// Which part renders tile (tx, ty): the bits of tx and ty interleaved (tx lowest) modulo the number of parts, so the
// parts sample the image on a 2-D lattice (mtsgpu_set_tiles in include/mtsgpu.h; the same function lives in
// oracle/orc_render.c and filmreduce.py)
uint32_t tileMorton(uint32_t tx, uint32_t ty) {
	uint32_t m = 0;
	for (int b = 0; b < 16; ++b)
		m |= ((tx >> b) & 1u) << (2 * b) | ((ty >> b) & 1u) << (2 * b + 1);
	return m;
}

// the samplers whose generate() fills per-pixel tables (permutations, scrambles)
bool samplerHasTables(const mtsgpu_ctx *c) {
	return c->samplerKind == MTSGPU_SAMPLER_LD_KEYED || c->samplerKind == MTSGPU_SAMPLER_STRATIFIED_KEYED;
}

// Device bytes ensurePaths() allocates per path of a pass: the 128-byte record, the shadow ray (3 x 16), the nine material
// bins sized for the all-in-one-bin case with a quarter of headroom (id 4 + hit 16 bytes per entry), the two next queues
// (id 4 + ray 32 bytes each) and the shadow queue's ids -- about 480 bytes, 34 GB for the default pass of 72 M paths.
constexpr size_t kBytesPerPath = kPathSlots * 16 + 3 * 16 + (size_t) (kNumBins * (4 + 16) * 5 / 4) + 2 * (4 + 32) + 4 + 4;
// Paths per pass when the caller set none (mtsgpu_set_options max_paths == 0): 72 M, or what 60 % of the free device memory
// holds if that is less (the sampler tables, the film and the scene of a later upload need room too)
uint64_t defaultMaxPaths(mtsgpu_ctx *c) {
	uint64_t paths = 72ull << 20;
	size_t freeB = 0, totalB = 0;
	if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
		// what this context already holds for its passes is free for the next one
		const uint64_t avail = (uint64_t) freeB + (uint64_t) c->pathCap * kBytesPerPath;
		paths = std::min<uint64_t>(paths, std::max<uint64_t>(1ull << 16, avail * 6 / 10 / kBytesPerPath));
	}
	return paths;
}

int ensurePaths(mtsgpu_ctx *c, size_t cap) {
	if (cap <= c->pathCap)
		return 0;
	freeAll(c->pathAllocs);
	c->pathCap = 0;
	cap = (cap + 255) & ~(size_t) 255;
	auto &o = c->pathAllocs;
	int rc = 0;
	rc |= devAlloc(c, &c->paths.base, cap * kPathSlots, o);
	rc |= devAlloc(c, &c->paths.shq_o, cap, o); rc |= devAlloc(c, &c->paths.shq_d, cap, o); rc |= devAlloc(c, &c->paths.shq_nee, cap, o);
	// With static dealing every shard (workgroups with blockIdx % kBinShards == s) sees at most 1/kBinShards of the
	// 256-ray batches plus one per workgroup, and all of them may land in one bin.  The dynamically claimed tail of a
	// large launch (a quarter of the queue) goes to whichever waves are free, so a shard can take more than its share:
	// a quarter of headroom covers what was ever observed; k_trace drops entries beyond the capacity and
	// traceAndBin() repeats such a launch with static dealing, for which the bound holds by construction.
	const unsigned gridBlocksMax = c->nCUs * kTraceBlocksPerCuMax;
	const size_t segCap = cap / kBinShards + cap / (4 * kBinShards) + (size_t) kTraceBlock * (gridBlocksMax / kBinShards + 2);
	rc |= devAlloc(c, &c->q.bins_base, segCap * kBinShards * kNumBins, o);
	rc |= devAlloc(c, &c->binHits, segCap * kBinShards * kNumBins, o);
	applyTuning(c);                  // DQueues::bin_hits follows the allocation
	c->q.bin_stride = (uint32_t) (segCap * kBinShards);
	if (segCap * kBinShards > 0xFFFFFFFFull) return fail(c, MTSGPU_EINVAL, "pass too large");
	c->q.bin_seg_cap = (uint32_t) segCap;
	rc |= devAlloc(c, &c->queueA, cap, o); rc |= devAlloc(c, &c->queueB, cap, o);
	for (int k = 0; k < 2; ++k) { rc |= devAlloc(c, &c->rayqA[k], cap, o); rc |= devAlloc(c, &c->rayqB[k], cap, o); }
	rc |= devAlloc(c, &c->q.shadow, cap, o);
	rc |= devAlloc(c, &c->q.redo, cap, o);          // closest-hit rays on which two primitives tied (k_trace TIE): traced again with the mailbox
	rc |= devAlloc(c, &c->counterSets, (size_t) kCounterSets * kNumCounters * kCounterStride, o);
	rc |= devAlloc(c, &c->viewsDev, kNumBins, o);
	rc |= devAlloc(c, &c->devStats, kNumDevStats, o);
	rc |= devAlloc(c, &c->q.trace_counts, kNumTraceCounts, o);
	rc |= devAlloc(c, &c->spillClosest, (size_t) gridBlocksMax * kTraceBlock * trace_spill_levels(), o);
	rc |= devAlloc(c, &c->spillShadow, (size_t) gridBlocksMax * kTraceBlock * trace_spill_levels(), o);
	if (rc) return rc;
	c->q.counters = c->counterSets; c->q.spill = c->spillClosest; c->q.dev_stats = nullptr;
	c->q.spill_stride = gridBlocksMax * kTraceBlock;
	c->q.n_cus = c->nCUs; c->q.force_static = 0;
	applyTuning(c);
	HIPCHK(c, hipMemset(c->q.trace_counts, 0, kNumTraceCounts * sizeof(unsigned long long)));
	c->q.rec = c->q.rec_len = nullptr; c->q.rec_cap = 0;
	c->pathCap = cap;
	return 0;
}

template <typename T> int ensureBuf(mtsgpu_ctx *c, T **p, size_t *cap, size_t need) {
	if (need <= *cap) return 0;
	if (*p) (void) hipFree(*p);
	*p = nullptr; *cap = 0;
	void *raw = nullptr;
	HIPCHK(c, hipMalloc(&raw, need * sizeof(T)));
	*p = static_cast<T *>(raw); *cap = need;
	return 0;
}

// what launch_ld_tables needs besides the tables: one stream word per slot and, above 512 samples per pixel, the scratch
// copy the tables are shuffled in
int ensureTableWork(mtsgpu_ctx *c, size_t nSlots, uint32_t spp) {
	int rc = ensureBuf(c, &c->ldState, &c->ldStateCap, nSlots); if (rc) return rc;
	return ensureBuf(c, &c->ldScratch, &c->ldScratchCap, ld_table_scratch_entries((uint32_t) nSlots, spp, c->ldDepth));
}

DConfig makeConfig(const mtsgpu_ctx *c, bool slotPerPath) {
	DConfig cfg{};
	std::memcpy(cfg.r2c, c->cam.raster_to_camera, sizeof(cfg.r2c));
	std::memcpy(cfg.c2w, c->cam.camera_to_world, sizeof(cfg.c2w));
	cfg.near_clip = c->cam.near_clip; cfg.far_clip = c->cam.far_clip;
	cfg.aperture_radius = c->cam.aperture_radius; cfg.focus_depth = c->cam.focus_depth; cfg.camera_kind = c->cam.kind;
	cfg.width = c->cam.width; cfg.height = c->cam.height;
	cfg.crop_x = c->cam.crop_offset_x; cfg.crop_y = c->cam.crop_offset_y;
	cfg.pix_w = filmWidth(c); cfg.pix_off = 0;
	cfg.max_depth = c->maxDepth; cfg.rr_depth = c->rrDepth; cfg.strict_normals = c->strictNormals;
	cfg.integrator = c->integrator; cfg.n_lum = c->nLumSamples; cfg.n_bsdf = c->nBsdfSamples;
	// MIDirectIntegrator::configure (direct.cpp:51-56)
	cfg.weight_bsdf = 1 / (float) c->nBsdfSamples; cfg.weight_lum = 1 / (float) c->nLumSamples;
	cfg.frac_bsdf = c->nBsdfSamples / (float) (c->nLumSamples + c->nBsdfSamples);
	cfg.frac_lum = c->nLumSamples / (float) (c->nLumSamples + c->nBsdfSamples);
	cfg.sampler_kind = c->samplerKind;
	cfg.spp = effectiveSpp(c); cfg.ld_depth = c->ldDepth; cfg.seed = c->seed;
	cfg.strat_res = 1; while ((uint32_t) cfg.strat_res * (uint32_t) cfg.strat_res < cfg.spp) ++cfg.strat_res;
	cfg.slot_per_path = slotPerPath ? 1 : 0;
	cfg.ld_scr = c->ldScr; cfg.ld_perm = c->ldPerm; cfg.primes = c->primes;
	// MIDirectIntegrator::configureSampler (direct.cpp:58-63): the luminaire array first
	if (c->integrator == 1) {
		if (c->nLumSamples > 1) cfg.arr_size[cfg.arr_n++] = (uint32_t) c->nLumSamples;
		if (c->nBsdfSamples > 1) cfg.arr_size[cfg.arr_n++] = (uint32_t) c->nBsdfSamples;
		for (int a = 0; a < cfg.arr_n; ++a) { cfg.arr_off[a] = cfg.arr_total; cfg.arr_total += cfg.spp * cfg.arr_size[a]; }
	}
	cfg.arr_scr = c->arrScr; cfg.arr_perm = c->arrPerm; cfg.arr_pts = c->arrPts;
	cfg.filt_size_x = c->filtSizeX; cfg.filt_size_y = c->filtSizeY; cfg.filt_border = c->filtBorder; cfg.filt_values = c->filtValues;
	return cfg;
}

// buffers of the sample arrays for nSlots sampler slots (and of the saved camera hits for nPaths paths)
int ensureSampleArrays(mtsgpu_ctx *c, size_t nSlots, size_t nPaths) {
	if (c->integrator != 1 || (c->nLumSamples <= 1 && c->nBsdfSamples <= 1))
		return 0;
	if (c->samplerKind == MTSGPU_SAMPLER_HALTON || c->samplerKind == MTSGPU_SAMPLER_HAMMERSLEY)
		return fail(c, MTSGPU_EINVAL, "request2DArray() is not supported by QMC samplers! (halton.cpp:102-104, hammersley.cpp:112-114)");
	const uint32_t spp = effectiveSpp(c);
	size_t total = 0; int nArr = 0;
	for (int n : { c->nLumSamples, c->nBsdfSamples })
		if (n > 1) {
			if ((uint64_t) spp * (uint64_t) n > 65536ull)
				return fail(c, MTSGPU_EINVAL, "sampleCount x samples per strategy = %llu exceeds 65536 points per pixel", (unsigned long long) spp * n);
			total += (size_t) spp * n; ++nArr;
		}
	int rc = 0;
	if (c->nBsdfSamples > 1) { rc = ensureBuf(c, &c->primSave, &c->primSaveCap, 3 * nPaths); if (rc) return rc; }
	c->paths.prim = c->primSave;
	if (!samplerHasTables(c)) return 0;
	rc = ensureBuf(c, &c->ldState, &c->ldStateCap, nSlots); if (rc) return rc;
	if (c->samplerKind == MTSGPU_SAMPLER_LD_KEYED) {
		rc = ensureBuf(c, &c->arrScr, &c->arrScrCap, nSlots * nArr * 2); if (rc) return rc;
		rc = ensureBuf(c, &c->arrPerm, &c->arrPermCap, nSlots * total); if (rc) return rc;
	} else {
		rc = ensureBuf(c, &c->arrPts, &c->arrPtsCap, nSlots * total); if (rc) return rc;
	}
	return 0;
}

hipEvent_t *nextEventPair(mtsgpu_ctx *c, std::vector<std::pair<hipEvent_t, hipEvent_t>> &pool, size_t &used) {
	if (used == pool.size()) {
		hipEvent_t a, b;
		if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
		pool.emplace_back(a, b);
	}
	return &pool[used++].first;
}

long tuningOr(const mtsgpu_ctx *c, const char *key, long dflt);
// The rays of the two closest-hit queues in queue order (kernels.h: DPaths::rq_*): the traversal reads those of `cur`, the
// kernel that fills `nxt` writes them.  NULL (or the "ray_queues" knob at 0): records only.
void rayQueues(mtsgpu_ctx *c, const uint32_t *cur, const uint32_t *nxt) {
	const bool on = tuningOr(c, "ray_queues", 1) != 0;
	auto of = [&](const uint32_t *q, int k) -> float4 * { return !on || !q ? nullptr : (q == c->queueA ? c->rayqA[k] : (q == c->queueB ? c->rayqB[k] : nullptr)); };
	c->paths.rq_o = of(cur, 0); c->paths.rq_d = of(cur, 1); c->paths.rqn_o = of(nxt, 0); c->paths.rqn_d = of(nxt, 1);
}
// ... and outside the bounce loops nobody reads or writes them (test hooks and the replay measurement trace rays they put
// into the records)
struct RayQueuesOff { mtsgpu_ctx *c; ~RayQueuesOff() { rayQueues(c, nullptr, nullptr); c->q.nee_parked = 0; } };

// an event pair for a traversal launch of class cls (ctx.h: traceEvClass)
hipEvent_t *nextTraceEvents(mtsgpu_ctx *c, int cls) {
	hipEvent_t *ev = nextEventPair(c, c->traceEvents, c->traceEvUsed);
	if (ev) {
		if (c->traceEvClass.size() < c->traceEvUsed) c->traceEvClass.resize(c->traceEvUsed);
		c->traceEvClass[c->traceEvUsed - 1] = (unsigned char) cls;
	}
	return ev;
}

int readCounters(mtsgpu_ctx *c) {
	HIPCHK(c, hipMemcpyAsync(c->hostCounters, c->q.counters, kNumCounters * kCounterStride * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}


// Closest-hit launch over queue[0..n) with the material sort, then the per-bin segment sizes (one blocking read of the
// counters).  A shard segment that overflowed (possible only with dynamically claimed batches, see ensurePaths) makes
// the launch run again with static dealing: tracing a ray twice writes the same hit twice.
// pairWith (knob "merged"): the any-hit queue of the previous bounce rides in the same launch (k_trace_pair)
struct PairedShadow { const DQueues *q; uint32_t n; bool coherent; };
int traceAndBin(mtsgpu_ctx *c, const uint32_t *queue, uint32_t n, bool coherent, BinView *views, const std::function<int()> &afterLaunch = nullptr,
                const PairedShadow *pairWith = nullptr) {
	hipStream_t s = c->stream;
	const size_t counterBytes = kNumCounters * kCounterStride * sizeof(uint32_t);
	for (int attempt = 0; attempt < 2; ++attempt) {
		if (attempt) HIPCHK(c, hipMemsetAsync(c->q.counters, 0, counterBytes, s));
		c->q.force_static = attempt ? 1u : 0u;
		hipEvent_t *ev = c->timeKernels ? nextTraceEvents(c, coherent ? 1 : 0) : nullptr;
		if (ev) HIPCHK(c, hipEventRecord(ev[0], s));
		// the closest-hit kernel without the mailbox (k_trace TIE, one more level of the tree in LDS); rays on which two primitives
		// tied come back in q.redo and go through the kernel with the mailbox
		const bool tie = tuningOr(c, "mailbox_free", 0) != 0 && !c->countTraversal && !(pairWith && attempt == 0);
		if (pairWith && attempt == 0 && !c->countTraversal)
			launch_trace_pair(s, c->dsc, c->paths, c->q, queue, n, coherent, *pairWith->q, pairWith->q->shadow, pairWith->n, pairWith->coherent);
		else
			launch_trace(s, 0, c->countTraversal && attempt == 0, true, c->dsc, c->paths, c->q, queue, n, coherent, nullptr, tie);
		if (ev) HIPCHK(c, hipEventRecord(ev[1], s));
		HIPCHK(c, hipGetLastError());
		c->stats.trace_launches++;
		if (attempt == 0 && afterLaunch) { const int rc2 = afterLaunch(); if (rc2) { c->q.force_static = 0; return rc2; } }
		int rc = readCounters(c); if (rc) { c->q.force_static = 0; return rc; }
		if (tie) {
			const uint32_t nRedo = c->hostCounters[kRedoWord];
			if (nRedo > n) { c->q.force_static = 0; return fail(c, MTSGPU_EHIP, "internal: redo list longer than the queue"); }
			if (nRedo) {
				// their rays come from the path records (the list is not in queue order); statically dealt: no segment can overflow
				DPaths viaRecords = c->paths; viaRecords.rq_o = viaRecords.rq_d = nullptr;
				c->q.force_static = 1;
				hipEvent_t *ev2 = c->timeKernels ? nextTraceEvents(c, 0) : nullptr;
				if (ev2) HIPCHK(c, hipEventRecord(ev2[0], s));
				launch_trace(s, 0, false, true, c->dsc, viaRecords, c->q, c->q.redo, nRedo, false);
				if (ev2) HIPCHK(c, hipEventRecord(ev2[1], s));
				HIPCHK(c, hipGetLastError());
				c->stats.trace_launches++;
				c->stats.rays_redone += nRedo;
				rc = readCounters(c); if (rc) { c->q.force_static = 0; return rc; }
			}
		}
		c->q.force_static = 0;
		bool overflow = false;
		for (int b = 0; b < kNumBins; ++b) {
			uint32_t acc = 0;
			for (int k = 0; k < kBinShards; ++k) {
				views[b].prefix[k] = acc;
				const uint32_t cnt = c->hostCounters[(b * kBinShards + k) * kCounterStride];
				if (cnt > c->q.bin_seg_cap) overflow = true;
				acc += cnt;
			}
			views[b].prefix[kBinShards] = acc;
		}
		if (attempt == 0 && c->tuning.count("test_retry") && c->tuning["test_retry"]) overflow = true;   // exercises the retry (tests)
		if (!overflow) return 0;
		c->stats.bin_overflow_retries++;
	}
	return fail(c, MTSGPU_EHIP, "internal: bin segment overflow with static dealing");
}

// MIDirectIntegrator with more than one sample per strategy (direct.cpp:129-150,163-195): after the camera rays are
// traced and sorted by material, the two sampling loops run as rounds over those queues -- round j of the first loop
// shades with luminaire sample j and traces its shadow rays, round j of the second loop shades with BSDF sample j,
// traces the sampled rays and adds what they hit.  Li therefore receives its terms in the order of the reference
// (emission, luminaire samples 0.., BSDF samples 0..), which float addition needs for identical results.
int runDirectRounds(mtsgpu_ctx *c, const DConfig &cfg0, uint32_t nPaths, volatile const int *cancel) {
	hipStream_t s = c->stream;
	DConfig cfg = cfg0;
	const size_t counterBytes = kNumCounters * kCounterStride * sizeof(uint32_t);
	auto timedTrace = [&](int mode, bool bin, const uint32_t *queue, uint32_t n, bool coherent) -> int {
		hipEvent_t *ev = c->timeKernels ? nextTraceEvents(c, mode == 1 ? 2 : 0) : nullptr;
		if (ev) HIPCHK(c, hipEventRecord(ev[0], s));
		launch_trace(s, mode, c->countTraversal, bin, c->dsc, c->paths, c->q, queue, n, coherent);
		if (ev) HIPCHK(c, hipEventRecord(ev[1], s));
		HIPCHK(c, hipGetLastError());
		c->stats.trace_launches++;
		return 0;
	};
	HIPCHK(c, hipMemsetAsync(c->q.counters, 0, counterBytes, s));
	c->q.next = c->queueB;
	BinView views[kNumBins];
	RayQueuesOff rqOff{ c };
	rayQueues(c, c->queueA, nullptr);        // the camera rays; the rays of the sampling rounds are traced from the records
	int rc = traceAndBin(c, c->queueA, nPaths, true, views); if (rc) return rc;
	rayQueues(c, nullptr, nullptr);
	c->stats.rays_closest += nPaths;
	auto shadeRound = [&](int mode, int index, bool withTerminal) -> int {
		hipEvent_t *sev = c->timeKernels ? nextEventPair(c, c->shadeEvents, c->shadeEvUsed) : nullptr;
		HIPCHK(c, hipMemsetAsync(c->q.counters, 0, counterBytes, s));     // the bins stay as they are; views[] holds their sizes
		if (sev) HIPCHK(c, hipEventRecord(sev[0], s));
		cfg.dr_mode = mode; cfg.dr_index = index;
		for (int b = 0; b < (withTerminal ? kNumBins : kNumBsdfTypes); ++b)
			launch_shade(s, b, c->dsc, c->paths, cfg, c->q, views[b]);
		if (sev) HIPCHK(c, hipEventRecord(sev[1], s));
		HIPCHK(c, hipGetLastError());
		return readCounters(c);
	};
	// direct.cpp:122-150
	for (int j = 0; j < std::max(1, c->nLumSamples); ++j) {
		if (cancel && *cancel) return fail(c, MTSGPU_ECANCEL, "render cancelled");
		rc = shadeRound(1, j, j == 0); if (rc) return rc;
		const uint32_t nShadow = c->hostCounters[kShadowWord];
		if (nShadow) {
			rc = timedTrace(1, false, c->q.shadow, nShadow, true); if (rc) return rc;
			c->stats.rays_shadow += nShadow;
		}
	}
	// direct.cpp:152-195
	for (int j = 0; j < std::max(1, c->nBsdfSamples); ++j) {
		if (cancel && *cancel) return fail(c, MTSGPU_ECANCEL, "render cancelled");
		rc = shadeRound(2, j, false); if (rc) return rc;
		const uint32_t nNext = c->hostCounters[kNextWord];
		if (!nNext) continue;
		rc = timedTrace(0, false, c->queueB, nNext, false); if (rc) return rc;
		c->stats.rays_closest += nNext;
		// what the sampled rays hit: the material-independent tail of the shading kernel over the ray queue
		BinView tail;
		tail.prefix[0] = 0;
		for (int k = 1; k <= kBinShards; ++k) tail.prefix[k] = nNext;
		hipEvent_t *sev = c->timeKernels ? nextEventPair(c, c->shadeEvents, c->shadeEvUsed) : nullptr;
		if (sev) HIPCHK(c, hipEventRecord(sev[0], s));
		cfg.dr_mode = 3; cfg.dr_index = j;
		launch_shade(s, kNumBsdfTypes, c->dsc, c->paths, cfg, c->q, tail, nullptr, 0, c->queueB);
		if (sev) HIPCHK(c, hipEventRecord(sev[1], s));
		HIPCHK(c, hipGetLastError());
	}
	return 0;
}

uint32_t *counterSet(mtsgpu_ctx *c, int bounce) { return c->counterSets + (size_t) (bounce & 1) * kNumCounters * kCounterStride; }

long tuningOr(const mtsgpu_ctx *c, const char *key, long dflt) {
	auto it = c->tuning.find(key);
	return it == c->tuning.end() ? dflt : it->second;
}

// Device-driven bounces: the whole chain of launches is enqueued without a single read-back of a queue size.
// k_trace reads its ray count from the counter the previous stage left in device memory, k_prep turns the shard
// counters of the closest-hit launch into the per-bin views k_shade reads, grids are sized for the upper bound
// (the number of paths of the pass) and surplus workgroups exit at once.  This is for frames whose launches are too
// short to hide a host round trip (a 1-spp frame: ~0.1-0.6 ms per launch, 2 round trips per bounce otherwise); large
// frames keep the host-sized grids of runBounces.  The host looks at a queue size once per chunk of bounces, to stop.
int runBouncesDevice(mtsgpu_ctx *c, const DConfig &cfg, uint32_t nPaths, volatile const int *cancel) {
	hipStream_t s1 = c->stream, s2 = c->stream2;
	const size_t setBytes = (size_t) kNumCounters * kCounterStride * sizeof(uint32_t);
	HIPCHK(c, hipMemsetAsync(c->counterSets, 0, kCounterSets * setBytes, s1));
	c->q.dev_stats = c->devStats;        // cleared by the caller (one frame may take several passes)
	c->devStatsUsed = true;
	// MIPathTracer traces at most maxDepth rays per path (path.cpp:87), the one-sample direct integrator two
	const int limit = cfg.integrator == 1 ? 2 : (cfg.max_depth > 0 ? cfg.max_depth : 0x7FFFFFFF);
	const int chunk = (int) tuningOr(c, "chunk", 8);
	uint32_t *cur = c->queueA, *nxt = c->queueB;
	uint32_t upper = nPaths;                   // what the host knows about the queue sizes
	bool shadowPending = false;
	int b = 0;
	// Every way out of this function -- cancel, a failed HIP call, the normal end -- leaves the context as the host-driven
	// loop expects it and the second stream joined: a shadow launch still running on it adds to Li with a plain
	// read-modify-write (k_trace MODE 1), and the next frame's clear / generate kernels on c->stream are not ordered
	// against it otherwise (a GUI cancels and re-renders at once).
	struct Restore {
		mtsgpu_ctx *c; hipStream_t s2; const bool &pending;
		~Restore() {
			if (pending) (void) hipStreamSynchronize(s2);
			c->q.dev_stats = nullptr; c->q.counters = c->counterSets; c->q.spill = c->spillClosest;
		}
	} restore{ c, s2, shadowPending };
	RayQueuesOff rqOff{ c };
	c->q.nee_parked = tuningOr(c, "nee_parked", 1) != 0 ? 1u : 0u;
	// cls >= 0: a traversal launch of that class (ctx.h: traceEvClass)
	auto timed = [&](std::vector<std::pair<hipEvent_t, hipEvent_t>> &pool, size_t &used, hipStream_t s, int which, int cls = -1) -> int {
		if (!c->timeKernels) return 0;
		hipEvent_t *ev = which == 0 ? (cls >= 0 ? nextTraceEvents(c, cls) : nextEventPair(c, pool, used)) : &pool[used - 1].first;
		if (!ev) return fail(c, MTSGPU_EHIP, "hipEventCreate failed");
		HIPCHK(c, hipEventRecord(ev[which], s));
		return 0;
	};
	while (b < limit && upper > 0) {
		if (cancel && *cancel) return fail(c, MTSGPU_ECANCEL, "render cancelled");
		const int end = (int) std::min<long long>((long long) b + chunk, limit);
		for (; b < end; ++b) {
			uint32_t *set = counterSet(c, b), *prev = counterSet(c, b - 1);
			c->q.counters = set; c->q.next = nxt; c->q.spill = c->spillClosest;
			rayQueues(c, cur, nxt);
			int rc = timed(c->traceEvents, c->traceEvUsed, s1, 0, b == 0 ? 1 : 0); if (rc) return rc;
			launch_trace(s1, 0, c->countTraversal, true, c->dsc, c->paths, c->q, cur, upper, b == 0,
			             b == 0 ? nullptr : prev + (size_t) kNextWord);
			rc = timed(c->traceEvents, c->traceEvUsed, s1, 1); if (rc) return rc;
			// the shading of this bounce adds to Li after the shadow rays of the previous one have (path.cpp:124 before :80)
			if (shadowPending) HIPCHK(c, hipStreamWaitEvent(s1, c->evShadow[(b - 1) & 1], 0));
			rc = timed(c->shadeEvents, c->shadeEvUsed, s1, 0); if (rc) return rc;
			launch_prep(s1, set, prev, c->viewsDev, c->q.bin_seg_cap, c->devStats);
			if (cfg.dr_mode == 0 && tuningOr(c, "shade_fused", 1) != 0) {
				launch_shade_all(s1, c->dsc, c->paths, cfg, c->q, c->viewsDev, c->binMask & ((1u << kNumBins) - 1u), upper);
			} else {
				BinView none{};
				for (int bin = 0; bin < kNumBins; ++bin)
					if (c->binMask & (1u << bin))
						launch_shade(s1, bin, c->dsc, c->paths, cfg, c->q, none, c->viewsDev, upper);
			}
			rc = timed(c->shadeEvents, c->shadeEvUsed, s1, 1); if (rc) return rc;
			HIPCHK(c, hipGetLastError());
			HIPCHK(c, hipEventRecord(c->evShade[b & 1], s1));
			// shadow rays of this bounce on the second stream, next to the closest-hit launch of the next bounce
			HIPCHK(c, hipStreamWaitEvent(s2, c->evShade[b & 1], 0));
			DQueues q2 = c->q; q2.spill = c->spillShadow;
			rc = timed(c->traceEvents, c->traceEvUsed, s2, 0, 2); if (rc) return rc;
			launch_trace(s2, 1, c->countTraversal, false, c->dsc, c->paths, q2, c->q.shadow, upper, b == 0,
			             set + (size_t) kShadowWord);
			rc = timed(c->traceEvents, c->traceEvUsed, s2, 1); if (rc) return rc;
			HIPCHK(c, hipGetLastError());
			HIPCHK(c, hipEventRecord(c->evShadow[b & 1], s2));
			shadowPending = true;
			std::swap(cur, nxt);
		}
		// how many paths are left: the one read-back of the chunk
		uint32_t *hostNext = &c->hostCounters[kNumCounters * kCounterStride + 2];
		HIPCHK(c, hipMemcpyAsync(hostNext, counterSet(c, b - 1) + (size_t) kNextWord, sizeof(uint32_t), hipMemcpyDeviceToHost, s1));
		HIPCHK(c, hipEventRecord(c->evCount, s1));
		HIPCHK(c, hipEventSynchronize(c->evCount));
		upper = *hostNext;
	}
	if (shadowPending) {
		HIPCHK(c, hipStreamWaitEvent(s1, c->evShadow[(b - 1) & 1], 0));      // the film kernels that follow wait on the device
		shadowPending = false;                                               // ... so the guard need not block the host
	}
	return 0;
}

// the statistics a device-driven frame kept on the device (the host-driven loop counts on the host)
int collectDeviceStats(mtsgpu_ctx *c) {
	if (!c->devStatsUsed) return 0;
	c->devStatsUsed = false;
	unsigned long long h[kNumDevStats];
	HIPCHK(c, hipMemcpy(h, c->devStats, sizeof(h), hipMemcpyDeviceToHost));
	c->stats.rays_closest += h[kStatClosest]; c->stats.rays_shadow += h[kStatShadow]; c->stats.trace_launches += h[kStatLaunches];
	if (h[kStatOverflow]) return fail(c, MTSGPU_EHIP, "internal: bin segment overflow in a device-driven frame");
	return 0;
}

// One wavefront pass: all bounces of the paths already generated into queueA[0..nPaths)
int runBounces(mtsgpu_ctx *c, const DConfig &cfg, uint32_t nPaths, volatile const int *cancel) {
	c->q.counters = c->counterSets; c->q.spill = c->spillClosest; c->q.dev_stats = nullptr;
	if (cfg.integrator == 1 && (cfg.n_lum > 1 || cfg.n_bsdf > 1))
		return runDirectRounds(c, cfg, nPaths, cancel);
	{
		// frames of few paths are chains of launches too short to hide a host round trip
		const long sf = tuningOr(c, "sync_free", -1);
		if (sf == 1 || (sf < 0 && nPaths <= (8u << 20)))
			return runBouncesDevice(c, cfg, nPaths, cancel);
	}
	// Two chip-filling persistent grids next to each other.  overlap = 1 (round 5, profiles/r05k_*): the any-hit launch of bounce b
	// first, the closest-hit launch of bounce b + 1 behind it on the other stream -- a third SLOWER, because the workgroups of the
	// larger footprint (80 VGPRs, 52 KB) do not pack into the holes the smaller ones (64 VGPRs, 36 KB) leave.  overlap = 2
	// (round 6): the other way round -- the closest-hit launch of bounce b + 1 goes first and takes the whole chip, the any-hit
	// launch of bounce b is enqueued behind it on the second stream and its workgroups move in where closest-hit workgroups
	// leave, i.e. into the 0.3-0.7 ms at the end of that launch in which its longest rays finish alone.  Legal either way: the
	// any-hit kernel only parks direct-light terms, the shading of bounce b + 1 waits for both.
	const long overlap = tuningOr(c, "overlap", 0);
	// merged = 1: the held-back any-hit queue rides in the next closest-hit LAUNCH (k_trace_pair: one footprint, no second stream)
	const bool merged = tuningOr(c, "merged", 0) != 0 && !overlap && !c->countTraversal;
	uint32_t nQ = nPaths;
	uint32_t *cur = c->queueA, *nxt = c->queueB;
	bool first = true;       // camera rays and their shadow rays are coherent: plain 64-ray batches win there
	hipStream_t s = c->stream, s2 = overlap ? c->stream2 : c->stream;
	const size_t setBytes = (size_t) kNumCounters * kCounterStride * sizeof(uint32_t);
	bool shadowPending = false;
	int b = 0;
	RayQueuesOff rqOff{ c };
	c->q.nee_parked = tuningOr(c, "nee_parked", 1) != 0 ? 1u : 0u;
	if (overlap && !c->q.nee_parked) return fail(c, MTSGPU_EINVAL, "overlap needs nee_parked");
	// the any-hit launch of a bounce: at once, or (overlap = 2) held back until the next closest-hit launch has been enqueued
	struct Deferred { bool pending = false; DQueues q; uint32_t n = 0; bool coherent = false; int bounce = 0; } held;
	auto launchShadow = [&](const DQueues &q2, uint32_t nShadow, bool coherent, int bounce) -> int {
		hipEvent_t *ev2 = c->timeKernels ? nextTraceEvents(c, 2) : nullptr;
		if (ev2) HIPCHK(c, hipEventRecord(ev2[0], s2));
		launch_trace(s2, 1, c->countTraversal, false, c->dsc, c->paths, q2, q2.shadow, nShadow, coherent);
		if (ev2) HIPCHK(c, hipEventRecord(ev2[1], s2));
		HIPCHK(c, hipGetLastError());
		if (overlap) HIPCHK(c, hipEventRecord(c->evShadow[bounce & 1], s2));
		return 0;
	};
	auto flushHeld = [&]() -> int {
		if (!held.pending) return 0;
		held.pending = false;
		if (const long us = tuningOr(c, "overlap_delay_us", 0)) {      // experiment: let the closest-hit grid take its slots first
			const auto t0 = std::chrono::steady_clock::now();
			while (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < us) { }
		}
		return launchShadow(held.q, held.n, held.coherent, held.bounce);
	};
	for (; nQ > 0; ++b) {
		if (cancel && *cancel)
			return fail(c, MTSGPU_ECANCEL, "render cancelled");
		// the counter set of this bounce; its last users (bounce b - 2) are done: the shading of bounce b - 1 waited for them
		c->q.counters = counterSet(c, b); c->q.spill = c->spillClosest;
		HIPCHK(c, hipMemsetAsync(c->q.counters, 0, setBytes, s));
		c->q.next = nxt;
		rayQueues(c, cur, nxt);
		// closest hit + material sort (and, behind it, the any-hit launch of the previous bounce that was held back)
		BinView views[kNumBins];
		int rc;
		if (merged && held.pending) {
			const PairedShadow pair{ &held.q, held.n, held.coherent };
			held.pending = false;
			rc = traceAndBin(c, cur, nQ, first, views, nullptr, &pair);
		} else {
			rc = traceAndBin(c, cur, nQ, first, views, flushHeld);
		}
		if (rc) return rc;
		c->stats.rays_closest += nQ;
		// the shading of this bounce adds to Li after the shadow rays of the previous one have (path.cpp:124 before :80)
		if (shadowPending && overlap) HIPCHK(c, hipStreamWaitEvent(s, c->evShadow[(b - 1) & 1], 0));
		// shade, one launch per BSDF type
		hipEvent_t *sev = c->timeKernels ? nextEventPair(c, c->shadeEvents, c->shadeEvUsed) : nullptr;
		if (sev) HIPCHK(c, hipEventRecord(sev[0], s));
		for (int bin = 0; bin < kNumBins; ++bin)
			launch_shade(s, bin, c->dsc, c->paths, cfg, c->q, views[bin]);
		if (sev) HIPCHK(c, hipEventRecord(sev[1], s));
		HIPCHK(c, hipGetLastError());
		rc = readCounters(c); if (rc) return rc;
		const uint32_t nNext = c->hostCounters[kNextWord], nShadow = c->hostCounters[kShadowWord];
		// shadow rays of this bounce (they add the direct-light term before the next bounce adds its own); the shading above has completed
		if (nShadow) {
			DQueues q2 = c->q; q2.spill = c->spillShadow;
			c->lastPass.shadowMax = std::max(c->lastPass.shadowMax, nShadow);
			if ((overlap == 2 || merged) && nNext > 0) {
				held.pending = true; held.q = q2; held.n = nShadow; held.coherent = first; held.bounce = b;
			} else {
				rc = launchShadow(q2, nShadow, first, b); if (rc) return rc;
			}
			shadowPending = true;
			c->stats.rays_shadow += nShadow; c->stats.trace_launches++;
		} else {
			shadowPending = false;
		}
		if (getenv("MTSGPU_DEBUG") && !overlap) {
			HIPCHK(c, hipStreamSynchronize(s)); HIPCHK(c, hipStreamSynchronize(s2));
			float a = 0, b2 = 0, c2 = 0;
			if (c->timeKernels) {
				const size_t ti = c->traceEvUsed - (nShadow ? 2 : 1);
				(void) hipEventElapsedTime(&a, c->traceEvents[ti].first, c->traceEvents[ti].second);
				(void) hipEventElapsedTime(&b2, c->shadeEvents[c->shadeEvUsed - 1].first, c->shadeEvents[c->shadeEvUsed - 1].second);
				if (nShadow) (void) hipEventElapsedTime(&c2, c->traceEvents[ti + 1].first, c->traceEvents[ti + 1].second);
			}
			fprintf(stderr, "bounce: closest %u rays %.3f ms (%.2f Grays/s) | shade %.3f ms | shadow %u rays %.3f ms (%.2f Grays/s)\n",
			        nQ, a, a > 0 ? nQ / a / 1e6 : 0.0, b2, nShadow, c2, c2 > 0 ? nShadow / c2 / 1e6 : 0.0);
		}
		std::swap(cur, nxt);
		nQ = nNext;
		first = false;
	}
	{ const int rc = flushHeld(); if (rc) return rc; }
	if (shadowPending && overlap) HIPCHK(c, hipStreamWaitEvent(s, c->evShadow[(b - 1) & 1], 0));
	c->q.counters = c->counterSets;
	return 0;
}

// ---- exact record-tail filter of k_trace: per leaf entry, one flag ----------------------------------------------------------
// TriAccel::rayIntersect (triaccel.h:141-158) first computes the plane distance t from the record's head, then
//     hu = o_u + t d_u - a_u,  hv = o_v + t d_v - a_v,  u = hv b_nu + hu b_nv,  v = hu c_nu + hv c_nv,  accept iff u >= 0 && v >= 0 && u + v <= 1
// from its tail.  k_trace may skip the tail -- two 16-byte requests -- of a candidate exactly when that test is CERTAIN to fail.  It
// knows the face through which the ray leaves the leaf it is visiting (axis, plane, side: the current exit point of the traversal,
// sahkdtree3.h:233,248-249) and the point p = o + t d in the arithmetic of the reference; what it cannot know without the tail is
// whether the triangle reaches beyond that face.  The flag says: for EVERY pair of binary32 values (p_u, p_v) with p_u beyond the
// leaf's box on the u axis (either side), or p_v beyond it on the v axis, the expressions above -- evaluated in binary32 in the
// reference's order, each operation rounded -- fail the test.  Proof per half-plane, here for p_u > hi_u:
//   * hu = fl(p_u - a_u) >= fl(hi_u - a_u) =: h (rounding is monotone); the flag needs h > 0.
//   * with U = hv b_nu + hu b_nv and V = hu c_nu + hv c_nv in exact arithmetic, the computed u, v differ from U, V by at most
//     E_u = g (|hv b_nu| + |hu b_nv|) + e and E_v likewise (two roundings per term: g = 2^-22 covers 2 ulp, e the subnormal range),
//     and fl(u + v) <= 1 needs u + v <= 1 + 2^-24.  So the test fails whenever U < -E_u, or V < -E_v, or U + V - E_u - E_v > 1 + 2^-22.
//   * all three conditions are stable under (hu, hv) -> lambda (hu, hv), lambda >= 1 (U, V and the g-part of E scale with lambda),
//     and every point of the half-plane hu >= h is lambda (h, w) for some real w: it suffices that for every real w one of the three
//     holds at (h, w).  On w >= 0 and on w <= 0 each condition is "an affine function of w is positive": the maximum of three
//     affine functions is convex, its minimum over a half-line lies at the end point or where two of them cross.
// The bounds are evaluated in binary64 (products of two binary32 values are exact there) with g doubled for its own rounding.
// Nothing here depends on where the triangle's vertices are: a flag that is set is a statement about the record's numbers alone.
bool rejectsOnHalfLine(const double m[3], const double c[3]) {
	// max_i (m_i z + c_i) > 0 for all z >= 0 ?
	auto value = [&](double z) { double v = -INFINITY; for (int i = 0; i < 3; ++i) v = std::max(v, m[i] * z + c[i]); return v; };
	if (!(value(0.0) > 0.0)) return false;
	// far out: some function must grow, or a constant one must stay positive
	double mmax = std::max(m[0], std::max(m[1], m[2]));
	if (!(mmax >= 0.0)) return false;
	if (mmax == 0.0) {
		double cbest = -INFINITY;
		for (int i = 0; i < 3; ++i) if (m[i] == 0.0) cbest = std::max(cbest, c[i]);
		if (!(cbest > 0.0)) return false;
	}
	for (int i = 0; i < 3; ++i)
		for (int j = i + 1; j < 3; ++j) {
			if (m[i] == m[j]) continue;
			const double z = (c[j] - c[i]) / (m[i] - m[j]);
			if (!(z > 0.0) || !std::isfinite(z)) continue;
			// the crossing point itself is rounded: look at it and at its neighbours
			for (double zz : { z, z * (1.0 - 1e-12), z * (1.0 + 1e-12) })
				if (!(value(zz) > 0.0)) return false;
		}
	return true;
}
// the half-plane {fixed coordinate beyond h (same sign as h), other coordinate w free}: U = af h' + aw w, V = bf h' + bw w
bool rejectsBeyond(double af, double aw, double bf, double bw, double h) {
	if (!(h != 0.0) || !std::isfinite(h) || !std::isfinite(af) || !std::isfinite(aw) || !std::isfinite(bf) || !std::isfinite(bw)) return false;
	const double g = 0x1p-21, e = 0x1p-140;
	for (int side = 0; side < 2; ++side) {              // w >= 0, then w = -z <= 0
		const double aW = side ? -aw : aw, bW = side ? -bw : bw;
		const double m[3] = { -aW - g * std::fabs(aW), -bW - g * std::fabs(bW), aW + bW - g * (std::fabs(aW) + std::fabs(bW)) };
		const double c[3] = { -af * h - g * std::fabs(af * h) - e, -bf * h - g * std::fabs(bf * h) - e,
		                      (af + bf) * h - g * (std::fabs(af * h) + std::fabs(bf * h)) - 1.0 - 0x1p-21 - 2 * e };
		if (!rejectsOnHalfLine(m, c)) return false;
	}
	return true;
}
// The SAH builder puts its planes on the bounds of the (clipped) triangles, so most triangles TOUCH faces of their leaf, and for
// a point one ulp beyond a touched face nothing can be proven (u + v may round to exactly 1).  The kernel therefore only skips a
// tail when the point lies beyond the face by more than a margin mu (one scene-wide binary32 constant, DTraceScene::tail_margin):
// it tests fl(p - plane) > mu, which implies p - plane > mu in exact arithmetic (rounding is monotone), and the plane is the
// leaf's own bound or lies beyond it.  The proof is made for the half-planes beyond lo - mu / hi + mu.
// rec: the 12 dwords of a TriAccel (triaccel.h:34-48); lo / hi: the leaf's box
float roundedTowardZero(double x) {
	float f = (float) x;
	if (std::fabs((double) f) > std::fabs(x)) f = std::nextafterf(f, 0.0f);
	return f;
}
bool tailFilterFlag(const uint32_t *rec, const float lo[3], const float hi[3], float margin) {
	const uint32_t k = rec[0];
	if (k > 2u || !(margin >= 0.0f) || !std::isfinite(margin)) return false;
	float f[12]; std::memcpy(f, rec, 48);
	const float a_u = f[4], a_v = f[5];
	const double b_nu = f[6], b_nv = f[7], c_nu = f[8], c_nv = f[9];
	const int ku = (int) ((k + 1u) % 3u), kv = (int) ((k + 2u) % 3u);          // triaccel.h:104-137
	// hu = fl(p_u - a_u) with p_u > hi_u + mu: hu >= fl(hi_u + mu - a_u) >= that value rounded toward zero (binary64 holds the sum of
	// three binary32 values to 2^-53, far inside the step to the next binary32 value toward zero); likewise below lo - mu
	const float hUhi = roundedTowardZero(((double) hi[ku] + margin - a_u) * (1.0 - 0x1p-50)), hUlo = roundedTowardZero(((double) lo[ku] - margin - a_u) * (1.0 - 0x1p-50));
	const float hVhi = roundedTowardZero(((double) hi[kv] + margin - a_v) * (1.0 - 0x1p-50)), hVlo = roundedTowardZero(((double) lo[kv] - margin - a_v) * (1.0 - 0x1p-50));
	if (!(hUhi > 0.0f) || !(hUlo < 0.0f) || !(hVhi > 0.0f) || !(hVlo < 0.0f)) return false;
	// fixed hu: U = hv b_nu + hu b_nv -> af = b_nv, aw = b_nu; V = hu c_nu + hv c_nv -> bf = c_nu, bw = c_nv
	if (!rejectsBeyond(b_nv, b_nu, c_nu, c_nv, hUhi) || !rejectsBeyond(b_nv, b_nu, c_nu, c_nv, hUlo)) return false;
	// fixed hv: af = b_nu, aw = b_nv; bf = c_nv, bw = c_nu
	return rejectsBeyond(b_nu, b_nv, c_nv, c_nu, hVhi) && rejectsBeyond(b_nu, b_nv, c_nv, c_nu, hVlo);
}
// one flag per entry of the index list: the leaves' boxes come from walking the tree with the scene's box (gkdtree.h:1170-1176)
// the margin of a scene: 2^-16 of its largest coordinate (some tens of ulps there; rays whose plane point lies closer to a face
// than this are not filtered, everything else about the filter is independent of the choice)
float tailFilterMargin(const mtsgpu_scene *sc) {
	float m = 0.0f;
	for (int a = 0; a < 3; ++a) m = std::max(m, std::max(std::fabs(sc->aabb_min[a]), std::fabs(sc->aabb_max[a])));
	m *= 0x1p-16f;
	return (std::isfinite(m) && m > 0.0f) ? m : 0x1p-100f;
}
void tailFilterFlags(const mtsgpu_scene *sc, float margin, std::vector<uint8_t> &flags) {
	flags.assign(sc->n_indices, 0);
	struct Item { uint32_t node; float lo[3], hi[3]; };
	std::vector<Item> stack;
	Item root; root.node = 0;
	for (int a = 0; a < 3; ++a) { root.lo[a] = sc->aabb_min[a]; root.hi[a] = sc->aabb_max[a]; }
	stack.push_back(root);
	while (!stack.empty()) {
		const Item it = stack.back(); stack.pop_back();
		const uint32_t a = sc->kd_nodes[2 * (size_t) it.node], b = sc->kd_nodes[2 * (size_t) it.node + 1];
		if (a & 0x80000000u) {
			for (uint32_t e = a & 0x7FFFFFFFu; e < b; ++e)
				flags[e] = tailFilterFlag(sc->triaccel + 12 * (size_t) sc->kd_indices[e], it.lo, it.hi, margin) ? 1 : 0;
			continue;
		}
		const int axis = (int) (a & 3u);
		float split; std::memcpy(&split, &b, 4);
		const uint32_t left = it.node + ((a & 0x3FFFFFFCu) >> 2);
		Item l = it, r = it;
		l.node = left; l.hi[axis] = split;
		r.node = left + 1; r.lo[axis] = split;
		stack.push_back(r); stack.push_back(l);
	}
}

int checkReady(mtsgpu_ctx *c) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	if (!c->haveScene) return fail(c, MTSGPU_ESTATE, "no scene uploaded");
	if (!c->haveCamera) return fail(c, MTSGPU_ESTATE, "no camera set");
	if (hipSetDevice(c->device) != hipSuccess) return fail(c, MTSGPU_EHIP, "hipSetDevice failed");
	return 0;
}

void collectTimings(mtsgpu_ctx *c) {
	float ms;
	// traversal launches may overlap (two streams): their summed durations and the length of the union of their intervals
	std::vector<std::pair<float, float>> iv;
	for (size_t i = 0; i < c->traceEvUsed; ++i) {
		if (hipEventElapsedTime(&ms, c->traceEvents[i].first, c->traceEvents[i].second) != hipSuccess) continue;
		c->stats.trace_ms += ms;
		const int cls = i < c->traceEvClass.size() ? c->traceEvClass[i] : 0;
		if (cls == 1) c->stats.trace_first_ms += ms;
		else if (cls == 2) c->stats.trace_shadow_ms += ms;
		float t0 = 0;
		if (i == 0 || hipEventElapsedTime(&t0, c->traceEvents[0].first, c->traceEvents[i].first) == hipSuccess)
			iv.emplace_back(t0, t0 + ms);
	}
	std::sort(iv.begin(), iv.end());
	float curA = 0, curB = -1;
	for (auto &p : iv) {
		if (curB < curA) { curA = p.first; curB = p.second; }
		else if (p.first <= curB) curB = std::max(curB, p.second);
		else { c->stats.trace_union_ms += curB - curA; curA = p.first; curB = p.second; }
	}
	if (curB >= curA) c->stats.trace_union_ms += curB - curA;
	for (size_t i = 0; i < c->shadeEvUsed; ++i)
		if (hipEventElapsedTime(&ms, c->shadeEvents[i].first, c->shadeEvents[i].second) == hipSuccess) c->stats.shade_ms += ms;
	c->traceEvUsed = c->shadeEvUsed = 0;
}

int fetchTraceCounts(mtsgpu_ctx *c) {
	unsigned long long h[kNumTraceCounts];
	HIPCHK(c, hipMemcpy(h, c->q.trace_counts, sizeof(h), hipMemcpyDeviceToHost));
	c->stats.n_inner = h[0]; c->stats.n_leaf = h[1]; c->stats.n_idx = h[2]; c->stats.n_tri_tested = h[3];
	c->stats.req_pair_global = h[kCntPairGlobal]; c->stats.req_pair_lds = h[kCntPairLds];
	c->stats.req_node_global = h[kCntNodeGlobal]; c->stats.req_node_lds = h[kCntNodeLds];
	c->stats.req_tail = h[kCntTail]; c->stats.req_spill = h[kCntSpill]; c->stats.req_head = h[kCntHead];
	if (getenv("MTSGPU_DEBUG"))     // candidates whose tail was fetched with a plane distance outside the visited leaf's own interval
		fprintf(stderr, "[mtsgpu] record tails: %llu requests, %llu candidates of which %llu (%.3f) lie outside the leaf's interval\n",
		        h[kCntTail], h[kCntTail] / 2, h[15], h[kCntTail] ? 2.0 * h[15] / h[kCntTail] : 0.0);
	if (getenv("MTSGPU_DEBUG"))     // SIMD utilisation of the traversal loops: lane steps / lane slots issued
		fprintf(stderr, "[mtsgpu] lanes: inner %llu/%llu (%.3f)  leaf-prims %llu/%llu (%.3f)  outer %llu/%llu (%.3f)  batch slots %llu\n",
		        h[0], h[4], h[4] ? (double) h[0] / h[4] : 0.0, h[2], h[5], h[5] ? (double) h[2] / h[5] : 0.0,
		        h[1], h[6], h[6] ? (double) h[1] / h[6] : 0.0, h[7]);
	return 0;
}

} // namespace

extern "C" {

int mtsgpu_abi_version(void) { return MTSGPU_ABI_VERSION; }

size_t mtsgpu_abi_sizeof(int which) {
	switch (which) {
		case 0: return sizeof(mtsgpu_scene);
		case 1: return sizeof(mtsgpu_camera);
		case 2: return sizeof(mtsgpu_stats);
		case 3: return sizeof(mtsgpu_mesh);
		case 4: return sizeof(mtsgpu_scene_desc);
		case 5: return sizeof(mtsgpu_kd_params);
		default: return 0;
	}
}

const char *mtsgpu_last_error(const mtsgpu_ctx *ctx) { return ctx ? ctx->error.c_str() : g_lastError.c_str(); }

int mtsgpu_create(int device, mtsgpu_ctx **out) {
	if (!out) return fail(nullptr, MTSGPU_EINVAL, "out is null");
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
		return fail(nullptr, MTSGPU_ENODEV, "no HIP device visible; libmtsgpu has no CPU fallback");
	if (device < 0 || device >= n)
		return fail(nullptr, MTSGPU_EINVAL, "device %d out of range (%d visible)", device, n);
	hipDeviceProp_t prop;
	HIPCHK(nullptr, hipGetDeviceProperties(&prop, device));
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(nullptr, MTSGPU_ENODEV, "device %d is %s; libmtsgpu is built for gfx950 only", device, prop.gcnArchName);
	HIPCHK(nullptr, hipSetDevice(device));
	mtsgpu_ctx *c = new mtsgpu_ctx();
	c->device = device;
	c->nCUs = (uint32_t) std::max(1, prop.multiProcessorCount);
	if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return fail(nullptr, MTSGPU_EHIP, "hipStreamCreate failed"); }
	c->ownStream = true;
	{
		bool ok = hipStreamCreate(&c->stream2) == hipSuccess && hipEventCreateWithFlags(&c->evCount, hipEventDisableTiming) == hipSuccess;
		for (int i = 0; i < 2 && ok; ++i)
			ok = hipEventCreateWithFlags(&c->evShade[i], kOrderEventFlags) == hipSuccess
			  && hipEventCreateWithFlags(&c->evShadow[i], kOrderEventFlags) == hipSuccess;
		if (!ok) { mtsgpu_destroy(c); return fail(nullptr, MTSGPU_EHIP, "stream / event creation failed"); }
	}
	if (hipHostMalloc((void **) &c->hostCounters, (kNumCounters * kCounterStride + 4) * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
		(void) hipStreamDestroy(c->stream); delete c; return fail(nullptr, MTSGPU_EHIP, "hipHostMalloc failed");
	}
	if (hipMalloc((void **) &c->pathLen, sizeof(unsigned long long)) != hipSuccess) {
		mtsgpu_destroy(c); return fail(nullptr, MTSGPU_EHIP, "hipMalloc failed");
	}
	{
		// the first 1000 primes (primeTable, src/libcore/util.cpp:64-122) for the halton / hammersley samplers
		std::vector<uint16_t> primes;
		for (uint32_t v = 2; primes.size() < 1000; ++v) {
			bool isPrime = true;
			for (uint32_t d = 2; d * d <= v; ++d) if (v % d == 0) { isPrime = false; break; }
			if (isPrime) primes.push_back((uint16_t) v);
		}
		if (hipMalloc((void **) &c->primes, primes.size() * sizeof(uint16_t)) != hipSuccess
		    || hipMemcpy(c->primes, primes.data(), primes.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) {
			mtsgpu_destroy(c); return fail(nullptr, MTSGPU_EHIP, "prime table upload failed");
		}
	}
	*out = c;
	return 0;
}

void mtsgpu_destroy(mtsgpu_ctx *c) {
	if (!c) return;
	(void) hipSetDevice(c->device);
	(void) hipDeviceSynchronize();
	freeAll(c->sceneAllocs); freeAll(c->pathAllocs);
	if (c->ownFilm && c->film) (void) hipFree(c->film);
	if (c->pixelList) (void) hipFree(c->pixelList);
	if (c->ldScr) (void) hipFree(c->ldScr);
	if (c->ldPerm) (void) hipFree(c->ldPerm);
	if (c->ldState) (void) hipFree(c->ldState);
	if (c->ldScratch) (void) hipFree(c->ldScratch);
	if (c->arrScr) (void) hipFree(c->arrScr);
	if (c->arrPerm) (void) hipFree(c->arrPerm);
	if (c->arrPts) (void) hipFree(c->arrPts);
	if (c->primSave) (void) hipFree(c->primSave);
	if (c->primes) (void) hipFree(c->primes);
	if (c->pathLen) (void) hipFree(c->pathLen);
	if (c->explicitSamples) (void) hipFree(c->explicitSamples);
	if (c->filtValues) (void) hipFree(c->filtValues);
	if (c->tileMeta) (void) hipFree(c->tileMeta);
	if (c->blocks) (void) hipFree(c->blocks);
	if (c->hostCounters) (void) hipHostFree(c->hostCounters);
	for (auto &e : c->traceEvents) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
	for (auto &e : c->shadeEvents) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
	if (c->ownStream && c->stream) (void) hipStreamDestroy(c->stream);
	if (c->stream2) (void) hipStreamDestroy(c->stream2);
	for (hipEvent_t e : { c->evShade[0], c->evShade[1], c->evShadow[0], c->evShadow[1], c->evCount }) if (e) (void) hipEventDestroy(e);
	delete c;
}

int mtsgpu_set_stream(mtsgpu_ctx *c, void *hip_stream) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	if (c->ownStream && c->stream) { (void) hipStreamSynchronize(c->stream); (void) hipStreamDestroy(c->stream); }
	if (hip_stream) { c->stream = (hipStream_t) hip_stream; c->ownStream = false; }
	else { HIPCHK(c, hipStreamCreate(&c->stream)); c->ownStream = true; }
	return 0;
}

int mtsgpu_upload_scene(mtsgpu_ctx *c, const mtsgpu_scene *sc) {
	if (!c || !sc) return fail(c, MTSGPU_EINVAL, "null argument");
	c->lastPass.valid = false;
	if (sc->abi_version != MTSGPU_ABI_VERSION) return fail(c, MTSGPU_EINVAL, "scene ABI version %u != %d", sc->abi_version, MTSGPU_ABI_VERSION);
	if (sc->n_nodes == 0 || !sc->kd_nodes) return fail(c, MTSGPU_EINVAL, "scene has no kd-tree");
	if (sc->n_lums == 0) return fail(c, MTSGPU_EINVAL, "scene has no luminaire (Scene::initialize would add a constant one, scene.cpp:310-318)");
	HIPCHK(c, hipSetDevice(c->device));
	// --- validate everything the kernels index with (an out-of-range index would fault the GPU) ---
	for (uint32_t i = 0; i < sc->n_nodes; ++i) {
		const uint32_t a = sc->kd_nodes[2 * (size_t) i], b = sc->kd_nodes[2 * (size_t) i + 1];
		if (a & 0x80000000u) {
			if ((a & 0x7FFFFFFFu) > b || b > sc->n_indices) return fail(c, MTSGPU_EINVAL, "kd leaf %u: index range out of bounds", i);
		} else {
			if (a & 0x40000000u) return fail(c, MTSGPU_EINVAL, "kd node %u: indirection nodes are not supported (gkdtree.h:1116-1126 removes them)", i);
			const uint64_t left = (uint64_t) i + ((a & 0x3FFFFFFCu) >> 2);
			if ((a & 3u) == 3u || left <= i || left + 1 >= sc->n_nodes) return fail(c, MTSGPU_EINVAL, "kd node %u: bad axis/child offset", i);
		}
	}
	{
		// every node is reached exactly once from the root (the device order below is built by walking the tree: an
		// unreached node would land on the root's slot, a shared one twice) and no branch is deeper than the traversal
		// stack of k_trace (LDS levels + spill levels; the reference's own limit is MTS_KD_MAXDEPTH = 48, gkdtree.h:35)
		const uint32_t depthLimit = (uint32_t) trace_stack_levels() - 2;
		std::vector<uint8_t> seen(sc->n_nodes, 0);
		std::vector<std::pair<uint32_t, uint32_t>> stack;       // node, depth
		stack.emplace_back(0u, 1u);
		uint32_t visited = 0;
		while (!stack.empty()) {
			const auto [i, depth] = stack.back(); stack.pop_back();
			if (seen[i]) return fail(c, MTSGPU_EINVAL, "kd node %u is the child of two nodes", i);
			seen[i] = 1; ++visited;
			const uint32_t a = sc->kd_nodes[2 * (size_t) i];
			if (a & 0x80000000u) continue;
			if (depth >= depthLimit) return fail(c, MTSGPU_EINVAL, "kd-tree deeper than %u levels", depthLimit);
			const uint32_t left = i + ((a & 0x3FFFFFFCu) >> 2);
			stack.emplace_back(left + 1, depth + 1); stack.emplace_back(left, depth + 1);
		}
		if (visited != sc->n_nodes) return fail(c, MTSGPU_EINVAL, "kd-tree: %u of %u nodes are unreachable from the root", sc->n_nodes - visited, sc->n_nodes);
	}
	for (uint32_t i = 0; i < sc->n_indices; ++i)
		if (sc->kd_indices[i] >= sc->n_tris) return fail(c, MTSGPU_EINVAL, "kd index %u out of range", i);
	if ((sc->shape_type == nullptr) != (sc->shape_params == nullptr)) return fail(c, MTSGPU_EINVAL, "shape_type and shape_params must both be given or both be NULL");
	auto shapeType = [&](uint32_t s) { return sc->shape_type ? sc->shape_type[s] : (uint32_t) MTSGPU_SHAPE_TRIMESH; };
	for (uint32_t s = 0; s < sc->n_shapes; ++s) {
		if (shapeType(s) > MTSGPU_SHAPE_SPHERE) return fail(c, MTSGPU_EINVAL, "shape %u: unknown shape type", s);
		if (sc->shape_tri_offset[s] > sc->shape_tri_offset[s + 1] || sc->shape_tri_offset[s + 1] > sc->n_tris) return fail(c, MTSGPU_EINVAL, "shape_tri_offset not monotone");
		const bool mesh = shapeType(s) == MTSGPU_SHAPE_TRIMESH;
		if (!mesh) {
			// any other shape is ONE kd-tree primitive (skdtree.cpp:54-57)
			if (sc->shape_tri_offset[s + 1] - sc->shape_tri_offset[s] != 1) return fail(c, MTSGPU_EINVAL, "shape %u: a non-mesh shape must own exactly one primitive", s);
			const float *SP = sc->shape_params + (size_t) MTSGPU_SHAPE_NPARAMS * s;
			for (int i = 0; i < MTSGPU_SHAPE_NPARAMS; ++i) if (!std::isfinite(SP[i])) return fail(c, MTSGPU_EINVAL, "shape %u: non-finite parameter", s);
			if (SP[3] == 0.0f) return fail(c, MTSGPU_EINVAL, "shape %u: zero radius", s);
		}
		for (uint32_t t = sc->shape_tri_offset[s]; t < sc->shape_tri_offset[s + 1]; ++t) {
			const uint32_t k = sc->triaccel[12 * (size_t) t];
			if (mesh) {
				if (k == MTSGPU_KNOTRIANGLE) return fail(c, MTSGPU_EINVAL, "TriAccel %u: a mesh triangle is marked as a shape", t);
				for (int j = 0; j < 3; ++j)
					if (sc->tri_idx[3 * (size_t) t + j] >= sc->n_verts) return fail(c, MTSGPU_EINVAL, "triangle vertex index out of range");
			} else if (k != MTSGPU_KNOTRIANGLE) {
				return fail(c, MTSGPU_EINVAL, "TriAccel %u: the primitive of a non-mesh shape must have k = KNoTriangleFlag", t);
			}
			if (sc->triaccel[12 * (size_t) t + 10] != s) return fail(c, MTSGPU_EINVAL, "TriAccel %u: shape index does not match shape_tri_offset", t);
		}
	}
	for (uint32_t s = 0; s < sc->n_shapes; ++s) {
		if (sc->shape_bsdf[s] >= (int32_t) sc->n_bsdfs || sc->shape_lum[s] >= (int32_t) sc->n_lums) return fail(c, MTSGPU_EINVAL, "shape %u: bad BSDF/luminaire index", s);
		if (sc->shape_tri_offset[s] > sc->shape_tri_offset[s + 1]) return fail(c, MTSGPU_EINVAL, "shape_tri_offset not monotone");
		if (sc->shape_lum[s] >= 0 && (sc->lum_type[sc->shape_lum[s]] != MTSGPU_LUM_AREA || sc->lum_shape[sc->shape_lum[s]] != (int32_t) s))
			return fail(c, MTSGPU_EINVAL, "shape %u: luminaire link is inconsistent", s);
	}
	if (sc->shape_tri_offset[sc->n_shapes] != sc->n_tris) return fail(c, MTSGPU_EINVAL, "shape_tri_offset does not cover all triangles");
	for (uint32_t t = 0; t < sc->n_tris; ++t)
		if (sc->triaccel[12 * (size_t) t + 10] >= sc->n_shapes) return fail(c, MTSGPU_EINVAL, "TriAccel %u: shape index out of range", t);
	for (uint32_t b = 0; b < sc->n_bsdfs; ++b)
		if ((sc->bsdf_type[b] & ~(uint32_t) MTSGPU_BSDF_TWOSIDED) >= MTSGPU_BSDF_NTYPES) return fail(c, MTSGPU_EINVAL, "BSDF %u: unknown type", b);
	for (uint32_t l = 0; l < sc->n_lums; ++l) {
		if (sc->lum_type[l] == MTSGPU_LUM_AREA) {
			const int32_t s = sc->lum_shape[l];
			if (s < 0 || s >= (int32_t) sc->n_shapes) return fail(c, MTSGPU_EINVAL, "luminaire %u: bad shape", l);
			const uint32_t n = sc->shape_tri_offset[s + 1] - sc->shape_tri_offset[s];
			if (shapeType((uint32_t) s) != MTSGPU_SHAPE_TRIMESH) {
				if (sc->lum_cdf_offset[l + 1] != sc->lum_cdf_offset[l]) return fail(c, MTSGPU_EINVAL, "luminaire %u: a non-mesh emitter has no triangle CDF", l);
			} else if (sc->lum_cdf_offset[l + 1] - sc->lum_cdf_offset[l] != n + 1) return fail(c, MTSGPU_EINVAL, "luminaire %u: CDF size mismatch", l);
		} else if (sc->lum_type[l] == MTSGPU_LUM_ENVMAP) {
			// everything env_triangle / env_pdf index with (envmap.cpp:123-193)
			if ((int32_t) l != sc->background_lum) return fail(c, MTSGPU_EINVAL, "luminaire %u: the envmap must be the background luminaire", l);
			const uint64_t np = (uint64_t) sc->env_pdf_width * sc->env_pdf_height;
			if (!sc->env_pixels || !sc->env_pdf || !sc->env_cdf || sc->env_width == 0 || sc->env_height == 0 || np == 0
			    || sc->env_width > 16384 || sc->env_height > 16384 || np > (1u << 26))
				return fail(c, MTSGPU_EINVAL, "luminaire %u: missing or oversized environment map", l);
			for (uint64_t i = 0; i < np; ++i)
				if (!(sc->env_cdf[i] <= sc->env_cdf[i + 1]) || !(sc->env_pdf[i] >= 0.0f)) return fail(c, MTSGPU_EINVAL, "luminaire %u: the environment map's CDF is not monotone", l);
			const float *LP = sc->lum_params + (size_t) MTSGPU_LUM_NPARAMS * l;
			for (int i = 0; i < 25; ++i) if (!std::isfinite(LP[i])) return fail(c, MTSGPU_EINVAL, "luminaire %u: non-finite parameter", l);
		} else if (sc->lum_type[l] > MTSGPU_LUM_COLLIMATED) {
			return fail(c, MTSGPU_EINVAL, "luminaire %u: unknown type", l);
		}
	}
	if (sc->background_lum >= (int32_t) sc->n_lums) return fail(c, MTSGPU_EINVAL, "background luminaire out of range");

	freeAll(c->sceneAllocs);
	c->haveScene = false;
	DScene d{};
	int rc = 0;
	// entryOld[e]: which entry of the index list the device's leaf-record slot e holds.  The identity, unless the experiment knob
	// MTSGPU_LEAF_ORDER=1 lays the leaves' runs out in the order of their nodes in the device tree (treelet order) instead of the
	// builder's index-list order (profiles/r06h_*); leafFirst[i]: first slot of the leaf with (old) node index i
	std::vector<uint32_t> entryOld(sc->n_indices), leafFirst;
	for (uint32_t e = 0; e < sc->n_indices; ++e) entryOld[e] = e;
	const bool leafReorder = getenv("MTSGPU_LEAF_ORDER") && atoi(getenv("MTSGPU_LEAF_ORDER")) == 1;
	{
		// Device node order.  The first trace_top_nodes() slots hold the root and the sibling pairs below it in
		// breadth-first order: k_trace keeps that prefix in LDS.  After it, 128-byte lines (16 nodes) are filled with
		// breadth-first pieces of subtrees ("treelets") so that one L1 miss serves several consecutive traversal steps.
		// The KDNode encoding is unchanged (siblings adjacent, relative offset to the left child,
		// gkdtree.h:442-470); only where a node lives changes, which traversal results do not depend on.
		const uint32_t N = sc->n_nodes;
		const uint32_t topSlots = trace_top_nodes();
		std::vector<uint32_t> newIndex(N, 0u);
		std::vector<uint32_t> stack, cand;           // entries: old index of the left node of a sibling pair
		auto leftOf = [&](uint32_t i) { return i + ((sc->kd_nodes[2 * (size_t) i] & 0x3FFFFFFCu) >> 2); };
		auto isLeaf = [&](uint32_t i) { return (sc->kd_nodes[2 * (size_t) i] & 0x80000000u) != 0; };
		uint32_t pos = 2;                            // slot 0 = root, slot 1 = padding
		newIndex[0] = 0;
		if (!isLeaf(0)) cand.push_back(leftOf(0));
		size_t head = 0;
		while (head < cand.size() || !stack.empty()) {
			if (head == cand.size()) { cand.clear(); head = 0; cand.push_back(stack.back()); stack.pop_back(); }
			const uint32_t l = cand[head++];
			newIndex[l] = pos; newIndex[l + 1] = pos + 1;
			pos += 2;
			for (uint32_t k = 0; k < 2; ++k)
				if (!isLeaf(l + k)) cand.push_back(leftOf(l + k));
			const bool full = pos < topSlots ? false : (pos == topSlots || (pos & 15u) == 0u);
			if (full) {
				// region full: the remaining frontier becomes the roots of later treelets (depth-first order)
				for (size_t k = cand.size(); k > head; --k) stack.push_back(cand[k - 1]);
				cand.clear(); head = 0;
			}
		}
		const uint32_t total = std::max<uint32_t>(pos, std::max(2u, topSlots));      // the LDS prefix is always there to copy
		if (total >= (1u << 29)) return fail(c, MTSGPU_EINVAL, "kd-tree too large");      // absolute child indices; k_trace's stack words keep node index * 2 below bit 30
		std::vector<uint32_t> dev(2 * (size_t) total, 0u);
		dev[2] = 0x80000000u; dev[3] = 0u;           // padding slot: empty leaf, never referenced
		if (leafReorder) {
			std::vector<std::pair<uint32_t, uint32_t>> leaves;      // (device node index, old node index)
			uint64_t covered = 0;
			for (uint32_t i = 0; i < N; ++i)
				if (isLeaf(i)) { leaves.emplace_back(newIndex[i], i); covered += sc->kd_nodes[2 * (size_t) i + 1] - (sc->kd_nodes[2 * (size_t) i] & 0x7FFFFFFFu); }
			if (covered == sc->n_indices) {           // every entry belongs to exactly one leaf (what the builders produce)
				std::sort(leaves.begin(), leaves.end());
				leafFirst.assign(N, 0u);
				uint32_t at = 0;
				for (const auto &lf : leaves) {
					const uint32_t first = sc->kd_nodes[2 * (size_t) lf.second] & 0x7FFFFFFFu, end = sc->kd_nodes[2 * (size_t) lf.second + 1];
					leafFirst[lf.second] = at;
					for (uint32_t e = first; e < end; ++e) entryOld[at++] = e;
				}
			}
		}
		for (uint32_t i = 0; i < N; ++i) {
			uint32_t a = sc->kd_nodes[2 * (size_t) i], b = sc->kd_nodes[2 * (size_t) i + 1];
			uint32_t *o = &dev[2 * (size_t) newIndex[i]];
			if ((a & 0x80000000u) && !leafFirst.empty()) { const uint32_t n = b - (a & 0x7FFFFFFFu); a = 0x80000000u | leafFirst[i]; b = leafFirst[i] + n; }
			if (a & 0x80000000u) { o[0] = a; o[1] = b; }
			else { o[0] = (a & 3u) | (newIndex[leftOf(i)] << 2); o[1] = b; }     // absolute left-child index (< 2^29)
		}
		rc |= upload(c, (const uint32_t **) &d.nodes, dev.data(), dev.size());
	}
	{
		// TriAccel records re-laid out in leaf order (one contiguous run per leaf, no index
		// indirection on the device); non-occluders are shapes without a BSDF
		// (Shape::isOccluder, shape.h:324)
		const size_t LS = 4 * (size_t) kLeafStride;                 // dwords per record slot
		std::vector<uint32_t> ta(LS * ((size_t) sc->n_indices + 1), 0u);
		std::vector<uint8_t> tailFlags;
		d.tail_margin = tailFilterMargin(sc);
		if (trace_tail_filter()) tailFilterFlags(sc, d.tail_margin, tailFlags);      // experiment builds only (trace.hip: MG_TAIL_FILTER)
		else tailFlags.assign(sc->n_indices, 0);
		for (uint32_t slot = 0; slot < sc->n_indices; ++slot) {
			const uint32_t e = entryOld[slot];
			const uint32_t prim = sc->kd_indices[e];
			uint32_t *dst = &ta[LS * (size_t) slot];
			std::memcpy(dst, sc->triaccel + 12 * (size_t) prim, 48);
			// dword 0 = k<<30 | non-occluder<<29 | primitive id (the head of the record decides everything
			// up to the plane distance); dword 10 stays the shape index
			if (prim >= (1u << 28)) return fail(c, MTSGPU_EINVAL, "more than 2^28 primitives");
			const bool isShape = dst[0] == MTSGPU_KNOTRIANGLE;
			// bit 28: the record's tail may be skipped for candidates beyond the leaf's box on a projection axis (tailFilterFlag)
			dst[0] = (std::min(dst[0], 3u) << 30) | (sc->shape_bsdf[dst[10]] < 0 ? 0x20000000u : 0u) | (tailFlags[e] ? 0x10000000u : 0u) | prim;
			dst[11] = 0;
			if ((dst[0] >> 30) == 3u) {
				// k == 3: a degenerate triangle (dword 1 = 0) or a non-triangle shape (dword 1 = shape type,
				// dwords 4..7 = centre and radius of the sphere)
				dst[1] = 0;
				if (isShape) {
					const float *SP = sc->shape_params + (size_t) MTSGPU_SHAPE_NPARAMS * dst[10];
					dst[1] = shapeType(dst[10]);
					std::memcpy(dst + 4, SP, 16);
				}
			}
		}
		rc |= upload(c, (const uint32_t **) &d.leaf_ta, ta.data(), ta.size());
		if (getenv("MTSGPU_DEBUG")) {
			size_t nf = 0; for (uint8_t f : tailFlags) nf += f;
			fprintf(stderr, "[mtsgpu] record-tail filter: %zu of %u leaf entries flagged (%.3f)\n", nf, sc->n_indices, sc->n_indices ? (double) nf / sc->n_indices : 0.0);
		}
		// per-primitive position / normal records for the shading kernels
		const size_t TS = 4 * (size_t) kTriStride;                    // floats per record (one 128-byte line)
		std::vector<float> triRec(TS * ((size_t) sc->n_tris + 1), 0.0f);
		for (uint32_t s = 0; s < sc->n_shapes; ++s)
			for (uint32_t t = sc->shape_tri_offset[s]; t < sc->shape_tri_offset[s + 1]; ++t) {
				float *P = &triRec[TS * (size_t) t], *Nn = P + 12;
				if (shapeType(s) != MTSGPU_SHAPE_TRIMESH) {
					// non-mesh shape: centre + radius, flag bit 31 (the rest comes from shape_params)
					const uint32_t flags = sc->shape_flags[s] | 0x80000000u;
					std::memcpy(P, sc->shape_params + (size_t) MTSGPU_SHAPE_NPARAMS * s, 16);
					std::memcpy(P + 10, &s, 4); std::memcpy(P + 11, &flags, 4);
					continue;
				}
				for (int k = 0; k < 3; ++k) {
					const uint32_t v = sc->tri_idx[3 * (size_t) t + k];
					std::memcpy(P + 3 * k, sc->vtx_pos + 3 * (size_t) v, 12);
					std::memcpy(Nn + 3 * k, sc->vtx_nrm + 3 * (size_t) v, 12);
				}
				const uint32_t flags = sc->shape_flags[s];
				std::memcpy(P + 10, &s, 4); std::memcpy(P + 11, &flags, 4);
			}
		rc |= upload(c, (const float **) &d.tri_pos, triRec.data(), triRec.size());
		d.tri_nrm = d.tri_pos ? d.tri_pos + 3 : nullptr;
	}
	rc |= upload(c, &d.shape_bsdf, sc->shape_bsdf, sc->n_shapes);
	rc |= upload(c, &d.shape_lum, sc->shape_lum, sc->n_shapes);
	rc |= upload(c, &d.shape_flags, sc->shape_flags, sc->n_shapes);
	{
		std::vector<uint32_t> bin(sc->n_shapes + 1, (uint32_t) kNumBsdfTypes);
		for (uint32_t sIdx = 0; sIdx < sc->n_shapes; ++sIdx)
			if (sc->shape_bsdf[sIdx] >= 0) bin[sIdx] = sc->bsdf_type[sc->shape_bsdf[sIdx]] & 0xFFu;
		rc |= upload(c, &d.shape_bin, bin.data(), bin.size());
	}
	{
		std::vector<uint32_t> st(sc->n_shapes + 1, (uint32_t) MTSGPU_SHAPE_TRIMESH);
		std::vector<float> sp((size_t) MTSGPU_SHAPE_NPARAMS * (sc->n_shapes + 1), 0.0f);
		if (sc->shape_type) {
			std::copy(sc->shape_type, sc->shape_type + sc->n_shapes, st.begin());
			std::copy(sc->shape_params, sc->shape_params + (size_t) MTSGPU_SHAPE_NPARAMS * sc->n_shapes, sp.begin());
		}
		rc |= upload(c, &d.shape_type, st.data(), st.size());
		rc |= upload(c, &d.shape_params, sp.data(), sp.size());
	}
	rc |= upload(c, &d.shape_tri_offset, sc->shape_tri_offset, (size_t) sc->n_shapes + 1);
	rc |= upload(c, &d.bsdf_type, sc->bsdf_type, sc->n_bsdfs);
	rc |= upload(c, &d.bsdf_params, sc->bsdf_params, (size_t) MTSGPU_BSDF_NPARAMS * sc->n_bsdfs);
	rc |= upload(c, &d.lum_type, sc->lum_type, sc->n_lums);
	rc |= upload(c, &d.lum_params, sc->lum_params, (size_t) MTSGPU_LUM_NPARAMS * sc->n_lums);
	rc |= upload(c, &d.lum_shape, sc->lum_shape, sc->n_lums);
	rc |= upload(c, &d.lum_inv_area, sc->lum_inv_area, sc->n_lums);
	rc |= upload(c, &d.lum_cdf_offset, sc->lum_cdf_offset, (size_t) sc->n_lums + 1);
	rc |= upload(c, &d.lum_tri_cdf, sc->lum_tri_cdf, sc->lum_cdf_offset[sc->n_lums]);
	rc |= upload(c, &d.lum_sel_cdf, sc->lum_sel_cdf, (size_t) sc->n_lums + 1);
	rc |= upload(c, &d.lum_sel_pdf, sc->lum_sel_pdf, sc->n_lums);
	if (sc->background_lum >= 0 && sc->lum_type[sc->background_lum] == MTSGPU_LUM_ENVMAP) {
		const size_t np = (size_t) sc->env_pdf_width * sc->env_pdf_height;
		rc |= upload(c, &d.env_pixels, sc->env_pixels, 3 * (size_t) sc->env_width * sc->env_height);
		rc |= upload(c, &d.env_pdf, sc->env_pdf, np);
		rc |= upload(c, &d.env_cdf, sc->env_cdf, np + 1);
		d.env_width = sc->env_width; d.env_height = sc->env_height;
		d.env_pdf_width = sc->env_pdf_width; d.env_pdf_height = sc->env_pdf_height;
	}
	if (rc) { freeAll(c->sceneAllocs); return rc; }
	d.lum_sel_sum = sc->lum_sel_sum; d.background_lum = sc->background_lum;
	d.has_shapes = 0;
	for (uint32_t s = 0; s < sc->n_shapes; ++s) if (shapeType(s) != MTSGPU_SHAPE_TRIMESH) d.has_shapes = 1;
	d.n_lums = sc->n_lums; d.n_nodes = sc->n_nodes; d.n_tris = sc->n_tris; d.n_shapes = sc->n_shapes;
	for (int a = 0; a < 3; ++a) { d.aabb_min[a] = sc->aabb_min[a]; d.aabb_max[a] = sc->aabb_max[a]; }
	c->dsc = d; c->nTris = sc->n_tris;
	// material queues that can ever be non-empty: the BSDF types of shapes that have one, and the "terminal" bin
	c->binMask = 1u << kNumBsdfTypes;
	for (uint32_t sIdx = 0; sIdx < sc->n_shapes; ++sIdx)
		if (sc->shape_bsdf[sIdx] >= 0) c->binMask |= 1u << (sc->bsdf_type[sc->shape_bsdf[sIdx]] & 0xFFu);
	c->haveScene = true;
	return 0;
}

int mtsgpu_set_camera(mtsgpu_ctx *c, const mtsgpu_camera *cam) {
	if (!c || !cam) return fail(c, MTSGPU_EINVAL, "null argument");
	c->lastPass.valid = false;
	if (cam->width <= 0 || cam->height <= 0 || (uint64_t) cam->width * (uint64_t) cam->height > 0x7FFFFFFFull)
		return fail(c, MTSGPU_EINVAL, "bad film size %dx%d", cam->width, cam->height);
	if (cam->kind != 0 && cam->kind != 1) return fail(c, MTSGPU_EINVAL, "unknown camera kind %d", cam->kind);
	if (cam->film_width != 0 || cam->film_height != 0 || cam->crop_offset_x != 0 || cam->crop_offset_y != 0) {
		// "Invalid crop window specification!" (film.cpp:41-45)
		if (cam->crop_offset_x < 0 || cam->crop_offset_y < 0 || cam->film_width <= 0 || cam->film_height <= 0
		    || (int64_t) cam->crop_offset_x + cam->width > cam->film_width || (int64_t) cam->crop_offset_y + cam->height > cam->film_height
		    || (uint64_t) cam->film_width * (uint64_t) cam->film_height > 0x7FFFFFFFull)
			return fail(c, MTSGPU_EINVAL, "invalid crop window %dx%d at (%d, %d) of a %dx%d film", cam->width, cam->height,
			            cam->crop_offset_x, cam->crop_offset_y, cam->film_width, cam->film_height);
	}
	c->cam = *cam; c->haveCamera = true;
	return 0;
}

int mtsgpu_set_integrator(mtsgpu_ctx *c, int max_depth, int rr_depth, int strict_normals) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	c->lastPass.valid = false;      // mtsgpu_replay_roof regenerates a pass from its saved configuration: not across a setter
	if (rr_depth <= 0) return fail(c, MTSGPU_EINVAL, "rrDepth == 0 breaks the computation of alpha values! (integrator.cpp:291)");
	c->maxDepth = max_depth; c->rrDepth = rr_depth; c->strictNormals = strict_normals ? 1 : 0;
	c->integrator = 0;
	return 0;
}

int mtsgpu_set_direct_integrator(mtsgpu_ctx *c, int luminaire_samples, int bsdf_samples) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	c->lastPass.valid = false;      // mtsgpu_replay_roof regenerates a pass from its saved configuration: not across a setter
	if (luminaire_samples < 0 || bsdf_samples < 0 || luminaire_samples + bsdf_samples <= 0)
		return fail(c, MTSGPU_EINVAL, "luminaireSamples + bsdfSamples must be > 0 (direct.cpp:41)");
	if (luminaire_samples > 65536 || bsdf_samples > 65536)
		return fail(c, MTSGPU_EINVAL, "at most 65536 samples per strategy");
	c->integrator = 1; c->nLumSamples = luminaire_samples; c->nBsdfSamples = bsdf_samples;
	return 0;
}

int mtsgpu_set_sampler(mtsgpu_ctx *c, int kind, uint32_t spp, int ld_depth, uint64_t seed) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	c->lastPass.valid = false;      // mtsgpu_replay_roof regenerates a pass from its saved configuration: not across a setter
	if (kind < MTSGPU_SAMPLER_INDEPENDENT_KEYED || kind > MTSGPU_SAMPLER_STRATIFIED_KEYED) return fail(c, MTSGPU_EINVAL, "unknown sampler kind %d", kind);
	if (spp == 0) return fail(c, MTSGPU_EINVAL, "sampleCount must be > 0");
	if (kind == MTSGPU_SAMPLER_LD_KEYED && (roundToPow2(spp) > 65536u || ld_depth < 1 || ld_depth > 64))
		return fail(c, MTSGPU_EINVAL, "ldsampler: sampleCount <= 65536 and 1 <= depth <= 64 required");
	if (kind == MTSGPU_SAMPLER_STRATIFIED_KEYED && (spp > 65536u || squareAtLeast(spp) > 65536u || ld_depth < 1 || ld_depth > 64))
		return fail(c, MTSGPU_EINVAL, "stratified: sampleCount <= 65536 and 1 <= depth <= 64 required");
	c->samplerKind = kind; c->spp = spp; c->ldDepth = ld_depth > 0 ? ld_depth : 3; c->seed = seed;
	return 0;
}

int mtsgpu_set_tiles(mtsgpu_ctx *c, int block_size, int part, int n_parts) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	c->lastPass.valid = false;      // mtsgpu_replay_roof regenerates a pass from its saved configuration: not across a setter
	if (block_size <= 0 || n_parts <= 0 || part < 0 || part >= n_parts) return fail(c, MTSGPU_EINVAL, "bad tile sharding %d/%d/%d", block_size, part, n_parts);
	c->blockSize = block_size; c->part = part; c->nParts = n_parts;
	return 0;
}

int mtsgpu_set_rfilter(mtsgpu_ctx *c, float size_x, float size_y, const float *values) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	c->lastPass.valid = false;      // mtsgpu_replay_roof regenerates a pass from its saved configuration: not across a setter
	if (!values) { c->filtSizeX = c->filtSizeY = 0.5f; c->filtBorder = 0; return 0; }
	if (!(size_x > 0) || !(size_y > 0) || size_x > 8 || size_y > 8) return fail(c, MTSGPU_EINVAL, "bad filter size");
	HIPCHK(c, hipSetDevice(c->device));
	if (!c->filtValues) HIPCHK(c, hipMalloc((void **) &c->filtValues, 256 * sizeof(float)));
	HIPCHK(c, hipMemcpy(c->filtValues, values, 256 * sizeof(float), hipMemcpyHostToDevice));
	c->filtSizeX = size_x; c->filtSizeY = size_y;
	c->filtBorder = (int) std::ceil(std::max(size_x, size_y) - 0.5f);       // renderproc.cpp:143-144
	return 0;
}

int mtsgpu_set_film_edges(mtsgpu_ctx *c, int high_quality_edges) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	c->lastPass.valid = false;      // mtsgpu_replay_roof regenerates a pass from its saved configuration: not across a setter
	c->hqEdges = high_quality_edges != 0;
	return 0;
}

struct mtsgpu_loaded_mesh { mg::LoadedMesh m; };

int mtsgpu_load_serialized(const char *path, int shape_index, mtsgpu_loaded_mesh **out, mtsgpu_mesh *mesh) {
	if (!path || !out || !mesh) return fail(nullptr, MTSGPU_EINVAL, "null argument");
	*out = nullptr;
	mtsgpu_loaded_mesh *lm = new mtsgpu_loaded_mesh();
	try {
		loadSerializedMesh(path, shape_index, lm->m);
	} catch (const std::exception &e) {
		delete lm;
		return fail(nullptr, MTSGPU_EINVAL, "%s", e.what());
	}
	std::memset(mesh, 0, sizeof(*mesh));
	mesh->n_verts = (uint32_t) (lm->m.positions.size() / 3); mesh->n_tris = (uint32_t) (lm->m.triangles.size() / 3);
	mesh->positions = lm->m.positions.data();
	mesh->normals = lm->m.normals.empty() ? nullptr : lm->m.normals.data();
	mesh->triangles = lm->m.triangles.data();
	mesh->face_normals = lm->m.faceNormals ? 1 : 0;
	mesh->bsdf = -1; mesh->lum = -1; mesh->shape_type = MTSGPU_SHAPE_TRIMESH;
	*out = lm;
	return 0;
}

void mtsgpu_loaded_mesh_free(mtsgpu_loaded_mesh *m) { delete m; }

int mtsgpu_tabulate_filter(int kind, float half_size, float p0, float p1, float *size_xy, float *values) {
	if (!size_xy || !values || kind < 0 || kind > 4) return fail(nullptr, MTSGPU_EINVAL, "bad filter arguments");
	tabulateFilter(kind, half_size, p0, p1, size_xy, values);
	return 0;
}

int mtsgpu_set_film_buffer(mtsgpu_ctx *c, void *device_ptr) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	if (c->ownFilm && c->film) (void) hipFree(c->film);
	c->film = (float *) device_ptr; c->ownFilm = false; c->filmPixels = 0;
	return 0;
}

int mtsgpu_set_tuning(mtsgpu_ctx *c, const char *key, long value) {
	if (!c || !key) return fail(c, MTSGPU_EINVAL, "null argument");
	struct Knob { const char *key; long lo, hi; };
	static const Knob knobs[] = { { "refill_min", 1, 64 }, { "desc_min", 1, 64 }, { "leaf_min", 1, 64 }, { "batch", 0, 64 },
	                              { "dyn_div", 0, 1 << 20 }, { "test_retry", 0, 1 }, { "sync_free", -1, 1 }, { "overlap", 0, 2 }, { "overlap_delay_us", 0, 100000 }, { "merged", 0, 1 }, { "mailbox_free", 0, 1 }, { "chunk", 1, 1024 }, { "blocks_per_cu", 0, (long) kTraceBlocksPerCuMax }, { "plain_below", 0, 1 << 30 }, { "dyn_min_rounds", 0, 1 << 20 }, { "shade_fused", 0, 1 }, { "ray_queues", 0, 1 }, { "nee_parked", 0, 1 } };
	for (const Knob &k : knobs)
		if (std::strcmp(k.key, key) == 0) {
			if (value < k.lo || value > k.hi) return fail(c, MTSGPU_EINVAL, "tuning knob %s: %ld outside [%ld, %ld]", key, value, k.lo, k.hi);
			c->tuning[key] = value;
			applyTuning(c);
			return 0;
		}
	return fail(c, MTSGPU_EINVAL, "unknown tuning knob '%s'", key);
}

int mtsgpu_set_options(mtsgpu_ctx *c, uint64_t max_paths, int count_traversal, int time_kernels) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	c->maxPaths = max_paths; c->countTraversal = count_traversal != 0; c->timeKernels = time_kernels != 0;
	return 0;
}

static int ensureFilm(mtsgpu_ctx *c) {
	const size_t px = (size_t) c->cam.width * c->cam.height;
	if (c->film && !c->ownFilm) return 0;
	if (c->film && c->filmPixels == px) return 0;
	if (c->film) (void) hipFree(c->film);
	c->film = nullptr;
	HIPCHK(c, hipMalloc((void **) &c->film, px * 5 * sizeof(float)));
	HIPCHK(c, hipMemset(c->film, 0, px * 5 * sizeof(float)));
	c->ownFilm = true; c->filmPixels = px;
	return 0;
}

int mtsgpu_clear_film(mtsgpu_ctx *c) {
	int rc = checkReady(c); if (rc) return rc;
	rc = ensureFilm(c); if (rc) return rc;
	HIPCHK(c, hipMemsetAsync(c->film, 0, (size_t) c->cam.width * c->cam.height * 5 * sizeof(float), c->stream));
	return 0;
}

int mtsgpu_sync(mtsgpu_ctx *c) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int mtsgpu_render(mtsgpu_ctx *c, volatile const int *cancel) {
	int rc = checkReady(c); if (rc) return rc;
	rc = ensureFilm(c); if (rc) return rc;
	const int W = c->cam.width, H = c->cam.height, bs = c->blockSize;
	const uint32_t spp = effectiveSpp(c);
	// ImageBlock work units (imageproc.cpp:43-78) owned by this context: tile (tx, ty) -> part morton(tx, ty) % n_parts
	// Film::hasHighQualityEdges: the rendered rectangle grows by the filter border (renderproc.cpp:146-153)
	const int off = c->hqEdges ? -c->filtBorder : 0;
	const int RW = W - 2 * off, RH = H - 2 * off;
	const int tx = (RW + bs - 1) / bs, ty = (RH + bs - 1) / bs;
	// keys: index of the raster pixel in the (full film + border) grid; the crop window only moves the rectangle
	const int cx = c->cam.crop_offset_x, cy = c->cam.crop_offset_y;
	const int keyW = filmWidth(c) - 2 * off;
	if ((uint64_t) keyW * (uint64_t) (filmHeight(c) - 2 * off) > 0xFFFFFFFFull) return fail(c, MTSGPU_EINVAL, "film too large");
	std::vector<uint32_t> &pixels = c->renderPixels;
	std::vector<TileMeta> &tiles = c->renderTiles;
	const std::vector<long long> key = { W, H, bs, off, cx, cy, keyW, c->nParts, c->part };
	const bool reuse = c->renderListValid && key == c->renderKey && c->pixelList && c->pixelListCap >= pixels.size();
	if (!reuse) {
		pixels.clear(); tiles.clear();
		for (int t = 0; t < tx * ty; ++t) {
			if (tileMorton((uint32_t) (t % tx), (uint32_t) (t / tx)) % (uint32_t) c->nParts != (uint32_t) c->part) continue;
			const int x0 = off + (t % tx) * bs, y0 = off + (t / tx) * bs;            // inside the crop window
			TileMeta tm{};
			tm.x0 = x0 + cx; tm.y0 = y0 + cy; tm.w = std::min(bs, off + RW - x0); tm.h = std::min(bs, off + RH - y0);
			tm.slot_base = (uint32_t) pixels.size(); tm.block_index = (uint32_t) tiles.size();
			tm.colour = (uint32_t) (((t % tx) & 1) + 2 * ((t / tx) & 1));
			tiles.push_back(tm);
			for (int y = tm.y0; y < tm.y0 + tm.h; ++y)
				for (int x = tm.x0; x < tm.x0 + tm.w; ++x)
					pixels.push_back((uint32_t) (y - off) * (uint32_t) keyW + (uint32_t) (x - off));
		}
		c->renderKey = key; c->renderListValid = false;
	}
	std::memset(&c->stats, 0, sizeof(c->stats));
	c->traceEvUsed = c->shadeEvUsed = 0;
	if (pixels.empty()) return 0;
	if (!reuse) {
		rc = ensureBuf(c, &c->pixelList, &c->pixelListCap, pixels.size()); if (rc) return rc;
		HIPCHK(c, hipMemcpyAsync(c->pixelList, pixels.data(), pixels.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
		// valid only once the copy has landed: a frame that leaves early (cancel, an error) and a caller that then switches
		// streams (mtsgpu_set_stream) must not find a list that is marked valid but not ordered before its next use
		HIPCHK(c, hipStreamSynchronize(c->stream));
		c->renderListValid = true;
	}

	const bool wideFilter = c->filtBorder > 0;
	if (wideFilter && 2 * c->filtBorder > bs) return fail(c, MTSGPU_EINVAL, "filter border %d too wide for block size %d", c->filtBorder, bs);
	const uint64_t maxPaths = c->maxPaths ? c->maxPaths : defaultMaxPaths(c);
	size_t slotsPerPass = (size_t) std::max<uint64_t>(1, std::min<uint64_t>(pixels.size(), maxPaths / spp));
	if (wideFilter) slotsPerPass = std::max<size_t>(slotsPerPass, (size_t) bs * bs);     // passes hold whole tiles
	if ((uint64_t) slotsPerPass * spp > 0x7FFFFFFFull) return fail(c, MTSGPU_EINVAL, "pass too large");
	rc = ensurePaths(c, slotsPerPass * spp); if (rc) return rc;
	if (samplerHasTables(c)) {
		rc = ensureBuf(c, &c->ldScr, &c->ldScrCap, slotsPerPass * 3 * c->ldDepth); if (rc) return rc;
		rc = ensureBuf(c, &c->ldPerm, &c->ldPermCap, slotsPerPass * 2 * c->ldDepth * spp); if (rc) return rc;
		rc = ensureTableWork(c, slotsPerPass, spp); if (rc) return rc;
	}
	rc = ensureSampleArrays(c, slotsPerPass, slotsPerPass * spp); if (rc) return rc;
	if (c->countTraversal) HIPCHK(c, hipMemsetAsync(c->q.trace_counts, 0, kNumTraceCounts * sizeof(unsigned long long), c->stream));
	DConfig cfg = makeConfig(c, false);
	cfg.pix_w = keyW; cfg.pix_off = off;
	HIPCHK(c, hipMemsetAsync(c->pathLen, 0, sizeof(unsigned long long), c->stream));
	HIPCHK(c, hipMemsetAsync(c->devStats, 0, kNumDevStats * sizeof(unsigned long long), c->stream));
	c->devStatsUsed = false;
	const size_t fullBlock = (size_t) (bs + 2 * c->filtBorder) * (bs + 2 * c->filtBorder) * 5;
	if (wideFilter) {
		rc = ensureBuf(c, &c->tileMeta, &c->tileMetaCap, tiles.size()); if (rc) return rc;
		rc = ensureBuf(c, &c->blocks, &c->blocksCap, tiles.size() * fullBlock); if (rc) return rc;
	}

	// frame timing events, released on every way out of this function
	struct EventPair {
		hipEvent_t a = nullptr, b = nullptr;
		~EventPair() { if (a) (void) hipEventDestroy(a); if (b) (void) hipEventDestroy(b); }
	} frameEv;
	HIPCHK(c, hipEventCreate(&frameEv.a)); HIPCHK(c, hipEventCreate(&frameEv.b));
	const hipEvent_t t0 = frameEv.a, t1 = frameEv.b;
	HIPCHK(c, hipEventRecord(t0, c->stream));
	size_t tileCursor = 0;
	for (size_t base = 0; base < pixels.size();) {
		uint32_t nSlots;
		size_t tileFirst = tileCursor;
		if (wideFilter) {
			// whole tiles only: a block gathers from all samples of its tile
			size_t acc = 0;
			while (tileCursor < tiles.size() && (acc == 0 || acc + (size_t) tiles[tileCursor].w * tiles[tileCursor].h <= slotsPerPass)) {
				// slot inside this pass.  The cached tile list is updated in place: every tile passes through this line in
				// every frame before its record is uploaded below, so nothing of an earlier frame's passes survives
				tiles[tileCursor].slot_base = (uint32_t) acc;
				acc += (size_t) tiles[tileCursor].w * tiles[tileCursor].h;
				++tileCursor;
			}
			nSlots = (uint32_t) acc;
		} else {
			nSlots = (uint32_t) std::min(slotsPerPass, pixels.size() - base);
		}
		const uint32_t nPaths = nSlots * spp;
		if (samplerHasTables(c)) {
			launch_ld_tables(c->stream, cfg, c->pixelList + base, nSlots, c->ldScr, c->ldPerm, c->ldState, c->ldScratch);
			launch_sample_arrays(c->stream, cfg, nSlots, c->ldState);
		}
		rayQueues(c, nullptr, c->queueA);
		launch_generate(c->stream, c->dsc, c->paths, cfg, c->pixelList + base, nSlots, nullptr, nPaths, c->queueA);
		rayQueues(c, nullptr, nullptr);
		HIPCHK(c, hipGetLastError());
		c->lastPass.valid = true; c->lastPass.cfg = cfg; c->lastPass.base = base;
		c->lastPass.nSlots = nSlots; c->lastPass.nPaths = nPaths; c->lastPass.shadowMax = 0;
		rc = runBounces(c, cfg, nPaths, cancel);
		if (rc) return rc;
		if (wideFilter) {
			const uint32_t nT = (uint32_t) (tileCursor - tileFirst);
			HIPCHK(c, hipMemcpyAsync(c->tileMeta + tileFirst, tiles.data() + tileFirst, nT * sizeof(TileMeta), hipMemcpyHostToDevice, c->stream));
			launch_splat_blocks(c->stream, c->paths, cfg, c->tileMeta + tileFirst, nT, spp, bs, c->blocks);
			launch_path_lengths(c->stream, c->paths, nPaths, c->pathLen);
		} else {
			launch_accumulate(c->stream, c->paths, cfg, nSlots, spp, c->film, c->pathLen);
		}
		HIPCHK(c, hipGetLastError());
		c->stats.camera_samples += nPaths;
		base += nSlots;
	}
	if (wideFilter)
		for (uint32_t colour = 0; colour < 4; ++colour)
			launch_add_blocks(c->stream, cfg, c->tileMeta, (uint32_t) tiles.size(), colour, bs, c->blocks, c->film);
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipEventRecord(t1, c->stream));
	HIPCHK(c, hipMemcpyAsync(&c->hostCounters[kNumCounters * kCounterStride], c->pathLen, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	std::memcpy(&c->stats.path_length_sum, &c->hostCounters[kNumCounters * kCounterStride], sizeof(uint64_t));
	float ms = 0;
	HIPCHK(c, hipEventElapsedTime(&ms, t0, t1));
	c->stats.total_ms = ms;
	collectTimings(c);
	rc = collectDeviceStats(c); if (rc) return rc;
	if (c->countTraversal) { rc = fetchTraceCounts(c); if (rc) return rc; }
	return 0;
}

int mtsgpu_read_film(mtsgpu_ctx *c, float *rgbaw) {
	int rc = checkReady(c); if (rc) return rc;
	if (!rgbaw) return fail(c, MTSGPU_EINVAL, "null output");
	rc = ensureFilm(c); if (rc) return rc;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	HIPCHK(c, hipMemcpy(rgbaw, c->film, (size_t) c->cam.width * c->cam.height * 5 * sizeof(float), hipMemcpyDeviceToHost));
	return 0;
}

int mtsgpu_get_stats(mtsgpu_ctx *c, mtsgpu_stats *out) {
	if (!c || !out) return fail(c, MTSGPU_EINVAL, "null argument");
	*out = c->stats;
	return 0;
}

int mtsgpu_trace_rays(mtsgpu_ctx *c, const float *rays, uint32_t n, int shadow, uint32_t *hits) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	if (!c->haveScene) return fail(c, MTSGPU_ESTATE, "no scene uploaded");
	if (!rays || !hits) return fail(c, MTSGPU_EINVAL, "null argument");
	HIPCHK(c, hipSetDevice(c->device));
	std::memset(&c->stats, 0, sizeof(c->stats));
	c->traceEvUsed = c->shadeEvUsed = 0;
	if (n == 0) return 0;
	c->lastPass.valid = false;
	int rc = ensurePaths(c, n); if (rc) return rc;
	{
		// host rays -> ray_o / ray_d slots of the path records
		std::vector<float> od(8 * (size_t) n);
		std::memcpy(od.data(), rays, od.size() * sizeof(float));
		HIPCHK(c, hipMemcpy2DAsync(c->paths.base, kPathSlots * sizeof(float4), od.data(), 32, 32, n, hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
	}
	launch_iota(c->stream, c->queueA, n);
	if (c->countTraversal) HIPCHK(c, hipMemsetAsync(c->q.trace_counts, 0, kNumTraceCounts * sizeof(unsigned long long), c->stream));
	hipEvent_t *ev = c->timeKernels ? nextTraceEvents(c, shadow ? 2 : 0) : nullptr;
	if (ev) HIPCHK(c, hipEventRecord(ev[0], c->stream));
	HIPCHK(c, hipMemsetAsync(c->q.counters, 0, kNumCounters * kCounterStride * sizeof(uint32_t), c->stream));     // dynamic batch head
	launch_trace(c->stream, shadow ? 2 : 0, c->countTraversal, false, c->dsc, c->paths, c->q, c->queueA, n, false);
	if (ev) HIPCHK(c, hipEventRecord(ev[1], c->stream));
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipMemcpy2DAsync(hits, 16, c->paths.base + 2, kPathSlots * sizeof(float4), 16, n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	c->stats.trace_launches = 1;
	if (shadow) c->stats.rays_shadow = n; else c->stats.rays_closest = n;
	collectTimings(c);
	if (c->countTraversal) { rc = fetchTraceCounts(c); if (rc) return rc; }
	return 0;
}

// The replay roof of the closest-hit traversal kernel (include/mtsgpu.h).
int mtsgpu_replay_roof(mtsgpu_ctx *c, int kind, uint32_t n, uint32_t stride, int reps, double *out) {
	if (!c) return fail(nullptr, MTSGPU_EINVAL, "null context");
	if (!c->haveScene) return fail(c, MTSGPU_ESTATE, "no scene uploaded");
	if (!out || n == 0 || stride == 0 || reps < 1 || kind < 0 || kind > 2) return fail(c, MTSGPU_EINVAL, "bad argument");
	const mtsgpu_ctx::LastPass lp = c->lastPass;
	if (kind == 1 && (!lp.valid || (uint64_t) n * stride > lp.nPaths))
		return fail(c, MTSGPU_EINVAL, "replay roof: %u camera rays x stride %u exceed the pass rendered last (%u paths)", n, stride, lp.valid ? lp.nPaths : 0u);
	if (kind == 2 && (!lp.valid || (uint64_t) n * stride > lp.shadowMax))
		return fail(c, MTSGPU_EINVAL, "replay roof: %u shadow rays x stride %u exceed the shadow queue of the pass rendered last (%u rays; host-driven passes only)",
		            n, stride, lp.valid ? lp.shadowMax : 0u);
	if ((uint64_t) n * stride > c->pathCap || ((uint64_t) (n - 1) * stride + 1) * kPathSlots > (1ull << 29))
		return fail(c, MTSGPU_EINVAL, "replay roof: %u rays x stride %u exceed the path records in memory (%zu) or the 2^29 slots a recorded index can name", n, stride, c->pathCap);
	HIPCHK(c, hipSetDevice(c->device));
	hipStream_t s = c->stream;
	const int mode = kind == 2 ? 1 : 0;
	const bool coherent = kind == 1, bin = kind != 2;      // as the bounces launch this class of rays
	const uint32_t cap = 256;                  // requests kept per ray (C3: 45 on average; longer lists are truncated and counted); % 4 == 0
	const uint32_t nBatches = (n + 63u) / 64u;
	uint32_t *rec = nullptr, *recLen = nullptr, *order = nullptr, *tr = nullptr, *batchLen = nullptr, *sink = nullptr;
	float4 *tmp = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	auto cleanup = [&]() {
		for (uint32_t *p : { rec, recLen, order, tr, batchLen, sink }) if (p) (void) hipFree(p);
		if (tmp) (void) hipFree(tmp);
		if (e0) (void) hipEventDestroy(e0);
		if (e1) (void) hipEventDestroy(e1);
		c->q.rec = c->q.rec_len = nullptr; c->q.rec_cap = 0;
	};
	#define RR_CHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); \
		return fail(c, MTSGPU_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)
	RR_CHK(hipMalloc((void **) &rec, (size_t) n * cap * 4)); RR_CHK(hipMalloc((void **) &recLen, (size_t) n * 4));
	RR_CHK(hipMalloc((void **) &order, (size_t) n * 4)); RR_CHK(hipMalloc((void **) &tr, (size_t) nBatches * cap * 64 * 4));
	RR_CHK(hipMalloc((void **) &batchLen, (size_t) nBatches * 4)); RR_CHK(hipMalloc((void **) &sink, 4));
	RR_CHK(hipEventCreate(&e0)); RR_CHK(hipEventCreate(&e1));
	const size_t counterBytes = kNumCounters * kCounterStride * sizeof(uint32_t);
	c->q.counters = c->counterSets; c->q.spill = mode == 1 ? c->spillShadow : c->spillClosest; c->q.dev_stats = nullptr;
	c->q.next = c->queueB;
	// the rays of the sample
	if (kind == 1) {
		// the camera rays of the pass rendered last, once more (its sampler tables are still in place)
		launch_generate(s, c->dsc, c->paths, lp.cfg, c->pixelList + lp.base, lp.nSlots, nullptr, lp.nPaths, c->queueA);
		RR_CHK(hipGetLastError());
		c->lastPass.valid = false;             // the path records no longer hold that pass
	}
	if (kind == 2) {
		// any-hit rays are addressed by their queue position: the sampled slots move to the front of the shadow queue
		RR_CHK(hipMalloc((void **) &tmp, (size_t) n * 3 * sizeof(float4)));
		launch_gather_strided(s, tmp, c->paths.shq_o, n, stride); launch_gather_strided(s, tmp + n, c->paths.shq_d, n, stride);
		launch_gather_strided(s, tmp + 2 * (size_t) n, c->paths.shq_nee, n, stride);
		RR_CHK(hipMemcpyAsync(c->paths.shq_o, tmp, (size_t) n * sizeof(float4), hipMemcpyDeviceToDevice, s));
		RR_CHK(hipMemcpyAsync(c->paths.shq_d, tmp + n, (size_t) n * sizeof(float4), hipMemcpyDeviceToDevice, s));
		RR_CHK(hipMemcpyAsync(c->paths.shq_nee, tmp + 2 * (size_t) n, (size_t) n * sizeof(float4), hipMemcpyDeviceToDevice, s));
		c->lastPass.shadowMax = 0;             // the queue is no longer the frame's
	} else {
		// every stride-th path record, its ray copied into queue order the way the bounces hand rays to this kernel
		launch_iota_strided(s, c->queueA, n, stride);
		launch_gather_strided(s, c->rayqA[0], c->paths.base + 0, n, stride * kPathSlots);
		launch_gather_strided(s, c->rayqA[1], c->paths.base + 1, n, stride * kPathSlots);
		rayQueues(c, c->queueA, nullptr);
	}
	RayQueuesOff rqOff{ c };
	const uint32_t *queue = mode == 1 ? c->q.shadow : c->queueA;
	// 1. the counting kernel records what every ray asks for
	RR_CHK(hipMemsetAsync(recLen, 0, (size_t) n * 4, s));
	RR_CHK(hipMemsetAsync(c->q.trace_counts, 0, kNumTraceCounts * sizeof(unsigned long long), s));
	RR_CHK(hipMemsetAsync(c->q.counters, 0, counterBytes, s));
	c->q.rec = rec; c->q.rec_len = recLen; c->q.rec_cap = cap;
	launch_trace(s, mode, true, bin, c->dsc, c->paths, c->q, queue, n, coherent);
	c->q.rec = c->q.rec_len = nullptr; c->q.rec_cap = 0;
	RR_CHK(hipGetLastError());
	std::vector<uint32_t> len(n);
	RR_CHK(hipMemcpyAsync(len.data(), recLen, (size_t) n * 4, hipMemcpyDeviceToHost, s));
	RR_CHK(hipStreamSynchronize(s));
	unsigned long long cnt[kNumTraceCounts];
	RR_CHK(hipMemcpy(cnt, c->q.trace_counts, sizeof(cnt), hipMemcpyDeviceToHost));
	// 2. rays by decreasing list length, 64 to a batch (lanes of a replay wave then finish together); a counting sort
	std::vector<uint32_t> ord(n), first(cap + 2, 0u);
	unsigned long long total = 0, truncated = 0;
	for (uint32_t i = 0; i < n; ++i) { const uint32_t l = std::min(len[i], cap); total += l; if (len[i] > cap) ++truncated; first[cap - l + 1]++; }
	for (uint32_t l = 0; l <= cap; ++l) first[l + 1] += first[l];
	for (uint32_t i = 0; i < n; ++i) ord[first[cap - std::min(len[i], cap)]++] = i;
	RR_CHK(hipMemcpyAsync(order, ord.data(), (size_t) n * 4, hipMemcpyHostToDevice, s));
	launch_build_replay(s, rec, recLen, order, n, cap, tr, batchLen);
	RR_CHK(hipGetLastError());
	// 3. the product kernel and the replay on the same rays, best of `reps`
	float prodMs = 1e30f, replayMs = 1e30f;
	for (int i = 0; i <= reps; ++i) {            // round 0 warms up
		RR_CHK(hipMemsetAsync(c->q.counters, 0, counterBytes, s));
		RR_CHK(hipEventRecord(e0, s));
		launch_trace(s, mode, false, bin, c->dsc, c->paths, c->q, queue, n, coherent);
		RR_CHK(hipEventRecord(e1, s));
		RR_CHK(hipEventSynchronize(e1));
		float ms = 0; RR_CHK(hipEventElapsedTime(&ms, e0, e1));
		if (i && ms < prodMs) prodMs = ms;
		RR_CHK(hipEventRecord(e0, s));
		launch_replay(s, c->dsc, c->paths, c->q, tr, batchLen, nBatches, cap, 0u, sink);
		RR_CHK(hipEventRecord(e1, s));
		RR_CHK(hipEventSynchronize(e1));
		RR_CHK(hipEventElapsedTime(&ms, e0, e1));
		if (i && ms < replayMs) replayMs = ms;
	}
	#undef RR_CHK
	cleanup();
	out[0] = (double) n; out[1] = (double) total; out[2] = (double) truncated;
	out[3] = prodMs; out[4] = replayMs;
	out[5] = (double) cnt[kCntPairGlobal]; out[6] = (double) cnt[kCntPairLds]; out[7] = (double) cnt[kCntNodeGlobal]; out[8] = (double) cnt[kCntNodeLds];
	out[9] = (double) cnt[kCntHead]; out[10] = (double) cnt[kCntTail]; out[11] = (double) cnt[kCntSpill];
	return 0;
}

int mtsgpu_ld_tables(mtsgpu_ctx *c, uint32_t pixel_key, float *out1d, float *out2d) {
	if (!c || !out1d || !out2d) return fail(c, MTSGPU_EINVAL, "null argument");
	if (c->samplerKind != MTSGPU_SAMPLER_LD_KEYED) return fail(c, MTSGPU_ESTATE, "sampler is not the low-discrepancy sampler");
	HIPCHK(c, hipSetDevice(c->device));
	const uint32_t spp = effectiveSpp(c);
	const int depth = c->ldDepth;
	int rc = ensureBuf(c, &c->ldScr, &c->ldScrCap, (size_t) 3 * depth); if (rc) return rc;
	rc = ensureBuf(c, &c->ldPerm, &c->ldPermCap, (size_t) 2 * depth * spp); if (rc) return rc;
	rc = ensureTableWork(c, 1, spp); if (rc) return rc;
	c->renderListValid = false; c->lastPass.valid = false;
	rc = ensureBuf(c, &c->pixelList, &c->pixelListCap, 1); if (rc) return rc;
	HIPCHK(c, hipMemcpyAsync(c->pixelList, &pixel_key, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	DConfig cfg{};
	cfg.spp = spp; cfg.ld_depth = depth; cfg.seed = c->seed; cfg.sampler_kind = 1;
	launch_ld_tables(c->stream, cfg, c->pixelList, 1, c->ldScr, c->ldPerm, c->ldState, c->ldScratch);
	HIPCHK(c, hipGetLastError());
	std::vector<uint32_t> scr((size_t) 3 * depth);
	std::vector<uint16_t> perm((size_t) 2 * depth * spp);
	HIPCHK(c, hipMemcpyAsync(scr.data(), c->ldScr, scr.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpyAsync(perm.data(), c->ldPerm, perm.size() * sizeof(uint16_t), hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	// the u32 -> f32 step of ldsampler.cpp:111,117 on the device-generated integer tables
	for (int i = 0; i < depth; ++i)
		for (uint32_t k = 0; k < spp; ++k) {
			out1d[(size_t) i * spp + k] = u32ToUnit(vdcBits(perm[((size_t) 2 * i) * spp + k], scr[3 * i]));
			const uint32_t p = perm[((size_t) 2 * i + 1) * spp + k];
			out2d[((size_t) i * spp + k) * 2 + 0] = u32ToUnit(vdcBits(p, scr[3 * i + 1]));
			out2d[((size_t) i * spp + k) * 2 + 1] = u32ToUnit(sobol2Bits(p, scr[3 * i + 2]));
		}
	return 0;
}

int mtsgpu_random_values(mtsgpu_ctx *c, int op, uint64_t seed, uint64_t arg, uint32_t clone, uint32_t n, uint64_t *out) {
	if (!c || !out) return fail(c, MTSGPU_EINVAL, "null argument");
	if (op < 0 || op > 3 || n == 0 || n > (1u << 20) || clone > 64 || (op == 2 && arg == 0)) return fail(c, MTSGPU_EINVAL, "bad Random request");
	HIPCHK(c, hipSetDevice(c->device));
	void *state = nullptr; unsigned long long *dOut = nullptr;
	HIPCHK(c, hipMalloc(&state, random_state_bytes()));
	hipError_t e = hipMalloc((void **) &dOut, (size_t) n * sizeof(unsigned long long));
	if (e == hipSuccess) {
		launch_random_values(c->stream, state, op, seed, arg, clone, n, dOut);
		e = hipGetLastError();
		if (e == hipSuccess) e = hipMemcpyAsync(out, dOut, (size_t) n * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	}
	(void) hipFree(state); if (dOut) (void) hipFree(dOut);
	if (e != hipSuccess) return fail(c, MTSGPU_EHIP, "Random on the device: %s", hipGetErrorString(e));
	return 0;
}

int mtsgpu_sampler_values(mtsgpu_ctx *c, uint32_t pixel_key, uint32_t sample_index, uint32_t n, int two_d, float *out) {
	if (!c || !out) return fail(c, MTSGPU_EINVAL, "null argument");
	if (n == 0) return 0;
	const uint32_t spp = effectiveSpp(c);
	if (n > 4096u || sample_index >= spp) return fail(c, MTSGPU_EINVAL, "sample index or count out of range");
	HIPCHK(c, hipSetDevice(c->device));
	c->renderListValid = false; c->lastPass.valid = false;
	int rc = ensureBuf(c, &c->pixelList, &c->pixelListCap, 1); if (rc) return rc;
	HIPCHK(c, hipMemcpyAsync(c->pixelList, &pixel_key, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	if (samplerHasTables(c)) {
		rc = ensureBuf(c, &c->ldScr, &c->ldScrCap, (size_t) 3 * c->ldDepth); if (rc) return rc;
		rc = ensureBuf(c, &c->ldPerm, &c->ldPermCap, (size_t) 2 * c->ldDepth * spp); if (rc) return rc;
		rc = ensureTableWork(c, 1, spp); if (rc) return rc;
	}
	const int integratorSaved = c->integrator;
	c->integrator = 0;                               // no sample arrays: the plain next1D / next2D sequence
	const DConfig cfg = makeConfig(c, true);
	c->integrator = integratorSaved;
	if (samplerHasTables(c))
		launch_ld_tables(c->stream, cfg, c->pixelList, 1, c->ldScr, c->ldPerm, c->ldState, c->ldScratch);
	float *dOut = nullptr;
	const size_t nOut = (size_t) n * (two_d ? 2 : 1);
	HIPCHK(c, hipMalloc((void **) &dOut, nOut * sizeof(float)));
	launch_sampler_values(c->stream, cfg, pixel_key, sample_index, n, two_d, dOut);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess) e = hipMemcpyAsync(out, dOut, nOut * sizeof(float), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	(void) hipFree(dOut);
	if (e != hipSuccess) return fail(c, MTSGPU_EHIP, "sampler read-out failed: %s", hipGetErrorString(e));
	return 0;
}

int mtsgpu_bsdf_eval(mtsgpu_ctx *c, uint32_t bsdf_type, const float *params, int op, uint32_t n, const float *queries, float *out) {
	if (!c || !params || !queries || !out) return fail(c, MTSGPU_EINVAL, "null argument");
	if ((bsdf_type & 0xFFu) >= (uint32_t) MTSGPU_BSDF_NTYPES || (bsdf_type & ~(0xFFu | (uint32_t) MTSGPU_BSDF_TWOSIDED)) || op < 0 || op > 2)
		return fail(c, MTSGPU_EINVAL, "bad BSDF type or operation");
	if (n == 0) return 0;
	if (n > (1u << 24)) return fail(c, MTSGPU_EINVAL, "at most 2^24 query records per call");
	HIPCHK(c, hipSetDevice(c->device));
	float *dQ = nullptr, *dOut = nullptr;
	HIPCHK(c, hipMalloc((void **) &dQ, (size_t) n * 6 * sizeof(float)));
	hipError_t e = hipMalloc((void **) &dOut, (size_t) n * 8 * sizeof(float));
	if (e == hipSuccess) e = hipMemcpyAsync(dQ, queries, (size_t) n * 6 * sizeof(float), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) {
		launch_bsdf_eval(c->stream, bsdf_type, params, op, n, dQ, dOut);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(out, dOut, (size_t) n * 8 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	(void) hipFree(dQ); if (dOut) (void) hipFree(dOut);
	if (e != hipSuccess) return fail(c, MTSGPU_EHIP, "BSDF read-out failed: %s", hipGetErrorString(e));
	return 0;
}

int mtsgpu_li_samples(mtsgpu_ctx *c, const uint32_t *pix_samples, uint32_t n, float *out) {
	int rc = checkReady(c); if (rc) return rc;
	if (!pix_samples || !out) return fail(c, MTSGPU_EINVAL, "null argument");
	std::memset(&c->stats, 0, sizeof(c->stats));
	c->traceEvUsed = c->shadeEvUsed = 0;
	if (n == 0) return 0;
	const uint32_t spp = effectiveSpp(c);
	std::vector<uint32_t> keys(n);
	for (uint32_t i = 0; i < n; ++i) {
		const uint32_t x = pix_samples[3 * (size_t) i], y = pix_samples[3 * (size_t) i + 1], j = pix_samples[3 * (size_t) i + 2];
		if (x >= (uint32_t) c->cam.width || y >= (uint32_t) c->cam.height || j >= spp)
			return fail(c, MTSGPU_EINVAL, "sample %u: pixel or sample index out of range", i);
		keys[i] = (y + (uint32_t) c->cam.crop_offset_y) * (uint32_t) filmWidth(c) + x + (uint32_t) c->cam.crop_offset_x;
	}
	rc = ensurePaths(c, n); if (rc) return rc;
	rc = ensureBuf(c, &c->explicitSamples, &c->explicitCap, 3 * (size_t) n); if (rc) return rc;
	c->renderListValid = false; c->lastPass.valid = false;
	rc = ensureBuf(c, &c->pixelList, &c->pixelListCap, n); if (rc) return rc;
	HIPCHK(c, hipMemcpyAsync(c->explicitSamples, pix_samples, 3 * (size_t) n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipMemcpyAsync(c->pixelList, keys.data(), (size_t) n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
	if (samplerHasTables(c)) {
		rc = ensureBuf(c, &c->ldScr, &c->ldScrCap, (size_t) n * 3 * c->ldDepth); if (rc) return rc;
		rc = ensureBuf(c, &c->ldPerm, &c->ldPermCap, (size_t) n * 2 * c->ldDepth * spp); if (rc) return rc;
		rc = ensureTableWork(c, n, spp); if (rc) return rc;
	}
	rc = ensureSampleArrays(c, n, n); if (rc) return rc;
	const DConfig cfg = makeConfig(c, true);
	if (samplerHasTables(c)) {
		launch_ld_tables(c->stream, cfg, c->pixelList, n, c->ldScr, c->ldPerm, c->ldState, c->ldScratch);
		launch_sample_arrays(c->stream, cfg, n, c->ldState);
	}
	rayQueues(c, nullptr, c->queueA);
	launch_generate(c->stream, c->dsc, c->paths, cfg, c->pixelList, n, c->explicitSamples, n, c->queueA);
	rayQueues(c, nullptr, nullptr);
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipMemsetAsync(c->devStats, 0, kNumDevStats * sizeof(unsigned long long), c->stream));
	c->devStatsUsed = false;
	rc = runBounces(c, cfg, n, nullptr); if (rc) return rc;
	std::vector<float> Li(4 * (size_t) n), thr(4 * (size_t) n), spos(4 * (size_t) n), parked(4 * (size_t) n);
	const size_t pitch = kPathSlots * sizeof(float4);
	HIPCHK(c, hipMemcpy2DAsync(Li.data(), 16, c->paths.base + 4, pitch, 16, n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpy2DAsync(parked.data(), 16, c->paths.base + 2, pitch, 16, n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpy2DAsync(thr.data(), 16, c->paths.base + 3, pitch, 16, n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipMemcpy2DAsync(spos.data(), 16, c->paths.base + 7, pitch, 16, n, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	for (size_t i = 0; i < n; ++i) {
		uint32_t flags; int depth;
		std::memcpy(&flags, &Li[4 * i + 3], 4); std::memcpy(&depth, &thr[4 * i + 3], 4);
		float *o = out + 8 * i;
		// a direct-light term still parked in the record of a path that has ended (DQueues::nee_parked)
		const float4 L = settled_Li(make_float4(Li[4 * i], Li[4 * i + 1], Li[4 * i + 2], 0.0f),
		                            make_float4(parked[4 * i], parked[4 * i + 1], parked[4 * i + 2], parked[4 * i + 3]));
		o[0] = L.x; o[1] = L.y; o[2] = L.z; o[3] = (flags & F_ALPHA) ? 1.0f : 0.0f;
		o[4] = spos[4 * i]; o[5] = spos[4 * i + 1]; o[6] = (float) depth; o[7] = 0.0f;
	}
	collectTimings(c);
	return collectDeviceStats(c);
}

// --- host-side flattening ------------------------------------------------------
struct mtsgpu_flat_scene { FlatScene fs; };

int mtsgpu_tail_filter_flag(const uint32_t *triaccel12, const float *box_min, const float *box_max, float margin) {
	if (!triaccel12 || !box_min || !box_max) return 0;
	return tailFilterFlag(triaccel12, box_min, box_max, margin) ? 1 : 0;
}

int mtsgpu_flatten(const mtsgpu_scene_desc *desc, const mtsgpu_kd_params *kd, mtsgpu_flat_scene **out) {
	if (!desc || !out) return fail(nullptr, MTSGPU_EINVAL, "null argument");
	*out = nullptr;
	try {
		std::unique_ptr<mtsgpu_flat_scene> p(new mtsgpu_flat_scene());
		flattenScene(*desc, kd, p->fs);
		*out = p.release();
	} catch (const std::exception &e) {
		return fail(nullptr, MTSGPU_EINVAL, "%s", e.what());
	}
	return 0;
}

const mtsgpu_scene *mtsgpu_flat_scene_get(const mtsgpu_flat_scene *fs) { return fs ? &fs->fs.sc : nullptr; }
void mtsgpu_flat_scene_free(mtsgpu_flat_scene *fs) { delete fs; }

int mtsgpu_flat_scene_kdstats(const mtsgpu_flat_scene *fs, double *out6) {
	if (!fs || !out6) return fail(nullptr, MTSGPU_EINVAL, "null argument");
	for (int i = 0; i < 6; ++i) out6[i] = fs->fs.kd.stats[i];
	return 0;
}

int mtsgpu_make_camera(const float origin[3], const float target[3], const float up[3],
                       float fov_deg, int width, int height, mtsgpu_camera *out) {
	if (!origin || !target || !up || !out || width <= 0 || height <= 0) return fail(nullptr, MTSGPU_EINVAL, "bad camera arguments");
	makeCamera(origin, target, up, fov_deg, width, height, *out);
	return 0;
}

int mtsgpu_make_camera_ortho(const float origin[3], const float target[3], const float up[3],
                             float scale_x, float scale_y, int width, int height, mtsgpu_camera *out) {
	if (!origin || !target || !up || !out || width <= 0 || height <= 0 || !(scale_x > 0) || !(scale_y > 0))
		return fail(nullptr, MTSGPU_EINVAL, "bad camera arguments");
	makeCameraOrtho(origin, target, up, scale_x, scale_y, width, height, *out);
	return 0;
}

} // extern "C"
