// kernels.h -- device-side data layout and launch wrappers of the wavefront path tracer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mg {

constexpr int kNumBsdfTypes = 8;      // lambertian, dielectric, roughmetal, microfacet, mirror, phong, roughglass, difftrans
constexpr int kNumBins = kNumBsdfTypes + 1;   // + "terminal" (miss / no BSDF)
// Threads per traversal workgroup and resident workgroups per CU.  Round 3: 512 threads, 3 workgroups for closest-hit rays
// (24 waves per CU, up to 80 VGPRs) and 4 for shadow rays (32 waves, 64 VGPRs), instead of 7 / 8 workgroups of 256: the
// LDS copy of the top of the tree is shared by 8 waves instead of 4, so the same LDS holds 10 levels of it (1 024 sibling
// pairs) instead of 8 -- every ray's descent from the root and every pop that lands there skip two more vector-memory
// requests, which buys more than the four waves per CU cost (205 -> 199 ms per C3 frame,
// profiles/r03e_exp_trace_workgroups.txt).  Workgroups whose wave count is not a multiple of four (384, 896 threads)
// lose a third of the CU's wave slots and are out.  MG_TRACE_BLOCK / MG_TRACE_WGS override the rule in experiment builds.
#ifndef MG_TRACE_BLOCK
#define MG_TRACE_BLOCK 512
#endif
constexpr int kTraceBlock = MG_TRACE_BLOCK;
static_assert(kTraceBlock % 256 == 0 && kTraceBlock <= 1024, "whole multiples of four waves");
#ifdef MG_TRACE_WGS
constexpr unsigned trace_blocks_per_cu(int) { return MG_TRACE_WGS; }
#else
constexpr unsigned trace_blocks_per_cu(int mode) { return (mode == 0 ? 24u : 32u) / (kTraceBlock / 64); }
#endif
constexpr unsigned trace_waves_per_simd(int mode) { return trace_blocks_per_cu(mode) * (kTraceBlock / 64) / 4; }
constexpr unsigned kTraceBlocksPerCuMax = 2048 / kTraceBlock;      // largest persistent traversal grid: 32 waves per CU
// LDS layout of the traversal kernels (trace.hip; k_replay of measure.hip keeps the same footprint)
#ifndef MG_STACK_LDS
#define MG_STACK_LDS 10
#endif
#ifndef MG_TOP_PAIRS
#define MG_TOP_PAIRS (MG_TRACE_BLOCK >= 512 ? 1024 : 128)
#endif
constexpr int kStackLDS = MG_STACK_LDS;       // stack levels kept in LDS (deeper ones spill: 1 push in 10^4 at 12 levels on C3)
// The first 2 * kTopPairs device nodes -- the root and the sibling pairs below it in breadth-first order, see
// mtsgpu_upload_scene -- are copied into LDS by every workgroup: each ray's descent from the root starts with 8-9
// levels that every other ray visits too, and a vector-memory request costs the CU ~0.5-1 ns per lane where an LDS
// read costs ~0.05 (profiles/r02_ta_gather_microbench.txt).  0 switches the cache off.
constexpr uint32_t kTopPairs = MG_TOP_PAIRS;
// The closest-hit kernel of host-driven bounces runs WITHOUT the hashed mailbox (trace.hip: TIE) and spends the 16 KB on one more
// level of the tree: 2 048 sibling pairs.  The breadth-first prefix of the device tree covers the larger of the two copies.
#ifndef MG_TOP_PAIRS_TIE
#define MG_TOP_PAIRS_TIE (2 * MG_TOP_PAIRS)
#endif
constexpr uint32_t kTopPairsTie = MG_TOP_PAIRS_TIE;
constexpr uint32_t kTopPairsMax = kTopPairsTie > kTopPairs ? kTopPairsTie : kTopPairs;
constexpr int kSpillLevels = 50 - kStackLDS;      // LDS + spill levels = MTS_KD_MAXDEPTH (48, gkdtree.h:35) + 2
static_assert(kStackLDS >= 1 && kStackLDS + kSpillLevels >= 48 + 2, "the traversal stack must hold every tree the reference can build");
constexpr uint32_t kSentinel = 0xFFFFFFFFu;
constexpr uint32_t kNullNode = 0xFFFFFFFFu;
constexpr uint32_t kNoPrim = 0xFFFFFFFFu;
// uint4 per leaf record: the 48-byte TriAccel records lie back to back (2.67 per 128-byte line; three of eight straddle two
// lines).  Measured and rejected (round 4, profiles/r04d_exp_trace_leaf_record_fetch.txt): one record per 64 bytes, so that
// a tail always lies on the line of its head: 194.2 -> 196.7 ms of traversal per C3 frame (fewer heads per line).
constexpr uint32_t kLeafStride = 3;
constexpr int kTriStride = 8;         // float4 per primitive record: one 128-byte line holds positions AND normals
constexpr int kLumStride = 32;        // MTSGPU_LUM_NPARAMS
constexpr int kCounterStride = 32;    // one 128-byte line per queue counter (atomics on one line serialise)
constexpr int kBsdfNParams = 16;      // MTSGPU_BSDF_NPARAMS
constexpr int kBinShards = 16;        // the closest-hit kernel appends to bins[b] through 16 independent segments
constexpr int kNumCounters = kNumBins * kBinShards + 5;   // bins x shards, next, shadow, dynamic heads of the two traversal launches, redo list
// Two sets of counters, used by alternate bounces: the shadow rays of bounce b are traced (second stream) while the
// closest-hit launch of bounce b + 1 already fills the next set
constexpr int kCounterSets = 2;
constexpr int kCntNext = kNumBins * kBinShards, kCntShadow = kCntNext + 1, kCntDynClosest = kCntNext + 2, kCntDynShadow = kCntNext + 3, kCntRedo = kCntNext + 4;
// word offsets of the two queue counters of k_shade inside a counter set (one 128-byte line each)
constexpr int kNextWord = kCntNext * kCounterStride, kShadowWord = kCntShadow * kCounterStride, kRedoWord = kCntRedo * kCounterStride;
#ifndef MG_SHADE_BLOCK
#define MG_SHADE_BLOCK 1024     // 512: two atomics-bound milliseconds more per 64-spp frame (one reservation per workgroup)
#endif
constexpr int kShadeBlock = MG_SHADE_BLOCK;
// device-side frame statistics (u64): rays of the closest-hit / shadow launches, segment-overflow flag, non-empty
// traversal launches -- what the host counts itself when it reads the queue sizes back every bounce
constexpr int kNumTraceCounts = 16;
enum { kCntPairGlobal = 8, kCntPairLds = 9, kCntNodeGlobal = 10, kCntNodeLds = 11, kCntTail = 12, kCntSpill = 13, kCntHead = 14 };
// request kinds of a recorded ray (DQueues::rec): index in units of the access size into the node array (16-byte sibling
// pairs, 8-byte nodes), the leaf records (16 bytes) or the path records (16-byte slots; loads of the ray, store of the hit)
enum : uint32_t { kReqPair = 1, kReqNode = 2, kReqLeaf = 3, kReqRay = 4, kReqHit = 5, kReqNone = 0xFFFFFFFFu };
enum { kStatClosest = 0, kStatShadow = 1, kStatOverflow = 2, kStatLaunches = 3, kNumDevStats = 4 };

// Scene in HBM (all pointers are device pointers); see DESIGN.md section 3
struct DScene {
	const uint2    *nodes;        // 8 B: (left child index << 2 | axis, split) or (1 << 31 | first record, end record)
	// TriAccel records in LEAF ORDER: entry e of the kd-tree index list holds the 48-byte
	// TriAccel of primitive kd_indices[e] (dword 0 = k << 30 | "not an occluder" << 29 | global
	// primitive id, dword 10 = shape), so a leaf's primitives are one contiguous run
	const uint4    *leaf_ta;
	// per-primitive gather record, one 128-byte line (kTriStride float4): p0.xyz p1.xyz p2.xyz -, shape, flags |
	// n0.xyz n1.xyz n2.xyz | pad -- one line instead of 3 index + 18 scattered vertex fetches; tri_nrm = tri_pos + 3
	const float4   *tri_pos;
	const float4   *tri_nrm;
	const int32_t  *shape_bsdf, *shape_lum;
	const uint32_t *shape_bin;    // per shape: the material queue of its hits = BSDF type, or kNumBsdfTypes without a BSDF
	const uint32_t *shape_type;   // MTSGPU_SHAPE_*
	const float    *shape_params; // [n_shapes][24]
	// environment map (level 0 of the MIPMap, RGB) and its sampling density (envmap.cpp:95-110)
	const float    *env_pixels, *env_pdf, *env_cdf;
	uint32_t        env_width, env_height, env_pdf_width, env_pdf_height;
	uint32_t        has_shapes;   // the scene has non-triangle shapes (lets the traversal skip their test with one scalar branch)
	const uint32_t *shape_flags, *shape_tri_offset;
	const uint32_t *bsdf_type;
	const float    *bsdf_params;
	const uint32_t *lum_type;
	const float    *lum_params;
	const int32_t  *lum_shape;
	const float    *lum_inv_area;
	const uint32_t *lum_cdf_offset;
	const float    *lum_tri_cdf, *lum_sel_cdf, *lum_sel_pdf;
	float lum_sel_sum;
	int32_t background_lum;
	uint32_t n_lums, n_nodes, n_tris, n_shapes;
	float aabb_min[3], aabb_max[3];
	float tail_margin;            // record-tail filter of k_trace (api.cpp: tailFilterFlag): how far beyond a leaf's face a plane point must lie
};

// What k_trace needs of the scene (a kernel argument: the fewer scalar registers it pins, the fewer get spilled)
struct DTraceScene {
	const uint2 *nodes;
	const uint4 *leaf_ta;
	const uint32_t *shape_bin;
	uint32_t has_shapes;
	float aabb_min[3], aabb_max[3];
	float tail_margin;
};
inline DTraceScene trace_scene(const DScene &sc) {
	DTraceScene t;
	t.nodes = sc.nodes; t.leaf_ta = sc.leaf_ta; t.shape_bin = sc.shape_bin; t.has_shapes = sc.has_shapes; t.tail_margin = sc.tail_margin;
	for (int i = 0; i < 3; ++i) { t.aabb_min[i] = sc.aabb_min[i]; t.aabb_max[i] = sc.aabb_max[i]; }
	return t;
}

// Per-path state, indexed by path id (paths never move; queues hold ids).
// One 128-byte record per path = one cache line, so that the id-indexed gathers of the shading and
// traversal kernels cost one L1 miss instead of one per field:
//   ray_o, ray_d, hit, thr, Li, bsdf, misc, spos   (everything a bounce reads)
// The pending shadow ray does not live here: it is stored in shadow-queue order (shq_*).
constexpr int kPathSlots = 8;         // float4 slots per record: one 128-byte line
struct DPaths {
	float4 *base;
	__host__ __device__ float4 &slot(size_t id, int k) const { return base[id * kPathSlots + k]; }
	__host__ __device__ float4 &ray_o(size_t id) const { return slot(id, 0); }   // o.xyz, mint
	__host__ __device__ float4 &ray_d(size_t id) const { return slot(id, 1); }   // d.xyz, maxt
	__host__ __device__ uint4  &hit(size_t id) const { return reinterpret_cast<uint4 &>(slot(id, 2)); }  // t, u, v bits, prim
	__host__ __device__ float4 &thr(size_t id) const { return slot(id, 3); }     // throughput rgb, w = depth (int bits)
	__host__ __device__ float4 &Li(size_t id) const { return slot(id, 4); }      // Li rgb, w = flags (uint bits)
	__host__ __device__ float4 &bsdf(size_t id) const { return slot(id, 5); }    // bsdfVal/pdf rgb, w = bsdfPdf
	__host__ __device__ uint4  &misc(size_t id) const { return reinterpret_cast<uint4 &>(slot(id, 6)); } // rng lo, rng hi, sample idx, pixel key
	__host__ __device__ float4 &spos(size_t id) const { return slot(id, 7); }    // raster position x, y
	// shadow rays live in queue order, not in the record: written coalesced by k_shade at the slot the stream
	// compaction assigns, read coalesced by k_trace<shadow>
	float4 *shq_o, *shq_d;        // origin p1, direction p2 - p1
	float4 *shq_nee;              // pending direct-light contribution rgb, w = path id (uint bits)
	// MIDirectIntegrator with several BSDF samples: the camera ray and its hit, [id][3] (ray_o, ray_d, hit), kept
	// while the record carries the ray of the current BSDF sample
	float4 *prim;
	// The rays of a closest-hit queue once more, in QUEUE order (entry i = ray_o / ray_d of the path queue[i]): the kernel that
	// fills a queue (k_generate, k_shade) streams them out next to the ids -- rqn_*, the queue being filled -- and
	// k_trace<closest> streams them in -- rq_*, the queue being traced -- instead of gathering one random 128-byte line
	// per ray behind the id (a dependent trip, 64 lines per load instruction).  NULL = not kept / read the records.
	float4 *rq_o, *rq_d, *rqn_o, *rqn_d;
};

// flags in Li.w
enum : uint32_t {
	F_EMITTED = 1u,       // rRec.type & EEmittedRadiance
	F_FIRST   = 2u,       // camera ray not yet processed
	F_ALPHA   = 4u,       // rRec.alpha == 1
	F_D1_SHIFT = 8, F_D2_SHIFT = 16, F_ST_SHIFT = 24
};

struct DConfig {
	float r2c[16], c2w[16];       // rasterToCamera, cameraToWorld (row major)
	float near_clip, far_clip;
	float aperture_radius, focus_depth;   // thin lens (perspective.cpp:90-103); 0 = pinhole
	int32_t camera_kind;                  // 0 perspective, 1 orthographic (orthographic.cpp:104-118)
	int32_t width, height;             // size of the film buffer = the crop window (Film::getCropSize)
	int32_t crop_x, crop_y;            // film pixel (x, y) is raster position (x + crop_x, y + crop_y) (mfilm.cpp:118-143)
	// pixel keys index a pix_w-wide grid over the FULL film's raster space whose origin is (pix_off, pix_off)
	// (pix_off = -border with highQualityEdges): key -> raster pixel, and the sampler streams are keyed by it
	int32_t pix_w, pix_off;
	int32_t max_depth, rr_depth, strict_normals;
	// integrator plugin: 0 = path (MIPathTracer), 1 = direct (MIDirectIntegrator, direct.cpp:51-56)
	int32_t integrator, n_lum, n_bsdf;
	float frac_lum, frac_bsdf, weight_lum, weight_bsdf;
	// luminaireSamples / bsdfSamples > 1: the loops of direct.cpp:129-150,163-195 run as rounds over the paths of the
	// camera hits -- 0 one pass (both counts <= 1), 1 luminaire sample dr_index, 2 BSDF sample dr_index,
	// 3 what follows the ray of a BSDF sample (direct.cpp:172-194)
	int32_t dr_mode, dr_index;
	// Sampler::request2DArray (sampler.cpp:71-74): arr_n arrays of arr_size[a] points per camera sample; element
	// (sample j, k) of array a is entry arr_off[a] + j * arr_size[a] + k of the pixel's arr_total points
	int32_t arr_n;
	uint32_t arr_size[2], arr_off[2], arr_total;
	const uint32_t *arr_scr;      // ldsampler: [slot][arr_n][2] scrambles
	const uint16_t *arr_perm;     // ldsampler: [slot][arr_total] shuffled point indices
	const float2 *arr_pts;        // stratified: [slot][arr_total] latin hypercube points
	int32_t sampler_kind;
	uint32_t spp; int32_t ld_depth;
	int32_t strat_res;            // StratifiedSampler::m_resolution (spp = strat_res^2)
	uint64_t seed;
	int32_t slot_per_path;        // 1: one sampler slot per path (explicit sample lists)
	// TabulatedFilter (rfilter.h:65-102); border = ceil(max(size) - 0.5) (renderproc.cpp:143-144)
	float filt_size_x, filt_size_y;
	int32_t filt_border;
	const float *filt_values;     // [16][16] on the device
	const uint32_t *ld_scr;       // [slot][3*ld_depth]
	const uint16_t *ld_perm;      // [slot][2*ld_depth][spp]
	const uint16_t *primes;       // primeTable (util.cpp:64-122) for the halton / hammersley samplers
};

struct DQueues {
	// per-material queues written by the closest-hit kernel: bin b = bins_base + b * bin_stride, kBinShards segments of
	// bin_seg_cap entries each (one base pointer instead of nine keeps the kernels' scalar registers free)
	uint32_t *bins_base;
	uint32_t bin_stride;
	uint32_t bin_seg_cap;
	// the hit (t, u, v, primitive) of every binned path next to its id, same index: the closest-hit kernel appends both in one
	// stream and the shading reads both in one -- otherwise the hit is a 16-byte store into a random path record (a partial
	// line: read, merged, written back).  Launches without the material sort leave their hits in the path records.
	uint4 *bin_hits;
	__host__ __device__ uint32_t *bin(int b) const { return bins_base + (size_t) b * bin_stride; }
	uint32_t *next;               // paths that continue (input of the next closest-hit launch)
	uint32_t *shadow;             // paths with a pending shadow ray
	// closest-hit launches without the mailbox (trace.hip: TIE): the path ids of rays on which two primitives tied in t -- the one
	// case in which the mailbox decides the result (sahkdtree3.h:130-144, :278-283); they are not binned but traced again by the
	// kernel with the mailbox.  Count in counters[kCntRedo]
	uint32_t *redo;
	uint32_t *counters;           // the counter set of this bounce, [i * kCounterStride]: i = b * kBinShards + shard for the bins, then kCnt*
	// counting builds (u64 x kNumTraceCounts): n_inner, n_leaf, n_idx, n_tri_tested, the lane slots of the three loops and of
	// the batches, then the vector-memory requests the kernel ISSUED: sibling pairs from global memory / from the LDS copy,
	// single nodes (pops) from global memory / from the LDS copy, 16-byte record tails, stack words spilled to HBM, record heads
	unsigned long long *trace_counts;
	// counting builds, optional (mtsgpu_replay_roof): ray number r of the queue writes its requests -- kind << 29 | index,
	// see kReq* -- to rec[r * rec_cap ..] and their number to rec_len[r]
	uint32_t *rec, *rec_len;
	uint32_t rec_cap;
	unsigned long long *dev_stats;     // kStat* (may be NULL)
	uint32_t *spill;              // traversal stack overflow: [level][thread]; one buffer per traversal mode (the two run concurrently)
	uint32_t spill_stride;
	uint32_t desc_min;                 // k_trace leaves its descent loop when fewer lanes than this are on inner nodes (>= 1)
	uint32_t leaf_min;                 // ... and its primitive loop when fewer lanes than this have leaf entries left (>= 1)
	uint32_t refill_min;               // k_trace refills its idle lanes once this many are idle (1..64)
	uint32_t coherent;                 // the rays of this launch are camera rays / their shadow rays: plain 64-ray batches
	uint32_t n_cus;                    // hipDeviceProp_t::multiProcessorCount: the persistent grids are sized from it
	uint32_t force_static;             // deal the whole queue statically (retry after a bin segment overflow; device-driven bounces)
	// experiment knobs of the k_trace schedule (mtsgpu_set_tuning; 0 = the default rule)
	uint32_t tune_batch;               // rays per wave and batch, 1..64
	uint32_t tune_dyn_div;             // 1/x of the rounds of a large launch are claimed dynamically (default 4)
	uint32_t tune_refill;              // refill threshold for coherent launches too (default: 64 there)
	uint32_t tune_plain_below;         // launches of fewer rays run the plain loops (0 = 8 rounds of the largest grid)
	uint32_t tune_dyn_min_rounds;      // launches of at least this many rounds claim their last rounds dynamically (0 = 8)
	uint32_t tune_blocks_per_cu;       // experiment: fewer resident workgroups of kTraceBlock threads per CU than trace_blocks_per_cu(mode)
	                                   // (3 closest-hit / 4 shadow at 512 threads); 0 or a value >= that = all.  Range 0..kTraceBlocksPerCuMax
	// The direct-light term of a shadow ray that came through (path.cpp:124: Li += ...): 0 = the any-hit kernel adds it to the
	// path's radiance itself (a random line read and a partial-line write per ray, with the wave waiting for the read);
	// 1 = it PARKS the term in slot 2 of the path record, tagged kNeeTag (one 16-byte store, nobody waits), and whoever reads
	// the record next adds it first -- the path's next shading, or the film kernels if the path has ended.  Same addition,
	// same place in the path's order of sums.  Slot 2 is free for it: the hits of binned paths travel with the bins.
	uint32_t nee_parked;
};
constexpr uint32_t kNeeTag = 0x4E454521u;      // not a primitive index (< 2^29) and not kNoPrim
// the radiance of a path record with a parked direct-light term added (what every reader of a finished path sees)
__host__ __device__ inline float4 settled_Li(float4 L, float4 parked) {
	union { float f; uint32_t u; } w; w.f = parked.w;
	if (w.u == kNeeTag) { L.x += parked.x; L.y += parked.y; L.z += parked.z; }
	return L;
}

// How one traversal launch over n rays is scheduled.  A pure function of (n, mode, q): the host evaluates it to size
// the grid when it knows n, the kernel evaluates it again -- with n read from device memory when the host does not
// know it (device-driven bounces) -- so both agree on who owns which batch.
struct TracePlan {
	uint32_t batch;        // rays per wave and batch (64; fewer when the launch cannot fill the chip)
	uint32_t blocks;       // workgroups that take part (the rest of a worst-case grid exits at once)
	uint32_t static_n;     // statically dealt queue prefix: whole rounds of the grid, or the whole queue
	uint32_t refill_min, desc_min, leaf_min;
};
__host__ __device__ inline TracePlan trace_plan(uint32_t n, int mode, const DQueues &q) {
	TracePlan p;
	const uint32_t wavesPerBlock = kTraceBlock / 64;
	uint32_t perCu = trace_blocks_per_cu(mode);
	if (q.tune_blocks_per_cu && q.tune_blocks_per_cu < perCu) perCu = q.tune_blocks_per_cu;
	const uint32_t maxBlocks = q.n_cus * perCu;
	// rays per wave: 64, or the smallest power of two (>= 8) with which the launch still fits into one round of the
	// persistent grid -- a launch that cannot fill the lanes of the chip trades idle lanes for shorter waves
	uint32_t batch = 64;
	if (!q.coherent) while (batch > 8u && (unsigned long long) (batch / 2) * wavesPerBlock * maxBlocks >= n) batch /= 2;
	if (q.tune_batch >= 1 && q.tune_batch <= 64) batch = q.tune_batch;
	p.batch = batch;
	const uint32_t need = (uint32_t) (((unsigned long long) n + batch * wavesPerBlock - 1) / (batch * wavesPerBlock));
	p.blocks = need < maxBlocks ? need : maxBlocks;
	// static share of the queue: whole rounds of the grid; the last quarter of the rounds and the remainder are
	// claimed dynamically (one atomic per 64-ray batch, far below the ~88 / us a single counter sustains)
	const uint32_t perRound = p.blocks * wavesPerBlock * batch;
	const uint32_t rounds = perRound ? n / perRound : 0u;
	const uint32_t dynDiv = q.tune_dyn_div ? q.tune_dyn_div : 4u;
	// small launches stay fully static: their waves finish together and would hit the counter in one burst
	const bool dynamic = rounds >= (q.tune_dyn_min_rounds ? q.tune_dyn_min_rounds : 8u) && !q.force_static;
	uint32_t dynRounds = dynamic ? rounds / dynDiv : 0u;
	if (dynamic && dynRounds < 1u) dynRounds = 1u;
	p.static_n = dynamic ? (rounds - dynRounds) * perRound : n;
	// the early loop exits trade the latency of a few straggling rays for throughput; with only a few
	// batches per wave the stragglers are the critical path, so small launches run the plain loops
	p.desc_min = q.desc_min; p.leaf_min = q.leaf_min; p.refill_min = q.refill_min;
	const uint32_t plainBelow = q.tune_plain_below ? q.tune_plain_below : 8u * (q.n_cus * kTraceBlocksPerCuMax) * kTraceBlock;
	if (n < plainBelow || q.coherent)
		p.desc_min = p.leaf_min = 1;
	if (q.coherent && !q.tune_refill)
		p.refill_min = 64;       // neighbouring camera samples finish together: refilling would only mix batches
	if (p.refill_min > batch) p.refill_min = batch;
	return p;
}

// --- launchers (sampler.hip, film.hip, trace.hip, shade.hip, measure.hip) -------------------------------------------------
// state (one word per slot): where the generate() stream of each slot stands after its tables; scratch: ld_table_scratch_entries()
// entries (0 up to 512 samples per pixel) in which the tables are shuffled
size_t ld_table_scratch_entries(uint32_t n_slots, uint32_t spp, int depth);
void launch_ld_tables(hipStream_t s, const DConfig &cfg, const uint32_t *pixel_keys, uint32_t n_slots,
                      uint32_t *scr, uint16_t *perm, unsigned long long *state, uint16_t *scratch);
// the requested sample arrays of the table-based samplers (ldsampler.cpp:152-153, stratified.cpp:136-138), continuing
// that stream; writes cfg.arr_scr / arr_perm / arr_pts
void launch_sample_arrays(hipStream_t s, const DConfig &cfg, uint32_t n_slots, const unsigned long long *state_in);
// the reference's Random (MT19937-64) on the device, one generator: see k_random_values
void launch_random_values(hipStream_t s, void *state, int op, unsigned long long seed, unsigned long long arg, uint32_t clone, uint32_t n,
                          unsigned long long *out);
size_t random_state_bytes();
void launch_sampler_values(hipStream_t s, const DConfig &cfg, uint32_t pixel_key, uint32_t j, uint32_t n, int two_d, float *out);
// BSDF::f (op 0), pdf (1), sample(bRec, pdf, sample) (2) for n query records [n][6] of one parameter block; out [n][8]
void launch_bsdf_eval(hipStream_t s, uint32_t type, const float *params, int op, uint32_t n, const float *queries, float *out);
void launch_generate(hipStream_t s, const DScene &sc, const DPaths &ps, const DConfig &cfg,
                     const uint32_t *pixel_list, uint32_t n_slots, const uint32_t *explicit_samples,
                     uint32_t n_paths, uint32_t *queue);
// mode 0: closest hit over ps.ray_* (writes ps.hit, bins ids by material)
// mode 1: shadow rays over ps.sh_* (adds ps.nee to ps.Li when unoccluded)
// mode 2: any-hit over ps.ray_* (writes ps.hit.w = occluded) -- test/benchmark API
// n_dev == NULL: n rays, grid sized for them.  n_dev != NULL: the count is read from device memory by the kernel and n
// is only its upper bound (device-driven bounces: the host never learns the queue sizes)
// tie: (mode 0 with bin, no counting) the mailbox-free kernel that lists tied rays in q.redo instead of binning them
void launch_trace(hipStream_t s, int mode, bool count, bool bin, const DScene &sc, const DPaths &ps,
                  const DQueues &q, const uint32_t *queue, uint32_t n, bool coherent, const uint32_t *n_dev = nullptr, bool tie = false);
// one launch for the closest-hit queue of a bounce (mode 0 with the material sort) and, behind it in the same waves, the any-hit
// queue of the bounce before (mode 1); host-sized grids only
void launch_trace_pair(hipStream_t s, const DScene &sc, const DPaths &ps, const DQueues &qc, const uint32_t *queue_c, uint32_t n_c, bool coherent_c,
                       const DQueues &qs, const uint32_t *queue_s, uint32_t n_s, bool coherent_s);
// prefix[s] = number of entries of the bin in segments < s (prefix[kBinShards] = total)
struct BinView { uint32_t prefix[kBinShards + 1]; };
// views_dev == NULL: the bin has view.prefix[kBinShards] entries.  Otherwise the view is views_dev[bin] (written by
// k_prep on the device) and n_bound bounds its size
// bin_ids: the bin's id segments (q.bin(bin) unless the caller shades another queue)
void launch_shade(hipStream_t s, int bin, const DScene &sc, const DPaths &ps, const DConfig &cfg,
                  const DQueues &q, const BinView &view, const BinView *views_dev = nullptr, uint32_t n_bound = 0,
                  const uint32_t *bin_ids = nullptr);
// device-driven bounces, path integrator / one-sample direct integrator: all bins of bin_mask in one launch; views_dev as
// above, n_bound bounds the sum of the bin sizes
void launch_shade_all(hipStream_t s, const DScene &sc, const DPaths &ps, const DConfig &cfg, const DQueues &q,
                      const BinView *views_dev, uint32_t bin_mask, uint32_t n_bound);
// device-driven bounces: per-bin views from the shard counters of the closest-hit launch that just ran (`cur`), and
// the counter set of the next bounce zeroed
void launch_prep(hipStream_t s, const uint32_t *cur, uint32_t *next_set, BinView *views_dev, uint32_t bin_seg_cap,
                 unsigned long long *dev_stats);
// path_len (may be NULL): u64 sum of the final path depths (the avgPathLength statistic, path.cpp:212-213)
void launch_accumulate(hipStream_t s, const DPaths &ps, const DConfig &cfg, uint32_t n_slots,
                       uint32_t spp_per_slot, float *film, unsigned long long *path_len);
void launch_path_lengths(hipStream_t s, const DPaths &ps, uint32_t n_paths, unsigned long long *path_len);
// a[i] = b[i] + s * c[i] over n float4
// `blocks` workgroups of 256 stride over the arrays (0: 4096)
void launch_triad(hipStream_t s, float4 *a, const float4 *b, const float4 *c, float scale, size_t n, unsigned blocks = 0);
// random 16-byte gathers, one element per lane and iteration, over (mask_elems + 1) elements
void launch_gather_roof(hipStream_t s, const uint4 *data, uint32_t mask_elems, int iters, unsigned blocks, uint32_t *sink);
// Replay roof of k_trace<closest> (mtsgpu_replay_roof): the recorded requests of n rays, re-issued with no arithmetic.
// order[] lists the rays by decreasing request count; build_replay lays the requests of 64 consecutive rays of that order
// out as tr[(batch * cap + k) * 64 + lane] (kReqNone past a ray's end) with batch_len[batch] = the longest of the 64
void launch_build_replay(hipStream_t s, const uint32_t *rec, const uint32_t *rec_len, const uint32_t *order, uint32_t n, uint32_t cap,
                         uint32_t *tr, uint32_t *batch_len);
// the persistent grid of k_trace<closest> (same workgroups per CU, same LDS footprint) walks the batches
void launch_replay(hipStream_t s, const DScene &sc, const DPaths &ps, const DQueues &q, const uint32_t *tr, const uint32_t *batch_len,
                   uint32_t n_batches, uint32_t cap, uint32_t zero, uint32_t *sink);
void launch_iota_strided(hipStream_t s, uint32_t *p, uint32_t n, uint32_t stride);
void launch_gather_strided(hipStream_t s, float4 *dst, const float4 *src, uint32_t n, uint32_t stride);      // dst[i] = src[i * stride]
// dst[i] += src[i] over n floats (the ordered film sum of a device group)
void launch_add_film(hipStream_t s, float *dst, const float *src, size_t n);
// ImageBlock tiles of one context: rect, first sampler slot of the tile inside its pass, block index
struct TileMeta { int32_t x0, y0, w, h; uint32_t slot_base; uint32_t block_index; uint32_t colour; uint32_t pad; };
void launch_splat_blocks(hipStream_t s, const DPaths &ps, const DConfig &cfg, const TileMeta *tiles, uint32_t n_tiles,
                         uint32_t spp, int block_size, float *blocks);
void launch_add_blocks(hipStream_t s, const DConfig &cfg, const TileMeta *tiles, uint32_t n_tiles, uint32_t colour,
                       int block_size, const float *blocks, float *film);
void launch_fill_u32(hipStream_t s, uint32_t *p, uint32_t v, size_t n);
void launch_iota(hipStream_t s, uint32_t *p, uint32_t n);
size_t trace_spill_levels();
int trace_tail_filter();          // MG_TAIL_FILTER of the build: scene upload computes the per-entry flags only when the kernels use them
uint32_t trace_top_nodes();       // device nodes k_trace copies into LDS: the breadth-first top of the tree
size_t trace_stack_levels();      // depth of the traversal stack (LDS + spill levels)

} // namespace mg
