/*
 * mtsgpu.h -- C ABI of the MI355X path-tracing hot path (libmtsgpu.so).
 *
 * This is the drop-in boundary for Mitsuba 0.2.1's unidirectional path tracer
 * (reference: src/integrators/path/path.cpp:47-216 driven by
 * src/librender/integrator.cpp:131-170).  Everything that crosses it is a plain
 * pointer, a size or a POD struct; no C++/torch types, no exceptions.
 *
 * Two clients bind exactly these entry points:
 *   (1) the Mitsuba integrator plugin `gpupath`, which fills a
 *       mtsgpu_scene from Scene/ShapeKDTree/TriMesh and overrides
 *       Integrator::render() (include/mitsuba/render/integrator.h:57);
 *   (2) the host-side mirror in mitsuba-renderer_amd/ (ctypes), used by tests/ and bench.py.
 *
 * All functions return 0 on success or a negative MTSGPU_E* code;
 * mtsgpu_last_error() gives the text.  One ctx = one GPU = one host thread; a
 * device group (mtsgpu_create_multi, below) drives several ctx -- one per GPU of the node -- from ONE host
 * process and sums their films over xGMI, which is what a Mitsuba process needs
 * (Scene::render calls Integrator::render once, src/librender/scene.cpp:356-359).
 */
#ifndef MTSGPU_H
#define MTSGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTSGPU_ABI_VERSION 6

enum {
	MTSGPU_OK = 0,
	MTSGPU_EINVAL = -1,   /* bad argument / inconsistent scene                  */
	MTSGPU_EHIP = -2,     /* a HIP runtime call failed                          */
	MTSGPU_ENODEV = -3,   /* no usable gfx950 device                            */
	MTSGPU_ECANCEL = -4,  /* *cancel became non-zero (Integrator::cancel)       */
	MTSGPU_ESTATE = -5    /* call sequence error (e.g. render before upload)    */
};

/* BSDF plugins on the path (src/bsdfs/{lambertian,dielectric,roughmetal,microfacet}.cpp) */
enum {
	MTSGPU_BSDF_LAMBERTIAN = 0, /* params: [0..2] reflectance                                  */
	MTSGPU_BSDF_DIELECTRIC = 1, /* params: [0] intIOR [1] extIOR [2..4] specRefl [5..7] specTrans */
	MTSGPU_BSDF_ROUGHMETAL = 2, /* params: [0] alphaB [1..3] ior [4..6] k [7..9] specRefl        */
	MTSGPU_BSDF_MICROFACET = 3, /* params: [0] alphaB [1] kd [2] ks [3] intIOR [4] extIOR
	                                       [5..7] diffuseRefl [8..10] specRefl                  */
	MTSGPU_BSDF_MIRROR = 4,     /* params: [0..2] specularReflectance            (src/bsdfs/mirror.cpp)      */
	MTSGPU_BSDF_PHONG = 5,      /* params: [0] exponent [1] kd [2] ks [3] specularSamplingWeight
	                                       [4] diffuseSamplingWeight [5..7] diffuseRefl [8..10] specRefl
	                                       (values after Phong::configure, src/bsdfs/phong.cpp:74-96)   */
	MTSGPU_BSDF_ROUGHGLASS = 6, /* params: [0] distribution (0 beckmann, 1 phong, 2 ggx) [1] alpha (phong: the exponent
	                                       2/alpha^2 - 2 the constructor derives, roughglass.cpp:130-136) [2] intIOR
	                                       [3] extIOR [4..6] specularReflectance [7..9] specularTransmittance
	                                       (src/bsdfs/roughglass.cpp)                                              */
	MTSGPU_BSDF_DIFFTRANS = 7,  /* params: [0..2] transmittance                          (src/bsdfs/difftrans.cpp)  */
	MTSGPU_BSDF_NTYPES = 8,
	/* OR-ed into bsdf_type: the BSDF is wrapped in a `twosided` adapter (src/bsdfs/twosided.cpp) */
	MTSGPU_BSDF_TWOSIDED = 0x100
};
#define MTSGPU_BSDF_NPARAMS 16

/* Luminaire plugins on the path (src/luminaires/{area,constant}.cpp) */
enum {
	MTSGPU_LUM_AREA = 0,     /* params: [0..2] intensity; shape = emitting TriMesh            */
	MTSGPU_LUM_CONSTANT = 1, /* params: [0..2] intensity [3..5] bsphere centre [6] radius      */
	/* delta luminaires (isIntersectable() == false, path.cpp:118-120) */
	MTSGPU_LUM_POINT = 2,    /* params: [0..2] intensity [3..5] position                (src/luminaires/point.cpp) */
	MTSGPU_LUM_DIRECTIONAL = 3, /* [0..2] intensity [3..5] direction (unit) [6] disk radius = scene bsphere radius
	                               (src/luminaires/directional.cpp:65-91)                                        */
	MTSGPU_LUM_SPOT = 4,     /* [0..2] intensity [3..5] position [6] cos(beamWidth) [7] cos(cutoffAngle)
	                            [8] cutoffAngle (rad) [9] 1/(cutoffAngle-beamWidth) [10..18] world->luminaire
	                            3x3 (row major) [19] beamWidth (rad)           (src/luminaires/spot.cpp:33-118) */
	MTSGPU_LUM_COLLIMATED = 6, /* [0..2] intensity [3] radius [4..15] world->luminaire 3x4 (row major, affine)
	                            [16..27] luminaire->world 3x4              (src/luminaires/collimated.cpp) */
	MTSGPU_LUM_ENVMAP = 5    /* [0] intensityScale [3..5] bsphere centre [6] radius (envmap.cpp:112-126)
	                            [7..15] world->luminaire 3x3 [16..24] luminaire->world 3x3 (row major); the image and
	                            its sampling density are the env_* arrays of the scene   (src/luminaires/envmap.cpp) */
};
#define MTSGPU_LUM_NPARAMS 32

/* Sampler kinds.  *_KEYED are the per-(pixel,sample)-keyed forms of the two
 * reference samplers (src/samplers/{independent,ldsampler}.cpp): identical
 * integer code, with Random (MT19937-64) replaced by a keyed SplitMix64 stream
 * so that samples are independent of traversal order. */
enum {
	MTSGPU_SAMPLER_INDEPENDENT_KEYED = 0,
	MTSGPU_SAMPLER_LD_KEYED = 1,
	/* The two purely deterministic QMC samplers (src/samplers/{halton,hammersley}.cpp): no Random involved, so the
	 * sample values are the reference's own, not a keyed variant: sample j of EVERY pixel is
	 * radicalInverse(primeTable[depth], j) (generate() resets the index for each pixel, halton.cpp:58-61).  The
	 * evaluation order inside Point2(nextValue(), nextValue()) (halton.cpp:88) is left to the compiler by the
	 * reference; x first is used here. */
	MTSGPU_SAMPLER_HALTON = 2,
	MTSGPU_SAMPLER_HAMMERSLEY = 3,
	/* src/samplers/stratified.cpp, keyed like the LD sampler: generate() shuffles the per-depth stratum
	 * permutations with the (seed, pixel, 0) stream, the jitter comes from the per-sample stream; sampleCount is
	 * rounded up to a perfect square (stratified.cpp:36-44) */
	MTSGPU_SAMPLER_STRATIFIED_KEYED = 4
};

/* shape_flags bits */
#define MTSGPU_SHAPE_HAS_NORMALS 1u  /* TriMesh has per-vertex normals (faceNormals=false) */

/* Shape plugins.  A TriMesh is expanded into one kd-tree primitive per triangle; every other shape is
 * ONE primitive that the tree only knows by its AABB (ShapeKDTree::addShape, src/librender/skdtree.cpp:43-60);
 * its tri_idx row is {NONE, NONE, NONE} and its TriAccel row has k = MTSGPU_KNOTRIANGLE (skdtree.cpp:92-96). */
enum {
	MTSGPU_SHAPE_TRIMESH = 0,
	MTSGPU_SHAPE_SPHERE = 1   /* src/shapes/sphere.cpp; params: [0..2] m_center [3] m_radius [4] m_inverted (0/1)
	                             [5..13] m_objectToWorld 3x3 (row major, scale removed: sphere.cpp:49-53)
	                             [14..22] m_worldToObject 3x3 [23] m_invSurfaceArea                           */
};
#define MTSGPU_SHAPE_NPARAMS 24
#define MTSGPU_KNOTRIANGLE 0xFFFFFFFFu

/*
 * Flattened scene.  This is what ShapeKDTree + TriMesh + BSDF/Luminaire
 * parameter blocks look like once laid out for HBM (Appendix A of SURVEY.md).
 *
 * Primitive index space = concatenation of the shapes' triangles in
 * ShapeKDTree::m_shapes order (src/librender/skdtree.cpp:43-65).
 */
typedef struct mtsgpu_scene {
	uint32_t abi_version;        /* MTSGPU_ABI_VERSION                                        */

	/* --- geometry (TriMesh storage, include/mitsuba/render/trimesh.h) --- */
	uint32_t n_shapes, n_tris, n_verts;   /* n_tris = number of kd-tree primitives (triangles + other shapes) */
	const float    *vtx_pos;     /* [n_verts][3]                                              */
	const float    *vtx_nrm;     /* [n_verts][3]; rows of shapes without normals are ignored  */
	const uint32_t *tri_idx;     /* [n_tris][3] indices into the global vertex pool           */
	const uint32_t *shape_tri_offset; /* [n_shapes+1] prefix sums (m_shapeMap)               */
	const int32_t  *shape_bsdf;  /* [n_shapes] index into bsdf_* or -1 (not an occluder)      */
	const int32_t  *shape_lum;   /* [n_shapes] index into lum_* or -1                         */
	const uint32_t *shape_flags; /* [n_shapes] MTSGPU_SHAPE_*                                 */
	const uint32_t *shape_type;  /* [n_shapes] MTSGPU_SHAPE_TRIMESH / _SPHERE; NULL = all TriMesh */
	const float    *shape_params;/* [n_shapes][MTSGPU_SHAPE_NPARAMS]; NULL iff shape_type is NULL  */

	/* --- SAH kd-tree (KDNode, include/mitsuba/render/gkdtree.h:442-470) ---
	 * node = 2 x u32: inner {axis | relOffsetToLeft<<2, float split},
	 *                 leaf  {1<<31 | primStart, primEnd}; children adjacent; root = 0 */
	uint32_t n_nodes, n_indices;
	const uint32_t *kd_nodes;    /* [n_nodes][2]                                              */
	const uint32_t *kd_indices;  /* [n_indices] primitive ids                                 */
	/* TriAccel, 12 x 4 B per primitive (include/mitsuba/render/triaccel.h:34-48) */
	const uint32_t *triaccel;    /* [n_tris][12]                                              */
	float aabb_min[3], aabb_max[3]; /* enlarged kd-tree AABB (gkdtree.h:1170-1176)          */

	/* --- BSDF parameter blocks --- */
	uint32_t n_bsdfs;
	const uint32_t *bsdf_type;   /* [n_bsdfs]                                                 */
	const float    *bsdf_params; /* [n_bsdfs][MTSGPU_BSDF_NPARAMS]                            */

	/* --- luminaires (Scene::m_luminaires order) --- */
	uint32_t n_lums;
	const uint32_t *lum_type;    /* [n_lums]                                                  */
	const float    *lum_params;  /* [n_lums][MTSGPU_LUM_NPARAMS]                              */
	const int32_t  *lum_shape;   /* [n_lums] emitting shape or -1                             */
	const float    *lum_inv_area;/* [n_lums] TriMesh::m_invSurfaceArea (trimesh.cpp:283)      */
	const uint32_t *lum_cdf_offset; /* [n_lums+1] into lum_tri_cdf; area: nTris+1 entries    */
	const float    *lum_tri_cdf; /* per-emitter triangle area CDFs (trimesh.cpp:279-283)      */
	const float    *lum_sel_cdf; /* [n_lums+1] Scene::m_luminairePDF cdf (scene.cpp:320-330)  */
	const float    *lum_sel_pdf; /* [n_lums]                                                  */
	float lum_sel_sum;           /* DiscretePDF::getOriginalSum()                             */
	int32_t background_lum;      /* index of the background luminaire or -1                   */

	/* --- the environment map of the MTSGPU_LUM_ENVMAP luminaire (at most one per scene) ---
	 * level 0 of MIPMap::fromBitmap (src/librender/mipmap.cpp:30-92,161-181: powers of two, ERepeat) and the
	 * DiscretePDF over level min(3, levels-1) that EnvMapLuminaire::configure builds (envmap.cpp:95-110) */
	uint32_t env_width, env_height;
	const float *env_pixels;     /* [env_height][env_width][3] linear RGB                     */
	uint32_t env_pdf_width, env_pdf_height;
	const float *env_pdf;        /* [w*h]   DiscretePDF::m_pdf after build()                  */
	const float *env_cdf;        /* [w*h+1] DiscretePDF::m_cdf                                */
} mtsgpu_scene;

/* PerspectiveCameraImpl state (src/cameras/perspective.cpp:43-112) */
typedef struct mtsgpu_camera {
	float raster_to_camera[16]; /* row-major 4x4, m_rasterToCamera                            */
	float camera_to_world[16];  /* row-major 4x4, m_cameraToWorld                             */
	float near_clip, far_clip;  /* camera.cpp:121-123                                         */
	int32_t width, height;      /* size of the rendered film = the crop window's size (Film::getCropSize,
	                               film.cpp:38-41); raster_to_camera maps ITS pixel coordinates            */
	float aperture_radius;      /* thin lens (perspective.cpp:90-103); 0 = pinhole            */
	float focus_depth;          /* camera.cpp:164                                             */
	int32_t kind;               /* 0 = perspective (src/cameras/perspective.cpp), 1 = orthographic
	                               (src/cameras/orthographic.cpp:104-118: origin = rasterToCamera(sample), d = +z,
	                               mint = 0, maxt = far - near; no lens sample)               */
	/* Film crop window (film.cpp:33-41: cropOffsetX/Y, cropWidth/Height inside a film_width x film_height film).
	 * The camera's raster space is that of the FULL film (perspective.cpp:52-63 scales NDC by m_film->getSize());
	 * the work units cover the crop window only and carry its offset (renderproc.cpp:146-154, imageproc.cpp:28-52:
	 * rect.setOffset(pos + m_offset)), so camera samples are generated at raster positions crop_offset +
	 * [0, width) x [0, height), and the film stores pixel (x, y) of the crop window at [y][x] (mfilm.cpp:118-143).
	 * Sampler keys are derived from the full-film pixel position: a cropped render equals the corresponding rectangle
	 * of the full render bit for bit.  film_width / film_height == 0 means no crop (film size = width x height). */
	int32_t crop_offset_x, crop_offset_y;
	int32_t film_width, film_height;
} mtsgpu_camera;

/* Per-kernel-class counters/timings of the last render (measurement, section 8d) */
typedef struct mtsgpu_stats {
	uint64_t camera_samples;
	uint64_t rays_closest, rays_shadow;
	/* algorithmic work of the traversal kernels (only filled when counting is on) */
	uint64_t n_inner, n_leaf, n_idx, n_tri_tested;
	uint64_t trace_launches;
	double trace_ms;     /* sum of HIP-event durations of all traversal launches        */
	double shade_ms;     /* shade / generate / accumulate kernels                       */
	double total_ms;     /* first launch -> last kernel end (device events)             */
	/* the `avgPathLength` statistic of the path tracer (path.cpp:24, :212-213: avgPathLength.incrementBase() per
	 * Li() call, += rRec.depth at its end): sum of the final depths; the average is path_length_sum / camera_samples */
	uint64_t path_length_sum;
	/* closest-hit launches repeated with static ray dealing because a material-queue segment overflowed */
	uint64_t bin_overflow_retries;
	/* time during which at least one traversal launch was running (the shadow rays of a bounce run next to the
	 * closest-hit launch of the following one when `overlap` is on: trace_ms then counts that time twice) */
	double trace_union_ms;
	/* vector-memory requests the traversal kernels ISSUED (lane level; only filled when counting is on): 16-byte sibling
	 * pairs fetched from global memory / served by the LDS copy of the top of the tree, single 8-byte nodes (two per pop)
	 * from global memory / from the LDS copy, 16-byte tails of leaf records (two per primitive whose plane distance lies
	 * inside the ray's interval), stack words spilled to HBM, 16-byte record heads (n_idx plus the heads fetched again when
	 * an interrupted leaf is resumed).  Ray and hit: 3 per ray. */
	uint64_t req_pair_global, req_pair_lds, req_node_global, req_node_lds, req_tail, req_spill, req_head;
	/* parts of trace_ms: the closest-hit launch of the first bounce of every pass (camera rays), and all any-hit launches;
	 * the remaining closest-hit launches are trace_ms - trace_first_ms - trace_shadow_ms */
	double trace_first_ms, trace_shadow_ms;
	/* closest-hit rays traced a second time, with the mailbox, because two primitives tied in t on them (the mailbox-free
	 * closest-hit kernel of host-driven bounces lists them instead of binning them; sahkdtree3.h:130-144, :278-283) */
	uint64_t rays_redone;
} mtsgpu_stats;

typedef struct mtsgpu_ctx mtsgpu_ctx;

/* --- lifecycle ----------------------------------------------------------- */
int  mtsgpu_create(int device, mtsgpu_ctx **out);
void mtsgpu_destroy(mtsgpu_ctx *ctx);
const char *mtsgpu_last_error(const mtsgpu_ctx *ctx); /* ctx may be NULL: global */
int  mtsgpu_abi_version(void);
/* first 16 hex digits of the sha256 of the library's sources, stamped in at build time (csrc/stamp.cpp): lets a
 * build script tell a stale binary from a current one */
const char *mtsgpu_source_hash(void);
/* sizeof() of the ABI structs as compiled into the library: 0 scene, 1 camera, 2 stats, 3 mesh,
 * 4 scene_desc, 5 kd_params (bindings check their own layout against these) */
size_t mtsgpu_abi_sizeof(int which);

/* Use a caller-owned HIP stream (hipStream_t passed as void*); NULL = own stream */
int  mtsgpu_set_stream(mtsgpu_ctx *ctx, void *hip_stream);

/* --- scene / integrator state (replaces Scene::initialize + plugin configure) */
int  mtsgpu_upload_scene(mtsgpu_ctx *ctx, const mtsgpu_scene *scene);
int  mtsgpu_set_camera(mtsgpu_ctx *ctx, const mtsgpu_camera *cam);
/* MonteCarloIntegrator properties (src/librender/integrator.cpp:272-292) */
int  mtsgpu_set_integrator(mtsgpu_ctx *ctx, int max_depth, int rr_depth, int strict_normals);
/* Switches to the `direct` integrator plugin (MIDirectIntegrator, src/integrators/direct/direct.cpp:33-56):
 * luminaireSamples, bsdfSamples >= 0 with a positive sum.  Counts above one draw from Sampler::next2DArray
 * (direct.cpp:58-63,122-127,156-161), which the independent, ldsampler and stratified samplers provide; with halton /
 * hammersley the render call fails as the reference does (halton.cpp:102-104).  sampleCount x count <= 65536.
 * mtsgpu_set_integrator() switches back to `path`. */
int  mtsgpu_set_direct_integrator(mtsgpu_ctx *ctx, int luminaire_samples, int bsdf_samples);
/* Sampler (src/samplers/{independent,ldsampler}.cpp): kind, sampleCount (LD: rounded up to pow2), LD depth, seed */
int  mtsgpu_set_sampler(mtsgpu_ctx *ctx, int kind, uint32_t spp, int ld_depth, uint64_t seed);
/* ImageBlock sharding (src/librender/imageproc.cpp:43-78): this ctx renders the tiles (tx, ty) of the
 * block_size^2 grid with morton(tx, ty) % n_parts == part (bits of tx and ty interleaved, tx lowest): with 8 parts
 * every 4 x 2 group of tiles holds one tile of each part, so the parts sample the image on a 2-D lattice and their
 * loads stay balanced whatever the picture shows.  The reference's own order (a spiral from the
 * centre) only serves the preview. */
int  mtsgpu_set_tiles(mtsgpu_ctx *ctx, int block_size, int part, int n_parts);
/* Reconstruction filter of the film as a TabulatedFilter (src/librender/rfilter.cpp:40-69,
 * include/mitsuba/render/rfilter.h:65-102): half extents and the 16x16 table Film::getTabulatedFilter()
 * holds.  values == NULL selects the box filter (src/rfilters/box.cpp).  Filters wider than half a
 * pixel make every ImageBlock carry a border of ceil(size - 0.5) pixels (renderproc.cpp:143-144). */
int  mtsgpu_set_rfilter(mtsgpu_ctx *ctx, float size_x, float size_y, const float *values);
/* Film property `highQualityEdges` (src/librender/film.cpp:51): the rendered rectangle grows by the filter
 * border on every side (renderproc.cpp:146-153), so pixels at the film's edge receive their full filter
 * support; the extra samples only land in block borders and are clipped by Film::putImageBlock. */
int  mtsgpu_set_film_edges(mtsgpu_ctx *ctx, int high_quality_edges);
/* Optional: render into a caller-owned device buffer [H][W][5] f32 (spec rgb, alpha, weight) */
int  mtsgpu_set_film_buffer(mtsgpu_ctx *ctx, void *device_ptr);
/* Tuning knobs (0 = default): paths in flight per pass; enable traversal counters */
int  mtsgpu_set_options(mtsgpu_ctx *ctx, uint64_t max_paths, int count_traversal, int time_kernels);

/* Scheduling knobs of the traversal kernel, for experiments and tests (none of them changes a result):
 *   refill_min / desc_min / leaf_min (1..64)  lane thresholds of k_trace
 *   batch (1..64, 0 = rule)                   rays per wave and batch
 *   dyn_div (0 = 4)                           1/dyn_div of a large launch's rounds are claimed dynamically
 *   sync_free (-1 rule, 0 off, 1 on)          bounce loop without host round trips (device-side counts); the rule
 *                                             turns it on for passes of at most 8 Mi paths
 *   chunk (1..1024, default 8)                bounces enqueued between two looks at the queue size (sync_free)
 *   overlap (0/1)                             host-driven loop: shadow rays of bounce b on a second stream, next to
 *                                             the closest-hit launch of bounce b + 1
 *   test_retry (0/1)                          treat every first closest-hit launch as overflowed (exercises the retry) */
int  mtsgpu_set_tuning(mtsgpu_ctx *ctx, const char *key, long value);

/* --- the hot path (replaces SampleIntegrator::render, integrator.cpp:87-120) */
int  mtsgpu_render(mtsgpu_ctx *ctx, volatile const int *cancel);
int  mtsgpu_sync(mtsgpu_ctx *ctx);
/* Film::putImageBlock sums: host copy of [H][W][5] f32 */
int  mtsgpu_read_film(mtsgpu_ctx *ctx, float *rgbaw);
int  mtsgpu_clear_film(mtsgpu_ctx *ctx);
int  mtsgpu_get_stats(mtsgpu_ctx *ctx, mtsgpu_stats *out);

/* --- several GPUs in one process -------------------------------------------------------------------------
 * A group owns one ctx per entry of devs[] (entries may repeat: two ctx on one GPU is how the single-GPU test box
 * exercises this path).  Configuration goes to every member through mtsgpu_group_ctx(g, i) with the ordinary
 * calls, or to all of them at once with the mtsgpu_group_* helpers; mtsgpu_group_render() then
 *   1. gives member i the tiles of part i of n (mtsgpu_set_tiles) and clears its film,
 *   2. renders all members concurrently, one host thread per member (SampleIntegrator::render, integrator.cpp:87-120,
 *      with the GPUs in the role of the scheduler's workers),
 *   3. sums the films into member 0's film (BlockedRenderProcess::processResult -> Film::putImageBlock,
 *      src/librender/renderproc.cpp:123-130, src/films/mfilm.cpp:118-143): ONE ncclReduce(sum, f32, W*H*5, root 0)
 *      over RCCL/xGMI when the devices are distinct and librccl can be loaded, otherwise -- and always when
 *      `ordered_reduce` is set -- peer copies into a staging buffer on device 0 added in member order 1, 2, ...,
 *      which makes the film independent of the collective's internal order for filters wider than a pixel.
 * mtsgpu_read_film(mtsgpu_group_ctx(g, 0)) returns the merged film.  The group call is blocking and must not be
 * entered concurrently; cancel is polled by every member. */
typedef struct mtsgpu_group mtsgpu_group;
int  mtsgpu_create_multi(int ndev, const int *devs, mtsgpu_group **out);
void mtsgpu_group_destroy(mtsgpu_group *g);
int  mtsgpu_group_size(const mtsgpu_group *g);
mtsgpu_ctx *mtsgpu_group_ctx(mtsgpu_group *g, int i);
const char *mtsgpu_group_last_error(const mtsgpu_group *g);
/* the same call on every member (scene upload runs on all devices concurrently) */
int  mtsgpu_group_upload_scene(mtsgpu_group *g, const mtsgpu_scene *scene);
int  mtsgpu_group_set_camera(mtsgpu_group *g, const mtsgpu_camera *cam);
int  mtsgpu_group_set_integrator(mtsgpu_group *g, int max_depth, int rr_depth, int strict_normals);
int  mtsgpu_group_set_sampler(mtsgpu_group *g, int kind, uint32_t spp, int ld_depth, uint64_t seed);
int  mtsgpu_group_set_rfilter(mtsgpu_group *g, float size_x, float size_y, const float *values);
/* block_size as in mtsgpu_set_tiles; ordered_reduce: 0 = RCCL when possible, 1 = always the ordered peer-copy sum,
 * 2 = fail when RCCL cannot be initialised and run the collective also for a single member (self-test of the
 * collective path on a one-GPU machine) */
int  mtsgpu_group_render(mtsgpu_group *g, int block_size, int ordered_reduce, volatile const int *cancel);
/* 0 = ordered peer-copy sum, 1 = RCCL ncclReduce: what the last mtsgpu_group_render used */
int  mtsgpu_group_last_reduce_kind(const mtsgpu_group *g);
/* The number of ranks of the group's RCCL communicator, once it exists and has passed its self-check (created by the first
 * mtsgpu_group_render that wants the collective: ncclCommInitAll gave every member a communicator, each reports the group's
 * size, and a one-float sum of ones arrived at the root as that size); 0 before that, when RCCL is not usable, or after a
 * collective failed and the group went back to the ordered sum. */
int  mtsgpu_group_rccl_ranks(const mtsgpu_group *g);
/* Why the last mtsgpu_group_render summed the films in member order although RCCL was asked for ("" when it did not have
 * to): librccl could not be loaded (the environment variable MTSGPU_RCCL_LIB names the library to load), the
 * communicator could not be created, or ncclReduce / ncclGroupEnd failed.  A failed collective does not lose the frame:
 * its result is received in a staging buffer, the members' films stay as rendered, the group stops using RCCL and adds
 * the films up in member order. */
const char *mtsgpu_group_reduce_note(const mtsgpu_group *g);
/* mtsgpu_set_tuning on every member.  One key belongs to the group itself and exists for tests: "rccl_fail" != 0 makes
 * the next collectives report a failure, which exercises the fall-back to the ordered sum. */
int  mtsgpu_group_set_tuning(mtsgpu_group *g, const char *key, long value);

/* Test hook of the traversal kernels' record-tail filter: 1 when mtsgpu_upload_scene would flag a leaf entry with this TriAccel
 * (12 dwords, include/mitsuba/render/triaccel.h:34-48) in a leaf with this box, i.e. when TriAccel::rayIntersect
 * (triaccel.h:141-158), evaluated in binary32, provably rejects EVERY projected point (o_u + t d_u, o_v + t d_v) that lies beyond
 * the box by more than `margin` on the triangle's u or v axis -- k_trace then does not fetch the record's tail for such a
 * candidate (its margin is 2^-16 of the scene's largest coordinate).  No GPU needed. */
int  mtsgpu_tail_filter_flag(const uint32_t *triaccel12, const float *box_min, const float *box_max, float margin);

/* HBM triad a[i] = b[i] + s * c[i] over three arrays of `bytes` each on `device` (float4 lanes, best of `iters`
 * launches): the practical bandwidth roof next to the 8 TB/s specification.  GB/s in *gbs. */
int  mtsgpu_hbm_triad(int device, size_t bytes, int iters, double *gbs);

/* Vector-memory request roof: lane-level 16-byte loads per second when every lane of every wave gathers its own random
 * element from `footprint_bytes` (a power of two; 4 MiB sits in the L2 like the upper kd-tree), with k_trace's grid
 * shape.  The traversal kernel is bound by this rate, not by DRAM bytes. */
int  mtsgpu_gather_roof(int device, size_t footprint_bytes, double *lane_requests_per_s);
/* Measurement: a traversal kernel against a replay of its own request stream.  A sample of n rays of the frame rendered
 * last is traced once with the counting kernel, which records every vector-memory request of every ray (which sibling
 * pair, node, leaf-record chunk; not recorded: what the kernel streams in and out in queue order -- queue ids, rays,
 * the ids and hits appended to the material bins -- stack words spilled to HBM, and the two extra chunks of a sphere
 * primitive); then (a) the product kernel of
 * that class and (b) a kernel that re-issues exactly those requests are timed: per ray in the recorded order, one 16-byte
 * load per request from the line the traversal asked for, eight independent requests in flight per lane, the grid shape
 * and LDS footprint of the closest-hit kernel, NO arithmetic, no stack, no mailbox, no dependence between the requests.
 * (b) is ONE throughput test of the memory system on this set of lines, not a bound: other issue orders may be faster.
 * kind selects the sample and the product kernel:
 *   0  every stride-th of the first n * stride path records -- after mtsgpu_render() each holds the LAST ray of its
 *      path; the rays are copied into queue order first -- with the closest-hit kernel as the bounces launch it
 *      (material binning on);
 *   1  the camera rays of the pass rendered last, generated again (the path records are overwritten), every stride-th,
 *      with the closest-hit kernel in the plain 64-ray batches of a first bounce;
 *   2  every stride-th slot of the shadow queue as the frame left it (slot i holds the ray of the deepest bounce that
 *      queued more than i shadow rays; host-driven passes only), moved to the front of the queue, with the any-hit
 *      kernel (which adds the rays' pending terms to the dead path records).
 * out[12]: rays, recorded requests, rays whose list was truncated (256 requests), product ms, replay ms (best of reps),
 * then the issued-request counters of the sample: pairs global / LDS, nodes global / LDS, record heads, tails, spills.
 * Overwrites the hits of the sampled path records; call after the film has been read. */
int  mtsgpu_replay_roof(mtsgpu_ctx *ctx, int kind, uint32_t n, uint32_t stride, int reps, double *out);

/* --- standalone kernels exposed for parity tests and the traversal benchmark */
/* ShapeKDTree::rayIntersect(ray, its) / (ray) on n host rays.
 * rays: [n][8] f32 = o.xyz, mint, d.xyz, maxt.   hits: [n][4] u32 = t(f32 bits), u, v, prim
 * (prim = 0xFFFFFFFF on miss); shadow != 0 -> hits[i][3] = 1/0 occluded (src/librender/skdtree.cpp:108-199) */
int  mtsgpu_trace_rays(mtsgpu_ctx *ctx, const float *rays, uint32_t n, int shadow, uint32_t *hits);
/* LowDiscrepancySampler::generate() for one pixel key (keyed stream), tables as f32:
 * out1d [depth][spp], out2d [depth][spp][2] */
int  mtsgpu_ld_tables(mtsgpu_ctx *ctx, uint32_t pixel_key, float *out1d, float *out2d);
/* What the configured Sampler hands out on the device: generate() for the pixel with key `pixel_key`, then, for camera
 * sample `sample_index`, n calls of next1D() (two_d == 0: out[n]) or next2D() (out[2n]) (src/librender/sampler.cpp,
 * the plugins under src/samplers).  With the halton / hammersley samplers these are the reference's own numbers: the tables of
 * src/tests/test_samplers.cpp:33-78 are checked against this call.  n <= 4096. */
int  mtsgpu_sampler_values(mtsgpu_ctx *ctx, uint32_t pixel_key, uint32_t sample_index, uint32_t n, int two_d, float *out);
/* The reference's `Random` (MT19937-64, src/libcore/random.cpp:99-227, random.h:82-148) run on the device, one generator:
 * seed == 0 is a default-constructed Random (seed 5489), otherwise Random::seed(seed); clone > 0 takes the clone-th
 * Random(Random *) copy of it (random.cpp:105-110), the way per-worker samplers are made.  op 0: n x nextULong();
 * 1: n x nextFloat() as bit patterns; 2: n x nextSize(arg); 3: Random::shuffle of 0 .. n-1.  A test hook: the library's
 * samplers draw from keyed streams; this pins the generator itself to the reference's known answers. */
int  mtsgpu_random_values(mtsgpu_ctx *ctx, int op, uint64_t seed, uint64_t arg, uint32_t clone, uint32_t n, uint64_t *out);
/* The BSDF plugins as the device runs them, for n query records of ONE parameter block (bsdf_type may carry
 * MTSGPU_BSDF_TWOSIDED; params[MTSGPU_BSDF_NPARAMS] as in the enum above).  queries [n][6], out [n][8]:
 *   op 0  BSDF::f(bRec)               queries = wi.xyz, wo.xyz      out = f.rgb
 *   op 1  BSDF::pdf(bRec)             queries = wi.xyz, wo.xyz      out = pdf
 *   op 2  BSDF::sample(bRec, pdf, s)  queries = wi.xyz, s.x, s.y, - out = wo.xyz, pdf, f.rgb, sampledType (bit pattern)
 *         (src/librender/bsdf.cpp:37-48 for the plugins that do not override it; f = 0 and pdf = 0 for a failed sample)
 * wi / wo are in the local shading frame (include/mitsuba/render/bsdf.h:34-132).  A test hook: the chi-square
 * procedure of the reference (src/tests/test_chisquare.cpp:299-420 with the BSDFs of data/tests/test_bsdf.xml) runs
 * against these values, i.e. against the code k_shade executes, with no CPU restatement in between. */
int  mtsgpu_bsdf_eval(mtsgpu_ctx *ctx, uint32_t bsdf_type, const float *params, int op, uint32_t n, const float *queries, float *out);
/* MIPathTracer::Li for explicit camera samples: in [n][3] u32 = pixel x, y, sample index;
 * out [n][8] f32 = Li rgb, alpha, raster x, raster y, depth, unused */
int  mtsgpu_li_samples(mtsgpu_ctx *ctx, const uint32_t *pix_samples, uint32_t n, float *out);

/* --- host-side flattening (what Scene::initialize does on the CPU) ---------
 * Builds everything a mtsgpu_scene needs from plain meshes: vertex normals
 * (trimesh.cpp:473-545), area CDFs, TriAccel table, SAH kd-tree (gkdtree.h). */
typedef struct mtsgpu_mesh {
	uint32_t n_verts, n_tris;
	const float    *positions;  /* [n_verts][3]                                   */
	const float    *normals;    /* [n_verts][3] or NULL                           */
	const uint32_t *triangles;  /* [n_tris][3]                                    */
	int32_t face_normals;       /* TriMesh 'faceNormals' property                 */
	int32_t bsdf;               /* index or -1                                    */
	int32_t lum;                /* index of the area luminaire attached, or -1    */
	int32_t shape_type;         /* MTSGPU_SHAPE_TRIMESH, or MTSGPU_SHAPE_SPHERE (then n_verts = n_tris = 0) */
	float   sphere_center[3];   /* `center` / `radius` properties (sphere.cpp:44-47)                       */
	float   sphere_radius;
	int32_t sphere_inverted;    /* `inverted` property (sphere.cpp:56)                                      */
} mtsgpu_mesh;

typedef struct mtsgpu_scene_desc {
	uint32_t n_meshes;
	const mtsgpu_mesh *meshes;
	uint32_t n_bsdfs;
	const uint32_t *bsdf_type;
	const float    *bsdf_params;
	uint32_t n_lums;
	const uint32_t *lum_type;    /* area luminaires must be referenced by exactly one mesh */
	const float    *lum_params;  /* constant/directional/envmap: bsphere-derived entries are computed; spot: [6],[7],[9] are computed */
	float camera_pos[3];         /* for ConstantLuminaire::preprocess (constant.cpp:49-63) */
	int32_t has_camera;
	/* bitmap of the envmap luminaire, any size: [env_height][env_width][3] linear RGB (what Bitmap::getFloatData
	 * holds after the EXR is read, mipmap.cpp:161-181); its lum_params carry [0] intensityScale and
	 * [16..24] luminaire->world rotation, everything else is derived */
	uint32_t env_width, env_height;
	const float *env_bitmap;
} mtsgpu_scene_desc;

/* kd-tree build parameters (gkdtree.h:711-724 defaults when 0 / negative) */
typedef struct mtsgpu_kd_params {
	float traversal_cost, query_cost, empty_space_bonus;
	int32_t stop_prims, max_bad_refines, exact_prim_threshold, max_depth, min_max_bins;
	int32_t clip, retract, n_threads;
	/* bit set -- 1: the min-max binning phase (> exact_prim_threshold primitives per node, gkdtree.h:1735-1867) runs on
	 * the current HIP device; 2: the exact O(n log n) sweep below that threshold (gkdtree.h:1898-2345) runs there too.
	 * The tree is the same bit for bit either way.  An error if there is no device; 0 = host */
	int32_t gpu_binning;
} mtsgpu_kd_params;

typedef struct mtsgpu_flat_scene mtsgpu_flat_scene; /* owns the arrays of a mtsgpu_scene */
int  mtsgpu_flatten(const mtsgpu_scene_desc *desc, const mtsgpu_kd_params *kd, mtsgpu_flat_scene **out);
const mtsgpu_scene *mtsgpu_flat_scene_get(const mtsgpu_flat_scene *fs);
void mtsgpu_flat_scene_free(mtsgpu_flat_scene *fs);
/* kd-tree statistics logged by the reference builder (gkdtree.h:1178-1213) */
int  mtsgpu_flat_scene_kdstats(const mtsgpu_flat_scene *fs, double *out6 /* inner, leaf, idx, expTrav, expLeaves, expPrims */);

/* One shape of a Mitsuba `.serialized` mesh file, as <shape type="serialized"> with `shapeIndex` loads it
 * (TriMesh::TriMesh(Stream *, int index), src/librender/trimesh.cpp:156-236): header 0x041C / version 3, zlib
 * stream, single or double precision.  *mesh points into *out and stays valid until mtsgpu_loaded_mesh_free;
 * `bsdf` and `lum` are -1, `normals` is NULL when the file has none, `face_normals` follows the file's flag. */
typedef struct mtsgpu_loaded_mesh mtsgpu_loaded_mesh;
int  mtsgpu_load_serialized(const char *path, int shape_index, mtsgpu_loaded_mesh **out, mtsgpu_mesh *mesh);
void mtsgpu_loaded_mesh_free(mtsgpu_loaded_mesh *m);

/* TabulatedFilter (src/librender/rfilter.cpp:40-69) of the reconstruction filter plugins: kind 0 `box`,
 * 1 `gaussian` (p0 = stddev; src/rfilters/gaussian.cpp:30-42,62-65), 2 `mitchell` (p0 = B, p1 = C; mitchell.cpp),
 * 3 `catmullrom`, 4 `wsinc` (p0 = cycles; wsinc.cpp).  half_size / p0 / p1 <= 0 (mitchell: < 0) select the
 * plugin's defaults.  size_xy[2], values[256] */
int  mtsgpu_tabulate_filter(int kind, float half_size, float p0, float p1, float *size_xy, float *values);

/* PerspectiveCameraImpl::configure for a lookAt camera (perspective.cpp:43-71,
 * transform.cpp:100-124,174-190).  fov in degrees along the smaller image side. */
int  mtsgpu_make_camera(const float origin[3], const float target[3], const float up[3],
                        float fov_deg, int width, int height, mtsgpu_camera *out);
/* the same camera looking through a crop window of a film_width x film_height film (perspective.cpp:52-63,
 * film.cpp:33-41): out->width/height = crop size, out->crop_offset_* = crop offset; fov along the smaller side of
 * the FULL film */
int  mtsgpu_make_camera_crop(const float origin[3], const float target[3], const float up[3], float fov_deg,
                             int film_width, int film_height, int crop_x, int crop_y, int crop_width, int crop_height,
                             mtsgpu_camera *out);
/* OrthographicCamera::configure (orthographic.cpp:46-82, transform.cpp:155-158) with toWorld =
 * lookAt(origin, target, up) * scale(scale_x, scale_y, 1): the view volume is 2*scale wide along its smaller side. */
int  mtsgpu_make_camera_ortho(const float origin[3], const float target[3], const float up[3],
                              float scale_x, float scale_y, int width, int height, mtsgpu_camera *out);

#ifdef __cplusplus
}
#endif
#endif /* MTSGPU_H */
