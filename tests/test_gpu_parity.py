"""GPU parity: the HIP path through the C ABI vs the CPU oracle, bit for bit."""
import os

import numpy as np
import pytest

from conftest import chord_rays

pytestmark = pytest.mark.gpu

SCENES = {
    "c1": lambda s: s.cornell_c1(),
    "c3_small": lambda s: s.cornell_c3(grid=24, sphere_subdiv=2),
    "c5_small": lambda s: s.cornell_c5(sphere_subdiv=2),
    "next_rows": lambda s: s.next_rows(sphere_subdiv=2),
    "spheres": lambda s: s.spheres(),
    "envlit": lambda s: s.envlit(),
    # the bunny of the reference's kd-tree test, written to a `.serialized` container (double precision, two shapes)
    # by the independent Python writer and read back by the product's loader
    "bunny": lambda s: s.bunny(_bunny_serialized(), _loader()),
}


def _bunny_serialized():
    import tempfile
    import ply_io
    import serialized_io as sio
    path = os.path.join(tempfile.gettempdir(), "mtsgpu_test_bunny.serialized")
    pos, tri = ply_io.read(os.path.join(os.path.dirname(__file__), "golden", "bunny.ply"))
    plane = dict(positions=np.array([[-1, 0.033, -1], [1, 0.033, -1], [1, 0.033, 1], [-1, 0.033, 1]], dtype=np.float64),
                 triangles=np.array([[0, 2, 1], [0, 3, 2]]), face_normals=True)
    sio.write(path, [dict(positions=pos.astype(np.float64), triangles=tri), plane], double=True)
    return path


def _loader():
    import _pkgload
    return _pkgload.load().load_serialized


def _setup(mts, orc, name, W=64, H=64, sampler="independent", spp=8, max_depth=None, seed=0x5EED):
    sd = SCENES[name](mts.scenes)
    scene = mts.Scene(sd)
    oscene = orc.FlatScene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, W, H)
    ocam = orc.make_camera(sd, W, H)
    md = sd.max_depth if max_depth is None else max_depth
    it = mts.MIPathTracer(maxDepth=md)
    it.preprocess(scene, cam, sampler=sampler, sampleCount=spp, seed=seed)
    kind = {"independent": mts.abi.SAMPLER_INDEPENDENT_KEYED, "ldsampler": mts.abi.SAMPLER_LD_KEYED,
            "halton": mts.abi.SAMPLER_HALTON, "hammersley": mts.abi.SAMPLER_HAMMERSLEY,
            "stratified": mts.abi.SAMPLER_STRATIFIED_KEYED}[sampler]
    op = orc.render_params(md, sampler=kind, spp=spp, seed=seed)
    return sd, scene, oscene, cam, ocam, it, op


@pytest.mark.parametrize("name", list(SCENES))
def test_trace_closest_and_shadow(gpu_lib, mts, orc, name):
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, name)
    rays = chord_rays(20000, (0, 1, 0), 2.2, seed=7)
    got = it.trace_rays(rays)
    exp = orc.trace_rays(oscene.scene, rays)
    assert np.array_equal(got, exp), "closest-hit (t,u,v,prim) differs in %d rays" % int((got != exp).any(axis=1).sum())
    assert (exp[:, 3] != 0xFFFFFFFF).sum() > 1000
    # shadow segments between surface points
    seg = rays.copy()
    seg[:, 3] = 1e-3
    seg[:, 7] = 1 - 1e-3
    seg[:, 4:7] *= np.float32(3.0)
    got = it.trace_rays(seg, shadow=True)
    exp = orc.trace_rays(oscene.scene, seg, shadow=True)
    assert np.array_equal(got[:, 3], exp[:, 3])
    assert 0 < exp[:, 3].sum() < len(seg)


def test_ld_tables_bit_exact(gpu_lib, mts, orc):
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c1", sampler="ldsampler", spp=64)
    for key in (0, 1, 4095, 123456):
        t1, t2 = it.ld_tables(key, 64, 3)
        e1 = np.zeros((3, 64), dtype=np.float32)
        e2 = np.zeros((3, 64, 2), dtype=np.float32)
        orc.lib().orc_ld_generate_keyed(0x5EED, key, 64, 3, mts.abi.ptr(e1, mts.abi.f32p), mts.abi.ptr(e2, mts.abi.f32p))
        assert np.array_equal(t1.view(np.uint32), e1.view(np.uint32))
        assert np.array_equal(t2.view(np.uint32), e2.view(np.uint32))


@pytest.mark.parametrize("name,sampler", [("c1", "independent"), ("c1", "ldsampler"), ("c3_small", "ldsampler"),
                                          ("c5_small", "independent"), ("c5_small", "ldsampler"),
                                          ("next_rows", "independent"), ("next_rows", "ldsampler"),
                                          ("spheres", "independent"), ("spheres", "ldsampler"),
                                          ("envlit", "independent"), ("envlit", "ldsampler"),
                                          ("c5_small", "halton"), ("c5_small", "hammersley"), ("spheres", "halton"),
                                          ("bunny", "ldsampler"), ("c5_small", "stratified"), ("next_rows", "stratified")])
def test_li_samples_bit_exact(gpu_lib, mts, orc, name, sampler):
    """MIPathTracer::Li per camera sample: radiance, alpha, raster position and path depth"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, name, W=32, H=32, sampler=sampler, spp=16)
    rng = np.random.RandomState(3)
    ps = np.stack([rng.randint(0, 32, 4000), rng.randint(0, 32, 4000), rng.randint(0, 16, 4000)], axis=1).astype(np.uint32)
    got = it.li_samples(ps)
    exp = orc.li_samples(oscene.scene, ocam, op, ps)
    bad = (got.view(np.uint32) != exp.view(np.uint32)).any(axis=1)
    assert not bad.any(), "%d of %d samples differ; first: got %s exp %s" % (bad.sum(), len(ps), got[bad][:1], exp[bad][:1])
    assert exp[:, :3].max() > 0


@pytest.mark.parametrize("name,sampler,spp", [("c1", "independent", 16), ("c1", "ldsampler", 32), ("c3_small", "ldsampler", 16),
                                              ("c5_small", "ldsampler", 16), ("next_rows", "ldsampler", 16),
                                              ("spheres", "ldsampler", 16), ("envlit", "ldsampler", 16),
                                              ("c5_small", "halton", 24), ("c5_small", "hammersley", 24), ("bunny", "ldsampler", 8), ("c5_small", "stratified", 16)])
def test_film_matches_oracle(gpu_lib, mts, orc, name, sampler, spp):
    """whole renderBlock + putSample pipeline; tolerance stated by north_star: pixel RMSE < 1e-5
    (the target is bit-identical, which is what is asserted first and reported)"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, name, W=48, H=40, sampler=sampler, spp=spp)
    assert it.render()
    film = it.film()
    ofilm, ost = orc.render(oscene.scene, ocam, op)
    st = it.stats()
    assert st["camera_samples"] == 48 * 40 * spp
    assert st["rays_closest"] == ost.rays_closest
    img, oimg = mts.develop(film), orc.develop(ofilm)
    rmse = float(np.sqrt(np.mean((img.astype(np.float64) - oimg.astype(np.float64)) ** 2)))
    assert rmse < 1e-5, "pixel RMSE %g" % rmse
    assert np.array_equal(film.view(np.uint32), ofilm.view(np.uint32)), "film not bit-identical (RMSE %g)" % rmse


@pytest.mark.parametrize("name,nl,nb,sampler", [
    ("c5_small", 1, 1, "ldsampler"), ("spheres", 1, 1, "ldsampler"), ("envlit", 1, 1, "ldsampler"), ("next_rows", 1, 0, "ldsampler"),
    ("c5_small", 0, 1, "ldsampler"),
    # more than one sample per strategy: Sampler::next2DArray (sampler.cpp:76-87) of the three samplers that have it
    ("c5_small", 4, 1, "ldsampler"), ("c5_small", 1, 4, "ldsampler"), ("c5_small", 3, 2, "ldsampler"), ("c5_small", 4, 0, "independent"),
    ("c5_small", 0, 3, "stratified"), ("spheres", 2, 2, "independent"), ("envlit", 3, 3, "stratified"), ("next_rows", 5, 2, "ldsampler"),
    ("c3_small", 2, 5, "independent"), ("c1", 8, 8, "stratified")])
def test_direct_integrator_matches_oracle(gpu_lib, mts, orc, name, nl, nb, sampler):
    """the `direct` integrator plugin (src/integrators/direct/direct.cpp): per-sample Li and the film"""
    sd = SCENES[name](mts.scenes)
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, 40, 32); ocam = orc.make_camera(sd, 40, 32)
    it = mts.MIDirectIntegrator(luminaireSamples=nl, bsdfSamples=nb)
    it.preprocess(scene, cam, sampler=sampler, sampleCount=16, seed=11)
    kind = {"independent": mts.abi.SAMPLER_INDEPENDENT_KEYED, "ldsampler": mts.abi.SAMPLER_LD_KEYED, "stratified": mts.abi.SAMPLER_STRATIFIED_KEYED}[sampler]
    op = orc.render_params(-1, sampler=kind, spp=16, seed=11, integrator="direct",
                           luminaire_samples=nl, bsdf_samples=nb)
    rng = np.random.RandomState(3)
    ps = np.stack([rng.randint(0, 40, 3000), rng.randint(0, 32, 3000), rng.randint(0, 16, 3000)], axis=1).astype(np.uint32)
    got = it.li_samples(ps)
    exp = orc.li_samples(oscene.scene, ocam, op, ps)
    bad = (got.view(np.uint32) != exp.view(np.uint32)).any(axis=1)
    assert not bad.any(), "%d of %d samples differ; first: got %s exp %s" % (bad.sum(), len(ps), got[bad][:1], exp[bad][:1])
    assert exp[:, :3].max() > 0
    assert it.render()
    ofilm, ost = orc.render(oscene.scene, ocam, op)
    assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32))
    st = it.stats()
    # Scene::sampleLuminaire tests visibility before the BSDF is evaluated; the wavefront only queues a shadow ray
    # when the term it guards is non-zero: fewer shadow rays, same sums
    assert st["rays_closest"] == ost.rays_closest and st["rays_shadow"] <= ost.rays_shadow and (st["rays_shadow"] > 0) == (nl > 0)
    # the QMC samplers have no sample arrays (halton.cpp:102-104): refused like the reference refuses it
    if nl > 1:
        it.preprocess(scene, cam, sampler="halton", sampleCount=16, seed=11)
        with pytest.raises(mts.MtsGpuError) as e:
            it.render()
        assert "not supported by QMC samplers" in str(e.value)


def test_direct_integrator_sample_arrays_with_tiles_and_filter(gpu_lib, mts, orc):
    """sample arrays through the bordered-ImageBlock path (gaussian filter, highQualityEdges, several passes, tile
    shards): the per-pixel tables follow the pixel keys of the rendered rectangle"""
    sd = mts.scenes.cornell_c5(sphere_subdiv=2)
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, 70, 50); ocam = orc.make_camera(sd, 70, 50)
    it = mts.MIDirectIntegrator(luminaireSamples=3, bsdfSamples=2)
    it.set_rfilter("gaussian"); of = orc.tabulate_filter("gaussian")
    for sampler, kind in (("ldsampler", mts.abi.SAMPLER_LD_KEYED), ("stratified", mts.abi.SAMPLER_STRATIFIED_KEYED)):
        it.preprocess(scene, cam, sampler=sampler, sampleCount=4, seed=2)
        op = orc.render_params(-1, sampler=kind, spp=4, seed=2, integrator="direct", luminaire_samples=3, bsdf_samples=2)
        it.set_film_edges(True); it.clear_film(); assert it.render()
        ofilm, _ = orc.render_tiles(oscene.scene, ocam, op, of, hq_edges=True)
        assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32)), sampler
        it.set_options(max_paths=32 * 32 * 4); it.clear_film(); assert it.render()       # one tile per pass
        assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32)), sampler
        it.set_options(max_paths=0); it.set_film_edges(False)
        acc = np.zeros((50, 70, 5), dtype=np.float32)
        for part in range(2):
            it.clear_film(); it.set_tiles(32, part, 2); assert it.render()
            acc += it.film()
        it.set_tiles(32, 0, 1)
        o2, _ = orc.render_tiles(oscene.scene, ocam, op, of)
        assert np.allclose(acc, o2, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_fuzz_scenes(gpu_lib, mts, orc, seed):
    """random scenes (soups with degenerate / duplicated triangles, spheres, every BSDF, mixed luminaires): both host
    builders agree on the tree, the traversal answers and per-sample radiance agree bit for bit"""
    sd = mts.scenes.fuzz(seed)
    kp = mts.abi.KdParams()
    if seed % 3 == 0: kp.exact_prim_threshold = 64                      # min-max binning phase too (on the device for seed % 6 == 0)
    scene = mts.Scene(sd, kd_params=kp, gpu_binning=(seed % 6 == 0), gpu_exact=(seed % 4 == 1)); oscene = orc.FlatScene(sd, kd_params=kp)
    a, b = scene.arrays(), oscene.arrays()
    for k in ("kd_nodes", "kd_indices", "triaccel", "vtx_nrm", "lum_tri_cdf"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    W, H = 48, 36
    cam = mts.PerspectiveCamera.for_description(sd, W, H); ocam = orc.make_camera(sd, W, H)
    sampler = ["independent", "ldsampler", "stratified", "halton"][seed % 4]
    kind = {"independent": 0, "ldsampler": 1, "halton": 2, "stratified": 4}[sampler]
    it = mts.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth, strictNormals=bool(seed & 1))
    it.preprocess(scene, cam, sampler=sampler, sampleCount=16, seed=seed)
    op = orc.render_params(sd.max_depth, rr_depth=sd.rr_depth, strict_normals=int(seed & 1), sampler=kind, spp=16, seed=seed)
    rng = np.random.RandomState(seed)
    n = 6000
    ps = np.stack([rng.randint(0, W, n), rng.randint(0, H, n), rng.randint(0, 16, n)], axis=1).astype(np.uint32)
    got, exp = it.li_samples(ps), orc.li_samples(oscene.scene, ocam, op, ps)
    bad = (got.view(np.uint32) != exp.view(np.uint32)).any(axis=1)
    assert not bad.any(), "%d of %d samples differ; first: %s got %s exp %s" % (bad.sum(), n, ps[bad][:1], got[bad][:1], exp[bad][:1])
    assert np.isfinite(exp[:, :3]).all() and exp[:, :3].max() > 0
    # rays through the whole box: closest hits and occlusion
    o = rng.uniform(-0.95, 0.95, (20000, 3)).astype(np.float32); o[:, 1] = rng.uniform(0.05, 1.95, 20000)
    d = rng.normal(size=(20000, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, np.full((20000, 1), 1e-4, np.float32), d, np.full((20000, 1), np.inf, np.float32)], axis=1).astype(np.float32)
    for shadow in (False, True):
        g = it.trace_rays(rays, shadow=shadow); e = orc.trace_rays(oscene.scene, rays, shadow=shadow)
        assert np.array_equal(np.asarray(g).view(np.uint32), np.asarray(e).view(np.uint32)), shadow


def test_tile_sharding_is_exact(gpu_lib, mts, orc):
    """ImageBlock sharding: the union of the parts equals the unsharded film bit for bit"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c1", W=80, H=72, sampler="ldsampler", spp=8)
    assert it.render()
    full = it.film()
    acc = np.zeros_like(full)
    for part in range(3):
        it.clear_film()
        it.set_tiles(32, part, 3)
        assert it.render()
        acc += it.film()
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))


def test_traversal_counters_match_oracle(gpu_lib, mts, orc):
    """the counting build of k_trace (bench.py's algorithmic bytes) counts what the oracle's counting mode counts"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c3_small")
    rays = chord_rays(5000, (0, 1, 0), 2.2, seed=21)
    it.set_options(count_traversal=True)
    it.trace_rays(rays)
    st = it.stats()
    _, tc = orc.trace_rays(oscene.scene, rays, counts=True)
    assert (st["n_inner"], st["n_leaf"], st["n_idx"], st["n_tri_tested"]) == (tc.n_inner, tc.n_leaf, tc.n_idx, tc.n_tri_tested)
    it.set_options(count_traversal=False)


def test_cancel_and_error_paths(gpu_lib, mts, orc):
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c1")
    import ctypes as C
    flag = C.c_int(1)                                   # Integrator::cancel() before the first stage
    assert mts.lib().mtsgpu_render(it._ctx, C.byref(flag)) == -4
    with pytest.raises(mts.MtsGpuError):
        mts.MIPathTracer(maxDepth=4, rrDepth=0).configure()   # "rrDepth == 0 breaks the computation of alpha values!"
    fresh = mts.MIPathTracer(maxDepth=4)
    with pytest.raises(mts.MtsGpuError):
        fresh.render()                                  # no scene uploaded


@pytest.mark.parametrize("case", ["c3_1M", "c5_threshold_300", "spheres_threshold_40", "bunny_threshold_2000", "bins_32"])
def test_gpu_binning_builds_the_same_tree(gpu_lib, mts, orc, case):
    """the device binning phase (k_kd_bin / k_kd_count / k_kd_scatter + the host's minimizeCost) yields the tree of the
    host builder bit for bit -- nodes, index lists, statistics, bounding box -- and therefore the oracle's tree"""
    kp = lambda: mts.abi.KdParams()
    a, b = kp(), kp()
    if case == "c3_1M":
        sd = mts.scenes.cornell_c3()
    elif case == "c5_threshold_300":
        sd = mts.scenes.cornell_c5(sphere_subdiv=3); a.exact_prim_threshold = b.exact_prim_threshold = 300
    elif case == "spheres_threshold_40":
        sd = mts.scenes.spheres(); a.exact_prim_threshold = b.exact_prim_threshold = 40
    elif case == "bunny_threshold_2000":
        sd = mts.scenes.bunny(_bunny_serialized(), _loader()); a.exact_prim_threshold = b.exact_prim_threshold = 2000
    else:
        sd = mts.scenes.cornell_c3(grid=60, sphere_subdiv=3); a.exact_prim_threshold = b.exact_prim_threshold = 1000
        a.min_max_bins = b.min_max_bins = 32
    host = mts.Scene(sd, kd_params=a)
    dev = mts.Scene(sd, kd_params=b, gpu_binning=True)
    ha, da = host.arrays(), dev.arrays()
    for k in ("kd_nodes", "kd_indices", "aabb_min", "aabb_max", "triaccel"):
        assert np.array_equal(ha[k].view(np.uint32), da[k].view(np.uint32)), k
    assert host.kdstats() == dev.kdstats()
    if case != "c3_1M":                                                 # the oracle's own builder, same parameters
        oa = orc.FlatScene(sd, kd_params=a).arrays()
        assert np.array_equal(oa["kd_nodes"], da["kd_nodes"]) and np.array_equal(oa["kd_indices"], da["kd_indices"])


def test_api_state_transitions(gpu_lib, mts, orc):
    """one context reused across scenes, film sizes, samplers, integrators and filters gives the same films as fresh
    contexts (no stale device state); C-ABI misuse returns error codes instead of crashing"""
    import ctypes as C
    L = mts.lib()
    sdA, sdB = mts.scenes.cornell_c1(), mts.scenes.cornell_c5(sphere_subdiv=2)
    scA, scB = mts.Scene(sdA), mts.Scene(sdB)
    oA, oB = orc.FlatScene(sdA), orc.FlatScene(sdB)
    it = mts.MIPathTracer(maxDepth=5)
    plan = [(sdA, scA, oA, 40, 24, "ldsampler", 8), (sdB, scB, oB, 17, 33, "independent", 5), (sdA, scA, oA, 64, 64, "stratified", 9),
            (sdB, scB, oB, 40, 24, "halton", 6)]
    for sd, sc, osc, w, h, sampler, spp in plan:
        cam = mts.PerspectiveCamera.for_description(sd, w, h)
        it.preprocess(sc, cam, sampler=sampler, sampleCount=spp, seed=21)
        it.clear_film()
        assert it.render()
        kind = {"independent": 0, "ldsampler": 1, "halton": 2, "hammersley": 3, "stratified": 4}[sampler]
        o, _ = orc.render(osc.scene, orc.make_camera(sd, w, h), orc.render_params(5, sampler=kind, spp=spp, seed=21))
        assert np.array_equal(it.film().view(np.uint32), o.view(np.uint32)), (w, h, sampler)
    # rendering twice without clearing accumulates (Film::putImageBlock semantics)
    f1 = it.film().copy(); assert it.render()
    f2 = it.film()                                                      # (f1 + a) + b, not f1 + (a + b): equal up to rounding
    assert np.allclose(f2, f1 + f1, rtol=1e-5, atol=1e-6) and not np.array_equal(f2, f1)
    # misuse of the C ABI
    assert L.mtsgpu_upload_scene(None, None) < 0 and L.mtsgpu_set_camera(it._ctx, None) < 0
    assert L.mtsgpu_read_film(it._ctx, None) < 0 and L.mtsgpu_trace_rays(it._ctx, None, 5, 0, None) < 0
    assert L.mtsgpu_set_tiles(it._ctx, 0, 0, 1) < 0 and L.mtsgpu_set_tiles(it._ctx, 32, 3, 2) < 0
    assert L.mtsgpu_set_sampler(it._ctx, 99, 4, 3, 0) < 0 and L.mtsgpu_set_sampler(it._ctx, 1, 0, 3, 0) < 0
    assert L.mtsgpu_set_rfilter(it._ctx, -1.0, 2.0, (C.c_float * 256)()) < 0
    bad = mts.abi.Camera(); bad.width = -4; bad.height = 10
    assert L.mtsgpu_set_camera(it._ctx, C.byref(bad)) < 0
    assert b"" != L.mtsgpu_last_error(it._ctx)
    L.mtsgpu_destroy(None)                                             # a no-op
    # a corrupted scene is refused before anything reaches the GPU
    arr = scA.arrays()
    broken = mts.abi.Scene.from_buffer_copy(scA.sc)
    nodes = arr["kd_nodes"].copy(); nodes[0, 0] = (nodes[0, 0] & 3) | (0x3FFFFFF << 2)      # child offset far outside the array
    broken.kd_nodes = mts.abi.ptr(nodes, mts.abi.u32p)
    assert L.mtsgpu_upload_scene(it._ctx, C.byref(broken)) == -1
    idx = arr["kd_indices"].copy(); idx[0] = 0x7FFFFFFF
    broken = mts.abi.Scene.from_buffer_copy(scA.sc); broken.kd_indices = mts.abi.ptr(idx, mts.abi.u32p)
    assert L.mtsgpu_upload_scene(it._ctx, C.byref(broken)) == -1
    # the context still works afterwards
    it.preprocess(scA, mts.PerspectiveCamera.for_description(sdA, 16, 16), sampler="independent", sampleCount=2)
    it.clear_film(); assert it.render()


def test_orthographic_camera(gpu_lib, mts, orc):
    """`orthographic` camera plugin (src/cameras/orthographic.cpp:104-118): parallel rays, mint = 0, maxt = far - near"""
    sd = mts.scenes.cornell_c5(sphere_subdiv=2)
    sd.camera = dict(origin=(0.1, 1.0, 3.4), target=(0.0, 0.9, 0.0), up=(0.0, 1.0, 0.0), ortho_scale=(0.95, 0.95))
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, 44, 36); ocam = orc.make_camera(sd, 44, 36)
    assert cam.c.kind == 1 and bytes(cam.c) == bytes(ocam)
    it = mts.MIPathTracer(maxDepth=6)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=16, seed=5)
    assert it.render()
    op = orc.render_params(6, sampler=mts.abi.SAMPLER_LD_KEYED, spp=16, seed=5)
    ofilm, _ = orc.render(oscene.scene, ocam, op)
    assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32)) and ofilm[..., :3].max() > 0
    ps = np.array([[x, y, j] for x in (0, 21, 43) for y in (0, 17, 35) for j in (0, 7)], dtype=np.uint32)
    assert np.array_equal(it.li_samples(ps).view(np.uint32), orc.li_samples(oscene.scene, ocam, op, ps).view(np.uint32))


def test_bunny_benchmark_rays(gpu_lib, mts, orc):
    """the reference's own traversal benchmark (src/tests/test_kd.cpp:85-130): chords of the test's sphere through the
    kd-tree of data/tests/bunny.ply (tests/golden/bunny.ply), rays drawn exactly as the test draws them"""
    import ply_io
    pos, tri = ply_io.read(os.path.join(os.path.dirname(__file__), "golden", "bunny.ply"))
    assert pos.shape == (35947, 3) and tri.shape == (69451, 3)
    sd = mts.scenes.SceneDescription("bunny")
    sd.add_mesh(pos, tri, bsdf=sd.lambertian(0.5), face_normals=False, name="bunny")
    sd.point_light((0.0, 0.5, 0.5), 1.0)                               # a scene needs a luminaire (scene.cpp:310-318)
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    a, b = scene.arrays(), oscene.arrays()
    assert all(np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes() for k in a)      # both builders, 160 701 nodes
    assert scene.sc.n_nodes == 160701 and scene.sc.n_indices == 235974
    it = mts.MIPathTracer(maxDepth=2)
    it.preprocess(scene, mts.PerspectiveCamera.for_description(sd, 16, 16), sampler="independent", sampleCount=1)
    rays = orc.chord_rays((-0.016840, 0.110154, -0.001537), 0.2, 300000)
    got = it.trace_rays(rays, shadow=True); exp = orc.trace_rays(oscene.scene, rays, shadow=True)
    assert np.array_equal(got[:, 3], exp[:, 3])
    assert abs(exp[:, 3].mean() - 0.1044) < 0.003                      # 10.44 % of the chords hit the bunny
    got = it.trace_rays(rays); exp = orc.trace_rays(oscene.scene, rays)
    assert np.array_equal(got, exp)


def test_edge_cases(gpu_lib, mts, orc):
    """empty and ragged inputs, degenerate geometry, non-finite rays, the largest LD sample count"""
    F = np.float32
    # --- a scene whose camera sees nothing: every ray misses, the film stays black, no kernel hangs ---
    sd = mts.scenes.cornell_c1()
    sd.camera = dict(origin=(0.0, 1.0, 5.0), target=(0.0, 1.0, 9.0), up=(0.0, 1.0, 0.0), fov=30.0)
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, 33, 17); ocam = orc.make_camera(sd, 33, 17)     # ragged tiles
    it = mts.MIPathTracer(maxDepth=4)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=4, seed=2)
    assert it.render()
    film = it.film()
    assert film[..., :4].max() == 0 and (film[..., 4] > 0).all()
    ofilm, _ = orc.render(oscene.scene, ocam, orc.render_params(4, sampler=mts.abi.SAMPLER_LD_KEYED, spp=4, seed=2))
    assert np.array_equal(film.view(np.uint32), ofilm.view(np.uint32))
    # --- empty inputs of the test hooks ---
    assert it.trace_rays(np.zeros((0, 8), dtype=np.float32)).shape == (0, 4)
    assert it.li_samples(np.zeros((0, 3), dtype=np.uint32)).shape == (0, 8)
    # --- non-finite and degenerate rays: same answers as the oracle, and the kernel terminates ---
    rays = np.zeros((8, 8), dtype=np.float32)
    rays[:, 0:3] = (0.0, 1.0, 3.0); rays[:, 3] = 1e-4; rays[:, 4:7] = (0.0, 0.0, -1.0); rays[:, 7] = np.inf
    rays[1, 4] = np.nan; rays[2, 4:7] = 0.0; rays[3, 4:7] = (np.inf, 0.0, -1.0); rays[4, 0] = np.nan
    rays[5, 7] = 0.0; rays[6, 3] = np.inf; rays[7, 4:7] = (0.0, 0.0, 1.0)
    sd1 = mts.scenes.cornell_c1(); sc1 = mts.Scene(sd1); osc1 = orc.FlatScene(sd1)
    it1 = mts.MIPathTracer(maxDepth=4)
    it1.preprocess(sc1, mts.PerspectiveCamera.for_description(sd1, 16, 16), sampler="independent", sampleCount=1)
    assert np.array_equal(it1.trace_rays(rays), orc.trace_rays(osc1.scene, rays))
    assert np.array_equal(it1.trace_rays(rays, shadow=True)[:, 3], orc.trace_rays(osc1.scene, rays, shadow=True)[:, 3])
    # --- degenerate (zero-area) and duplicated triangles next to ordinary ones ---
    sd2 = mts.scenes.cornell_c1()
    pos = np.array([[-0.3, 0.5, 0.0], [0.3, 0.5, 0.0], [0.0, 1.1, 0.0], [0.1, 0.1, 0.1]], dtype=F)
    tri = np.array([[0, 1, 2], [0, 1, 2], [3, 3, 3], [0, 1, 1], [2, 1, 0]], dtype=np.uint32)
    sd2.add_mesh(pos, tri, bsdf=sd2.lambertian(0.5), face_normals=True, name="degenerate")
    sc2 = mts.Scene(sd2); osc2 = orc.FlatScene(sd2)
    assert np.array_equal(sc2.arrays()["kd_nodes"], osc2.arrays()["kd_nodes"])
    it2 = mts.MIPathTracer(maxDepth=5)
    cam2 = mts.PerspectiveCamera.for_description(sd2, 31, 29)
    it2.preprocess(sc2, cam2, sampler="independent", sampleCount=8, seed=3)
    assert it2.render()
    o2, _ = orc.render(osc2.scene, orc.make_camera(sd2, 31, 29), orc.render_params(5, spp=8, seed=3))
    assert np.array_equal(it2.film().view(np.uint32), o2.view(np.uint32))
    chords = chord_rays(5000, (0, 1, 0), 2.2, seed=5)
    assert np.array_equal(it2.trace_rays(chords), orc.trace_rays(osc2.scene, chords))      # equal-t ties between the duplicates
    # --- the largest sample count of the LD sampler's 16-bit permutations, on a 2 x 1 image ---
    it3 = mts.MIPathTracer(maxDepth=3)
    cam3 = mts.PerspectiveCamera.for_description(sd1, 2, 1)
    it3.preprocess(sc1, cam3, sampler="ldsampler", sampleCount=65536, seed=4)
    assert it3.render()
    o3, _ = orc.render(osc1.scene, orc.make_camera(sd1, 2, 1), orc.render_params(3, sampler=mts.abi.SAMPLER_LD_KEYED, spp=65536, seed=4))
    assert np.array_equal(it3.film().view(np.uint32), o3.view(np.uint32))
    with pytest.raises(mts.MtsGpuError):
        it3.preprocess(sc1, cam3, sampler="ldsampler", sampleCount=65537)                 # would not fit the tables
    # 512 samples: the largest tables the LDS kernel holds (64 KB); 1024: the first size of the plain kernel
    for spp in (512, 1024):
        cam4 = mts.PerspectiveCamera.for_description(sd1, 12, 10)
        it3.preprocess(sc1, cam4, sampler="ldsampler", sampleCount=spp, seed=9)
        it3.clear_film(); assert it3.render()
        o4, _ = orc.render(osc1.scene, orc.make_camera(sd1, 12, 10), orc.render_params(3, sampler=mts.abi.SAMPLER_LD_KEYED, spp=spp, seed=9))
        assert np.array_equal(it3.film().view(np.uint32), o4.view(np.uint32)), spp


# --------------------------------------------------------------------------------------------
# BASELINE.json full size: the 1 044 482-triangle C3 scene at 1024 x 1024
# --------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c3_full(gpu_lib, mts, orc):
    sd = mts.scenes.cornell_c3()                      # grid 320 -> 1 024 000 wall triangles
    scene = mts.Scene(sd)
    oscene = orc.FlatScene(sd)
    assert scene.sc.n_tris == 1044482
    # the two independent host implementations agree on the whole tree
    assert np.array_equal(scene.arrays()["kd_nodes"], oscene.arrays()["kd_nodes"])
    assert np.array_equal(scene.arrays()["kd_indices"], oscene.arrays()["kd_indices"])
    return sd, scene, oscene


def test_full_size_li_samples_bit_exact(c3_full, mts, orc):
    sd, scene, oscene = c3_full
    W = H = 1024
    cam = mts.PerspectiveCamera.for_description(sd, W, H)
    it = mts.MIPathTracer(maxDepth=sd.max_depth)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=64, seed=0x5EED)
    rng = np.random.RandomState(11)
    ps = np.stack([rng.randint(0, W, 3000), rng.randint(0, H, 3000), rng.randint(0, 64, 3000)], axis=1).astype(np.uint32)
    got = it.li_samples(ps)
    op = orc.render_params(sd.max_depth, sampler=mts.abi.SAMPLER_LD_KEYED, spp=64, seed=0x5EED)
    exp = orc.li_samples(oscene.scene, orc.make_camera(sd, W, H), op, ps)
    bad = (got.view(np.uint32) != exp.view(np.uint32)).any(axis=1)
    assert not bad.any(), "%d of %d samples differ" % (bad.sum(), len(ps))
    assert exp[:, 6].max() >= 10                       # deep paths (glass + russian roulette) are covered


def test_full_size_frame_properties(c3_full, mts, orc):
    """size-independent properties at 1024^2: determinism, tile sharding, weight/alpha checksums,
    and a crop rendered by the oracle"""
    sd, scene, oscene = c3_full
    W = H = 1024
    spp = 4
    cam = mts.PerspectiveCamera.for_description(sd, W, H)
    it = mts.MIPathTracer(maxDepth=sd.max_depth)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=3)
    assert it.render()
    a = it.film()
    it.clear_film(); assert it.render()
    assert np.array_equal(a.view(np.uint32), it.film().view(np.uint32))          # run-to-run identical
    st = it.stats()
    assert st["camera_samples"] == W * H * spp
    # sharding over 5 parts (what 5 GPUs would render) sums to the same film
    acc = np.zeros_like(a)
    for part in range(5):
        it.clear_film(); it.set_tiles(32, part, 5); assert it.render()
        acc += it.film()
    assert np.array_equal(acc.view(np.uint32), a.view(np.uint32))
    # checksums: every sample lands in exactly one pixel with weight 1 (or 0 on a pixel border)
    w = a[..., 4].astype(np.float64)
    assert w.max() <= spp and w.sum() >= W * H * spp * 0.999
    assert (a[..., 3] <= a[..., 4]).all() and np.isfinite(a).all() and (a[..., :3] >= 0).all()
    # oracle crop at matched seeds
    op = orc.render_params(sd.max_depth, sampler=mts.abi.SAMPLER_LD_KEYED, spp=spp, seed=3)
    crop, _ = orc.render(oscene.scene, orc.make_camera(sd, W, H), op, rect=(600, 640, 664, 680))
    assert np.array_equal(crop[640:680, 600:664].view(np.uint32), a[640:680, 600:664].view(np.uint32))
    assert crop[640:680, 600:664, :3].max() > 0


def test_c4_at_full_size_against_oracle_crops(c3_full, mts, orc):
    """BASELINE.json configs[3] exactly as bench.py times it at N = 1: the 1 044 482-triangle scene, 1024 x 1024, ldsampler at
    4096 samples per pixel, maxDepth 16, the default pass size (57 passes of 18 432 pixels: k_ld_scout + k_ld_apply_lds,
    k_accumulate_wave at full size).  Three 2 x 2 crops of the oracle at matched seeds -- one under the glass sphere, one on
    the back wall, one in the last (ragged) pass -- are equal bit for bit, and every camera sample was drawn"""
    sd, scene, oscene = c3_full
    W = H = 1024
    spp = 4096
    cam = mts.PerspectiveCamera.for_description(sd, W, H)
    it = mts.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=0x5EED)
    assert it.render()
    film = it.film()
    st = it.stats()
    assert st["camera_samples"] == W * H * spp
    w = film[..., 4].astype(np.float64)
    # every sample lands in one pixel with weight 1, except the few that ImageBlock::putSample refuses (Spectrum::isValid: the
    # 0/0 of the power heuristic, path.cpp:218-222) -- 8e-5 of them on this scene; the crops below pin the exact values
    assert abs(w.sum() - float(W * H) * spp) <= 5e-4 * W * H * spp and w.max() <= spp + 16 and np.isfinite(film).all()
    # the pixel the centre of the glass sphere projects to
    c2w = np.array(list(cam.c.camera_to_world), dtype=np.float64).reshape(4, 4)
    r2c = np.array(list(cam.c.raster_to_camera), dtype=np.float64).reshape(4, 4)
    p = np.linalg.inv(r2c) @ np.linalg.inv(c2w) @ np.array([0.3, 0.4, 0.2, 1.0])
    gx, gy = int(p[0] / p[3]) & ~1, int(p[1] / p[3]) & ~1
    assert 0 < gx < W - 2 and 0 < gy < H - 2
    ocam = orc.make_camera(sd, W, H)
    op = orc.render_params(sd.max_depth, rr_depth=sd.rr_depth, sampler=mts.abi.SAMPLER_LD_KEYED, spp=spp, seed=0x5EED)
    for x0, y0 in ((gx, gy), (512, 300), (W - 2, H - 2)):
        o, ost = orc.render(oscene.scene, ocam, op, rect=(x0, y0, x0 + 2, y0 + 2))
        assert np.array_equal(film[y0:y0 + 2, x0:x0 + 2].view(np.uint32), o[y0:y0 + 2, x0:x0 + 2].view(np.uint32)), (x0, y0)
        assert o[y0:y0 + 2, x0:x0 + 2, :3].max() > 0
    # the sphere crop really is glass: paths through it run deeper than the diffuse walls' (dielectric: no russian roulette)
    og, ostg = orc.render(oscene.scene, ocam, op, rect=(gx, gy, gx + 2, gy + 2))
    ob, ostb = orc.render(oscene.scene, ocam, op, rect=(512, 300, 514, 302))
    assert ostg.path_length_sum > ostb.path_length_sum


def test_gaussian_rfilter_matches_oracle(gpu_lib, mts, orc):
    """the default film filter (film.cpp:89-95): bordered ImageBlocks + Film::putImageBlock"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c5_small", W=80, H=72, sampler="ldsampler", spp=8)
    size, values = it.set_rfilter("gaussian")
    of = orc.tabulate_filter("gaussian")
    assert np.array_equal(values.view(np.uint32), np.array(of.values, dtype=np.float32).view(np.uint32)) and size[0] == of.size_x
    assert it.render()
    film = it.film()
    ofilm, ost = orc.render_tiles(oscene.scene, ocam, op, of)
    assert np.array_equal(film.view(np.uint32), ofilm.view(np.uint32))
    # many small passes (whole tiles per pass) give the same film
    it.set_options(max_paths=32 * 32 * 8 * 2)
    it.clear_film(); assert it.render()
    assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32))
    it.set_options(max_paths=0)
    # tile sharding: blocks of different parts overlap in their borders; the sum order differs by association only
    acc = np.zeros_like(film)
    for part in range(3):
        it.clear_film(); it.set_tiles(32, part, 3); assert it.render()
        acc += it.film()
    assert np.allclose(acc, film, rtol=2e-6, atol=1e-7)
    # Film::hasHighQualityEdges: tiles start at (-border, -border), samples outside the film are traced too
    it.set_tiles(32, 0, 1); it.set_film_edges(True); it.clear_film(); assert it.render()
    ohq, ohst = orc.render_tiles(oscene.scene, ocam, op, of, hq_edges=True)
    assert np.array_equal(it.film().view(np.uint32), ohq.view(np.uint32))
    assert it.stats()["camera_samples"] == ohst.camera_samples == (80 + 4) * (72 + 4) * 8
    assert not np.array_equal(ohq.view(np.uint32), ofilm.view(np.uint32))
    it.set_film_edges(False)
    # the other reconstruction filter plugins (negative lobes)
    for kind in ("mitchell", "catmullrom", "wsinc"):
        it.set_rfilter(kind); it.clear_film(); assert it.render()
        ok, _ = orc.render_tiles(oscene.scene, ocam, op, orc.tabulate_filter(kind))
        assert np.array_equal(it.film().view(np.uint32), ok.view(np.uint32)), kind
    # back to the box filter
    it.set_rfilter("box"); it.set_tiles(32, 0, 1); it.clear_film(); assert it.render()
    obox, _ = orc.render(oscene.scene, ocam, op)
    assert np.array_equal(it.film().view(np.uint32), obox.view(np.uint32))
