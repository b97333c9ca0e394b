"""GPU parity, third set: the reference's chi-square procedure run against the DEVICE BSDF code (no oracle in between),
and the BASELINE.json configurations C2 and C5 at their stated parameters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from chisquare_ref import WI_SAMPLES, bsdf_models, chi_square, square_to_sphere


@pytest.mark.parametrize("index", range(10))
def test_device_bsdfs_pass_the_reference_chi_square(gpu_lib, mts, index):
    """HIP <-> reference-held test material, directly: the procedure of src/tests/test_chisquare.cpp:299-420 (10 x 20
    (theta, phi) contingency table, 200 000 samples and 20 incident directions per model, cells pooled below an expected
    frequency of 5, significance 0.005 with the Sidak correction) on sample() / pdf() / f() as k_shade executes them
    (mtsgpu_bsdf_eval), for the models of data/tests/test_bsdf.xml"""
    name, btype, params, back = bsdf_models(mts)[index]
    it = mts.MIPathTracer()
    failures = chi_square(it.bsdf_eval, btype, params, back, np.random.RandomState(1000 + index))
    assert not failures, "%s: chi-square rejected for %d of %d incident directions: %s" % (name, len(failures), WI_SAMPLES, failures[:3])


def test_device_bsdf_eval_is_the_oracle_bit_for_bit(gpu_lib, mts, orc):
    """the same read-out against the CPU restatement: f, pdf and sample of every model, random query records"""
    it = mts.MIPathTracer()
    rng = np.random.RandomState(5)
    extra = [("dielectric", 1, np.array([1.5046, 1, 1, 1, 1, 1, 1, 1] + [0] * 8, dtype=np.float32), True),
             ("mirror", 4, np.array([0.8, 0.8, 0.8] + [0] * 13, dtype=np.float32), False)]
    for name, btype, params, back in bsdf_models(mts) + extra:
        n = 20000
        wi = square_to_sphere(rng.random_sample((n, 2)).astype(np.float32))
        wo = square_to_sphere(rng.random_sample((n, 2)).astype(np.float32))
        s = rng.random_sample((n, 2)).astype(np.float32)
        for op, aux in ((0, wo), (1, wo), (2, s)):
            g, e = it.bsdf_eval(btype, params, op, wi, aux), orc.bsdf_eval(btype, params, op, wi, aux)
            if op == 2:
                # a failed sample (f = 0 or pdf = 0) only has to say "no contribution" on both sides
                dead_g = (g[:, 3] == 0) | ~g[:, 4:7].any(axis=1); dead_e = (e[:, 3] == 0) | ~e[:, 4:7].any(axis=1)
                assert np.array_equal(dead_g, dead_e), (name, "sample validity")
                g, e = g[~dead_e], e[~dead_e]
                assert len(g) > n // 10 or btype in (1, 4), name
            assert np.array_equal(g.view(np.uint32), e.view(np.uint32)), (name, op)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] and configs[4] at their stated parameters (C2, C5)
# ---------------------------------------------------------------------------------------------------------------------
def test_c2_as_stated(gpu_lib, mts, orc):
    """C2: the C1 Cornell box (12 triangles), `path` maxDepth 4, `ldsampler` 1024 spp, seed 0x5EED -- 64 x 64 pixels, the whole
    film against the oracle bit for bit, and the 256^2 frame of SURVEY.md 8d (P size) against oracle crops"""
    sd = mts.scenes.cornell_c1()
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    spp, depth = 1024, 4
    it = mts.MIPathTracer(maxDepth=depth)
    op = orc.render_params(depth, sampler=mts.abi.SAMPLER_LD_KEYED, spp=spp, seed=0x5EED)
    cam = mts.PerspectiveCamera.for_description(sd, 64, 64)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=0x5EED)
    assert it.render()
    film = it.film()
    ofilm, _ = orc.render(oscene.scene, orc.make_camera(sd, 64, 64), op)
    assert np.array_equal(film.view(np.uint32), ofilm.view(np.uint32))
    # (the LD sampler can return exactly 1.0, ldsampler.cpp:111: such a sample lands in the next pixel)
    assert film[..., 4].min() >= spp - 2 and film[..., 4].sum() >= 64 * 64 * spp - 64 and film[..., :3].max() > 0
    st = it.stats()
    assert st["camera_samples"] == 64 * 64 * spp
    W = H = 256
    cam = mts.PerspectiveCamera.for_description(sd, W, H)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=0x5EED)
    assert it.render()
    film = it.film()
    ocam = orc.make_camera(sd, W, H)
    for (x0, y0, x1, y1) in ((0, 0, 24, 16), (116, 120, 140, 136), (232, 240, 256, 256)):
        crop, _ = orc.render(oscene.scene, ocam, op, rect=(x0, y0, x1, y1))
        assert np.array_equal(crop[y0:y1, x0:x1].view(np.uint32), film[y0:y1, x0:x1].view(np.uint32)), (x0, y0)


def test_c5_as_stated(gpu_lib, mts, orc):
    """C5: C1 box + four icospheres of subdivision 4 (lambertian, roughmetal, dielectric, microfacet) + constant environment,
    `path` maxDepth 32, 256 spp, 256^2: oracle crops on every sphere and on the background, bit for bit"""
    sd = mts.scenes.cornell_c5(sphere_subdiv=4)
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    W = H = 256; spp, depth = 256, 32
    assert sd.max_depth == depth
    cam = mts.PerspectiveCamera.for_description(sd, W, H)
    it = mts.MIPathTracer(maxDepth=depth)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=0x5EED)
    assert it.render()
    film = it.film()
    st = it.stats()
    assert st["camera_samples"] == W * H * spp and st["path_length_sum"] > st["camera_samples"]
    op = orc.render_params(depth, sampler=mts.abi.SAMPLER_LD_KEYED, spp=spp, seed=0x5EED)
    ocam = orc.make_camera(sd, W, H)
    # project the sphere centres to find crops that see each material
    c = sd.camera
    o = np.asarray(c["origin"], dtype=np.float64); t = np.asarray(c["target"], dtype=np.float64)
    d = (t - o) / np.linalg.norm(t - o); right = np.cross(d, np.asarray(c["up"], dtype=np.float64)); right /= np.linalg.norm(right)
    up = np.cross(right, d)
    tanh = np.tan(np.radians(c["fov"]) / 2)
    rects = [(0, 0, 16, 12)]
    for centre in ((-0.5, 0.3, -0.4), (0.5, 0.3, -0.4), (-0.5, 0.3, 0.45), (0.5, 0.3, 0.45)):
        v = np.asarray(centre) - o
        z = v @ d
        px = int((0.5 - (v @ right) / (z * tanh) / 2) * W); py = int((0.5 - (v @ up) / (z * tanh) / 2) * H)
        # Mitsuba's lookAt has its x axis pointing left on the image (transform.cpp:174-190); either way the crop is valid
        px = min(max(px, 8), W - 8); py = min(max(py, 6), H - 6)
        rects.append((px - 8, py - 6, px + 8, py + 6))
    lit = 0
    for (x0, y0, x1, y1) in rects:
        crop, _ = orc.render(oscene.scene, ocam, op, rect=(x0, y0, x1, y1))
        assert np.array_equal(crop[y0:y1, x0:x1].view(np.uint32), film[y0:y1, x0:x1].view(np.uint32)), (x0, y0)
        lit += crop[y0:y1, x0:x1, :3].max() > 0
    assert lit == len(rects)


# ---------------------------------------------------------------------------------------------------------------------
# the film reduce of a device group when RCCL lets it down (csrc/group.cpp; reference merge: renderproc.cpp:123-130)
# ---------------------------------------------------------------------------------------------------------------------
def _group(mts, devices):
    sd = mts.scenes.cornell_c1()
    scene = mts.Scene(sd); cam = mts.PerspectiveCamera.for_description(sd, 64, 48)
    g = mts.DeviceGroup(devices, maxDepth=4)
    g.preprocess(scene, cam, sampler="independent", sampleCount=4)
    return g, scene, cam


def test_group_reports_an_rccl_library_that_cannot_be_loaded(gpu_lib, mts, monkeypatch):
    """MTSGPU_RCCL_LIB names the library; a missing one is an error only when RCCL is demanded (ordered_reduce = 2)"""
    monkeypatch.setenv("MTSGPU_RCCL_LIB", "/nonexistent/librccl.so")
    g, scene, cam = _group(mts, [0])
    assert g.render()                                    # one member, RCCL not needed
    ref = g.film()
    with pytest.raises(mts.MtsGpuError, match="dlopen"):
        g.render(ordered_reduce=2)
    # two members on this one GPU: RCCL is not usable anyway (one GPU per rank), the ordered sum gives the film
    g2, _, _ = _group(mts, [0, 0])
    assert g2.render() and g2.reduce_kind() == "ordered peer-copy sum" and "share a device" in g2.reduce_note()
    assert np.array_equal(g2.film().view(np.uint32), ref.view(np.uint32))


def test_group_keeps_the_frame_when_the_collective_fails(gpu_lib, mts, monkeypatch):
    """a failing ncclReduce (injected) must not lose the frame: the films stay as rendered, the group adds them up in
    member order, says why, and stops using RCCL"""
    g, scene, cam = _group(mts, [0])
    assert g.render()
    ref = g.film()
    assert g.render(ordered_reduce=2) and g.reduce_kind() == "rccl ncclReduce" and g.reduce_note() == ""
    assert np.array_equal(g.film().view(np.uint32), ref.view(np.uint32))
    g.set_tuning(rccl_fail=1)
    assert g.render(ordered_reduce=2)
    assert g.reduce_kind() == "ordered peer-copy sum" and "injected" in g.reduce_note()
    assert np.array_equal(g.film().view(np.uint32), ref.view(np.uint32))
    g.set_tuning(rccl_fail=0)
    assert g.render() and np.array_equal(g.film().view(np.uint32), ref.view(np.uint32))


def test_group_rccl_reduce_equals_the_ordered_sum_on_two_gpus(gpu_lib, mts):
    """needs two GPUs (skipped on the one-GPU test box): ncclReduce over two distinct devices against the ordered
    peer-copy sum; with the box filter every pixel has one writer, so the two films are equal bit for bit"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU")
    g, scene, cam = _group(mts, [0, 1])
    assert g.render(ordered_reduce=1) and g.reduce_kind() == "ordered peer-copy sum"
    a = g.film()
    try:
        assert g.render(ordered_reduce=2)
    except mts.MtsGpuError as e:                         # no usable RCCL on this machine: nothing to compare
        pytest.skip("RCCL could not be initialised here: %s" % e)
    # a collective that failed has fallen back to the ordered sum and said why: the film must be right either way
    assert np.array_equal(g.film().view(np.uint32), a.view(np.uint32)), g.reduce_note()
    if g.reduce_kind() != "rccl ncclReduce":
        pytest.skip("the collective fell back to the ordered sum (film still equal): %s" % g.reduce_note())
    one, _, _ = _group(mts, [0])
    assert one.render() and np.array_equal(one.film().view(np.uint32), a.view(np.uint32))
