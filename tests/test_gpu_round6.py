"""GPU tests, sixth set: what round 6 added or left uncovered, each against the oracle."""
import numpy as np
import pytest

from test_gpu_parity import _setup

pytestmark = pytest.mark.gpu


def test_ld_apply_in_memory_with_many_slots_groups_and_chunked_launches(gpu_lib, mts, orc):
    """LowDiscrepancySampler::generate() above 16 384 samples per pixel (k_ld_scout + k_ld_apply: a lane per table, the 64
    slots of a wave interleaved in a scratch copy, the result transposed through LDS into the per-pixel rows) with everything
    the single-slot table read-out cannot reach: 384 pixels = six groups of 64 slots, 48 tables per pixel = 288 (group, table)
    pairs = two chunked launches (block0 > 0), and a second pass whose last group is ragged (200 + 184 slots).  The film
    equals the oracle's in crops from the first group, a middle group, the pass boundary and the ragged last group"""
    W, H, spp, depth = 24, 16, 32768, 24
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c1", W=W, H=H, sampler="ldsampler", spp=spp, max_depth=2)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, depth=depth, seed=0x5EED)
    op = orc.render_params(2, sampler=mts.abi.SAMPLER_LD_KEYED, spp=spp, ld_depth=depth, seed=0x5EED)
    it.set_options(max_paths=200 * spp)
    assert it.render()
    film = it.film()
    assert it.stats()["camera_samples"] == W * H * spp
    for x0, y0 in ((0, 0), (10, 5), (6, 8), (8, 8), (22, 15)):      # slot 200 = pixel (8, 8): the first of the second pass
        o, _ = orc.render(oscene.scene, ocam, op, rect=(x0, y0, x0 + 2, y0 + 1))
        assert np.array_equal(film[y0:y0 + 1, x0:x0 + 2].view(np.uint32), o[y0:y0 + 1, x0:x0 + 2].view(np.uint32)), (x0, y0)
        assert o[y0, x0, 4] > 0


def _group(mts, devices):
    sd = mts.scenes.cornell_c1()
    scene = mts.Scene(sd); cam = mts.PerspectiveCamera.for_description(sd, 64, 48)
    g = mts.DeviceGroup(devices, maxDepth=4)
    g.preprocess(scene, cam, sampler="independent", sampleCount=4)
    return g, scene, cam


def test_group_rccl_self_check_and_rank_count(gpu_lib, mts):
    """mtsgpu_group_render checks its communicator before the first frame depends on it (ncclCommInitAll gave every member a
    communicator, a one-float sum of ones arrives as the number of members) and reports the verified rank count; a probe
    that fails (injected) leaves the group on the ordered sum with the reason, and is an error when RCCL was demanded"""
    g, scene, cam = _group(mts, [0])
    assert g.rccl_ranks() == 0                           # no collective asked for yet
    assert g.render(ordered_reduce=2) and g.reduce_kind() == "rccl ncclReduce"
    assert g.rccl_ranks() == 1
    ref = g.film()
    g.set_tuning(rccl_fail=1)                            # the collective of the next frame fails: RCCL is given up
    assert g.render(ordered_reduce=2) and g.rccl_ranks() == 0 and np.array_equal(g.film().view(np.uint32), ref.view(np.uint32))
    g2, _, _ = _group(mts, [0])
    g2.set_tuning(rccl_fail=1)                           # the probe itself fails
    with pytest.raises(mts.MtsGpuError, match="self-check"):
        g2.render(ordered_reduce=2)
    assert g2.rccl_ranks() == 0
    g2.set_tuning(rccl_fail=0)
    assert g2.render() and np.array_equal(g2.film().view(np.uint32), ref.view(np.uint32))


def test_group_rccl_reduce_equals_the_ordered_sum_on_eight_gpus(gpu_lib, mts):
    """needs eight GPUs (skipped below): the node-sized form of the two-GPU test -- ncclCommInitAll over eight devices, the
    self-check, ONE ncclReduce of the films (renderproc.cpp:123-130 in one collective) against the ordered peer-copy sum and
    against the unsharded render; with the box filter every pixel has one writer, so all three films are equal bit for bit"""
    import torch
    if torch.cuda.device_count() < 8:
        pytest.skip("fewer than eight GPUs")
    g, scene, cam = _group(mts, list(range(8)))
    assert g.render(ordered_reduce=1) and g.reduce_kind() == "ordered peer-copy sum"
    a = g.film()
    try:
        assert g.render(ordered_reduce=2)
    except mts.MtsGpuError as e:
        pytest.skip("RCCL could not be initialised here: %s" % e)
    assert np.array_equal(g.film().view(np.uint32), a.view(np.uint32)), g.reduce_note()
    if g.reduce_kind() != "rccl ncclReduce":
        pytest.skip("the collective fell back to the ordered sum (film still equal): %s" % g.reduce_note())
    assert g.rccl_ranks() == 8
    one, _, _ = _group(mts, [0])
    assert one.render() and np.array_equal(one.film().view(np.uint32), a.view(np.uint32))


def test_exact_tail_filter_variant_matches_the_oracle(gpu_lib, mts, orc, tmp_path):
    """the record-tail filter build of the traversal kernels (tools/build_variant.sh tf -DMG_TAIL_FILTER=1; not the product:
    it is time-neutral, profiles/r06g_*) skips a quarter of the record tails and must change nothing: the film of a child
    process that loads the variant equals the oracle's bit for bit.  Skipped when the variant library has not been built"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "mitsuba-renderer_amd", "libmtsgpu_tf.so")
    if not os.path.exists(lib):
        pytest.skip("libmtsgpu_tf.so not built")
    out = str(tmp_path / "film.npy")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import _pkgload; pkg = _pkgload.load(); "
            "sd = pkg.scenes.cornell_c5(sphere_subdiv=3); it = pkg.MIPathTracer(maxDepth=12); "
            "it.preprocess(pkg.Scene(sd), pkg.PerspectiveCamera.for_description(sd, 96, 96), sampler='ldsampler', sampleCount=16, seed=11); "
            "assert it.render(); assert pkg.lib().mtsgpu_source_hash; np.save(%r, it.film())") % (root, out)
    env = dict(os.environ, MTSGPU_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    sd = mts.scenes.cornell_c5(sphere_subdiv=3)
    oscene = orc.FlatScene(sd)                              # kept alive: the oracle's scene lives as long as this object
    ofilm, _ = orc.render(oscene.scene, orc.make_camera(sd, 96, 96),
                          orc.render_params(12, sampler=mts.abi.SAMPLER_LD_KEYED, spp=16, seed=11))
    assert np.array_equal(np.load(out).view(np.uint32), ofilm.view(np.uint32))


def test_traversal_launch_orders_do_not_change_the_film(gpu_lib, mts, orc):
    """host-driven bounces with the any-hit launch of a bounce in front of / behind the next closest-hit launch on a second
    stream (overlap = 1 / 2) and inside it (merged = 1: k_trace_pair, one launch per bounce; also when that launch is repeated
    with static dealing): the any-hit kernel only parks direct-light terms, so every order gives the oracle's film"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c5_small", W=48, H=40, sampler="ldsampler", spp=16, max_depth=8)
    ofilm, _ = orc.render(oscene.scene, ocam, op)
    for knobs in (dict(sync_free=0), dict(sync_free=0, overlap=1), dict(sync_free=0, overlap=2), dict(sync_free=0, overlap=2, overlap_delay_us=20),
                  dict(sync_free=0, overlap=0, merged=1), dict(sync_free=0, merged=1, test_retry=1)):
        it.set_tuning(**knobs)
        it.clear_film(); assert it.render()
        assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32)), knobs
    it.set_tuning(sync_free=0, merged=1, test_retry=0)
    it.set_options(max_paths=16 * 500)                     # ragged passes
    it.clear_film(); assert it.render()
    assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32))


def test_closest_hit_without_the_mailbox_hands_ties_back(gpu_lib, mts, orc):
    """host-driven bounces trace closest hits WITHOUT the hashed mailbox (one more level of the tree in its LDS instead); the one
    case in which the mailbox decides a result -- two primitives tied in t: the later test wins, and which tests run depends on
    the mailbox (sahkdtree3.h:130-144, :278-283) -- is detected per ray and traced again by the kernel with the mailbox.  A box whose
    every wall exists twice, with different reflectances, makes every hit a tie: the film equals the oracle's bit for bit, with the
    knob on and off, and the statistics show the rays that went back"""
    S = mts.scenes
    sd = S.SceneDescription("tied_walls")
    a, b = sd.lambertian(0.7, 0.2, 0.2), sd.lambertian(0.2, 0.7, 0.2)
    for name, p0, e1, e2, nrm in S._box_faces():
        pos, tri = S._quad(p0, e1, e2, nrm)
        sd.add_mesh(pos, tri, bsdf=a, face_normals=True, name=name)
        sd.add_mesh(pos.copy(), tri.copy(), bsdf=b, face_normals=True, name=name + "_again")
    pos, tri = S.icosphere(2, 0.35, (0.1, 0.45, 0.0))
    sd.add_mesh(pos, tri, bsdf=sd.dielectric(), face_normals=False, name="glass")
    S._add_light(sd)
    sd.max_depth = 6
    W, H, spp = 48, 40, 16
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, W, H); ocam = orc.make_camera(sd, W, H)
    ofilm, _ = orc.render(oscene.scene, ocam, orc.render_params(6, sampler=mts.abi.SAMPLER_LD_KEYED, spp=spp, seed=5))
    it = mts.MIPathTracer(maxDepth=6)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=5)
    redone = {}
    for knobs in (dict(sync_free=0, mailbox_free=1), dict(sync_free=0, mailbox_free=0), dict(sync_free=0, mailbox_free=1, test_retry=1),
                  dict(sync_free=1, mailbox_free=1, test_retry=0)):
        it.set_tuning(**knobs)
        it.clear_film(); assert it.render()
        assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32)), knobs
        redone[tuple(sorted(knobs.items()))] = it.stats()["rays_redone"]
    vals = list(redone.values())
    assert vals[0] > 0.3 * it.stats()["rays_closest"] and vals[1] == 0 and vals[3] == 0, redone      # device-driven bounces keep the mailbox
    # an ordinary scene hands back next to nothing
    sd2, scene2, oscene2, cam2, ocam2, it2, op2 = _setup(mts, orc, "c5_small", W=48, H=40, sampler="ldsampler", spp=16, max_depth=8)
    it2.set_tuning(sync_free=0, mailbox_free=1)
    assert it2.render()
    assert np.array_equal(it2.film().view(np.uint32), orc.render(oscene2.scene, ocam2, op2)[0].view(np.uint32))
    assert it2.stats()["rays_redone"] < 1e-3 * it2.stats()["rays_closest"]
