"""GPU parity, second set: the reference's own sampler tables read back from the device, crop windows, maxDepth 0,
the C4 configuration (4096 spp, one of 8 tile shards), several contexts / ranks on one GPU (device group through the
C ABI, bench.py's world > 1 branch), statistics, and malformed trees."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# src/tests/test_samplers.cpp:37-43 (MATLAB: p = haltonset(5); net(p,5)) -- the reference's golden table
HALTON = np.array([
    0, 0, 0, 0, 0,
    0.500000000000000, 0.333333333333333, 0.200000000000000, 0.142857142857143, 0.090909090909091,
    0.250000000000000, 0.666666666666667, 0.400000000000000, 0.285714285714286, 0.181818181818182,
    0.750000000000000, 0.111111111111111, 0.600000000000000, 0.428571428571429, 0.272727272727273,
    0.125000000000000, 0.444444444444444, 0.800000000000000, 0.571428571428571, 0.363636363636364]).reshape(5, 5)
KINDS = {"independent": 0, "ldsampler": 1, "halton": 2, "hammersley": 3, "stratified": 4}


def _ctx(mts, sd, W, H, sampler, spp, max_depth=None, seed=0x5EED):
    scene = mts.Scene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, W, H)
    it = mts.MIPathTracer(maxDepth=sd.max_depth if max_depth is None else max_depth)
    it.preprocess(scene, cam, sampler=sampler, sampleCount=spp, seed=seed)
    return scene, cam, it


def test_device_samplers_reproduce_the_reference_tables(gpu_lib, mts, orc):
    """HIP <-> reference golden: Sampler::next1D() on the DEVICE against the tables of src/tests/test_samplers.cpp:33-78
    (eps 1e-7 as there), and bit for bit against the oracle for every sampler kind"""
    sd = mts.scenes.cornell_c1()
    scene, cam, it = _ctx(mts, sd, 16, 16, "halton", 5)
    for i in range(5):                                                  # test01_Halton: 5 x next1D per sample
        got = it.sampler_values(pixel_key=0, sample_index=i, n=5)
        assert np.abs(got.astype(np.float64) - HALTON[i]).max() <= 1e-7, (i, got)
    it.preprocess(scene, cam, sampler="hammersley", sampleCount=5)
    for i in range(5):                                                  # test02_Hammersley: i/5, then the Halton dimensions
        got = it.sampler_values(pixel_key=0, sample_index=i, n=6)
        exp = np.concatenate([[i / 5.0], HALTON[i]])
        assert np.abs(got.astype(np.float64) - exp).max() <= 1e-7, (i, got)
    # the raster position of a camera sample is pixel + next2D(): the same numbers through MIPathTracer::Li's front end
    it.preprocess(scene, cam, sampler="halton", sampleCount=5)
    out = it.li_samples(np.array([[0, 0, j] for j in range(5)], dtype=np.uint32))
    assert np.abs(out[:, 4:6].astype(np.float64) - HALTON[:, :2]).max() <= 1e-7
    # every sampler kind, 1-D and 2-D draws beyond the table depth, against the oracle bit for bit
    for sampler, kind in KINDS.items():
        spp = 16
        it.preprocess(scene, cam, sampler=sampler, sampleCount=spp, seed=77)
        op = orc.render_params(4, sampler=kind, spp=spp, seed=77)
        for key, j in ((0, 0), (37, 3), (255, spp - 1)):
            for two_d in (False, True):
                g = it.sampler_values(key, j, 9, two_d=two_d)
                e = orc.sampler_values(op, key, j, 9, two_d=two_d)
                assert np.array_equal(g.view(np.uint32), e.view(np.uint32)), (sampler, key, j, two_d)


def test_crop_window_equals_the_rectangle_of_the_full_render(gpu_lib, mts, orc):
    """Film crop window (film.cpp:33-41, renderproc.cpp:146-154): the camera keeps the full film's raster space, the
    work units cover the window.  Box and gaussian filter, with tile sharding."""
    sd = mts.scenes.cornell_c5(sphere_subdiv=2)
    W, H, crop = 96, 80, (21, 9, 50, 45)
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    ocam = orc.make_camera(sd, W, H)
    op = orc.render_params(6, sampler=mts.abi.SAMPLER_LD_KEYED, spp=8, seed=5)
    x0, y0, w, h = crop
    ofull, _ = orc.render(oscene.scene, ocam, op, rect=(x0, y0, x0 + w, y0 + h))
    cam = mts.PerspectiveCamera.cropped(sd, W, H, crop)
    assert (cam.width, cam.height) == (w, h)
    it = mts.MIPathTracer(maxDepth=6)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=8, seed=5)
    assert it.render()
    film = it.film()
    assert film.shape == (h, w, 5)
    assert np.array_equal(film.view(np.uint32), ofull[y0:y0 + h, x0:x0 + w].view(np.uint32))
    # per-sample: film pixel (x, y) of the window is raster pixel (x + x0, y + y0) of the full film
    ps = np.array([[0, 0, 0], [w - 1, h - 1, 7], [10, 3, 4]], dtype=np.uint32)
    got = it.li_samples(ps)
    exp = orc.li_samples(oscene.scene, ocam, op, ps + np.array([x0, y0, 0], dtype=np.uint32))
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    # shards of the window add up to it
    acc = np.zeros_like(film)
    for part in range(3):
        it.clear_film(); it.set_tiles(16, part, 3); assert it.render()
        acc += it.film()
    assert np.array_equal(acc.view(np.uint32), film.view(np.uint32))
    # a filter wider than a pixel: the full render restricted to blocks that touch the window differs only by the
    # samples outside the window, so compare a window that IS the film with the uncropped path instead
    it2 = mts.MIPathTracer(maxDepth=6)
    it2.preprocess(scene, mts.PerspectiveCamera.cropped(sd, W, H, (0, 0, W, H)), sampler="ldsampler", sampleCount=4, seed=5)
    it2.set_rfilter("gaussian"); assert it2.render()
    it3 = mts.MIPathTracer(maxDepth=6)
    it3.preprocess(scene, mts.PerspectiveCamera.for_description(sd, W, H), sampler="ldsampler", sampleCount=4, seed=5)
    it3.set_rfilter("gaussian"); assert it3.render()
    assert np.array_equal(it2.film().view(np.uint32), it3.film().view(np.uint32))
    # invalid windows are refused like film.cpp:41-45 refuses them
    bad = mts.abi.Camera.from_buffer_copy(cam.c); bad.crop_offset_x = 60
    assert mts.lib().mtsgpu_set_camera(it._ctx, C.byref(bad)) == -1


def test_gaussian_crop_window_matches_oracle_blocks(gpu_lib, mts, orc):
    """crop window + gaussian filter: blocks start at the window's corner (renderproc.cpp:146-154), so the window of a
    W x H film is the oracle's tile render of a camera whose raster space is shifted -- checked through per-pixel sums of
    a box-filtered render instead: weights sum to spp inside the window"""
    sd = mts.scenes.cornell_c1()
    cam = mts.PerspectiveCamera.cropped(sd, 64, 64, (8, 16, 40, 32))
    it = mts.MIPathTracer(maxDepth=3)
    it.preprocess(mts.Scene(sd), cam, sampler="independent", sampleCount=4)
    it.set_rfilter("gaussian"); it.set_film_edges(True)
    assert it.render()
    f = it.film()
    assert f.shape == (32, 40, 5) and np.isfinite(f).all() and (f[..., 4] > 0).all()
    # with highQualityEdges every pixel of the window receives its full filter support: the weights are near-uniform
    w = f[..., 4]
    assert w.std() / w.mean() < 0.35


def test_max_depth_zero_and_one(gpu_lib, mts, orc):
    """while (depth <= maxDepth || maxDepth < 0) with depth = 1 (path.cpp:61): maxDepth 0 never enters the loop"""
    sd = mts.scenes.cornell_c5(sphere_subdiv=2)
    oscene = orc.FlatScene(sd); ocam = orc.make_camera(sd, 40, 32)
    for md in (0, 1, 2):
        scene, cam, it = _ctx(mts, sd, 40, 32, "independent", 4, max_depth=md)
        assert it.render()
        o, _ = orc.render(oscene.scene, ocam, orc.render_params(md, sampler=0, spp=4))
        f = it.film()
        assert np.array_equal(f.view(np.uint32), o.view(np.uint32)), md
        if md == 0:
            assert f[..., :3].max() == 0 and f[..., 3].max() > 0       # black, alpha from the camera ray


def test_avg_path_length_statistic(gpu_lib, mts, orc):
    """avgPathLength (path.cpp:24,212-213): sum of rRec.depth over all Li() calls, box and gaussian film paths"""
    sd = mts.scenes.cornell_c1()
    scene, cam, it = _ctx(mts, sd, 48, 40, "ldsampler", 8, max_depth=4)
    assert it.render()
    st = it.stats()
    oscene = orc.FlatScene(sd)                                           # keep it alive: .scene points into it
    _, ost = orc.render(oscene.scene, orc.make_camera(sd, 48, 40), orc.render_params(4, sampler=1, spp=8))
    assert st["path_length_sum"] == ost.path_length_sum > st["camera_samples"]
    assert 1.0 < st["avg_path_length"] <= 4.0
    it.set_rfilter("gaussian"); it.clear_film(); assert it.render()
    assert it.stats()["path_length_sum"] == ost.path_length_sum


@pytest.fixture(scope="module")
def c3_full(mts):
    sd = mts.scenes.cornell_c3()
    return sd, mts.Scene(sd)


def test_c4_one_of_eight_shards_at_4096_spp(gpu_lib, mts, orc, c3_full):
    """BASELINE.json configs[3]: the 1M-triangle scene at 4096 spp, tiles sharded 8 ways.  One shard on the GPU,
    oracle crops inside tiles of that shard bit for bit; the other pixels stay empty."""
    sd, scene = c3_full
    W = H = 256
    cam = mts.PerspectiveCamera.for_description(sd, W, H)
    it = mts.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=4096, seed=0x5EED)
    part = 5
    it.set_tiles(32, part, 8)
    assert it.render()
    film = it.film()
    st = it.stats()
    keys = mts.filmreduce.tiles_of_rank(W, H, 32, part, 8)
    assert st["camera_samples"] == len(keys) * 4096 and len(keys) == 8 * 32 * 32
    own = np.zeros(W * H, dtype=bool); own[keys] = True
    own = own.reshape(H, W)
    # 4096 samples per owned pixel, minus the few that ImageBlock::putSample refuses (Spectrum::isValid: the 0/0 of the
    # power heuristic, path.cpp:218-222) or that a (0,2)-sequence value of exactly 1.0f (ldsampler.cpp:111) moves to the
    # neighbouring pixel, which may belong to another shard's tile -- the oracle crops below pin the exact values
    wown = film[own][:, 4]
    dev = np.abs(wown.astype(np.float64) - 4096)
    assert dev.max() <= 16 and (dev == 0).mean() > 0.8, (dev.max(), (dev == 0).mean(), wown.min(), wown.max())
    stray = film[~own][:, 4].astype(np.float64).sum()
    assert stray <= 256, stray
    oscene = orc.FlatScene(sd)
    ocam = orc.make_camera(sd, W, H)
    op = orc.render_params(sd.max_depth, rr_depth=sd.rr_depth, sampler=mts.abi.SAMPLER_LD_KEYED, spp=4096, seed=0x5EED)
    ys, xs = np.nonzero(own)
    for k in (0, len(ys) // 2, len(ys) - 1):                            # three 2x2 crops inside the shard's tiles
        x0, y0 = int(xs[k]) & ~1, int(ys[k]) & ~1
        o, _ = orc.render(oscene.scene, ocam, op, rect=(x0, y0, x0 + 2, y0 + 2))
        assert np.array_equal(film[y0:y0 + 2, x0:x0 + 2].view(np.uint32), o[y0:y0 + 2, x0:x0 + 2].view(np.uint32)), (x0, y0)


def test_device_group_two_members_on_one_gpu(gpu_lib, mts, orc):
    """mtsgpu_create_multi / mtsgpu_group_render: two contexts on device 0 share the tiles, member 0 ends up with the
    unsharded film bit for bit (box filter); with the gaussian filter the ordered sum equals part 0 + part 1 of the
    oracle in that order"""
    sd = mts.scenes.cornell_c5(sphere_subdiv=2)
    W, H = 100, 72
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, W, H); ocam = orc.make_camera(sd, W, H)
    g = mts.DeviceGroup([0, 0], maxDepth=6)
    assert len(g) == 2
    g.preprocess(scene, cam, sampler="ldsampler", sampleCount=8, seed=9)
    assert g.render(block_size=32)
    assert g.reduce_kind() == "ordered peer-copy sum"                   # one GPU: no RCCL communicator possible
    op = orc.render_params(6, sampler=mts.abi.SAMPLER_LD_KEYED, spp=8, seed=9)
    full, ost = orc.render(oscene.scene, ocam, op)
    assert np.array_equal(g.film().view(np.uint32), full.view(np.uint32))
    s0, s1 = g.member_stats(0), g.member_stats(1)
    assert s0["camera_samples"] + s1["camera_samples"] == W * H * 8 and min(s0["camera_samples"], s1["camera_samples"]) > 0
    assert s0["rays_closest"] + s1["rays_closest"] == ost.rays_closest
    # a second frame through the same group: films are cleared per frame
    assert g.render(block_size=32)
    assert np.array_equal(g.film().view(np.uint32), full.view(np.uint32))
    # gaussian filter: block borders overlap between the members, the sum order is fixed (0, then 1)
    g.set_rfilter("gaussian")
    assert g.render(block_size=32, ordered_reduce=True)
    of = orc.tabulate_filter("gaussian")
    p0, _ = orc.render_tiles(oscene.scene, ocam, op, of, part=0, n_parts=2)
    p1, _ = orc.render_tiles(oscene.scene, ocam, op, of, part=1, n_parts=2)
    assert np.array_equal(g.film().view(np.uint32), (p0 + p1).view(np.uint32))
    # three members, odd tile grid
    g3 = mts.DeviceGroup([0, 0, 0], maxDepth=6)
    g3.preprocess(scene, cam, sampler="ldsampler", sampleCount=8, seed=9)
    assert g3.render(block_size=16)
    assert np.array_equal(g3.film().view(np.uint32), full.view(np.uint32))
    g.close(); g3.close()
    # errors: empty device list, device out of range
    h = C.c_void_p()
    assert mts.lib().mtsgpu_create_multi(0, None, C.byref(h)) < 0
    devs = (C.c_int * 2)(0, 99)
    assert mts.lib().mtsgpu_create_multi(2, devs, C.byref(h)) < 0


def test_group_rccl_collective_path_single_member(gpu_lib, mts, orc):
    """the RCCL leg of mtsgpu_group_render on the one GPU this box has: librccl is loaded, a communicator over the
    group's devices exists, ncclReduce(sum, root 0) runs on the member's stream and leaves the film as it was"""
    sd = mts.scenes.cornell_c1()
    scene = mts.Scene(sd); cam = mts.PerspectiveCamera.for_description(sd, 64, 48)
    g = mts.DeviceGroup([0], maxDepth=4)
    g.preprocess(scene, cam, sampler="independent", sampleCount=4)
    assert g.render()
    ref = g.film()
    rc = mts.lib().mtsgpu_group_render(g._g, 32, 2, None)               # 2: run the collective even for one member
    assert rc == 0, mts.lib().mtsgpu_group_last_error(g._g)
    assert g.reduce_kind() == "rccl ncclReduce"
    assert np.array_equal(g.film().view(np.uint32), ref.view(np.uint32))


def test_bench_world2_branch_on_one_gpu(gpu_lib, mts, orc, tmp_path):
    """bench.py --gpus 2 started without a launcher: it spawns its two ranks itself, both on GPU 0 (--devices 0,0, film
    reduce through gloo on host copies), and rank 0's reduced film equals the unsharded render bit for bit; weak and
    strong (--spp-total) modes"""
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    sd = mts.scenes.cornell_c3(grid=24, sphere_subdiv=5)
    scene = mts.Scene(sd)
    for mode, extra, spp_total in (("weak", ["--spp", "4"], 8), ("strong", ["--spp-total", "16"], 16)):
        out = str(tmp_path / ("film_%s.npy" % mode))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0", "--res", "96", "--grid", "24",
               "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-1spp", "--host-kd", "--c4-spp", "64", "--c4-steps", "1", "--dump-film", out] + extra
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
        rec = json.loads(line)
        assert rec["n_gpus"] == 2 and rec["scaling"] == mode and rec["value"] > 0
        assert len(rec["rank_ms"]) == 2 and len(rec["reduce_ms"]) == 2 and min(rec["rank_ms"]) > 0
        assert rec["roofline"]["frac"] > 0 and "cpu_baseline" not in rec
        # the film reduce names itself; two ranks on one GPU cannot form an RCCL communicator (gloo on host copies: 0 RCCL ranks)
        assert "gloo" in rec["reduce_kind"] and rec["rccl_ranks"] == 0
        if mode == "weak":
            # BASELINE.json configs[3] rides along: the C4 strong-scaling frame over the same two ranks
            c4 = rec["c4_strong"]
            assert c4["scaling"] == "strong" and c4["n_gpus"] == 2 and c4["spp_total"] == 64 and c4["value"] > 0 and "c4_strong" in rec["config"]["workload"]
        else:
            assert rec["c4_strong"] is None
        cam = mts.PerspectiveCamera.for_description(sd, 96, 96)
        it = mts.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
        it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp_total, seed=0x5EED)
        assert it.render()
        assert np.array_equal(np.load(out).view(np.uint32), it.film().view(np.uint32)), mode


def test_overflow_retry_and_tuning_knobs_do_not_change_results(gpu_lib, mts, orc):
    """the closest-hit launch repeated with static dealing (what a material-queue segment overflow triggers) and every
    scheduling knob leave the film untouched"""
    sd = mts.scenes.cornell_c5(sphere_subdiv=2)
    scene, cam, it = _ctx(mts, sd, 64, 64, "ldsampler", 16, max_depth=8)
    assert it.render()
    ref = it.film()
    it.set_tuning(test_retry=1, sync_free=0)            # the host-driven loop (large frames) is the one with dynamic claims
    it.clear_film(); assert it.render()
    assert it.stats()["bin_overflow_retries"] > 0
    assert np.array_equal(it.film().view(np.uint32), ref.view(np.uint32))
    it.set_tuning(test_retry=0, sync_free=-1)
    for knobs in (dict(refill_min=8), dict(desc_min=1, leaf_min=1), dict(batch=16), dict(dyn_div=1), dict(refill_min=64, batch=64),
                  dict(sync_free=1), dict(sync_free=1, shade_fused=0), dict(sync_free=0, shade_fused=1)):
        it.set_tuning(**knobs)
        it.clear_film(); assert it.render()
        assert it.stats()["bin_overflow_retries"] == 0
        assert np.array_equal(it.film().view(np.uint32), ref.view(np.uint32)), knobs
    with pytest.raises(mts.MtsGpuError):
        it.set_tuning(no_such_knob=1)


def test_malformed_trees_are_refused(gpu_lib, mts):
    """mtsgpu_upload_scene validates reachability and depth of caller-supplied trees (the kernels' stack has
    trace_stack_levels entries; an unreached node would be laid over the root by the device re-ordering)"""
    sd = mts.scenes.cornell_c1()
    sc = mts.Scene(sd)
    it = mts.MIPathTracer(maxDepth=4)
    L = mts.lib()
    # 1. two inner nodes share their children, the last two nodes are unreachable
    def inner(i, left, axis=0, split=0.0):
        return [((left - i) << 2) | axis, int(np.float32(split).view(np.uint32))]
    leaf = [0x80000000, 0]
    shared = np.array([inner(0, 1), inner(1, 3), inner(2, 3), leaf, leaf, leaf, leaf], dtype=np.uint32)
    broken = mts.abi.Scene.from_buffer_copy(sc.sc)
    broken.kd_nodes = mts.abi.ptr(shared, mts.abi.u32p); broken.n_nodes = len(shared)
    assert L.mtsgpu_upload_scene(it._ctx, C.byref(broken)) == -1
    assert b"child of two nodes" in L.mtsgpu_last_error(it._ctx)
    orphan = np.array([inner(0, 1), leaf, leaf, leaf, leaf], dtype=np.uint32)
    broken.kd_nodes = mts.abi.ptr(orphan, mts.abi.u32p); broken.n_nodes = len(orphan)
    assert L.mtsgpu_upload_scene(it._ctx, C.byref(broken)) == -1
    assert b"unreachable" in L.mtsgpu_last_error(it._ctx)
    # 2. a degenerate chain deeper than the traversal stack: node i = inner(left = i + 1 .. ), 60 levels
    depth = 60
    n = 2 * depth + 1
    chain = np.zeros((n, 2), dtype=np.uint32)
    # layout: node 0 root; children of the k-th inner node at 2k+1 (inner, next level) and 2k+2 (empty leaf)
    for k in range(depth):
        i = 0 if k == 0 else 2 * k - 1
        chain[i, 0] = ((2 * k + 1 - i) << 2) | 0                        # axis 0, relative offset to the left child
        chain[i, 1] = np.float32(0.0).view(np.uint32)
        chain[2 * k + 2, 0] = 0x80000000; chain[2 * k + 2, 1] = 0       # empty leaf
    last = 2 * depth - 1
    chain[last, 0] = 0x80000000; chain[last, 1] = 0
    broken = mts.abi.Scene.from_buffer_copy(sc.sc)
    broken.kd_nodes = mts.abi.ptr(chain, mts.abi.u32p); broken.n_nodes = n
    assert L.mtsgpu_upload_scene(it._ctx, C.byref(broken)) == -1
    assert b"deeper than" in L.mtsgpu_last_error(it._ctx)
    # the intact scene still uploads
    assert L.mtsgpu_upload_scene(it._ctx, sc.ptr) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["c1", "c5_threshold_300", "spheres", "c3_grid60", "c3_grid60_noclip", "c3_grid60_noretract",
                                  "bunny", "fuzz", "c3_1M"])
def test_device_exact_phase_builds_the_same_tree(gpu_lib, mts, orc, case):
    """the exact O(n log n) sweep of the kd-tree build (gkdtree.h:1898-2345: event lists, SAH sweep, classification,
    perfect splits, bad refines, retraction) run level by level on the device yields the host builder's tree bit for
    bit -- nodes, index lists, statistics, bounding box -- with and without the device binning phase above it, and
    therefore the oracle's tree"""
    from test_gpu_parity import _bunny_serialized, _loader
    def params(**kw):
        k = mts.abi.KdParams()
        for a, b in kw.items():
            setattr(k, a, b)
        return k
    kw, sds = {}, []
    if case == "c1":
        sds = [mts.scenes.cornell_c1()]
    elif case == "c5_threshold_300":
        sds = [mts.scenes.cornell_c5(sphere_subdiv=3)]; kw = dict(exact_prim_threshold=300)
    elif case == "spheres":
        sds = [mts.scenes.spheres()]
    elif case.startswith("c3_grid60"):
        sds = [mts.scenes.cornell_c3(grid=60, sphere_subdiv=3)]
        kw = dict(clip=-1) if case.endswith("noclip") else dict(retract=-1) if case.endswith("noretract") else {}
    elif case == "bunny":
        sds = [mts.scenes.bunny(_bunny_serialized(), _loader())]; kw = dict(exact_prim_threshold=20000)
    elif case == "fuzz":
        sds = [mts.scenes.fuzz(seed) for seed in range(8)]
    else:
        sds = [mts.scenes.cornell_c3()]
    for sd in sds:
        host = mts.Scene(sd, kd_params=params(**kw))
        ha = host.arrays()
        for binning in (False, True):
            dev = mts.Scene(sd, kd_params=params(**kw), gpu_binning=binning, gpu_exact=True)
            da = dev.arrays()
            for k in ("kd_nodes", "kd_indices", "aabb_min", "aabb_max", "triaccel"):
                assert np.array_equal(ha[k].view(np.uint32), da[k].view(np.uint32)), (case, k, binning)
            assert host.kdstats() == dev.kdstats()
        if case != "c3_1M":                                                 # the oracle's own builder, same parameters
            oa = orc.FlatScene(sd, kd_params=params(**kw)).arrays()
            assert np.array_equal(oa["kd_nodes"], ha["kd_nodes"]) and np.array_equal(oa["kd_indices"], ha["kd_indices"])


def test_device_random_is_the_reference_mt19937_64(gpu_lib, mts, orc):
    """`Random` (src/libcore/random.cpp:99-227, random.h:82-148) on the device against the reference's known answers
    directly (values measured from the reference, and ISO C++ [rand.predef] for std::mt19937_64, which
    the reference's generator is) -- no oracle in between -- and against the oracle's generator for long sequences"""
    import orc as O
    it = mts.MIPathTracer(maxDepth=4)
    out = it.random_values(0, 10000)                                   # default-constructed Random: seed 5489
    assert int(out[0]) == 14514284786278117030
    assert int(out[9999]) == 9981545732273789042
    assert np.array_equal(out, it.random_values(0, 10000, seed=5489))
    bits = it.random_values(1, 2).astype(np.uint32)
    assert [float(v).hex() for v in bits.view(np.float32)] == ["0x1.eded5c0000000p-1", "0x1.17901c0000000p-1"]
    assert int(it.random_values(0, 1, clone=1)[0]) == 13719712115898985683      # Random(Random *) clone #0 of a fresh parent
    assert it.random_values(3, 8).tolist() == [7, 3, 1, 5, 2, 0, 4, 6]        # Random::shuffle of 0..7
    # the oracle's generator (pinned by the same numbers in tests/test_oracle_kats.py): other seeds, nextSize, clones
    L = orc.lib()
    for seed in (1, 0x5EED, 2 ** 63 + 12345):
        r = O.Random(); r.mti = 313
        L.orc_random_seed(C.byref(r), seed)
        ref = [L.orc_random_next_ulong(C.byref(r)) for _ in range(1000)]
        assert it.random_values(0, 1000, seed=seed).tolist() == ref
    for n in (1, 2, 3, 1000, 2 ** 32 + 1):
        r = O.Random(); r.mti = 313
        ref = [L.orc_random_next_size(C.byref(r), n) for _ in range(300)]
        assert it.random_values(2, 300, arg=n).tolist() == ref
    parent = O.Random(); parent.mti = 313
    for clone in (1, 2, 3):
        child = O.Random(); child.mti = 313
        L.orc_random_seed_from(C.byref(child), C.byref(parent))
        ref = [L.orc_random_next_ulong(C.byref(child)) for _ in range(500)]
        assert it.random_values(0, 500, clone=clone).tolist() == ref
    r = O.Random(); r.mti = 313
    a = np.arange(1000, dtype=np.uint32)
    L.orc_random_shuffle_u32(C.byref(r), a.ctypes.data_as(C.POINTER(C.c_uint32)), 1000)
    assert it.random_values(3, 1000).tolist() == a.tolist()
