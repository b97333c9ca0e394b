"""GPU tests, fourth set: the measurement hooks added in round 4 (issued-request counters, replay roof) and the
pipeline changes of the round, each against the oracle or against an invariant of the traversal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_issued_request_counters_add_up(gpu_lib, mts):
    """Counting build: every inner-node step fetches exactly one sibling pair (from global memory or from the LDS copy of
    the top of the tree), every pop fetches the popped node and -- unless the stack entry is the sentinel -- its parent
    (n_inner itself is checked against the oracle's count in test_traversal_counters_match_oracle)"""
    sd = mts.scenes.cornell_c3(grid=48, sphere_subdiv=3)
    scene = mts.Scene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, 96, 96)
    it = mts.MIPathTracer(maxDepth=8)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=4, seed=3)
    it.set_options(count_traversal=True)
    assert it.render()
    st = it.stats()
    rays = st["rays_closest"] + st["rays_shadow"]
    assert rays > 0 and st["n_inner"] > 0
    assert st["req_pair_global"] + st["req_pair_lds"] == st["n_inner"]
    pops_max = st["n_leaf"]                      # a ray pops at most once per leaf it visits
    nodes = st["req_node_global"] + st["req_node_lds"]
    assert 0 < nodes <= 2 * pops_max + rays      # + the root, fetched once per ray
    assert st["req_tail"] <= 2 * st["n_idx"] and st["req_head"] >= st["n_idx"]


def test_replay_roof_replays_what_the_kernel_asked_for(gpu_lib, mts):
    """mtsgpu_replay_roof, all three classes of launches: the recorded lists hold exactly the gathers the counters saw
    (pairs + nodes from global memory, heads, tails -- rays, ids and hits move in queue order and are not part of the
    lists), and both timings are positive"""
    sd = mts.scenes.cornell_c3(grid=48, sphere_subdiv=3)
    scene = mts.Scene(sd)
    cam = mts.PerspectiveCamera.for_description(sd, 128, 128)
    it = mts.MIPathTracer(maxDepth=8)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=8, seed=3)
    it.set_tuning(sync_free=0)                # host-driven bounces: the shadow class needs the size of the shadow queue
    assert it.render()
    film = it.film().copy()
    n = 128 * 128 * 8 // 2
    for kind in ("deep", "shadow", "camera"):
        nk = n // 4 if kind == "shadow" else n
        rr = it.replay_roof(nk, stride=2, reps=1, kind=kind)
        assert rr["rays"] == nk and rr["truncated_rays"] == 0, kind
        assert rr["requests"] == rr["pair_global"] + rr["node_global"] + rr["heads"] + rr["tails"], (kind, rr)
        assert rr["product_ms"] > 0 and rr["replay_ms"] > 0
    # camera rays start at the root: every one of them reads the LDS copy of the top of the tree
    assert rr["pair_lds"] >= nk
    # once the camera rays have been generated again the pass is gone: no second sample of it
    with pytest.raises(mts.MtsGpuError):
        it.replay_roof(n, stride=2, reps=1, kind="camera")
    with pytest.raises(mts.MtsGpuError):
        it.replay_roof(n // 4, stride=2, reps=1, kind="shadow")
    # the measurement leaves the renderer usable: the next frame is the same film
    it.clear_film()
    assert it.render()
    assert np.array_equal(it.film().view(np.uint32), film.view(np.uint32))
    with pytest.raises(mts.MtsGpuError):
        it.replay_roof(1 << 30, stride=4)


def _bench(args, timeout=900):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])


def test_bench_line_carries_the_group_form_and_the_replay_roof(gpu_lib, mts, tmp_path):
    """bench.py on one GPU, small frame: the JSON line has the one-process device-group timing (`group`, rendered by a
    fresh child process after the ranks have finished) with a film equal to the unsharded render bit for bit, the
    replay roof with frac <= 1, and a CPU baseline that says what an ungated host would extrapolate to"""
    out = str(tmp_path / "group.npy")
    rec = _bench(["--res", "256", "--grid", "48", "--spp", "16", "--steps", "1", "--warmup", "1", "--host-kd",
                  "--no-1spp", "--no-cpu-baseline", "--group", "--dump-group-film", out])
    g = rec["group"]
    assert "error" not in g, g
    assert g["group_ms_per_step"] > 0 and g["value"] > 0 and g["devices"] == [0]
    rq = rec["roofline_requests"]
    # (on the full-size frame the replay takes 0.86 of the kernel's time; a 1 M-ray sample is a launch of three rounds,
    # where both kernels are mostly ramp and tail, so only the plumbing is asserted here)
    assert rq["note"] is None and set(rq["classes"]) == {"camera", "deep", "shadow"}, rq
    for kind, cl in rq["classes"].items():
        if kind == "shadow":           # a frame this small runs device-driven bounces, which leave no shadow-queue size behind
            assert "host-driven passes only" in cl["error"] and rq["replay_ms_per_step"] is None, rq
            continue
        assert 0 < cl["replay_ratio"] < 2.5 and cl["product_ms"] > 0 and cl["replay_ms"] > 0 and cl["launches_ms_per_step"] > 0, rq
    parts = sum(cl["launches_ms_per_step"] for cl in rq["classes"].values())
    assert abs(parts - rq["trace_ms_per_step"]) < 1e-6 * max(1.0, parts)
    assert rq["issued_requests_per_ray"] > 0 and rq["lds_served_requests_per_ray"] > 0
    assert rec["roofline"]["frac"] > 0 and rec["roofline_shade"]["frac"] > 0
    sd = mts.scenes.cornell_c3(grid=48, sphere_subdiv=5)
    it = mts.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
    it.preprocess(mts.Scene(sd), mts.PerspectiveCamera.for_description(sd, 256, 256), sampler="ldsampler", sampleCount=16, seed=0x5EED)
    assert it.render()
    assert np.array_equal(np.load(out).view(np.uint32), it.film().view(np.uint32))


def test_bench_over_two_gpus_uses_the_nccl_backend(gpu_lib, mts, tmp_path):
    """`bench.py --gpus 2` as the driver starts it (no --devices): the ranks form an RCCL process group, rank 0's reduced film
    equals the unsharded render, and the group child renders the same film through mtsgpu_group_render over both GPUs.
    Needs two GPUs: skipped on the one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the nccl branch of bench.py needs two")
    out, gout = str(tmp_path / "film.npy"), str(tmp_path / "group.npy")
    rec = _bench(["--gpus", "2", "--res", "128", "--grid", "24", "--spp", "4", "--steps", "1", "--warmup", "0", "--host-kd",
                  "--no-1spp", "--no-cpu-baseline", "--c4-spp", "64", "--c4-steps", "1", "--dump-film", out, "--dump-group-film", gout])
    assert rec["n_gpus"] == 2 and len(rec["rank_ms"]) == 2
    assert rec["rccl_ranks"] == 2 and "rccl" in rec["reduce_kind"]           # RCCL's first collective summed one per rank
    assert rec["c4_strong"]["value"] > 0 and rec["c4_strong"]["n_gpus"] == 2
    assert rec["group"].get("rccl_ranks") in (0, 2)                          # 2 unless the group fell back to the ordered sum (it says why)
    assert "error" not in rec["group"], rec["group"]
    sd = mts.scenes.cornell_c3(grid=24, sphere_subdiv=5)
    it = mts.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
    it.preprocess(mts.Scene(sd), mts.PerspectiveCamera.for_description(sd, 128, 128), sampler="ldsampler", sampleCount=8, seed=0x5EED)
    assert it.render()
    assert np.array_equal(np.load(out).view(np.uint32), it.film().view(np.uint32))
    assert np.array_equal(np.load(gout).view(np.uint32), it.film().view(np.uint32))


def test_group_tuning_reaches_every_member(gpu_lib, mts):
    """mtsgpu_group_set_tuning forwards a knob to all members (the film does not change), refuses an unknown one, and keeps
    its own test knob (`rccl_fail`) to itself"""
    sd = mts.scenes.cornell_c1()
    scene = mts.Scene(sd); cam = mts.PerspectiveCamera.for_description(sd, 64, 48)
    g = mts.DeviceGroup([0, 0], maxDepth=4)
    g.preprocess(scene, cam, sampler="independent", sampleCount=4)
    assert g.render()
    ref = g.film()
    g.set_tuning(sync_free=0, refill_min=8)
    assert g.render() and np.array_equal(g.film().view(np.uint32), ref.view(np.uint32))
    g.set_tuning(rccl_fail=1); g.set_tuning(rccl_fail=0)
    with pytest.raises(mts.MtsGpuError):
        g.set_tuning(no_such_knob=1)
