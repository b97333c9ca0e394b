"""GPU tests, fifth set: the pipeline changes of round 5, each against the oracle or against the unchanged path."""
import numpy as np
import pytest

from test_gpu_parity import _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("spp", [1024, 2048, 4096, 32768])
def test_ld_tables_above_512_samples(gpu_lib, mts, orc, spp):
    """LowDiscrepancySampler::generate() (ldsampler.cpp:125-158) above 512 samples per pixel: one wave per pixel finds the
    draws Random::nextSize accepts (random.cpp:196-215), one lane per table applies the swaps of Random::shuffle
    (random.h:145-148) -- in LDS, a wave per table, up to 16 384 samples; in memory, eight steps at a time, above -- scrambles
    and permutations equal the oracle's sequential loop"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c1", sampler="ldsampler", spp=spp)
    for key in (0, 7, 65535, 123456):
        t1, t2 = it.ld_tables(key, spp, 3)
        e1 = np.zeros((3, spp), dtype=np.float32)
        e2 = np.zeros((3, spp, 2), dtype=np.float32)
        orc.lib().orc_ld_generate_keyed(0x5EED, key, spp, 3, mts.abi.ptr(e1, mts.abi.f32p), mts.abi.ptr(e2, mts.abi.f32p))
        assert np.array_equal(t1.view(np.uint32), e1.view(np.uint32))
        assert np.array_equal(t2.view(np.uint32), e2.view(np.uint32))


@pytest.mark.parametrize("sampler", ["ldsampler", "stratified"])
def test_films_at_1024_samples_in_one_and_in_many_passes(gpu_lib, mts, orc, sampler):
    """the tables of 192 pixels (three waves of k_ld_apply per table, the last one ragged) and of ragged passes"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c5_small", W=16, H=12, sampler=sampler, spp=1024, max_depth=6)
    ofilm, _ = orc.render(oscene.scene, ocam, op)
    assert it.render()
    assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32))
    for max_paths in (1024 * 100, 1024 * 7):          # 2 and 28 passes (the last ones ragged)
        it.set_options(max_paths=max_paths)
        it.clear_film(); assert it.render()
        assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32)), max_paths


def _run_bench(args, timeout=900):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])


def _unsharded_c3(mts, res, grid, spp_total):
    sd = mts.scenes.cornell_c3(grid=grid, sphere_subdiv=5)
    it = mts.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
    it.preprocess(mts.Scene(sd), mts.PerspectiveCamera.for_description(sd, res, res), sampler="ldsampler", sampleCount=spp_total, seed=0x5EED)
    assert it.render()
    return it.film()


def test_five_ranks_on_one_gpu(gpu_lib, mts, tmp_path):
    """bench.py's process-per-GPU form at a world size above two: five ranks (the GPU boxes allow six processes on a card,
    this test runner is one of them) share GPU 0, tiles morton(tx, ty) % 5 (imageproc.cpp:43-78), the films summed on
    rank 0 (renderproc.cpp:123-130; gloo on host copies, because ranks on one device cannot form an RCCL communicator):
    every rank reports its times and the reduced film equals the unsharded render bit for bit"""
    out = str(tmp_path / "film5.npy")
    rec = _run_bench(["--gpus", "5", "--devices", "0,0,0,0,0", "--res", "160", "--grid", "24", "--spp-total", "8", "--steps", "1",
                      "--warmup", "0", "--no-cpu-baseline", "--no-1spp", "--host-kd", "--no-group", "--max-paths", str(1 << 18),
                      "--dump-film", out])
    assert rec["n_gpus"] == 5 and rec["scaling"] == "strong" and rec["value"] > 0
    assert len(rec["rank_ms"]) == 5 and len(rec["reduce_ms"]) == 5 and min(rec["rank_ms"]) > 0
    assert np.array_equal(np.load(out).view(np.uint32), _unsharded_c3(mts, 160, 24, 8).view(np.uint32))


def test_device_group_of_eight_members_on_one_gpu(gpu_lib, mts, tmp_path):
    """the drop-in's own multi-GPU form at the node's size: mtsgpu_create_multi over eight members (all on GPU 0 here),
    mtsgpu_group_render gives member i the tiles of part i of 8 and sums the films in member order -- one process, eight
    contexts, eight host threads; the film equals the unsharded render bit for bit.  With eight distinct GPUs the only
    new thing is the RCCL reduce itself"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "film8.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--group-child", "8", "--devices", "0,0,0,0,0,0,0,0", "--res", "256",
                        "--grid", "24", "--spp-total", "8", "--steps", "1", "--host-kd", "--dump-film", out],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    g = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert g["devices"] == [0] * 8 and g["group_ms_per_step"] > 0
    assert g["reduce_kind"] == "ordered peer-copy sum", g
    assert np.array_equal(np.load(out).view(np.uint32), _unsharded_c3(mts, 256, 24, 8).view(np.uint32))


def test_fused_shading_launch_equals_one_launch_per_bsdf_type(gpu_lib, mts, orc):
    """k_shade_all (device-driven bounces: every material queue of a bounce in ONE launch) against one k_shade launch per
    BSDF type, on the scene with the most BSDF types, both against the oracle"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "next_rows", W=48, H=48, sampler="ldsampler", spp=8)
    ofilm, _ = orc.render(oscene.scene, ocam, op)
    for fused in (1, 0):
        it.set_tuning(sync_free=1, shade_fused=fused)
        it.clear_film(); assert it.render()
        assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32)), fused


def test_rccl_process_group_of_one_rank(gpu_lib, mts, tmp_path):
    """what bench.py does first with more than one GPU, as far as one GPU allows: torch.distributed over backend nccl (= RCCL)
    forms a group, the probing all-reduce and the film reduce (filmreduce.reduce_film on a 1024 x 1024 x 5 tensor) run on the
    device; a fresh process, because a process group is per process"""
    import os, subprocess, sys, socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import datetime, os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import _pkgload
pkg = _pkgload.load()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=torch.device("cuda", 0))
probe = torch.ones(1, device="cuda"); dist.all_reduce(probe); torch.cuda.synchronize()
assert int(probe.item()) == 1
film = torch.arange(1024 * 1024 * 5, dtype=torch.float32, device="cuda").reshape(1024, 1024, 5)
ref = film.clone()
dist.reduce(film, dst=0, op=dist.ReduceOp.SUM); torch.cuda.synchronize()      # what reduce_film issues with more than one rank
assert torch.equal(film, ref)
assert pkg.filmreduce.reduce_film(film, dst=0) is film
dist.barrier(); dist.destroy_process_group()
print("rccl ok", dist.Backend.NCCL)
''' % root
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"rccl ok" in r.stdout, r.stderr.decode()[-2000:]


@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_random_scenes_whoever_drives_the_bounces(gpu_lib, mts, orc, seed):
    """random scenes (every BSDF type, mixed luminaires, spheres) rendered with each way of driving the bounces of a pass --
    the device (one shading launch for all material queues, or one per BSDF type), the host (read-back per bounce, dynamically
    claimed batches), the host over three ragged passes: the film equals the oracle's bit for bit every time
    (tools/fuzz_parity.py sweeps the same over thousands of seeds: profiles/r05w_*)"""
    sd = mts.scenes.fuzz(seed, n_meshes=6 + seed % 25)
    scene = mts.Scene(sd); oscene = orc.FlatScene(sd)
    W, H, spp = 40, 30, 8
    sampler = ["independent", "ldsampler", "halton"][seed % 3]
    kind = {"independent": 0, "ldsampler": 1, "halton": 2}[sampler]
    cam = mts.PerspectiveCamera.for_description(sd, W, H); ocam = orc.make_camera(sd, W, H)
    op = orc.render_params(sd.max_depth, rr_depth=sd.rr_depth, strict_normals=0, sampler=kind, spp=spp, seed=seed)
    ofilm, _ = orc.render(oscene.scene, ocam, op)
    it = mts.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
    it.preprocess(scene, cam, sampler=sampler, sampleCount=spp, seed=seed)
    for name, tuning, max_paths in (("device", dict(sync_free=1, shade_fused=1), 0), ("device, per type", dict(sync_free=1, shade_fused=0), 0),
                                    ("host", dict(sync_free=0), 0), ("host, three passes", dict(sync_free=0), spp * (W * H // 3 + 1))):
        it.set_tuning(**tuning); it.set_options(max_paths=max_paths)
        it.clear_film(); assert it.render()
        assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32)), name
