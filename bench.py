#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on N MI355X GPUs of one node.

Workload (BASELINE.json configs[2]/[3], SURVEY.md 8d "C3/C4"): procedural Cornell variant with one
1 024 000-triangle displaced walls TriMesh (lambertian), a 20 480-triangle dielectric icosphere and
an area luminaire; `path` integrator maxDepth=16, rrDepth=10; 1024x1024, low-discrepancy sampler.
A "step" is one full-frame render.  Every GPU renders its share of the ImageBlock tiles (Morton index of the
tile modulo N, mtsgpu_set_tiles) into a full-frame film; the films are summed once per step with a reduce over
RCCL/xGMI (Film::putImageBlock).  Scene upload is excluded, the film reduce is included.
value = total camera samples / wall time.

Two scaling modes:
  weak   (default)         every GPU renders 64 spp worth of samples: the frame has 64*N spp, per-GPU work constant
  strong (--spp-total S)   the frame has S spp whatever N is (BASELINE.json configs[3] "C4": --spp-total 4096)

`python bench.py --gpus N` needs no launcher: with N > 1 and no WORLD_SIZE in the environment it starts N
worker processes itself (before anything touches a GPU) and prints rank 0's JSON line.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is a worker right away.

One JSON line on rank 0 (see DESIGN.md section 8 for the definition of every field)."""
import argparse
import datetime
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def kernel_source_hash():
    """identifies the kernels a committed PMC traffic record belongs to"""
    h = hashlib.sha256()
    for f in ("trace.hip", "kernels.h", "kdevice.h", "devmath.h"):      # what k_trace is compiled from
        h.update(open(os.path.join(ROOT, "mitsuba-renderer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(st):
    """B_ray = 8 n_inner + 8 n_leaf + 4 n_idx + 48 n_tri + 48 (ray in 32 B + hit out 16 B),
    n_tri counted without mailbox credit (= index entries visited)."""
    rays = st["rays_closest"] + st["rays_shadow"]
    return 8 * st["n_inner"] + 8 * st["n_leaf"] + 4 * st["n_idx"] + 48 * st["n_idx"] + 48 * rays


def issued_requests(st):
    """Lane-level vector-memory GATHERS the traversal kernels issued in the counting step (mtsgpu_stats.req_*): sibling
    pairs and single nodes that were not served by the LDS copy of the top of the tree, one record head per index entry,
    the record tails, stack spills.  Rays, queue ids, binned ids + hits and shadow rays move in queue order (streams: about
    five 4- to 16-byte accesses per closest-hit ray, three per any-hit ray) and are not counted here."""
    return st["req_pair_global"] + st["req_node_global"] + st["req_head"] + st["req_tail"] + st["req_spill"]


def shade_algorithmic_bytes(st):
    """k_shade, DESIGN.md section 6: per shaded path-bounce the state it must read and write (ray 32 B, hit 16 B,
    throughput + depth 16 B, Li + flags 16 B, bsdfVal + pdf 16 B, sampler state 16 B -> 112 B read; the same minus
    the hit written back: 96 B), the 48-byte triangle record of the hit (84 B with vertex normals, counted as 48),
    and 48 B per shadow ray appended to the shadow queue."""
    return st["rays_closest"] * (112 + 96 + 48) + st["rays_shadow"] * 48


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes show 256
    CPUs and grant 16: cpu.max = '1600000 100000')"""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, len(os.sched_getaffinity(0)), quota


def cpu_baseline(pkg, sd, res, spp, max_depth, seconds=12.0):
    """The oracle (CPU restatement, kind 'port') on a bounded centre crop of the same frame: one thread, then all
    usable cores (OpenMP), built with -O3 -march=native on this machine (SURVEY.md 8d "CPU reference timing (ii)").
    Test infrastructure used as a reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    flags = "-O3 -march=native -ffp-contract=off"
    try:
        orc.use_native_build()
        orc.lib()
    except Exception as e:                      # no compiler on the box: the portable -O2 build
        orc._variant = "liboracle.so"
        flags = "-O2 -ffp-contract=off (native build failed: %s)" % type(e).__name__
    cores, visible, quota = usable_cpus()
    oscene = orc.FlatScene(sd)
    cam = orc.make_camera(sd, res, res)
    c = res // 2

    def timed(side, threads):
        prm = orc.render_params(max_depth, sampler=pkg.abi.SAMPLER_LD_KEYED, spp=spp, seed=0x5EED, n_threads=threads)
        h = side // 2
        t0 = time.perf_counter()
        _, st = orc.render(oscene.scene, cam, prm, rect=(c - h, c - h, c + h, c + h))
        dt = time.perf_counter() - t0
        return side * side * spp / dt, (st.rays_closest + st.rays_shadow) / dt, dt

    def side_for(rate, secs):
        s = int(min(res, max(8, (rate * secs / spp) ** 0.5)))
        return s - s % 2

    rate1, _, _ = timed(8, 1)                                           # calibration
    s1 = side_for(rate1, seconds / 3)
    rate1, rays1, dt1 = timed(s1, 1)
    sN = side_for(rate1 * cores * 0.8, seconds)
    rateN, raysN, dtN = timed(sN, cores)
    return {"value": rateN / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "threads": cores, "visible_cpus": visible, "cgroup_cpu_quota": quota,
            "one_thread_msamples_per_s": rate1 / 1e6,
            "mrays_per_s_per_thread": rays1 / 1e6,
            "mrays_per_s_per_thread_all_threads": raysN / 1e6 / cores,
            "scaling_efficiency": rateN / (rate1 * cores),
            # NOT measured: the same host without the cgroup quota -- every visible CPU at the per-thread rate and
            # parallel efficiency measured on the granted ones -- so that nobody reads value as "one whole host"
            "extrapolated_all_visible_cpus": {"value": rateN / cores * visible / 1e6, "unit": "Msamples/s", "cpus": visible,
                                              "note": "measured %d-thread rate x %d / %d; an extrapolation, not a measurement" % (cores, visible, cores)},
            "build": flags,
            "sample": "centre crops of the %dx%d frame x %d spp: %dx%d pixels on 1 thread (%.1f s), %dx%d pixels on %d threads "
                      "(%d camera samples, %.1f s); oracle/liboracle_native.so, OpenMP over pixels"
                      % (res, res, spp, s1, s1, dt1, sN, sN, cores, sN * sN * spp, dtN)}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_workers(n):
    """`python bench.py --gpus N` without a launcher: N fresh worker processes (this parent never touches a GPU and
    never execs), one per GPU, rank 0 prints the JSON line on our stdout."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:           # one rank failed: the others would wait in the rendezvous forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def group_child(args):
    """--group-child N: the drop-in's own multi-GPU form, timed in a fresh process.  A Mitsuba process calls the
    integrator once (src/librender/scene.cpp:356-359), so the plugin drives all GPUs from ONE process through
    mtsgpu_create_multi / mtsgpu_group_render (one host thread per GPU, the films summed by one ncclReduce through a
    dlopen'ed librccl, or by ordered peer copies; renderproc.cpp:123-130 is the merge being replaced).  Same workload and
    sharding as the process-per-GPU form above; prints one JSON object."""
    import _pkgload
    pkg = _pkgload.load()
    n = args.group_child
    devices = [int(x) for x in args.devices.split(",")][:n] if args.devices else list(range(n))
    sd = pkg.scenes.cornell_c3(grid=args.grid, sphere_subdiv=5)
    n_tris = 5 * 2 * args.grid * args.grid + 20480
    scene = pkg.Scene(sd, None, gpu_binning=not args.host_kd, gpu_exact=(not args.host_kd) and n_tris > 2_000_000)
    spp_total = args.spp_total if args.spp_total > 0 else args.spp * n
    cam = pkg.PerspectiveCamera.for_description(sd, args.res, args.res)
    g = pkg.DeviceGroup(devices, maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
    g.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp_total, seed=0x5EED)
    assert g.render()                           # warm-up: buffers, RCCL communicators
    times = []
    for _ in range(max(1, args.steps)):
        t0 = time.perf_counter()
        assert g.render()                       # returns with the reduced film in member 0
        times.append((time.perf_counter() - t0) * 1e3)
    ms = sum(times) / len(times)
    out = {"form": "one process, %d GPUs: mtsgpu_create_multi + mtsgpu_group_render" % n, "devices": devices,
           "group_ms_per_step": ms, "group_ms_best": min(times), "steps": len(times),
           "value": args.res * args.res * spp_total / (ms * 1e-3) / 1e6, "unit": "Msamples/s",
           "reduce_kind": g.reduce_kind(), "reduce_note": g.reduce_note(),
           "rccl_ranks": g.rccl_ranks()}            # ranks of the communicator that passed the library's self-check (0: ordered sum)
    if args.dump_film:
        import numpy as np
        np.save(args.dump_film, g.film())
    print(json.dumps(out), flush=True)


def run_group_child(args, world):
    """rank 0, after every rank has finished: the group form in a fresh child process (never exec; 120 s; a failure
    leaves a note instead of the numbers)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                             "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.abspath(__file__), "--group-child", str(world), "--steps", "3", "--res", str(args.res),
           "--spp", str(args.spp), "--spp-total", str(args.spp_total), "--grid", str(args.grid)]
    if args.host_kd:
        cmd.append("--host-kd")
    if args.devices:
        cmd += ["--devices", args.devices]
    if args.dump_group_film:
        cmd += ["--dump-film", args.dump_group_film]
    # the result line of the ranks must not depend on this child: its output goes to files, it gets 120 s, and a child
    # that does not die within 5 s of being killed is left behind rather than waited for
    import tempfile
    try:
        with tempfile.TemporaryFile() as fo, tempfile.TemporaryFile() as fe:
            p = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe)
            deadline = time.time() + 120
            while p.poll() is None and time.time() < deadline:
                time.sleep(0.1)
            if p.poll() is None:
                p.kill()
                try:
                    p.wait(timeout=5)
                except subprocess.TimeoutExpired:
                    pass
                return {"error": "no result within 120 s"}
            fo.seek(0); fe.seek(0)
            lines = [l for l in fo.read().decode(errors="replace").splitlines() if l.startswith("{")]
            if p.returncode == 0 and lines:
                return json.loads(lines[-1])
            return {"error": "exit code %d: %s" % (p.returncode, fe.read().decode(errors="replace")[-400:])}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=64, help="weak scaling: samples per pixel per GPU (the frame has spp * N)")
    ap.add_argument("--spp-total", type=int, default=0, help="strong scaling: samples per pixel of the frame, whatever N is (C4: 4096)")
    ap.add_argument("--grid", type=int, default=320, help="wall grid resolution (320 -> 1 024 000 triangles)")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-1spp", action="store_true", help="skip the time-to-1spp frames")
    ap.add_argument("--no-c4", action="store_true", help="with --gpus > 1 in weak mode: skip the C4 strong-scaling frames (4096 spp over all GPUs) that are timed after the weak steps")
    ap.add_argument("--c4-spp", type=int, default=4096, help="samples per pixel of the C4 strong-scaling frame (BASELINE.json configs[3]: 4096)")
    ap.add_argument("--c4-steps", type=int, default=2)
    ap.add_argument("--host-kd", action="store_true", help="kd-tree build (binning and exact phase) on the host instead of the GPU (same tree)")
    ap.add_argument("--devices", default="", help="comma list: HIP device of each local rank (default: LOCAL_RANK). "
                    "Ranks sharing a device reduce their films through gloo on host copies (test mode)")
    ap.add_argument("--dump-film", default="", help="rank 0 writes the reduced film of the last step to this .npy file")
    ap.add_argument("--group-child", type=int, default=0, help="(internal) time the one-process device group over N GPUs and print its JSON")
    ap.add_argument("--no-group", action="store_true", help="skip the one-process device-group timing")
    ap.add_argument("--group", action="store_true", help="time the one-process device group with one GPU too (default: only with --gpus > 1)")
    ap.add_argument("--dump-group-film", default="", help="the group child writes its reduced film to this .npy file")
    args = ap.parse_args()

    if args.group_child > 0:
        return group_child(args)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_workers(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    import _pkgload
    pkg = _pkgload.load()
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(world))
    if len(devices) < world:
        sys.exit("--devices names %d devices for %d ranks" % (len(devices), world))
    device = devices[local_rank]
    shared_device = len(set(devices[:world])) < world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X; libmtsgpu has no CPU path")
    if device >= torch.cuda.device_count():
        sys.exit("rank %d: device %d requested, %d visible" % (rank, device, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    collective_ranks = 0
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if shared_device else "nccl"
        try:
            # a rendezvous or communicator that cannot form must end the run with a reason, not hang it
            kw = {} if shared_device else {"device_id": torch.device("cuda", device)}
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120), **kw)
            probe = torch.ones(1, device="cpu" if shared_device else "cuda")
            dist.all_reduce(probe)                       # first collective: RCCL builds its rings here
            if not shared_device:
                torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("all_reduce of ones over %d ranks returned %s" % (world, probe.item()))
            collective_ranks = int(probe.item())         # what the backend's first collective summed up: one per rank
        except Exception as e:
            sys.stderr.write("bench.py rank %d/%d (device %d): torch.distributed backend %s failed: %s: %s\n"
                             % (rank, world, device, backend, type(e).__name__, e))
            sys.stderr.flush()
            os._exit(3)

    # --- scene (host side: generate, flatten, upload; not timed) ---
    sd = pkg.scenes.cornell_c3(grid=args.grid, sphere_subdiv=5)
    t0 = time.time()
    # kd-tree build: the binning phase on the device; the exact phase there too from 2 M triangles up (below that its
    # per-level launches cost more than the host's job pool: 0.157 s against 0.120 s at 1 M, profiles/r02e_kdbuild_bench.txt).
    # The tree is the same bit for bit either way.
    n_tris = 5 * 2 * args.grid * args.grid + 20480
    scene = pkg.Scene(sd, None, gpu_binning=not args.host_kd, gpu_exact=(not args.host_kd) and n_tris > 2_000_000)
    flatten_s = time.time() - t0
    W = H = args.res
    strong = args.spp_total > 0
    spp_total = args.spp_total if strong else args.spp * world
    cam = pkg.PerspectiveCamera.for_description(sd, W, H)
    it = pkg.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth, device=device)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp_total, seed=0x5EED)
    it.set_tiles(32, rank, world)
    film = torch.zeros((H, W, 5), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    it.set_stream(stream.cuda_stream)
    it.set_film_buffer(film.data_ptr(), keepalive=film)
    host_film = torch.zeros((H, W, 5), dtype=torch.float32).pin_memory() if (world > 1 and shared_device) else None

    split = {"render_s": 0.0, "reduce_s": 0.0}     # this rank's time inside the steps: its own frame, the film reduce

    def step():
        t_a = time.perf_counter()
        film.zero_()
        if not it.render():                     # returns with the stream idle
            raise RuntimeError("render cancelled")
        t_b = time.perf_counter()
        if world > 1:
            if shared_device:                   # test mode: two ranks on one GPU cannot form an RCCL communicator
                host_film.copy_(film)
                pkg.filmreduce.reduce_film(host_film, dst=0)
                if rank == 0:
                    film.copy_(host_film)
            else:
                pkg.filmreduce.reduce_film(film, dst=0)
            torch.cuda.synchronize()            # the next step's film.zero_() would wait for the reduce anyway
        split["render_s"] += t_b - t_a
        split["reduce_s"] += time.perf_counter() - t_b

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # --- untimed: warmup, one counting step (algorithmic work of the traversal kernels), and KT steps with a HIP-event
    # pair around every launch on the library's streams: the per-kernel durations of the roofline blocks.  The timed
    # steps below run WITHOUT those events (about 130 event pairs per frame are not part of the hot path).
    it.set_options(max_paths=args.max_paths, count_traversal=False, time_kernels=False)
    for _ in range(args.warmup):
        step()
    it.set_options(max_paths=args.max_paths, count_traversal=True, time_kernels=False)
    step()
    torch.cuda.synchronize()
    counts = it.stats()
    bytes_per_step = algorithmic_bytes(counts)
    it.set_options(max_paths=args.max_paths, count_traversal=False, time_kernels=True)
    trace_ms = trace_first_ms = trace_shadow_ms = 0.0
    trace_launches = 0
    shade_ms = 0.0
    kt_steps = 2
    for _ in range(kt_steps):
        step()
        st = it.stats()                     # per-launch HIP-event durations on the library's streams
        trace_ms += st["trace_ms"]; shade_ms += st["shade_ms"]; trace_launches += st["trace_launches"]
        trace_first_ms += st["trace_first_ms"]; trace_shadow_ms += st["trace_shadow_ms"]
    trace_ms /= kt_steps; shade_ms /= kt_steps; trace_launches /= kt_steps      # per step from here on
    trace_first_ms /= kt_steps; trace_shadow_ms /= kt_steps
    it.set_options(max_paths=args.max_paths, count_traversal=False, time_kernels=False)

    # --- timed region: exactly K steps between barriers ---
    fence()
    split["render_s"] = split["reduce_s"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    rank_ms = [split["render_s"] / args.steps * 1e3]
    reduce_ms = [split["reduce_s"] / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own render time and the time it spent in the reduce (waiting for the slowest rank included):
        # one SCALE run then tells load imbalance from collective cost
        mine = torch.tensor([rank_ms[0], reduce_ms[0]], dtype=torch.float64, device="cpu" if shared_device else "cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(e[0].item()) for e in every]
        reduce_ms = [float(e[1].item()) for e in every]
    if args.dump_film and rank == 0:
        np.save(args.dump_film, film.cpu().numpy())

    # --- the traversal kernels against replays of their own request streams: samples of the frame's own rays, one per class
    # of launch (untimed; rank 0).  Deep rays first: the camera class generates the last pass's camera rays again. ---
    replay_raw = {}
    replay_note = None
    r_stride = 1
    if rank == 0:
        try:
            avail = min(W * H * spp_total // world, 72 << 20)    # paths of one pass
            r_stride = 8 if avail >= (32 << 20) else 1
            r_n = min(8 << 20, avail // r_stride)          # 8 M rays: twice the size from which the early loop exits are on
            if r_n >= 65536:
                for kind in ("deep", "shadow", "camera"):
                    n_k = r_n // 4 if kind == "shadow" else r_n      # about a third of the paths queue a shadow ray per bounce
                    try:
                        replay_raw[kind] = it.replay_roof(n_k, r_stride, reps=3, kind=kind)
                    except pkg.MtsGpuError as e:                     # e.g. device-driven passes leave no shadow-queue size behind
                        replay_raw[kind] = {"error": str(e)}
            else:
                replay_note = "frame too small for a replay sample"
        except Exception as e:
            replay_note = "%s: %s" % (type(e).__name__, e)

    # --- time to a 1-spp frame (second half of BASELINE.json's metric), untimed region ---
    one_spp_ms = None
    if not args.no_1spp:
        it.preprocess(scene, cam, sampler="ldsampler", sampleCount=1, seed=0x5EED)
        it.set_tiles(32, rank, world)
        it.set_options(max_paths=args.max_paths, count_traversal=False, time_kernels=False)
        step(); fence()
        best = 1e30
        for _ in range(3):
            t1 = time.perf_counter()
            step(); fence()
            best = min(best, (time.perf_counter() - t1) * 1e3)
        one_spp_ms = best

    # --- BASELINE.json configs[3] next to the weak line: with N > 1 GPUs the same scene at 4096 spp per FRAME (strong
    # scaling: every GPU renders its tiles at 4096 spp, one film reduce per frame), timed like the steps above ---
    c4 = None
    if world > 1 and not strong and not args.no_c4:
        it.preprocess(scene, cam, sampler="ldsampler", sampleCount=args.c4_spp, seed=0x5EED)
        it.set_tiles(32, rank, world)
        it.set_options(max_paths=args.max_paths, count_traversal=False, time_kernels=False)
        step(); fence()                              # warm-up: the 4096-spp tables and passes
        t1 = time.perf_counter()
        for _ in range(max(1, args.c4_steps)):
            step()
        fence()
        c4_elapsed = time.perf_counter() - t1
        t = torch.tensor([c4_elapsed], dtype=torch.float64, device="cpu" if shared_device else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        c4_elapsed = float(t.item())
        n_c4 = max(1, args.c4_steps)
        c4 = {"config": "BASELINE.json configs[3]: the 1M-triangle scene at %d spp per frame, tiles sharded over %d GPUs, one film reduce per frame (strong scaling)" % (args.c4_spp, world),
              "metric": "Msamples/s", "value": W * H * args.c4_spp * n_c4 / c4_elapsed / 1e6, "n_gpus": world, "scaling": "strong",
              "steps": n_c4, "warmup": 1, "ms_per_step": c4_elapsed / n_c4 * 1e3, "spp_total": args.c4_spp}

    if rank == 0:
        # HBM traffic of the traversal launches from a separate PMC pass of the same frame (profiles/); only a
        # record taken with exactly these kernel sources counts
        traffic = None
        fabric_rec = None
        l2 = {"l2_hit_rate": None, "l2_miss_per_ray": None}
        traffic_note = "no PMC record for these kernel sources (tools/profile_round.sh writes profiles/*_traffic.json)"
        try:
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
            for f in reversed(cands):
                tj = json.load(open(os.path.join(ROOT, "profiles", f)))
                if tj.get("kernel_source_hash") == kernel_source_hash() and world == 1 and not strong \
                        and tj["workload"] == {"grid": args.grid, "res": args.res, "spp": args.spp}:
                    traffic = tj["hbm_bytes_per_launch"]; traffic_note = "profiles/" + f
                    fabric_rec = tj
                    l2 = {"l2_hit_rate": tj.get("l2_hit_rate"), "l2_miss_per_ray": tj.get("l2_miss_per_ray")}
                    break
        except Exception:
            pass
        total_samples = W * H * spp_total * args.steps
        value = total_samples / elapsed / 1e6
        achieved = bytes_per_step / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        rays = counts["rays_closest"] + counts["rays_shadow"]
        triad = pkg.hbm_triad_gbs(device)
        gather = pkg.gather_roof(device, 4)
        req_per_step = issued_requests(counts)
        replay = {"kernel": "k_trace (closest-hit and any-hit launches, by class)",
                  "what": "each class of traversal launch against a replay of its own request stream (mtsgpu_replay_roof: the same lines in "
                          "the same per-ray order, one 16-byte load each, eight independent requests in flight per lane, no arithmetic, "
                          "grid and LDS footprint of the closest-hit kernel).  A throughput test of the memory system on these lines, "
                          "not a bound: other issue orders may be faster.  Not replayed: what moves in queue order (queue ids, rays, "
                          "binned ids and hits, shadow rays), stack spills, the extra chunks of sphere primitives",
                  "issued_requests_per_ray": req_per_step / max(rays, 1),
                  "lds_served_requests_per_ray": (counts["req_pair_lds"] + counts["req_node_lds"]) / max(rays, 1),
                  "issued_breakdown_per_ray": {"pairs": counts["req_pair_global"] / max(rays, 1), "pop_nodes": counts["req_node_global"] / max(rays, 1),
                                               "record_heads": counts["req_head"] / max(rays, 1), "record_tails": counts["req_tail"] / max(rays, 1),
                                               "stack_spills": counts["req_spill"] / max(rays, 1)},
                  "random_gather_4MiB_G_per_s": gather / 1e9 if gather else None,
                  "note": replay_note}
        if replay_raw:
            samples = {"camera": "camera rays of the last pass, generated again, every %dth; closest-hit kernel in plain 64-ray batches",
                       "deep": "the last ray of every %dth path of the frame just rendered; closest-hit kernel with material binning",
                       "shadow": "every %dth slot of the shadow queue as the frame left it; any-hit kernel"}
            class_ms = {"camera": trace_first_ms, "shadow": trace_shadow_ms, "deep": trace_ms - trace_first_ms - trace_shadow_ms}
            classes = {}
            replay_ms_per_step = 0.0
            complete = True
            for kind, rr in replay_raw.items():
                if "error" in rr:
                    classes[kind] = {"error": rr["error"], "launches_ms_per_step": class_ms[kind]}
                    complete = False
                    continue
                ratio = rr["replay_ms"] / rr["product_ms"]
                classes[kind] = {"sample": ("%d rays: " % rr["rays"]) + samples[kind] % r_stride,
                                 "sample_requests_per_ray": rr["requests"] / rr["rays"], "sample_truncated_rays": rr["truncated_rays"],
                                 "product_ms": rr["product_ms"], "replay_ms": rr["replay_ms"], "replay_ratio": ratio,
                                 "product_G_requests_per_s": rr["requests"] / (rr["product_ms"] * 1e-3) / 1e9,
                                 "replay_G_requests_per_s": rr["requests"] / (rr["replay_ms"] * 1e-3) / 1e9,
                                 "launches_ms_per_step": class_ms[kind],
                                 "launches": {"camera": "the closest-hit launch of every pass's first bounce", "shadow": "all any-hit launches",
                                              "deep": "all other closest-hit launches"}[kind]}
                replay_ms_per_step += class_ms[kind] * ratio
            replay.update({"classes": classes, "unit": "G lane-requests/s", "trace_ms_per_step": trace_ms,
                           # sum over the frame's launches of (launch time x the replay ratio measured on its class's sample)
                           "replay_ms_per_step": replay_ms_per_step if complete else None,
                           "replay_ratio": replay_ms_per_step / trace_ms if complete and trace_ms > 0 else None})
        # what crosses the fabric (L2 misses, 128-byte lines; calibrated: profiles/r06a_fetch_size_calibration.txt) during the
        # traversal launches, against the rate the chip sustains for random 128-byte lines that live in the Infinity Cache
        fabric = None
        if fabric_rec and trace_ms > 0:
            fb = fabric_rec.get("fabric_bytes_per_frame")
            ceil = (fabric_rec.get("random_line_rate_G_per_s") or {}).get("k_sustained_150MB")
            if fb:
                lines = fb / 128.0 / (trace_ms * 1e-3) / 1e9
                fabric = {"bytes_per_step": fb, "tb_per_s": fb / (trace_ms * 1e-3) / 1e12, "G_lines_per_s": lines,
                          "random_line_ceiling_G_per_s": ceil, "frac_of_ceiling": lines / ceil if ceil else None,
                          "read_requests_per_ray": (fabric_rec.get("fabric_read_requests_per_frame") or 0) / max(rays, 1) or None,
                          "what": "bytes between the L2s and the Infinity Cache / HBM per step (2 x FETCH_SIZE + WRITE_SIZE, the x2 measured) over the traversal "
                                  "launches' time; ceiling = sustained random 16-byte gathers, eight in flight per lane, on distinct lines of a 150 MB footprint (tools/micro/gather_calib.hip k_sustained)"}
        sh_bytes = shade_algorithmic_bytes(counts)
        sh_achieved = sh_bytes / (shade_ms * 1e-3) / 1e9 if shade_ms > 0 else 0.0
        out = {
            "metric": "Msamples/s", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the film reduce: which collective summed the per-GPU films, and how many ranks its first collective (a sum of ones) saw
            "reduce_kind": "none (one GPU)" if world == 1 else ("rccl all-ranks reduce (torch.distributed nccl backend)" if backend == "nccl" else "gloo on host copies (ranks share a device: test mode)"),
            "rccl_ranks": collective_ranks if backend == "nccl" else 0,
            "config": {
                "workload": "%s 1M-tri Cornell: %d-tri displaced walls TriMesh + 20480-tri dielectric icosphere + area light, "
                            "path maxDepth=%d rrDepth=%d, %dx%d, ldsampler %s, box filter"
                            % ("C4" if strong else "C3", 5 * 2 * args.grid * args.grid, sd.max_depth, sd.rr_depth, W, H,
                               ("%d spp per frame (strong scaling)" % spp_total) if strong
                               else ("%d spp per GPU (%d per frame, weak scaling)" % (args.spp, spp_total)))
                            + ("; the C4 frame (configs[3]: %d spp per frame over the same %d GPUs, strong scaling) is timed after it: c4_strong" % (args.c4_spp, world) if c4 else ""),
                "triangles": int(scene.sc.n_tris), "kd_nodes": int(scene.sc.n_nodes), "kd_indices": int(scene.sc.n_indices),
                "parallelism": "ImageBlock tiles, morton(tx, ty) %% %d + one RCCL film reduce per frame" % world,
                "host_flatten_s": flatten_s,
            },
            # per rank: ms per step of its own frame (film clear + all kernels), and of the film reduce that follows
            # (it includes the wait for the slowest rank; on rank 0 also the receive)
            "rank_ms": rank_ms, "reduce_ms": reduce_ms,
            "c4_strong": c4,
            "time_to_1spp_frame_ms": one_spp_ms,
            "avg_path_length": counts.get("avg_path_length"),
            "roofline": {
                "kernel": "k_trace (closest-hit + shadow kd-tree traversal)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                "l2_hit_rate": l2["l2_hit_rate"], "l2_miss_per_ray": l2["l2_miss_per_ray"],     # TCC_HIT / TCC_MISS of the same record
                "fabric": fabric,
                "peak_measured_triad": triad, "frac_of_triad": achieved / triad if triad else None,
                "algorithmic_bytes_per_launch": bytes_per_step / max(counts["trace_launches"], 1),
                "algorithmic_bytes_per_step": bytes_per_step, "bytes_per_ray": bytes_per_step / max(rays, 1),
                "rays_per_step": rays, "launches_per_step": counts["trace_launches"],
                "avg_launch_ms": trace_ms / max(trace_launches, 1), "trace_ms_per_step": trace_ms,
                "kernel_times_from": "%d untimed steps with HIP events around every launch, right before the timed steps" % kt_steps,
                "shade_ms_per_step": shade_ms,
                "n_inner_per_ray": counts["n_inner"] / max(rays, 1), "n_leaf_per_ray": counts["n_leaf"] / max(rays, 1),
                "n_idx_per_ray": counts["n_idx"] / max(rays, 1), "n_tri_tested_per_ray": counts["n_tri_tested"] / max(rays, 1),
            },
            "roofline_requests": replay,
            "roofline_shade": {
                "kernel": "k_shade (one Li iteration per path: emitter hit, MIS, RR, NEE sample, BSDF sample); ms_per_step brackets the k_shade launches only",
                "bound": "hbm", "achieved": sh_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": sh_achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_step": sh_bytes, "ms_per_step": shade_ms,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, sd, args.res, args.spp, sd.max_depth)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if (world > 1 or args.group) and not args.no_group:
            # every rank is done (the others exit now): the form the Mitsuba plugin uses, all GPUs behind one process.
            # With one GPU it is the same frame once more (--group asks for it anyway)
            out["group"] = run_group_child(args, world)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
