#!/usr/bin/env python3
"""bench.py -- Msamples/s of the path-tracing hot path on N MI355X GPUs of one node.

Workload (BASELINE.json configs[2]/[3], SURVEY.md 8d "C3/C4"): procedural Cornell variant with one
1 024 000-triangle displaced walls TriMesh (lambertian), a 20 480-triangle dielectric icosphere and
an area luminaire; `path` integrator maxDepth=16, rrDepth=10; 1024x1024, low-discrepancy sampler.
A "step" is one full-frame render.  Weak scaling: every GPU renders 64 spp worth of samples for its
share of the ImageBlock tiles (spp = 64*N, tiles t % N == rank), then the per-GPU films are summed
once per step with a reduce over RCCL/xGMI (Film::putImageBlock).  Scene upload is excluded, the film
reduce is included.  value = total camera samples / wall time.

One JSON line on rank 0 (see DESIGN.md section 8 for the definition of every field)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(st):
    """B_ray = 8 n_inner + 8 n_leaf + 4 n_idx + 48 n_tri + 48 (ray in 32 B + hit out 16 B),
    n_tri counted without mailbox credit (= index entries visited)."""
    rays = st["rays_closest"] + st["rays_shadow"]
    return 8 * st["n_inner"] + 8 * st["n_leaf"] + 4 * st["n_idx"] + 48 * st["n_idx"] + 48 * rays


def cpu_baseline(pkg, sd, res, spp, max_depth, seconds=15.0):
    """The oracle (CPU restatement, kind 'port') on a bounded centre crop of the same frame,
    all host cores (OpenMP).  Test infrastructure used as a reported baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    oscene = orc.FlatScene(sd)
    cam = orc.make_camera(sd, res, res)
    prm = orc.render_params(max_depth, sampler=pkg.abi.SAMPLER_LD_KEYED, spp=spp, seed=0x5EED)
    cores = os.cpu_count() or 1
    c = res // 2
    # calibrate on a small crop, then size the measured crop for ~`seconds` of wall time
    t0 = time.time()
    orc.render(oscene.scene, cam, prm, rect=(c - 8, c - 8, c + 8, c + 8))
    rate = 16 * 16 * spp / max(time.time() - t0, 1e-3)
    side = int(min(res, max(16, (rate * seconds / spp) ** 0.5)))
    side -= side % 2
    h = side // 2
    t0 = time.time()
    _, st = orc.render(oscene.scene, cam, prm, rect=(c - h, c - h, c + h, c + h))
    dt = time.time() - t0
    n = side * side * spp
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%dx%d-pixel centre crop of the %dx%d frame x %d spp (%d camera samples, %.1f s), oracle/liboracle.so with OpenMP"
                      % (side, side, res, res, spp, n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel per GPU")
    ap.add_argument("--grid", type=int, default=320, help="wall grid resolution (320 -> 1 024 000 triangles)")
    ap.add_argument("--max-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-kd", action="store_true", help="kd-tree binning phase on the host instead of the GPU (same tree)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    import _pkgload
    pkg = _pkgload.load()
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X; libmtsgpu has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # --- scene (host side: generate, flatten, upload; not timed) ---
    sd = pkg.scenes.cornell_c3(grid=args.grid, sphere_subdiv=5)
    t0 = time.time()
    kp = None
    if os.environ.get("MTSGPU_KD_TRAV"):          # experiment knob: Scene property kdTraversalCost (scene.cpp:54-88)
        kp = pkg.abi.KdParams(); kp.traversal_cost = float(os.environ["MTSGPU_KD_TRAV"])
    scene = pkg.Scene(sd, kp, gpu_binning=not args.host_kd)
    flatten_s = time.time() - t0
    W = H = args.res
    spp_total = args.spp * world
    cam = pkg.PerspectiveCamera.for_description(sd, W, H)
    it = pkg.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth, device=local_rank)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp_total, seed=0x5EED)
    it.set_tiles(32, rank, world)
    film = torch.zeros((H, W, 5), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    it.set_stream(stream.cuda_stream)
    it.set_film_buffer(film.data_ptr(), keepalive=film)

    def step():
        film.zero_()
        if not it.render():
            raise RuntimeError("render cancelled")
        if world > 1:
            pkg.filmreduce.reduce_film(film, dst=0)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # --- untimed: warmup + one counting step (algorithmic work of the traversal kernels) ---
    it.set_options(max_paths=args.max_paths, count_traversal=False, time_kernels=True)
    for _ in range(args.warmup):
        step()
    it.set_options(max_paths=args.max_paths, count_traversal=True, time_kernels=True)
    step()
    torch.cuda.synchronize()
    counts = it.stats()
    bytes_per_step = algorithmic_bytes(counts)
    it.set_options(max_paths=args.max_paths, count_traversal=False, time_kernels=True)

    # --- timed region: exactly K steps between barriers ---
    trace_ms = 0.0
    trace_launches = 0
    shade_ms = 0.0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = it.stats()                     # per-launch HIP-event durations on the library's stream
        trace_ms += st["trace_ms"]; shade_ms += st["shade_ms"]; trace_launches += st["trace_launches"]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # --- time to a 1-spp frame (second half of BASELINE.json's metric), untimed region ---
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=1, seed=0x5EED)
    it.set_tiles(32, rank, world)
    step(); fence()
    t1 = time.perf_counter()
    step(); fence()
    one_spp_ms = (time.perf_counter() - t1) * 1e3

    if rank == 0:
        # HBM traffic of the traversal launches from a separate PMC pass of the same frame (profiles/)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tj["workload"] == {"grid": args.grid, "res": args.res, "spp": args.spp} and world == 1:
                traffic = tj["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        total_samples = W * H * spp_total * args.steps
        value = total_samples / elapsed / 1e6
        achieved = bytes_per_step * args.steps / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        rays = counts["rays_closest"] + counts["rays_shadow"]
        out = {
            "metric": "Msamples/s", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "C3/C4 1M-tri Cornell: %d-tri displaced walls TriMesh + 20480-tri dielectric icosphere + area light, "
                            "path maxDepth=%d rrDepth=%d, %dx%d, ldsampler %d spp per GPU (%d total), box filter"
                            % (5 * 2 * args.grid * args.grid, sd.max_depth, sd.rr_depth, W, H, args.spp, spp_total),
                "triangles": int(scene.sc.n_tris), "kd_nodes": int(scene.sc.n_nodes), "kd_indices": int(scene.sc.n_indices),
                "parallelism": "ImageBlock tiles t%%%d + one RCCL film reduce per frame" % world,
                "host_flatten_s": flatten_s,
            },
            "time_to_1spp_frame_ms": one_spp_ms,
            "roofline": {
                "kernel": "k_trace (closest-hit + shadow kd-tree traversal)",
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per_step / max(counts["trace_launches"], 1),
                "algorithmic_bytes_per_step": bytes_per_step, "bytes_per_ray": bytes_per_step / max(rays, 1),
                "rays_per_step": rays, "launches_per_step": counts["trace_launches"],
                "avg_launch_ms": trace_ms / max(trace_launches, 1), "trace_ms_per_step": trace_ms / args.steps,
                "shade_ms_per_step": shade_ms / args.steps,
                "n_inner_per_ray": counts["n_inner"] / max(rays, 1), "n_leaf_per_ray": counts["n_leaf"] / max(rays, 1),
                "n_idx_per_ray": counts["n_idx"] / max(rays, 1), "n_tri_tested_per_ray": counts["n_tri_tested"] / max(rays, 1),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, sd, args.res, args.spp, sd.max_depth)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
