"""Per-bounce kernel times of one C3 frame (MTSGPU_DEBUG=1 prints them): python3 tools/bounce_times.py [spp] [res] [knob=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkgload
pkg = _pkgload.load()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
knobs = dict(a.split("=") for a in sys.argv[3:])
if knobs.pop("debug", "0") == "1":
    os.environ["MTSGPU_DEBUG"] = "1"
count = knobs.pop("count", "0") == "1"
sampler = knobs.pop("sampler", "ldsampler")
grid = int(knobs.pop("grid", "320"))          # 1000 = the 10 M-triangle scene
timing = knobs.pop("timing", "1") == "1"      # timing=0: no HIP events around the launches (the frame as bench.py times it)
sd = pkg.scenes.cornell_c3(grid=grid)
scene = pkg.Scene(sd, None, gpu_binning=True, gpu_exact=True)
cam = pkg.PerspectiveCamera.for_description(sd, res, res)
it = pkg.MIPathTracer(maxDepth=sd.max_depth)
it.preprocess(scene, cam, sampler=sampler, sampleCount=spp, seed=0x5EED)
if knobs:
    it.set_tuning(**{k: int(v) for k, v in knobs.items()})
it.set_options(time_kernels=timing, count_traversal=count)
assert it.render()                      # warm-up (allocations)
sys.stderr.write("---- frame ----\n")
if "quiet" not in os.environ.get("MTSGPU_BT", ""):
    pass
best = 1e30
for _ in range(3):
    t0 = time.perf_counter()
    assert it.render()
    best = min(best, (time.perf_counter() - t0) * 1e3)
dt = best
st = it.stats()
print("wall(best of 3) %.2f ms  total(dev) %.2f ms  trace %.2f ms (union %.2f)  shade %.2f ms  launches %d  rays %d+%d" % (
    dt, st["total_ms"], st["trace_ms"], st["trace_union_ms"], st["shade_ms"], st["trace_launches"], st["rays_closest"], st["rays_shadow"]))
