"""BASELINE.json's other configs as timed frames (the bench line is C3 / C4): C2 -- the C1 box, ldsampler 1024 spp, maxDepth 4 --
and C5 -- mixed BSDFs (lambertian / roughmetal / dielectric / microfacet icospheres of subdivision 4), constant environment,
maxDepth 32, 256 spp -- both at 512 x 512, with HIP-event kernel times:  python3 tools/config_frames.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkgload
pkg = _pkgload.load()
res = 512
for name, sd, spp, sampler in (("C2", pkg.scenes.cornell_c1(), 1024, "ldsampler"), ("C5", pkg.scenes.cornell_c5(sphere_subdiv=4), 256, "ldsampler"),
                               ("C5", pkg.scenes.cornell_c5(sphere_subdiv=4), 256, "independent")):
    scene = pkg.Scene(sd)
    cam = pkg.PerspectiveCamera.for_description(sd, res, res)
    it = pkg.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
    it.preprocess(scene, cam, sampler=sampler, sampleCount=spp, seed=0x5EED)
    assert it.render()
    best = 1e30
    for timing in (False, False, True):
        it.set_options(time_kernels=timing)
        t0 = time.perf_counter(); assert it.render(); dt = (time.perf_counter() - t0) * 1e3
        if not timing: best = min(best, dt)
    st = it.stats()
    n = res * res * spp
    print("%s (%d tris, maxDepth %d, %s %d spp, %dx%d): %.1f ms  %.1f Msamples/s | traversal %.1f ms (%d launches, %.2f + %.2f rays per sample), shading %.1f ms, avg path length %.2f"
          % (name, sd.n_tris, sd.max_depth, sampler, spp, res, res, best, n / best / 1e3, st["trace_ms"], st["trace_launches"],
             st["rays_closest"] / n, st["rays_shadow"] / n, st["shade_ms"], st["avg_path_length"]), flush=True)
