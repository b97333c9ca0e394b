"""One C4 frame (C3 scene, 1024^2, ldsampler 4096 spp, 58 passes) per setting of the knobs given:
python3 tools/c4_frame.py [spp] [knob=v0,v1 ...]   e.g.  tools/c4_frame.py 4096 nee_parked=0,1"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkgload
pkg = _pkgload.load()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sweeps = [(a.split("=")[0], [int(v) for v in a.split("=")[1].split(",")]) for a in sys.argv[2:]]
sd = pkg.scenes.cornell_c3()
scene = pkg.Scene(sd, None, gpu_binning=True, gpu_exact=False)
cam = pkg.PerspectiveCamera.for_description(sd, 1024, 1024)
it = pkg.MIPathTracer(maxDepth=sd.max_depth)
it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=0x5EED)
settings = [{}]
for k, vs in sweeps:
    settings = [dict(s, **{k: v}) for s in settings for v in vs]
for rep in range(int(os.environ.get("C4_REPS", "2"))):
    for s in settings:
        if s:
            it.set_tuning(**s)
        for timing in ((False, True) if "kt" in os.environ.get("C4_FRAME", "") else (False,)):
            it.set_options(time_kernels=timing)
            t0 = time.perf_counter()
            assert it.render()
            dt = (time.perf_counter() - t0) * 1e3
            st = it.stats()
            print("%s time_kernels=%d: wall %.1f ms (%.1f Msamples/s)  trace %.1f  shade %.1f" % (s, timing, dt, 1024 * 1024 * spp / dt / 1e3, st["trace_ms"], st["shade_ms"]), flush=True)
