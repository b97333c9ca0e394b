"""Two passes of a C4-like frame (C3 scene, 192 x 192, ldsampler at the given spp) -- for rocprofv3 --stats runs that isolate
the sampler-table kernels:  python3 tools/k0_time.py [spp] [res]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkgload
pkg = _pkgload.load()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
res = int(sys.argv[2]) if len(sys.argv) > 2 else 192
sd = pkg.scenes.cornell_c3()
scene = pkg.Scene(sd, None, gpu_binning=True, gpu_exact=False)
cam = pkg.PerspectiveCamera.for_description(sd, res, res)
it = pkg.MIPathTracer(maxDepth=sd.max_depth)
it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=0x5EED)
for _ in range(2):
    t0 = time.perf_counter(); assert it.render(); print("frame %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
