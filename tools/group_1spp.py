"""1-spp frame through a device group whose members all sit on GPU 0 (K independent bounce chains on one chip):
python3 tools/group_1spp.py K [spp]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkgload
pkg = _pkgload.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sd = pkg.scenes.cornell_c3()
scene = pkg.Scene(sd, None, gpu_binning=True, gpu_exact=True)
cam = pkg.PerspectiveCamera.for_description(sd, 1024, 1024)
g = pkg.DeviceGroup([0] * K, maxDepth=sd.max_depth)
g.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=0x5EED)
assert g.render()
best = 1e30
for _ in range(5):
    t0 = time.perf_counter()
    assert g.render()
    best = min(best, (time.perf_counter() - t0) * 1e3)
print("group of %d on one GPU, %d spp: wall(best of 5) %.2f ms (%s)" % (K, spp, best, g.reduce_kind()))
