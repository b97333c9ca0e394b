"""One untimed C3 frame for counter collection: python3 tools/pmc_workload.py [spp] [res] [count]
(production kernels unless the third argument is "count": the counting build spills and writes more; MTSGPU_PMC_GRID=<n> sets the
wall grid -- 1000 = the 10 M-triangle scene)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _pkgload
pkg = _pkgload.load()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
grid = int(os.environ.get("MTSGPU_PMC_GRID", "320"))
sd = pkg.scenes.cornell_c3(grid=grid)
scene = pkg.Scene(sd, None, gpu_binning=True, gpu_exact=grid > 440)
cam = pkg.PerspectiveCamera.for_description(sd, res, res)
it = pkg.MIPathTracer(maxDepth=sd.max_depth)
it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=0x5EED)
it.set_options(count_traversal=(len(sys.argv) > 3 and sys.argv[3] == "count"))
assert it.render()
st = it.stats()
rays = st["rays_closest"] + st["rays_shadow"]
alg = 8 * st["n_inner"] + 8 * st["n_leaf"] + 52 * st["n_idx"] + 48 * rays
print("STATS", st, "algorithmic_bytes", alg)
