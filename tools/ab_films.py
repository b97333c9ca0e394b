"""Films of one C3 frame under different tuning knobs must be identical: python3 tools/ab_films.py <spp> <res> <grid> knob=value[,knob=value] ...
(each argument after the first three is one configuration; the first one is the reference; save=<file.npy> / ref=<file.npy>
in a configuration store the film / take the reference from a file, to compare library variants)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import _pkgload
pkg = _pkgload.load()
spp, res, grid = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sd = pkg.scenes.cornell_c3(grid=grid)
scene = pkg.Scene(sd, None, gpu_binning=True, gpu_exact=True)
cam = pkg.PerspectiveCamera.for_description(sd, res, res)
ref = None
for conf in sys.argv[4:]:
    it = pkg.MIPathTracer(maxDepth=sd.max_depth)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, seed=0x5EED)
    knobs = dict(kv.split("=") for kv in conf.split(",") if kv and kv != "default")
    save, load = knobs.pop("save", None), knobs.pop("ref", None)      # films across libraries (MTSGPU_LIB): save=<npy> / ref=<npy>
    if load:
        ref = np.load(load)
    if knobs:
        it.set_tuning(**{k: int(v) for k, v in knobs.items()})
    assert it.render()
    f = it.film().copy()
    st = it.stats()
    if save:
        np.save(save, f)
    if ref is None:
        ref = f
    same = np.array_equal(ref.view(np.uint32), f.view(np.uint32))
    print("%-40s film %s  rays %d+%d  total %.2f ms" % (conf, "identical" if same else "DIFFERS", st["rays_closest"], st["rays_shadow"], st["total_ms"]))
    assert same, conf
