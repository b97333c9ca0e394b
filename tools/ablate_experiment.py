"""k_trace timing on a fixed set of incoherent rays (results are wrong under ablation; only time matters)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import _pkgload
pkg = _pkgload.load()
sd = pkg.scenes.cornell_c3()
scene = pkg.Scene(sd)
cam = pkg.PerspectiveCamera.for_description(sd, 64, 64)
it = pkg.MIPathTracer(maxDepth=16)
it.preprocess(scene, cam)
it.set_options(time_kernels=True)
n = 8_000_000
rng = np.random.RandomState(1)
o = np.stack([rng.rand(n) * 1.9 - 0.95, rng.rand(n) * 1.9 + 0.05, rng.rand(n) * 1.9 - 0.95], axis=1)
d = rng.randn(n, 3); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((n, 8), dtype=np.float32)
rays[:, 0:3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.inf
it.trace_rays(rays[:100000])
best = 1e9
for _ in range(3):
    it.trace_rays(rays); best = min(best, it.stats()["trace_ms"])
it.set_options(time_kernels=True, count_traversal=True)
it.trace_rays(rays); st = it.stats()
print("ablate=%s  %.2f ms  %.2f Grays/s   inner/ray %.1f leaf/ray %.1f idx/ray %.1f" % (os.environ.get("MTSGPU_ABLATE", "0"), best, n / best / 1e6,
      st["n_inner"] / n, st["n_leaf"] / n, st["n_idx"] / n))
