"""The reference's own traversal benchmark (src/tests/test_kd.cpp:85-130, "bunny benchmark") on the MI355X path:
10 M random chords of the sphere (centre (-0.016840, 0.110154, -0.001537), radius 0.2) through the kd-tree of
data/tests/bunny.ply, any-hit query ShapeKDTree::rayIntersect(ray).  Prints Mrays/s from the HIP-event time of the
traversal kernel (rays resident in HBM).  The chords are drawn with numpy here (same distribution as the reference's
default-seeded Random; the exact reference rays are used by tests/test_gpu_parity.py::test_bunny_benchmark_rays).
    python3 tools/bunny_bench.py [n_rays]"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkgload
import ply_io
pkg = _pkgload.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pos, tri = ply_io.read(os.path.join(ROOT, "tests", "golden", "bunny.ply"))
sd = pkg.scenes.SceneDescription("bunny")
sd.add_mesh(pos, tri, bsdf=sd.lambertian(0.5), face_normals=False, name="bunny")
sd.point_light((0.0, 0.5, 0.5), 1.0)       # a scene needs a luminaire (scene.cpp:310-318)
scene = pkg.Scene(sd)
it = pkg.MIPathTracer(maxDepth=2)
it.preprocess(scene, pkg.PerspectiveCamera.for_description(sd, 16, 16), sampler="independent", sampleCount=1)
it.set_options(time_kernels=True)
rng = np.random.default_rng(5489)
def sphere(u):
    z = 1.0 - 2.0 * u[:, 1]; r = np.sqrt(np.maximum(0.0, 1.0 - z * z)); phi = 2.0 * np.pi * u[:, 0]
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)
c = np.array([-0.016840, 0.110154, -0.001537])
p1 = c + 0.2 * sphere(rng.random((n, 2))); p2 = c + 0.2 * sphere(rng.random((n, 2)))
d = p2 - p1; d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((n, 8), dtype=np.float32)
rays[:, 0:3] = p1; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.inf
for mode, shadow in (("any-hit  ShapeKDTree::rayIntersect(ray)", True), ("closest  ShapeKDTree::rayIntersect(ray, its)", False)):
    it.trace_rays(rays[:1000], shadow=shadow)                     # warm-up
    hits = it.trace_rays(rays, shadow=shadow)
    ms = it.stats()["trace_ms"]
    found = hits[:, 3].mean() if shadow else (hits[:, 3] != 0xFFFFFFFF).mean()
    print("%s: %d rays, %.3f %% intersections, kernel %.2f ms -> %.1f Mrays/s" % (mode, n, 100 * found, ms, n / ms / 1e3))
