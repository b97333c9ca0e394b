"""Upper bound of what ray re-ordering could buy k_trace: same rays, different queue order."""
import os, sys, time
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
# origins on the five walls, cosine-ish directions into the box
face = rng.randint(0, 5, n)
u, v = rng.rand(n) * 2 - 1, rng.rand(n) * 2
o = np.zeros((n, 3)); nrm = np.zeros((n, 3))
for f, (ax, val, sgn) in enumerate([(1, 0.0, 1), (1, 2.0, -1), (2, -1.0, 1), (0, -1.0, 1), (0, 1.0, -1)]):
    m = face == f
    a, b = [k for k in range(3) if k != ax]
    o[m, ax] = val + sgn * 0.02
    o[m, a] = u[m] if a != 1 else v[m]
    o[m, b] = u[m] if (b != 1 and a == 1) else (v[m] if b == 1 else rng.rand(m.sum()) * 2 - 1)
    nrm[m, ax] = sgn
d = rng.randn(n, 3); d /= np.linalg.norm(d, axis=1, keepdims=True)
flip = (d * nrm).sum(axis=1) < 0
d[flip] *= -1
rays = np.zeros((n, 8), dtype=np.float32)
rays[:, 0:3] = o; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.inf

def run(order, label):
    r = rays[order] if order is not None else rays
    it.trace_rays(r[:100000])          # warm
    best = 1e9
    for _ in range(3):
        it.trace_rays(r)
        best = min(best, it.stats()["trace_ms"])
    print("%-28s %.2f ms  %.2f Grays/s" % (label, best, n / best / 1e6))

run(None, "random order")
octant = (d[:, 0] > 0).astype(np.int64) | ((d[:, 1] > 0).astype(np.int64) << 1) | ((d[:, 2] > 0).astype(np.int64) << 2)
for g in (8, 16, 64, 256):
    cell = np.clip(((o + [1, 0, 1]) / 2 * g).astype(np.int64), 0, g - 1)
    key = ((cell[:, 0] * g + cell[:, 1]) * g + cell[:, 2]) * 8 + octant
    run(np.argsort(key, kind="stable"), "grid %d^3 + octant" % g)
    key2 = (cell[:, 0] * g + cell[:, 1]) * g + cell[:, 2]
    run(np.argsort(key2, kind="stable"), "grid %d^3 only" % g)
run(np.argsort(octant, kind="stable"), "octant only")

# --- destination-based order: where the ray leaves the scene box decides which deep part of the tree it visits ---
lo, hi = np.array([-1.0, 0.0, -1.0]), np.array([1.0, 2.0, 1.0])
with np.errstate(divide="ignore", invalid="ignore"):
    tfar = np.where(d > 0, (hi - o) / d, np.where(d < 0, (lo - o) / d, np.inf))
axis = np.argmin(tfar, axis=1)
texit = tfar[np.arange(n), axis]
p = o + texit[:, None] * d
face = axis * 2 + (d[np.arange(n), axis] > 0)
def dest_key(g):
    a = (axis + 1) % 3; b = (axis + 2) % 3
    ca = np.clip(((p[np.arange(n), a] - lo[a]) / (hi[a] - lo[a]) * g).astype(np.int64), 0, g - 1)
    cb = np.clip(((p[np.arange(n), b] - lo[b]) / (hi[b] - lo[b]) * g).astype(np.int64), 0, g - 1)
    return (face * g + ca) * g + cb
for g in (1, 4, 16, 64):
    order = np.argsort(dest_key(g), kind="stable")
    for xcd in (0, 1):
        it.set_tuning(xcd_segments=xcd)
        run(order, "exit face + %dx%d cells, xcd=%d" % (g, g, xcd))
it.set_tuning(xcd_segments=1)
run(None, "random order, xcd=1")
g = 16
cell = np.clip(((o + [1, 0, 1]) / 2 * g).astype(np.int64), 0, g - 1)
run(np.argsort((cell[:, 0] * g + cell[:, 1]) * g + cell[:, 2], kind="stable"), "origin grid 16^3, xcd=1")
# balanced version: equal-count segments are what contiguous eighths give; shuffle inside each eighth to separate
# the effect of the L2 (per XCD) from that of coherent waves
order = np.argsort(dest_key(16), kind="stable")
seg = (n + 7) // 8
sh = order.copy()
for k in range(8):
    s_ = sh[k * seg:(k + 1) * seg]; rng.shuffle(s_)
run(sh, "exit 16x16 per XCD, shuffled inside, xcd=1")
