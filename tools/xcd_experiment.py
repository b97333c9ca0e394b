"""Round 4, VERDICT item 1: what do XCD-local ray queues buy k_trace?

Same 8 M incoherent rays (origins on the five displaced walls of C3, directions into the box), different queue
orders, with and without `xcd_segments` (XCD x = workgroups with blockIdx % 8 == x takes the x-th eighth of the
queue).  Every XCD has its own 4 MB L2 and the deep part of the tree (nodes + leaf records: 57 MB) is what misses
in it, so the question is whether handing each XCD the rays of ONE region of the scene turns misses into hits.

The `xcd_segments` tuning knob this script drives was part of k_trace up to commit 95795f9 and was taken out once the
result was in (profiles/r04a_exp_xcd_queues_*.txt: L2 hit rate 0.58 -> 0.80, kernel time unchanged); check that commit out
to run it again.

  python3 tools/xcd_experiment.py            best-of-3 kernel time per order (HIP events)
  python3 tools/xcd_experiment.py --pmc      one launch per order, no warm-up: under
        rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ... the i-th k_trace dispatch is the i-th label printed
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import _pkgload
pkg = _pkgload.load()
PMC = "--pmc" in sys.argv
sd = pkg.scenes.cornell_c3()
scene = pkg.Scene(sd)
cam = pkg.PerspectiveCamera.for_description(sd, 64, 64)
it = pkg.MIPathTracer(maxDepth=16)
it.preprocess(scene, cam)
it.set_options(time_kernels=True)
n = 8_000_000
rng = np.random.RandomState(1)
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

# where a ray leaves the scene's box: the deep part of the tree it ends in
lo, hi = np.array([-1.0, 0.0, -1.0]), np.array([1.0, 2.0, 1.0])
with np.errstate(divide="ignore", invalid="ignore"):
    tfar = np.where(d > 0, (hi - o) / d, np.where(d < 0, (lo - o) / d, np.inf))
axis = np.argmin(tfar, axis=1)
texit = tfar[np.arange(n), axis]
pexit = o + texit[:, None] * d
eface = axis * 2 + (d[np.arange(n), axis] > 0)

# the reference tree's top three levels: 3 bits per point (side of the root split, of that child's split, ...)
nodes = scene.arrays()["kd_nodes"]
def kd_region(p, levels=3):
    """walks gkdtree.h:442-470 nodes: inner = (left child offset << 2 | axis, split); children adjacent, relative offset"""
    idx = np.zeros(len(p), dtype=np.int64)
    key = np.zeros(len(p), dtype=np.int64)
    for _ in range(levels):
        w0 = nodes[idx, 0]; split = nodes[idx, 1].view(np.float32)
        leaf = (w0 & 0x80000000) != 0
        ax = (w0 & 3).astype(np.int64)
        right = p[np.arange(len(p)), np.where(leaf, 0, ax)] > split
        left = idx + ((w0 & 0x7FFFFFFF) >> 2).astype(np.int64)
        nxt = np.where(right, left + 1, left)
        key = key * 2 + np.where(leaf, 0, right.astype(np.int64))
        idx = np.where(leaf, idx, nxt)
    return key

def cells(p, g):
    c = np.clip(((p - lo) / (hi - lo) * g).astype(np.int64), 0, g - 1)
    return (c[:, 0] * g + c[:, 1]) * g + c[:, 2]

def exit_key(g):
    a = (axis + 1) % 3; b = (axis + 2) % 3
    ca = np.clip(((pexit[np.arange(n), a] - lo[a]) / (hi[a] - lo[a]) * g).astype(np.int64), 0, g - 1)
    cb = np.clip(((pexit[np.arange(n), b] - lo[b]) / (hi[b] - lo[b]) * g).astype(np.int64), 0, g - 1)
    return (eface * g + ca) * g + cb

def shuffled_eighths(order):
    """equal eighths of a sorted order, shuffled inside: what a per-XCD queue WITHOUT a sort can look like"""
    sh = order.copy(); seg = (n + 7) // 8
    for k in range(8):
        s_ = sh[k * seg:(k + 1) * seg]; rng.shuffle(s_)
    return sh

labels = []
def run(order, label, xcd):
    it.set_tuning(xcd_segments=xcd)
    r = rays[order] if order is not None else rays
    if PMC:
        it.trace_rays(r)
        labels.append((label, xcd, it.stats()["trace_ms"]))
        print("PMC %-52s xcd=%d  %.2f ms" % (label, xcd, it.stats()["trace_ms"]), flush=True)
        return
    it.trace_rays(r[:100000])
    best = 1e9
    for _ in range(3):
        it.trace_rays(r)
        best = min(best, it.stats()["trace_ms"])
    print("%-52s xcd=%d  %.2f ms  %.2f Grays/s" % (label, xcd, best, n / best / 1e6), flush=True)

try:
    okey3 = kd_region(o); ekey3 = kd_region(pexit)
    kd_ok = True
except Exception as e:              # node layout differs from what kd_region assumes: the grid keys still run
    print("kd_region failed:", e); kd_ok = False

run(None, "random", 0)
run(None, "random", 1)
o16 = np.argsort(cells(o, 16), kind="stable")
e16 = np.argsort(exit_key(16), kind="stable")
e4 = np.argsort(exit_key(4), kind="stable")
for lab, order in (("sorted by origin cell 16^3", o16), ("sorted by exit face + 16x16", e16), ("sorted by exit face + 4x4", e4)):
    run(order, lab, 0); run(order, lab, 1)
run(shuffled_eighths(o16), "eighths of origin-cell order, shuffled inside", 1)
run(shuffled_eighths(e16), "eighths of exit-cell order, shuffled inside", 1)
run(shuffled_eighths(e16), "eighths of exit-cell order, shuffled inside", 0)
if kd_ok:
    for lab, key in (("3-bit kd region of the origin", okey3), ("3-bit kd region of the exit point", ekey3),
                     ("3-bit kd region of origin, then of exit (6 bits)", okey3 * 8 + ekey3)):
        print("   sizes of the 8 regions:", np.bincount(key % 8 if key.max() < 8 else key // 8, minlength=8).tolist())
        order = shuffled_eighths(np.argsort(key, kind="stable"))
        run(order, lab + ", equal eighths, shuffled inside", 1)
    # origin AND exit region equal (the rays whose whole deep working set is one region) first, per region
    both = np.argsort(okey3 * 8 + ekey3, kind="stable")
    run(both, "sorted by (origin region, exit region), 64 classes", 1)
    run(both, "sorted by (origin region, exit region), 64 classes", 0)
if PMC:
    print("LABELS", labels)
