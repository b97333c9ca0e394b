"""Load balance of the tile sharding (mtsgpu_set_tiles: morton(tx, ty) % N): every part of N rendered on ONE GPU, rays and
device time per part.  The slowest part bounds the N-GPU frame: efficiency <= mean / max.
usage: python tools/shard_balance.py [N] [spp]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import _pkgload
pkg = _pkgload.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sd = pkg.scenes.cornell_c3()
scene = pkg.Scene(sd, None, gpu_binning=True, gpu_exact=True)
cam = pkg.PerspectiveCamera.for_description(sd, 1024, 1024)
it = pkg.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth)
for n in sorted({1, 2, 4, N}):
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp * n, seed=0x5EED)      # weak scaling: spp * n per frame
    rays, ms = [], []
    for part in range(n):
        it.set_tiles(32, part, n)
        it.set_options(time_kernels=True)
        assert it.render(); assert it.render()
        st = it.stats()
        rays.append(st["rays_closest"] + st["rays_shadow"]); ms.append(st["total_ms"])
    rays, ms = np.array(rays, dtype=np.float64), np.array(ms)
    print("N = %d (%d spp per frame): rays per part min / mean / max = %.4g / %.4g / %.4g (max / mean %.4f); device ms per part min / mean / max = %.1f / %.1f / %.1f -> balance bound %.3f"
          % (n, spp * n, rays.min(), rays.mean(), rays.max(), rays.max() / rays.mean(), ms.min(), ms.mean(), ms.max(), ms.mean() / ms.max()), flush=True)
