"""Experiment: wall time of one 1-spp C3 frame rendered by K contexts (one HIP stream + host thread each) that
shard the tiles and share one film buffer:  python3 tools/multictx_1spp.py"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import _pkgload
pkg = _pkgload.load()
sd = pkg.scenes.cornell_c3()
scene = pkg.Scene(sd)
W = H = 1024
cam = pkg.PerspectiveCamera.for_description(sd, W, H)
film = torch.zeros((H, W, 5), dtype=torch.float32, device="cuda:0")
for K in (1, 2, 3, 4, 6, 8):
    its = []
    for k in range(K):
        it = pkg.MIPathTracer(maxDepth=sd.max_depth)
        it.preprocess(scene, cam, sampler="ldsampler", sampleCount=1, seed=0x5EED)
        it.set_tiles(32, k, K)
        it.set_film_buffer(film.data_ptr())
        its.append(it)
    def run(it):
        assert it.render()
    best = 1e9
    for rep in range(4):
        film.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(it,)) for it in its]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("contexts %d: 1-spp frame %.2f ms  (film sum %.1f)" % (K, best * 1e3, float(film[..., 4].sum())))
    del its
