"""Device exact phase vs host builder: python3 tools/kdexact_check.py [case ...]  (prints the first difference)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import _pkgload
mts = _pkgload.load()

def cases():
    kp = lambda **kw: [setattr(k, a, b) for k in [mts.abi.KdParams()] for a, b in kw.items()] and None
    def P(**kw):
        k = mts.abi.KdParams()
        for a, b in kw.items(): setattr(k, a, b)
        return k
    yield "c1", mts.scenes.cornell_c1(), {}
    yield "c5_sub2", mts.scenes.cornell_c5(sphere_subdiv=2), {}
    yield "c5_sub3_thr300", mts.scenes.cornell_c5(sphere_subdiv=3), dict(exact_prim_threshold=300)
    yield "spheres", mts.scenes.spheres(), {}
    yield "c3_grid60", mts.scenes.cornell_c3(grid=60, sphere_subdiv=3), {}
    yield "c3_grid60_noclip", mts.scenes.cornell_c3(grid=60, sphere_subdiv=3), dict(clip=-1)
    yield "c3_grid60_noretract", mts.scenes.cornell_c3(grid=60, sphere_subdiv=3), dict(retract=-1)
    for seed in range(6):
        yield "fuzz%d" % seed, mts.scenes.fuzz(seed), {}
    yield "c3_1M", mts.scenes.cornell_c3(), {}

want = sys.argv[1:]
bad = 0
for name, sd, kw in cases():
    if want and name not in want: continue
    def P():
        k = mts.abi.KdParams()
        for a, b in kw.items(): setattr(k, a, b)
        return k
    t0 = time.time(); host = mts.Scene(sd, kd_params=P(), gpu_binning=True); t1 = time.time()
    dev = mts.Scene(sd, kd_params=P(), gpu_binning=True, gpu_exact=True); t2 = time.time()
    ha, da = host.arrays(), dev.arrays()
    ok = True
    for k in ("kd_nodes", "kd_indices", "aabb_min", "aabb_max"):
        x, y = ha[k].view(np.uint32).ravel(), da[k].view(np.uint32).ravel()
        if x.shape != y.shape or not np.array_equal(x, y):
            ok = False
            n = min(len(x), len(y)); d = np.nonzero(x[:n] != y[:n])[0]
            print("  %s differs: sizes %d / %d, first at %s" % (k, len(x), len(y), d[:5]))
    if host.kdstats() != dev.kdstats():
        ok = False; print("  stats", host.kdstats(), dev.kdstats())
    print("%-22s %s  tris %d nodes %d  host %.3f s  device-exact %.3f s" % (name, "OK " if ok else "DIFF", sd_n if False else host.sc.n_tris, host.sc.n_nodes, t1 - t0, t2 - t1), flush=True)
    bad += not ok
sys.exit(1 if bad else 0)
