"""Host vs device phases of the kd-tree build ("kd-tree build").
usage: MTSGPU_KDTIMING=1 python tools/kdbuild_bench.py [grid ...]     (320 -> 1 M triangles, 1000 -> 10 M)"""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
mts = importlib.import_module("mitsuba-renderer_amd")
_kp = mts.abi.KdParams(); _kp.exact_prim_threshold = 300
mts.Scene(mts.scenes.cornell_c5(sphere_subdiv=3), kd_params=_kp, gpu_binning=True, gpu_exact=True)      # HIP runtime + code objects, not part of any build
for grid in [int(a) for a in sys.argv[1:]] or [320, 1000]:
    sd = mts.scenes.cornell_c3(grid=grid)
    res = {}
    for name, binning, exact in (("host", False, False), ("device binning", True, False), ("device binning + exact", True, True)):
        best = 1e30
        for _ in range(2):
            t = time.time()
            sc = mts.Scene(sd, gpu_binning=binning, gpu_exact=exact)
            best = min(best, time.time() - t)
        res[name] = sc.arrays()
        print("grid %d: %-24s flatten + build %.3f s" % (grid, name, best), flush=True)
    same = all(np.array_equal(res["host"][k], res[n][k]) for n in res for k in ("kd_nodes", "kd_indices"))
    print("grid %d: %d nodes, trees identical: %s" % (grid, res["host"]["kd_nodes"].shape[0], same), flush=True)
