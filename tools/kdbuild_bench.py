"""Host vs device binning phase of the kd-tree build ("kd-tree build").
usage: MTSGPU_KDTIMING=1 python tools/kdbuild_bench.py [grid ...]     (320 -> 1 M triangles, 1000 -> 10 M)"""
import importlib
import sys
import time

import numpy as np

sys.path.insert(0, ".")
mts = importlib.import_module("mitsuba-renderer_amd")
_kp = mts.abi.KdParams(); _kp.exact_prim_threshold = 300
mts.Scene(mts.scenes.cornell_c5(sphere_subdiv=3), kd_params=_kp, gpu_binning=True)      # HIP runtime start-up, not part of any build
for grid in [int(a) for a in sys.argv[1:]] or [320, 1000]:
    sd = mts.scenes.cornell_c3(grid=grid)
    res = {}
    for dev in (False, True):
        t = time.time()
        sc = mts.Scene(sd, gpu_binning=dev)
        res[dev] = (time.time() - t, sc.arrays())
        print("grid %d: %s binning, flatten + build %.2f s" % (grid, "device" if dev else "host", res[dev][0]), flush=True)
    same = all(np.array_equal(res[False][1][k], res[True][1][k]) for k in ("kd_nodes", "kd_indices"))
    print("grid %d: %d nodes, trees identical: %s" % (grid, res[True][1]["kd_nodes"].shape[0], same), flush=True)
