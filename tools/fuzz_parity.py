"""Parity sweep over random scenes (mitsuba-renderer_amd/scenes.py::fuzz): HIP path vs the oracle, bit for bit.
usage: python tools/fuzz_parity.py [first_seed] [count]     (test infrastructure: uses oracle/)"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _pkgload
mts = _pkgload.load()
from oracle import orc

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bad_total = 0
for seed in range(first, first + count):
    sd = mts.scenes.fuzz(seed, n_meshes=6 + seed % 25)
    kp = mts.abi.KdParams()
    if seed % 3 == 0: kp.exact_prim_threshold = 32 + seed % 100
    scene = mts.Scene(sd, kd_params=kp, gpu_binning=(seed % 2 == 0), gpu_exact=(seed % 4 < 2)); oscene = orc.FlatScene(sd, kd_params=kp)
    a, b = scene.arrays(), oscene.arrays()
    same = all(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) for k in ("kd_nodes", "kd_indices", "triaccel", "vtx_nrm"))
    sampler = ["independent", "ldsampler", "stratified", "halton", "hammersley"][seed % 5]
    kind = {"independent": 0, "ldsampler": 1, "halton": 2, "hammersley": 3, "stratified": 4}[sampler]
    direct = seed % 7 == 0
    nl, nb = (1 + seed % 4, seed % 3) if direct and kind in (0, 1, 4) else (1, 1)
    it = mts.MIDirectIntegrator(luminaireSamples=nl, bsdfSamples=nb) if direct else \
        mts.MIPathTracer(maxDepth=sd.max_depth, rrDepth=sd.rr_depth, strictNormals=bool(seed & 1))
    # every 15th seed: the table samplers above 512 samples per pixel (k_ld_scout / k_ld_apply_lds) on a frame of 8 x 6 pixels
    many = seed % 15 == 1 and kind in (1, 4) and not direct
    spp = (1024 if kind == 1 else 900) if many else (9 if kind == 4 else 8)
    W, H = (8, 6) if many else (40, 30)
    cam = mts.PerspectiveCamera.for_description(sd, W, H); ocam = orc.make_camera(sd, W, H)
    it.preprocess(scene, cam, sampler=sampler, sampleCount=spp, seed=seed)
    op = orc.render_params(sd.max_depth, rr_depth=sd.rr_depth, strict_normals=int(seed & 1), sampler=kind, spp=spp, seed=seed,
                           integrator="direct" if direct else "path", luminaire_samples=nl, bsdf_samples=nb)
    # who drives the bounces: 0 the default for a frame this small (device-driven, all material queues in one launch), 1 the host
    # (one read-back per bounce, exact grids, dynamically claimed batches, one shading launch per BSDF type), 2 the host over
    # several ragged passes, 3 device-driven with one shading launch per BSDF type
    # 4 / 5 (round 6): the host with the any-hit launch of a bounce BEHIND / in front of the next closest-hit launch on a second stream
    # 6: the host with ONE traversal launch per bounce (k_trace_pair); 7: the host with the mailbox-free closest-hit kernel (tied rays traced again)
    drive = (seed // 5) % 8
    if drive == 1: it.set_tuning(sync_free=0)
    elif drive == 2: it.set_tuning(sync_free=0); it.set_options(max_paths=spp * (W * H // 3 + 1))
    elif drive == 3: it.set_tuning(sync_free=1, shade_fused=0)
    elif drive == 4: it.set_tuning(sync_free=0, overlap=2)
    elif drive == 5: it.set_tuning(sync_free=0, overlap=1); it.set_options(max_paths=spp * (W * H // 2 + 1))
    elif drive == 6: it.set_tuning(sync_free=0, merged=1)
    elif drive == 7: it.set_tuning(sync_free=0, mailbox_free=1)
    assert it.render()
    film = it.film()
    ofilm, _ = orc.render(oscene.scene, ocam, op)
    nbad = int((film.view(np.uint32) != ofilm.view(np.uint32)).any(axis=2).sum())
    bad_total += nbad + (0 if same else 1)
    print("seed %d: %d tris, %s%s%s drive %d, tree %s, %d of %d pixels differ" % (seed, sd.n_tris, sampler, " %d spp" % spp if many else "", " direct(%d,%d)" % (nl, nb) if direct else "", drive,
                                                                       "same" if same else "DIFFERENT", nbad, W * H), flush=True)
print("TOTAL mismatches:", bad_total)
