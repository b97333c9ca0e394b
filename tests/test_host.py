"""Host-side logic of the product (no GPU needed): ABI surface, flattening, kd-tree builder, error paths."""
import ctypes as C
import re
import os
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    a = np.atleast_1d(np.asarray(a)); b = np.atleast_1d(np.asarray(b))
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def test_library_exports_every_declared_symbol(mts):
    L = mts.lib()
    header = open(os.path.join(ROOT, "include", "mtsgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(mtsgpu_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), "libmtsgpu.so does not export %s" % name
    assert sorted(mts.EXPORTS) == declared
    assert L.mtsgpu_abi_version() == mts.abi.ABI_VERSION


def test_ctypes_layout_matches_the_compiled_structs(mts):
    L, a = mts.lib(), mts.abi
    for which, typ in enumerate([a.Scene, a.Camera, a.Stats, a.Mesh, a.SceneDesc, a.KdParams]):
        assert L.mtsgpu_abi_sizeof(which) == C.sizeof(typ), typ.__name__


def test_no_gpu_means_a_loud_error_not_a_fallback(mts):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(mts.MtsGpuError) as e:
        mts.MIPathTracer(maxDepth=4)
    assert "no HIP device" in str(e.value) or "CPU fallback" in str(e.value)
    # the device binning phase of the kd-tree build does not quietly run on the host either
    kp = mts.abi.KdParams(); kp.exact_prim_threshold = 300
    with pytest.raises(mts.MtsGpuError) as e:
        mts.Scene(mts.scenes.cornell_c5(sphere_subdiv=2), kd_params=kp, gpu_binning=True)
    assert "no HIP device" in str(e.value)
    # ... nor does the device exact phase
    with pytest.raises(mts.MtsGpuError) as e:
        mts.Scene(mts.scenes.cornell_c1(), gpu_exact=True)
    assert "no HIP device" in str(e.value)


@pytest.mark.parametrize("maker", [
    lambda s: s.cornell_c1(),
    lambda s: s.cornell_c3(grid=12, sphere_subdiv=2),
    lambda s: s.cornell_c5(sphere_subdiv=2),
    lambda s: s.cornell_c3(grid=40, sphere_subdiv=3),
])
def test_flatten_matches_oracle_bit_for_bit(mts, orc, maker):
    """normals, area CDFs, TriAccel table, SAH kd-tree (nodes + indices), enlarged AABB, bsphere"""
    sd = maker(mts.scenes)
    A, B = orc.FlatScene(sd).arrays(), mts.Scene(sd).arrays()
    assert sorted(A) == sorted(B)
    bad = [k for k in A if not _same(A[k], B[k])]
    assert not bad, bad


def test_min_max_binning_phase_matches_oracle(mts, orc):
    """force the > exactPrimThreshold code path (binning, tight boxes, parallel sub-tree jobs) on a small scene"""
    sd = mts.scenes.cornell_c3(grid=24, sphere_subdiv=2)
    kp = mts.abi.KdParams(); kp.exact_prim_threshold = 500
    A, B = orc.FlatScene(sd, kp).arrays(), mts.Scene(sd, kp).arrays()
    assert _same(A["kd_nodes"], B["kd_nodes"]) and _same(A["kd_indices"], B["kd_indices"])
    kp1 = mts.abi.KdParams(); kp1.exact_prim_threshold = 500; kp1.n_threads = 1
    C1 = mts.Scene(sd, kp1).arrays()
    assert _same(C1["kd_nodes"], B["kd_nodes"]), "tree depends on the number of build threads"


def test_kdtree_is_structurally_valid(mts):
    sd = mts.scenes.cornell_c3(grid=16, sphere_subdiv=2)
    sc = mts.Scene(sd)
    A = sc.arrays()
    nodes, idx = A["kd_nodes"], A["kd_indices"]
    seen = np.zeros(sd.n_tris, dtype=bool)
    stack, visited, depth_max = [(0, 1)], 0, 0
    while stack:
        n, d = stack.pop()
        visited += 1; depth_max = max(depth_max, d)
        a, b = int(nodes[n, 0]), int(nodes[n, 1])
        if a & 0x80000000:
            seen[idx[(a & 0x7FFFFFFF):b]] = True
        else:
            assert (a & 3) < 3
            left = n + ((a & 0x3FFFFFFC) >> 2)
            stack += [(left, d + 1), (left + 1, d + 1)]
    assert visited == len(nodes) and seen.all()
    assert depth_max <= 48                       # MTS_KD_MAXDEPTH
    st = sc.kdstats()
    assert st["inner"] + st["leaf"] == len(nodes) and st["indices"] == len(idx)


def test_camera_matches_oracle(mts, orc):
    sd = mts.scenes.cornell_c1()
    for (w, h) in ((256, 256), (320, 200), (200, 320)):
        a = mts.PerspectiveCamera.for_description(sd, w, h).c
        b = orc.make_camera(sd, w, h)
        assert bytes(a) == bytes(b)
    # raster centre maps to the optical axis
    m = np.array(list(mts.PerspectiveCamera.for_description(sd, 256, 256).c.raster_to_camera)).reshape(4, 4)
    p = m @ np.array([128, 128, 0, 1.0]); p = p[:3] / p[3]
    assert abs(p[0]) < 1e-6 and abs(p[1]) < 1e-6
    # orthographic camera (src/cameras/orthographic.cpp): both hosts agree; raster corners map to +-scale
    sd.camera = dict(origin=(0.0, 1.0, 3.4), target=(0.0, 1.0, 0.0), up=(0.0, 1.0, 0.0), ortho_scale=(1.1, 1.1))
    for (w, h) in ((256, 256), (320, 200), (200, 320)):
        a = mts.PerspectiveCamera.for_description(sd, w, h).c
        b = orc.make_camera(sd, w, h)
        assert bytes(a) == bytes(b) and a.kind == 1
    c = mts.PerspectiveCamera.for_description(sd, 256, 256).c
    r2c = np.array(list(c.raster_to_camera)).reshape(4, 4); c2w = np.array(list(c.camera_to_world)).reshape(4, 4)
    q = c2w @ (r2c @ np.array([0, 0, 0, 1.0]))
    assert abs(abs(q[0]) - 1.1) < 1e-5 and abs(abs(q[1] - 1.0) - 1.1) < 1e-5


def test_flatten_rejects_bad_input(mts):
    sd = mts.scenes.cornell_c1()
    sd.meshes[0].triangles[0, 0] = 99
    with pytest.raises(mts.MtsGpuError):
        mts.Scene(sd)
    sd = mts.scenes.cornell_c1()
    sd.meshes[0].bsdf = 42
    with pytest.raises(mts.MtsGpuError):
        mts.Scene(sd)


def test_tiles_of_rank_partition(mts):
    W, H = 100, 70
    parts = [mts.filmreduce.tiles_of_rank(W, H, 32, r, 3) for r in range(3)]
    allp = np.concatenate(parts)
    assert len(allp) == W * H and len(np.unique(allp)) == W * H


def test_tile_ownership_is_a_balanced_2d_lattice(mts, orc):
    """mtsgpu_set_tiles: tile (tx, ty) belongs to part morton(tx, ty) % n.  At 1024^2 / 32 every part of 2, 4 or 8 owns
    the same number of tiles and every 4 x 2 group of tiles holds all 8 parts; the oracle's tile renderer uses the same
    rule (one sample per pixel, box filter: the weight channel marks the owned pixels)"""
    fr = mts.filmreduce
    assert [fr.tile_morton(x, y) for x, y in ((0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (3, 5))] == [0, 1, 2, 3, 4, 39]
    for n in (2, 4, 8):
        owner = np.array([[fr.tile_morton(x, y) % n for x in range(32)] for y in range(32)])
        assert set(np.bincount(owner.reshape(-1), minlength=n)) == {1024 // n}
    owner = np.array([[fr.tile_morton(x, y) % 8 for x in range(32)] for y in range(32)])
    for y in range(0, 32, 2):
        for x in range(0, 32, 4):
            assert sorted(owner[y:y + 2, x:x + 4].reshape(-1)) == list(range(8))
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    W, H, bs = 70, 50, 16
    cam = orc.make_camera(sd, W, H)
    prm = orc.render_params(1, sampler=0, spp=1)
    box = orc.tabulate_filter("box")
    for r in range(3):
        film, _ = orc.render_tiles(fs.scene, cam, prm, box, block_size=bs, part=r, n_parts=3)
        mine = np.flatnonzero(film[..., 4].reshape(-1) > 0)
        assert np.array_equal(np.sort(fr.tiles_of_rank(W, H, bs, r, 3)), mine)


def test_bench_helpers():
    """bench.py without a GPU: the CPU count it reports honours the cgroup quota, the kernel-source hash is stable"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    n, visible, quota = bench.usable_cpus()
    assert 1 <= n <= visible == len(os.sched_getaffinity(0))
    if quota is not None:
        assert n <= max(1, int(quota + 0.5))
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    assert bench.free_port() > 0
    st = dict(rays_closest=10, rays_shadow=5, n_inner=100, n_leaf=20, n_idx=30, n_tri_tested=30)
    assert bench.algorithmic_bytes(st) == 8 * 100 + 8 * 20 + 52 * 30 + 48 * 15


def test_serialized_mesh_loader(mts, tmp_path):
    """mtsgpu_load_serialized == TriMesh::TriMesh(Stream *, int) (src/librender/trimesh.cpp:156-236): on files written
    by an independent Python implementation of the format (tests/serialized_io.py) and, where the reference checkout
    is present (not on the GPU box), on its own data/blender/mitsuba/matpreview/matpreview.serialized, whose counts
    and hashes are recorded here"""
    import hashlib
    import os
    import serialized_io as sio
    path = "/root/reference/data/blender/mitsuba/matpreview/matpreview.serialized"
    golden = {0: (2078, 3936, "ebdc487819cb"), 1: (7529, 14288, "f67fc789ac10"), 2: (25, 32, "7aa5b07f5656")}
    for i in range(3 if os.path.exists(path) else 0):
        assert sio.shape_count(open(path, "rb").read()) == 3
        ref = sio.read(path, i)
        m = mts.load_serialized(path, i)
        assert np.array_equal(ref["positions"].view(np.uint32), m.positions.view(np.uint32))
        assert np.array_equal(ref["triangles"], m.triangles)
        assert np.array_equal(ref["normals"].view(np.uint32), m.normals.view(np.uint32)) and not m.face_normals
        nv, nt, sha = golden[i]
        assert (len(m.positions), len(m.triangles)) == (nv, nt)
        assert hashlib.sha1(m.positions.tobytes() + m.triangles.tobytes()).hexdigest()[:12] == sha
    # double precision, texture coordinates and colours (skipped), face normals; several shapes per file
    rng = np.random.RandomState(4)
    meshes = []
    for k in range(3):
        nv = 50 + 17 * k
        meshes.append(dict(positions=rng.randn(nv, 3), triangles=rng.randint(0, nv, (80 + k, 3)),
                           normals=rng.randn(nv, 3) if k != 1 else None, texcoords=rng.rand(nv, 2) if k == 0 else None,
                           colors=rng.rand(nv, 3) if k == 2 else None, face_normals=(k == 1)))
    for double in (False, True):
        f = str(tmp_path / ("m%d.serialized" % double))
        sio.write(f, meshes, double=double)
        for k, src in enumerate(meshes):
            m = mts.load_serialized(f, k)
            assert np.array_equal(m.positions, np.asarray(src["positions"]).astype(np.float32))
            assert np.array_equal(m.triangles, np.asarray(src["triangles"], dtype=np.uint32))
            assert (m.normals is None) == (src["normals"] is None) and m.face_normals == bool(src["face_normals"])
            if m.normals is not None:
                assert np.array_equal(m.normals, np.asarray(src["normals"]).astype(np.float32))
    # malformed input is refused with an error, never read out of bounds
    f = str(tmp_path / "m0.serialized")
    good = open(f, "rb").read()
    bad = {"index": (good, 7), "header": (b"\x04\x1c" + good[2:], 0), "version": (good[:2] + b"\x02\x00" + good[4:], 0),
           "truncated": (good[:200], 0), "garbage": (good[:4] + b"\x00" * 64, 0), "empty": (b"", 0)}
    for name, (blob, idx) in bad.items():
        g = str(tmp_path / ("bad_%s.serialized" % name))
        open(g, "wb").write(blob)
        with pytest.raises(mts.MtsGpuError):
            mts.load_serialized(g, idx)
    with pytest.raises(mts.MtsGpuError):
        mts.load_serialized(str(tmp_path / "does_not_exist.serialized"), 0)
    # an index that points outside the vertex array
    broken = [dict(positions=rng.randn(4, 3), triangles=np.array([[0, 1, 9]]))]
    sio.write(str(tmp_path / "oob.serialized"), broken)
    with pytest.raises(mts.MtsGpuError):
        mts.load_serialized(str(tmp_path / "oob.serialized"), 0)


def test_library_is_built_from_the_sources_on_disk(mts):
    """mtsgpu_source_hash() (stamped in by csrc/Makefile, csrc/stamp.cpp) equals the hash of the sources: the binary that
    ships to the GPU box is not a stale one"""
    assert mts.lib().mtsgpu_source_hash().decode() == mts.source_hash()


REFERENCE_INCLUDE = "/root/reference/include"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_INCLUDE), reason="the reference checkout is only present in the build container")
@pytest.mark.parametrize("plugin", ["gpupath", "gpudirect"])
def test_plugin_sources_pass_the_compiler_front_end(plugin):
    """integration/<plugin>.cpp + gpucommon.h type-check against Mitsuba's own declarations (Integrator, Scene, ShapeKDTree,
    ImageBlock, Sampler, InstanceManager, ...: include/mitsuba/render/integrator.h:45-88, core/cobject.h:79-87) and against
    include/mtsgpu.h.  Syntax only: nothing is compiled to code, nothing of the reference is built or run.  The Boost headers
    Mitsuba's headers pull in are absent from this image; tests/plugin_syntax/ holds stand-ins for this check alone."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    cmd = ["g++", "-std=gnu++17", "-fsyntax-only", "-Wall", "-DSINGLE_PRECISION", "-include", "unistd.h",
           "-I" + os.path.join(root, "tests", "plugin_syntax"), "-isystem", REFERENCE_INCLUDE, "-I" + os.path.join(root, "include"),
           os.path.join(root, "integration", plugin + ".cpp")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert r.returncode == 0, r.stdout[-4000:]
    ours = [l for l in r.stdout.splitlines() if "integration/" in l and "warning" in l]
    assert not ours, "\n".join(ours)
