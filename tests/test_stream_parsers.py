"""integration/streamparse.h on bytes: the field-order logic of the reference-side binding (integration/gpucommon.h) without
Mitsuba.  The streams are written by tests/mts_stream_writer.py, which follows each class's serialize(); the parsed parameter
blocks must equal what the library's own flattener (mtsgpu_flatten) builds for the same scene description, bit for bit."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import mts_stream_writer as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sp(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    so = str(tmp_path_factory.mktemp("harness") / "libstreamharness.so")
    # undefined-behaviour checks stay on in the harness (misaligned reads, shifts, overflow: the parsers take bytes as they come)
    subprocess.check_call(["g++", "-std=gnu++11", "-O1", "-Wall", "-Werror", "-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "integration"), os.path.join(ROOT, "tests", "stream_harness", "harness.cpp"), "-o", so])
    return C.CDLL(so)


def _call(fn, data, prec, *outs):
    msg = C.create_string_buffer(512)
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    rc = fn(buf, C.c_size_t(len(data)), prec, *outs, msg, C.c_size_t(512))
    return rc, msg.value.decode(errors="replace")


def _parse_bsdf(sp, data, prec=4):
    t = C.c_uint32(0)
    P = np.zeros(16, dtype=np.float32)
    rc, msg = _call(sp.sp_parse_bsdf, data, prec, C.byref(t), P.ctypes.data_as(C.POINTER(C.c_float)))
    return rc, msg, t.value, P


def _parse_block(sp, fn, data, n, prec=4):
    P = np.zeros(n, dtype=np.float32)
    rc, msg = _call(fn, data, prec, P.ctypes.data_as(C.POINTER(C.c_float)))
    return rc, msg, P


def _flat(mts, sd):
    return mts.Scene(sd).arrays()


@pytest.mark.parametrize("scene", ["c5", "next", "spheres", "envlit"])
@pytest.mark.parametrize("form", ["constants", "children", "shared", "double"])
def test_bsdf_blocks_from_bytes(sp, mts, scene, form):
    """every BSDF of the test scenes (all eight classes, the twosided adapter): Mitsuba's stream -> type word + block == the
    flattener's, with the textures as constructor constants (no parent), as addChild children (parent = the BSDF, a known
    id), as one texture object in both slots (the second reference is a bare id), and in a DOUBLE_PRECISION build"""
    sd = {"c5": mts.scenes.cornell_c5, "next": mts.scenes.next_rows, "spheres": mts.scenes.spheres, "envlit": mts.scenes.envlit}[scene]()
    fa = _flat(mts, sd)
    assert fa["bsdf_params"].shape[0] == len(sd.bsdf_type) > 0
    seen = set()
    for i, (t, P) in enumerate(zip(fa["bsdf_type"], fa["bsdf_params"])):
        btype, two = int(t) & 0xFF, bool(int(t) & mts.abi.BSDF_TWOSIDED)
        seen.add(btype)
        P = np.array(P, dtype=np.float32)
        if form == "shared":
            if btype not in (1, 3, 5, 6):
                continue
            # one texture in both slots: the blocks agree only if both slots hold the first slot's value
            a, b = {1: (2, 5), 3: (5, 8), 5: (5, 8), 6: (4, 7)}[btype]
            P = P.copy(); P[b:b + 3] = P[a:a + 3]
        s = W.Stream(8 if form == "double" else 4)
        W.bsdf(s, ("bsdf", i), btype, P, twosided=two, name="m%d" % i, tex_parent=(form == "children"), share_textures=(form == "shared"))
        rc, msg, gt, gP = _parse_bsdf(sp, s.bytes(), 8 if form == "double" else 4)
        assert rc == 0, msg
        assert gt == int(t) if form != "shared" else (gt & 0xFF) == btype
        assert np.array_equal(gP.view(np.uint32), P.view(np.uint32)), (scene, i, btype, gP, P)
    assert seen, scene


def test_every_bsdf_class_is_covered(mts):
    sds = [mts.scenes.cornell_c5(), mts.scenes.next_rows(), mts.scenes.spheres(), mts.scenes.envlit()]
    classes = {int(t) & 0xFF for sd in sds for t in sd.bsdf_type}
    assert classes == set(range(8))
    assert any(int(t) & mts.abi.BSDF_TWOSIDED for sd in sds for t in sd.bsdf_type)


def test_bsdf_errors_are_reported(sp, mts):
    P = np.zeros(16, dtype=np.float32); P[0:3] = 0.5
    s = W.Stream(); W.bsdf(s, "b", 0, P)
    good = s.bytes()
    rc, msg, _, _ = _parse_bsdf(sp, good[:-5])
    assert rc != 0 and "end of the serialized stream" in msg
    # a bitmap texture where a constant is expected (what a textured scene would send)
    s = W.Stream()
    def body(s):
        W.configurable(s); s.string("")
        s.ref("tex", "BitmapTexture", lambda s: W.configurable(s))
    s.ref("b", "Lambertian", body)
    rc, msg, _, _ = _parse_bsdf(sp, s.bytes())
    assert rc != 0 and "BitmapTexture" in msg and "only constant" in msg
    # a class that is not on the path
    s = W.Stream(); s.ref("b", "Ward", lambda s: (W.configurable(s), s.string("")))
    rc, msg, _, _ = _parse_bsdf(sp, s.bytes())
    assert rc != 0 and "Ward" in msg
    # twosided without a nested BRDF
    s = W.Stream(); s.ref("b", "TwoSidedBRDF", lambda s: (W.configurable(s), s.string(""), s.ref(None)))
    rc, msg, _, _ = _parse_bsdf(sp, s.bytes())
    assert rc != 0 and "nested" in msg


def _rigid(P_w2l_3x4=None, rot=None, pos=None):
    """4x4 world->luminaire and luminaire->world from a rotation (rows = luminaire axes in world space) and a position"""
    R = np.asarray(rot, dtype=np.float32).reshape(3, 3)
    p = np.asarray(pos, dtype=np.float32)
    l2w = np.eye(4, dtype=np.float32); l2w[:3, :3] = R.T; l2w[:3, 3] = p
    w2l = np.eye(4, dtype=np.float32); w2l[:3, :3] = R; w2l[:3, 3] = -(R @ p)
    return w2l, l2w


@pytest.mark.parametrize("prec", [4, 8])
def test_delta_luminaires_from_bytes(sp, mts, prec):
    """spot, directional and collimated luminaires of the `next_rows` scene: Luminaire::serialize + the class's own fields ->
    the block the flattener builds (spot: position from luminaireToWorld, cosines and transition width from configure();
    directional: the disk radius preprocess() left; collimated: both 3x4 matrices)"""
    sd = mts.scenes.next_rows()
    fa = _flat(mts, sd)
    kinds = {}
    for l, (t, P) in enumerate(zip(fa["lum_type"], fa["lum_params"])):
        P = np.array(P, dtype=np.float32)
        s = W.Stream(prec)
        if t == mts.abi.LUM_SPOT:
            w2l, l2w = _rigid(rot=P[10:19], pos=P[3:6])
            W.spot(s, "l", w2l, l2w, P[0:3], P[19], P[8])
            rc, msg, g = _parse_block(sp, sp.sp_parse_spot, s.bytes(), mts.abi.LUM_NPARAMS, prec)
        elif t == mts.abi.LUM_DIRECTIONAL:
            W.directional(s, "l", np.eye(4), np.eye(4), P[3:6], P[0:3], (0.0, 0.0, 0.0), P[6])
            rc, msg, g = _parse_block(sp, sp.sp_parse_directional, s.bytes(), mts.abi.LUM_NPARAMS, prec)
        elif t == mts.abi.LUM_COLLIMATED:
            w2l = np.eye(4, dtype=np.float32); w2l[:3, :] = P[4:16].reshape(3, 4)
            l2w = np.eye(4, dtype=np.float32); l2w[:3, :] = P[16:28].reshape(3, 4)
            W.collimated(s, "l", w2l, l2w, P[0:3], P[3])
            rc, msg, g = _parse_block(sp, sp.sp_parse_collimated, s.bytes(), mts.abi.LUM_NPARAMS, prec)
        else:
            continue
        assert rc == 0, msg
        kinds[int(t)] = kinds.get(int(t), 0) + 1
        if prec == 4:
            assert np.array_equal(g.view(np.uint32), P.view(np.uint32)), (int(t), g, P)
        else:
            # a double-precision build evaluates cos() and the reciprocal in double before the block is rounded to float
            assert np.allclose(g, P, rtol=2e-7, atol=0) and np.array_equal(g[[0, 1, 2, 3, 4, 5, 8, 19]], P[[0, 1, 2, 3, 4, 5, 8, 19]])
    assert set(kinds) == {mts.abi.LUM_SPOT, mts.abi.LUM_DIRECTIONAL, mts.abi.LUM_COLLIMATED}


def test_spot_projection_texture_is_refused(sp, mts):
    w2l, l2w = _rigid(rot=np.eye(3), pos=(0, 1, 0))
    for kw in (dict(texture=(0.5, 1.0, 1.0)), dict(texture_class="BitmapTexture")):
        s = W.Stream(); W.spot(s, "l", w2l, l2w, (1, 1, 1), 0.2, 0.3, **kw)
        rc, msg, _ = _parse_block(sp, sp.sp_parse_spot, s.bytes(), mts.abi.LUM_NPARAMS)
        assert rc != 0 and "projection textures" in msg
    s = W.Stream(); W.collimated(s, "l", w2l, l2w, (1, 1, 1), 0.1)
    rc, msg, _ = _parse_block(sp, sp.sp_parse_spot, s.bytes(), mts.abi.LUM_NPARAMS)
    assert rc != 0 and "expected a SpotLuminaire" in msg


def test_envmap_header_from_bytes(sp, mts):
    """the environment map of the `envlit` scene: intensity scale, bounding sphere (as preprocess() left it), both rotations,
    and where the EXR file's bytes start"""
    sd = mts.scenes.envlit()
    fa = _flat(mts, sd)
    l = int(np.nonzero(fa["lum_type"] == mts.abi.LUM_ENVMAP)[0][0])
    P = np.array(fa["lum_params"][l], dtype=np.float32)
    w2l = np.eye(4, dtype=np.float32); w2l[:3, :3] = P[7:16].reshape(3, 3)
    l2w = np.eye(4, dtype=np.float32); l2w[:3, :3] = P[16:25].reshape(3, 3)
    exr = bytes(range(256)) * 3
    s = W.Stream(); W.envmap(s, "l", w2l, l2w, P[0], "/data/envmap.exr", P[3:7], exr)
    data = s.bytes()
    g = np.zeros(mts.abi.LUM_NPARAMS, dtype=np.float32)
    off, size = C.c_uint64(0), C.c_uint32(0)
    rc, msg = _call(sp.sp_parse_envmap, data, 4, g.ctypes.data_as(C.POINTER(C.c_float)), C.byref(off), C.byref(size))
    assert rc == 0, msg
    assert size.value == len(exr) and data[off.value:off.value + size.value] == exr
    keep = np.r_[0, 3:25]
    assert np.array_equal(g[keep].view(np.uint32), P[keep].view(np.uint32)), (g, P)
    rc, msg = _call(sp.sp_parse_envmap, data[:-10], 4, g.ctypes.data_as(C.POINTER(C.c_float)), C.byref(off), C.byref(size))
    assert rc != 0 and "truncated" in msg


@pytest.mark.parametrize("prec", [4, 8])
def test_spheres_from_bytes(sp, mts, prec):
    """every `sphere` of the `spheres` scene behind Shape::serialize with its nested BSDF (each BSDF class the scene uses) and,
    for the emitter, its nested area luminaire: centre, radius, orientation, both transforms, 1 / area"""
    sd = mts.scenes.spheres()
    fa = _flat(mts, sd)
    n = 0
    for sidx in np.nonzero(fa["shape_type"] == mts.abi.SHAPE_SPHERE)[0]:
        SP = np.array(fa["shape_params"][sidx], dtype=np.float32)
        o2w = np.eye(4, dtype=np.float32); o2w[:3, :3] = SP[5:14].reshape(3, 3); o2w[:3, 3] = SP[0:3]
        w2o = np.eye(4, dtype=np.float32); w2o[:3, :3] = SP[14:23].reshape(3, 3); w2o[:3, 3] = -SP[0:3]
        b = int(fa["shape_bsdf"][sidx]); l = int(fa["shape_lum"][sidx])
        bargs = None
        if b >= 0:
            t = int(fa["bsdf_type"][b])
            bargs = (t & 0xFF, np.array(fa["bsdf_params"][b], dtype=np.float32), bool(t & mts.abi.BSDF_TWOSIDED))
        s = W.Stream(prec)
        W.sphere(s, ("shape", int(sidx)), o2w, w2o, SP[3], SP[0:3], SP[4] != 0, bsdf_args=bargs,
                 lum_intensity=None if l < 0 else fa["lum_params"][l][0:3])
        rc, msg, g = _parse_block(sp, sp.sp_parse_sphere, s.bytes(), mts.abi.SHAPE_NPARAMS, prec)
        assert rc == 0, msg
        if prec == 4:
            assert np.array_equal(g.view(np.uint32), SP.view(np.uint32)), (int(sidx), g, SP)
        else:
            assert np.array_equal(g[:23].view(np.uint32), SP[:23].view(np.uint32)) and np.isclose(g[23], SP[23], rtol=2e-7)
        n += 1
    assert n == 8 and (fa["shape_lum"][fa["shape_type"] == mts.abi.SHAPE_SPHERE] >= 0).sum() == 1
    rc, msg, _ = _parse_block(sp, sp.sp_parse_sphere, s.bytes()[:40], mts.abi.SHAPE_NPARAMS, prec)
    assert rc != 0


def test_parsers_survive_truncated_and_corrupted_streams(sp, mts):
    """every prefix of a valid stream and a few hundred corrupted copies: the parsers answer (an error or a block), they never
    read past the buffer (the harness is built with -fsanitize=undefined; lengths and ids come from the bytes)"""
    rng = np.random.RandomState(3)
    sd = mts.scenes.spheres()
    fa = _flat(mts, sd)
    streams = []
    for i, (t, P) in enumerate(zip(fa["bsdf_type"], fa["bsdf_params"])):
        s = W.Stream(); W.bsdf(s, ("b", i), int(t) & 0xFF, np.array(P, dtype=np.float32), twosided=bool(int(t) & mts.abi.BSDF_TWOSIDED), name="x")
        streams.append(("bsdf", s.bytes()))
    w2l, l2w = _rigid(rot=np.eye(3), pos=(0, 1, 0))
    s = W.Stream(); W.spot(s, "l", w2l, l2w, (1, 1, 1), 0.2, 0.3); streams.append(("spot", s.bytes()))
    s = W.Stream(); W.envmap(s, "l", w2l, l2w, 1.0, "x.exr", (0, 0, 0, 1), b"\x01" * 64); streams.append(("envmap", s.bytes()))
    s = W.Stream(); W.sphere(s, "s", np.eye(4), np.eye(4), 1.0, (0, 0, 0), False, bsdf_args=(0, np.full(16, 0.5, dtype=np.float32), False), lum_intensity=(1, 1, 1))
    streams.append(("sphere", s.bytes()))
    def run(kind, data):
        if kind == "bsdf": return _parse_bsdf(sp, data)[0]
        if kind == "spot": return _parse_block(sp, sp.sp_parse_spot, data, mts.abi.LUM_NPARAMS)[0]
        if kind == "sphere": return _parse_block(sp, sp.sp_parse_sphere, data, mts.abi.SHAPE_NPARAMS)[0]
        g = np.zeros(mts.abi.LUM_NPARAMS, dtype=np.float32); off, size = C.c_uint64(0), C.c_uint32(0)
        return _call(sp.sp_parse_envmap, data, 4, g.ctypes.data_as(C.POINTER(C.c_float)), C.byref(off), C.byref(size))[0]
    for kind, data in streams:
        assert run(kind, data) == 0
        for n in range(1, len(data)):
            rc = run(kind, data[:n])
            assert rc != 0 or kind == "sphere"           # a sphere's fields are read from the END: a cut in the middle may still parse
        for _ in range(60):
            b = bytearray(data)
            for _ in range(rng.randint(1, 4)):
                b[rng.randint(len(b))] = rng.randint(256)
            run(kind, bytes(b))                          # any answer, no crash
