"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product package."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import _pkgload  # noqa: E402

abi = _pkgload.load().abi


class RenderParams(C.Structure):
    _fields_ = [("max_depth", C.c_int), ("rr_depth", C.c_int), ("strict_normals", C.c_int),
                ("sampler_kind", C.c_int), ("spp", C.c_uint32), ("ld_depth", C.c_int),
                ("seed", C.c_uint64), ("n_threads", C.c_int),
                ("integrator", C.c_int), ("luminaire_samples", C.c_int), ("bsdf_samples", C.c_int)]


class TabFilter(C.Structure):
    _fields_ = [("size_x", C.c_float), ("size_y", C.c_float), ("values", (C.c_float * 16) * 16)]


class TraceCounts(C.Structure):
    _fields_ = [("n_inner", C.c_uint64), ("n_leaf", C.c_uint64), ("n_idx", C.c_uint64), ("n_tri_tested", C.c_uint64)]


class Random(C.Structure):
    _fields_ = [("mt", C.c_uint64 * 312), ("mti", C.c_int)]


_variant = "liboracle.so"


def use_native_build():
    """bench.py's cpu_baseline leg: the same sources with -O3 -march=native, compiled on the machine that runs it
    (oracle/Makefile).  Must be called before the first lib()."""
    global _variant
    assert _lib is None, "the oracle library is already loaded"
    _variant = "liboracle_native.so"


def build(force=False):
    so = os.path.join(_HERE, _variant)
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "mtsgpu.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", _variant], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    f32p, u32p = abi.f32p, abi.u32p
    L.orc_random_seed.argtypes = [C.POINTER(Random), C.c_uint64]
    L.orc_random_seed_from.argtypes = [C.POINTER(Random), C.POINTER(Random)]
    L.orc_random_next_ulong.argtypes = [C.POINTER(Random)]; L.orc_random_next_ulong.restype = C.c_uint64
    L.orc_random_next_float.argtypes = [C.POINTER(Random)]; L.orc_random_next_float.restype = C.c_float
    L.orc_random_next_size.argtypes = [C.POINTER(Random), C.c_uint64]; L.orc_random_next_size.restype = C.c_uint64
    L.orc_random_shuffle_u32.argtypes = [C.POINTER(Random), u32p, C.c_size_t]
    L.orc_keyed_init.argtypes = [C.c_uint64] * 3; L.orc_keyed_init.restype = C.c_uint64
    L.orc_keyed_next.argtypes = [C.POINTER(C.c_uint64)]; L.orc_keyed_next.restype = C.c_uint64
    L.orc_vdc_bits.argtypes = [C.c_uint32, C.c_uint32]; L.orc_vdc_bits.restype = C.c_uint32
    L.orc_sobol2_bits.argtypes = [C.c_uint32, C.c_uint32]; L.orc_sobol2_bits.restype = C.c_uint32
    L.orc_u32_to_unit.argtypes = [C.c_uint32]; L.orc_u32_to_unit.restype = C.c_float
    L.orc_ld_generate_mt.argtypes = [C.POINTER(Random), C.c_uint32, C.c_int, f32p, f32p]
    L.orc_ld_generate_keyed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, f32p, f32p]
    L.orc_radical_inverse.argtypes = [C.c_int, C.c_uint64]; L.orc_radical_inverse.restype = C.c_float
    L.orc_radical_inverse_incremental.argtypes = [C.c_int, C.c_float]; L.orc_radical_inverse_incremental.restype = C.c_float
    L.orc_atan2f.argtypes = [C.c_float, C.c_float]; L.orc_atan2f.restype = C.c_float
    for n in ("orc_sinf", "orc_cosf", "orc_expf", "orc_logf", "orc_atanf", "orc_pow4f"):
        getattr(L, n).argtypes = [C.c_float]; getattr(L, n).restype = C.c_float
    L.orc_square_to_sphere.argtypes = [f32p, f32p]
    L.orc_square_to_hemisphere_psa.argtypes = [f32p, f32p]
    L.orc_square_to_triangle.argtypes = [f32p, f32p]
    L.orc_coordinate_system.argtypes = [f32p, f32p, f32p]
    L.orc_fresnel_dielectric.argtypes = [C.c_float] * 4; L.orc_fresnel_dielectric.restype = C.c_float
    L.orc_fresnel.argtypes = [C.c_float] * 3; L.orc_fresnel.restype = C.c_float
    L.orc_fresnel_conductor.argtypes = [C.c_float, f32p, f32p, f32p]
    L.orc_clipped_aabb.argtypes = [f32p] * 7; L.orc_clipped_aabb.restype = C.c_int
    L.orc_triaccel_load.argtypes = [f32p, f32p, f32p, u32p]; L.orc_triaccel_load.restype = C.c_int
    L.orc_triaccel_intersect.argtypes = [u32p, f32p, f32p, C.c_float, C.c_float, f32p, f32p, f32p]
    L.orc_triaccel_intersect.restype = C.c_int
    L.orc_flatten.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.KdParams), C.POINTER(C.c_void_p)]
    L.orc_flatten.restype = C.c_int
    L.orc_flat_scene_get.argtypes = [C.c_void_p]; L.orc_flat_scene_get.restype = C.POINTER(abi.Scene)
    L.orc_flat_scene_free.argtypes = [C.c_void_p]
    L.orc_flat_scene_kdstats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.orc_make_camera.argtypes = [f32p, f32p, f32p, C.c_float, C.c_int, C.c_int, C.POINTER(abi.Camera)]
    L.orc_make_camera_ortho.argtypes = [f32p, f32p, f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(abi.Camera)]
    L.orc_trace_rays.argtypes = [C.POINTER(abi.Scene), f32p, C.c_uint32, C.c_int, u32p, C.POINTER(TraceCounts)]
    L.orc_render_rect.argtypes = [C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(RenderParams),
                                  C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.POINTER(abi.Stats)]
    L.orc_li_samples.argtypes = [C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(RenderParams),
                                 u32p, C.c_uint32, f32p]
    L.orc_sampler_values.argtypes = [C.POINTER(RenderParams), C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, f32p]
    L.orc_render_rect_mt.argtypes = [C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(RenderParams),
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_tabulate_filter.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(TabFilter)]
    L.orc_render_tiles.argtypes = [C.POINTER(abi.Scene), C.POINTER(abi.Camera), C.POINTER(RenderParams), C.POINTER(TabFilter),
                                   C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.POINTER(abi.Stats)]
    L.orc_chord_rays.argtypes = [f32p, C.c_float, C.c_uint32, f32p]
    L.orc_luminaire_sample.argtypes = [C.POINTER(abi.Scene), C.c_int, f32p, f32p, f32p]
    L.orc_luminaire_pdf.argtypes = [C.POINTER(abi.Scene), C.c_int, f32p, f32p, f32p, f32p]; L.orc_luminaire_pdf.restype = C.c_float
    L.orc_bsdf_f.argtypes = [C.c_uint32, f32p, f32p, f32p, f32p]
    L.orc_bsdf_pdf.argtypes = [C.c_uint32, f32p, f32p, f32p]; L.orc_bsdf_pdf.restype = C.c_float
    L.orc_bsdf_eval.argtypes = [C.c_uint32, f32p, C.c_int, C.c_uint32, f32p, f32p]; L.orc_bsdf_eval.restype = None
    L.orc_bsdf_sample.argtypes = [C.c_uint32, f32p, f32p, f32p, f32p, f32p, u32p, f32p]
    _lib = L
    return L


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class FlatScene:
    """orc_flatten() result"""

    def __init__(self, desc, kd_params=None):
        self._desc, self._keep = desc.to_ctypes()
        self._h = C.c_void_p()
        kp = kd_params if kd_params is not None else abi.KdParams()
        rc = lib().orc_flatten(C.byref(self._desc), C.byref(kp), C.byref(self._h))
        if rc != 0:
            raise RuntimeError("orc_flatten failed: %d" % rc)
        self.scene = lib().orc_flat_scene_get(self._h)

    @property
    def sc(self):
        return self.scene.contents

    def arrays(self):
        return abi.scene_arrays(self.sc)

    def kdstats(self):
        out = (C.c_double * 6)()
        lib().orc_flat_scene_kdstats(self._h, out)
        return list(out)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_flat_scene_free(self._h)
            self._h = None


def make_camera(desc, width, height):
    cam = abi.Camera()
    c = desc.camera
    if "ortho_scale" in c:                       # <camera type="orthographic"> with toWorld = lookAt * scale
        sx, sy = c["ortho_scale"]
        lib().orc_make_camera_ortho(abi.ptr(_f(c["origin"]), abi.f32p), abi.ptr(_f(c["target"]), abi.f32p),
                                    abi.ptr(_f(c["up"]), abi.f32p), C.c_float(sx), C.c_float(sy), width, height, C.byref(cam))
        return cam
    lib().orc_make_camera(abi.ptr(_f(c["origin"]), abi.f32p), abi.ptr(_f(c["target"]), abi.f32p),
                          abi.ptr(_f(c["up"]), abi.f32p), C.c_float(c["fov"]), width, height, C.byref(cam))
    if c.get("aperture", 0.0) > 0:
        cam.aperture_radius = c["aperture"]
        cam.focus_depth = c.get("focus", cam.far_clip)
    return cam


def render_params(max_depth, rr_depth=10, strict_normals=0, sampler=abi.SAMPLER_INDEPENDENT_KEYED,
                  spp=4, ld_depth=3, seed=0x5EED, n_threads=0, integrator="path", luminaire_samples=1, bsdf_samples=1):
    return RenderParams(max_depth, rr_depth, strict_normals, sampler, spp, ld_depth, seed, n_threads,
                        {"path": 0, "direct": 1}[integrator], luminaire_samples, bsdf_samples)


def render(scene_ptr, cam, params, rect=None):
    W, H = cam.width, cam.height
    film = np.zeros((H, W, 5), dtype=np.float32)
    st = abi.Stats()
    x0, y0, x1, y1 = rect if rect else (0, 0, W, H)
    lib().orc_render_rect(scene_ptr, C.byref(cam), C.byref(params), x0, y0, x1, y1, abi.ptr(film, abi.f32p), C.byref(st))
    return film, st


def bsdf_eval(bsdf_type, params, op, wi, aux):
    """the oracle's BSDF::f (op 0) / pdf (1) / sample(bRec, pdf, sample) (2) for n query records, laid out like
    mtsgpu_bsdf_eval: returns [n][8]"""
    aux = np.atleast_2d(np.asarray(aux, dtype=np.float32))
    n = aux.shape[0]
    q = np.zeros((n, 6), dtype=np.float32)
    q[:, :3] = np.asarray(wi, dtype=np.float32).reshape(-1, 3)
    q[:, 3:3 + aux.shape[1]] = aux
    P = np.zeros(abi.BSDF_NPARAMS, dtype=np.float32); P[:len(params)] = params
    out = np.zeros((n, 8), dtype=np.float32)
    lib().orc_bsdf_eval(int(bsdf_type), abi.ptr(P, abi.f32p), int(op), n, abi.ptr(q, abi.f32p), abi.ptr(out, abi.f32p))
    return out


def li_samples(scene_ptr, cam, params, pix_samples):
    ps = np.ascontiguousarray(pix_samples, dtype=np.uint32).reshape(-1, 3)
    out = np.zeros((ps.shape[0], 8), dtype=np.float32)
    lib().orc_li_samples(scene_ptr, C.byref(cam), C.byref(params), abi.ptr(ps, abi.u32p), ps.shape[0], abi.ptr(out, abi.f32p))
    return out


def sampler_values(params, pixel_key, sample_index, n, two_d=False):
    """generate() for the pixel, then n x next1D() (or next2D()) of camera sample `sample_index`"""
    out = np.zeros((n, 2) if two_d else (n,), dtype=np.float32)
    lib().orc_sampler_values(C.byref(params), int(pixel_key), int(sample_index), int(n), int(bool(two_d)), abi.ptr(out, abi.f32p))
    return out


def trace_rays(scene_ptr, rays, shadow=False, counts=False):
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    hits = np.zeros((r.shape[0], 4), dtype=np.uint32)
    tc = TraceCounts()
    lib().orc_trace_rays(scene_ptr, abi.ptr(r, abi.f32p), r.shape[0], 1 if shadow else 0, abi.ptr(hits, abi.u32p),
                         C.byref(tc) if counts else None)
    return (hits, tc) if counts else hits


def develop(film):
    """Film::develop: spec / weight (mfilm.cpp:108-116), weight 0 -> 0"""
    w = film[..., 4:5]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(w > 0, np.float32(1.0) / w, np.float32(0)).astype(np.float32)
    return (film[..., :3] * inv).astype(np.float32)


RFILTERS = {"box": 0, "gaussian": 1, "mitchell": 2, "catmullrom": 3, "wsinc": 4}


def chord_rays(center, radius, n):
    """the rays of src/tests/test_kd.cpp:96-116 (default-seeded Random)"""
    c = np.asarray(center, dtype=np.float32)
    rays = np.zeros((n, 8), dtype=np.float32)
    lib().orc_chord_rays(abi.ptr(c, abi.f32p), float(radius), n, abi.ptr(rays, abi.f32p))
    return rays


def tabulate_filter(kind="gaussian", half_size=-1.0, p0=-1.0, p1=-1.0):
    """half_size / p0 / p1 <= 0 select the plugin defaults (gaussian: stddev; mitchell: B, C; wsinc: cycles)"""
    f = TabFilter()
    lib().orc_tabulate_filter(RFILTERS[kind], half_size, p0, p1, C.byref(f))
    return f


def render_tiles(scene_ptr, cam, params, filt, block_size=32, part=0, n_parts=1, hq_edges=False):
    film = np.zeros((cam.height, cam.width, 5), dtype=np.float32)
    st = abi.Stats()
    lib().orc_render_tiles(scene_ptr, C.byref(cam), C.byref(params), C.byref(filt), block_size, part, n_parts, int(bool(hq_edges)),
                           abi.ptr(film, abi.f32p), C.byref(st))
    return film, st
