"""mitsuba-renderer_amd: MI355X-native path-tracing hot path behind Mitsuba 0.2.1's
integrator interface.

Host-side mirror of the reference's objects for this path, over the C ABI of
include/mtsgpu.h (libmtsgpu.so, built by csrc/Makefile):

    Scene          <- Scene + ShapeKDTree after Scene::initialize()   (src/librender/scene.cpp:291-332)
    PerspectiveCamera  <- PerspectiveCameraImpl                      (src/cameras/perspective.cpp)
    MIPathTracer   <- the `path` integrator plugin                   (src/integrators/path/path.cpp)

There is no CPU fallback: every compute call raises if libmtsgpu.so or a gfx950
device is missing.  (The CPU oracle lives in oracle/ and is test infrastructure.)"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi, scenes, filmreduce  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MTSGPU_LIB selects an experiment build (tools/build_variant.sh); the product is libmtsgpu.so next to this file
LIB_PATH = os.environ.get("MTSGPU_LIB") or os.path.join(_HERE, "libmtsgpu.so")

EXPORTS = [
    "mtsgpu_create", "mtsgpu_destroy", "mtsgpu_last_error", "mtsgpu_abi_version", "mtsgpu_source_hash", "mtsgpu_abi_sizeof", "mtsgpu_set_stream",
    "mtsgpu_upload_scene", "mtsgpu_set_camera", "mtsgpu_set_integrator", "mtsgpu_set_direct_integrator", "mtsgpu_set_sampler",
    "mtsgpu_set_tiles", "mtsgpu_set_rfilter", "mtsgpu_set_film_edges", "mtsgpu_tabulate_filter", "mtsgpu_set_film_buffer", "mtsgpu_set_options", "mtsgpu_render", "mtsgpu_sync",
    "mtsgpu_read_film", "mtsgpu_clear_film", "mtsgpu_get_stats", "mtsgpu_trace_rays", "mtsgpu_ld_tables",
    "mtsgpu_li_samples", "mtsgpu_flatten", "mtsgpu_flat_scene_get", "mtsgpu_flat_scene_free",
    "mtsgpu_flat_scene_kdstats", "mtsgpu_make_camera", "mtsgpu_make_camera_ortho", "mtsgpu_load_serialized", "mtsgpu_loaded_mesh_free",
    "mtsgpu_make_camera_crop", "mtsgpu_hbm_triad", "mtsgpu_sampler_values", "mtsgpu_random_values", "mtsgpu_set_tuning", "mtsgpu_gather_roof",
    "mtsgpu_create_multi", "mtsgpu_group_destroy", "mtsgpu_group_size", "mtsgpu_group_ctx", "mtsgpu_group_last_error",
    "mtsgpu_group_upload_scene", "mtsgpu_group_set_camera", "mtsgpu_group_set_integrator", "mtsgpu_group_set_sampler",
    "mtsgpu_group_set_rfilter", "mtsgpu_group_render", "mtsgpu_group_last_reduce_kind", "mtsgpu_group_rccl_ranks", "mtsgpu_group_reduce_note", "mtsgpu_bsdf_eval", "mtsgpu_replay_roof", "mtsgpu_group_set_tuning", "mtsgpu_tail_filter_flag",
]


class MtsGpuError(RuntimeError):
    pass


def load_serialized(path, shape_index=0, bsdf=-1, lum=-1, name=None):
    """<shape type="serialized"> (TriMesh::TriMesh(Stream *, int), src/librender/trimesh.cpp:156-236) -> MeshDesc"""
    h = C.c_void_p(); m = abi.Mesh()
    rc = lib().mtsgpu_load_serialized(os.fsencode(path), int(shape_index), C.byref(h), C.byref(m))
    if rc != 0:
        raise MtsGpuError("mtsgpu_load_serialized: %s" % lib().mtsgpu_last_error(None).decode())
    try:
        pos = abi.np_from(m.positions, (m.n_verts, 3), np.float32)
        tri = abi.np_from(m.triangles, (m.n_tris, 3), np.uint32)
        nrm = abi.np_from(m.normals, (m.n_verts, 3), np.float32) if m.normals else None
        return scenes.MeshDesc(pos, tri, bsdf=bsdf, lum=lum, face_normals=bool(m.face_normals), normals=nrm,
                               name=name or "%s#%d" % (os.path.basename(path), shape_index))
    finally:
        lib().mtsgpu_loaded_mesh_free(h)


_SOURCES = ["api.cpp", "group.cpp", "sampler.hip", "film.hip", "trace.hip", "shade.hip", "measure.hip",   # SRCS, then HDRS of csrc/Makefile
            "kdbuild.cpp", "flatten.cpp", "serialized.cpp",
            "host.h", "ctx.h", "kernels.h", "kdevice.h", "sampler.h", "devmath.h", os.path.join("..", "..", "include", "mtsgpu.h")]


def source_hash():
    """the hash csrc/Makefile stamps into the library (csrc/stamp.cpp), recomputed from the sources on disk"""
    import hashlib
    h = hashlib.sha256()
    for f in _SOURCES:
        h.update(open(os.path.join(_HERE, "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def built_hash(path=None):
    """mtsgpu_source_hash() of a built library, None if it cannot be loaded or predates the stamp"""
    try:
        L = C.CDLL(path or os.path.join(_HERE, "libmtsgpu.so"))
        L.mtsgpu_source_hash.restype = C.c_char_p
        return L.mtsgpu_source_hash().decode()
    except (OSError, AttributeError):
        return None


def _probe_hash(lib_path):
    """mtsgpu_source_hash() of the file at lib_path, asked in a child process (a library already loaded here would answer
    for the OLD file).  Returns (hash, None) or (None, why the library could not be loaded or asked)."""
    probe = "import ctypes as C; L = C.CDLL(%r); L.mtsgpu_source_hash.restype = C.c_char_p; print(L.mtsgpu_source_hash().decode())" % lib_path
    r = subprocess.run([os.sys.executable, "-c", probe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        return None, r.stderr.decode(errors="replace").strip().splitlines()[-1:] or ["exit code %d" % r.returncode]
    return r.stdout.decode().strip(), None


def build(force=False):
    """Compile libmtsgpu.so for gfx950 (hipcc cross-compiles without a GPU).  make is incremental; a binary whose stamped
    source hash differs from the sources on disk (e.g. a copied-in .so newer than the files) is rebuilt from scratch.  A
    library that cannot be LOADED is reported as such (with the loader's message), not mistaken for a stale one."""
    csrc = os.path.join(_HERE, "csrc")
    lib_path = os.path.join(_HERE, "libmtsgpu.so")          # what csrc/Makefile builds; MTSGPU_LIB variants are built by tools/build_variant.sh
    subprocess.check_call(["make", "-C", csrc, "-j4"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    got, why = _probe_hash(lib_path)
    if why is not None:
        raise MtsGpuError("libmtsgpu.so was built but cannot be loaded: %s" % "; ".join(why))
    if got != source_hash():
        subprocess.check_call(["make", "-C", csrc, "-j4", "-B"], stdout=subprocess.DEVNULL)
        got, why = _probe_hash(lib_path)
        if why is not None:
            raise MtsGpuError("libmtsgpu.so cannot be loaded after a full rebuild: %s" % "; ".join(why))
        if got != source_hash():
            raise MtsGpuError("libmtsgpu.so reports source hash %r after a full rebuild, the sources hash to %r" % (got, source_hash()))
    return lib_path


_lib = None


def lib():
    """Load libmtsgpu.so; fails loudly when it is missing (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MtsGpuError("libmtsgpu.so not built: run __graft_entry__.build() or `make -C mitsuba-renderer_amd/csrc`")
    L = C.CDLL(LIB_PATH)
    vp, f32p, u32p = C.c_void_p, abi.f32p, abi.u32p
    L.mtsgpu_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mtsgpu_destroy.argtypes = [vp]; L.mtsgpu_destroy.restype = None
    L.mtsgpu_last_error.argtypes = [vp]; L.mtsgpu_last_error.restype = C.c_char_p
    L.mtsgpu_abi_version.argtypes = []
    L.mtsgpu_source_hash.argtypes = []; L.mtsgpu_source_hash.restype = C.c_char_p
    L.mtsgpu_abi_sizeof.argtypes = [C.c_int]; L.mtsgpu_abi_sizeof.restype = C.c_size_t
    L.mtsgpu_set_stream.argtypes = [vp, vp]
    L.mtsgpu_upload_scene.argtypes = [vp, C.POINTER(abi.Scene)]
    L.mtsgpu_set_camera.argtypes = [vp, C.POINTER(abi.Camera)]
    L.mtsgpu_set_integrator.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.mtsgpu_set_sampler.argtypes = [vp, C.c_int, C.c_uint32, C.c_int, C.c_uint64]
    L.mtsgpu_set_direct_integrator.argtypes = [vp, C.c_int, C.c_int]
    L.mtsgpu_set_tiles.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.mtsgpu_set_film_buffer.argtypes = [vp, vp]
    L.mtsgpu_set_rfilter.argtypes = [vp, C.c_float, C.c_float, f32p]
    L.mtsgpu_set_film_edges.argtypes = [vp, C.c_int]
    L.mtsgpu_tabulate_filter.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, f32p, f32p]
    L.mtsgpu_set_options.argtypes = [vp, C.c_uint64, C.c_int, C.c_int]
    L.mtsgpu_render.argtypes = [vp, C.POINTER(C.c_int)]
    L.mtsgpu_sync.argtypes = [vp]
    L.mtsgpu_read_film.argtypes = [vp, f32p]
    L.mtsgpu_clear_film.argtypes = [vp]
    L.mtsgpu_get_stats.argtypes = [vp, C.POINTER(abi.Stats)]
    L.mtsgpu_trace_rays.argtypes = [vp, f32p, C.c_uint32, C.c_int, u32p]
    L.mtsgpu_ld_tables.argtypes = [vp, C.c_uint32, f32p, f32p]
    L.mtsgpu_li_samples.argtypes = [vp, u32p, C.c_uint32, f32p]
    L.mtsgpu_flatten.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.KdParams), C.POINTER(vp)]
    L.mtsgpu_flat_scene_get.argtypes = [vp]; L.mtsgpu_flat_scene_get.restype = C.POINTER(abi.Scene)
    L.mtsgpu_flat_scene_free.argtypes = [vp]; L.mtsgpu_flat_scene_free.restype = None
    L.mtsgpu_flat_scene_kdstats.argtypes = [vp, C.POINTER(C.c_double)]
    L.mtsgpu_make_camera.argtypes = [f32p, f32p, f32p, C.c_float, C.c_int, C.c_int, C.POINTER(abi.Camera)]
    L.mtsgpu_make_camera_ortho.argtypes = [f32p, f32p, f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.POINTER(abi.Camera)]
    L.mtsgpu_load_serialized.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp), C.POINTER(abi.Mesh)]
    L.mtsgpu_loaded_mesh_free.argtypes = [vp]; L.mtsgpu_loaded_mesh_free.restype = None
    L.mtsgpu_make_camera_crop.argtypes = [f32p, f32p, f32p, C.c_float] + [C.c_int] * 6 + [C.POINTER(abi.Camera)]
    L.mtsgpu_gather_roof.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_double)]
    L.mtsgpu_set_tuning.argtypes = [vp, C.c_char_p, C.c_long]
    L.mtsgpu_sampler_values.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, f32p]
    L.mtsgpu_random_values.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.mtsgpu_bsdf_eval.argtypes = [vp, C.c_uint32, f32p, C.c_int, C.c_uint32, f32p, f32p]
    L.mtsgpu_hbm_triad.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.mtsgpu_replay_roof.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_double)]
    L.mtsgpu_tail_filter_flag.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float]
    L.mtsgpu_create_multi.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    L.mtsgpu_group_destroy.argtypes = [vp]; L.mtsgpu_group_destroy.restype = None
    L.mtsgpu_group_size.argtypes = [vp]
    L.mtsgpu_group_ctx.argtypes = [vp, C.c_int]; L.mtsgpu_group_ctx.restype = vp
    L.mtsgpu_group_last_error.argtypes = [vp]; L.mtsgpu_group_last_error.restype = C.c_char_p
    L.mtsgpu_group_upload_scene.argtypes = [vp, C.POINTER(abi.Scene)]
    L.mtsgpu_group_set_camera.argtypes = [vp, C.POINTER(abi.Camera)]
    L.mtsgpu_group_set_integrator.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.mtsgpu_group_set_sampler.argtypes = [vp, C.c_int, C.c_uint32, C.c_int, C.c_uint64]
    L.mtsgpu_group_set_rfilter.argtypes = [vp, C.c_float, C.c_float, f32p]
    L.mtsgpu_group_render.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.mtsgpu_group_last_reduce_kind.argtypes = [vp]
    L.mtsgpu_group_rccl_ranks.argtypes = [vp]
    L.mtsgpu_group_reduce_note.argtypes = [vp]; L.mtsgpu_group_reduce_note.restype = C.c_char_p
    L.mtsgpu_group_set_tuning.argtypes = [vp, C.c_char_p, C.c_long]
    _lib = L
    return L


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Scene:
    """A flattened, render-ready scene: what Scene::initialize() leaves behind
    (kd-tree, TriAccel table, vertex normals, luminaire CDFs), built by mtsgpu_flatten()."""

    def __init__(self, description, kd_params=None, gpu_binning=False, gpu_exact=False):
        """gpu_binning: run the min-max binning phase of the kd-tree build (nodes of more than
        exactPrimThreshold primitives, gkdtree.h:1735-1867) on the current HIP device -- same tree, bit for bit.
        gpu_exact: the exact O(n log n) sweep below that threshold (gkdtree.h:1898-2345) on the device as well."""
        self.description = description
        self._desc, self._keep = description.to_ctypes()
        self._h = C.c_void_p()
        kp = kd_params if kd_params is not None else abi.KdParams()
        kp.gpu_binning = (kp.gpu_binning & ~3) | (1 if gpu_binning else 0) | (2 if gpu_exact else 0)
        rc = lib().mtsgpu_flatten(C.byref(self._desc), C.byref(kp), C.byref(self._h))
        if rc != 0:
            raise MtsGpuError("mtsgpu_flatten: %s" % lib().mtsgpu_last_error(None).decode())
        self.ptr = lib().mtsgpu_flat_scene_get(self._h)

    @property
    def sc(self):
        return self.ptr.contents

    def arrays(self):
        return abi.scene_arrays(self.sc)

    def kdstats(self):
        out = (C.c_double * 6)()
        lib().mtsgpu_flat_scene_kdstats(self._h, out)
        return dict(zip(["inner", "leaf", "indices", "exp_traversals", "exp_leaves", "exp_prims"], list(out)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:
            _lib.mtsgpu_flat_scene_free(h)
            self._h = None


class PerspectiveCamera:
    """`perspective` camera plugin: lookAt toWorld transform, fov along the smaller side, pinhole."""

    def __init__(self, origin, target, up, fov, width, height, apertureRadius=0.0, focusDepth=None):
        self.c = abi.Camera()
        rc = lib().mtsgpu_make_camera(abi.ptr(_f(origin), abi.f32p), abi.ptr(_f(target), abi.f32p), abi.ptr(_f(up), abi.f32p),
                                      C.c_float(fov), int(width), int(height), C.byref(self.c))
        if rc != 0:
            raise MtsGpuError("mtsgpu_make_camera: %s" % lib().mtsgpu_last_error(None).decode())
        if apertureRadius > 0:                       # thin lens (camera.cpp:164-166, perspective.cpp:90-103)
            self.c.aperture_radius = apertureRadius
            self.c.focus_depth = self.c.far_clip if focusDepth is None else focusDepth

    @classmethod
    def cropped(cls, desc, film_width, film_height, crop):
        """the camera of `desc` behind a film with a crop window (film.cpp:33-41): crop = (offsetX, offsetY, width, height)"""
        c = desc.camera
        self = cls.__new__(cls)
        self.c = abi.Camera()
        rc = lib().mtsgpu_make_camera_crop(abi.ptr(_f(c["origin"]), abi.f32p), abi.ptr(_f(c["target"]), abi.f32p), abi.ptr(_f(c["up"]), abi.f32p),
                                           C.c_float(c["fov"]), int(film_width), int(film_height), *[int(v) for v in crop], C.byref(self.c))
        if rc != 0:
            raise MtsGpuError("mtsgpu_make_camera_crop: %s" % lib().mtsgpu_last_error(None).decode())
        return self

    @classmethod
    def for_description(cls, desc, width, height):
        c = desc.camera
        if "ortho_scale" in c:
            return OrthographicCamera(c["origin"], c["target"], c["up"], c["ortho_scale"], width, height)
        return cls(c["origin"], c["target"], c["up"], c["fov"], width, height,
                   apertureRadius=c.get("aperture", 0.0), focusDepth=c.get("focus"))

    @property
    def width(self):
        return self.c.width

    @property
    def height(self):
        return self.c.height


class OrthographicCamera(PerspectiveCamera):
    """`orthographic` camera plugin (src/cameras/orthographic.cpp): toWorld = lookAt * scale(sx, sy, 1)"""

    def __init__(self, origin, target, up, scale, width, height):
        self.c = abi.Camera()
        sx, sy = (scale, scale) if np.isscalar(scale) else scale
        rc = lib().mtsgpu_make_camera_ortho(abi.ptr(_f(origin), abi.f32p), abi.ptr(_f(target), abi.f32p), abi.ptr(_f(up), abi.f32p),
                                            C.c_float(sx), C.c_float(sy), int(width), int(height), C.byref(self.c))
        if rc != 0:
            raise MtsGpuError("mtsgpu_make_camera_ortho: %s" % lib().mtsgpu_last_error(None).decode())


class MIPathTracer:
    """The `path` integrator (MIPathTracer : MonteCarloIntegrator) on one MI355X.

    Properties as in src/librender/integrator.cpp:272-292: maxDepth (-1 = unbounded),
    rrDepth (10), strictNormals (false).  The call order mirrors the host's:
    configure() -> preprocess(scene, camera, sampler...) -> render() [-> cancel()]."""

    def __init__(self, maxDepth=-1, rrDepth=10, strictNormals=False, device=0):
        self.maxDepth, self.rrDepth, self.strictNormals = int(maxDepth), int(rrDepth), bool(strictNormals)
        self._ctx = C.c_void_p()
        self._cancel = C.c_int(0)
        self._film_keep = None
        rc = lib().mtsgpu_create(int(device), C.byref(self._ctx))
        if rc != 0:
            raise MtsGpuError("mtsgpu_create: %s" % lib().mtsgpu_last_error(None).decode())
        self.camera = None

    def _chk(self, rc, what):
        if rc != 0:
            raise MtsGpuError("%s: %s (code %d)" % (what, lib().mtsgpu_last_error(self._ctx).decode(), rc))

    def configure(self):
        self._configure_integrator()
        return self

    def _configure_integrator(self):
        self._chk(lib().mtsgpu_set_integrator(self._ctx, self.maxDepth, self.rrDepth, int(self.strictNormals)), "set_integrator")

    def preprocess(self, scene, camera, sampler="independent", sampleCount=4, depth=3, seed=0x5EED):
        """Scene::preprocess -> Integrator::preprocess: upload the flattened scene, camera and sampler."""
        self.configure()
        sp = scene.ptr if isinstance(scene, Scene) else scene
        self._chk(lib().mtsgpu_upload_scene(self._ctx, sp), "upload_scene")
        self.camera = camera
        self._chk(lib().mtsgpu_set_camera(self._ctx, C.byref(camera.c if hasattr(camera, "c") else camera)), "set_camera")
        kind = {"independent": abi.SAMPLER_INDEPENDENT_KEYED, "ldsampler": abi.SAMPLER_LD_KEYED, "halton": abi.SAMPLER_HALTON,
                "hammersley": abi.SAMPLER_HAMMERSLEY, "stratified": abi.SAMPLER_STRATIFIED_KEYED}[sampler] if isinstance(sampler, str) else int(sampler)
        self._chk(lib().mtsgpu_set_sampler(self._ctx, kind, int(sampleCount), int(depth), int(seed)), "set_sampler")
        return True

    def set_tiles(self, block_size=32, part=0, n_parts=1):
        self._chk(lib().mtsgpu_set_tiles(self._ctx, block_size, part, n_parts), "set_tiles")

    def set_rfilter(self, kind="box", halfSize=-1.0, stddev=-1.0, B=-1.0, C=-1.0, cycles=-1.0):
        """Film reconstruction filter plugin (src/rfilters/{box,gaussian,mitchell,catmullrom,wsinc}.cpp); negative
        values select the plugin defaults"""
        if kind == "box":
            self._chk(lib().mtsgpu_set_rfilter(self._ctx, 0.5, 0.5, None), "set_rfilter")
            return
        size = np.zeros(2, dtype=np.float32); values = np.zeros(256, dtype=np.float32)
        k = {"gaussian": 1, "mitchell": 2, "catmullrom": 3, "wsinc": 4}[kind]
        p0, p1 = {1: (stddev, -1.0), 2: (B, C), 3: (-1.0, -1.0), 4: (cycles, -1.0)}[k]
        rc = lib().mtsgpu_tabulate_filter(k, halfSize, p0, p1, abi.ptr(size, abi.f32p), abi.ptr(values, abi.f32p))
        if rc != 0:
            raise MtsGpuError("mtsgpu_tabulate_filter: %s" % lib().mtsgpu_last_error(None).decode())
        self._chk(lib().mtsgpu_set_rfilter(self._ctx, float(size[0]), float(size[1]), abi.ptr(values, abi.f32p)), "set_rfilter")
        return size, values.reshape(16, 16)

    def set_film_edges(self, highQualityEdges=False):
        """Film property `highQualityEdges` (src/librender/film.cpp:51, renderproc.cpp:146-153)"""
        self._chk(lib().mtsgpu_set_film_edges(self._ctx, int(bool(highQualityEdges))), "set_film_edges")

    def set_options(self, max_paths=0, count_traversal=False, time_kernels=False):
        self._chk(lib().mtsgpu_set_options(self._ctx, int(max_paths), int(count_traversal), int(time_kernels)), "set_options")

    def set_tuning(self, **knobs):
        """scheduling knobs of the traversal kernel (mtsgpu_set_tuning); results never depend on them"""
        for k, v in knobs.items():
            self._chk(lib().mtsgpu_set_tuning(self._ctx, k.encode(), int(v)), "set_tuning")

    def set_stream(self, hip_stream):
        self._chk(lib().mtsgpu_set_stream(self._ctx, C.c_void_p(hip_stream)), "set_stream")

    def set_film_buffer(self, device_ptr, keepalive=None):
        self._film_keep = keepalive
        self._chk(lib().mtsgpu_set_film_buffer(self._ctx, C.c_void_p(device_ptr)), "set_film_buffer")

    def clear_film(self):
        self._chk(lib().mtsgpu_clear_film(self._ctx), "clear_film")

    def render(self):
        """SampleIntegrator::render: returns True, or False when cancelled (integrator.cpp:87-120)."""
        self._cancel.value = 0
        rc = lib().mtsgpu_render(self._ctx, C.byref(self._cancel))
        if rc == -4:
            return False
        self._chk(rc, "render")
        return True

    def cancel(self):
        self._cancel.value = 1

    def sync(self):
        self._chk(lib().mtsgpu_sync(self._ctx), "sync")

    def film(self):
        """[H][W][5] float32: spectrum rgb sum, alpha sum, weight sum (ImageBlock layout)"""
        cam = self.camera.c if hasattr(self.camera, "c") else self.camera
        out = np.zeros((cam.height, cam.width, 5), dtype=np.float32)
        self._chk(lib().mtsgpu_read_film(self._ctx, abi.ptr(out, abi.f32p)), "read_film")
        return out

    def stats(self):
        st = abi.Stats()
        self._chk(lib().mtsgpu_get_stats(self._ctx, C.byref(st)), "get_stats")
        return st.as_dict()

    # --- kernels exposed for parity tests --------------------------------------
    def trace_rays(self, rays, shadow=False):
        r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        hits = np.zeros((r.shape[0], 4), dtype=np.uint32)
        self._chk(lib().mtsgpu_trace_rays(self._ctx, abi.ptr(r, abi.f32p), r.shape[0], int(shadow), abi.ptr(hits, abi.u32p)), "trace_rays")
        return hits

    REPLAY_KINDS = {"deep": 0, "camera": 1, "shadow": 2}

    def replay_roof(self, n, stride=1, reps=3, kind="deep"):
        """mtsgpu_replay_roof: the request stream of n rays of the frame rendered last replayed without arithmetic, next to
        the product kernel on the same rays.  kind: "deep" = the last ray of every stride-th path (closest-hit kernel),
        "camera" = the camera rays of the last pass, generated again (closest-hit kernel, first-bounce launch shape),
        "shadow" = every stride-th slot of the shadow queue (any-hit kernel)"""
        out = (C.c_double * 12)()
        self._chk(lib().mtsgpu_replay_roof(self._ctx, self.REPLAY_KINDS[kind], int(n), int(stride), int(reps), out), "replay_roof")
        keys = ["rays", "requests", "truncated_rays", "product_ms", "replay_ms", "pair_global", "pair_lds", "node_global", "node_lds",
                "heads", "tails", "spills"]
        return dict(zip(keys, list(out)))

    def ld_tables(self, pixel_key, spp, depth):
        t1 = np.zeros((depth, spp), dtype=np.float32)
        t2 = np.zeros((depth, spp, 2), dtype=np.float32)
        self._chk(lib().mtsgpu_ld_tables(self._ctx, int(pixel_key), abi.ptr(t1, abi.f32p), abi.ptr(t2, abi.f32p)), "ld_tables")
        return t1, t2

    def sampler_values(self, pixel_key, sample_index, n, two_d=False):
        """generate() for the pixel, then n x next1D() (or next2D()) of camera sample `sample_index`, on the device"""
        out = np.zeros((n, 2) if two_d else (n,), dtype=np.float32)
        self._chk(lib().mtsgpu_sampler_values(self._ctx, int(pixel_key), int(sample_index), int(n), int(bool(two_d)),
                                              abi.ptr(out, abi.f32p)), "sampler_values")
        return out

    def random_values(self, op, n, seed=0, arg=0, clone=0):
        """the reference's Random (MT19937-64) on the device: op 0 nextULong, 1 nextFloat bits, 2 nextSize(arg), 3 shuffle(0..n-1)"""
        out = np.zeros(n, dtype=np.uint64)
        self._chk(lib().mtsgpu_random_values(self._ctx, int(op), int(seed), int(arg), int(clone), int(n),
                                             out.ctypes.data_as(C.POINTER(C.c_uint64))), "random_values")
        return out

    def bsdf_eval(self, bsdf_type, params, op, wi, aux):
        """BSDF::f (op 0), pdf (1), sample(bRec, pdf, sample) (2) on the device for n query records (mtsgpu_bsdf_eval):
        wi [n][3] or [3]; aux = wo [n][3] (op 0, 1) or the 2D sample [n][2] (op 2).  Returns [n][8]."""
        aux = np.atleast_2d(np.asarray(aux, dtype=np.float32))
        n = aux.shape[0]
        q = np.zeros((n, 6), dtype=np.float32)
        q[:, :3] = np.asarray(wi, dtype=np.float32).reshape(-1, 3)
        q[:, 3:3 + aux.shape[1]] = aux
        P = np.zeros(abi.BSDF_NPARAMS, dtype=np.float32); P[:len(params)] = params
        out = np.zeros((n, 8), dtype=np.float32)
        self._chk(lib().mtsgpu_bsdf_eval(self._ctx, int(bsdf_type), abi.ptr(P, abi.f32p), int(op), n, abi.ptr(q, abi.f32p),
                                         abi.ptr(out, abi.f32p)), "bsdf_eval")
        return out

    def li_samples(self, pix_samples):
        ps = np.ascontiguousarray(pix_samples, dtype=np.uint32).reshape(-1, 3)
        out = np.zeros((ps.shape[0], 8), dtype=np.float32)
        self._chk(lib().mtsgpu_li_samples(self._ctx, abi.ptr(ps, abi.u32p), ps.shape[0], abi.ptr(out, abi.f32p)), "li_samples")
        return out

    def close(self):
        if self._ctx and _lib is not None:
            _lib.mtsgpu_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceGroup:
    """Several GPUs behind one host process (mtsgpu_create_multi): what the Mitsuba plugin uses, since Scene::render
    calls Integrator::render once (src/librender/scene.cpp:356-359).  devices may repeat (two contexts on one GPU)."""

    def __init__(self, devices, maxDepth=-1, rrDepth=10, strictNormals=False):
        self.maxDepth, self.rrDepth, self.strictNormals = int(maxDepth), int(rrDepth), bool(strictNormals)
        self._g = C.c_void_p()
        self._cancel = C.c_int(0)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = lib().mtsgpu_create_multi(len(devices), devs, C.byref(self._g))
        if rc != 0:
            raise MtsGpuError("mtsgpu_create_multi: %s" % lib().mtsgpu_last_error(None).decode())
        self.camera = None

    def _chk(self, rc, what):
        if rc != 0:
            raise MtsGpuError("%s: %s (code %d)" % (what, lib().mtsgpu_group_last_error(self._g).decode(), rc))

    def __len__(self):
        return lib().mtsgpu_group_size(self._g)

    def member(self, i):
        return lib().mtsgpu_group_ctx(self._g, int(i))

    def preprocess(self, scene, camera, sampler="independent", sampleCount=4, depth=3, seed=0x5EED):
        sp = scene.ptr if isinstance(scene, Scene) else scene
        self._chk(lib().mtsgpu_group_set_integrator(self._g, self.maxDepth, self.rrDepth, int(self.strictNormals)), "set_integrator")
        self._chk(lib().mtsgpu_group_upload_scene(self._g, sp), "upload_scene")
        self.camera = camera
        self._chk(lib().mtsgpu_group_set_camera(self._g, C.byref(camera.c)), "set_camera")
        kind = {"independent": abi.SAMPLER_INDEPENDENT_KEYED, "ldsampler": abi.SAMPLER_LD_KEYED, "halton": abi.SAMPLER_HALTON,
                "hammersley": abi.SAMPLER_HAMMERSLEY, "stratified": abi.SAMPLER_STRATIFIED_KEYED}[sampler]
        self._chk(lib().mtsgpu_group_set_sampler(self._g, kind, int(sampleCount), int(depth), int(seed)), "set_sampler")
        return True

    def set_rfilter(self, kind="box", halfSize=-1.0, stddev=-1.0):
        if kind == "box":
            self._chk(lib().mtsgpu_group_set_rfilter(self._g, 0.5, 0.5, None), "set_rfilter")
            return
        size = np.zeros(2, dtype=np.float32); values = np.zeros(256, dtype=np.float32)
        rc = lib().mtsgpu_tabulate_filter({"gaussian": 1, "mitchell": 2, "catmullrom": 3, "wsinc": 4}[kind], halfSize, stddev, -1.0,
                                          abi.ptr(size, abi.f32p), abi.ptr(values, abi.f32p))
        if rc != 0:
            raise MtsGpuError("mtsgpu_tabulate_filter: %s" % lib().mtsgpu_last_error(None).decode())
        self._chk(lib().mtsgpu_group_set_rfilter(self._g, float(size[0]), float(size[1]), abi.ptr(values, abi.f32p)), "set_rfilter")

    def render(self, block_size=32, ordered_reduce=False):
        """ordered_reduce: False / 0 = RCCL when possible, True / 1 = the ordered peer-copy sum, 2 = RCCL must initialise"""
        self._cancel.value = 0
        rc = lib().mtsgpu_group_render(self._g, int(block_size), int(ordered_reduce), C.byref(self._cancel))
        if rc == -4:
            return False
        self._chk(rc, "group_render")
        return True

    def cancel(self):
        self._cancel.value = 1

    def reduce_kind(self):
        return {0: "ordered peer-copy sum", 1: "rccl ncclReduce"}.get(lib().mtsgpu_group_last_reduce_kind(self._g))

    def rccl_ranks(self):
        """ranks of the group's RCCL communicator once it has passed its self-check, else 0"""
        return int(lib().mtsgpu_group_rccl_ranks(self._g))

    def set_tuning(self, **knobs):
        """mtsgpu_set_tuning on every member (and the group's own test knob rccl_fail)"""
        for k, v in knobs.items():
            self._chk(lib().mtsgpu_group_set_tuning(self._g, k.encode(), int(v)), "group_set_tuning")

    def reduce_note(self):
        """why the last render fell back to the ordered sum ("" when it did not)"""
        return lib().mtsgpu_group_reduce_note(self._g).decode()

    def film(self):
        out = np.zeros((self.camera.c.height, self.camera.c.width, 5), dtype=np.float32)
        rc = lib().mtsgpu_read_film(self.member(0), abi.ptr(out, abi.f32p))
        if rc != 0:
            raise MtsGpuError("read_film: %s" % lib().mtsgpu_last_error(self.member(0)).decode())
        return out

    def member_stats(self, i):
        st = abi.Stats()
        lib().mtsgpu_get_stats(self.member(i), C.byref(st))
        return st.as_dict()

    def close(self):
        if self._g and _lib is not None:
            _lib.mtsgpu_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hbm_triad_gbs(device=0, gib=1.0, iters=5):
    """measured HBM triad bandwidth in GB/s (the practical roof next to the 8 TB/s specification), None on failure"""
    out = C.c_double(0)
    rc = lib().mtsgpu_hbm_triad(int(device), int(gib * (1 << 30)), int(iters), C.byref(out))
    return float(out.value) if rc == 0 and out.value > 0 else None


def gather_roof(device=0, footprint_mib=4):
    """measured lane-level 16-byte gather requests per second (the roof of the traversal kernel), None on failure"""
    out = C.c_double(0)
    rc = lib().mtsgpu_gather_roof(int(device), int(footprint_mib) << 20, C.byref(out))
    return float(out.value) if rc == 0 and out.value > 0 else None


class MIDirectIntegrator(MIPathTracer):
    """The `direct` integrator (src/integrators/direct/direct.cpp); more than one sample per strategy needs a sampler
    with sample arrays (independent, ldsampler, stratified), as in the reference"""

    def __init__(self, luminaireSamples=1, bsdfSamples=1, device=0):
        MIPathTracer.__init__(self, device=device)
        self.luminaireSamples, self.bsdfSamples = int(luminaireSamples), int(bsdfSamples)

    def _configure_integrator(self):
        self._chk(lib().mtsgpu_set_direct_integrator(self._ctx, self.luminaireSamples, self.bsdfSamples), "set_direct_integrator")


def develop(film):
    """Film::develop: pixel = spectrum * (1 / weight) (src/films/mfilm.cpp:108-116)"""
    w = film[..., 4:5]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(w > 0, np.float32(1.0) / w, np.float32(0)).astype(np.float32)
    return (film[..., :3] * inv).astype(np.float32)
