"""ctypes view of include/mtsgpu.h (the C ABI of libmtsgpu.so).

Field order mirrors the header exactly; tests/test_abi.py checks the sizes
against the sizes the C compiler reports (mtsgpu_abi_sizeof)."""
import ctypes as C
import numpy as np

ABI_VERSION = 6
BSDF_LAMBERTIAN, BSDF_DIELECTRIC, BSDF_ROUGHMETAL, BSDF_MICROFACET, BSDF_MIRROR, BSDF_PHONG, BSDF_ROUGHGLASS, BSDF_DIFFTRANS = 0, 1, 2, 3, 4, 5, 6, 7
BSDF_TWOSIDED = 0x100
BSDF_NPARAMS = 16
LUM_AREA, LUM_CONSTANT, LUM_POINT, LUM_DIRECTIONAL, LUM_SPOT, LUM_ENVMAP, LUM_COLLIMATED = 0, 1, 2, 3, 4, 5, 6
LUM_NPARAMS = 32
SAMPLER_INDEPENDENT_KEYED, SAMPLER_LD_KEYED, SAMPLER_HALTON, SAMPLER_HAMMERSLEY, SAMPLER_STRATIFIED_KEYED = 0, 1, 2, 3, 4
SHAPE_HAS_NORMALS = 1
SHAPE_TRIMESH, SHAPE_SPHERE = 0, 1
SHAPE_NPARAMS = 24
KNOTRIANGLE = 0xFFFFFFFF

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)


class Scene(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_shapes", C.c_uint32), ("n_tris", C.c_uint32), ("n_verts", C.c_uint32),
        ("vtx_pos", f32p), ("vtx_nrm", f32p), ("tri_idx", u32p),
        ("shape_tri_offset", u32p), ("shape_bsdf", i32p), ("shape_lum", i32p), ("shape_flags", u32p),
        ("shape_type", u32p), ("shape_params", f32p),
        ("n_nodes", C.c_uint32), ("n_indices", C.c_uint32),
        ("kd_nodes", u32p), ("kd_indices", u32p), ("triaccel", u32p),
        ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3),
        ("n_bsdfs", C.c_uint32), ("bsdf_type", u32p), ("bsdf_params", f32p),
        ("n_lums", C.c_uint32), ("lum_type", u32p), ("lum_params", f32p), ("lum_shape", i32p),
        ("lum_inv_area", f32p), ("lum_cdf_offset", u32p), ("lum_tri_cdf", f32p),
        ("lum_sel_cdf", f32p), ("lum_sel_pdf", f32p),
        ("lum_sel_sum", C.c_float), ("background_lum", C.c_int32),
        ("env_width", C.c_uint32), ("env_height", C.c_uint32), ("env_pixels", f32p),
        ("env_pdf_width", C.c_uint32), ("env_pdf_height", C.c_uint32), ("env_pdf", f32p), ("env_cdf", f32p),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("raster_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 16),
        ("near_clip", C.c_float), ("far_clip", C.c_float),
        ("width", C.c_int32), ("height", C.c_int32),
        ("aperture_radius", C.c_float), ("focus_depth", C.c_float), ("kind", C.c_int32),
        ("crop_offset_x", C.c_int32), ("crop_offset_y", C.c_int32), ("film_width", C.c_int32), ("film_height", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("camera_samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64),
        ("n_inner", C.c_uint64), ("n_leaf", C.c_uint64), ("n_idx", C.c_uint64), ("n_tri_tested", C.c_uint64),
        ("trace_launches", C.c_uint64),
        ("trace_ms", C.c_double), ("shade_ms", C.c_double), ("total_ms", C.c_double),
        ("path_length_sum", C.c_uint64), ("bin_overflow_retries", C.c_uint64),
        ("trace_union_ms", C.c_double),
        ("req_pair_global", C.c_uint64), ("req_pair_lds", C.c_uint64), ("req_node_global", C.c_uint64), ("req_node_lds", C.c_uint64),
        ("req_tail", C.c_uint64), ("req_spill", C.c_uint64), ("req_head", C.c_uint64),
        ("trace_first_ms", C.c_double), ("trace_shadow_ms", C.c_double),
        ("rays_redone", C.c_uint64),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        # the reference's `avgPathLength` statistic (path.cpp:24,212-213)
        d["avg_path_length"] = self.path_length_sum / self.camera_samples if self.camera_samples else None
        return d


class Mesh(C.Structure):
    _fields_ = [
        ("n_verts", C.c_uint32), ("n_tris", C.c_uint32),
        ("positions", f32p), ("normals", f32p), ("triangles", u32p),
        ("face_normals", C.c_int32), ("bsdf", C.c_int32), ("lum", C.c_int32),
        ("shape_type", C.c_int32), ("sphere_center", C.c_float * 3), ("sphere_radius", C.c_float),
        ("sphere_inverted", C.c_int32),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("n_meshes", C.c_uint32), ("meshes", C.POINTER(Mesh)),
        ("n_bsdfs", C.c_uint32), ("bsdf_type", u32p), ("bsdf_params", f32p),
        ("n_lums", C.c_uint32), ("lum_type", u32p), ("lum_params", f32p),
        ("camera_pos", C.c_float * 3), ("has_camera", C.c_int32),
        ("env_width", C.c_uint32), ("env_height", C.c_uint32), ("env_bitmap", f32p),
    ]


class KdParams(C.Structure):
    _fields_ = [
        ("traversal_cost", C.c_float), ("query_cost", C.c_float), ("empty_space_bonus", C.c_float),
        ("stop_prims", C.c_int32), ("max_bad_refines", C.c_int32), ("exact_prim_threshold", C.c_int32),
        ("max_depth", C.c_int32), ("min_max_bins", C.c_int32),
        ("clip", C.c_int32), ("retract", C.c_int32), ("n_threads", C.c_int32), ("gpu_binning", C.c_int32),
    ]


def ptr(a, typ):
    """numpy array -> typed pointer (the array must outlive the pointer)"""
    if a is None:
        return C.cast(None, typ)
    return a.ctypes.data_as(typ)


def np_from(p, shape, dtype):
    """typed pointer -> numpy copy"""
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    arr = np.ctypeslib.as_array(p, shape=(n,))
    return np.array(arr, dtype=dtype, copy=True).reshape(shape)


def scene_arrays(sc):
    """Copy every array of a mtsgpu_scene into a dict of numpy arrays (for bit-exact comparisons)."""
    ns, nt, nv = sc.n_shapes, sc.n_tris, sc.n_verts
    nl = sc.n_lums
    out = {
        "vtx_pos": np_from(sc.vtx_pos, (nv, 3), np.float32),
        "vtx_nrm": np_from(sc.vtx_nrm, (nv, 3), np.float32),
        "tri_idx": np_from(sc.tri_idx, (nt, 3), np.uint32),
        "shape_tri_offset": np_from(sc.shape_tri_offset, (ns + 1,), np.uint32),
        "shape_bsdf": np_from(sc.shape_bsdf, (ns,), np.int32),
        "shape_lum": np_from(sc.shape_lum, (ns,), np.int32),
        "shape_flags": np_from(sc.shape_flags, (ns,), np.uint32),
        "shape_type": np_from(sc.shape_type, (ns,), np.uint32),
        "shape_params": np_from(sc.shape_params, (ns, SHAPE_NPARAMS), np.float32),
        "kd_nodes": np_from(sc.kd_nodes, (sc.n_nodes, 2), np.uint32),
        "kd_indices": np_from(sc.kd_indices, (sc.n_indices,), np.uint32),
        "triaccel": np_from(sc.triaccel, (nt, 12), np.uint32),
        "aabb_min": np.array(list(sc.aabb_min), dtype=np.float32),
        "aabb_max": np.array(list(sc.aabb_max), dtype=np.float32),
        "bsdf_type": np_from(sc.bsdf_type, (sc.n_bsdfs,), np.uint32),
        "bsdf_params": np_from(sc.bsdf_params, (sc.n_bsdfs, BSDF_NPARAMS), np.float32),
        "lum_type": np_from(sc.lum_type, (nl,), np.uint32),
        "lum_params": np_from(sc.lum_params, (nl, LUM_NPARAMS), np.float32),
        "lum_shape": np_from(sc.lum_shape, (nl,), np.int32),
        "lum_inv_area": np_from(sc.lum_inv_area, (nl,), np.float32),
        "lum_cdf_offset": np_from(sc.lum_cdf_offset, (nl + 1,), np.uint32),
        "lum_sel_cdf": np_from(sc.lum_sel_cdf, (nl + 1,), np.float32),
        "lum_sel_pdf": np_from(sc.lum_sel_pdf, (nl,), np.float32),
        "lum_sel_sum": np.float32(sc.lum_sel_sum),
        "background_lum": int(sc.background_lum),
    }
    ncdf = int(out["lum_cdf_offset"][-1]) if nl else 0
    out["lum_tri_cdf"] = np_from(sc.lum_tri_cdf, (ncdf,), np.float32)
    out["env_size"] = (int(sc.env_width), int(sc.env_height), int(sc.env_pdf_width), int(sc.env_pdf_height))
    out["env_pixels"] = np_from(sc.env_pixels, (sc.env_height, sc.env_width, 3), np.float32)
    npdf = sc.env_pdf_width * sc.env_pdf_height
    out["env_pdf"] = np_from(sc.env_pdf, (npdf,), np.float32)
    out["env_cdf"] = np_from(sc.env_cdf, (npdf + 1 if npdf else 0,), np.float32)
    return out
