"""Procedural scene descriptions for the configurations of SURVEY.md section 8(d).

A description is what Mitsuba's XML loader would hand to Scene::addChild: plain
triangle meshes (TriMesh ctor, include/mitsuba/render/trimesh.h:52-56), BSDF and
luminaire property blocks and a lookAt camera.  Nothing here is on the hot path;
flattening (normals, CDFs, TriAccel, kd-tree) happens in mtsgpu_flatten().
All arithmetic is float32 so the arrays are identical on every host."""
import ctypes as C
import numpy as np
from . import abi

F = np.float32


class MeshDesc:
    def __init__(self, positions, triangles, bsdf=-1, lum=-1, face_normals=True, normals=None, name=""):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        self.triangles = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.bsdf, self.lum, self.face_normals, self.name = int(bsdf), int(lum), bool(face_normals), name
        self.shape_type, self.sphere = abi.SHAPE_TRIMESH, None


class SphereDesc(MeshDesc):
    """<shape type="sphere"> with `center` and `radius` (src/shapes/sphere.cpp:44-47): one kd-tree primitive"""

    def __init__(self, center, radius, bsdf=-1, lum=-1, inverted=False, name="sphere"):
        MeshDesc.__init__(self, np.zeros((0, 3)), np.zeros((0, 3), dtype=np.uint32), bsdf, lum, True, None, name)
        self.shape_type = abi.SHAPE_SPHERE
        self.sphere = (tuple(float(F(v)) for v in center), float(F(radius)), bool(inverted))


class SceneDescription:
    """meshes + BSDF blocks + luminaires + camera (+ integrator defaults for the config)"""

    def __init__(self, name):
        self.name = name
        self.meshes = []
        self.bsdf_type = []
        self.bsdf_params = []
        self.lum_type = []
        self.lum_params = []
        self.camera = dict(origin=(0.0, 1.0, 3.4), target=(0.0, 1.0, 0.0), up=(0.0, 1.0, 0.0), fov=39.3)
        self.max_depth = -1
        self.rr_depth = 10
        self.env_bitmap = None      # [H][W][3] float32 of the envmap luminaire

    # --- property blocks --------------------------------------------------
    def add_bsdf(self, btype, params):
        p = np.zeros(abi.BSDF_NPARAMS, dtype=np.float32)
        p[:len(params)] = np.asarray(params, dtype=np.float32)
        self.bsdf_type.append(int(btype))
        self.bsdf_params.append(p)
        return len(self.bsdf_type) - 1

    def lambertian(self, r, g=None, b=None):
        g = r if g is None else g
        b = r if b is None else b
        return self.add_bsdf(abi.BSDF_LAMBERTIAN, [r, g, b])

    def dielectric(self, int_ior=1.5046, ext_ior=1.0):
        return self.add_bsdf(abi.BSDF_DIELECTRIC, [int_ior, ext_ior, 1, 1, 1, 1, 1, 1])

    def roughmetal(self, alpha=0.1, ior=0.37, k=2.82):
        return self.add_bsdf(abi.BSDF_ROUGHMETAL, [alpha, ior, ior, ior, k, k, k, 1, 1, 1])

    def microfacet(self, alpha=0.1, kd=0.5, ks=0.5, int_ior=1.5, ext_ior=1.0, rd=1.0, rs=1.0):
        return self.add_bsdf(abi.BSDF_MICROFACET, [alpha, kd, ks, int_ior, ext_ior, rd, rd, rd, rs, rs, rs])

    def mirror(self, r=0.8):
        return self.add_bsdf(abi.BSDF_MIRROR, [r, r, r])

    def phong(self, exponent=10.0, rd=0.5, rs=0.2, kd=1.0, ks=1.0):
        """parameter block as Phong::configure() leaves it (src/bsdfs/phong.cpp:74-96), float32 arithmetic"""
        kd, ks, rd, rs = F(kd), F(ks), F(rd), F(rs)
        if kd * rd + ks * rs > F(1.0):                       # verifyEnergyConservation
            norm = F(1) / (kd * rd + ks * rs)
            kd, ks = kd * norm, ks * norm
        avg_d = (rd + rd + rd) * F(1.0 / 3) * kd             # Spectrum::average() * m_kd
        avg_s = (rs + rs + rs) * F(1.0 / 3) * ks
        ssw = avg_s / (avg_d + avg_s)
        dsw = F(1.0) - ssw
        return self.add_bsdf(abi.BSDF_PHONG, [exponent, kd, ks, ssw, dsw, rd, rd, rd, rs, rs, rs])

    def roughglass(self, alpha=0.1, int_ior=1.5046, ext_ior=1.0, distribution="beckmann", refl=1.0, trans=1.0):
        """src/bsdfs/roughglass.cpp: for `phong` the constructor maps alpha to the exponent 2/alpha^2 - 2 (:130-136)"""
        d = {"beckmann": 0, "phong": 1, "ggx": 2}[distribution]
        a = F(alpha)
        if d == 1:
            a = F(2) / (a * a) - F(2)
        return self.add_bsdf(abi.BSDF_ROUGHGLASS, [d, a, int_ior, ext_ior, refl, refl, refl, trans, trans, trans])

    def difftrans(self, t=0.5):
        return self.add_bsdf(abi.BSDF_DIFFTRANS, [t, t, t])

    def twosided(self, bsdf):
        """wrap an existing BSDF block in the `twosided` adapter (src/bsdfs/twosided.cpp)"""
        self.bsdf_type[bsdf] |= abi.BSDF_TWOSIDED
        return bsdf

    def point_light(self, position, intensity):
        l = self.add_lum(abi.LUM_POINT, [intensity] * 3 if np.isscalar(intensity) else intensity)
        self.lum_params[l][3:6] = np.asarray(position, dtype=np.float32)
        return l

    def directional_light(self, direction, intensity):
        l = self.add_lum(abi.LUM_DIRECTIONAL, [intensity] * 3 if np.isscalar(intensity) else intensity)
        d = np.asarray(direction, dtype=np.float32)
        self.lum_params[l][3:6] = d / np.sqrt((d * d).sum(), dtype=np.float32)
        return l

    def spot_light(self, position, target, intensity, cutoff_deg=20.0, beam_deg=None):
        """SpotLuminaire with toWorld = lookAt(position, target, up) (src/luminaires/spot.cpp:33-62)"""
        l = self.add_lum(abi.LUM_SPOT, [intensity] * 3 if np.isscalar(intensity) else intensity)
        P = self.lum_params[l]
        pos, tgt = np.asarray(position, dtype=np.float32), np.asarray(target, dtype=np.float32)
        d = tgt - pos; d = d / np.sqrt((d * d).sum(), dtype=np.float32)
        up = np.array([0, 0, 1], dtype=np.float32) if abs(d[2]) < 0.9 else np.array([1, 0, 0], dtype=np.float32)
        right = np.cross(d, up).astype(np.float32); right /= np.sqrt((right * right).sum(), dtype=np.float32)
        new_up = np.cross(right, d).astype(np.float32)
        P[3:6] = pos
        beam = cutoff_deg * 0.75 if beam_deg is None else beam_deg
        P[8] = F(cutoff_deg) * (F(np.pi) / F(180.0))         # degToRad
        P[19] = F(beam) * (F(np.pi) / F(180.0))
        P[10:13] = right; P[13:16] = new_up; P[16:19] = d    # world -> luminaire rotation (rows)
        return l

    def collimated_beam(self, position, target, intensity, radius=0.01):
        """<luminaire type="collimated"> with toWorld = lookAt(position, target) (src/luminaires/collimated.cpp)"""
        l = self.add_lum(abi.LUM_COLLIMATED, [intensity] * 3 if np.isscalar(intensity) else intensity)
        P = self.lum_params[l]
        pos, tgt = np.asarray(position, dtype=np.float32), np.asarray(target, dtype=np.float32)
        d = tgt - pos; d = d / np.sqrt((d * d).sum(), dtype=np.float32)
        up = np.array([0, 0, 1], dtype=np.float32) if abs(d[2]) < 0.9 else np.array([1, 0, 0], dtype=np.float32)
        right = np.cross(d, up).astype(np.float32); right /= np.sqrt((right * right).sum(), dtype=np.float32)
        new_up = np.cross(right, d).astype(np.float32)
        R = np.stack([right, new_up, d], axis=1).astype(np.float32)          # luminaire -> world (columns)
        P[3] = radius
        L2W = np.concatenate([R, pos.reshape(3, 1)], axis=1).astype(np.float32)
        Rt = R.T.astype(np.float32)
        W2L = np.concatenate([Rt, (-(Rt @ pos)).reshape(3, 1)], axis=1).astype(np.float32)
        P[4:16] = W2L.ravel(); P[16:28] = L2W.ravel()
        return l

    def envmap(self, bitmap, intensity_scale=1.0, to_world=None):
        """<luminaire type="envmap"> (src/luminaires/envmap.cpp): lat-long bitmap [H][W][3], optional rotation"""
        l = self.add_lum(abi.LUM_ENVMAP, [intensity_scale, 0, 0])
        R = np.eye(3, dtype=np.float32) if to_world is None else np.asarray(to_world, dtype=np.float32).reshape(3, 3)
        self.lum_params[l][16:25] = R.ravel()
        self.env_bitmap = np.ascontiguousarray(bitmap, dtype=np.float32)
        assert self.env_bitmap.ndim == 3 and self.env_bitmap.shape[2] == 3
        return l

    def add_lum(self, ltype, intensity):
        p = np.zeros(abi.LUM_NPARAMS, dtype=np.float32)
        p[:3] = np.asarray(intensity, dtype=np.float32)
        self.lum_type.append(int(ltype))
        self.lum_params.append(p)
        return len(self.lum_type) - 1

    def add_mesh(self, *a, **k):
        self.meshes.append(MeshDesc(*a, **k))
        return self.meshes[-1]

    def add_sphere(self, *a, **k):
        self.meshes.append(SphereDesc(*a, **k))
        return self.meshes[-1]

    @property
    def n_tris(self):
        return sum(m.triangles.shape[0] for m in self.meshes)

    # --- ctypes view --------------------------------------------------------
    def to_ctypes(self):
        """-> (mtsgpu_scene_desc, keepalive list)"""
        keep = []
        meshes = (abi.Mesh * len(self.meshes))()
        for i, m in enumerate(self.meshes):
            keep += [m.positions, m.triangles, m.normals]
            meshes[i].n_verts = m.positions.shape[0]
            meshes[i].n_tris = m.triangles.shape[0]
            meshes[i].positions = abi.ptr(m.positions, abi.f32p)
            meshes[i].normals = abi.ptr(m.normals, abi.f32p)
            meshes[i].triangles = abi.ptr(m.triangles, abi.u32p)
            meshes[i].face_normals = 1 if m.face_normals else 0
            meshes[i].bsdf = m.bsdf
            meshes[i].lum = m.lum
            meshes[i].shape_type = m.shape_type
            if m.sphere is not None:
                meshes[i].sphere_center = (C.c_float * 3)(*m.sphere[0])
                meshes[i].sphere_radius = m.sphere[1]
                meshes[i].sphere_inverted = 1 if m.sphere[2] else 0
        bt = np.asarray(self.bsdf_type, dtype=np.uint32)
        bp = np.ascontiguousarray(np.stack(self.bsdf_params) if self.bsdf_params else np.zeros((0, abi.BSDF_NPARAMS)), dtype=np.float32)
        lt = np.asarray(self.lum_type, dtype=np.uint32)
        lp = np.ascontiguousarray(np.stack(self.lum_params) if self.lum_params else np.zeros((0, abi.LUM_NPARAMS)), dtype=np.float32)
        d = abi.SceneDesc()
        d.n_meshes = len(self.meshes)
        d.meshes = C.cast(meshes, C.POINTER(abi.Mesh))
        d.n_bsdfs = len(bt)
        d.bsdf_type = abi.ptr(bt, abi.u32p)
        d.bsdf_params = abi.ptr(bp, abi.f32p)
        d.n_lums = len(lt)
        d.lum_type = abi.ptr(lt, abi.u32p)
        d.lum_params = abi.ptr(lp, abi.f32p)
        d.camera_pos = (C.c_float * 3)(*[float(v) for v in self.camera["origin"]])
        d.has_camera = 1
        if self.env_bitmap is not None:
            d.env_width, d.env_height = self.env_bitmap.shape[1], self.env_bitmap.shape[0]
            d.env_bitmap = abi.ptr(self.env_bitmap, abi.f32p)
            keep.append(self.env_bitmap)
        keep += [meshes, bt, bp, lt, lp]
        return d, keep


# ---------------------------------------------------------------------------
# geometry helpers
# ---------------------------------------------------------------------------
def _quad(p0, e1, e2, want_normal):
    """two triangles covering p0 + s*e1 + t*e2, wound so that cross(p1-p0, p2-p0) points along want_normal"""
    p0, e1, e2 = (np.asarray(v, dtype=np.float32) for v in (p0, e1, e2))
    if np.dot(np.cross(e1, e2), np.asarray(want_normal, dtype=np.float32)) < 0:
        e1, e2 = e2, e1
    pos = np.stack([p0, p0 + e1, p0 + e1 + e2, p0 + e2]).astype(np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32)
    return pos, tri


def hash32(a, b, c, seed=1):
    """fixed integer mixer (uint32 arithmetic), vectorised"""
    with np.errstate(over="ignore"):
        h = (np.asarray(a, dtype=np.uint32) * np.uint32(0x9E3779B1)
             ^ np.asarray(b, dtype=np.uint32) * np.uint32(0x85EBCA77)
             ^ np.asarray(c, dtype=np.uint32) * np.uint32(0xC2B2AE3D)
             ^ np.uint32(seed) * np.uint32(0x27D4EB2F))
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x2C1B3C6D)
        h ^= h >> np.uint32(12)
        h *= np.uint32(0x297A2D39)
        h ^= h >> np.uint32(15)
    return h


def _grid_face(face_id, p0, e1, e2, normal, n, amp):
    """n x n grid of cells x 2 triangles with unshared vertices; interior grid points are
    displaced along `normal` by amp*(hash/2^32 - 0.5); border points stay put (watertight box)."""
    p0, e1, e2, normal = (np.asarray(v, dtype=np.float32) for v in (p0, e1, e2, normal))
    if np.dot(np.cross(e1, e2), normal) < 0:
        e1, e2 = e2, e1
    ii, jj = np.meshgrid(np.arange(n + 1, dtype=np.uint32), np.arange(n + 1, dtype=np.uint32), indexing="ij")
    s = ii.astype(np.float32) / F(n)
    t = jj.astype(np.float32) / F(n)
    disp = (hash32(np.uint32(face_id), ii, jj).astype(np.float32) / F(4294967296.0) - F(0.5)) * F(amp)
    border = (ii == 0) | (ii == n) | (jj == 0) | (jj == n)
    disp = np.where(border, F(0), disp).astype(np.float32)
    grid = (p0[None, None, :] + s[..., None] * e1[None, None, :] + t[..., None] * e2[None, None, :]
            + disp[..., None] * normal[None, None, :]).astype(np.float32)
    a = grid[:-1, :-1]; b = grid[1:, :-1]; c = grid[1:, 1:]; d = grid[:-1, 1:]
    # two triangles per cell: (a, b, c), (a, c, d); 6 unshared vertices per cell
    pos = np.stack([a, b, c, a, c, d], axis=2).reshape(-1, 3).astype(np.float32)
    tri = np.arange(pos.shape[0], dtype=np.uint32).reshape(-1, 3)
    return pos, tri


def icosphere(subdiv, radius, centre):
    """icosahedron subdivided `subdiv` times, shared vertices, outward winding"""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
                  [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
                  [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(subdiv):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        es = np.sort(edges, axis=1)
        uniq, inv = np.unique(es, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = v.shape[0]
        v = np.concatenate([v, mid], axis=0)
        nf = f.shape[0]
        inv = inv.reshape(-1)
        m01, m12, m20 = base + inv[:nf], base + inv[nf:2 * nf], base + inv[2 * nf:]
        f = np.concatenate([
            np.stack([f[:, 0], m01, m20], axis=1), np.stack([f[:, 1], m12, m01], axis=1),
            np.stack([f[:, 2], m20, m12], axis=1), np.stack([m01, m12, m20], axis=1)], axis=0)
    pos = (v * radius + np.asarray(centre, dtype=np.float64)[None, :]).astype(np.float32)
    # make sure the winding is outward (cross(p1-p0, p2-p0) . (centroid - centre) > 0)
    p = v[f]
    n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    flip = np.einsum("ij,ij->i", n, p.mean(axis=1)) < 0
    f[flip] = f[flip][:, [0, 2, 1]]
    return pos, f.astype(np.uint32)


# ---------------------------------------------------------------------------
# configurations
# ---------------------------------------------------------------------------
def _box_faces():
    # (name, p0, e1, e2, inward normal)
    return [
        ("floor",   (-1, 0, -1), (2, 0, 0), (0, 0, 2), (0, 1, 0)),
        ("ceiling", (-1, 2, -1), (2, 0, 0), (0, 0, 2), (0, -1, 0)),
        ("back",    (-1, 0, -1), (2, 0, 0), (0, 2, 0), (0, 0, 1)),
        ("left",    (-1, 0, -1), (0, 0, 2), (0, 2, 0), (1, 0, 0)),
        ("right",   (1, 0, -1),  (0, 0, 2), (0, 2, 0), (-1, 0, 0)),
    ]


def _add_light(sd, intensity=15.0):
    black = sd.lambertian(0.0)
    lum = sd.add_lum(abi.LUM_AREA, [intensity] * 3)
    pos, tri = _quad((-0.25, 1.99, -0.25), (0.5, 0, 0), (0, 0, 0.5), (0, -1, 0))
    sd.add_mesh(pos, tri, bsdf=black, lum=lum, face_normals=True, name="light")


def cornell_c1():
    """C1/C2: 6 quads / 12 triangles, lambertian walls, one area luminaire"""
    sd = SceneDescription("cornell_c1")
    white = sd.lambertian(0.73)
    red = sd.lambertian(0.63, 0.065, 0.05)
    green = sd.lambertian(0.14, 0.45, 0.091)
    for name, p0, e1, e2, nrm in _box_faces():
        pos, tri = _quad(p0, e1, e2, nrm)
        sd.add_mesh(pos, tri, bsdf={"left": red, "right": green}.get(name, white), face_normals=True, name=name)
    _add_light(sd)
    sd.max_depth = 4
    return sd


def cornell_c3(grid=320, sphere_subdiv=5):
    """C3/C4: one displaced walls TriMesh (5 faces x grid^2 x 2 tris; 1 024 000 at grid=320),
    a dielectric icosphere and the light quad"""
    sd = SceneDescription("cornell_c3_g%d" % grid)
    white = sd.lambertian(0.73)
    glass = sd.dielectric(1.5046, 1.0)
    ps, ts, base = [], [], 0
    for fid, (name, p0, e1, e2, nrm) in enumerate(_box_faces()):
        pos, tri = _grid_face(fid, p0, e1, e2, nrm, grid, 0.01)
        ps.append(pos); ts.append(tri + np.uint32(base)); base += pos.shape[0]
    sd.add_mesh(np.concatenate(ps), np.concatenate(ts), bsdf=white, face_normals=True, name="walls")
    pos, tri = icosphere(sphere_subdiv, 0.4, (0.3, 0.4, 0.2))
    sd.add_mesh(pos, tri, bsdf=glass, face_normals=False, name="glass")
    _add_light(sd)
    sd.max_depth = 16
    return sd


def cornell_c5(sphere_subdiv=4):
    """C5: C1 box + four icospheres (lambertian, roughmetal, dielectric, microfacet) + constant env"""
    sd = cornell_c1()
    sd.name = "cornell_c5"
    mats = [sd.lambertian(0.5), sd.roughmetal(), sd.dielectric(), sd.microfacet()]
    centres = [(-0.5, 0.3, -0.4), (0.5, 0.3, -0.4), (-0.5, 0.3, 0.45), (0.5, 0.3, 0.45)]
    for m, c in zip(mats, centres):
        pos, tri = icosphere(sphere_subdiv, 0.3, c)
        sd.add_mesh(pos, tri, bsdf=m, face_normals=False, name="sphere")
    sd.add_lum(abi.LUM_CONSTANT, [1.0, 1.0, 1.0])
    sd.max_depth = 32
    return sd


def next_rows(sphere_subdiv=2):
    """mirror / phong / twosided BSDFs, point / spot / directional luminaires, thin lens"""
    sd = SceneDescription("next_rows")
    white = sd.twosided(sd.lambertian(0.73))
    red = sd.twosided(sd.lambertian(0.63, 0.065, 0.05))
    green = sd.lambertian(0.14, 0.45, 0.091)
    for name, p0, e1, e2, nrm in _box_faces():
        pos, tri = _quad(p0, e1, e2, nrm)
        if name == "back":
            tri = tri[:, [0, 2, 1]]                 # wound the wrong way round: only a twosided BSDF shades it
        sd.add_mesh(pos, tri, bsdf={"left": red, "right": green}.get(name, white), face_normals=True, name=name)
    _add_light(sd, intensity=6.0)
    mats = [sd.mirror(0.8), sd.phong(20.0, 0.4, 0.3), sd.twosided(sd.roughmetal(0.2)), sd.phong(5.0, 0.9, 0.6)]
    centres = [(-0.5, 0.3, -0.4), (0.5, 0.3, -0.4), (-0.5, 0.3, 0.45), (0.5, 0.3, 0.45)]
    for m, c in zip(mats, centres):
        pos, tri = icosphere(sphere_subdiv, 0.3, c)
        sd.add_mesh(pos, tri, bsdf=m, face_normals=False, name="sphere")
    # a diffusely transmitting sheet in front of the back wall and a collimated beam onto the floor
    pos, tri = _quad((-0.6, 0.9, -0.7), (1.2, 0, 0), (0, 0.7, 0), (0, 0, 1))
    sd.add_mesh(pos, tri, bsdf=sd.difftrans(0.6), face_normals=True, name="sheet")
    sd.collimated_beam((0.1, 1.9, 0.1), (0.1, 0.0, 0.1), 6.0, radius=0.15)
    sd.point_light((0.0, 1.2, 0.6), 0.6)
    sd.spot_light((-0.8, 1.8, 0.8), (0.4, 0.0, -0.3), 25.0, cutoff_deg=25.0)
    sd.directional_light((0.3, -1.0, -0.4), 0.4)
    sd.camera["aperture"] = 0.04
    sd.camera["focus"] = 3.3
    sd.max_depth = 8
    return sd


def spheres():
    """analytic `sphere` shapes (glass, mirror, diffuse, one inverted) and a sphere-shaped
    area luminaire (Sphere::sampleSolidAngle) next to a mesh luminaire, inside the Cornell box"""
    sd = SceneDescription("spheres")
    white = sd.lambertian(0.73)
    red = sd.lambertian(0.63, 0.065, 0.05)
    green = sd.lambertian(0.14, 0.45, 0.091)
    for name, p0, e1, e2, nrm in _box_faces():
        pos, tri = _quad(p0, e1, e2, nrm)
        sd.add_mesh(pos, tri, bsdf={"left": red, "right": green}.get(name, white), face_normals=True, name=name)
    _add_light(sd, intensity=4.0)
    sd.add_sphere((-0.45, 0.35, -0.3), 0.35, bsdf=sd.dielectric())
    sd.add_sphere((0.5, 0.3, 0.3), 0.3, bsdf=sd.mirror(0.9))
    sd.add_sphere((0.0, 0.2, 0.55), 0.2, bsdf=sd.lambertian(0.3, 0.4, 0.8))
    sd.add_sphere((0.45, 1.3, -0.4), 0.25, bsdf=sd.twosided(sd.phong(30.0, 0.3, 0.5)), inverted=True)
    sd.add_sphere((-0.75, 0.15, 0.6), 0.15, bsdf=sd.roughglass(0.15, 1.5, 1.0, "beckmann"))
    sd.add_sphere((-0.25, 0.12, 0.8), 0.12, bsdf=sd.roughglass(0.4, 1.5, 1.0, "ggx"))
    sd.add_sphere((0.75, 0.12, 0.8), 0.12, bsdf=sd.roughglass(0.3, 1.33, 1.0, "phong", trans=0.9))
    lum = sd.add_lum(abi.LUM_AREA, [9.0, 7.0, 4.0])
    sd.add_sphere((-0.55, 1.45, 0.35), 0.12, bsdf=sd.lambertian(0.0), lum=lum)
    sd.max_depth = 8
    return sd


def env_bitmap(width=96, height=40, seed=5):
    """a synthetic HDR lat-long image: sky gradient, a small hot sun, ground colour, hash noise (not a power of two,
    so MIPMap's Lanczos up-sampling runs)"""
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    v = (y + F(0.5)) / F(height)
    img = np.zeros((height, width, 3), dtype=np.float32)
    sky = (F(1) - v)[..., None] * np.array([0.5, 0.7, 1.2], dtype=np.float32) + F(0.15)
    ground = np.array([0.25, 0.2, 0.12], dtype=np.float32)
    img[:] = np.where((v < F(0.55))[..., None], sky, ground)
    sx, sy = int(width * 0.3), int(height * 0.22)
    img[sy:sy + 2, sx:sx + 3] = np.array([60.0, 55.0, 40.0], dtype=np.float32)
    h = (x.astype(np.uint32) * np.uint32(73856093) ^ y.astype(np.uint32) * np.uint32(19349663) ^ np.uint32(seed * 83492791)) & np.uint32(0xFFFF)
    img *= (F(0.9) + F(0.2) * h.astype(np.float32) / F(65535.0))[..., None]
    return img


def envlit():
    """an `envmap` luminaire (rotated) lights glossy / diffuse / glass objects on a ground plane"""
    sd = SceneDescription("envlit")
    grey = sd.lambertian(0.5)
    pos, tri = _quad((-3.0, 0.0, -3.0), (6.0, 0, 0), (0, 0, 6.0), (0, 1, 0))
    sd.add_mesh(pos, tri, bsdf=grey, face_normals=True, name="ground")
    pos, tri = icosphere(2, 0.4, (-0.7, 0.4, 0.0))
    sd.add_mesh(pos, tri, bsdf=sd.roughmetal(0.15), face_normals=False, name="metal")
    sd.add_sphere((0.3, 0.45, -0.2), 0.45, bsdf=sd.dielectric())
    sd.add_sphere((1.1, 0.3, 0.5), 0.3, bsdf=sd.phong(40.0, 0.5, 0.3))
    c, s_ = F(np.cos(0.7)), F(np.sin(0.7))
    sd.envmap(env_bitmap(), 0.4, to_world=[[c, 0, s_], [0, 1, 0], [-s_, 0, c]])
    sd.camera = dict(origin=(0.2, 1.3, 3.6), target=(0.1, 0.4, 0.0), up=(0.0, 1.0, 0.0), fov=40.0)
    sd.max_depth = 6
    return sd


def matpreview(serialized_path, loader, material="roughglass"):
    """The reference's material-preview scene (data/blender/mitsuba/matpreview/matpreview.xml): the three shapes of
    matpreview.serialized (interior, exterior, ground plane) with the toWorld transforms, camera and envmap rotation
    of the XML; the EXR of the original is replaced by the synthetic lat-long bitmap (no OpenEXR here) and the
    checkerboard of the plane by its mean reflectance.  `loader(path, index)` -> MeshDesc (the product's
    mtsgpu_load_serialized in tests)."""
    sd = SceneDescription("matpreview")
    diff = sd.lambertian(0.18)
    plane = sd.lambertian(0.3)
    mat = {"roughglass": lambda: sd.roughglass(0.1, 1.5046, 1.0, "beckmann"), "roughmetal": lambda: sd.roughmetal(0.1),
           "phong": lambda: sd.phong(30.0, 0.2, 0.7), "dielectric": lambda: sd.dielectric()}[material]()
    inner = loader(serialized_path, 0)
    inner.positions = (inner.positions + np.array([0, 0, 0.0252155], dtype=np.float32)).astype(np.float32)
    inner.bsdf = diff
    outer = loader(serialized_path, 1)
    outer.bsdf = mat
    ground = loader(serialized_path, 2)
    ground.positions = (ground.positions * F(20.0) + np.array([-10, 10, 0], dtype=np.float32)).astype(np.float32)
    ground.bsdf = plane
    sd.meshes += [inner, outer, ground]
    sd.envmap(env_bitmap(), 1.0, to_world=[[-0.224951, -0.000001, -0.974370], [-0.974370, 0.0, 0.224951], [0.0, 1.0, -0.000001]])
    o = np.array([3.69558, -3.46243, 3.25463], dtype=np.float32)
    fwd = np.array([-0.654862, 0.610666, -0.445245], dtype=np.float32)
    sd.camera = dict(origin=tuple(o), target=tuple(o + fwd), up=(-0.31737, 0.312469, 0.895343), fov=28.8415)
    sd.max_depth = 8
    return sd


def bunny(serialized_path, loader, material="roughglass"):
    """The Stanford bunny of the reference's kd-tree test (data/tests/bunny.ply) read back from a `.serialized`
    container through `loader(path, index)`, on a ground plane under the synthetic environment map"""
    sd = SceneDescription("bunny")
    mat = {"roughglass": lambda: sd.roughglass(0.1, 1.5046, 1.0, "beckmann"), "roughmetal": lambda: sd.roughmetal(0.1),
           "lambertian": lambda: sd.lambertian(0.6)}[material]()
    mesh = loader(serialized_path, 0)
    mesh.bsdf = mat
    mesh.face_normals = False
    ground = loader(serialized_path, 1)
    ground.bsdf = sd.lambertian(0.4)
    sd.meshes += [mesh, ground]
    sd.envmap(env_bitmap(), 0.6)
    sd.camera = dict(origin=(-0.05, 0.2, 0.35), target=(-0.02, 0.1, 0.0), up=(0.0, 1.0, 0.0), fov=35.0)
    sd.max_depth = 8
    return sd


def fuzz(seed, n_meshes=14):
    """Random triangle soups, spheres, materials and luminaires inside the Cornell box: shared-vertex meshes with
    smooth normals, degenerate and duplicated triangles, axis-aligned slabs, every BSDF type with random parameters,
    area / point / spot / constant luminaires.  Meant for parity runs (tests/test_gpu_parity.py::test_fuzz_scenes)."""
    rng = np.random.RandomState(seed)
    sd = SceneDescription("fuzz_%d" % seed)
    u = lambda lo, hi, n=None: rng.uniform(lo, hi, n)
    mats = [sd.lambertian(*u(0.1, 0.9, 3)), sd.dielectric(u(1.2, 1.8), 1.0), sd.roughmetal(u(0.05, 0.4)),
            sd.microfacet(u(0.05, 0.4), u(0.2, 0.8), u(0.2, 0.8)), sd.mirror(u(0.5, 0.95)), sd.phong(u(5, 60), u(0.2, 0.6), u(0.1, 0.4)),
            sd.roughglass(u(0.05, 0.3), distribution=["beckmann", "phong", "ggx"][rng.randint(3)]), sd.difftrans(u(0.2, 0.8))]
    mats += [sd.twosided(mats[0]), sd.twosided(mats[3]), sd.twosided(mats[5])]
    white = sd.lambertian(0.73)
    for name, p0, e1, e2, nrm in _box_faces():
        pos, tri = _quad(p0, e1, e2, nrm)
        sd.add_mesh(pos, tri, bsdf=white if rng.rand() < 0.6 else mats[rng.randint(len(mats))], face_normals=True, name=name)
    for m in range(n_meshes):
        kind = rng.randint(5)
        c = np.array([u(-0.8, 0.8), u(0.2, 1.7), u(-0.8, 0.8)])
        bs = mats[rng.randint(len(mats))]
        if kind == 0:                                    # random soup, unshared vertices, some degenerate triangles
            nt = rng.randint(4, 60)
            pos = (c + u(-0.25, 0.25, (nt * 3, 3))).astype(np.float32)
            tri = np.arange(nt * 3, dtype=np.uint32).reshape(nt, 3)
            pos[3 * (nt // 2) + 1] = pos[3 * (nt // 2)]  # two equal vertices
            pos[3 * (nt // 3) + 2] = pos[3 * (nt // 3)] * 0.5 + pos[3 * (nt // 3) + 1] * 0.5      # collinear
            sd.add_mesh(pos, tri, bsdf=bs, face_normals=bool(rng.randint(2)), name="soup%d" % m)
        elif kind == 1:                                  # icosphere with smooth normals
            pos, tri = icosphere(rng.randint(0, 3), u(0.08, 0.3), tuple(c))
            sd.add_mesh(pos, tri, bsdf=bs, face_normals=False, name="ico%d" % m)
        elif kind == 2:                                  # axis-aligned slab: coplanar, duplicated triangles
            pos, tri = _quad(tuple(c), (u(0.1, 0.5), 0, 0), (0, 0, u(0.1, 0.5)), (0, 1, 0))
            tri = np.concatenate([tri, tri])             # every triangle twice
            sd.add_mesh(pos, tri, bsdf=bs, face_normals=True, name="slab%d" % m)
        elif kind == 3:                                  # analytic sphere
            sd.add_sphere(tuple(c), u(0.05, 0.3), bsdf=bs)
        else:                                            # wavy grid with shared vertices (smooth normals)
            n = rng.randint(3, 12)
            gx, gz = np.meshgrid(np.linspace(-0.3, 0.3, n), np.linspace(-0.3, 0.3, n), indexing="ij")
            gy = 0.05 * np.sin(7 * gx + seed) * np.cos(5 * gz)
            pos = (c + np.stack([gx, gy, gz], axis=-1).reshape(-1, 3)).astype(np.float32)
            idx = np.arange(n * n).reshape(n, n)
            a, b, cc, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
            tri = np.concatenate([np.stack([a, b, cc], 1), np.stack([a, cc, d], 1)]).astype(np.uint32)
            sd.add_mesh(pos, tri, bsdf=bs, face_normals=False, name="grid%d" % m)
    _add_light(sd, intensity=float(u(5, 20)))
    extra = rng.randint(4)
    if extra == 0: sd.point_light((u(-0.5, 0.5), u(0.5, 1.5), u(-0.5, 0.5)), (2.0, 2.0, 1.5))
    elif extra == 1: sd.spot_light((0.0, 1.8, 0.5), (u(-0.5, 0.5), 0.0, u(-0.5, 0.5)), (8.0, 8.0, 8.0), cutoff_deg=float(u(15, 40)))
    elif extra == 2: sd.add_lum(abi.LUM_CONSTANT, [0.3, 0.35, 0.5])
    else:                                                # a second area luminaire: an emitting sphere
        lum = sd.add_lum(abi.LUM_AREA, [6.0, 5.0, 4.0])
        sd.add_sphere((u(-0.6, 0.6), u(1.2, 1.7), u(-0.6, 0.6)), 0.08, bsdf=sd.lambertian(0.0), lum=lum)
    sd.max_depth = int(rng.randint(3, 12))
    sd.rr_depth = int(rng.randint(2, 6))
    return sd


def by_name(name, **kw):
    return {"c1": cornell_c1, "c3": cornell_c3, "c5": cornell_c5, "next": next_rows, "spheres": spheres, "envlit": envlit}[name](**kw)
