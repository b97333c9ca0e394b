"""An independent writer of Mitsuba 0.2.1's object streams (what InstanceManager::serialize produces), for the byte-level tests
of integration/streamparse.h.  Test infrastructure: every function follows the serialize() of the class it is named after and
cites it (paths relative to /root/reference); nothing here is used by the product.

Wire format: src/libcore/serialization.cpp:72-85 (object references), src/libcore/stream.cpp (host byte order; strings are
NUL-terminated, :214-216), include/mitsuba/core/stream.h:192 (bool = one byte)."""
import struct

import numpy as np


class Stream:
    def __init__(self, float_bytes=4):
        self.b = bytearray()
        self.fb = float_bytes
        self.ids = {}            # InstanceManager::m_objToId
        self.counter = 0

    # --- Stream primitives ---
    def uint(self, v): self.b += struct.pack("<I", v)
    def int(self, v): self.b += struct.pack("<i", v)
    def bool(self, v): self.b += struct.pack("<B", 1 if v else 0)
    def float(self, v): self.b += struct.pack("<f" if self.fb == 4 else "<d", float(v))
    def string(self, s): self.b += s.encode() + b"\0"
    def spectrum(self, rgb):                     # Spectrum::serialize (spectrum.h:438-440), SPECTRUM_SAMPLES = 3
        for v in rgb: self.float(v)
    def vec3(self, v):                           # TPoint3 / TVector3::serialize (point.h, vector.h)
        for x in v: self.float(x)
    def matrix(self, m):                         # Matrix::serialize (matrix.h:387-389): row major
        for x in np.asarray(m, dtype=np.float64).reshape(16): self.float(x)
    def transform(self, fwd, inv):               # Transform::serialize (transform.h:304-307)
        self.matrix(fwd); self.matrix(inv)

    # --- InstanceManager::serialize (serialization.cpp:72-85) ---
    def ref(self, key, cls=None, body=None):
        """key None -> NULL; a key seen before -> its id; otherwise a new id, the class name and body(self)"""
        if key is None:
            self.uint(0); return
        if key in self.ids:
            self.uint(self.ids[key]); return
        self.counter += 1
        self.uint(self.counter)
        self.string(cls)
        self.ids[key] = self.counter
        body(self)

    def bytes(self): return bytes(self.b)


def configurable(s, parent_key=None):
    """ConfigurableObject::serialize (src/libcore/properties.cpp:358-363): the parent"""
    s.ref(parent_key)


def const_spectrum_texture(s, key, rgb, parent_key=None):
    """ConstantSpectrumTexture::serialize (src/librender/texture.cpp:89-93) behind Texture::serialize (:39-41)"""
    def body(s):
        configurable(s, parent_key)
        s.spectrum(rgb)
    s.ref(key, "ConstantSpectrumTexture", body)


def const_float_texture(s, key, value, parent_key=None):
    """ConstantFloatTexture::serialize (texture.cpp:100-103)"""
    def body(s):
        configurable(s, parent_key)
        s.float(value)
    s.ref(key, "ConstantFloatTexture", body)


# ---------------------------------------------------------------------------------------------------------------
# BSDFs: P is the parameter block of include/mtsgpu.h (MTSGPU_BSDF_*), the source of the values that are written
# ---------------------------------------------------------------------------------------------------------------
def bsdf(s, key, btype, P, twosided=False, name="", tex_parent=False, share_textures=False):
    """BSDF::serialize (src/librender/bsdf.cpp:50-53) + the plugin's own fields.  tex_parent: the textures are children added
    through addChild (their parent is the BSDF: a known id) instead of constants made in the constructor (no parent);
    share_textures: both texture slots of a two-texture BSDF hold the same object (the second reference is just its id)"""
    if twosided:                                 # TwoSidedBRDF::serialize (src/bsdfs/twosided.cpp:52-56)
        def body(s):
            configurable(s); s.string(name)
            bsdf(s, (key, "nested"), btype, P, False, name, tex_parent, share_textures)
        s.ref(key, "TwoSidedBRDF", body)
        return
    tp = key if tex_parent else None
    def tex(slot, rgb):
        k = (key, "tex") if share_textures else (key, slot)
        const_spectrum_texture(s, k, rgb, tp)
    cls = {0: "Lambertian", 1: "Dielectric", 2: "RoughMetal", 3: "Microfacet", 4: "Mirror", 5: "Phong", 6: "RoughGlass",
           7: "DiffuseTransmitter"}[btype]
    def body(s):
        configurable(s); s.string(name)
        if btype == 0:                           # lambertian.cpp:137-141
            tex("reflectance", P[0:3])
        elif btype == 1:                         # dielectric.cpp:88-95
            s.float(P[0]); s.float(P[1])
            tex("specularReflectance", P[2:5]); tex("specularTransmittance", P[5:8])
        elif btype == 2:                         # roughmetal.cpp:169-176
            tex("specularReflectance", P[7:10])
            s.float(P[0]); s.spectrum(P[1:4]); s.spectrum(P[4:7])
        elif btype == 3:                         # microfacet.cpp:283-293
            tex("diffuseReflectance", P[5:8]); tex("specularReflectance", P[8:11])
            for k in range(5): s.float(P[k])
        elif btype == 4:                         # mirror.cpp:51-55
            s.spectrum(P[0:3])
        elif btype == 5:                         # phong.cpp:246-256
            tex("diffuseReflectance", P[5:8]); tex("specularReflectance", P[8:11])
            for k in range(5): s.float(P[k])
        elif btype == 6:                         # roughglass.cpp:735-744
            s.int(int(P[0]))
            const_float_texture(s, (key, "alpha"), P[1], tp)
            tex("specularReflectance", P[4:7]); tex("specularTransmittance", P[7:10])
            s.float(P[2]); s.float(P[3])
        elif btype == 7:                         # difftrans.cpp:142-146
            tex("transmittance", P[0:3])
    s.ref(key, cls, body)


# ---------------------------------------------------------------------------------------------------------------
# luminaires (serialized with the parent taken off: integration/gpucommon.h serializedDetached)
# ---------------------------------------------------------------------------------------------------------------
def luminaire_base(s, w2l, l2w, parent_key=None, weight=1.0, ltype=0, intersectable=False, name=""):
    """Luminaire::serialize (src/librender/luminaire.cpp:65-73)"""
    configurable(s, parent_key)
    s.ref(None)                                  # m_medium
    s.float(weight); s.int(ltype); s.bool(intersectable)
    s.transform(w2l, l2w)                        # m_worldToLuminaire (and its inverse)
    s.string(name)


def directional(s, key, w2l, l2w, direction, intensity, disk_origin, disk_radius):
    """DirectionalLuminaire::serialize (src/luminaires/directional.cpp:57-63)"""
    def body(s):
        luminaire_base(s, w2l, l2w, ltype=1)
        s.vec3(direction); s.spectrum(intensity); s.vec3(disk_origin); s.float(disk_radius)
    s.ref(key, "DirectionalLuminaire", body)


def spot(s, key, w2l, l2w, intensity, beam_width, cutoff_angle, texture=(1.0, 1.0, 1.0), texture_class="ConstantSpectrumTexture"):
    """SpotLuminaire::serialize (src/luminaires/spot.cpp:63-70): the texture made in the constructor has no parent (:42-43)"""
    def body(s):
        luminaire_base(s, w2l, l2w, ltype=2)
        def tbody(s):
            configurable(s); s.spectrum(texture)
        s.ref((key, "texture"), texture_class, tbody)
        s.spectrum(intensity); s.float(beam_width); s.float(cutoff_angle)
    s.ref(key, "SpotLuminaire", body)


def collimated(s, key, w2l, l2w, intensity, radius):
    """CollimatedBeamLuminaire::serialize (src/luminaires/collimated.cpp:47-51)"""
    def body(s):
        luminaire_base(s, w2l, l2w, ltype=1)
        s.spectrum(intensity); s.float(radius)
    s.ref(key, "CollimatedBeamLuminaire", body)


def envmap(s, key, w2l, l2w, intensity_scale, path, bsphere, exr_bytes):
    """EnvMapLuminaire::serialize (src/luminaires/envmap.cpp:79-93)"""
    def body(s):
        luminaire_base(s, w2l, l2w, ltype=4)
        s.float(intensity_scale); s.string(path)
        s.vec3(bsphere[0:3]); s.float(bsphere[3])            # BSphere::serialize (bsphere.h:121-124)
        s.uint(len(exr_bytes)); s.b += exr_bytes
    s.ref(key, "EnvMapLuminaire", body)


def area_luminaire(s, key, shape_key, intensity, w2l, l2w):
    """AreaLuminaire::serialize (src/luminaires/area.cpp:51-56): its parent and m_shape are the shape"""
    def body(s):
        luminaire_base(s, w2l, l2w, parent_key=shape_key, ltype=8, intersectable=True)
        s.spectrum(intensity)
        s.ref(shape_key)
    s.ref(key, "AreaLuminaire", body)


def sphere(s, key, o2w, w2o, radius, center, inverted, bsdf_args=None, lum_intensity=None, occluder=True):
    """Sphere::serialize (src/shapes/sphere.cpp:72-78) behind Shape::serialize (src/librender/shape.cpp:130-138)"""
    eye = np.eye(4)
    def body(s):
        configurable(s)                          # the parent (the Scene) was taken off
        if bsdf_args is None: s.ref(None)
        else: bsdf(s, (key, "bsdf"), *bsdf_args)
        s.ref(None)                              # m_subsurface
        if lum_intensity is None: s.ref(None)
        else: area_luminaire(s, (key, "lum"), key, lum_intensity, eye, eye)
        s.ref(None); s.ref(None)                 # interior / exterior medium
        s.bool(occluder)
        s.transform(o2w, w2o); s.float(radius); s.vec3(center); s.bool(inverted)
    s.ref(key, "Sphere", body)
