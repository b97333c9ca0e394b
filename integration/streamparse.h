/*
 * streamparse.h -- the field-order logic of the reference-side binding (integration/gpucommon.h), free of Mitsuba's headers:
 * it takes the bytes `InstanceManager::serialize` produced for a BSDF, a delta / environment luminaire or a sphere and fills
 * the parameter blocks of include/mtsgpu.h.  gpucommon.h produces the bytes (MemoryStream) and calls in here; the CPU tests
 * (tests/test_stream_parsers.py) produce them with an independent writer that follows each class's serialize() and compare
 * the blocks with what the library's own flattener builds for the same scene description, bit for bit.
 *
 * Wire format (src/libcore/serialization.cpp:72-85, src/libcore/stream.cpp): host byte order; an object reference is a uint32
 * id -- 0 for NULL, a known id for an object written before, otherwise a new id followed by the NUL-terminated class name
 * (stream.cpp:214-216, :391-402) and the object's serialize().  Float is `FloatT` (float with SINGLE_PRECISION).  A Spectrum is
 * SPECTRUM_SAMPLES = 3 Floats (spectrum.h:144-146, :438-440), a Point / Vector 3 Floats (point.h:264-268), a Matrix4x4 16 Floats
 * row major (matrix.h:74-76, :387-389), a Transform its matrix and its inverse (transform.h:40-43, :304-307), a BSphere centre +
 * radius (bsphere.h:40-43, :121-124), a bool one byte (stream.h:192, :279).
 */
#ifndef MTSGPU_STREAMPARSE_H
#define MTSGPU_STREAMPARSE_H

#include <mtsgpu.h>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <stdint.h>

namespace mtsgpu_stream {

template <typename FloatT> struct Matrix44 { FloatT m[4][4]; };
template <typename FloatT> struct Xform { Matrix44<FloatT> fwd, inv; };      /* Transform: m_transform, m_invTransform */

/* A cursor over the bytes of a MemoryStream.  Reading past the end or an unexpected field sets `error` (first one wins) and
 * makes every later read return zeros; callers check ok() once at the end, as SLog(EError) would have thrown. */
template <typename FloatT> class ByteReader {
public:
	ByteReader(const uint8_t *data, size_t size) : m_data(data), m_size(size), m_pos(0) { }
	bool ok() const { return m_error.empty(); }
	const std::string &error() const { return m_error; }
	void fail(const std::string &what) { if (m_error.empty()) m_error = what; }
	size_t pos() const { return m_pos; }
	size_t size() const { return m_size; }
	void setPos(size_t p) { if (p > m_size) fail("seek beyond the end of the stream"); else m_pos = p; }

	uint32_t readUInt() { uint32_t v = 0; raw(&v, 4); return v; }                      /* stream.cpp:275-281 */
	int32_t readInt() { int32_t v = 0; raw(&v, 4); return v; }
	bool readBool() { uint8_t v = 0; raw(&v, 1); return v != 0; }                     /* stream.h:279 */
	FloatT readFloat() { FloatT v = 0; raw(&v, sizeof(FloatT)); return v; }
	std::string readString() {                                                        /* stream.cpp:391-402 */
		std::string s;
		while (ok()) {
			char c = 0; raw(&c, 1);
			if (c == 0) break;
			s += c;
		}
		return s;
	}
	void readSpectrum(float rgb[3]) {                                                 /* rgbOf(): toLinearRGB is the identity for 3 samples */
		for (int i = 0; i < 3; ++i) rgb[i] = (float) readFloat();
	}
	void readVec3(FloatT v[3]) { for (int i = 0; i < 3; ++i) v[i] = readFloat(); }
	Matrix44<FloatT> readMatrix() {
		Matrix44<FloatT> m;
		for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.m[i][j] = readFloat();
		return m;
	}
	Xform<FloatT> readTransform() { Xform<FloatT> t; t.fwd = readMatrix(); t.inv = readMatrix(); return t; }
private:
	void raw(void *dst, size_t n) {
		if (!ok()) { memset(dst, 0, n); return; }
		if (m_pos + n > m_size) { fail("unexpected end of the serialized stream"); memset(dst, 0, n); return; }
		memcpy(dst, m_data + m_pos, n); m_pos += n;
	}
	const uint8_t *m_data; size_t m_size, m_pos;
	std::string m_error;
};

template <typename FloatT> inline void copy3x3(float *dst, const Matrix44<FloatT> &m) {
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) dst[3 * i + j] = (float) m.m[i][j];
}
template <typename FloatT> inline void copy3x4(float *dst, const Matrix44<FloatT> &m) {
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) dst[4 * i + j] = (float) m.m[i][j];
}

/* ------------------------------------------------------------------------------------------------------------------
 * BSDFs.  What <BSDF class>::serialize wrote -- the only access to the plugins' private parameters.  BSDFs have no parent
 * (BSDF::setParent is empty, bsdf.cpp:55-57) and their only children are textures.
 * ---------------------------------------------------------------------------------------------------------------- */
template <typename FloatT> class BSDFStream {
public:
	BSDFStream(const uint8_t *data, size_t size) : r(data, size) {
		if (!openObject(m_className)) r.fail("cannot read back the parameters of a BSDF");
		/* BSDF::serialize (bsdf.cpp:50-53): ConfigurableObject::serialize = the parent (none), then m_name */
		skipReference(); r.readString();
	}
	const std::string &className() const { return m_className; }
	/* a texture child: [id][class name][Texture::serialize = parent reference][ConstantSpectrumTexture: the value]
	 * (src/librender/texture.cpp:39-41, :89-93).  Anything but a constant needs computePartials + MIPMap: out of scope. */
	void readConstantTexture(const char *what, float rgb[3]) {
		const uint32_t id = r.readUInt();
		if (id != 0 && m_values.count(id)) { memcpy(rgb, m_values[id].v, sizeof(float) * 3); return; }      /* one texture shared by two slots */
		if (id == 0) { r.fail(std::string(what) + " of " + m_className + " is missing"); return; }
		m_seen.insert(id);
		const std::string cls = r.readString();
		if (cls != "ConstantSpectrumTexture") {                                                 /* consttexture.h:28-60 */
			r.fail(std::string(what) + " of " + m_className + " is a " + cls + "; only constant reflectances are supported");
			return;
		}
		skipReference();
		r.readSpectrum(rgb);
		memcpy(m_values[id].v, rgb, sizeof(float) * 3);
	}
	/* a ConstantFloatTexture child (roughglass' alpha): [id][class name][Texture::serialize][the value] (texture.cpp:95-103) */
	FloatT readConstantFloatTexture(const char *what) {
		const uint32_t id = r.readUInt();
		if (id == 0 || m_seen.count(id)) { r.fail(std::string(what) + " of " + m_className + " is missing or shared"); return 0; }
		m_seen.insert(id);
		const std::string cls = r.readString();
		if (cls != "ConstantFloatTexture") { r.fail(std::string(what) + " of " + m_className + " is a " + cls + "; only constant values are supported"); return 0; }
		skipReference();
		return r.readFloat();
	}
	/* the nested BRDF of a `twosided` adapter: positions the reader on its fields */
	void enterNestedBSDF() {
		if (!openObject(m_className)) { r.fail("twosided BRDF without a nested BRDF"); return; }
		skipReference(); r.readString();
	}
	ByteReader<FloatT> r;
private:
	struct RGB { float v[3]; };
	bool openObject(std::string &cls) {
		const uint32_t id = r.readUInt();
		if (id == 0 || m_seen.count(id)) return false;
		m_seen.insert(id);
		cls = r.readString();
		return true;
	}
	void skipReference() {
		const uint32_t id = r.readUInt();
		if (id != 0 && !m_seen.count(id)) r.fail("unexpected nested object in the serialized form of " + m_className);
	}
	std::set<uint32_t> m_seen;
	std::map<uint32_t, RGB> m_values;
	std::string m_className;
};

/* One BSDF instance -> its type word (MTSGPU_BSDF_* | MTSGPU_BSDF_TWOSIDED) and parameter block P[MTSGPU_BSDF_NPARAMS]
 * (zeroed by the caller).  Returns false with `err` set for classes and textures that are not on this path. */
template <typename FloatT> inline bool parseBSDF(const uint8_t *data, size_t size, uint32_t *type, float *P, std::string *err) {
	BSDFStream<FloatT> rd(data, size);
	uint32_t flags = 0;
	if (rd.className() == "TwoSidedBRDF") {                                  /* twosided.cpp:52-56: the nested BRDF follows */
		flags |= MTSGPU_BSDF_TWOSIDED;
		rd.enterNestedBSDF();
	}
	const std::string cls = rd.className();
	if (cls == "Lambertian") {                                               /* lambertian.cpp:137-141 */
		*type = MTSGPU_BSDF_LAMBERTIAN | flags;
		rd.readConstantTexture("reflectance", P);
	} else if (cls == "Dielectric") {                                        /* dielectric.cpp:88-95 */
		*type = MTSGPU_BSDF_DIELECTRIC | flags;
		P[0] = (float) rd.r.readFloat(); P[1] = (float) rd.r.readFloat();
		rd.readConstantTexture("specularReflectance", P + 2);
		rd.readConstantTexture("specularTransmittance", P + 5);
	} else if (cls == "RoughMetal") {                                        /* roughmetal.cpp:169-176 */
		*type = MTSGPU_BSDF_ROUGHMETAL | flags;
		rd.readConstantTexture("specularReflectance", P + 7);
		P[0] = (float) rd.r.readFloat();
		rd.r.readSpectrum(P + 1); rd.r.readSpectrum(P + 4);                  /* m_ior, m_k */
	} else if (cls == "Microfacet") {                                        /* microfacet.cpp:283-293 */
		*type = MTSGPU_BSDF_MICROFACET | flags;
		rd.readConstantTexture("diffuseReflectance", P + 5);
		rd.readConstantTexture("specularReflectance", P + 8);
		for (int k = 0; k < 5; ++k) P[k] = (float) rd.r.readFloat();         /* alphaB, kd, ks, intIOR, extIOR */
	} else if (cls == "Mirror") {                                            /* mirror.cpp:51-55 */
		*type = MTSGPU_BSDF_MIRROR | flags;
		rd.r.readSpectrum(P);
	} else if (cls == "Phong") {                                             /* phong.cpp:246-256 (values after configure()) */
		*type = MTSGPU_BSDF_PHONG | flags;
		rd.readConstantTexture("diffuseReflectance", P + 5);
		rd.readConstantTexture("specularReflectance", P + 8);
		for (int k = 0; k < 5; ++k) P[k] = (float) rd.r.readFloat();         /* exponent, kd, ks, specular / diffuse sampling weight */
	} else if (cls == "RoughGlass") {                                        /* roughglass.cpp:735-744 */
		*type = MTSGPU_BSDF_ROUGHGLASS | flags;
		P[0] = (float) rd.r.readInt();                                       /* EBeckmann 0, EPhong 1, EGGX 2 (:84-91) = the ABI's codes */
		P[1] = (float) rd.readConstantFloatTexture("alpha");                 /* phong: already the exponent (:130-136) */
		rd.readConstantTexture("specularReflectance", P + 4);
		rd.readConstantTexture("specularTransmittance", P + 7);
		P[2] = (float) rd.r.readFloat(); P[3] = (float) rd.r.readFloat();    /* intIOR, extIOR */
	} else if (cls == "DiffuseTransmitter") {                                /* difftrans.cpp:142-146 */
		*type = MTSGPU_BSDF_DIFFTRANS | flags;
		rd.readConstantTexture("transmittance", P);
	} else {
		rd.r.fail("BSDF class " + cls + " is not on this path (lambertian, dielectric, roughmetal, microfacet, mirror, phong, "
		          "roughglass, difftrans and the twosided adapter are)");
	}
	if (!rd.r.ok()) { if (err) *err = rd.r.error(); return false; }
	return true;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Scene-level objects (delta luminaires, the environment map, spheres) serialized with their parent taken off
 * (gpucommon.h: serializedDetached): [id]["<class>"][parent = 0][the class's fields]
 * ---------------------------------------------------------------------------------------------------------------- */
template <typename FloatT> inline void openDetached(ByteReader<FloatT> &r, const char *expectedClass) {
	r.readUInt();
	const std::string cls = r.readString();
	if (r.ok() && cls != expectedClass) r.fail(std::string("expected a ") + expectedClass + ", found a " + cls);
	if (r.readUInt() != 0 && r.ok()) r.fail(std::string(expectedClass) + " still has a parent");            /* ConfigurableObject(Stream *), properties.cpp:346-349 */
}
/* Luminaire(Stream *, InstanceManager *) (src/librender/luminaire.cpp:42-51): medium, sampling weight, type, intersectable,
 * worldToLuminaire, name */
template <typename FloatT> inline Xform<FloatT> readLuminaireBase(ByteReader<FloatT> &r) {
	if (r.readUInt() != 0 && r.ok()) r.fail("luminaires inside participating media are not on this path");
	r.readFloat(); r.readInt(); r.readBool();
	const Xform<FloatT> worldToLuminaire = r.readTransform();
	r.readString();
	return worldToLuminaire;
}

/* DirectionalLuminaire (directional.cpp:57-63): direction, intensity, disk origin, disk radius (after preprocess, :65-72) */
template <typename FloatT> inline bool parseDirectional(const uint8_t *data, size_t size, float *P, std::string *err) {
	ByteReader<FloatT> r(data, size);
	openDetached(r, "DirectionalLuminaire");
	readLuminaireBase(r);
	FloatT dir[3], origin[3];
	r.readVec3(dir);
	r.readSpectrum(P);
	r.readVec3(origin); (void) origin;
	P[6] = (float) r.readFloat();
	P[3] = (float) dir[0]; P[4] = (float) dir[1]; P[5] = (float) dir[2];
	if (!r.ok()) { if (err) *err = r.error(); return false; }
	return true;
}

/* SpotLuminaire (spot.cpp:63-70): texture, intensity, beam width, cutoff angle; configure() (:56-62) derives the rest */
template <typename FloatT> inline bool parseSpot(const uint8_t *data, size_t size, float *P, std::string *err) {
	ByteReader<FloatT> r(data, size);
	openDetached(r, "SpotLuminaire");
	const Xform<FloatT> w2l = readLuminaireBase(r);
	{	/* m_texture: only the default constant 1 is on this path (spot.cpp:42-43, :96-102) */
		r.readUInt();
		if (r.readString() != "ConstantSpectrumTexture" && r.ok()) r.fail("projection textures of spot luminaires are not on this path");
		r.readUInt();                                                        /* Texture::serialize: its parent (the luminaire: a known id, or none) */
		float tex[3]; r.readSpectrum(tex);
		if (r.ok() && (tex[0] != 1.0f || tex[1] != 1.0f || tex[2] != 1.0f)) r.fail("projection textures of spot luminaires are not on this path");
	}
	r.readSpectrum(P);
	const FloatT beamWidth = r.readFloat(), cutoffAngle = r.readFloat();
	/* m_luminaireToWorld(Point(0, 0, 0)) (transform.h:133-149): the translation column, divided by w unless it is 1 */
	FloatT x = w2l.inv.m[0][3], y = w2l.inv.m[1][3], z = w2l.inv.m[2][3];
	const FloatT w = w2l.inv.m[3][3];
	if (w != (FloatT) 1) { const FloatT rc = (FloatT) 1 / w; x *= rc; y *= rc; z *= rc; }
	P[3] = (float) x; P[4] = (float) y; P[5] = (float) z;
	P[6] = (float) std::cos(beamWidth); P[7] = (float) std::cos(cutoffAngle);
	P[8] = (float) cutoffAngle; P[9] = (float) ((FloatT) 1 / (cutoffAngle - beamWidth));
	copy3x3(P + 10, w2l.fwd);
	P[19] = (float) beamWidth;
	if (!r.ok()) { if (err) *err = r.error(); return false; }
	return true;
}

/* CollimatedBeamLuminaire (collimated.cpp:47-51): intensity, radius */
template <typename FloatT> inline bool parseCollimated(const uint8_t *data, size_t size, float *P, std::string *err) {
	ByteReader<FloatT> r(data, size);
	openDetached(r, "CollimatedBeamLuminaire");
	const Xform<FloatT> w2l = readLuminaireBase(r);
	r.readSpectrum(P);
	P[3] = (float) r.readFloat();
	copy3x4(P + 4, w2l.fwd);
	copy3x4(P + 16, w2l.inv);
	if (!r.ok()) { if (err) *err = r.error(); return false; }
	return true;
}

/* EnvMapLuminaire (envmap.cpp:79-93): intensity scale, path, bounding sphere (after preprocess, :112-126), then the size and
 * the bytes of the EXR file.  Fills the parameter block and tells where the bitmap's bytes are; decoding them needs Mitsuba's
 * Bitmap / MIPMap and stays in gpucommon.h */
template <typename FloatT> inline bool parseEnvMapHeader(const uint8_t *data, size_t size, float *P, size_t *exrOffset, uint32_t *exrSize, std::string *err) {
	ByteReader<FloatT> r(data, size);
	openDetached(r, "EnvMapLuminaire");
	const Xform<FloatT> w2l = readLuminaireBase(r);
	P[0] = (float) r.readFloat();                                            /* m_intensityScale */
	r.readString();                                                          /* m_path */
	FloatT c[3]; r.readVec3(c);
	P[3] = (float) c[0]; P[4] = (float) c[1]; P[5] = (float) c[2]; P[6] = (float) r.readFloat();
	copy3x3(P + 7, w2l.fwd); copy3x3(P + 16, w2l.inv);
	*exrSize = r.readUInt();
	*exrOffset = r.pos();
	if (r.ok() && r.pos() + *exrSize > r.size()) r.fail("the environment map's bitmap is truncated");
	if (!r.ok()) { if (err) *err = r.error(); return false; }
	return true;
}

/* Sphere (sphere.cpp:72-78) behind Shape::serialize (shape.cpp:130-138), whose nested BSDF / luminaire are skipped by seeking
 * from the END, where the sphere's own fields have a fixed size: objectToWorld, radius, centre, inverted */
template <typename FloatT> inline bool parseSphere(const uint8_t *data, size_t size, float *SP, std::string *err) {
	ByteReader<FloatT> r(data, size);
	openDetached(r, "Sphere");
	const size_t tail = 2 * 16 * sizeof(FloatT) + sizeof(FloatT) + 3 * sizeof(FloatT) + 1;
	if (r.ok() && size < tail + r.pos()) r.fail("the serialized sphere is too short");
	if (r.ok()) r.setPos(size - tail);
	const Xform<FloatT> o2w = r.readTransform();
	const FloatT radius = r.readFloat();
	FloatT c[3]; r.readVec3(c);
	const bool inverted = r.readBool();
	SP[0] = (float) c[0]; SP[1] = (float) c[1]; SP[2] = (float) c[2]; SP[3] = (float) radius;
	SP[4] = inverted ? 1.0f : 0.0f;
	copy3x3(SP + 5, o2w.fwd); copy3x3(SP + 14, o2w.inv);
	const FloatT pi = (FloatT) 3.14159265358979323846;                       /* Mitsuba's M_PI has Float's precision (constants.h:38-53) */
	SP[23] = (float) (1 / (4 * pi * radius * radius));                       /* m_invSurfaceArea (sphere.cpp:59, :69) */
	if (!r.ok()) { if (err) *err = r.error(); return false; }
	return true;
}

} /* namespace mtsgpu_stream */

#endif /* MTSGPU_STREAMPARSE_H */
