// Host side of the drop-in boundary: reader for Mitsuba's `.serialized` triangle-mesh container, i.e. what
// TriMesh::TriMesh(Stream *, int index) does (src/librender/trimesh.cpp:156-236): optional seek to a sub-shape
// through the offset table at the end of the file, the 0x041C / version 3 header, then a zlib stream holding
// flags, vertex / triangle counts (64 bit), positions, optional normals / texture coordinates / colours in single
// or double precision, and the 32-bit index triples.  The result is handed over as a mtsgpu_mesh.
#include "host.h"
#include <zlib.h>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace mg {
namespace {

enum : uint32_t {      // ETriMeshFlags (trimesh.cpp:72-80)
	kHasNormals = 0x0001, kHasTexcoords = 0x0002, kHasTangents = 0x0004, kHasColors = 0x0008,
	kFaceNormals = 0x0010, kSinglePrecision = 0x1000, kDoublePrecision = 0x2000
};

std::vector<uint8_t> readFile(const char *path) {
	FILE *f = std::fopen(path, "rb");
	if (!f) throw std::runtime_error(std::string("cannot open ") + path);
	std::fseek(f, 0, SEEK_END);
	const long size = std::ftell(f);
	std::fseek(f, 0, SEEK_SET);
	std::vector<uint8_t> data(size > 0 ? (size_t) size : 0);
	const size_t got = data.empty() ? 0 : std::fread(data.data(), 1, data.size(), f);
	std::fclose(f);
	if (got != data.size()) throw std::runtime_error(std::string("short read on ") + path);
	return data;
}

// ZStream over the rest of the file (src/libcore/zstream.cpp:39: inflateInit, i.e. the zlib container)
struct Inflater {
	z_stream zs;
	explicit Inflater(const uint8_t *src, size_t n) {
		std::memset(&zs, 0, sizeof(zs));
		if (inflateInit(&zs) != Z_OK) throw std::runtime_error("inflateInit failed");
		zs.next_in = const_cast<Bytef *>(src);
		zs.avail_in = (uInt) std::min<size_t>(n, 0xFFFFFFFFu);
	}
	~Inflater() { inflateEnd(&zs); }
	void read(void *dst, size_t n) {
		uint8_t *out = static_cast<uint8_t *>(dst);
		while (n > 0) {
			const uInt chunk = (uInt) std::min<size_t>(n, 1u << 30);
			zs.next_out = out; zs.avail_out = chunk;
			const int rc = inflate(&zs, Z_NO_FLUSH);
			const size_t produced = chunk - zs.avail_out;
			out += produced; n -= produced;
			if (rc == Z_STREAM_END && n > 0) throw std::runtime_error("serialized mesh: compressed stream ends early");
			if (rc != Z_OK && rc != Z_STREAM_END) throw std::runtime_error("serialized mesh: corrupt compressed stream");
			if (produced == 0 && rc == Z_OK && zs.avail_in == 0) throw std::runtime_error("serialized mesh: truncated file");
		}
	}
	template <typename T> T get() { T v; read(&v, sizeof(T)); return v; }
	// readHelper (trimesh.cpp:122-153): file precision -> float
	void readFloats(bool fileDouble, float *dst, size_t count) {
		if (!fileDouble) { read(dst, count * sizeof(float)); return; }
		std::vector<double> tmp(count);
		read(tmp.data(), count * sizeof(double));
		for (size_t i = 0; i < count; ++i) dst[i] = (float) tmp[i];
	}
};

} // namespace

void loadSerializedMesh(const char *path, int index, LoadedMesh &out) {
	const std::vector<uint8_t> file = readFile(path);
	size_t pos = 0;
	if (index != 0) {
		// offset table at the end of the file: [offset_0 ... offset_{count-1}] count  (trimesh.cpp:160-173)
		if (file.size() < 4) throw std::runtime_error("serialized mesh: file too small");
		uint32_t count; std::memcpy(&count, &file[file.size() - 4], 4);
		if (index < 0 || (uint32_t) index > count || (size_t) 4 * (1 + (size_t) count) > file.size())
			throw std::runtime_error("serialized mesh: shape index is out of range");
		uint32_t off; std::memcpy(&off, &file[file.size() - 4 * (size_t) (1 + count - (uint32_t) index)], 4);
		pos = off;
	}
	if (pos + 4 > file.size()) throw std::runtime_error("serialized mesh: truncated header");
	uint16_t format, version;
	std::memcpy(&format, &file[pos], 2); std::memcpy(&version, &file[pos + 2], 2);
	if (format == 0x1C04) throw std::runtime_error("serialized mesh: file written by an old version of Mitsuba");
	if (format != 0x041C) throw std::runtime_error("serialized mesh: invalid file format");
	if (version != 0x03) throw std::runtime_error("serialized mesh: incompatible file version");
	Inflater z(&file[pos + 4], file.size() - pos - 4);
	const uint32_t flags = z.get<uint32_t>();
	const uint64_t nv = z.get<uint64_t>(), nt = z.get<uint64_t>();
	if (nv == 0 || nt == 0 || nv >= 0xFFFFFFFFull || nt >= 0x7FFFFFFFull)
		throw std::runtime_error("serialized mesh: empty or oversized mesh");
	const bool dbl = (flags & kDoublePrecision) != 0;
	out.positions.resize(3 * nv);
	z.readFloats(dbl, out.positions.data(), 3 * nv);
	out.normals.clear();
	if (flags & kHasNormals) { out.normals.resize(3 * nv); z.readFloats(dbl, out.normals.data(), 3 * nv); }
	if (flags & kHasTexcoords) { std::vector<float> skip(2 * nv); z.readFloats(dbl, skip.data(), 2 * nv); }     // no textures on this path
	if (flags & kHasColors) { std::vector<float> skip(3 * nv); z.readFloats(dbl, skip.data(), 3 * nv); }
	out.triangles.resize(3 * nt);
	z.read(out.triangles.data(), 3 * nt * sizeof(uint32_t));
	for (uint32_t v : out.triangles)
		if (v >= nv) throw std::runtime_error("serialized mesh: vertex index out of range");
	out.faceNormals = (flags & kFaceNormals) != 0;
}

} // namespace mg
