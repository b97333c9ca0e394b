// host.h -- host-side pieces of libmtsgpu shared between translation units.
#pragma once
#include "../../include/mtsgpu.h"
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mg {

struct KdTree {
	std::vector<uint32_t> nodes;      // 2 dwords per node
	std::vector<uint32_t> indices;
	float aabbMin[3] = { 0, 0, 0 }, aabbMax[3] = { 0, 0, 0 };
	double stats[6] = { 0, 0, 0, 0, 0, 0 };
};

// genBox: [nTris][6] boxes, read for the primitives whose tri row is {MTSGPU_KNOTRIANGLE x 3}; may be NULL without such
void buildKdTree(const float *vtx, const uint32_t *tri, uint32_t nTris, const float *genBox, const mtsgpu_kd_params *kp, KdTree &out);
bool clippedTriangleBox(const float *p0, const float *p1, const float *p2, const float *bmin, const float *bmax,
                        float *omin, float *omax);

// Owns every array a mtsgpu_scene points to
struct FlatScene {
	mtsgpu_scene sc;
	KdTree kd;
	std::vector<float> envPixels, envPdf, envCdf;
	std::vector<float> vtxPos, vtxNrm, shapeParams, bsdfParams, lumParams, lumInvArea, lumTriCdf, lumSelCdf, lumSelPdf;
	std::vector<uint32_t> triIdx, shapeTriOffset, shapeFlags, shapeType, triaccel, bsdfType, lumType, lumCdfOffset;
	std::vector<int32_t> shapeBsdf, shapeLum, lumShape;
};

void flattenScene(const mtsgpu_scene_desc &d, const mtsgpu_kd_params *kp, FlatScene &fs);

// One shape of a `.serialized` file (TriMesh::TriMesh(Stream *, int), src/librender/trimesh.cpp:156-236)
struct LoadedMesh {
	std::vector<float> positions, normals;     // normals empty when the file has none
	std::vector<uint32_t> triangles;
	bool faceNormals = false;
};
void loadSerializedMesh(const char *path, int index, LoadedMesh &out);   // throws std::runtime_error
// TabulatedFilter of the box / gaussian / mitchell / catmullrom / wsinc plugins (kinds 0..4): sizeXY[2], values[16*16]
void tabulateFilter(int kind, float halfSize, float p0, float p1, float *sizeXY, float *values);
void makeCamera(const float origin[3], const float target[3], const float up[3], float fovDeg, int width, int height,
                mtsgpu_camera &out);
void makeCameraOrtho(const float origin[3], const float target[3], const float up[3], float scaleX, float scaleY,
                     int width, int height, mtsgpu_camera &out);

} // namespace mg
