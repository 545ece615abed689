// Bottom-level acceleration structures (BLAS) of the ray-trace path: one
// 8-wide bounding volume hierarchy per object, built once on the host at load
// time.  This is the counterpart of the reference's
//   render::AssetProcessor::makeBVHData      /root/reference/src/mgr.cpp:472-473
// (the un-vendored Madrona MeshBVH builder); the top level over a world's
// instances is rebuilt on the device every step (bvh.hip).
//
// Node width 8 x 8 box corners = the 64 lanes of a wavefront: one lane projects
// one corner of one child box, an 8-lane reduction gives the child's screen
// rectangle (bvh.hip, node visit).
#pragma once

#include <cstdint>
#include <vector>

#include "raster.hpp"

namespace mrx {

constexpr uint32_t kBvhWidth = 8;
constexpr uint32_t kBvhLeafMax = 16;      // triangles per leaf
constexpr uint32_t kBvhFlatMax = 32;      // objects up to this size have no BLAS: the
                                          // instance test alone selects their triangles
constexpr uint32_t kBvhEmpty = 0xFFFFFFFFu;
constexpr uint32_t kBvhLeafBit = 0x80000000u;
constexpr uint32_t kBvhLeafStartBits = 26;     // leaf = bit31 | (count-1) << 26 | start
constexpr uint32_t kBvhStackCap = 64;          // traversal stack entries per wave

// 256 bytes.  child[c]: kBvhEmpty, a leaf (kBvhLeafBit | (count - 1) << 26 |
// first entry of the leaf in the leaf-triangle list) or the index of an inner node.
struct alignas(16) BvhNode {
    float bmin[kBvhWidth][3];
    float bmax[kBvhWidth][3];
    uint32_t child[kBvhWidth];
    uint32_t pad[8];
};
static_assert(sizeof(BvhNode) == 256, "BvhNode layout");

// 48 bytes, read as three float4 by the instance phase.
struct alignas(16) ObjInfo {
    uint32_t firstTri, numTris;   // the object's range of the ObjTri pool
    int32_t root;                 // root node, -1: flat (<= kBvhFlatMax triangles)
    uint32_t pad;
    float bbMin[4], bbMax[4];     // exact bounds of the object's vertices (w unused)
};
static_assert(sizeof(ObjInfo) == 48, "ObjInfo layout");

struct BlasSet {
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> leafTris;   // indices into the ObjTri pool
    std::vector<ObjInfo> objects;
    uint32_t maxDepth = 0;            // deepest inner-node chain of any object
};

// Builds the BLAS of every object (binned surface-area heuristic, leaves of at
// most kBvhLeafMax triangles, binary tree collapsed to 8-wide nodes).  Objects
// whose tree would need more than kBvhStackCap stack entries are rebuilt with
// balanced median splits.
void buildBlas(const ObjTri *tris, const std::vector<int32_t> &objFirst,
               const std::vector<int32_t> &objCount, BlasSet &out);

}  // namespace mrx
