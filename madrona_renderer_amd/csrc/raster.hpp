// Device-side parameter block of the tiled raster / ray-cast kernels and the
// host launcher.  Layouts are described in DESIGN.md section 4.
#pragma once

#include <cstdint>
#include <hip/hip_runtime_api.h>

namespace mrx {

struct BvhNode;
struct ObjInfo;

// Object-space triangle: 16 dwords, one 64-byte line.
struct alignas(16) ObjTri {
    float p[9];      // 3 vertices x xyz
    float uv[6];     // 3 vertices x uv
    int32_t mat;     // material index, -1 = default material
};
static_assert(sizeof(ObjTri) == 64, "ObjTri must be one 64-byte line");

// Material of one object triangle, resolved on the host (mesh material or the
// default colour; texture index validated): one 32-byte record next to ObjTri.
struct alignas(16) TriMat {
    // rgb of the resolved material; the alpha slot (nothing shades with it) holds the
    // int32 id of the object the triangle belongs to -- the segmask label, so that label
    // and geometry agree whatever non-negative value the live ObjectID column holds
    float color[4];
    int32_t tex;       // texture index, -1 = untextured
    // S6b back-face culling data of the triangle's shell (edge-connected
    // component of its object): orient = +1 / -1 when the shell is a closed,
    // consistently wound mesh (sign of its volume), 0 otherwise; bb = the
    // shell's object-space bounding box, padded
    float orient;
    float bbMin[3];
    float bbMax[3];
    int32_t texDesc[4];   // TexDesc of `tex` (offset, width, height, 0), zero when untextured
};
static_assert(sizeof(TriMat) == 64, "TriMat layout");

struct TexDesc {
    uint32_t offset;  // texel offset into the RGBA8 pool
    uint32_t width, height;
    uint32_t pad;
};

// One world-triangle slot: which instance row draws which object triangle.
struct WorldTri {
    uint32_t inst;   // row of the world-major instance tables
    uint32_t tri;    // index into the ObjTri pool
};

struct RasterParams {
    // shared read-only scene
    const ObjTri *tris;
    const TriMat *triMats;           // parallel to tris
    const TexDesc *textures;
    const uint32_t *texels;          // RGBA8
    int32_t anyTextured;             // some drawn triangle has a texture
    // per-view draw lists: view v draws viewTris[v * viewTriStride + k],
    // k < viewTriCount[v] -- one load level between the view index and the
    // triangle's pose / geometry rows
    const WorldTri *viewTris;
    const uint32_t *viewTriCount;    // [views]
    uint32_t viewTriStride;
    // Uniform worlds (every world: the same <= 4 objects in the same order, the
    // same camera count): the draw list is arithmetic on kernel arguments --
    // slot k of a view belongs to local instance i with uniPrefix[i] <= k <
    // uniPrefix[i+1], object triangle uniFirstTri[i] + k - uniPrefix[i] -- so
    // no table load sits between the view index and the pose / geometry rows.
    uint32_t uniInstances;           // instances per world, 0 = not uniform
    uint32_t uniCamsPerWorld;
    uint32_t uniPrefix[5];
    uint32_t uniFirstTri[4];
    // the blocks the pose tensors / geometry tables are slices of (GroupHeader; filled in by mrx_create)
    const char *poseBlock;
    const char *geomBlock;
    uint32_t numInstances, poolTris;
    // pose state (the exported, mutable tensors)
    const float *instPos;            // [I][3]
    const float *instRot;            // [I][4] w,x,y,z
    const float *instScale;          // [I][3]
    const int32_t *instObj;          // [I]
    const float *camPos;             // [V][3]
    const float *camRot;             // [V][4]
    // outputs, storage order [view][slow][fast]
    uint32_t *rgb;
    float *depth;
    int32_t *ids;                    // visibility ids or segmask, may be null
    uint32_t numViews;
    uint32_t nfast, nslow;           // pixels per row, rows per view
    uint32_t tilesFast, tilesSlow;   // 64x64 tiles per view
    // pixel -> ray constants
    float sx, ox, sz, oz;
    float invNear, invFar;
    // S6b: the eye counts as outside a shell only when it is further than this
    // from the shell's box -- no front face can then be cut by the near plane
    float s6bPad;
    float toLight[3];
    float ambient, diffuse;
    int32_t transposed;              // Raytracer-mode [x][y] storage
    int32_t idsAreSegmask;           // ids buffer holds objectID instead of tri index
    // Timing-only ablation switches (MRX_DEBUG_SKIP env, never set in
    // production): 1 skip stores, 2 skip raster, 4 skip classification,
    // 8 skip triangle setup.  Outputs are wrong when any bit is set.
    uint32_t debugSkip;
    // Output store policy: 1 = write-through (sc1), 0 = plain write-back.
    uint32_t writeThrough;
    // Shape of the group kernel's workgroups (filled in by launchRaster):
    // grpPerView == 1: a workgroup owns grpViews whole views; otherwise it owns
    // grpChunkTiles tiles of one view and a view takes grpPerView workgroups.
    // grpViewsWanted / grpTilesWanted: MRX_GROUP_VIEWS / MRX_GROUP_TILES
    // tuning overrides (0 = automatic).
    uint32_t grpViews, grpChunkTiles, grpPerView;
    int32_t grpViewsWanted, grpTilesWanted;
    // XCD-aware split of the group kernel (filled in by launchRaster);
    // xcdSkew = strips moved per workgroup pair; xcdSkewWanted is the
    // MRX_XCD_SKEW override: -1 automatic, 0 off, 1..7 strips.
    uint32_t xcdSkew;
    int32_t xcdSkewWanted;
    // XCD phase feedback of the split: workgroup 0 of every launch writes its XCC id to
    // a host-mapped word; its parity, read back by the host before a later launch, makes
    // the workgroups of each pair trade places (raster.hip)
    uint32_t *xccReport;
    uint32_t xcdPhase;
    // rotate the group <-> XCD relation by two every round of eight workgroups
    // (MRX_XCD_ROTATE override: -1 automatic, 0 off, 1 on)
    uint32_t xcdRotate;
    int32_t xcdRotateWanted;
    int32_t debugSlots;              // MRX_DEBUG_SLOTS: force 32 / 64 triangle slots per tile
    // Diagnostic only (MRX_DEBUG_STAMPS=1): per-wave s_memrealtime stamps,
    // [workgroup][wave][8], written to memory nothing else reads.
    unsigned long long *debugStamps;
    // ---- BVH path (bvh.hip): per-object BLAS built at load (bvh.hpp), the
    //      world -> instance-row and view -> world tables the per-step TLAS is
    //      built from, and each instance's first world-local triangle index
    const BvhNode *bvhNodes;
    const uint32_t *bvhLeafTris;
    const ObjInfo *instInfo;         // [I]: the ObjInfo of each instance's (creation-time) object
    uint32_t numObjects;
    uint32_t bvhUniInst, bvhUniCams; // every world: this many instances / cameras (0: look the tables up)
    const uint32_t *worldInstStart;  // [worlds + 1]
    const uint32_t *viewWorld;       // [views]
    const uint32_t *instKBase;       // [I]
    uint32_t bvhPassInst;            // instances whose TLAS records fit LDS at once (a multiple of 8)
    int32_t bvhTile;                 // tile of a workgroup: 0 = 64x64, 1 = 64 wide x 32, 2 = 32x32 (MRX_BVH_TILE)
    int32_t bvhSmallArea;            // boxes of up to this many pixels are walked by their triangle's lane
    int32_t bvhClassify;             // 64x64 tiles: the instantiation that classifies listed triangles per strip
    // tiles of a view one workgroup renders in turn over one TLAS build (1 when a world needs several
    // TLAS passes; filled in by the host, MRX_BVH_GROUP_TILES overrides)
    uint32_t bvhGroupTiles;
    // one-tile views a workgroup renders in turn, their TLASes built side by side in one phase I (a power of
    // two; 1 unless every world fits one TLAS pass; filled in by the host, MRX_BVH_GROUP_VIEWS overrides);
    // bit 16 (two views): a launch of as many workgroups as the chip holds, pairs on the first of them and single
    // views on the others; bits 17..19: wave-priority mode of the younger workgroups (bvh.hip; 0 off); bits 20..31:
    // the index of the first workgroup that is a CU's second (= the number of CUs)
    uint32_t bvhGroupViews;
    // every world holds at most 64 triangles in at most 64 instance rows: the kernel that sets a view's
    // triangles up once and shares the per-tile work among all waves (bvh.hip, bvhFlatKernel; MRX_BVH_FLAT=0: never)
    uint32_t bvhFlat;
    // textured BVH instantiations: 48-byte shading records per round of a tile (64 ... 1023; chosen by the host so that
    // the workgroup's LDS stays within half a CU's: mrx_api.cpp chooseBvhGroups)
    uint32_t bvhTexCap;
    // compute units of the device the renderer runs on (hipDeviceAttributeMultiprocessorCount at creation):
    // every "does the batch fill the chip" decision of the launchers follows from it (groupFill below)
    uint32_t numCUs;
};

// ---- the argument header of the group kernel's fast prologue (raster.hip, FAST) -------------------------------------
// The command processor can write the first dwords of a kernel's argument block into SGPRs at wave launch
// (-mllvm -amdgpu-kernarg-preload-count): a kernel whose first loads need only those starts them without the
// s_load round trip to the argument block (a fresh copy, hence a cache miss, on every launch: 0.24 us in
// scripts/micro/kernarg_preload.hip).  Fourteen dwords at most, so everything the set-up waves of a
// uniform-world, one-tile-per-view launch need ahead of their pose loads is packed into twelve: the pose
// tensors live in ONE block and the geometry tables in another, at offsets that follow from the counts.
#if defined(__HIP__) || defined(__HIPCC__)
#define MRX_HD __host__ __device__
#else
#define MRX_HD
#endif
struct PoseLayout {
    uint32_t camRot, camPos, instRot, instPos, instScale, instObj, total;   // byte offsets, 256-byte aligned
};
MRX_HD inline uint32_t mrxAlign256(uint32_t x) { return (x + 255u) & ~255u; }
MRX_HD inline PoseLayout poseLayout(uint32_t views, uint32_t instances)
{
    PoseLayout l;
    l.camRot = 0;
    l.camPos = l.camRot + mrxAlign256(views * 16u);
    l.instRot = l.camPos + mrxAlign256(views * 12u);
    l.instPos = l.instRot + mrxAlign256(instances * 16u);
    l.instScale = l.instPos + mrxAlign256(instances * 12u);
    l.instObj = l.instScale + mrxAlign256(instances * 12u);
    l.total = l.instObj + mrxAlign256(instances * 4u);
    return l;
}
// geometry block: ObjTri[pool] at 0, TriMat[pool] at geomMatsOffset(pool)
MRX_HD inline uint32_t geomMatsOffset(uint32_t poolTris) { return mrxAlign256(poolTris * 64u); }

struct GroupHeader {
    const char *pose;        // base of the pose block (poseLayout(views, instances))
    const char *geom;        // base of the geometry block
    uint32_t views, instances, poolTris;
    // grpViews | xcdSkew << 8 | xcdRotate << 12 | uniInstances << 13 | uniCamsPerWorld << 16 | valid << 31
    uint32_t shape;
    uint32_t groups;         // workgroups of the launch
    uint32_t prefix;         // uniPrefix[1..4], eight bits each
    uint32_t first01, first23;   // uniFirstTri[0..3], sixteen bits each
};

// Default dispatch: worlds of this many triangles and more take the BVH path
// (measured crossover, profiles/r02_bvh_crossover.txt; MRX_BVH_MIN_TRIS overrides; small batches of 64x64 views
// cross earlier -- from 65 triangles up to 640 views, from 91 up to 1024 untextured ones: mrx_api.cpp).
constexpr uint32_t kBvhMinTris = 129;

// Kernel variants (mrx_config.kernel_variant).
enum KernelVariant : int32_t {
    kVariantDefault = 0,     // group kernel up to kBvhMinTris-1 triangles per world, BVH above
    kVariantBrute = 1,       // v1: every triangle tested at every pixel
    kVariantBvh = 2,         // BVH path whatever the scene size
    kVariantRaster = 3,      // never the BVH path (group kernel, chunked kernel above 256)
    kNumVariants
};

// ---- launch-shape constants as functions of the device (they were MI355X literals) ----------------
// Workgroups of the group kernel a launch needs before the chip counts as full: a CU holds 2048 threads = four
// workgroups of 512 (the untextured kernel; the textured one has 256 threads and longer tiles -- same target).
inline uint32_t groupFill(uint32_t numCUs) { return 4u * (numCUs ? numCUs : 256u); }
// Triangles per world from which the default dispatch takes the BVH path (mrx_api.cpp bindGeometry).  `base` is the
// general threshold (kBvhMinTris or MRX_BVH_MIN_TRIS).  Small batches of one-tile views cross earlier, because up to two
// workgroups per CU the BVH kernel's launch costs what its slowest workgroup costs while the raster kernels' 128-slot
// shape pays for its slots -- measured on 256 CUs (profiles/r03_bvh_threshold.txt): from 65 triangles up to 640 views =
// 2.5 per CU (512 views of 74 triangles 10.7 against 12.3 us, textured 13.0 against 15.5), from 91 up to 1024 = 4 per CU
// for untextured worlds (1024 views of 98 triangles 18.0 against 18.6); larger batches cross at `base`.
inline uint32_t bvhDispatchMinTris(uint32_t base, uint32_t numViews, bool anyTextured, uint32_t nfast, uint32_t nslow,
                                   uint32_t numCUs)
{
    const uint32_t cus = numCUs ? numCUs : 256u;
    if (nfast > 64u || nslow > 64u)
        return base;
    if (2ull * numViews <= 5ull * cus)
        return base < 65u ? base : 65u;
    if (numViews <= 4u * cus && !anyTextured)
        return base < 91u ? base : 91u;
    return base;
}

// Raytracer-mode batches of small worlds that the default dispatch gives to the BVH path's flat kernel (bvh.hip,
// bvhFlatKernel: worlds of <= 64 triangles in <= 64 rows) instead of the raster group kernel: views of at least 16
// tiles, at least 192 tiles per compute unit in the batch -- BASELINE configs[4], 4096 views of 256x256 = 65536 tiles on 256
// CUs, is the measured case (474 - 483 us against 488 - 494 on the same boxes, profiles/r04_c5_shape_by_views.txt,
// r04_bench_k20_kernel_stats.csv); at 2048 views and fewer, and in Rasterizer mode (no segmask: two output tensors), the
// raster kernel is ahead by 1 - 3 % and keeps the batch.
inline bool bvhDispatchFlat(bool raytracer, uint32_t numViews, uint32_t nfast, uint32_t nslow, uint32_t maxWorldTris,
                            uint32_t maxWorldInstances, uint32_t numCUs)
{
    const uint32_t cus = numCUs ? numCUs : 256u;
    const uint64_t tpv = (uint64_t)((nfast + 63u) / 64u) * ((nslow + 63u) / 64u);
    return raytracer && maxWorldTris >= 1u && maxWorldTris <= 64u && maxWorldInstances <= 64u && tpv >= 16u &&
           (uint64_t)numViews * tpv >= 192ull * cus;
}

hipError_t launchRaster(const RasterParams &p, uint32_t maxWorldTris,
                        int32_t variant, hipStream_t stream);

// BVH path: per-step TLAS in LDS, wave-packet traversal of TLAS + BLAS,
// exact S6 leaf test (bvh.hip).  p.bvhPassInst is filled in by the caller.
hipError_t launchBvh(const RasterParams &p, hipStream_t stream);
constexpr uint32_t kBvhMaxWorldTris = 0x1FFFFEu;   // the depth buffer's key holds 21 bits of triangle index
// dynamic LDS bytes one workgroup of the BVH kernel needs for `passInst` instance records
size_t bvhLdsBytes(uint32_t passInst, bool textured, bool classify, uint32_t groupViews, uint32_t texCap);

}  // namespace mrx
