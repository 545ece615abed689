// Device-side parameter block of the tiled raster / ray-cast kernels and the
// host launcher.  Layouts are described in DESIGN.md section 4.
#pragma once

#include <cstdint>
#include <hip/hip_runtime_api.h>

namespace mrx {

// Object-space triangle: 16 dwords, one 64-byte line.
struct alignas(16) ObjTri {
    float p[9];      // 3 vertices x xyz
    float uv[6];     // 3 vertices x uv
    int32_t mat;     // material index, -1 = default material
};
static_assert(sizeof(ObjTri) == 64, "ObjTri must be one 64-byte line");

struct alignas(16) Material {
    float color[4];
    int32_t tex;     // texture index, -1 = untextured
    int32_t pad[3];
};
static_assert(sizeof(Material) == 32, "Material layout");

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
    const Material *materials;
    const TexDesc *textures;
    const uint32_t *texels;          // RGBA8
    uint32_t numMaterials, numTextures;
    // per-world tables
    const WorldTri *worldTris;
    const uint32_t *worldTriStart;   // [worlds + 1]
    const uint32_t *viewWorld;       // [views]
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
    float toLight[3];
    float ambient, diffuse;
    float defaultColor[4];
    int32_t transposed;              // Raytracer-mode [x][y] storage
    int32_t idsAreSegmask;           // ids buffer holds objectID instead of tri index
};

// Kernel variants (mrx_config.kernel_variant).
enum KernelVariant : int32_t {
    kVariantDefault = 0,
    kVariantBrute = 1,       // v1: every triangle tested at every pixel
    kNumVariants
};

hipError_t launchRaster(const RasterParams &p, uint32_t maxWorldTris,
                        int32_t variant, hipStream_t stream);

}  // namespace mrx
