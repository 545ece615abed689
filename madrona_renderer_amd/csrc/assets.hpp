// Host-side asset ingestion for the batch renderer: a minimal Wavefront OBJ
// reader and a PNG decoder.  These stand in for the un-vendored
// madrona::imp::AssetImporter / ImageImporter the reference calls at
// /root/reference/src/mgr.cpp:294-307 and :318-319.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace mrx {

// Object-space triangle soup of one OBJ file.  objStart lists the first triangle
// of each `o` / `g` block that has faces (faces ahead of the first statement form a
// block of their own).  mrx_create makes one object of the whole file, like the
// reference (importFromDisk(..., one_object_per_asset = true),
// /root/reference/src/mgr.cpp:301-303; objects[i] <-> asset path i, :340-345); the
// blocks only become objects of their own under MRX_OBJ_SPLIT_BLOCKS=1.
struct TriSoup {
    std::vector<float> pos;  // [T][3 verts][xyz]
    std::vector<float> uv;   // [T][3 verts][uv]
    std::vector<uint32_t> objStart;   // first triangle of each block, ascending; at least one entry
    // `usemtl` of each triangle as an index into mtlNames, -1 = none
    std::vector<int32_t> triMtl;
    std::vector<std::string> mtlNames;
    std::vector<std::string> mtlLibs;   // `mtllib` files, resolved against the OBJ's directory
    uint32_t numTris() const { return (uint32_t)(pos.size() / 9); }
};

// One `newmtl` block of a Wavefront MTL file (the subset this renderer shades with).
struct MtlMaterial {
    std::string name;
    float kd[3] = { 1.0f, 1.0f, 1.0f };
    std::string mapKd;                  // resolved against the MTL's directory, "" = none
};

struct Image {
    std::vector<uint8_t> rgba;  // [h][w][4]
    uint32_t width = 0, height = 0;
};

// Both return false and fill `err` on failure.
bool loadOBJ(const std::string &path, TriSoup &out, std::string &err);
bool loadMTL(const std::string &path, std::vector<MtlMaterial> &out, std::string &err);
bool decodePNG(const std::string &path, Image &out, std::string &err);
bool decodePNGMem(const uint8_t *data, size_t size, Image &out, std::string &err);
// KTX2 container with a BC7 or RGBA8 base level -> RGBA8 (ktx2.cpp; the
// reference's "ktx2" handler, /root/reference/src/mgr.cpp:199-212,297-298).
bool decodeKTX2(const std::string &path, Image &out, std::string &err);
bool decodeKTX2Mem(const uint8_t *data, size_t size, Image &out, std::string &err);
void decodeBC7Block(const uint8_t block[16], uint8_t out[16][4]);
// By extension: .ktx2 -> decodeKTX2, anything else -> decodePNG.
bool decodeTexture(const std::string &path, Image &out, std::string &err);
// 8-bit RGBA, non-interlaced, zlib-deflated, filter type 0 on every scanline.
bool encodePNG(const std::string &path, const uint8_t *rgba, uint32_t width, uint32_t height,
               std::string &err);

}  // namespace mrx
