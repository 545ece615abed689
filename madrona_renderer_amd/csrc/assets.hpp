// Host-side asset ingestion for the batch renderer: a minimal Wavefront OBJ
// reader and a PNG decoder.  These stand in for the un-vendored
// madrona::imp::AssetImporter / ImageImporter the reference calls at
// /root/reference/src/mgr.cpp:294-307 and :318-319.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace mrx {

// Object-space triangle soup of one object (one per OBJ file / raw mesh).
struct TriSoup {
    std::vector<float> pos;  // [T][3 verts][xyz]
    std::vector<float> uv;   // [T][3 verts][uv]
    uint32_t numTris() const { return (uint32_t)(pos.size() / 9); }
};

struct Image {
    std::vector<uint8_t> rgba;  // [h][w][4]
    uint32_t width = 0, height = 0;
};

// Both return false and fill `err` on failure.
bool loadOBJ(const std::string &path, TriSoup &out, std::string &err);
bool decodePNG(const std::string &path, Image &out, std::string &err);
bool decodePNGMem(const uint8_t *data, size_t size, Image &out, std::string &err);
// 8-bit RGBA, non-interlaced, zlib-deflated, filter type 0 on every scanline.
bool encodePNG(const std::string &path, const uint8_t *rgba, uint32_t width, uint32_t height,
               std::string &err);

}  // namespace mrx
