// KTX2 textures: container reader and BC7 block decoder (to RGBA8, at load).
//
// The reference registers a "ktx2" image handler that hands the file to the
// un-vendored madrona-ktx and uploads the BC7 blocks it returns
// (/root/reference/src/mgr.cpp:199-212, :297-298).  This renderer samples
// RGBA8 texels, so a .ktx2 texture is decoded once on the host: the container
// per the Khronos KTX 2.0 specification (header, index, level index), BC7 per
// the Khronos Data Format Specification ("BPTC compressed texture image
// formats").  Supported: vkFormat BC7_UNORM / BC7_SRGB (145 / 146) and
// R8G8B8A8_UNORM / _SRGB (37 / 43), 2D, one layer, one face, base level;
// supercompression none, Zstandard (scheme 2, through the system's libzstd) or
// ZLIB (scheme 3).  Files whose payload is Basis Universal (BasisLZ scheme 1 or
// UASTC, vkFormat 0) need a transcoder and are refused with a message saying so
// (the UASTC / ETC1S block formats are defined by tables -- mode layouts,
// partition maps, BISE sequences -- that are neither in this image nor checkable
// here without a Basis encoder or a reference file; a decoder written from
// memory that could not be validated would be worse than the refusal).  sRGB formats are read as they are (the
// PNG path applies no transfer function either).
#include "assets.hpp"

#include <cstdio>
#include <cstring>
#include <mutex>

#include <dlfcn.h>
#include <zlib.h>

namespace mrx {

namespace {

#include "bc7_tables.inc"

struct BitReader {
    const uint8_t *p;
    uint32_t pos = 0;
    uint32_t get(uint32_t n)                     // n <= 8 bits, LSB first
    {
        uint32_t v = 0;
        for (uint32_t i = 0; i < n; ++i, ++pos)
            v |= (uint32_t)((p[pos >> 3] >> (pos & 7)) & 1u) << i;
        return v;
    }
};

struct Bc7Mode {
    uint8_t subsets, partBits, rotBits, idxSelBits, colorBits, alphaBits, endpointPBits, sharedPBits,
        indexBits, index2Bits;
};
const Bc7Mode kModes[8] = {
    { 3, 4, 0, 0, 4, 0, 1, 0, 3, 0 }, { 2, 6, 0, 0, 6, 0, 0, 1, 3, 0 }, { 3, 6, 0, 0, 5, 0, 0, 0, 2, 0 },
    { 2, 6, 0, 0, 7, 0, 1, 0, 2, 0 }, { 1, 0, 2, 1, 5, 6, 0, 0, 2, 3 }, { 1, 0, 2, 0, 7, 8, 0, 0, 2, 2 },
    { 1, 0, 0, 0, 7, 7, 1, 0, 4, 0 }, { 2, 6, 0, 0, 5, 5, 1, 0, 2, 0 },
};
const uint8_t kWeights2[4] = { 0, 21, 43, 64 };
const uint8_t kWeights3[8] = { 0, 9, 18, 27, 37, 46, 55, 64 };
const uint8_t kWeights4[16] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };

inline uint32_t weightOf(uint32_t bits, uint32_t idx)
{
    return bits == 2 ? kWeights2[idx] : bits == 3 ? kWeights3[idx] : kWeights4[idx];
}
inline uint8_t lerp64(uint32_t a, uint32_t b, uint32_t w)
{
    return (uint8_t)((a * (64u - w) + b * w + 32u) >> 6);
}

}  // namespace

// One 16-byte BC7 block -> 16 RGBA8 pixels, row-major within the 4x4 block.
void decodeBC7Block(const uint8_t block[16], uint8_t out[16][4])
{
    uint32_t mode = 0;
    while (mode < 8 && !((block[0] >> mode) & 1u))
        ++mode;
    if (mode == 8) {                              // reserved encoding: transparent black
        std::memset(out, 0, 64);
        return;
    }
    const Bc7Mode &m = kModes[mode];
    BitReader br { block, mode + 1 };
    const uint32_t part = br.get(m.partBits);
    const uint32_t rot = br.get(m.rotBits);
    const uint32_t idxSel = br.get(m.idxSelBits);
    const uint32_t numEnds = 2u * m.subsets;
    uint32_t ep[6][4];
    for (int ch = 0; ch < 3; ++ch)
        for (uint32_t e = 0; e < numEnds; ++e)
            ep[e][ch] = br.get(m.colorBits);
    for (uint32_t e = 0; e < numEnds; ++e)
        ep[e][3] = m.alphaBits ? br.get(m.alphaBits) : 255u;
    uint32_t cbits = m.colorBits, abits = m.alphaBits;
    if (m.endpointPBits || m.sharedPBits) {
        uint32_t pb[6];
        if (m.endpointPBits) {
            for (uint32_t e = 0; e < numEnds; ++e)
                pb[e] = br.get(1);
        } else {
            for (uint32_t s = 0; s < m.subsets; ++s)
                pb[2 * s] = pb[2 * s + 1] = br.get(1);
        }
        for (uint32_t e = 0; e < numEnds; ++e) {
            for (int ch = 0; ch < 3; ++ch)
                ep[e][ch] = (ep[e][ch] << 1) | pb[e];
            if (m.alphaBits)
                ep[e][3] = (ep[e][3] << 1) | pb[e];
        }
        ++cbits;
        if (abits)
            ++abits;
    }
    for (uint32_t e = 0; e < numEnds; ++e) {      // widen to eight bits by bit replication
        for (int ch = 0; ch < 3; ++ch) {
            const uint32_t v = ep[e][ch] << (8 - cbits);
            ep[e][ch] = v | (v >> cbits);
        }
        if (abits) {
            const uint32_t v = ep[e][3] << (8 - abits);
            ep[e][3] = v | (v >> abits);
        }
    }
    const uint8_t *shape = m.subsets == 2 ? kBc7Part2[part] : m.subsets == 3 ? kBc7Part3[part] : nullptr;
    uint32_t anchors[3] = { 0, 0, 0 };
    if (m.subsets == 2)
        anchors[1] = kBc7Anchor2[part];
    if (m.subsets == 3) {
        anchors[1] = kBc7Anchor3a[part];
        anchors[2] = kBc7Anchor3b[part];
    }
    uint32_t idx[16], idx2[16];
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t s = shape ? shape[i] : 0u;
        idx[i] = br.get(m.indexBits - (i == anchors[s] ? 1u : 0u));
    }
    for (uint32_t i = 0; i < 16; ++i)
        idx2[i] = m.index2Bits ? br.get(m.index2Bits - (i == 0 ? 1u : 0u)) : 0u;
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t s = shape ? shape[i] : 0u;
        const uint32_t *e0 = ep[2 * s], *e1 = ep[2 * s + 1];
        uint32_t cw, aw;
        if (!m.index2Bits) {
            cw = aw = weightOf(m.indexBits, idx[i]);
        } else if (idxSel) {                      // mode 4 with the index sets swapped
            cw = weightOf(m.index2Bits, idx2[i]);
            aw = weightOf(m.indexBits, idx[i]);
        } else {
            cw = weightOf(m.indexBits, idx[i]);
            aw = weightOf(m.index2Bits, idx2[i]);
        }
        uint8_t px[4] = { lerp64(e0[0], e1[0], cw), lerp64(e0[1], e1[1], cw), lerp64(e0[2], e1[2], cw),
                          lerp64(e0[3], e1[3], aw) };
        if (rot) {                                // modes 4 / 5: alpha swapped with a colour channel
            const uint8_t t = px[3];
            px[3] = px[rot - 1];
            px[rot - 1] = t;
        }
        std::memcpy(out[i], px, 4);
    }
}

namespace {

inline uint32_t le32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
inline uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

// size_t ZSTD_decompress(void *dst, size_t dstCapacity, const void *src, size_t compressedSize)
// and unsigned ZSTD_isError(size_t) of libzstd.so.1 (stable since zstd 1.0), bound once.
bool zstdDecompress(uint8_t *dst, size_t need, const uint8_t *src, size_t len, std::string &err)
{
    typedef size_t (*DecompressFn)(void *, size_t, const void *, size_t);
    typedef unsigned (*IsErrorFn)(size_t);
    static DecompressFn decompress = nullptr;
    static IsErrorFn isError = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h)
            h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
        if (h) {
            decompress = (DecompressFn)dlsym(h, "ZSTD_decompress");
            isError = (IsErrorFn)dlsym(h, "ZSTD_isError");
        }
    });
    if (!decompress || !isError) {
        err = "KTX2: Zstandard payload, but libzstd.so.1 is not installed";
        return false;
    }
    const size_t got = decompress(dst, need, src, len);
    if (isError(got) || got != need) {
        err = "KTX2: Zstandard payload does not decompress to the image size";
        return false;
    }
    return true;
}

}  // namespace

bool decodeKTX2Mem(const uint8_t *data, size_t size, Image &out, std::string &err)
{
    static const uint8_t ident[12] = { 0xAB, 'K', 'T', 'X', ' ', '2', '0', 0xBB, 0x0D, 0x0A, 0x1A, 0x0A };
    if (size < 80 + 24 || std::memcmp(data, ident, 12) != 0) {
        err = "not a KTX2 file";
        return false;
    }
    const uint32_t vkFormat = le32(data + 12), width = le32(data + 20), height = le32(data + 24);
    const uint32_t depth = le32(data + 28), layers = le32(data + 32), faces = le32(data + 36);
    const uint32_t levels = le32(data + 40), scheme = le32(data + 44);
    if (width == 0 || height == 0 || width > 16384 || height > 16384 || depth > 1 || layers > 1 || faces != 1) {
        err = "KTX2: only single 2D images are supported";
        return false;
    }
    if (vkFormat == 0 || scheme == 1) {
        err = "KTX2: Basis Universal payload (UASTC / BasisLZ) needs a transcoder; store BC7 or RGBA8";
        return false;
    }
    const bool bc7 = vkFormat == 145 || vkFormat == 146;
    const bool rgba8 = vkFormat == 37 || vkFormat == 43;
    if (!bc7 && !rgba8) {
        err = "KTX2: unsupported vkFormat " + std::to_string(vkFormat) + " (BC7 and R8G8B8A8 are read)";
        return false;
    }
    if (scheme != 0 && scheme != 2 && scheme != 3) {
        err = "KTX2: unsupported supercompression scheme " + std::to_string(scheme) + " (none, Zstandard and ZLIB are read)";
        return false;
    }
    if ((size_t)80 + 24 * (size_t)(levels ? levels : 1) > size) {
        err = "KTX2: truncated level index";
        return false;
    }
    // level 0 is the base (largest) image
    const uint64_t off = le64(data + 80), len = le64(data + 88), ulen = le64(data + 96);
    if (off > size || len > size - off) {
        err = "KTX2: level data out of range";
        return false;
    }
    const uint32_t bw = (width + 3) / 4, bh = (height + 3) / 4;
    const size_t need = bc7 ? (size_t)bw * bh * 16 : (size_t)width * height * 4;
    std::vector<uint8_t> inflated;
    const uint8_t *src = data + off;
    if (scheme == 2) {
        // Zstandard (what `toktx --zcmp` and the Basis tools write for non-Basis payloads): the
        // system's libzstd, looked up at run time -- the image ships the library without its header
        inflated.resize(need);
        if (!zstdDecompress(inflated.data(), need, src, (size_t)len, err))
            return false;
        src = inflated.data();
    } else if (scheme == 3) {
        inflated.resize(need);
        uLongf got = (uLongf)need;
        if (uncompress(inflated.data(), &got, src, (uLong)len) != Z_OK || got != need) {
            err = "KTX2: zlib payload does not inflate to the image size";
            return false;
        }
        src = inflated.data();
    } else if (len < need || (ulen && ulen < need)) {
        err = "KTX2: level data shorter than the image";
        return false;
    }
    out.width = width;
    out.height = height;
    out.rgba.assign((size_t)width * height * 4, 0);
    if (rgba8) {
        std::memcpy(out.rgba.data(), src, need);
        return true;
    }
    uint8_t px[16][4];
    for (uint32_t by = 0; by < bh; ++by)
        for (uint32_t bx = 0; bx < bw; ++bx) {
            decodeBC7Block(src + ((size_t)by * bw + bx) * 16, px);
            for (uint32_t y = 0; y < 4 && by * 4 + y < height; ++y)
                for (uint32_t x = 0; x < 4 && bx * 4 + x < width; ++x)
                    std::memcpy(&out.rgba[(((size_t)by * 4 + y) * width + bx * 4 + x) * 4], px[y * 4 + x], 4);
        }
    return true;
}

bool decodeKTX2(const std::string &path, Image &out, std::string &err)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open '" + path + "'";
        return false;
    }
    std::vector<uint8_t> buf;
    uint8_t chunk[65536];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof chunk, f)) > 0)
        buf.insert(buf.end(), chunk, chunk + n);
    std::fclose(f);
    if (!decodeKTX2Mem(buf.data(), buf.size(), out, err)) {
        err = path + ": " + err;
        return false;
    }
    return true;
}

// Texture files by extension: .ktx2 through the reader above, everything else as PNG.
bool decodeTexture(const std::string &path, Image &out, std::string &err)
{
    const size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? std::string() : path.substr(dot + 1);
    for (char &c : ext)
        c = (char)(c >= 'A' && c <= 'Z' ? c - 'A' + 'a' : c);
    if (ext == "ktx2")
        return decodeKTX2(path, out, err);
    return decodePNG(path, out, err);
}

}  // namespace mrx
