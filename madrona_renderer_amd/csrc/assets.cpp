#include "assets.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <zlib.h>

namespace mrx {

namespace {

bool readFile(const std::string &path, std::vector<uint8_t> &buf, std::string &err)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open '" + path + "'";
        return false;
    }
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    if (got != buf.size()) {
        err = "short read on '" + path + "'";
        return false;
    }
    return true;
}

// Text numbers go through strtod and one rounding to float, the same path a
// Python float() -> numpy.float32 conversion takes.
inline float parseFloat(const char *&p)
{
    char *end = nullptr;
    double d = std::strtod(p, &end);
    p = end;
    return (float)d;
}

inline void skipSpace(const char *&p)
{
    while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
}

struct Corner {
    int v;
    int vt;  // -1: none
};

std::string dirOf(const std::string &path)
{
    const size_t slash = path.find_last_of('/');
    return slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
}

// Text files are parsed line by line in place: bytes that would end the C string early (an
// embedded NUL -- found by the sanitizer run over damaged files: strchr() then returned no
// line end) become blanks, a final newline is added, and every line is terminated for the
// time it is parsed, so that strtod / strtol can never run on into the next line.
void prepareText(std::vector<uint8_t> &file)
{
    for (uint8_t &c : file)
        if (c == 0)
            c = ' ';
    file.push_back('\n');
    file.push_back(0);
}

// rest of the line, trimmed
std::string restOfLine(const char *p, const char *eol)
{
    while (p < eol && (*p == ' ' || *p == '\t')) ++p;
    const char *e = eol;
    while (e > p && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r' || e[-1] == '\n')) --e;
    return std::string(p, e);
}

}  // namespace

bool loadOBJ(const std::string &path, TriSoup &out, std::string &err)
{
    std::vector<uint8_t> file;
    if (!readFile(path, file, err))
        return false;
    prepareText(file);

    std::vector<float> vs, vts;
    std::vector<Corner> corners;
    out.pos.clear();
    out.uv.clear();
    out.triMtl.clear();
    out.mtlNames.clear();
    out.mtlLibs.clear();
    out.objStart.assign(1, 0u);
    int32_t curMtl = -1;

    const char *p = (const char *)file.data();
    const char *const fileEnd = p + file.size() - 1;      // the terminating NUL
    int lineNo = 0;
    while (p < fileEnd) {
        ++lineNo;
        char *eol = const_cast<char *>((const char *)std::memchr(p, '\n', (size_t)(fileEnd - p)));
        if (!eol)
            break;
        *eol = 0;                                         // the line is a C string of its own
        skipSpace(p);
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            p += 1;
            for (int i = 0; i < 3; ++i) {
                skipSpace(p);
                vs.push_back(parseFloat(p));
            }
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            p += 2;
            for (int i = 0; i < 2; ++i) {
                skipSpace(p);
                vts.push_back(parseFloat(p));
            }
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 1;
            corners.clear();
            const int nv = (int)(vs.size() / 3), nt = (int)(vts.size() / 2);
            while (true) {
                skipSpace(p);
                if (p >= eol || *p == 0)
                    break;
                char *end = nullptr;
                long vi = std::strtol(p, &end, 10);
                if (end == p) {
                    err = path + ": bad face at line " + std::to_string(lineNo);
                    return false;
                }
                p = end;
                Corner c;
                c.v = vi > 0 ? (int)vi - 1 : nv + (int)vi;
                c.vt = -1;
                if (*p == '/') {
                    ++p;
                    if (*p != '/' ) {
                        long ti = std::strtol(p, &end, 10);
                        if (end != p) {
                            c.vt = ti > 0 ? (int)ti - 1 : nt + (int)ti;
                            p = end;
                        }
                    }
                    if (*p == '/') {  // normal index: parsed and ignored
                        ++p;
                        std::strtol(p, &end, 10);
                        p = end;
                    }
                }
                if (c.v < 0 || c.v >= nv || c.vt >= nt) {
                    err = path + ": index out of range at line " +
                          std::to_string(lineNo);
                    return false;
                }
                corners.push_back(c);
            }
            for (size_t j = 1; j + 1 < corners.size(); ++j) {
                const Corner tri[3] = { corners[0], corners[j], corners[j + 1] };
                for (const Corner &c : tri) {
                    out.pos.push_back(vs[3 * c.v + 0]);
                    out.pos.push_back(vs[3 * c.v + 1]);
                    out.pos.push_back(vs[3 * c.v + 2]);
                    out.uv.push_back(c.vt >= 0 ? vts[2 * c.vt + 0] : 0.0f);
                    out.uv.push_back(c.vt >= 0 ? vts[2 * c.vt + 1] : 0.0f);
                }
                out.triMtl.push_back(curMtl);
            }
        } else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t' || p[1] == '\r' || p[1] == 0)) {
            // a new object opens here unless the current one has no face yet
            if (out.numTris() > out.objStart.back())
                out.objStart.push_back(out.numTris());
        } else if (!std::strncmp(p, "usemtl", 6) && (p[6] == ' ' || p[6] == '\t')) {
            const std::string name = restOfLine(p + 6, eol);
            curMtl = -1;
            for (size_t i = 0; i < out.mtlNames.size(); ++i)
                if (out.mtlNames[i] == name)
                    curMtl = (int32_t)i;
            if (curMtl < 0) {
                curMtl = (int32_t)out.mtlNames.size();
                out.mtlNames.push_back(name);
            }
        } else if (!std::strncmp(p, "mtllib", 6) && (p[6] == ' ' || p[6] == '\t')) {
            const std::string name = restOfLine(p + 6, eol);
            if (!name.empty())
                out.mtlLibs.push_back(name[0] == '/' ? name : dirOf(path) + name);
        }
        p = eol + 1;
    }
    return true;
}

bool loadMTL(const std::string &path, std::vector<MtlMaterial> &out, std::string &err)
{
    std::vector<uint8_t> file;
    if (!readFile(path, file, err))
        return false;
    prepareText(file);
    const char *p = (const char *)file.data();
    const char *const fileEnd = p + file.size() - 1;
    while (p < fileEnd) {
        char *eol = const_cast<char *>((const char *)std::memchr(p, '\n', (size_t)(fileEnd - p)));
        if (!eol)
            break;
        *eol = 0;
        skipSpace(p);
        if (!std::strncmp(p, "newmtl", 6) && (p[6] == ' ' || p[6] == '\t')) {
            out.emplace_back();
            out.back().name = restOfLine(p + 6, eol);
        } else if (!out.empty() && p[0] == 'K' && p[1] == 'd' && (p[2] == ' ' || p[2] == '\t')) {
            const char *q = p + 2;
            for (int i = 0; i < 3; ++i) {
                skipSpace(q);
                out.back().kd[i] = parseFloat(q);
            }
        } else if (!out.empty() && !std::strncmp(p, "map_Kd", 6) && (p[6] == ' ' || p[6] == '\t')) {
            // options (-s, -o ...) are not supported: the last token is the file
            std::string rest = restOfLine(p + 6, eol);
            const size_t sp = rest.find_last_of(" \t");
            if (sp != std::string::npos)
                rest = rest.substr(sp + 1);
            if (!rest.empty())
                out.back().mapKd = rest[0] == '/' ? rest : dirOf(path) + rest;
        }
        p = eol + 1;
    }
    return true;
}

// ---------------------------------------------------------------------------
// PNG: colour types 0/2/3/4/6, bit depths 1-16, non-interlaced.
// ---------------------------------------------------------------------------
namespace {

inline uint32_t be32(const uint8_t *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) |
           ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

inline int paeth(int a, int b, int c)
{
    int p = a + b - c;
    int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

}  // namespace

bool decodePNGMem(const uint8_t *data, size_t size, Image &out, std::string &err)
{
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (size < 8 || std::memcmp(data, sig, 8) != 0) {
        err = "not a PNG";
        return false;
    }
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t off = 8;
    bool sawEnd = false;
    while (off + 12 <= size && !sawEnd) {
        uint32_t len = be32(data + off);
        const uint8_t *type = data + off + 4;
        const uint8_t *body = data + off + 8;
        if (off + 12 + (size_t)len > size) {
            err = "truncated PNG chunk";
            return false;
        }
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            sawEnd = true;
        }
        off += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || ctype < 0) {
        err = "PNG without IHDR";
        return false;
    }
    if (interlace != 0) {
        err = "interlaced PNG not supported";
        return false;
    }
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: err = "bad PNG colour type"; return false;
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) ||
        (ctype == 3 && depth == 16) ||
        ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8)) {
        err = "bad PNG bit depth";
        return false;
    }
    const size_t bitsPerPixel = (size_t)channels * depth;
    const size_t stride = ((size_t)w * bitsPerPixel + 7) / 8;
    const size_t bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;

    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf rawLen = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || rawLen != raw.size()) {
        err = "PNG inflate failed";
        return false;
    }

    // undo the scanline filters in place
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t *line = raw.data() + (size_t)y * (stride + 1);
        const uint8_t ft = line[0];
        uint8_t *cur = line + 1;
        const uint8_t *up = y ? line - stride : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= bpp ? cur[i - bpp] : 0;
            int b = up[i];
            int c = i >= bpp ? up[i - bpp] : 0;
            int x = cur[i];
            switch (ft) {
            case 0: break;
            case 1: x += a; break;
            case 2: x += b; break;
            case 3: x += (a + b) >> 1; break;
            case 4: x += paeth(a, b, c); break;
            default: err = "bad PNG filter"; return false;
            }
            cur[i] = (uint8_t)x;
        }
    }

    out.width = w;
    out.height = h;
    out.rgba.assign((size_t)w * h * 4, 255);
    const int maxv = (1 << (depth > 8 ? 8 : depth)) - 1;
    auto sample = [&](const uint8_t *row, size_t idx) -> int {
        // idx-th sample of the row, reduced to 8 bits for depth 16
        if (depth == 8) return row[idx];
        if (depth == 16) return row[2 * idx];
        const size_t bit = idx * depth;
        const int shift = 8 - depth - (int)(bit & 7);
        return (row[bit >> 3] >> shift) & maxv;
    };
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *row = raw.data() + (size_t)y * (stride + 1) + 1;
        uint8_t *dst = out.rgba.data() + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; ++x, dst += 4) {
            switch (ctype) {
            case 0: {
                int g = sample(row, x);
                int g8 = depth < 8 ? g * 255 / maxv : g;
                dst[0] = dst[1] = dst[2] = (uint8_t)g8;
                if (trns.size() >= 2) {
                    int key = depth == 16 ? trns[0] : trns[1];
                    if (depth == 16 ? (row[2 * x] == trns[0] && row[2 * x + 1] == trns[1])
                                    : g == key)
                        dst[3] = 0;
                }
                break;
            }
            case 2: {
                dst[0] = (uint8_t)sample(row, 3 * x + 0);
                dst[1] = (uint8_t)sample(row, 3 * x + 1);
                dst[2] = (uint8_t)sample(row, 3 * x + 2);
                if (trns.size() >= 6) {
                    bool hit;
                    if (depth == 16)
                        hit = !std::memcmp(row + 6 * x, trns.data(), 6);
                    else
                        hit = dst[0] == trns[1] && dst[1] == trns[3] && dst[2] == trns[5];
                    if (hit) dst[3] = 0;
                }
                break;
            }
            case 3: {
                size_t i = (size_t)sample(row, x);
                if (3 * i + 2 < plte.size()) {
                    dst[0] = plte[3 * i];
                    dst[1] = plte[3 * i + 1];
                    dst[2] = plte[3 * i + 2];
                } else {
                    dst[0] = dst[1] = dst[2] = 0;
                }
                if (i < trns.size()) dst[3] = trns[i];
                break;
            }
            case 4: {
                int g = sample(row, 2 * x);
                dst[0] = dst[1] = dst[2] = (uint8_t)g;
                dst[3] = (uint8_t)sample(row, 2 * x + 1);
                break;
            }
            case 6: {
                for (int c = 0; c < 4; ++c)
                    dst[c] = (uint8_t)sample(row, 4 * x + c);
                break;
            }
            }
        }
    }
    return true;
}

namespace {
void putChunk(std::vector<uint8_t> &out, const char type[4], const uint8_t *body, uint32_t len)
{
    const uint8_t hdr[4] = { (uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8),
                             (uint8_t)len };
    out.insert(out.end(), hdr, hdr + 4);
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (len)
        out.insert(out.end(), body, body + len);
    const uint32_t crc = (uint32_t)crc32(0L, out.data() + at, (uInt)(len + 4));
    const uint8_t tail[4] = { (uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8),
                              (uint8_t)crc };
    out.insert(out.end(), tail, tail + 4);
}
}  // namespace

bool encodePNG(const std::string &path, const uint8_t *rgba, uint32_t width, uint32_t height,
               std::string &err)
{
    std::vector<uint8_t> raw((size_t)height * ((size_t)width * 4 + 1));
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t *line = raw.data() + (size_t)y * ((size_t)width * 4 + 1);
        line[0] = 0;
        std::memcpy(line + 1, rgba + (size_t)y * width * 4, (size_t)width * 4);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
        err = "PNG deflate failed";
        return false;
    }
    std::vector<uint8_t> out = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    const uint8_t ihdr[13] = { (uint8_t)(width >> 24), (uint8_t)(width >> 16), (uint8_t)(width >> 8),
                               (uint8_t)width, (uint8_t)(height >> 24), (uint8_t)(height >> 16),
                               (uint8_t)(height >> 8), (uint8_t)height, 8, 6, 0, 0, 0 };
    putChunk(out, "IHDR", ihdr, 13);
    putChunk(out, "IDAT", z.data(), (uint32_t)zlen);
    putChunk(out, "IEND", nullptr, 0);
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) {
        err = "cannot write '" + path + "'";
        return false;
    }
    const size_t put = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    if (put != out.size()) {
        err = "short write on '" + path + "'";
        return false;
    }
    return true;
}

bool decodePNG(const std::string &path, Image &out, std::string &err)
{
    std::vector<uint8_t> file;
    if (!readFile(path, file, err))
        return false;
    if (!decodePNGMem(file.data(), file.size(), out, err)) {
        err = path + ": " + err;
        return false;
    }
    return true;
}

}  // namespace mrx
