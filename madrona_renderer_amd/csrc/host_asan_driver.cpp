// Sanitizer driver for the host-side code that reads untrusted input: the OBJ / MTL / PNG /
// KTX2 / BC7 readers (assets.cpp, ktx2.cpp) and the BLAS builder (bvh.cpp).  Built with
// -fsanitize=address,undefined by `python -m madrona_renderer_amd.build --asan` (g++, host
// only) and run by tests/test_sanitizers.py in the CPU suite -- never on the GPU box (GPU
// AddressSanitizer is not available on the pool).
//
//   host_asan_driver parse FILE...                 read each file by its extension
//   host_asan_driver fuzz FILE ITERATIONS SEED     the file with random bytes overwritten,
//                                                  truncated or extended, ITERATIONS times
//   host_asan_driver blas NUM_TRIS SEED KIND       build + walk the BLAS of a random soup
//                                                  (KIND: 0 cloud, 1 sliver chain, 2 coincident)
// Parse failures are expected and fine; what must not happen is a sanitizer report.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <unistd.h>

#include "assets.hpp"
#include "bvh.hpp"

namespace {

std::string extOf(const std::string &path)
{
    const size_t dot = path.find_last_of('.');
    std::string e = dot == std::string::npos ? std::string() : path.substr(dot + 1);
    for (char &c : e)
        c = (char)(c >= 'A' && c <= 'Z' ? c - 'A' + 'a' : c);
    return e;
}

// returns a small digest of what was read, so nothing can be optimised away
uint64_t readOne(const std::string &path, bool &ok)
{
    std::string err;
    const std::string ext = extOf(path);
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    ok = false;
    if (ext == "obj") {
        mrx::TriSoup soup;
        ok = mrx::loadOBJ(path, soup, err);
        if (ok) {
            mix(soup.numTris());
            for (float f : soup.pos) { uint32_t u; std::memcpy(&u, &f, 4); mix(u); }
            for (uint32_t s : soup.objStart) mix(s);
            for (int32_t m : soup.triMtl) mix((uint32_t)m);
            std::vector<mrx::MtlMaterial> lib;
            for (const std::string &ml : soup.mtlLibs) {
                std::string merr;
                (void)mrx::loadMTL(ml, lib, merr);
            }
            mix(lib.size());
            if (soup.numTris() > 0 && soup.numTris() < 200000) {
                std::vector<mrx::ObjTri> tris(soup.numTris());
                for (uint32_t t = 0; t < soup.numTris(); ++t)
                    std::memcpy(tris[t].p, &soup.pos[9 * (size_t)t], 36);
                mrx::BlasSet b;
                mrx::buildBlas(tris.data(), { 0 }, { (int32_t)soup.numTris() }, b);
                mix(b.nodes.size());
                mix(b.maxDepth);
            }
        }
    } else if (ext == "mtl") {
        std::vector<mrx::MtlMaterial> lib;
        ok = mrx::loadMTL(path, lib, err);
        for (const auto &m : lib) mix(m.name.size() + m.mapKd.size());
    } else {
        mrx::Image img;
        ok = mrx::decodeTexture(path, img, err);
        if (ok) {
            mix(img.width); mix(img.height);
            for (uint8_t b : img.rgba) mix(b);
        }
    }
    return h;
}

uint64_t rng(uint64_t &s)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int fuzz(const std::string &path, int iters, uint64_t seed)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path.c_str());
        return 2;
    }
    std::vector<uint8_t> orig;
    uint8_t chunk[65536];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof chunk, f)) > 0)
        orig.insert(orig.end(), chunk, chunk + n);
    std::fclose(f);
    char tmpl[] = "/tmp/mrx_fuzz_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0)
        return 2;
    close(fd);
    const std::string tmp = std::string(tmpl) + "." + extOf(path);
    int accepted = 0;
    uint64_t digest = 0;
    for (int it = 0; it < iters; ++it) {
        std::vector<uint8_t> d = orig;
        const int kind = (int)(rng(seed) % 4);
        if (kind == 0 && !d.empty()) {                // a few random bytes
            const int k = 1 + (int)(rng(seed) % 8);
            for (int i = 0; i < k; ++i)
                d[rng(seed) % d.size()] = (uint8_t)rng(seed);
        } else if (kind == 1 && !d.empty()) {         // truncate
            d.resize(rng(seed) % d.size());
        } else if (kind == 2 && d.size() > 8) {       // a 32-bit field set to an extreme value
            const size_t at = rng(seed) % (d.size() - 4);
            const uint32_t v[4] = { 0u, 0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u };
            std::memcpy(&d[at], &v[rng(seed) % 4], 4);
        } else {                                      // garbage appended / a run of one byte
            const size_t at = d.empty() ? 0 : rng(seed) % d.size();
            d.insert(d.begin() + (long)at, 64 + rng(seed) % 512, (uint8_t)rng(seed));
        }
        FILE *o = std::fopen(tmp.c_str(), "wb");
        if (!o)
            return 2;
        if (!d.empty())
            std::fwrite(d.data(), 1, d.size(), o);
        std::fclose(o);
        bool ok = false;
        digest ^= readOne(tmp, ok);
        accepted += ok;
    }
    std::remove(tmp.c_str());
    std::remove(tmpl);
    std::printf("fuzz %s: %d iterations, %d still parsed, digest %016llx\n", path.c_str(), iters, accepted,
                (unsigned long long)digest);
    return 0;
}

int blas(uint32_t numTris, uint64_t seed, int kind)
{
    std::vector<mrx::ObjTri> tris(numTris);
    auto uf = [&]() { return (float)((rng(seed) >> 40) * (1.0 / 16777216.0)); };
    for (uint32_t t = 0; t < numTris; ++t) {
        float c[3] = { uf() * 20.f - 10.f, uf() * 20.f - 10.f, uf() * 4.f };
        if (kind == 1) {            // a chain of slivers along one axis with exponentially growing gaps:
            c[0] = std::ldexp(1.0f, (int)(t % 60)) * 1e-6f * (float)(t + 1);   // lopsided SAH splits
            c[1] = c[2] = 0.0f;
        } else if (kind == 2) {     // every centroid coincides
            c[0] = c[1] = c[2] = 1.0f;
        }
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a)
                tris[t].p[3 * v + a] = c[a] + (kind == 2 ? 0.0f : (uf() - 0.5f) * 0.3f);
    }
    mrx::BlasSet b;
    mrx::buildBlas(tris.data(), { 0 }, { (int32_t)numTris }, b);
    // walk the tree: every leaf entry in range, every triangle exactly once
    std::vector<uint32_t> seen(numTris, 0);
    uint64_t leaves = 0;
    if (b.objects[0].root >= 0) {
        std::vector<uint32_t> todo = { (uint32_t)b.objects[0].root };
        while (!todo.empty()) {
            const uint32_t ni = todo.back();
            todo.pop_back();
            if (ni >= b.nodes.size())
                return 3;
            for (uint32_t c = 0; c < mrx::kBvhWidth; ++c) {
                const uint32_t ref = b.nodes[ni].child[c];
                if (ref == mrx::kBvhEmpty)
                    continue;
                if (ref & mrx::kBvhLeafBit) {
                    const uint32_t cnt = ((ref >> mrx::kBvhLeafStartBits) & 15u) + 1u;
                    const uint32_t start = ref & ((1u << mrx::kBvhLeafStartBits) - 1u);
                    if (start + cnt > b.leafTris.size())
                        return 3;
                    for (uint32_t i = 0; i < cnt; ++i) {
                        const uint32_t tt = b.leafTris[start + i];
                        if (tt >= numTris || seen[tt]++)
                            return 3;
                    }
                    ++leaves;
                } else {
                    todo.push_back(ref);
                }
            }
        }
        for (uint32_t t = 0; t < numTris; ++t)
            if (seen[t] != 1)
                return 3;
    }
    std::printf("blas %u triangles kind %d: %zu nodes, depth %u, %llu leaves, stack bound %s\n", numTris, kind,
                b.nodes.size(), b.maxDepth, (unsigned long long)leaves,
                1 + 7 * b.maxDepth <= mrx::kBvhStackCap ? "holds" : "EXCEEDED (the scene would be refused)");
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc >= 3 && !std::strcmp(argv[1], "parse")) {
        for (int i = 2; i < argc; ++i) {
            bool ok = false;
            const uint64_t h = readOne(argv[i], ok);
            std::printf("%s: %s %016llx\n", argv[i], ok ? "ok" : "refused", (unsigned long long)h);
        }
        return 0;
    }
    if (argc == 5 && !std::strcmp(argv[1], "fuzz"))
        return fuzz(argv[2], std::atoi(argv[3]), (uint64_t)std::atoll(argv[4]));
    if (argc == 5 && !std::strcmp(argv[1], "blas"))
        return blas((uint32_t)std::atoi(argv[2]), (uint64_t)std::atoll(argv[3]), std::atoi(argv[4]));
    std::fprintf(stderr, "usage: host_asan_driver parse FILE... | fuzz FILE ITERATIONS SEED | blas NUM_TRIS SEED KIND\n");
    return 2;
}
