// renderer_headless -- the reference's command-line front end
// (/root/reference/src/headless.cpp:31-79, src/args.cpp:52-98, src/dump.cpp:45-119)
// over the MI355X Manager:
//
//   renderer_headless NUM_WORLDS NUM_STEPS rt|rast BATCH_WIDTH BATCH_HEIGHT
//                     [--dump-last-frame file_name_without_extension]
//                     [--scene synthetic|demo] [--depth] [--gpus N]
//
// --gpus N (no counterpart upstream: the reference has a single gpuID,
// mgr.hpp:50) renders the worlds on N devices of the node through ONE Manager
// (Config::deviceIDs): contiguous world ranges whose sizes differ by at most
// one, one shard per device, step() launches on all of them, no exchange
// between the devices.  A line per device and the two reference lines for the
// whole node are printed.
//
// It steps the renderer NUM_STEPS times, prints the reference's two lines
// (`FPS`, `Average total step time`) and optionally writes the last frame of
// every world as one tiled PNG.  The reference constructs its Manager without
// a scene (headless.cpp:48-55 passes no rcfg); here the scene is either the
// synthetic cube+plane worlds of the benchmark or the reference's demo scene
// (viewer.cpp:74-164 / scripts/test.py:11-130).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/madrona_mi355/manager.hpp"
#include "../../include/mrx.h"
#include "assets.hpp"

#ifndef MRX_DATA_DIR
#define MRX_DATA_DIR "data"
#endif

using namespace madRender;
using madrona::math::Quat;
using madrona::math::Vector3;

namespace {

enum class Mode { Rasterizer, Raycaster };

struct Args {
    uint32_t numWorlds = 0, numSteps = 0, width = 64, height = 64;
    Mode mode = Mode::Rasterizer;
    bool dump = false, dumpDepth = false, demo = false;
    uint32_t gpus = 1;
    std::string outName;
};

[[noreturn]] void usage(const char *argv0)
{
    std::fprintf(stderr,
                 "%s [NUM_WORLDS] [NUM_STEPS] [rt|rast] [BATCH_WIDTH] [BATCH_HEIGHT] "
                 "[--dump-last-frame file_name_without_extension] [--scene synthetic|demo] [--depth] [--gpus N]\n",
                 argv0);
    std::exit(EXIT_FAILURE);
}

Args parse(int argc, char **argv)
{
    if (argc < 6)
        usage(argv[0]);
    Args a;
    a.numWorlds = (uint32_t)std::atoi(argv[1]);
    a.numSteps = (uint32_t)std::atoi(argv[2]);
    if (!std::strcmp(argv[3], "rt")) a.mode = Mode::Raycaster;
    else if (!std::strcmp(argv[3], "rast")) a.mode = Mode::Rasterizer;
    else usage(argv[0]);
    a.width = (uint32_t)std::atoi(argv[4]);
    a.height = (uint32_t)std::atoi(argv[5]);
    for (int i = 6; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--dump-last-frame") && i + 1 < argc) {
            a.dump = true;
            a.outName = argv[++i];
        } else if (!std::strcmp(argv[i], "--scene") && i + 1 < argc) {
            a.demo = !std::strcmp(argv[++i], "demo");
        } else if (!std::strcmp(argv[i], "--depth")) {
            a.dumpDepth = true;
        } else if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) {
            a.gpus = (uint32_t)std::atoi(argv[++i]);
        } else {
            usage(argv[0]);
        }
    }
    if (a.numWorlds == 0 || a.width == 0 || a.height == 0 || a.gpus == 0 || a.gpus > a.numWorlds)
        usage(argv[0]);
    return a;
}

// u(k) = (splitmix64(seed ^ k) >> 40) * 2^-24
double uniform(uint64_t k)
{
    uint64_t z = (0x4D52584Dull ^ k) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 40) * 0x1p-24;
}

// camera at `eye` looking at `tgt`: local +Y forward, +X right, +Z up, no roll
Quat lookAt(const float eye[3], const double tgt[3])
{
    double f[3] = { tgt[0] - eye[0], tgt[1] - eye[1], tgt[2] - eye[2] };
    double n = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (double &x : f) x /= n;
    double r[3] = { f[1] * 1.0 - f[2] * 0.0, f[2] * 0.0 - f[0] * 1.0, f[0] * 0.0 - f[1] * 0.0 };
    n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (double &x : r) x /= n;
    double u[3] = { r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0] };
    // rotation with columns (right, fwd, up) -> quaternion
    const double m[3][3] = { { r[0], f[0], u[0] }, { r[1], f[1], u[1] }, { r[2], f[2], u[2] } };
    const double tr = m[0][0] + m[1][1] + m[2][2];
    double q[4];
    if (tr > 0) {
        const double s = std::sqrt(tr + 1.0) * 2;
        q[0] = 0.25 * s; q[1] = (m[2][1] - m[1][2]) / s;
        q[2] = (m[0][2] - m[2][0]) / s; q[3] = (m[1][0] - m[0][1]) / s;
    } else if (m[0][0] > m[1][1] && m[0][0] > m[2][2]) {
        const double s = std::sqrt(1.0 + m[0][0] - m[1][1] - m[2][2]) * 2;
        q[0] = (m[2][1] - m[1][2]) / s; q[1] = 0.25 * s;
        q[2] = (m[0][1] + m[1][0]) / s; q[3] = (m[0][2] + m[2][0]) / s;
    } else if (m[1][1] > m[2][2]) {
        const double s = std::sqrt(1.0 + m[1][1] - m[0][0] - m[2][2]) * 2;
        q[0] = (m[0][2] - m[2][0]) / s; q[1] = (m[0][1] + m[1][0]) / s;
        q[2] = 0.25 * s; q[3] = (m[1][2] + m[2][1]) / s;
    } else {
        const double s = std::sqrt(1.0 + m[2][2] - m[0][0] - m[1][1]) * 2;
        q[0] = (m[1][0] - m[0][1]) / s; q[1] = (m[0][2] + m[2][0]) / s;
        q[2] = (m[1][2] + m[2][1]) / s; q[3] = 0.25 * s;
    }
    return Quat { (float)q[0], (float)q[1], (float)q[2], (float)q[3] };
}

struct Scene {
    std::vector<std::string> paths;
    std::vector<const char *> pathPtrs;
    std::vector<int32_t> matAssign;
    std::vector<AdditionalMaterial> mats;
    std::vector<std::string> texPaths;
    std::vector<const char *> texPtrs;
    std::vector<ImportedInstance> instances;
    std::vector<ImportedCamera> cameras;
    std::vector<Sim::WorldInit> worlds;
    std::vector<Vector3> verts;
    std::vector<madrona::math::Vector2> uvs;
    std::vector<uint32_t> indices, vertOff, idxOff;
    std::vector<int32_t> meshMats;
};

// worlds [first, first + n) of the synthetic job: world ids are global, so a
// shard holds exactly the rows it would own of the whole job's scene
void buildSynthetic(Scene &s, uint32_t first, uint32_t n, const std::string &dataDir)
{
    s.paths = { dataDir + "/cube.obj", dataDir + "/plane.obj" };
    s.matAssign = { 0, 0 };
    s.mats = { { { 0.588f, 0.588f, 0.588f, 1.0f }, -1, 0.8f, 0.2f },
               { { 1.0f, 1.0f, 1.0f, 1.0f }, 0, 0.8f, 0.2f } };
    s.texPaths = { dataDir + "/cube.png" };
    const double pi = 3.14159265358979323846;
    for (uint32_t w = 0; w < n; ++w) {
        double u[12];
        for (int j = 0; j < 12; ++j)
            u[j] = uniform((uint64_t)(first + w) * 16 + j);
        const double sc = 1.0 + 2.0 * u[2], th = 2.0 * pi * u[3];
        s.instances.push_back({ { 0.f, 0.f, 0.f }, { 1.f, 0.f, 0.f, 0.f }, { 1.f, 1.f, 1.f }, 1 });
        s.instances.push_back({ { (float)(-4 + 8 * u[0]), (float)(-4 + 8 * u[1]), (float)(0.5 * sc) },
                                { (float)std::cos(th / 2), 0.f, 0.f, (float)std::sin(th / 2) },
                                { (float)sc, (float)sc, (float)sc }, 0 });
        const double r = 10.0 + 6.0 * u[7], hgt = 3.0 + 5.0 * u[8], az = 2.0 * pi * u[9];
        const float eye[3] = { (float)(r * std::cos(az)), (float)(r * std::sin(az)), (float)hgt };
        const double tgt[3] = { 0.0, 0.0, 1.0 };
        s.cameras.push_back({ { eye[0], eye[1], eye[2] }, lookAt(eye, tgt) });
        s.worlds.push_back({ 2, w * 2, 1, w });
    }
}

void buildDemo(Scene &s, uint32_t n, const std::string &dataDir)
{
    s.paths = { dataDir + "/cube.obj" };
    s.matAssign = { 0 };
    s.mats = { { { 1.f, 1.f, 1.f, 1.f }, 0, 0.8f, 0.2f } };
    s.texPaths = { dataDir + "/cube.png" };
    s.verts = { { 0.f, 0.f, 0.f }, { 5.f, 0.f, 10.f }, { 10.f, 0.f, 0.f } };
    s.uvs = { { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f } };
    s.indices = { 0, 1, 2 };
    s.vertOff = { 0 };
    s.idxOff = { 0 };
    s.meshMats = { -1 };
    s.instances = { { { 0.f, 0.f, 15.f }, { 0.707107f, 0.707107f, 0.f, 0.f }, { 3.f, 3.f, 3.f }, 0 },
                    { { 0.f, 0.f, 15.f }, { 0.707107f, 0.707107f, 0.f, 0.f }, { 10.f, 10.f, 10.f }, 1 } };
    s.cameras = { { { -22.343935f, -21.845375f, 27.061676f },
                    { 0.913407f, -0.112268f, 0.047731f, -0.388336f } } };
    for (uint32_t w = 0; w < n; ++w)
        s.worlds.push_back({ 2, 0, 1, 0 });
}

// Tiled dump, as /root/reference/src/dump.cpp:45-119: ceil(sqrt(N)) rows of
// images; depth as grey 255 * min(d / 255, 1).  Raytracer storage is [x][y]
// and is transposed back (dump.cpp:9-21); rasterizer storage is row-major.
bool dumpTiled(const std::string &name, mrx_renderer *shard, uint32_t numImages, uint32_t resX,
               uint32_t resY, bool depth, bool transpose)
{
    const size_t bytesPerImage = (size_t)4 * resX * resY;
    std::vector<uint8_t> host(bytesPerImage * numImages);
    if (mrx_copy_to_host(shard, depth ? MRX_BUF_DEPTH : MRX_BUF_RGB,
                         host.data(), host.size()) != MRX_OK) {
        std::fprintf(stderr, "%s\n", mrx_last_error());
        return false;
    }
    const uint32_t tilesY = (uint32_t)std::ceil(std::sqrt((double)numImages));
    const uint32_t tilesX = (uint32_t)std::ceil((double)numImages / tilesY);
    const uint32_t outW = tilesX * resX, outH = tilesY * resY;
    std::vector<uint8_t> img((size_t)outW * outH * 4, 0);
    for (uint32_t i = 0; i < numImages; ++i) {
        const uint32_t tx = i % tilesX, ty = i / tilesX;
        const uint8_t *src = host.data() + bytesPerImage * i;
        for (uint32_t y = 0; y < resY; ++y)
            for (uint32_t x = 0; x < resX; ++x) {
                const size_t si = transpose ? ((size_t)x * resY + y) : ((size_t)y * resX + x);
                uint8_t *dst = &img[(((size_t)ty * resY + y) * outW + tx * resX + x) * 4];
                if (depth) {
                    float d;
                    std::memcpy(&d, src + 4 * si, 4);
                    const uint8_t g = (uint8_t)(255.0f * std::fmin(d / 255.0f, 1.0f));
                    dst[0] = dst[1] = dst[2] = g;
                    dst[3] = 255;
                } else {
                    std::memcpy(dst, src + 4 * si, 4);
                }
            }
    }
    std::string err;
    if (!mrx::encodePNG(name + ".png", img.data(), outW, outH, err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return false;
    }
    return true;
}

}  // namespace

int main(int argc, char **argv)
{
    const Args args = parse(argc, argv);
    const char *dd = std::getenv("MADRONA_MI355_DATA");
    const std::string dataDir = dd ? dd : MRX_DATA_DIR;
    // MRX_HEADLESS_REHEARSAL=1: every shard on device 0 -- walks the N-device
    // control flow on a one-GPU box (the numbers then mean nothing)
    const char *reh = std::getenv("MRX_HEADLESS_REHEARSAL");
    const bool rehearsal = reh && reh[0] == '1';
    if (!rehearsal && (int)args.gpus > mrx_device_count()) {
        std::fprintf(stderr, "--gpus %u but %d HIP device(s) visible\n", args.gpus, mrx_device_count());
        return EXIT_FAILURE;
    }

    // the whole job's scene; ONE Manager spans the devices (Config::deviceIDs): it splits the
    // worlds into contiguous ranges, one shard per device, and step() launches on all of them
    Scene s;
    if (args.demo) buildDemo(s, args.numWorlds, dataDir);
    else buildSynthetic(s, 0, args.numWorlds, dataDir);
    for (auto &p : s.paths) s.pathPtrs.push_back(p.c_str());
    for (auto &p : s.texPaths) s.texPtrs.push_back(p.c_str());
    std::vector<int> devices(args.gpus);
    for (uint32_t g = 0; g < args.gpus; ++g)
        devices[g] = rehearsal ? 0 : (int)g;

    Manager::Config cfg {};
    cfg.gpuID = devices[0];
    cfg.numWorlds = args.numWorlds;
    cfg.renderMode = args.mode == Mode::Raycaster ? Manager::RenderMode::Raytracer
                                                  : Manager::RenderMode::Rasterizer;
    cfg.batchRenderViewWidth = args.width;
    cfg.batchRenderViewHeight = args.height;
    cfg.headlessMode = true;
    auto &rc = cfg.rcfg;
    rc.geoCfg = { s.verts.data(), s.uvs.data(), s.indices.data(), s.vertOff.data(), s.idxOff.data(),
                  s.meshMats.data(), (uint32_t)s.verts.size(), (uint32_t)s.indices.size(),
                  (uint32_t)s.vertOff.size() };
    rc.assetPaths = s.pathPtrs.data();
    rc.numAssetPaths = (uint32_t)s.pathPtrs.size();
    rc.matAssignments = s.matAssign.data();
    rc.numMatAssignments = (uint32_t)s.matAssign.size();
    rc.additionalMats = s.mats.data();
    rc.numAdditionalMats = (uint32_t)s.mats.size();
    rc.additionalTextures = s.texPtrs.data();
    rc.numAdditionalTextures = (uint32_t)s.texPtrs.size();
    rc.importedInstances = s.instances.data();
    rc.numInstances = (uint32_t)s.instances.size();
    rc.cameras = s.cameras.data();
    rc.numCameras = (uint32_t)s.cameras.size();
    rc.worlds = s.worlds.data();
    if (args.gpus > 1) {
        cfg.deviceIDs = devices.data();
        cfg.numDevices = args.gpus;
    }

    Manager mgr(cfg);              // aborts (FATAL) on failure, like the reference
    mgr.sync();

    const auto start = std::chrono::system_clock::now();
    mgr.mark(0);                   // an event on every device's stream, for the per-device lines
    for (uint32_t i = 0; i < args.numSteps; ++i)
        mgr.step();
    mgr.mark(1);
    mgr.sync();
    const auto end = std::chrono::system_clock::now();
    const double seconds = std::chrono::duration<double>(end - start).count();

    bool ok = true;
    mrx_renderer *top = (mrx_renderer *)mgr.nativeHandle();
    for (uint32_t g = 0; g < mgr.numShards(); ++g) {
        const uint32_t lo = mgr.shardFirstWorld(g), hi = mgr.shardFirstWorld(g + 1);
        mrx_renderer *sh = mrx_shard(top, (int)g);
        if (args.gpus > 1) {
            float ms = 0.0f;
            if (mrx_elapsed_ms(sh, &ms) != MRX_OK || !(ms > 0.0f))
                ms = (float)(seconds * 1000.0);
            std::printf("GPU %d: worlds [%u, %u) FPS %f\n", devices[g], lo, hi,
                        (double)args.numSteps * (double)(hi - lo) / (ms * 1e-3));
        }
        if (args.dump) {
            const bool rt = args.mode == Mode::Raycaster;
            const uint32_t resY = rt ? args.width : args.height;
            const std::string name = args.gpus == 1 ? args.outName : args.outName + ".gpu" + std::to_string(g);
            ok = dumpTiled(name, sh, hi - lo, args.width, resY, args.dumpDepth, rt) && ok;
        }
    }
    if (!ok)
        return EXIT_FAILURE;
    // whole node: every world of the job over the time all devices took
    const double fps = (double)args.numSteps * (double)args.numWorlds / seconds;
    std::printf("FPS %f\n", fps);
    std::printf("Average total step time: %f ms\n",
                1000.0 * seconds / (double)(args.numSteps ? args.numSteps : 1));
    return 0;
}
