// madRender::Manager -- the host C++ API of the batch renderer, kept
// member-for-member with the reference's class
// (/root/reference/src/mgr.hpp:29-120) so callers such as
// /root/reference/src/headless.cpp:48-61 and src/bindings.cpp:183-233 keep
// working.  The implementation (madrona_renderer_amd/csrc/manager.cpp) is a
// thin layer over the C-ABI in include/mrx.h; all rendering happens in HIP
// kernels on the MI355X.
#pragma once

#include <memory>
#include <string>

#include "types.hpp"

namespace madRender {

using AdditionalMaterial = madrona::imp::SourceMaterial;

struct ImportedAsset {
    std::string path;
    int32_t matID;   // index into the additional materials, -1 = none
};

class Manager {
public:
    enum class RenderMode { Rasterizer, Raytracer };

    struct GeometryConfig {
        const madrona::math::Vector3 *vertices;
        const madrona::math::Vector2 *uvs;
        const uint32_t *indices;
        const uint32_t *meshVertexOffsets;
        const uint32_t *meshIndexOffsets;
        const int32_t *meshMaterials;
        uint32_t numVertices;
        uint32_t numIndices;
        uint32_t numMeshes;
    };

    struct Config {
        int gpuID;
        uint32_t numWorlds;
        RenderMode renderMode;
        uint32_t batchRenderViewWidth = 64;
        uint32_t batchRenderViewHeight = 64;
        madrona::render::APIBackend *extRenderAPI = nullptr;  // ignored
        madrona::render::GPUDevice *extRenderDev = nullptr;   // ignored
        bool headlessMode = false;

        struct RenderConfig {
            GeometryConfig geoCfg;
            const char **assetPaths;
            uint32_t numAssetPaths;
            int32_t *matAssignments;
            uint32_t numMatAssignments;
            const AdditionalMaterial *additionalMats;
            uint32_t numAdditionalMats;
            const char **additionalTextures;
            uint32_t numAdditionalTextures;
            ImportedInstance *importedInstances;
            uint32_t numInstances;
            ImportedCamera *cameras;
            uint32_t numCameras;
            Sim::WorldInit *worlds;
        } rcfg;

        // ---- additions (trailing, defaulted: the reference's initialisers keep compiling) ----
        // Single-process multi-device: with numDevices > 1 this one Manager spans deviceIDs[0 ..
        // numDevices) -- the worlds are split into contiguous ranges, one shard (own tensors, own
        // launch) per listed device, step() launches on all of them; gpuID is then ignored.
        const int *deviceIDs = nullptr;
        uint32_t numDevices = 0;
        // Rows per world at least (the reference's maxInstancesPerWorld, src/mgr.cpp:378-388):
        // spare rows start hidden and unbound, see refreshObjects().
        uint32_t maxInstancesPerWorld = 0;
    };

    // Aborts (FATAL-style, like the reference) when construction fails.
    Manager(const Config &cfg);
    ~Manager();

    void step();     // advance + render, asynchronous on the renderer's stream
    void render();   // the render half of step()
    void sync();     // wait for everything enqueued so far

    // (the tensor getters take the shard of a multi-device Manager; a Manager of one device
    // has shard 0 only, so the reference's argument-less calls are unchanged)
    madrona::py::Tensor rgbTensor(uint32_t shard = 0) const;
    madrona::py::Tensor depthTensor(uint32_t shard = 0) const;
    madrona::py::Tensor segmaskTensor(uint32_t shard = 0) const;

    madrona::py::Tensor instancePositionTensor(uint32_t shard = 0) const;
    madrona::py::Tensor instanceRotationTensor(uint32_t shard = 0) const;

    madrona::py::Tensor cameraPositionTensor(uint32_t shard = 0) const;
    madrona::py::Tensor cameraRotationTensor(uint32_t shard = 0) const;

    uint64_t rgbCudaPtr(uint32_t shard = 0) const;
    uint64_t depthCudaPtr(uint32_t shard = 0) const;
    uint64_t segmaskCudaPtr(uint32_t shard = 0) const;

    // Additions with no counterpart in the reference (measurement / tests).
    madrona::py::Tensor visibilityTensor(uint32_t shard = 0) const;   // needs MADRONA_MI355_VISIBILITY=1
    // i32 [instances], mutable: negative hides the instance from the next step on
    // (the ObjectID column, src/sim.cpp:152-156; src/sim.inl:5-16)
    madrona::py::Tensor instanceObjectTensor(uint32_t shard = 0) const;
    madrona::py::Tensor instanceScaleTensor(uint32_t shard = 0) const;
    // binds every row to the (non-negative) object id its ObjectID column now holds: a spare
    // row gets its geometry, an existing row swaps it (makeEntityRenderable at run time,
    // src/sim.inl:5-8); waits for the device
    void refreshObjects();
    uint32_t numShards() const;                     // devices this Manager spans
    // worlds [shardFirstWorld(i), shardFirstWorld(i + 1)) live on shard i
    uint32_t shardFirstWorld(uint32_t shard) const;
    float timeRenders(int steps);                   // device ms for `steps` renders
    double timeStepsHost(int steps);                // host us per step() call, `steps` calls back to back
    void mark(int which);                           // HIP event 0/1 on the stream
    float elapsedMs();                              // event1 - event0, waits for 1
    uint64_t bytesPerStep() const;                  // algorithmic HBM bytes / render
    void *nativeHandle() const;                     // mrx_renderer *
    // output placement as mrx_placement reports it: candidates timed at creation
    int placement(float *candUs, int capacity, float *keptUs) const;
    void setStream(void *hipStream);                // launch on this stream from now on
    void setShardStream(uint32_t shard, void *hipStream);   // the same for one shard of several
    const char *renderPath() const;                 // "raster" (tiled raster kernels) or "bvh"

    uint32_t numAgents;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace madRender
