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
    };

    // Aborts (FATAL-style, like the reference) when construction fails.
    Manager(const Config &cfg);
    ~Manager();

    void step();     // advance + render, asynchronous on the renderer's stream
    void render();   // the render half of step()
    void sync();     // wait for everything enqueued so far

    madrona::py::Tensor rgbTensor() const;
    madrona::py::Tensor depthTensor() const;
    madrona::py::Tensor segmaskTensor() const;

    madrona::py::Tensor instancePositionTensor() const;
    madrona::py::Tensor instanceRotationTensor() const;

    madrona::py::Tensor cameraPositionTensor() const;
    madrona::py::Tensor cameraRotationTensor() const;

    uint64_t rgbCudaPtr() const;
    uint64_t depthCudaPtr() const;
    uint64_t segmaskCudaPtr() const;

    // Additions with no counterpart in the reference (measurement / tests).
    madrona::py::Tensor visibilityTensor() const;   // needs MADRONA_MI355_VISIBILITY=1
    // i32 [instances], mutable: negative hides the instance from the next step on
    // (the ObjectID column, src/sim.cpp:152-156; src/sim.inl:5-16)
    madrona::py::Tensor instanceObjectTensor() const;
    float timeRenders(int steps);                   // device ms for `steps` renders
    void mark(int which);                           // HIP event 0/1 on the stream
    float elapsedMs();                              // event1 - event0, waits for 1
    uint64_t bytesPerStep() const;                  // algorithmic HBM bytes / render
    void *nativeHandle() const;                     // mrx_renderer *
    // output placement as mrx_placement reports it: candidates timed at creation
    int placement(float *candUs, int capacity, float *keptUs) const;
    void setStream(void *hipStream);                // launch on this stream from now on
    const char *renderPath() const;                 // "raster" (tiled raster kernels) or "bvh"

    uint32_t numAgents;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace madRender
