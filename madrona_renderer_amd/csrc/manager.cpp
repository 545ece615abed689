// madRender::Manager over the C-ABI (include/mrx.h).  Mirrors the control
// flow of the reference's Manager (/root/reference/src/mgr.cpp:505-665):
// construct -> first frame rendered; step(); tensor getters that wrap device
// pointers without copying.
#include "../../include/madrona_mi355/manager.hpp"
#include "../../include/mrx.h"

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace madRender {

namespace detail {
static bool g_throwOnError = false;
// The reference aborts on every error (FATAL / REQ_CUDA / assert:
// mgr.cpp:306,320,420,595).  Language bindings flip this so the same
// conditions surface as exceptions instead of killing the interpreter.
void setThrowOnError(bool v) { g_throwOnError = v; }

[[noreturn]] static void fatal(const std::string &msg)
{
    if (g_throwOnError)
        throw std::runtime_error(msg);
    std::fprintf(stderr, "FATAL: %s\n", msg.c_str());
    std::fflush(stderr);
    std::abort();
}
}  // namespace detail

using madrona::py::Tensor;
using madrona::py::TensorElementType;

struct Manager::Impl {
    mrx_renderer *r = nullptr;
    ~Impl() { mrx_destroy(r); }

    Tensor wrap(int which, uint32_t shard) const
    {
        int64_t dims[4] = { 0, 0, 0, 0 };
        int ndim = 0, dtype = 0, dev = 0;
        void *p = mrx_buffer_shard(r, (int)shard, which, dims, &ndim, &dtype, &dev);
        if (!p)
            detail::fatal(mrx_last_error());
        TensorElementType t = dtype == MRX_DTYPE_U8 ? TensorElementType::UInt8
                            : dtype == MRX_DTYPE_I32 ? TensorElementType::Int32
                                                     : TensorElementType::Float32;
        return Tensor(p, t, dims, ndim, dev);
    }
};

Manager::Manager(const Config &cfg)
    : impl_(new Impl())
{
    static_assert(sizeof(mrx_instance) == sizeof(ImportedInstance), "instance ABI");
    static_assert(sizeof(mrx_camera) == sizeof(ImportedCamera), "camera ABI");
    static_assert(sizeof(mrx_world_init) == sizeof(Sim::WorldInit), "world ABI");
    static_assert(sizeof(mrx_material) == sizeof(AdditionalMaterial), "material ABI");

    mrx_config c {};
    c.struct_size = sizeof(mrx_config);
    c.gpu_id = cfg.gpuID;
    c.num_worlds = cfg.numWorlds;
    c.render_mode = cfg.renderMode == RenderMode::Raytracer ? MRX_MODE_RAYTRACER
                                                            : MRX_MODE_RASTERIZER;
    c.view_width = cfg.batchRenderViewWidth;
    c.view_height = cfg.batchRenderViewHeight;
    const Config::RenderConfig &rc = cfg.rcfg;
    c.geo.vertices = reinterpret_cast<const float *>(rc.geoCfg.vertices);
    c.geo.uvs = reinterpret_cast<const float *>(rc.geoCfg.uvs);
    c.geo.indices = rc.geoCfg.indices;
    c.geo.mesh_vertex_offsets = rc.geoCfg.meshVertexOffsets;
    c.geo.mesh_index_offsets = rc.geoCfg.meshIndexOffsets;
    c.geo.mesh_materials = rc.geoCfg.meshMaterials;
    c.geo.num_vertices = rc.geoCfg.numVertices;
    c.geo.num_indices = rc.geoCfg.numIndices;
    c.geo.num_meshes = rc.geoCfg.numMeshes;
    c.asset_paths = rc.assetPaths;
    c.num_asset_paths = rc.numAssetPaths;
    c.mat_assignments = rc.matAssignments;
    c.num_mat_assignments = rc.numMatAssignments;
    c.materials = reinterpret_cast<const mrx_material *>(rc.additionalMats);
    c.num_materials = rc.numAdditionalMats;
    c.texture_paths = rc.additionalTextures;
    c.num_textures = rc.numAdditionalTextures;
    c.instances = reinterpret_cast<const mrx_instance *>(rc.importedInstances);
    c.num_instances = rc.numInstances;
    c.cameras = reinterpret_cast<const mrx_camera *>(rc.cameras);
    c.num_cameras = rc.numCameras;
    c.worlds = reinterpret_cast<const mrx_world_init *>(rc.worlds);
    c.stream = nullptr;
    c.device_ids = cfg.deviceIDs;
    c.num_devices = cfg.numDevices;
    c.max_instances_per_world = cfg.maxInstancesPerWorld;
    // build-only knobs travel by environment so Config stays field-compatible
    if (const char *v = std::getenv("MADRONA_MI355_VISIBILITY"))
        if (std::atoi(v) != 0)
            c.flags |= MRX_FLAG_VISIBILITY_IDS;
    if (const char *k = std::getenv("MADRONA_MI355_KERNEL"))
        c.kernel_variant = std::atoi(k);

    if (mrx_create(&c, &impl_->r) != MRX_OK)
        detail::fatal(mrx_last_error());

    // vestigial in the reference too (mgr.cpp:516-522)
    const char *num_agents_str = std::getenv("HIDESEEK_NUM_AGENTS");
    numAgents = num_agents_str ? (uint32_t)std::atoi(num_agents_str) : 1u;
    // mrx_create already rendered the first frame (mgr.cpp:524)
}

Manager::~Manager() {}

void Manager::step()
{
    if (mrx_step(impl_->r) != MRX_OK)
        detail::fatal(mrx_last_error());
}

void Manager::render()
{
    if (mrx_render(impl_->r) != MRX_OK)
        detail::fatal(mrx_last_error());
}

void Manager::sync()
{
    if (mrx_sync(impl_->r) != MRX_OK)
        detail::fatal(mrx_last_error());
}

Tensor Manager::rgbTensor(uint32_t shard) const { return impl_->wrap(MRX_BUF_RGB, shard); }
Tensor Manager::depthTensor(uint32_t shard) const { return impl_->wrap(MRX_BUF_DEPTH, shard); }
Tensor Manager::segmaskTensor(uint32_t shard) const { return impl_->wrap(MRX_BUF_SEGMASK, shard); }
Tensor Manager::visibilityTensor(uint32_t shard) const { return impl_->wrap(MRX_BUF_VISIBILITY, shard); }

Tensor Manager::instanceObjectTensor(uint32_t shard) const { return impl_->wrap(MRX_BUF_INSTANCE_OBJECT, shard); }
Tensor Manager::instanceScaleTensor(uint32_t shard) const { return impl_->wrap(MRX_BUF_INSTANCE_SCALE, shard); }

Tensor Manager::instancePositionTensor(uint32_t shard) const
{
    return impl_->wrap(MRX_BUF_INSTANCE_POSITION, shard);
}
Tensor Manager::instanceRotationTensor(uint32_t shard) const
{
    return impl_->wrap(MRX_BUF_INSTANCE_ROTATION, shard);
}
Tensor Manager::cameraPositionTensor(uint32_t shard) const
{
    return impl_->wrap(MRX_BUF_CAMERA_POSITION, shard);
}
Tensor Manager::cameraRotationTensor(uint32_t shard) const
{
    return impl_->wrap(MRX_BUF_CAMERA_ROTATION, shard);
}

uint64_t Manager::rgbCudaPtr(uint32_t shard) const { return (uint64_t)rgbTensor(shard).devicePtr(); }
uint64_t Manager::depthCudaPtr(uint32_t shard) const { return (uint64_t)depthTensor(shard).devicePtr(); }
uint64_t Manager::segmaskCudaPtr(uint32_t shard) const { return (uint64_t)segmaskTensor(shard).devicePtr(); }

void Manager::refreshObjects()
{
    if (mrx_refresh_objects(impl_->r) != MRX_OK)
        detail::fatal(mrx_last_error());
}

uint32_t Manager::numShards() const { return (uint32_t)mrx_num_shards(impl_->r); }

uint32_t Manager::shardFirstWorld(uint32_t shard) const
{
    const int64_t w = mrx_shard_first_world(impl_->r, (int)shard);
    if (w < 0)
        detail::fatal(mrx_last_error());
    return (uint32_t)w;
}

float Manager::timeRenders(int steps)
{
    float ms = 0.f;
    if (mrx_time_renders(impl_->r, steps, &ms) != MRX_OK)
        detail::fatal(mrx_last_error());
    return ms;
}

double Manager::timeStepsHost(int steps)
{
    double us = 0.0;
    if (mrx_time_steps_host(impl_->r, steps, &us) != MRX_OK)
        detail::fatal(mrx_last_error());
    return us;
}

void Manager::mark(int which)
{
    if (mrx_mark(impl_->r, which) != MRX_OK)
        detail::fatal(mrx_last_error());
}

float Manager::elapsedMs()
{
    float ms = 0.f;
    if (mrx_elapsed_ms(impl_->r, &ms) != MRX_OK)
        detail::fatal(mrx_last_error());
    return ms;
}

uint64_t Manager::bytesPerStep() const
{
    mrx_info_t info {};
    if (mrx_info_sized(impl_->r, &info, sizeof info) != MRX_OK)
        detail::fatal(mrx_last_error());
    return info.bytes_per_step;
}

void *Manager::nativeHandle() const { return impl_->r; }

int Manager::placement(float *candUs, int capacity, float *keptUs) const
{
    return mrx_placement(impl_->r, candUs, capacity, keptUs);
}

const char *Manager::renderPath() const
{
    mrx_info_t info {};
    if (mrx_info_sized(impl_->r, &info, sizeof info) != MRX_OK)
        detail::fatal(mrx_last_error());
    return info.render_path == 1 ? "bvh" : "raster";
}

void Manager::setStream(void *hipStream)
{
    if (mrx_set_stream(impl_->r, hipStream) != MRX_OK)
        detail::fatal(mrx_last_error());
}

void Manager::setShardStream(uint32_t shard, void *hipStream)
{
    mrx_renderer *sh = mrx_shard(impl_->r, (int)shard);
    if (!sh || mrx_set_stream(sh, hipStream) != MRX_OK)
        detail::fatal(mrx_last_error());
}

}  // namespace madRender
