// Python module `madrona_renderer` (pybind11): the same classes, keyword
// arguments and methods as the reference's nanobind module
// (/root/reference/src/bindings.cpp:18-236), bound to the MI355X Manager.
// Tensors are exported through DLPack (device type ROCm), so
// `tensor.to_torch()` aliases the renderer's HBM buffers without a copy.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <array>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/madrona_mi355/manager.hpp"
#include "../../include/mrx.h"

namespace py = pybind11;

namespace madRender {
namespace detail { void setThrowOnError(bool v); }
}

namespace {

using namespace madRender;
using madrona::py::Tensor;
using madrona::py::TensorElementType;

// ---- minimal DLPack (v0.8 ABI) ------------------------------------------
enum { kDLCPU = 1, kDLROCM = 10 };
enum { kDLInt = 0, kDLUInt = 1, kDLFloat = 2 };
struct DLDevice { int32_t device_type; int32_t device_id; };
struct DLDataType { uint8_t code; uint8_t bits; uint16_t lanes; };
struct DLTensor {
    void *data;
    DLDevice device;
    int32_t ndim;
    DLDataType dtype;
    int64_t *shape;
    int64_t *strides;
    uint64_t byte_offset;
};
struct DLManagedTensor {
    DLTensor dl_tensor;
    void *manager_ctx;
    void (*deleter)(DLManagedTensor *);
};

struct DLHolder {
    DLManagedTensor managed;
    int64_t shape[4];
    PyObject *owner;   // keeps the renderer alive while torch holds the view
};

void dlDeleter(DLManagedTensor *m)
{
    DLHolder *h = static_cast<DLHolder *>(m->manager_ctx);
    if (h->owner && Py_IsInitialized()) {
        py::gil_scoped_acquire gil;
        Py_DECREF(h->owner);
    }
    delete h;
}

void capsuleDestructor(PyObject *cap)
{
    // consumed capsules are renamed "used_dltensor" and own nothing
    if (PyCapsule_IsValid(cap, "dltensor")) {
        auto *m = static_cast<DLManagedTensor *>(PyCapsule_GetPointer(cap, "dltensor"));
        if (m && m->deleter)
            m->deleter(m);
    }
}

// Python-side tensor handle: the view plus a reference to its renderer.
struct PyTensor {
    Tensor t;
    py::object owner;
};

py::object makeCapsule(const PyTensor &self)
{
    auto *h = new DLHolder();
    const Tensor &t = self.t;
    DLTensor &d = h->managed.dl_tensor;
    d.data = t.devicePtr();
    d.device = DLDevice { kDLROCM, t.gpuID() };
    d.ndim = (int32_t)t.numDims();
    switch (t.type()) {
    case TensorElementType::UInt8: d.dtype = DLDataType { kDLUInt, 8, 1 }; break;
    case TensorElementType::Int32: d.dtype = DLDataType { kDLInt, 32, 1 }; break;
    default: d.dtype = DLDataType { kDLFloat, 32, 1 }; break;
    }
    for (int i = 0; i < d.ndim; ++i)
        h->shape[i] = t.dims()[i];
    d.shape = h->shape;
    d.strides = nullptr;
    d.byte_offset = 0;
    h->managed.manager_ctx = h;
    h->managed.deleter = dlDeleter;
    h->owner = self.owner.ptr();
    Py_XINCREF(h->owner);
    PyObject *cap = PyCapsule_New(&h->managed, "dltensor", capsuleDestructor);
    if (!cap) {
        dlDeleter(&h->managed);
        throw py::error_already_set();
    }
    return py::reinterpret_steal<py::object>(cap);
}

template <typename T, int N>
using FArr = py::array_t<T, py::array::c_style | py::array::forcecast>;

PyTensor wrapTensor(py::object self, Tensor t) { return PyTensor { t, std::move(self) }; }

// `shard` of the tensor getters: None means "the renderer's one shard" -- an error when the
// renderer spans several devices (each shard's tensors live on its own device)
uint32_t shardOf(const py::object &self, const py::object &shard)
{
    const uint32_t n = self.cast<Manager &>().numShards();
    if (shard.is_none()) {
        if (n > 1)
            throw py::value_error("this renderer spans " + std::to_string(n) +
                                  " devices: say which shard's tensor (shard=i)");
        return 0;
    }
    const long i = shard.cast<long>();
    if (i < 0 || (unsigned long)i >= n)
        throw py::index_error("shard " + std::to_string(i) + " of " + std::to_string(n));
    return (uint32_t)i;
}

}  // namespace

PYBIND11_MODULE(madrona_renderer, m)
{
    m.doc() = "MI355X-native batch renderer with the madrona_renderer API";
    madRender::detail::setThrowOnError(true);

    py::enum_<Manager::RenderMode>(m, "RenderMode")
        .value("Rasterizer", Manager::RenderMode::Rasterizer)
        .value("Raytracer", Manager::RenderMode::Raytracer);

    py::class_<ImportedAsset>(m, "ImportedAsset")
        .def(py::init([](std::string path, int64_t mat_id) {
                 return ImportedAsset { std::move(path), (int32_t)mat_id };
             }),
             py::arg("path"), py::arg("mat_id"));

    py::class_<AdditionalMaterial>(m, "AdditionalMaterial")
        .def(py::init([](const std::array<float, 4> &color, int64_t texture_id,
                         float roughness, float metalness) {
                 AdditionalMaterial mat {};
                 mat.color = { color[0], color[1], color[2], color[3] };
                 mat.textureIdx = (int32_t)texture_id;
                 mat.roughness = roughness;
                 mat.metalness = metalness;
                 return mat;
             }),
             py::arg("color"), py::arg("texture_id"), py::arg("roughness"),
             py::arg("metalness"));

    py::class_<ImportedInstance>(m, "ImportedInstance")
        .def(py::init([](const std::array<float, 3> &pos, const std::array<float, 4> &rot,
                         const std::array<float, 3> &scale, int64_t object_id) {
                 ImportedInstance i {};
                 i.position = { pos[0], pos[1], pos[2] };
                 i.rotation = { rot[0], rot[1], rot[2], rot[3] };
                 i.scale = { scale[0], scale[1], scale[2] };
                 i.objectID = (int32_t)object_id;
                 return i;
             }),
             py::arg("position"), py::arg("rotation"), py::arg("scale"),
             py::arg("object_id"));

    py::class_<ImportedCamera>(m, "ImportedCamera")
        .def(py::init([](const std::array<float, 3> &pos, const std::array<float, 4> &rot) {
                 ImportedCamera c {};
                 c.position = { pos[0], pos[1], pos[2] };
                 c.rotation = { rot[0], rot[1], rot[2], rot[3] };
                 return c;
             }),
             py::arg("position"), py::arg("rotation"));

    py::class_<Sim::WorldInit>(m, "WorldInit")
        .def(py::init([](int64_t num_instances, int64_t instance_offset, int64_t num_cameras,
                         int64_t camera_offset) {
                 return Sim::WorldInit { (uint32_t)num_instances, (uint32_t)instance_offset,
                                         (uint32_t)num_cameras, (uint32_t)camera_offset };
             }),
             py::arg("num_instances"), py::arg("instance_offset"), py::arg("num_cameras"),
             py::arg("camera_offset"));

    m.def("inspect", [](py::array_t<uint32_t, py::array::c_style> a) {
        std::printf("Array data pointer : %p\n", (const void *)a.data());
        std::printf("Array dimension : %zu\n", (size_t)a.ndim());
        for (py::ssize_t i = 0; i < a.ndim(); ++i) {
            std::printf("Array dimension [%zu] : %zu\n", (size_t)i, (size_t)a.shape(i));
            std::printf("Array stride    [%zu] : %zd\n", (size_t)i,
                        (ssize_t)(a.strides(i) / (py::ssize_t)sizeof(uint32_t)));
        }
        std::printf("Device ID = 0 (cpu=1, cuda=0)\n");
        std::printf("Array dtype: int16=0, uint32=1, float32=0\n");
    });

    py::class_<PyTensor>(m, "Tensor")
        .def("__dlpack__",
             [](const PyTensor &self, py::kwargs) { return makeCapsule(self); })
        .def("__dlpack_device__",
             [](const PyTensor &self) { return py::make_tuple((int)kDLROCM, self.t.gpuID()); })
        .def("to_torch",
             [](py::object self) {
                 py::object torch = py::module_::import("torch");
                 return torch.attr("from_dlpack")(self);
             })
        .def("device_ptr", [](const PyTensor &self) { return (uint64_t)self.t.devicePtr(); })
        .def_property_readonly("shape",
                               [](const PyTensor &self) {
                                   py::tuple s(self.t.numDims());
                                   for (int i = 0; i < self.t.numDims(); ++i)
                                       s[i] = self.t.dims()[i];
                                   return s;
                               })
        .def_property_readonly("gpu_id", [](const PyTensor &self) { return self.t.gpuID(); });

    py::class_<Manager>(m, "MadronaRenderer")
        .def(py::init([](int gpu_id, int num_worlds, Manager::RenderMode render_mode,
                         int batch_render_view_width, int batch_render_view_height,
                         const std::vector<ImportedAsset> &asset_paths,
                         py::array_t<float, py::array::c_style | py::array::forcecast> mesh_vertices,
                         py::array_t<float, py::array::c_style | py::array::forcecast> mesh_uvs,
                         py::array_t<uint32_t, py::array::c_style | py::array::forcecast> mesh_indices,
                         py::array_t<uint32_t, py::array::c_style | py::array::forcecast> mesh_vertex_offsets,
                         py::array_t<uint32_t, py::array::c_style | py::array::forcecast> mesh_indices_offsets,
                         py::array_t<int32_t, py::array::c_style | py::array::forcecast> mesh_materials,
                         const std::vector<AdditionalMaterial> &mats,
                         const std::vector<std::string> &texture_paths,
                         const std::vector<ImportedInstance> &instances,
                         const std::vector<ImportedCamera> &cameras,
                         const std::vector<Sim::WorldInit> &worlds,
                         const std::vector<int> &device_ids, int max_instances_per_world) {
                 if (mesh_vertices.size() && (mesh_vertices.ndim() != 2 || mesh_vertices.shape(1) != 3))
                     throw py::value_error("mesh_vertices must have shape [N, 3]");
                 if (mesh_uvs.size() && (mesh_uvs.ndim() != 2 || mesh_uvs.shape(1) != 2))
                     throw py::value_error("mesh_uvs must have shape [N, 2]");
                 if ((size_t)num_worlds != worlds.size())
                     throw py::value_error("num_worlds does not match len(worlds)");
                 // the C ABI carries no lengths for the per-mesh arrays and reads
                 // uvs row-for-row with vertices: check them here
                 if (mesh_indices_offsets.size() != mesh_vertex_offsets.size() ||
                     mesh_materials.size() != mesh_vertex_offsets.size())
                     throw py::value_error("mesh_vertex_offsets, mesh_indices_offsets and "
                                           "mesh_materials must have one entry per mesh");
                 if ((mesh_uvs.size() ? mesh_uvs.shape(0) : 0) !=
                     (mesh_vertices.size() ? mesh_vertices.shape(0) : 0))
                     throw py::value_error("mesh_uvs must have one row per row of mesh_vertices");
                 std::vector<const char *> cstrs(asset_paths.size());
                 std::vector<int32_t> mat_assignments(asset_paths.size());
                 for (size_t i = 0; i < asset_paths.size(); ++i) {
                     cstrs[i] = asset_paths[i].path.c_str();
                     mat_assignments[i] = asset_paths[i].matID;
                 }
                 std::vector<const char *> texture_cstrs(texture_paths.size());
                 for (size_t i = 0; i < texture_paths.size(); ++i)
                     texture_cstrs[i] = texture_paths[i].c_str();

                 Manager::Config cfg {};
                 cfg.gpuID = gpu_id;
                 cfg.numWorlds = (uint32_t)num_worlds;
                 cfg.renderMode = render_mode;
                 cfg.batchRenderViewWidth = (uint32_t)batch_render_view_width;
                 cfg.batchRenderViewHeight = (uint32_t)batch_render_view_height;
                 auto &g = cfg.rcfg.geoCfg;
                 g.vertices = (const madrona::math::Vector3 *)mesh_vertices.data();
                 g.uvs = (const madrona::math::Vector2 *)mesh_uvs.data();
                 g.indices = mesh_indices.data();
                 g.meshVertexOffsets = mesh_vertex_offsets.data();
                 g.meshIndexOffsets = mesh_indices_offsets.data();
                 g.meshMaterials = mesh_materials.data();
                 g.numVertices = mesh_vertices.size() ? (uint32_t)mesh_vertices.shape(0) : 0;
                 g.numIndices = (uint32_t)mesh_indices.size();
                 g.numMeshes = (uint32_t)mesh_vertex_offsets.size();
                 cfg.rcfg.assetPaths = cstrs.data();
                 cfg.rcfg.numAssetPaths = (uint32_t)cstrs.size();
                 cfg.rcfg.matAssignments = mat_assignments.data();
                 cfg.rcfg.numMatAssignments = (uint32_t)mat_assignments.size();
                 cfg.rcfg.additionalMats = mats.data();
                 cfg.rcfg.numAdditionalMats = (uint32_t)mats.size();
                 cfg.rcfg.additionalTextures = texture_cstrs.data();
                 cfg.rcfg.numAdditionalTextures = (uint32_t)texture_cstrs.size();
                 cfg.rcfg.importedInstances = const_cast<ImportedInstance *>(instances.data());
                 cfg.rcfg.numInstances = (uint32_t)instances.size();
                 cfg.rcfg.cameras = const_cast<ImportedCamera *>(cameras.data());
                 cfg.rcfg.numCameras = (uint32_t)cameras.size();
                 cfg.rcfg.worlds = const_cast<Sim::WorldInit *>(worlds.data());
                 // additions: one renderer over several devices; spare instance rows
                 cfg.deviceIDs = device_ids.empty() ? nullptr : device_ids.data();
                 cfg.numDevices = (uint32_t)device_ids.size();
                 if (max_instances_per_world < 0)
                     throw py::value_error("max_instances_per_world must not be negative");
                 cfg.maxInstancesPerWorld = (uint32_t)max_instances_per_world;
                 return new Manager(cfg);
             }),
             py::arg("gpu_id"), py::arg("num_worlds"), py::arg("render_mode"),
             py::arg("batch_render_view_width"), py::arg("batch_render_view_height"),
             py::arg("asset_paths"), py::arg("mesh_vertices"), py::arg("mesh_uvs"),
             py::arg("mesh_indices"), py::arg("mesh_vertex_offsets"),
             py::arg("mesh_indices_offsets"), py::arg("mesh_materials"), py::arg("materials"),
             py::arg("texture_paths"), py::arg("instances"), py::arg("cameras"),
             py::arg("worlds"),
             // not in the reference (its callers never pass them): device_ids = [d0, d1, ...] makes this
             // one renderer span several devices (gpu_id is then ignored), max_instances_per_world
             // reserves hidden, unbound rows per world (see refresh_objects)
             py::arg("device_ids") = std::vector<int>(), py::arg("max_instances_per_world") = 0)
        .def("step", &Manager::step)
        .def("render", &Manager::render)
        .def("sync", &Manager::sync)
        .def("rgb_tensor",
             [](py::object self, py::object shard) { return wrapTensor(self, self.cast<Manager &>().rgbTensor(shardOf(self, shard))); },
             py::arg("shard") = py::none())
        .def("depth_tensor",
             [](py::object self, py::object shard) { return wrapTensor(self, self.cast<Manager &>().depthTensor(shardOf(self, shard))); },
             py::arg("shard") = py::none())
        .def("segmask_tensor",
             [](py::object self, py::object shard) { return wrapTensor(self, self.cast<Manager &>().segmaskTensor(shardOf(self, shard))); },
             py::arg("shard") = py::none())
        .def("visibility_tensor",
             [](py::object self, py::object shard) { return wrapTensor(self, self.cast<Manager &>().visibilityTensor(shardOf(self, shard))); },
             py::arg("shard") = py::none())
        .def("rgb_cuda_ptr", [](py::object self, py::object shard) { return self.cast<Manager &>().rgbCudaPtr(shardOf(self, shard)); },
             py::arg("shard") = py::none())
        .def("depth_cuda_ptr", [](py::object self, py::object shard) { return self.cast<Manager &>().depthCudaPtr(shardOf(self, shard)); },
             py::arg("shard") = py::none())
        .def("segmask_cuda_ptr", [](py::object self, py::object shard) { return self.cast<Manager &>().segmaskCudaPtr(shardOf(self, shard)); },
             py::arg("shard") = py::none())
        .def("instance_scale_tensor",
             [](py::object self, py::object shard) {
                 return wrapTensor(self, self.cast<Manager &>().instanceScaleTensor(shardOf(self, shard)));
             },
             py::arg("shard") = py::none())
        // binds rows to the object ids their ObjectID column holds (spare rows: max_instances_per_world)
        .def("refresh_objects", &Manager::refreshObjects)
        .def_property_readonly("num_shards", &Manager::numShards)
        .def("shard_first_world", &Manager::shardFirstWorld, py::arg("shard"))
        .def("instance_position_tensor",
             [](py::object self, py::object shard) {
                 return wrapTensor(self, self.cast<Manager &>().instancePositionTensor(shardOf(self, shard)));
             },
             py::arg("shard") = py::none())
        .def("instance_object_tensor",
             [](py::object self, py::object shard) {
                 return wrapTensor(self, self.cast<Manager &>().instanceObjectTensor(shardOf(self, shard)));
             },
             py::arg("shard") = py::none())
        .def("instance_rotation_tensor",
             [](py::object self, py::object shard) {
                 return wrapTensor(self, self.cast<Manager &>().instanceRotationTensor(shardOf(self, shard)));
             },
             py::arg("shard") = py::none())
        .def("camera_position_tensor",
             [](py::object self, py::object shard) {
                 return wrapTensor(self, self.cast<Manager &>().cameraPositionTensor(shardOf(self, shard)));
             },
             py::arg("shard") = py::none())
        .def("camera_rotation_tensor",
             [](py::object self, py::object shard) {
                 return wrapTensor(self, self.cast<Manager &>().cameraRotationTensor(shardOf(self, shard)));
             },
             py::arg("shard") = py::none())
        .def("time_renders", &Manager::timeRenders, py::arg("steps"))
        .def("time_steps_host", &Manager::timeStepsHost, py::arg("steps"))
        .def("mark", &Manager::mark, py::arg("which"))
        .def("elapsed_ms", &Manager::elapsedMs)
        .def("bytes_per_step", &Manager::bytesPerStep)
        .def("render_path", [](Manager &self) { return std::string(self.renderPath()); })
        .def("placement",
             [](Manager &self) {
                 float us[16] = {}, kept = 0.f;
                 const int n = self.placement(us, 16, &kept);
                 py::list cands;
                 for (int i = 0; i < n && i < 16; ++i)
                     cands.append(us[i]);
                 py::dict d;
                 d["tries"] = n > 0 ? n : 1;
                 d["candidates_us"] = cands;
                 d["kept_us"] = n > 0 ? py::object(py::float_(kept)) : py::object(py::none());
                 return d;
             })
        .def("native_handle", [](Manager &self) { return (uint64_t)self.nativeHandle(); })
        // e.g. r.set_stream(torch.cuda.current_stream().cuda_stream): pose writes and
        // step() are then ordered on that stream without a host synchronisation
        .def("set_stream",
             [](Manager &self, uint64_t stream, py::object shard) {
                 if (shard.is_none())
                     self.setStream((void *)stream);
                 else
                     self.setShardStream(shard.cast<uint32_t>(), (void *)stream);
             },
             py::arg("stream"), py::arg("shard") = py::none())
        .def_readonly("num_agents", &Manager::numAgents);
}
