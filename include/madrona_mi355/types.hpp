// Plain-old-data types the Manager API is written in.  They stand in for the
// un-vendored <madrona/math.hpp>, <madrona/importer.hpp> and
// <madrona/py/utils.hpp> declarations the reference headers pull in
// (/root/reference/src/mgr.hpp:5-9, src/sim.hpp:31-50), keeping the same names,
// member order and sizes so reference-side code compiles against them.
#pragma once

#include <array>
#include <cstdint>

namespace madrona {

namespace math {
struct Vector2 { float x, y; };
struct Vector3 { float x, y, z; };
struct Vector4 { float x, y, z, w; };
struct Quat { float w, x, y, z; };
struct Diag3x3 { float d0, d1, d2; };
}  // namespace math

namespace imp {
// as filled at /root/reference/src/bindings.cpp:44-49
struct SourceMaterial {
    math::Vector4 color;
    int32_t textureIdx;
    float roughness;
    float metalness;
};
}  // namespace imp

namespace render {
// External Vulkan objects of the reference's viewer path; accepted and ignored.
struct APIBackend;
struct GPUDevice;
}  // namespace render

namespace py {

enum class TensorElementType { UInt8, Int8, Int16, Int32, Int64, Float16, Float32 };

// Non-owning view of a device buffer (stand-in for madrona::py::Tensor as the
// reference uses it: /root/reference/src/mgr.cpp:192,552-557,609).
class Tensor {
public:
    Tensor() = default;
    Tensor(void *dev_ptr, TensorElementType type,
           std::initializer_list<int64_t> dims, int gpu_id)
        : ptr_(dev_ptr), type_(type), ndim_(0), gpu_(gpu_id)
    {
        for (int64_t d : dims)
            if (ndim_ < 4) dims_[ndim_++] = d;
    }
    Tensor(void *dev_ptr, TensorElementType type, const int64_t *dims, int ndim,
           int gpu_id)
        : ptr_(dev_ptr), type_(type), ndim_(ndim < 4 ? ndim : 4), gpu_(gpu_id)
    {
        for (int i = 0; i < ndim_; ++i) dims_[i] = dims[i];
    }
    void *devicePtr() const { return ptr_; }
    TensorElementType type() const { return type_; }
    int64_t numDims() const { return ndim_; }
    const int64_t *dims() const { return dims_.data(); }
    bool isOnGPU() const { return gpu_ >= 0; }
    int gpuID() const { return gpu_; }
    int64_t numBytesPerItem() const
    {
        switch (type_) {
        case TensorElementType::UInt8: case TensorElementType::Int8: return 1;
        case TensorElementType::Int16: case TensorElementType::Float16: return 2;
        case TensorElementType::Int64: return 8;
        default: return 4;
        }
    }
private:
    void *ptr_ = nullptr;
    TensorElementType type_ = TensorElementType::UInt8;
    std::array<int64_t, 4> dims_ { 0, 0, 0, 0 };
    int ndim_ = 0;
    int gpu_ = -1;
};

}  // namespace py
}  // namespace madrona

namespace madRender {

// /root/reference/src/sim.hpp:31-36 (44 bytes)
struct ImportedInstance {
    madrona::math::Vector3 position;
    madrona::math::Quat rotation;
    madrona::math::Diag3x3 scale;
    int32_t objectID;
};

// /root/reference/src/sim.hpp:47-50 (28 bytes)
struct ImportedCamera {
    madrona::math::Vector3 position;
    madrona::math::Quat rotation;
};

// Only the part of the reference's Sim that the Manager API names
// (/root/reference/src/sim.hpp:76-82); the ECS program itself is replaced by
// static world-major tables on the device.
struct Sim {
    struct WorldInit {
        uint32_t numInstances;
        uint32_t instancesOffset;
        uint32_t numCameras;
        uint32_t camerasOffset;
    };
};

static_assert(sizeof(ImportedInstance) == 44, "ImportedInstance layout");
static_assert(sizeof(ImportedCamera) == 28, "ImportedCamera layout");
static_assert(sizeof(Sim::WorldInit) == 16, "WorldInit layout");
static_assert(sizeof(madrona::imp::SourceMaterial) == 28, "SourceMaterial layout");

}  // namespace madRender
