/*
 * mrx.h -- C-ABI of the MI355X batch renderer (libmrx_hip.so).
 *
 * This is the drop-in boundary for the reference's per-frame hot path
 *   Manager::step()                      /root/reference/src/mgr.cpp:529-546
 * and the state around it that the path needs (construction, tensor export).
 * The reference's own boundary is the C++ class madRender::Manager
 * (/root/reference/src/mgr.hpp:29-120); its Python module binds that class
 * (/root/reference/src/bindings.cpp:123-233).  Everything below Manager --
 * the un-vendored Madrona executor, RenderingSystem, BatchRenderer and BVH
 * tracer -- is replaced by the HIP kernels behind these entry points.
 *
 * Plain C types only: pointers, sizes, PODs.  No torch, no C++ types.
 * Every function returns 0 on success or a negative MRX_E_* code;
 * mrx_last_error() gives the message of the calling thread's last failure.
 * The library is HIP-only: mrx_create() fails with MRX_E_NO_DEVICE when no
 * gfx950 device is usable -- there is no CPU fallback.
 */
#ifndef MRX_H
#define MRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRX_ABI_VERSION 4

enum {
    MRX_OK = 0,
    MRX_E_INVALID = -1,     /* bad argument / inconsistent config          */
    MRX_E_NO_DEVICE = -2,   /* no usable HIP device                        */
    MRX_E_HIP = -3,         /* a HIP runtime call failed                   */
    MRX_E_ASSET = -4,       /* an asset file could not be read or parsed   */
    MRX_E_UNSUPPORTED = -5  /* e.g. segmask in Rasterizer mode             */
};

/* Manager::RenderMode, /root/reference/src/mgr.hpp:31-34 */
enum { MRX_MODE_RASTERIZER = 0, MRX_MODE_RAYTRACER = 1 };

/* madRender::ImportedInstance, /root/reference/src/sim.hpp:31-36 (44 bytes,
 * rotation is w,x,y,z). */
typedef struct {
    float position[3];
    float rotation[4];
    float scale[3];
    int32_t object_id;
} mrx_instance;

/* madRender::ImportedCamera, /root/reference/src/sim.hpp:47-50 (28 bytes). */
typedef struct {
    float position[3];
    float rotation[4];
} mrx_camera;

/* madRender::Sim::WorldInit, /root/reference/src/sim.hpp:76-82 (16 bytes). */
typedef struct {
    uint32_t num_instances;
    uint32_t instances_offset;
    uint32_t num_cameras;
    uint32_t cameras_offset;
} mrx_world_init;

/* madrona::imp::SourceMaterial as the reference fills it,
 * /root/reference/src/bindings.cpp:44-49 (28 bytes). */
typedef struct {
    float color[4];
    int32_t texture_idx;    /* -1: untextured */
    float roughness;
    float metalness;
} mrx_material;

/* Manager::GeometryConfig, /root/reference/src/mgr.hpp:36-47. */
typedef struct {
    const float *vertices;              /* [num_vertices][3] */
    const float *uvs;                   /* [num_vertices][2] */
    const uint32_t *indices;            /* [num_indices]     */
    const uint32_t *mesh_vertex_offsets;/* [num_meshes]      */
    const uint32_t *mesh_index_offsets; /* [num_meshes]      */
    const int32_t *mesh_materials;      /* [num_meshes]      */
    uint32_t num_vertices;
    uint32_t num_indices;
    uint32_t num_meshes;
} mrx_geometry;

enum {
    /* also write a per-pixel int32 visibility buffer (world-local triangle
     * index, -1 = background); parity tests use it for bit-exact checks */
    MRX_FLAG_VISIBILITY_IDS = 1u << 0,
    /* several devices (device_ids): mrx_step only posts the render to the per-device host threads and returns;
     * every other entry point joins them first (the same as MRX_SHARD_ASYNC=1 in the environment) */
    MRX_FLAG_SHARD_ASYNC = 1u << 1
};

/* Manager::Config + Config::RenderConfig, /root/reference/src/mgr.hpp:49-88.
 * All pointers are borrowed for the duration of mrx_create() only. */
typedef struct {
    uint32_t struct_size;       /* = sizeof(mrx_config), ABI check */
    int32_t gpu_id;
    uint32_t num_worlds;
    int32_t render_mode;
    uint32_t view_width;
    uint32_t view_height;
    mrx_geometry geo;
    const char *const *asset_paths;
    uint32_t num_asset_paths;
    const int32_t *mat_assignments;     /* per asset path, -1 = none */
    uint32_t num_mat_assignments;
    const mrx_material *materials;
    uint32_t num_materials;
    const char *const *texture_paths;
    uint32_t num_textures;
    const mrx_instance *instances;
    uint32_t num_instances;
    const mrx_camera *cameras;
    uint32_t num_cameras;
    const mrx_world_init *worlds;       /* [num_worlds] */
    /* build-only knobs (no counterpart in the reference) */
    void *stream;               /* hipStream_t to launch on; NULL = null stream */
    uint32_t flags;             /* MRX_FLAG_* */
    int32_t kernel_variant;     /* 0 = default (raster kernels up to 128 triangles per
                                 * world, BVH path from 129 -- from 65 for batches of up to 640
                                 * 64x64 views, from 91 for up to 1024 untextured ones; and for
                                 * Raytracer-mode batches of many large views of small worlds:
                                 * mrx_dispatch_flat);
                                 * 1 = brute-force cross-check;
                                 * 2 = BVH path always; 3 = raster kernels always */
    /* -- ABI 3 (a caller that sets struct_size = MRX_CONFIG_V2_SIZE passes none of these) --
     * Single-process multi-device: with num_devices > 1 the renderer spans device_ids[0 ..
     * num_devices): the worlds are split into contiguous ranges (sizes differ by at most one,
     * as scenes.shard_range), one shard per listed device -- its own tensors, launched on that
     * device's null stream -- and mrx_step launches on all of them.  gpu_id is then ignored;
     * an id may repeat (several shards on one device).  The reference has a single gpuID
     * (/root/reference/src/mgr.hpp:50) and every caller constructs ONE Manager
     * (scripts/test.py:112-130, src/bindings.cpp:183-205): this is that constructor, wider. */
    const int32_t *device_ids;
    uint32_t num_devices;       /* 0 or 1: one device, gpu_id */
    /* Rows per world at least (0 = exactly num_instances of each world): the reference sizes
     * its renderer by maxInstancesPerWorld (/root/reference/src/mgr.cpp:378-388) and creates
     * renderables at run time (src/sim.inl:5-8).  Spare rows start hidden and unbound
     * (ObjectID -1, identity pose); see mrx_refresh_objects. */
    uint32_t max_instances_per_world;
} mrx_config;
#define MRX_CONFIG_V2_SIZE ((uint32_t)offsetof(mrx_config, device_ids))

typedef struct mrx_renderer mrx_renderer;

/* Buffers of mrx_buffer(): the tensors Manager exports,
 * /root/reference/src/mgr.cpp:547-665 and ExportID /root/reference/src/sim.hpp:19-29. */
enum {
    MRX_BUF_RGB = 0,            /* u8  [views,H,W,4]  (Raytracer: [views,res,res,4]) */
    MRX_BUF_DEPTH = 1,          /* f32 [views,H,W,1]  (Raytracer: [views,res,res])   */
    MRX_BUF_SEGMASK = 2,        /* i32 [views,res,res], Raytracer only               */
    MRX_BUF_INSTANCE_POSITION = 3, /* f32 [instances,3]                              */
    MRX_BUF_INSTANCE_ROTATION = 4, /* f32 [instances,4]  w,x,y,z                     */
    MRX_BUF_CAMERA_POSITION = 5,   /* f32 [cameras,3]                                */
    MRX_BUF_CAMERA_ROTATION = 6,   /* f32 [cameras,4]                                */
    MRX_BUF_VISIBILITY = 7,     /* i32 [views,H,W], needs MRX_FLAG_VISIBILITY_IDS    */
    MRX_BUF_INSTANCE_SCALE = 8, /* f32 [instances,3]                                 */
    /* i32 [instances], mutable: the ObjectID column of the renderables
     * (/root/reference/src/sim.cpp:152-156).  A negative value hides the instance
     * from the next step on (cleanupRenderableEntity, src/sim.inl:10-16), writing
     * the id back shows it again (makeEntityRenderable, src/sim.inl:5-8).  Only
     * the sign is interpreted: the geometry an instance draws is bound when the
     * renderer is created, triangle slots / visibility ids stay where they are, and
     * the segmask shows the id of the bound object (label and geometry always agree:
     * writing a different non-negative id changes neither) -- until
     * mrx_refresh_objects() re-binds the rows to the ids the column holds. */
    MRX_BUF_INSTANCE_OBJECT = 9,
    MRX_NUM_BUFFERS = 10
};

enum { MRX_DTYPE_U8 = 0, MRX_DTYPE_I32 = 1, MRX_DTYPE_F32 = 2 };

typedef struct {
    uint32_t num_worlds, num_views, num_instances;
    uint32_t num_objects, num_triangles, num_materials, num_textures;
    uint32_t max_world_triangles;   /* most triangles any one world draws  */
    uint32_t storage_fast, storage_slow; /* pixels per row / rows per view */
    int32_t device_id;
    int32_t kernel_variant;
    uint64_t bytes_per_step;        /* algorithmic HBM bytes of one render */
    int32_t render_path;            /* 0 = tiled raster kernels, 1 = BVH ray-trace path */
    uint32_t bvh_nodes;             /* 8-wide BLAS nodes of all objects     */
    uint32_t bvh_depth;             /* deepest BLAS                         */
    uint32_t max_world_instances;   /* most instances any one world holds   */
    /* -- ABI 3 (mrx_info writes none of these: see mrx_info_sized) -- */
    uint32_t num_shards;            /* devices the renderer spans (1 unless device_ids) */
} mrx_info_t;
#define MRX_INFO_V2_SIZE ((size_t)offsetof(mrx_info_t, num_shards))

/* -- lifetime: replaces Manager::Manager / ~Manager (mgr.cpp:505-527).
 *    Like the reference constructor, mrx_create renders the first frame. */
int mrx_create(const mrx_config *cfg, mrx_renderer **out);
void mrx_destroy(mrx_renderer *r);

/* -- per-frame: mrx_step == Manager::step (mgr.cpp:529-546); mrx_render is
 *    its render half (the north star's Manager::render()).  Both enqueue on
 *    the renderer's stream and return without waiting. */
int mrx_step(mrx_renderer *r);
int mrx_render(mrx_renderer *r);
int mrx_sync(mrx_renderer *r);

/* -- tensor export: replaces Manager::*Tensor()/ *CudaPtr() (mgr.cpp:547-665).
 *    Returns the device pointer (owned by the renderer, valid until
 *    mrx_destroy) or NULL on error. */
void *mrx_buffer(mrx_renderer *r, int which, int64_t dims[4], int *ndim,
                 int *dtype, int *device);

/* -- debug readback: waits for the stream, then copies the first `bytes`
 *    bytes of a buffer to host memory (what /root/reference/src/dump.cpp:53-70
 *    does with cudaMemcpy). */
int mrx_copy_to_host(mrx_renderer *r, int which, void *dst, uint64_t bytes);

/* -- re-binding (makeEntityRenderable at run time, /root/reference/src/sim.inl:5-8): reads
 *    the ObjectID column back and binds every row whose id is non-negative and differs from
 *    the object it draws to that object -- a spare row (max_instances_per_world) gets its
 *    geometry this way, an existing row swaps it.  Rows holding a negative id stay bound to
 *    what they drew (hidden).  World-local triangle indices (visibility ids) are renumbered in
 *    row order; the kernel and its launch shape are chosen again for the new triangle counts.
 *    Waits for the stream; pose tensors and outputs keep their addresses. */
int mrx_refresh_objects(mrx_renderer *r);

/* -- shards (device_ids): mrx_num_shards = devices the renderer spans; mrx_shard returns shard i
 *    as a renderer of its own -- valid with every function here until the parent is destroyed
 *    (never destroy a shard) -- whose tensors hold its world range; mrx_buffer_shard is
 *    mrx_buffer(mrx_shard(r, i), ...).  On a renderer of several shards mrx_step / mrx_render /
 *    mrx_sync / mrx_refresh_objects / mrx_time_renders act on all of them, mrx_info adds up,
 *    and mrx_buffer / mrx_copy_to_host / mrx_set_stream want a shard (MRX_E_UNSUPPORTED). */
/*    Host side: mrx_step has the launches enqueued by one thread per device and returns when all are queued
 *    (MRX_SHARD_THREADS=0: by the calling thread, one after the other).  MRX_SHARD_ASYNC=1 in the environment of
 *    mrx_create: mrx_step only posts the render to those threads and returns; every other entry point here, called
 *    on the renderer or on one of its shards, waits for them to have enqueued everything posted. */
int mrx_num_shards(mrx_renderer *r);
mrx_renderer *mrx_shard(mrx_renderer *r, int shard);
void *mrx_buffer_shard(mrx_renderer *r, int shard, int which, int64_t dims[4], int *ndim,
                       int *dtype, int *device);
/*    first world of shard i in the renderer's world order (shard i owns worlds
 *    [mrx_shard_first_world(i), mrx_shard_first_world(i + 1)); i = num_shards gives num_worlds) */
int64_t mrx_shard_first_world(mrx_renderer *r, int shard);
/*    the split itself (host only, no renderer needed): first world of shard `shard` of
 *    `num_shards` over `num_worlds` worlds; shard = num_shards gives num_worlds */
int64_t mrx_shard_split(uint32_t num_worlds, uint32_t shard, uint32_t num_shards);

/* -- mrx_info writes the first MRX_INFO_V2_SIZE bytes of mrx_info_t -- the struct as ABI 2 declared
 *    it, whatever the caller was compiled against; mrx_info_sized(r, &info, sizeof info) writes
 *    min(size, sizeof(mrx_info_t)) bytes, so a caller gets exactly the fields it knows (ABI 4).
 *    The struct only ever grows at its end. */
int mrx_info(mrx_renderer *r, mrx_info_t *out);
int mrx_info_sized(mrx_renderer *r, void *out, size_t size);
void *mrx_stream(mrx_renderer *r);
/* -- stream: later launches are enqueued on `stream` (a hipStream_t; NULL = the
 *    device's null stream).  Work already enqueued on the old stream is waited
 *    for first, so the switch never reorders two renders.  A caller that
 *    writes the pose tensors from its own stream (e.g. a non-default torch
 *    stream) passes that stream here and needs no host synchronisation between
 *    the write and mrx_step(): both are ordered on the one stream. */
int mrx_set_stream(mrx_renderer *r, void *stream);

/* -- measurement: enqueue `steps` back-to-back mrx_render launches between
 *    two HIP events on the renderer's stream; *ms_total = elapsed device ms.
 *    Synchronises the stream before returning. */
int mrx_time_renders(mrx_renderer *r, int steps, float *ms_total);
/*    Host side of the same loop: `steps` mrx_step calls back to back with no synchronisation in
 *    between; *us_per_step = host wall time until the last call returned, per call (what the
 *    calling thread pays to have one step enqueued on every device).  Synchronises afterwards. */
int mrx_time_steps_host(mrx_renderer *r, int steps, double *us_per_step);
/*    mrx_mark(r, 0 | 1) records HIP event 0 / 1 on the renderer's stream;
 *    mrx_elapsed_ms waits for event 1 and returns event1 - event0.  They let a
 *    caller bracket its own timed region of mrx_step calls with device time. */
/*    Diagnostic: with MRX_DEBUG_STAMPS=1 in the environment at mrx_create, the
 *    kernels record per-wave timestamps; mrx_debug_stamps copies them out
 *    ([workgroups][4 waves][8] u64, 100 MHz ticks). Returns the count copied. */
int64_t mrx_debug_stamps(mrx_renderer *r, uint64_t *dst, int64_t capacity);
int mrx_mark(mrx_renderer *r, int which);
int mrx_elapsed_ms(mrx_renderer *r, float *ms);
/*    Output placement (outputs of 256 MiB and more: mrx_create times at most two
 *    candidate allocations and keeps the faster): writes the us per
 *    render of the candidates timed, in order (up to `capacity`), and of the one kept;
 *    returns how many were timed -- 0 when the layout needed no search. */
int mrx_placement(mrx_renderer *r, float *cand_us, int capacity, float *kept_us);

/* -- loader cross-check (host copies of what was uploaded) */
int mrx_copy_triangles(mrx_renderer *r, float *tri_pos /*[T][9]*/,
                       float *tri_uv /*[T][6]*/, int32_t *tri_mat /*[T]*/,
                       int32_t *obj_first /*[O]*/, int32_t *obj_count /*[O]*/);

/* -- host-only asset readers (no device needed); free results with mrx_free */
int mrx_load_obj(const char *path, float **tri_pos, float **tri_uv,
                 uint32_t *num_tris);
/*    The `o` / `g` blocks (with faces) of an OBJ file: writes up to `capacity`
 *    first-triangle indices and returns the block count, or a negative MRX_E_*.
 *    mrx_create makes ONE object of an asset file, as the reference does
 *    (importFromDisk(..., one_object_per_asset = true), /root/reference/src/mgr.cpp:301-303;
 *    objects[i] <-> asset path i, :340-345) -- the blocks are its meshes; with
 *    MRX_OBJ_SPLIT_BLOCKS=1 in the environment every block becomes an object of its own. */
int mrx_obj_objects(const char *path, uint32_t *first_tri, uint32_t capacity);
int mrx_decode_png(const char *path, uint8_t **rgba, uint32_t *width,
                   uint32_t *height);
/*    Texture files as mrx_create reads them: .ktx2 (BC7 or RGBA8 base level,
 *    decoded to RGBA8 on the host -- the reference's "ktx2" handler,
 *    /root/reference/src/mgr.cpp:199-212,297-298) or PNG. */
int mrx_decode_texture(const char *path, uint8_t **rgba, uint32_t *width,
                       uint32_t *height);
/*    BC7 blocks (16 bytes each) -> RGBA8, 16 pixels per block, row-major in the block. */
int mrx_decode_bc7(const uint8_t *blocks, uint32_t num_blocks, uint8_t *rgba /*[num_blocks][16][4]*/);
/*    What the OBJ reader makes of a file's material statements, as JSON text:
 *    {"num_tris":N,"tri_mtl":[...],"names":[...],"libs":[...],
 *     "materials":[{"name":..,"kd":[r,g,b],"map_kd":..},...]} (materials = every
 *    newmtl of the file's mtllibs).  Returns the length, or a negative MRX_E_*. */
int64_t mrx_describe_obj_materials(const char *path, char *json, uint64_t capacity);
/*    Builds the bottom-level BVH of one triangle soup exactly as mrx_create does
 *    for an object (the counterpart of AssetProcessor::makeBVHData,
 *    /root/reference/src/mgr.cpp:472-473) and checks its invariants: every
 *    triangle sits in exactly one leaf, every child box contains all that hangs
 *    below it, leaves hold at most 16 triangles, the traversal stack bound
 *    holds.  Returns MRX_OK or MRX_E_INVALID; counts are 0 for soups small
 *    enough to need no hierarchy. */
int mrx_blas_check(const float *tri_pos /*[T][9]*/, uint32_t num_tris, uint32_t *num_nodes,
                   uint32_t *depth, uint32_t *num_leaves);
void mrx_free(void *p);
/*    The dispatch rules that depend on the size of the device, as pure functions (host only): the triangles
 *    per world from which the default dispatch takes the BVH path for a batch of `num_views` views of
 *    width x height pixels on a device of `num_cus` compute units (`base` = the general threshold, 0 = the
 *    built-in 129), and the workgroups of the raster group kernel from which a launch counts as filling the
 *    chip.  mrx_create reads the CU count from the device (hipDeviceAttributeMultiprocessorCount). */
uint32_t mrx_dispatch_min_tris(uint32_t base, uint32_t num_views, int textured, uint32_t width, uint32_t height,
                               uint32_t num_cus);
uint32_t mrx_group_fill(uint32_t num_cus);
/*    ... and whether the default dispatch gives a Raytracer-mode batch of small worlds (<= 64 triangles in <= 64
 *    rows) to the BVH path (its flat kernel) rather than to the raster kernels: views of >= 16 tiles, >= 192 tiles per
 *    compute unit in the batch (BASELINE configs[4]); 1 / 0. */
int mrx_dispatch_flat(int raytracer, uint32_t num_views, uint32_t width, uint32_t height, uint32_t max_world_triangles,
                      uint32_t max_world_instances, uint32_t num_cus);

int mrx_device_count(void);
int mrx_abi_version(void);
const char *mrx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MRX_H */
