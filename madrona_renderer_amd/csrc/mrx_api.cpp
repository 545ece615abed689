// C-ABI implementation (include/mrx.h): scene ingestion, world assembly,
// device state and the per-frame launch.  HIP only -- no CPU rendering path.
#include "../../include/mrx.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <map>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#include "assets.hpp"
#include "bvh.hpp"
#include "raster.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define MRX_HIP(call)                                                          \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess)                                                  \
            return fail(MRX_E_HIP, std::string(#call) + ": " +                 \
                                       hipGetErrorString(e_));                 \
    } while (0)

// Build-defined constants of the rendering spec.
constexpr double kVfovDeg = 90.0;      // /root/reference/src/sim.cpp:170
constexpr float kRasterZNear = 0.001f; // /root/reference/src/sim.cpp:170
constexpr float kRtZNear = 0.1f;       // /root/reference/src/mgr.cpp:477
constexpr float kRtZFar = 1000.f;      // /root/reference/src/mgr.cpp:478
constexpr double kLightDir[3] = { 1.0, -1.0, -0.05 };  // mgr.cpp:357
constexpr float kAmbient = 0.25f;
constexpr float kDiffuse = 0.75f;

template <typename T>
struct DevBuf {
    T *ptr = nullptr;
    size_t count = 0;
    bool owned = true;
    void *base = nullptr;                       // what hipMalloc returned (ptr may sit inside it)
    // diagnostic (MRX_OUT_ALLOC=vmm): the block comes from the virtual-memory API instead --
    // one physical handle mapped at a reserved address
    hipMemGenericAllocationHandle_t vmmHandle {};
    size_t vmmSize = 0;
    hipError_t alloc(size_t n, size_t tailBytes = 0)     // tail: room for slices (view) behind the data
    {
        count = n;
        owned = true;
        const hipError_t e = hipMalloc(&base, (n ? n : 1) * sizeof(T) + tailBytes);
        ptr = e == hipSuccess ? static_cast<T *>(base) : nullptr;
        return e;
    }
    hipError_t allocVmm(size_t n, size_t tailBytes, int device)
    {
        count = n;
        owned = true;
        hipMemAllocationProp prop {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        if (e != hipSuccess)
            return e;
        if (const char *dbg = std::getenv("MRX_OUT_VMM_GRAN_MB"))
            gran = std::max<size_t>(gran, (size_t)std::atoll(dbg) << 20);
        const size_t bytes = ((n ? n : 1) * sizeof(T) + tailBytes + gran - 1) / gran * gran;
        e = hipMemCreate(&vmmHandle, bytes, &prop, 0);
        if (e != hipSuccess)
            return e;
        e = hipMemAddressReserve(&base, bytes, gran, nullptr, 0);
        if (e == hipSuccess)
            e = hipMemMap(base, bytes, 0, vmmHandle, 0);
        if (e == hipSuccess) {
            hipMemAccessDesc acc {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(base, bytes, &acc, 1);
        }
        if (e != hipSuccess) {
            (void)hipMemRelease(vmmHandle);
            base = nullptr;
            return e;
        }
        vmmSize = bytes;
        ptr = static_cast<T *>(base);
        return hipSuccess;
    }
    void view(void *base, size_t n)             // a slice of another allocation (not owned)
    {
        ptr = static_cast<T *>(base);
        this->base = nullptr;
        count = n;
        owned = false;
    }
    hipError_t upload(const std::vector<T> &h)
    {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty())
            return e;
        return hipMemcpy(ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    // upload again after the host table changed (mrx_refresh_objects): in place when the size fits
    hipError_t reupload(const std::vector<T> &h)
    {
        if (ptr && owned && h.size() <= count && !vmmSize) {
            if (h.empty())
                return hipSuccess;
            return hipMemcpy(ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
        }
        release();
        return upload(h);
    }
    void release()
    {
        if (base && owned && vmmSize) {
            (void)hipMemUnmap(base, vmmSize);
            (void)hipMemRelease(vmmHandle);
            (void)hipMemAddressFree(base, vmmSize);
            vmmSize = 0;
        } else if (base && owned) {
            (void)hipFree(base);
        }
        ptr = nullptr;
        base = nullptr;
    }
};

// The output tensors of a renderer normally live in ONE allocation, rgb first.  A lane
// stores the same pixels of every tensor back to back, and when two tensors
// lie a multiple of 512 KiB apart those stores collide in the memory system
// (measured: a 64 MiB + 64 MiB render takes 24.1 us at distance = 0 mod 512 KiB
// and 22.5 us at 256 KiB mod 512 KiB, profiles/r01_placement.txt).  Separate
// hipMalloc blocks are 2 MiB-aligned -- the bad case -- and their physical
// distance is anybody's guess; inside one allocation the distance is ours:
// depth starts at phase 256 KiB of the 512 KiB period, ids at phase 64 KiB.
constexpr size_t kOutPeriod = 512u << 10;
size_t outPhase(const char *env, size_t dflt)
{
    if (const char *dbg = std::getenv(env))
        return (size_t)std::atoll(dbg) << 10;
    return dflt;
}
// Which kind of candidate the placement search starts with: tensors of 64 MiB
// were reliably fast in one block, tensors of 256 MiB and 1 GiB one block each.
bool firstKindIsOneBlock(size_t px)
{
    if (const char *dbg = std::getenv("MRX_OUT_KIND"))      // "one" | "split": diagnostic override
        return dbg[0] == 'o';
    return px * 4 < (128ull << 20);
}

hipError_t allocOutputs(size_t px, bool wantIds, bool oneAllocation, DevBuf<uint32_t> &rgb,
                        DevBuf<float> &depth, DevBuf<int32_t> &ids)
{
    const size_t depthPhase = outPhase("MRX_OUT_SKEW_DEPTH_KB", 256u << 10);
    const size_t idsPhase = outPhase("MRX_OUT_SKEW_IDS_KB", 64u << 10);
    if (!oneAllocation) {
        // One allocation per tensor, made back to back, the phases applied
        // inside the (2 MiB-aligned) blocks.  Whether that yields the intended
        // distance is up to the allocator -- this is the other kind of
        // candidate of the placement search: half-GiB outputs were only ever
        // fast this way, 128 MiB ones reliably only in one allocation.
        hipError_t e = rgb.alloc(px);
        size_t gapBytes = 0;
        if (const char *dbg = std::getenv("MRX_OUT_GAP_MB"))    // diagnostic: a held allocation between the tensors
            gapBytes = (size_t)std::atoll(dbg) << 20;
        static std::vector<DevBuf<uint8_t>> heldGaps;            // (only ever filled by the diagnostic knob)
        auto gap = [&]() {
            DevBuf<uint8_t> sp;
            if (gapBytes && sp.alloc(gapBytes) == hipSuccess)
                heldGaps.push_back(sp);
        };
        if (e == hipSuccess) {
            gap();
            e = depth.alloc(px, depthPhase);
            depth.ptr = reinterpret_cast<float *>(static_cast<char *>(depth.base) + depthPhase);
        }
        if (e == hipSuccess && wantIds) {
            gap();
            e = ids.alloc(px, idsPhase);
            ids.ptr = reinterpret_cast<int32_t *>(static_cast<char *>(ids.base) + idsPhase);
        }
        if (e != hipSuccess) {
            rgb.release(); depth.release(); ids.release();
        }
        return e;
    }
    const size_t tb = (px * 4 + kOutPeriod - 1) / kOutPeriod * kOutPeriod;
    const size_t depthOff = tb + depthPhase;
    const size_t idsOff = depthOff + tb + kOutPeriod - (depthOff % kOutPeriod) + idsPhase;
    const size_t total = (wantIds ? idsOff : depthOff) + px * 4;
    const char *how = std::getenv("MRX_OUT_ALLOC");
    int dev = 0;
    (void)hipGetDevice(&dev);
    const hipError_t e = (how && how[0] == 'v') ? rgb.allocVmm(px, total - px * 4, dev) : rgb.alloc(px, total - px * 4);
    if (e != hipSuccess)
        return e;
    char *base = static_cast<char *>(rgb.base);
    depth.view(base + depthOff, px);
    if (wantIds)
        ids.view(base + idsOff, px);
    return hipSuccess;
}

}  // namespace

struct ShardWorker;
static void stopShardWorkers(mrx_renderer *r);

struct mrx_renderer {
    int device = 0;
    hipStream_t stream = nullptr;
    int32_t mode = MRX_MODE_RASTERIZER;
    uint32_t flags = 0;
    int32_t variant = 0;
    uint64_t stepCount = 0;
    mrx_info_t info {};
    mrx::RasterParams params {};
    // host copies kept for mrx_copy_triangles
    std::vector<mrx::ObjTri> hostTris;
    std::vector<int32_t> objFirst, objCount;
    // device state
    DevBuf<mrx::ObjTri> tris;
    DevBuf<mrx::TriMat> triMats;
    DevBuf<mrx::TexDesc> textures;
    DevBuf<uint32_t> texels;
    DevBuf<mrx::WorldTri> viewTris;
    DevBuf<uint32_t> viewTriCount;
    // the pose tensors are slices of one block, the geometry tables of another, at offsets that follow
    // from the counts (raster.hpp poseLayout / geomMatsOffset): the group kernel's fast prologue derives
    // their addresses from two base pointers that reach it preloaded in SGPRs
    DevBuf<uint8_t> poseBlock, geomBlock;
    DevBuf<float> instPos, instRot, instScale, camPos, camRot;
    DevBuf<int32_t> instObj;
    DevBuf<uint32_t> rgb;
    DevBuf<float> depth;
    DevBuf<int32_t> ids;
    DevBuf<unsigned long long> stamps;
    // XCD phase feedback (raster.hip): a host-mapped word workgroup 0 reports its XCC id to
    uint32_t *xccHost = nullptr, *xccDev = nullptr;
    int32_t xcdPhaseForced = -1;                // MRX_XCD_PHASE: 0 / 1 override (tests)
    uint32_t launches = 0;
    // BVH path: BLAS (built at load) and the tables the per-step TLAS reads
    DevBuf<mrx::BvhNode> bvhNodes;
    DevBuf<uint32_t> bvhLeafTris, worldInstStart, viewWorld, instKBase;
    DevBuf<mrx::ObjInfo> objInfo;
    bool useBvh = false;
    // host copies the geometry binding is (re)built from (bindGeometry): the object each
    // instance row is bound to, the world -> row and view -> world tables, the BLAS set
    std::vector<int32_t> boundObj;
    std::vector<uint32_t> worldInstStartHost, viewWorldHost, worldCams;
    std::vector<mrx::TriMat> triMatsHost;
    mrx::BlasSet blas;
    uint32_t bvhMinTris = mrx::kBvhMinTris;
    uint32_t bvhGroupTilesGeneral = 1;          // tiles of a view per workgroup as bvhTileKernel wants them (buildScene)
    // single-process multi-device (mrx_config.device_ids): a renderer that only fans out to
    // one sub-renderer per device, each owning a contiguous range of the worlds
    std::vector<mrx_renderer *> shards;
    std::vector<uint32_t> shardFirstWorld;      // [shards + 1]
    // one persistent host thread per device but the first (whose shards the calling thread
    // launches): mrx_step posts the launch to all of them and returns when every device has its
    // kernel enqueued -- the host cost of a step is the slowest enqueue, not their sum
    std::vector<ShardWorker *> workers;
    std::vector<mrx_renderer *> ownShards;      // (with workers) the shards the calling thread launches
    uint32_t workerSeq = 0;
    bool shardAsync = false;                    // MRX_SHARD_ASYNC=1 (startShardWorkers)
    uint32_t rendersPosted = 0;
    mrx_renderer *parent = nullptr;             // of a shard: the renderer it belongs to
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // what choosePlacement measured (mrx_placement): us per render of every candidate
    // output allocation it timed, in order, and of the one it kept
    std::vector<float> placementUs;
    float placementKeptUs = 0.0f;

    hipError_t launch()
    {
        // the parity workgroup 0 reported some launches ago (the word is written by the
        // device without synchronisation: any value it ever held is a valid prediction)
        if (xcdPhaseForced >= 0)
            params.xcdPhase = (uint32_t)xcdPhaseForced;
        else if (xccHost)
            params.xcdPhase = *(volatile uint32_t *)xccHost & 1u;
        // (the report is a write over PCIe: asked for in the first launches and then in
        // every 32nd -- the phase of a queue changes rarely)
        ++launches;
        params.xccReport = (xccDev && (launches <= 4 || (launches & 31u) == 0)) ? xccDev : nullptr;
        if (useBvh)
            return mrx::launchBvh(params, stream);
        return mrx::launchRaster(params, info.max_world_triangles, variant, stream);
    }

    ~mrx_renderer()
    {
        stopShardWorkers(this);
        for (mrx_renderer *sh : shards)
            delete sh;
        if (!shards.empty())
            return;
        (void)hipSetDevice(device);
        tris.release(); triMats.release(); textures.release(); texels.release();
        viewTris.release(); viewTriCount.release();
        instPos.release(); instRot.release(); instScale.release();
        camPos.release(); camRot.release(); instObj.release();
        poseBlock.release(); geomBlock.release();
        rgb.release(); depth.release(); ids.release(); stamps.release();
        if (xccHost) (void)hipHostFree(xccHost);
        bvhNodes.release(); bvhLeafTris.release(); worldInstStart.release();
        viewWorld.release(); instKBase.release(); objInfo.release();
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};


// ---- shard workers (single-process multi-device) ----------------------------
// A kernel launch costs the host 3 - 4 us (MI355X_MICROARCH.md), so eight launches issued one
// after the other from the calling thread cost more than the 13 us a 2048-world shard renders
// in: the one-Manager form every reference caller uses (/root/reference/scripts/test.py:112-130,
// one ctor, one step()) would be host-bound at 8 devices.  Each shard but the first therefore
// has a host thread of its own, created by mrx_create, bound to its device once, parked on a
// sequence word.  mrx_step stores the next sequence number into every worker's word, launches
// shard 0 itself and waits until every worker has acknowledged -- when it returns, every
// device has the kernel in its queue (a caller's next operation on those streams is ordered
// behind it, exactly as in the serial form), and the host paid for one launch plus a cache-line
// handshake.  Workers spin on their word for a while after a command (a simulation loop calls
// step() every few tens of microseconds) and then sleep in a futex; MRX_SHARD_SPIN_US sets how
// long (default 200), MRX_SHARD_THREADS=0 restores the serial form (startShardWorkers).
enum : int { kCmdNone = 0, kCmdRender, kCmdSync, kCmdTimed, kCmdExit };

struct ShardWorker {
    std::vector<mrx_renderer *> shards;         // the shards of ONE device, launched in order
    std::thread th;
    // written by the master, read by the worker
    alignas(64) std::atomic<uint32_t> posted { 0 };
    std::atomic<int> cmd { kCmdNone };          // (ordered by `posted`; atomic because MRX_SHARD_ASYNC re-posts kCmdRender
    std::atomic<int> steps { 0 };               //  while the worker may still be looking at the previous one)
    alignas(64) std::atomic<uint32_t> sleeping { 0 };
    // written by the worker, read by the master
    alignas(64) std::atomic<uint32_t> done { 0 };
    int rc = MRX_OK;
    std::string err;
    int64_t tSeen = 0, tDone = 0;               // MRX_SHARD_TRACE: when the command was seen / finished
    alignas(64) std::atomic<uint32_t> masterSleeping { 0 };
    // MRX_SHARD_ASYNC=1: renders are counted, not acknowledged one by one -- the caller posts (renderPosted)
    // and goes on, the worker launches until it has caught up (renderDone); every other command joins first
    bool async = false;
    alignas(64) std::atomic<uint32_t> renderPosted { 0 };
    alignas(64) std::atomic<uint32_t> renderDone { 0 };
};

namespace {

inline void cpuRelax() { __builtin_ia32_pause(); }

inline long futexWait(std::atomic<uint32_t> *word, uint32_t expected)
{
    return syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAIT_PRIVATE, expected, nullptr, nullptr, 0);
}

inline long futexWake(std::atomic<uint32_t> *word)
{
    return syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

int64_t shardSpinNs()
{
    static const int64_t ns = [] {
        const char *e = std::getenv("MRX_SHARD_SPIN_US");
        return (int64_t)(e ? std::max(0, std::atoi(e)) : 200) * 1000;
    }();
    return ns;
}

bool shardTrace()
{
    static const bool on = std::getenv("MRX_SHARD_TRACE") != nullptr;
    return on;
}

// MRX_SHARD_TRACE: sums over the render commands of a renderer (ns): post -> worker saw it (worst
// worker), the worker's launch (worst), the calling thread's own launch, post -> all acknowledged
struct ShardTrace { int64_t wake = 0, launch = 0, own = 0, total = 0, n = 0; };
thread_local ShardTrace g_trace;

int64_t nowNs()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

// what the shards of one device do for a command, on whichever thread runs it (the shards of a
// group share a device; it is made current here unless the thread is bound to it for good)
int groupRun(const std::vector<mrx_renderer *> &shards, int cmd, int steps, bool bindDevice)
{
    if (shards.empty())
        return MRX_OK;
    if (bindDevice)
        MRX_HIP(hipSetDevice(shards[0]->device));
    switch (cmd) {
    case kCmdRender:
        for (mrx_renderer *sh : shards)
            MRX_HIP(sh->launch());
        return MRX_OK;
    case kCmdSync:
        for (mrx_renderer *sh : shards)
            MRX_HIP(hipStreamSynchronize(sh->stream));
        return MRX_OK;
    case kCmdTimed:
        // every shard's launches between its own two events, the shards of the device interleaved
        // step by step: the slowest shard's interval spans the device's whole job
        for (mrx_renderer *sh : shards)
            MRX_HIP(hipEventRecord(sh->ev0, sh->stream));
        for (int i = 0; i < steps; ++i)
            for (mrx_renderer *sh : shards)
                MRX_HIP(sh->launch());
        for (mrx_renderer *sh : shards)
            MRX_HIP(hipEventRecord(sh->ev1, sh->stream));
        return MRX_OK;
    default:
        return MRX_OK;
    }
}

void shardWorkerMain(ShardWorker *w)
{
    const hipError_t bound = hipSetDevice(w->shards[0]->device);
    uint32_t seen = 0;
    int64_t lastWork = nowNs();
    for (;;) {
        uint32_t seq;
        // park: spin while commands keep coming, then sleep
        for (uint32_t it = 0;; ++it) {
            seq = w->posted.load(std::memory_order_acquire);
            if (seq != seen)
                break;
            if ((it & 255u) == 255u && nowNs() - lastWork > shardSpinNs()) {
                w->sleeping.store(1, std::memory_order_seq_cst);
                if (w->posted.load(std::memory_order_seq_cst) == seen)
                    futexWait(&w->posted, seen);
                w->sleeping.store(0, std::memory_order_relaxed);
                continue;
            }
            cpuRelax();
        }
        seen = seq;
        const int cmd = w->cmd.load(std::memory_order_relaxed);
        if (shardTrace())
            w->tSeen = nowNs();
        if (w->async) {
            // counted renders: launch until caught up (a failure is kept for the next join and the
            // count still advances, so that nobody waits for a launch that will not come)
            uint32_t doneR = w->renderDone.load(std::memory_order_relaxed);
            while (doneR != w->renderPosted.load(std::memory_order_acquire)) {
                int rc = bound == hipSuccess ? groupRun(w->shards, kCmdRender, 0, false)
                                             : fail(MRX_E_HIP, std::string("hipSetDevice (shard worker): ") + hipGetErrorString(bound));
                if (rc != MRX_OK && w->rc == MRX_OK) {
                    w->err = g_err;
                    w->rc = rc;
                }
                w->renderDone.store(++doneR, std::memory_order_release);
            }
            lastWork = nowNs();
            if (cmd == kCmdRender)
                continue;                             // (nothing to acknowledge: the count is the acknowledgement)
        }
        if (cmd == kCmdExit) {
            w->done.store(seq, std::memory_order_release);
            return;
        }
        int rc = MRX_OK;
        if (bound != hipSuccess)
            rc = fail(MRX_E_HIP, std::string("hipSetDevice (shard worker): ") + hipGetErrorString(bound));
        if (rc == MRX_OK)
            rc = groupRun(w->shards, cmd, w->steps.load(std::memory_order_relaxed), false);
        if (!(w->async && w->rc != MRX_OK)) {         // (an error of a counted render waits for its join)
            w->rc = rc;
            if (rc != MRX_OK)
                w->err = g_err;
        }
        if (shardTrace())
            w->tDone = nowNs();
        w->done.store(seq, std::memory_order_seq_cst);
        if (w->masterSleeping.load(std::memory_order_seq_cst))
            futexWake(&w->done);
        lastWork = nowNs();
    }
}

// MRX_SHARD_ASYNC: wait until every device thread has enqueued every render posted so far; the
// first failure among them is reported here (and cleared)
int joinRenders(mrx_renderer *r)
{
    if (!r->shardAsync)
        return MRX_OK;
    int rc = MRX_OK;
    std::string err;
    for (ShardWorker *w : r->workers) {
        for (uint32_t it = 0; w->renderDone.load(std::memory_order_acquire) != r->rendersPosted; ++it) {
            if ((it & 1023u) == 1023u)
                std::this_thread::yield();
            else
                cpuRelax();
        }
        if (w->rc != MRX_OK) {
            if (rc == MRX_OK) {
                rc = w->rc;
                err = w->err;
            }
            w->rc = MRX_OK;
        }
    }
    return rc == MRX_OK ? MRX_OK : fail(rc, err);
}

// a renderer, or the renderer a shard handle belongs to: device threads caught up (MRX_SHARD_ASYNC)
int settle(mrx_renderer *r)
{
    if (r && r->parent)
        r = r->parent;
    if (!r || !r->shardAsync)
        return MRX_OK;
    return joinRenders(r);
}

// run `cmd` on every shard of a multi-device renderer: workers in parallel, shard 0 (and any
// shard without a worker) on the calling thread; returns the first failure
int shardsRun(mrx_renderer *r, int cmd, int steps = 0)
{
    if (r->workers.empty()) {
        // no threads: device by device from the calling thread (consecutive shards of one device as a group)
        std::vector<mrx_renderer *> group;
        for (size_t i = 0; i < r->shards.size(); ++i) {
            group.push_back(r->shards[i]);
            if (i + 1 == r->shards.size() || r->shards[i + 1]->device != r->shards[i]->device) {
                const int rc = groupRun(group, cmd, steps, true);
                if (rc != MRX_OK)
                    return rc;
                group.clear();
            }
        }
        return MRX_OK;
    }
    if (r->shardAsync) {
        if (cmd == kCmdRender) {
            // post and go on; at most 256 renders ahead of the slowest device thread
            const uint32_t n = ++r->rendersPosted;
            const uint32_t seq = ++r->workerSeq;
            for (ShardWorker *w : r->workers) {
                while (n - w->renderDone.load(std::memory_order_acquire) > 256u)
                    cpuRelax();
                w->cmd = kCmdRender;
                w->renderPosted.store(n, std::memory_order_release);
                w->posted.store(seq, std::memory_order_seq_cst);
                if (w->sleeping.load(std::memory_order_seq_cst))
                    futexWake(&w->posted);
            }
            return MRX_OK;
        }
        const int jrc = joinRenders(r);
        if (jrc != MRX_OK)
            return jrc;
    }
    const uint32_t seq = ++r->workerSeq;
    const bool trace = shardTrace() && cmd == kCmdRender;
    const int64_t tPost = trace ? nowNs() : 0;
    int64_t tOwn = 0;
    for (ShardWorker *w : r->workers) {
        w->cmd = cmd;
        w->steps = steps;
        w->posted.store(seq, std::memory_order_seq_cst);
        if (w->sleeping.load(std::memory_order_seq_cst))
            futexWake(&w->posted);
    }
    int rc = groupRun(r->ownShards, cmd, steps, true);
    if (trace)
        tOwn = nowNs();
    std::string err = rc != MRX_OK ? g_err : std::string();
    for (ShardWorker *w : r->workers) {
        // an enqueue takes microseconds: spin; a stream synchronisation may take long: sleep
        const int64_t t0 = nowNs();
        for (uint32_t it = 0; w->done.load(std::memory_order_acquire) != seq; ++it) {
            if ((it & 255u) == 255u && nowNs() - t0 > 50000) {
                w->masterSleeping.store(1, std::memory_order_seq_cst);
                const uint32_t cur = w->done.load(std::memory_order_seq_cst);
                if (cur != seq)
                    futexWait(&w->done, cur);
                w->masterSleeping.store(0, std::memory_order_relaxed);
                continue;
            }
            cpuRelax();
        }
        if (w->rc != MRX_OK && rc == MRX_OK) {
            rc = w->rc;
            err = w->err;
        }
    }
    if (trace) {
        int64_t wake = 0, launch = 0;
        for (ShardWorker *w : r->workers) {
            wake = std::max(wake, w->tSeen - tPost);
            launch = std::max(launch, w->tDone - w->tSeen);
        }
        g_trace.wake += wake;
        g_trace.launch += launch;
        g_trace.own += tOwn - tPost;
        g_trace.total += nowNs() - tPost;
        g_trace.n++;
    }
    if (rc != MRX_OK)
        return fail(rc, err);
    return MRX_OK;
}

void startShardWorkers(mrx_renderer *r)
{
    // One host thread per DEVICE: shards that share a device share its queues and (by default)
    // its null stream, where concurrent launches only queue up behind the runtime's stream lock
    // (profiles/r04_multidev_host.txt: eight threads on one null stream 48 us per step against
    // 32 us from one thread).  The first device's shards stay with the calling thread.
    // MRX_SHARD_THREADS=0: no threads at all; =2: a thread per shard whatever its device (the
    // one-GPU rehearsal of a node, with a stream per shard).
    int mode = 1;
    if (const char *e = std::getenv("MRX_SHARD_THREADS"))
        mode = std::atoi(e);
    if (mode <= 0)
        return;
    std::vector<std::vector<mrx_renderer *>> groups;
    for (mrx_renderer *sh : r->shards) {
        bool placed = false;
        for (auto &g : groups)
            if (mode == 1 && g[0]->device == sh->device) {
                g.push_back(sh);
                placed = true;
                break;
            }
        if (!placed)
            groups.push_back({ sh });
    }
    if (groups.size() <= 1)
        return;
    // MRX_SHARD_ASYNC=1 (opt-in): mrx_step only POSTS the render -- every device, the first included, has a
    // thread -- and returns; the threads enqueue behind its back, and every other mrx_* call on the renderer or
    // on one of its shards joins them first (mrx_sync, mrx_buffer ..., the next non-render command).  The host
    // pays a cache line per device for a step, not a launch.  What the caller gives up: an operation it enqueues
    // ITSELF on a shard's stream right after mrx_step (a torch kernel on tensors fetched earlier) is no longer
    // ordered behind the render unless an mrx_* call came in between.
    r->shardAsync = (r->flags & MRX_FLAG_SHARD_ASYNC) != 0;
    if (const char *e = std::getenv("MRX_SHARD_ASYNC"))
        r->shardAsync = std::atoi(e) != 0;
    const size_t firstWorker = r->shardAsync ? 0 : 1;
    if (!r->shardAsync)
        r->ownShards = groups[0];
    for (size_t g = firstWorker; g < groups.size(); ++g) {
        ShardWorker *w = new ShardWorker();
        w->shards = groups[g];
        w->async = r->shardAsync;
        w->th = std::thread(shardWorkerMain, w);
        r->workers.push_back(w);
    }
}

}  // namespace

static void stopShardWorkers(mrx_renderer *r)
{
    if (r->workers.empty())
        return;
    const uint32_t seq = ++r->workerSeq;
    for (ShardWorker *w : r->workers) {
        w->cmd = kCmdExit;
        w->posted.store(seq, std::memory_order_seq_cst);
        futexWake(&w->posted);
    }
    for (ShardWorker *w : r->workers) {
        w->th.join();
        delete w;
    }
    r->workers.clear();
}

namespace {

// S6b support.  An object may hold several shells (edge-connected components);
// each is judged on its own: closed and consistently wound <=> every directed
// edge of the component occurs once and so does its reverse (vertices welded
// by exact position).  Per triangle: orient = the sign of its component's
// enclosed volume (+1 outward winding, -1 inward; 0 when the component is
// open / inconsistent, which leaves its triangles two-sided) and the
// component's bounding box, padded by 1e-4 of its extent + 1e-6.
void shellOrientation(const mrx::ObjTri *tris, uint32_t n, mrx::TriMat *mats)
{
    std::map<std::array<uint32_t, 3>, uint32_t> ids;
    std::vector<std::array<uint32_t, 3>> tv(n);
    for (uint32_t t = 0; t < n; ++t)
        for (int c = 0; c < 3; ++c) {
            std::array<uint32_t, 3> key;
            std::memcpy(key.data(), tris[t].p + 3 * c, 12);
            for (uint32_t &w : key)
                if (w == 0x80000000u) w = 0;            // -0 == +0
            auto it = ids.find(key);
            if (it == ids.end())
                it = ids.emplace(key, (uint32_t)ids.size()).first;
            tv[t][c] = it->second;
        }
    // components: triangles joined through a shared (undirected) edge
    std::vector<uint32_t> parent(n);
    for (uint32_t t = 0; t < n; ++t)
        parent[t] = t;
    auto find = [&](uint32_t x) {
        while (parent[x] != x)
            x = parent[x] = parent[parent[x]];
        return x;
    };
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> firstOnEdge;
    for (uint32_t t = 0; t < n; ++t)
        for (int c = 0; c < 3; ++c) {
            uint32_t a = tv[t][c], b = tv[t][(c + 1) % 3];
            if (a > b) std::swap(a, b);
            auto ins = firstOnEdge.emplace(std::make_pair(a, b), t);
            if (!ins.second)
                parent[find(t)] = find(ins.first->second);
        }
    struct Shell {
        std::map<std::pair<uint32_t, uint32_t>, int> edges;
        double vol = 0.0;
        bool bad = false;
        uint32_t count = 0;
        float lo[3], hi[3];
    };
    std::map<uint32_t, Shell> shells;
    for (uint32_t t = 0; t < n; ++t) {
        Shell &sh = shells[find(t)];
        const uint32_t *v = tv[t].data();
        if (v[0] == v[1] || v[1] == v[2] || v[2] == v[0])
            sh.bad = true;
        for (int c = 0; c < 3; ++c)
            if (++sh.edges[{ v[c], v[(c + 1) % 3] }] > 1)
                sh.bad = true;
        const float *a = tris[t].p, *b = a + 3, *c3 = a + 6;
        sh.vol += (double)a[0] * ((double)b[1] * c3[2] - (double)b[2] * c3[1]) +
                  (double)a[1] * ((double)b[2] * c3[0] - (double)b[0] * c3[2]) +
                  (double)a[2] * ((double)b[0] * c3[1] - (double)b[1] * c3[0]);
        for (int c = 0; c < 3; ++c)
            for (int ax = 0; ax < 3; ++ax) {
                const float x = tris[t].p[3 * c + ax];
                if ((sh.count == 0 && c == 0) || x < sh.lo[ax]) sh.lo[ax] = x;
                if ((sh.count == 0 && c == 0) || x > sh.hi[ax]) sh.hi[ax] = x;
            }
        sh.count++;
    }
    for (auto &kv : shells) {
        Shell &sh = kv.second;
        if (sh.count < 4)
            sh.bad = true;
        for (const auto &e : sh.edges)
            if (!sh.bad && sh.edges.find({ e.first.second, e.first.first }) == sh.edges.end())
                sh.bad = true;
        for (int ax = 0; ax < 3; ++ax) {
            const float pad = 1e-4f * (sh.hi[ax] - sh.lo[ax]) + 1e-6f;
            sh.lo[ax] -= pad;
            sh.hi[ax] += pad;
        }
    }
    for (uint32_t t = 0; t < n; ++t) {
        const Shell &sh = shells[find(t)];
        mats[t].orient = sh.bad ? 0.0f : sh.vol > 0.0 ? 1.0f : sh.vol < 0.0 ? -1.0f : 0.0f;
        std::memcpy(mats[t].bbMin, sh.lo, 12);
        std::memcpy(mats[t].bbMax, sh.hi, 12);
    }
}

// One-tile views whose worlds fit one pass: two views per workgroup, their TLASes built side by side (bvh.hip,
// MULTI) -- phase I is as long as the latency of its pose loads whether one wave works in it or two, and the
// texel loads at the end of the first view's tile overlap the traversal of the second.  As long as the groups
// still fill the chip (two resident workgroups per CU) and two TLASes leave room for two workgroups in a CU's
// LDS: untextured worlds of up to 104 instances, textured ones of up to 64 (profiles/r03_bvh_group_views.txt).
// Called whenever the bound geometry changes (textured or not decides the kernel's tables).
int chooseBvhGroups(mrx_renderer &r)
{
    using namespace mrx;
    RasterParams &p = r.params;
    const uint32_t nviews = p.numViews, maxWorldInst = r.info.max_world_instances;
    p.bvhGroupViews = 1;
    // worlds that fit one 64-lane set-up take the flat kernel (bvh.hip): 64x64 tiles, one view per workgroup
    p.bvhFlat = (p.bvhTile == 0 && r.info.max_world_triangles <= 64u && maxWorldInst <= 64u) ? 1u : 0u;
    if (const char *dbg = std::getenv("MRX_BVH_FLAT"))
        p.bvhFlat = p.bvhFlat && std::atoi(dbg) != 0;
    p.bvhGroupTiles = r.bvhGroupTilesGeneral;
    if (p.bvhFlat) {
        // Tiles of a view per workgroup for the flat kernel (three workgroups per CU: 46 KB of LDS each).  A group pays
        // one set-up of the view (S) and then a tile time (T) per tile, and the launch runs in ceil(groups / resident)
        // generations: the run length -- the view's tiles divided by a power of two, at most 16 -- that minimises
        // generations x (S + tiles x T).  S / T = 2.5 / 3.0 from the stamps (profiles/r04_flat_stamps.txt); measured
        // (profiles/r04_flat_zbufs_ab.txt): 1024 x 128^2 picks 2 (39.0 us against 41.4 at 4 and 48.3 at 1), 4096 x 128^2
        // picks 4 (115 against 137 / 166), configs[4] (4096 views of 16 tiles) picks 8.
        const uint32_t tpvF = ((p.nfast + 63u) / 64u) * ((p.nslow + 63u) / 64u);
        const uint32_t residentF = 3u * std::max(p.numCUs, 1u);
        uint32_t best = 1;
        double bestCost = 1e300;
        for (uint32_t g = std::min(tpvF, 16u); g >= 1u; g = (g + 1u) / 2u) {
            const uint64_t groups = (uint64_t)nviews * ((tpvF + g - 1) / g);
            const double cost = (double)((groups + residentF - 1) / residentF) * (2.5 + 3.0 * (double)g);
            if (cost < bestCost) {
                bestCost = cost;
                best = g;
            }
            if (g == 1u)
                break;
        }
        p.bvhGroupTiles = best;
        if (const char *dbg = std::getenv("MRX_BVH_GROUP_TILES"))
            p.bvhGroupTiles = (uint32_t)std::max(1, std::min((int)tpvF, std::atoi(dbg)));
        return MRX_OK;
    }
    const uint32_t resident = 2u * std::max(p.numCUs, 1u);
    const uint32_t tpv = ((p.nfast + 63u) / 64u) * ((p.nslow + 63u) / 64u);
    if (tpv == 1 && p.bvhTile == 0 && maxWorldInst <= p.bvhPassInst) {
        const bool tex = p.anyTextured != 0;
        // (by view count, profiles/r03_bvh_group_views.txt: two views per workgroup win where the pairs fit the chip at
        // once and fill at least three quarters of it -- 768 ... 1024 views on 256 CUs -- and again from four
        // generations of single views on; in between, two generations of single views beat one and a bit of pairs)
        // Between one and two views per resident workgroup the launch is as many workgroups as the chip holds, pairs
        // on the first nviews - resident of them and single views on the others (bit 16; bvh.hip): one generation
        // for 513 ... 1024 views.
        // (worlds of more than 64 instances: only from three quarters of the chip in pairs on -- 576 ... 700 views of
        // 101-instance worlds lose 4 % to two generations of single views, profiles/r03_bvh_group_views.txt)
        const bool mixed = nviews > resident && nviews <= 2u * resident &&
                           (maxWorldInst <= 64u || 2u * nviews >= 3u * resident);
        const bool pairsPay = mixed || nviews >= 4u * resident;
        // (textured: pairs as long as two TLAS blocks leave room for 256 records)
        p.bvhGroupViews = pairsPay && bvhLdsBytes(p.bvhPassInst, tex, p.bvhClassify != 0, 2u, 256u) <= 80u * 1024u ? 2u : 1u;
        if (p.bvhGroupViews == 2u && mixed && !std::getenv("MRX_BVH_NO_MIXED"))
            p.bvhGroupViews |= 0x10000u;
        if (const char *dbg = std::getenv("MRX_BVH_GROUP_VIEWS")) {
            const int want = std::atoi(dbg);
            if (want == 1 || want == 2 || want == 4 || want == 8)
                p.bvhGroupViews = (uint32_t)want | (want == 2 ? (p.bvhGroupViews & 0x10000u) : 0u);
        }
    }
    // Launches of one-tile views whose workgroups all run at once (more than one per CU, at most two): the
    // workgroup dispatched second to a CU runs the first part of its work at wave priority 1 (bvh.hip: the arbiters
    // serve the older workgroup first, and the launch ends with the younger half).  MRX_BVH_PRIO: 0 off, 1 / 2 / 3
    // the modes measured there.  (Views of several tiles, one generation of workgroups: 256 x 128^2 gains 4 - 6 %,
    // 128 x 128^2 of 100 cubes loses 7 %, C5 / 8 loses 1 % -- tiles of a view differ too much in what they hold;
    // left alone.  profiles/r03_bvh_priority.txt)
    int prio = 2;
    if (const char *dbg = std::getenv("MRX_BVH_PRIO"))
        prio = std::max(0, std::min(3, std::atoi(dbg)));
    const uint32_t tw = p.bvhTile == 2 ? 32 : 64, th = p.bvhTile == 0 ? 64 : 32;
    const uint32_t tiles = ((p.nfast + tw - 1) / tw) * ((p.nslow + th - 1) / th);
    const uint32_t groupTiles = std::max(1u, std::min(p.bvhGroupTiles, tiles));
    const uint32_t gv = p.bvhGroupViews & 0xFFFFu;
    const uint32_t wgs = (p.bvhGroupViews & 0x10000u) ? resident
                         : gv > 1 ? (nviews + gv - 1) / gv : nviews * ((tiles + groupTiles - 1) / groupTiles);
    // textured worlds: as many 48-byte shading records per round as fit the half CU beside the TLAS block(s), in 32s
    // (profiles/r04_bvh_textured.txt; MRX_BVH_TEX_CAP overrides)
    {
        uint32_t cap = 64;
        while (cap + 32u <= 1008u &&
               bvhLdsBytes(p.bvhPassInst, true, p.bvhClassify != 0, gv, cap + 32u) <= 80u * 1024u)
            cap += 32u;
        if (const char *dbg = std::getenv("MRX_BVH_TEX_CAP"))
            cap = (uint32_t)std::max(64, std::min((int)cap, std::atoi(dbg)));
        p.bvhTexCap = cap;
    }
    p.bvhGroupViews |= std::min(resident / 2u, 4095u) << 20;      // (the first index of a CU's second workgroup)
    if (prio && p.bvhTile == 0 && tiles == 1 && wgs <= resident && wgs > resident / 2u)
        p.bvhGroupViews |= (uint32_t)prio << 17;
    return MRX_OK;
}

// Everything that follows from which object each instance row is bound to (r.boundObj):
// the world-local triangle numbering, the per-view draw lists of the raster kernels, the
// per-instance BLAS records of the BVH path, which kernel renders, the uniform-world fast
// paths.  Runs at creation and again from mrx_refresh_objects; the pose tensors, the
// outputs and the object pool are untouched.
int bindGeometry(mrx_renderer &r)
{
    using namespace mrx;
    const std::vector<int32_t> &bound = r.boundObj;
    const std::vector<uint32_t> &worldInstStart = r.worldInstStartHost, &viewWorld = r.viewWorldHost;
    const uint32_t numWorlds = (uint32_t)worldInstStart.size() - 1u;
    auto drawn = [&](int32_t o) { return o >= 0 && (size_t)o < r.objFirst.size(); };
    std::vector<uint32_t> worldTriStart(1, 0), instKBase(bound.size(), 0);
    std::vector<WorldTri> worldTris;   // per world; expanded per view below
    uint32_t maxWorldTris = 0;
    for (uint32_t w = 0; w < numWorlds; ++w) {
        for (uint32_t row = worldInstStart[w]; row < worldInstStart[w + 1]; ++row) {
            // first world-local triangle index of the instance (the visibility id space)
            instKBase[row] = (uint32_t)worldTris.size() - worldTriStart[w];
            if (drawn(bound[row])) {
                const uint32_t f = (uint32_t)r.objFirst[bound[row]];
                const uint32_t n = (uint32_t)r.objCount[bound[row]];
                for (uint32_t t = 0; t < n; ++t)
                    worldTris.push_back(WorldTri { row, f + t });
            }
        }
        worldTriStart.push_back((uint32_t)worldTris.size());
        maxWorldTris = std::max(maxWorldTris, worldTriStart[w + 1] - worldTriStart[w]);
    }
    // Which kernel renders: the group kernel (raster.hip) wins while a world fits its
    // 128-slot instantiation, the BVH path (bvh.hip) as soon as the 256-slot one
    // would be needed (measured at 1024 worlds x 64x64: 122 triangles 20.9 against
    // 25.7 us, 134 triangles 28.6 against 27.0, 482 triangles 83 against 31.4 --
    // profiles/r02_bvh_crossover.txt); it serves both render modes.
    // MRX_BVH_MIN_TRIS moves the threshold; kernel_variant 2 / 3 force the BVH /
    // the raster kernels.
    //   Small batches of one-tile views cross over earlier: up to two workgroups per CU the BVH kernel's
    // launch costs what its slowest workgroup costs, while the raster kernels' 128-slot shape pays for its slots
    // (profiles/r03_bvh_threshold.txt: 512 views of 74 triangles 10.7 against 12.3 us, 1024 views of 98 triangles
    // 18.0 against 18.6; 2048 views and more cross at 122 ... 129).
    uint32_t minTris = r.bvhMinTris;
    {
        bool anyTex = false;
        for (const WorldTri &wt : worldTris)
            if (r.triMatsHost[wt.tri].tex >= 0) {
                anyTex = true;
                break;
            }
        const uint32_t nviews = (uint32_t)viewWorld.size();
        // (textured worlds: the same up to 2.5 views per CU -- 512 views of 74 triangles 13.0 against 15.5 us -- and at
        // 4 per CU only from ~115 triangles on, left at the general threshold; raster.hpp bvhDispatchMinTris)
        if (!std::getenv("MRX_BVH_MIN_TRIS"))
            minTris = bvhDispatchMinTris(minTris, nviews, anyTex, r.params.nfast, r.params.nslow, r.params.numCUs);
    }
    // (Raytracer-mode batches of many large views of small worlds: the BVH path's flat kernel -- what BASELINE configs[4]
    // names -- is ahead of the raster kernel there; raster.hpp bvhDispatchFlat.  MRX_BVH_FLAT=0 / MRX_BVH_MIN_TRIS: off)
    bool rtFlat = bvhDispatchFlat(r.mode == MRX_MODE_RAYTRACER, (uint32_t)viewWorld.size(), r.params.nfast, r.params.nslow,
                                  maxWorldTris, r.info.max_world_instances, r.params.numCUs) &&
                  !std::getenv("MRX_BVH_MIN_TRIS") && r.params.bvhTile == 0;
    if (const char *dbg = std::getenv("MRX_BVH_FLAT"))
        rtFlat = rtFlat && std::atoi(dbg) != 0;
    r.useBvh = r.variant == kVariantBvh || (r.variant == kVariantDefault && (maxWorldTris >= minTris || rtFlat));
    if (r.useBvh && maxWorldTris > kBvhMaxWorldTris)
        return fail(MRX_E_UNSUPPORTED, "more than 2M triangles in one world");
    // per-view draw lists with a fixed stride (one load level in the kernel);
    // the BVH path walks the world's instances instead and needs none
    const uint32_t stride = maxWorldTris ? maxWorldTris : 1u;
    {
        const size_t nv = r.useBvh ? 0 : viewWorld.size();
        std::vector<WorldTri> viewTris(nv * stride, WorldTri { 0, 0 });
        std::vector<uint32_t> viewTriCount(nv);
        for (size_t v = 0; v < nv; ++v) {
            const uint32_t w = viewWorld[v];
            const uint32_t b = worldTriStart[w], n = worldTriStart[w + 1] - b;
            viewTriCount[v] = n;
            std::memcpy(viewTris.data() + v * stride, worldTris.data() + b, n * sizeof(WorldTri));
        }
        MRX_HIP(r.viewTris.reupload(viewTris));
        MRX_HIP(r.viewTriCount.reupload(viewTriCount));
    }
    // per instance: range, root and box of its bound object, so the per-step TLAS build has
    // no load that depends on another
    std::vector<ObjInfo> instInfo(bound.size());
    for (size_t i = 0; i < bound.size(); ++i) {
        ObjInfo info {};
        info.root = -1;
        if (drawn(bound[i]))
            info = r.blas.objects[bound[i]];
        instInfo[i] = info;
    }
    if (instInfo.empty())
        instInfo.emplace_back();                      // idle lanes read row 0
    MRX_HIP(r.objInfo.reupload(instInfo));
    MRX_HIP(r.instKBase.reupload(instKBase));

    RasterParams &p = r.params;
    p.anyTextured = 0;
    for (const mrx::WorldTri &wt : worldTris)
        if (r.triMatsHost[wt.tri].tex >= 0) {
            p.anyTextured = 1;
            break;
        }
    p.viewTris = r.viewTris.ptr;
    p.viewTriCount = r.viewTriCount.ptr;
    p.viewTriStride = stride;
    p.instInfo = r.objInfo.ptr;
    p.instKBase = r.instKBase.ptr;
    // uniform worlds -> arithmetic draw list (raster.hpp): every world the same 1..4 bound
    // objects in the same order, the same camera count
    p.uniInstances = 0;
    p.uniCamsPerWorld = 0;
    if (numWorlds > 0) {
        const uint32_t n0 = worldInstStart[1] - worldInstStart[0], c0 = r.worldCams[0];
        bool uni = n0 >= 1 && n0 <= 4 && c0 >= 1;
        for (uint32_t w = 0; uni && w < numWorlds; ++w) {
            uni = worldInstStart[w + 1] - worldInstStart[w] == n0 && r.worldCams[w] == c0;
            for (uint32_t i = 0; uni && i < n0; ++i) {
                const int32_t oa = bound[worldInstStart[w] + i], ob = bound[worldInstStart[0] + i];
                uni = oa == ob && drawn(oa);
            }
        }
        if (uni) {
            p.uniInstances = n0;
            p.uniCamsPerWorld = c0;
            uint32_t acc = 0;
            for (uint32_t i = 0; i < 4; ++i) {
                p.uniPrefix[i] = acc;
                p.uniFirstTri[i] = 0;
                if (i < n0) {
                    const int32_t o = bound[worldInstStart[0] + i];
                    p.uniFirstTri[i] = (uint32_t)r.objFirst[o];
                    acc += (uint32_t)r.objCount[o];
                }
            }
            p.uniPrefix[4] = acc;
        }
    }
    r.info.max_world_triangles = maxWorldTris;
    r.info.render_path = r.useBvh ? 1 : 0;
    return chooseBvhGroups(r);
}

int buildScene(const mrx_config &cfg, mrx_renderer &r)
{
    using namespace mrx;
    // ---- objects: disk assets in path order, then one object per raw mesh
    //      (/root/reference/src/mgr.cpp:267-270, scripts/test.py:7-10)
    std::vector<ObjTri> &tris = r.hostTris;
    const uint32_t maxInstancesPerWorld = cfg.max_instances_per_world;
    auto appendObject = [&](const float *pos, const float *uv, uint32_t n, int32_t mat) {
        r.objFirst.push_back((int32_t)tris.size());
        r.objCount.push_back((int32_t)n);
        for (uint32_t t = 0; t < n; ++t) {
            ObjTri o {};
            std::memcpy(o.p, pos + 9 * (size_t)t, sizeof(o.p));
            std::memcpy(o.uv, uv + 6 * (size_t)t, sizeof(o.uv));
            o.mat = mat;
            tris.push_back(o);
        }
    };
    // File materials (MTL) are appended after the API materials and textures:
    // an asset with mat_id >= 0 is drawn with that API material; with mat_id
    // -1 each face keeps the material its OBJ names through mtllib / usemtl
    // (Kd colour, map_Kd PNG texture), or the default when it names none.
    struct FileMat { float kd[3]; std::string mapKd; };
    std::vector<FileMat> fileMats;
    std::vector<std::string> fileTexPaths;
    for (uint32_t a = 0; a < cfg.num_asset_paths; ++a) {
        TriSoup soup;
        std::string err;
        if (!loadOBJ(cfg.asset_paths[a], soup, err))
            return fail(MRX_E_ASSET, "Failed to load render assets: " + err);
        // The reference carries a per-asset material id through its API
        // (bindings.cpp:26-36) but applies it only inside a disabled block
        // (mgr.cpp:339-349); the block's evident intent is implemented here.
        int32_t mat = -1;
        if (cfg.mat_assignments && a < cfg.num_mat_assignments)
            mat = cfg.mat_assignments[a];
        // One object per asset FILE: the reference imports with one_object_per_asset
        // (the `true` of importFromDisk, mgr.cpp:301-303) and addresses objects[i] by asset
        // path i (mgr.cpp:340-345), so the `o` / `g` blocks of a file are the meshes of one
        // object and later assets / raw meshes keep the ids the reference gives them.
        // MRX_OBJ_SPLIT_BLOCKS=1 opts into one object per block instead.
        const char *split = std::getenv("MRX_OBJ_SPLIT_BLOCKS");
        if (split && split[0] == '1') {
            for (size_t o = 0; o < soup.objStart.size(); ++o) {
                const uint32_t t0 = soup.objStart[o];
                const uint32_t t1 = o + 1 < soup.objStart.size() ? soup.objStart[o + 1] : soup.numTris();
                appendObject(soup.pos.data() + 9 * (size_t)t0, soup.uv.data() + 6 * (size_t)t0, t1 - t0, mat);
            }
        } else {
            appendObject(soup.pos.data(), soup.uv.data(), soup.numTris(), mat);
        }
        if (mat < 0 && !soup.mtlNames.empty()) {
            std::vector<MtlMaterial> lib;
            for (const std::string &ml : soup.mtlLibs) {
                std::string merr;
                (void)loadMTL(ml, lib, merr);       // a missing library leaves faces on the default
            }
            std::vector<int32_t> nameToMat(soup.mtlNames.size(), -1);
            for (size_t n = 0; n < soup.mtlNames.size(); ++n)
                for (const MtlMaterial &mm : lib)
                    if (mm.name == soup.mtlNames[n]) {
                        nameToMat[n] = (int32_t)(cfg.num_materials + fileMats.size());
                        fileMats.push_back(FileMat { { mm.kd[0], mm.kd[1], mm.kd[2] }, mm.mapKd });
                        break;
                    }
            const size_t first = tris.size() - soup.numTris();
            for (uint32_t t = 0; t < soup.numTris(); ++t)
                if (soup.triMtl[t] >= 0)
                    tris[first + t].mat = nameToMat[soup.triMtl[t]];
        }
    }
    const mrx_geometry &g = cfg.geo;
    for (uint32_t m = 0; m < g.num_meshes; ++m) {   // mgr.cpp:214-272
        const uint32_t v0 = g.mesh_vertex_offsets[m];
        const uint32_t v1 = m + 1 < g.num_meshes ? g.mesh_vertex_offsets[m + 1] : g.num_vertices;
        const uint32_t i0 = g.mesh_index_offsets[m];
        const uint32_t i1 = m + 1 < g.num_meshes ? g.mesh_index_offsets[m + 1] : g.num_indices;
        if (v1 < v0 || v1 > g.num_vertices || i1 < i0 || i1 > g.num_indices)
            return fail(MRX_E_INVALID, "raw mesh offsets out of range");
        const uint32_t nt = (i1 - i0) / 3;
        std::vector<float> pos(9 * (size_t)nt), uv(6 * (size_t)nt);
        for (uint32_t t = 0; t < nt; ++t)
            for (int c = 0; c < 3; ++c) {
                const uint32_t vi = g.indices[i0 + 3 * t + c];
                if (vi >= v1 - v0)
                    return fail(MRX_E_INVALID, "raw mesh index out of range");
                std::memcpy(&pos[9 * (size_t)t + 3 * c], g.vertices + 3 * (size_t)(v0 + vi), 12);
                std::memcpy(&uv[6 * (size_t)t + 2 * c], g.uvs + 2 * (size_t)(v0 + vi), 8);
            }
        appendObject(pos.data(), uv.data(), nt, g.mesh_materials[m]);
    }

    // ---- textures and materials (mgr.cpp:316-337; no disk textures or
    //      materials exist, so the index shift there is zero)
    std::vector<TexDesc> texDescs;
    std::vector<uint32_t> texels;
    for (uint32_t t = 0; t < cfg.num_textures; ++t) {
        Image img;
        std::string err;
        if (!decodeTexture(cfg.texture_paths[t], img, err))
            return fail(MRX_E_ASSET, "Failed to load texture: " + err);
        // (the BVH kernels' shading records hold width | height << 16)
        if (img.width == 0 || img.height == 0 || img.width > 65535u || img.height > 65535u)
            return fail(MRX_E_UNSUPPORTED, std::string("texture ") + cfg.texture_paths[t] +
                                               ": sides of 1 ... 65535 texels are supported");
        TexDesc d {};
        d.offset = (uint32_t)texels.size();
        d.width = img.width;
        d.height = img.height;
        texDescs.push_back(d);
        const size_t n = (size_t)img.width * img.height;
        texels.resize(texels.size() + n);
        std::memcpy(texels.data() + d.offset, img.rgba.data(), n * 4);
    }
    // combined material table: API materials, then file materials whose map_Kd
    // textures (PNG or KTX2; anything unreadable means untextured) follow the API textures
    std::vector<mrx_material> allMats(cfg.materials, cfg.materials + cfg.num_materials);
    for (const FileMat &fm : fileMats) {
        mrx_material m {};
        m.color[0] = fm.kd[0]; m.color[1] = fm.kd[1]; m.color[2] = fm.kd[2]; m.color[3] = 1.0f;
        m.texture_idx = -1;
        m.roughness = 0.8f;
        m.metalness = 0.2f;
        if (!fm.mapKd.empty()) {
            int32_t found = -1;
            for (size_t i = 0; i < fileTexPaths.size(); ++i)
                if (fileTexPaths[i] == fm.mapKd)
                    found = (int32_t)(cfg.num_textures + i);
            if (found < 0) {
                Image img;
                std::string terr;
                if (decodeTexture(fm.mapKd, img, terr) && img.width >= 1 && img.height >= 1 && img.width <= 65535u &&
                    img.height <= 65535u) {
                    TexDesc d {};
                    d.offset = (uint32_t)texels.size();
                    d.width = img.width;
                    d.height = img.height;
                    texDescs.push_back(d);
                    const size_t n = (size_t)img.width * img.height;
                    texels.resize(texels.size() + n);
                    std::memcpy(texels.data() + d.offset, img.rgba.data(), n * 4);
                    found = (int32_t)(cfg.num_textures + fileTexPaths.size());
                    fileTexPaths.push_back(fm.mapKd);
                }
            }
            m.texture_idx = found;
        }
        allMats.push_back(m);
    }
    // resolve every object triangle's material once: mesh material if it
    // exists, else the default colour (white, untextured); a texture index
    // with no texture behind it means untextured
    std::vector<TriMat> triMats(tris.size());
    for (size_t t = 0; t < tris.size(); ++t) {
        TriMat &tm = triMats[t];
        tm.color[0] = tm.color[1] = tm.color[2] = tm.color[3] = 1.0f;
        tm.tex = -1;
        const int32_t m = tris[t].mat;
        if (m >= 0 && (size_t)m < allMats.size()) {
            std::memcpy(tm.color, allMats[m].color, 16);
            tm.tex = allMats[m].texture_idx;
        }
        if (tm.tex < 0 || (uint32_t)tm.tex >= (uint32_t)texDescs.size())
            tm.tex = -1;
        if (tm.tex >= 0) {
            // the texture's descriptor rides along (BVH path: no second load level per pixel)
            tm.texDesc[0] = (int32_t)texDescs[tm.tex].offset;
            tm.texDesc[1] = (int32_t)texDescs[tm.tex].width;
            tm.texDesc[2] = (int32_t)texDescs[tm.tex].height;
        }
    }
    // the object a triangle belongs to rides in the alpha slot (segmask label, raster.hpp)
    for (size_t o = 0; o < r.objFirst.size(); ++o)
        for (int32_t t = 0; t < r.objCount[o]; ++t) {
            const int32_t id = (int32_t)o;
            std::memcpy(&triMats[(size_t)r.objFirst[o] + t].color[3], &id, 4);
        }
    // S6b: orientation and padded bounding box of every triangle's shell
    for (size_t o = 0; o < r.objFirst.size(); ++o)
        shellOrientation(tris.data() + r.objFirst[o], (uint32_t)r.objCount[o],
                         triMats.data() + r.objFirst[o]);

    // ---- world assembly: per-world copies of the table rows, world-major
    //      (/root/reference/src/sim.cpp:143-175).  With max_instances_per_world a world owns
    //      that many rows at least (the reference sizes its renderer by maxInstancesPerWorld,
    //      src/mgr.cpp:378-388, and creates renderables at run time, src/sim.inl:5-8): the
    //      spare rows start hidden and unbound (ObjectID -1, identity pose) and are bound to
    //      an object by writing its id and calling mrx_refresh_objects.
    std::vector<float> instPos, instRot, instScale, camPos, camRot;
    std::vector<int32_t> instObj;
    std::vector<uint32_t> &worldInstStart = r.worldInstStartHost, &viewWorld = r.viewWorldHost;
    worldInstStart.assign(1, 0u);
    viewWorld.clear();
    r.worldCams.clear();
    uint32_t maxWorldInst = 0;
    for (uint32_t w = 0; w < cfg.num_worlds; ++w) {
        const mrx_world_init &wi = cfg.worlds[w];
        if ((uint64_t)wi.instances_offset + wi.num_instances > cfg.num_instances ||
            (uint64_t)wi.cameras_offset + wi.num_cameras > cfg.num_cameras)
            return fail(MRX_E_INVALID, "world " + std::to_string(w) +
                                           " addresses rows outside the tables");
        const uint32_t rows = std::max(wi.num_instances, maxInstancesPerWorld);
        for (uint32_t i = 0; i < rows; ++i) {
            if (i < wi.num_instances) {
                const mrx_instance &in = cfg.instances[wi.instances_offset + i];
                instPos.insert(instPos.end(), in.position, in.position + 3);
                instRot.insert(instRot.end(), in.rotation, in.rotation + 4);
                instScale.insert(instScale.end(), in.scale, in.scale + 3);
                instObj.push_back(in.object_id);
            } else {
                const float zero[3] = { 0.f, 0.f, 0.f }, ident[4] = { 1.f, 0.f, 0.f, 0.f }, one[3] = { 1.f, 1.f, 1.f };
                instPos.insert(instPos.end(), zero, zero + 3);
                instRot.insert(instRot.end(), ident, ident + 4);
                instScale.insert(instScale.end(), one, one + 3);
                instObj.push_back(-1);
            }
        }
        worldInstStart.push_back((uint32_t)instObj.size());
        maxWorldInst = std::max(maxWorldInst, rows);
        r.worldCams.push_back(wi.num_cameras);
        for (uint32_t c = 0; c < wi.num_cameras; ++c) {
            const mrx_camera &cam = cfg.cameras[wi.cameras_offset + c];
            camPos.insert(camPos.end(), cam.position, cam.position + 3);
            camRot.insert(camRot.end(), cam.rotation, cam.rotation + 4);
            viewWorld.push_back(w);
        }
    }
    // the object each row draws: bound here, re-bound by mrx_refresh_objects
    r.boundObj = instObj;

    // ---- upload what never changes shape
    {
        // geometry block: ObjTri[pool], then TriMat[pool] at the next 256-byte boundary
        const uint32_t pool = (uint32_t)tris.size();
        const size_t matsOff = geomMatsOffset(pool);
        MRX_HIP(r.geomBlock.alloc(matsOff + (size_t)pool * sizeof(TriMat) + 256));
        r.tris.view(r.geomBlock.ptr, pool);
        r.triMats.view(r.geomBlock.ptr + matsOff, pool);
        if (pool) {
            MRX_HIP(hipMemcpy(r.tris.ptr, tris.data(), (size_t)pool * sizeof(ObjTri), hipMemcpyHostToDevice));
            MRX_HIP(hipMemcpy(r.triMats.ptr, triMats.data(), (size_t)pool * sizeof(TriMat), hipMemcpyHostToDevice));
        }
    }
    MRX_HIP(r.textures.upload(texDescs));
    MRX_HIP(r.texels.upload(texels));
    r.triMatsHost = triMats;
    r.bvhMinTris = kBvhMinTris;
    if (const char *dbg = std::getenv("MRX_BVH_MIN_TRIS"))
        r.bvhMinTris = (uint32_t)std::max(0, std::atoi(dbg));
    // BLAS per object (the reference builds them at load too: mgr.cpp:472-473)
    buildBlas(tris.data(), r.objFirst, r.objCount, r.blas);
    const BlasSet &blas = r.blas;
    if (blas.leafTris.size() >= (1u << kBvhLeafStartBits))
        return fail(MRX_E_UNSUPPORTED, "too many triangles in BLAS leaves");
    // (the kernel's traversal stack is kBvhStackCap entries of LDS per wave: a deeper tree
    // would write past it -- bvh.cpp rebuilds such trees balanced, which bounds the depth
    // by about log8 of the triangle count; refuse what still does not fit)
    if (1 + 7 * blas.maxDepth > kBvhStackCap)
        return fail(MRX_E_UNSUPPORTED, "a BLAS is too deep for the traversal stack (" +
                                           std::to_string(blas.maxDepth) + " levels)");
    MRX_HIP(r.bvhNodes.upload(blas.nodes));
    MRX_HIP(r.bvhLeafTris.upload(blas.leafTris));
    MRX_HIP(r.worldInstStart.upload(worldInstStart));
    MRX_HIP(r.viewWorld.upload(viewWorld));
    {
        // pose block (the exported, user-mutable tensors are slices of it)
        const uint32_t nv = (uint32_t)viewWorld.size(), ni = (uint32_t)instObj.size();
        const PoseLayout lay = poseLayout(nv, ni);
        MRX_HIP(r.poseBlock.alloc((size_t)lay.total + 256));
        MRX_HIP(hipMemset(r.poseBlock.ptr, 0, (size_t)lay.total + 256));
        uint8_t *b = r.poseBlock.ptr;
        r.camRot.view(b + lay.camRot, camRot.size());
        r.camPos.view(b + lay.camPos, camPos.size());
        r.instRot.view(b + lay.instRot, instRot.size());
        r.instPos.view(b + lay.instPos, instPos.size());
        r.instScale.view(b + lay.instScale, instScale.size());
        r.instObj.view(b + lay.instObj, instObj.size());
        auto up = [](void *dst, const void *src, size_t bytes) {
            return bytes ? hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
        };
        MRX_HIP(up(r.camRot.ptr, camRot.data(), camRot.size() * 4));
        MRX_HIP(up(r.camPos.ptr, camPos.data(), camPos.size() * 4));
        MRX_HIP(up(r.instRot.ptr, instRot.data(), instRot.size() * 4));
        MRX_HIP(up(r.instPos.ptr, instPos.data(), instPos.size() * 4));
        MRX_HIP(up(r.instScale.ptr, instScale.data(), instScale.size() * 4));
        MRX_HIP(up(r.instObj.ptr, instObj.data(), instObj.size() * 4));
    }

    const bool rt = cfg.render_mode == MRX_MODE_RAYTRACER;
    const uint32_t W = cfg.view_width;
    // Raytracer output is square, res = view width (mgr.cpp:130,443)
    const uint32_t H = rt ? cfg.view_width : cfg.view_height;
    const uint32_t nviews = (uint32_t)viewWorld.size();
    // storage [view][slow][fast]: raster fast = image x; Raytracer fast =
    // image y (callers read it as [x][y]: scripts/test.py:160, dump.cpp:9-21)
    const uint32_t nfast = rt ? H : W, nslow = rt ? W : H;
    const size_t px = (size_t)nviews * nfast * nslow;
    {
        // diagnostic: a block allocated ahead of the outputs (MRX_OUT_PRE_MB), freed
        // again right after them unless MRX_OUT_PRE_HOLD=1
        DevBuf<uint8_t> pre;
        if (const char *dbg = std::getenv("MRX_OUT_PRE_MB"))
            if (pre.alloc((size_t)std::atoll(dbg) << 20) != hipSuccess)
                (void)hipGetLastError();
        MRX_HIP(allocOutputs(px, rt || (cfg.flags & MRX_FLAG_VISIBILITY_IDS), firstKindIsOneBlock(px), r.rgb, r.depth, r.ids));
        const char *hold = std::getenv("MRX_OUT_PRE_HOLD");
        if (!(hold && hold[0] == '1'))
            pre.release();
    }
    const bool wantIds = rt || (cfg.flags & MRX_FLAG_VISIBILITY_IDS);

    RasterParams &p = r.params;
    p.tris = r.tris.ptr;
    p.triMats = r.triMats.ptr;
    p.textures = r.textures.ptr;
    p.texels = r.texels.ptr;
    p.poseBlock = reinterpret_cast<const char *>(r.poseBlock.ptr);
    p.geomBlock = reinterpret_cast<const char *>(r.geomBlock.ptr);
    p.numInstances = (uint32_t)instObj.size();
    p.poolTris = (uint32_t)tris.size();
    p.instPos = r.instPos.ptr;
    p.instRot = r.instRot.ptr;
    p.instScale = r.instScale.ptr;
    p.instObj = r.instObj.ptr;
    p.camPos = r.camPos.ptr;
    p.camRot = r.camRot.ptr;
    p.rgb = r.rgb.ptr;
    p.depth = r.depth.ptr;
    p.ids = wantIds ? r.ids.ptr : nullptr;
    p.numViews = nviews;
    {
        int cus = 0;
        MRX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, r.device));
        p.numCUs = (uint32_t)std::max(cus, 1);
        if (const char *dbg = std::getenv("MRX_FAKE_CUS"))      // tests: the launch shapes of a smaller device
            p.numCUs = (uint32_t)std::max(1, std::atoi(dbg));
    }
    p.nfast = nfast;
    p.nslow = nslow;
    p.tilesFast = (nfast + 63) / 64;
    p.tilesSlow = (nslow + 63) / 64;
    // S5: pixel -> ray constants, double math rounded once to float
    const float th = (float)std::tan(kVfovDeg * M_PI / 360.0);
    const double asp = (double)W / (double)H;
    p.sx = (float)(2.0 * (double)th * asp / (double)W);
    p.ox = (float)((1.0 / (double)W - 1.0) * (double)th * asp);
    p.sz = (float)(-2.0 * (double)th / (double)H);
    p.oz = (float)((1.0 - 1.0 / (double)H) * (double)th);
    p.invNear = 1.0f / (rt ? kRtZNear : kRasterZNear);
    p.invFar = rt ? 1.0f / kRtZFar : 0.0f;
    // S6b pad: a surface point nearer than znear along +Y that still projects
    // into the image lies within znear * |longest ray| of the eye
    p.s6bPad = (float)((double)(rt ? kRtZNear : kRasterZNear) *
                       std::sqrt(1.0 + (double)th * (double)th * (1.0 + asp * asp)) * 1.001);
    const double ln = std::sqrt(kLightDir[0] * kLightDir[0] + kLightDir[1] * kLightDir[1] +
                                kLightDir[2] * kLightDir[2]);
    for (int c = 0; c < 3; ++c)
        p.toLight[c] = (float)(-kLightDir[c] / ln);
    p.ambient = kAmbient;
    p.diffuse = kDiffuse;
    p.transposed = rt ? 1 : 0;
    // Raytracer ids are the segmask unless the caller asked for visibility ids
    p.idsAreSegmask = (rt && !(cfg.flags & MRX_FLAG_VISIBILITY_IDS)) ? 1 : 0;
    p.debugSkip = 0;
    if (const char *dbg = std::getenv("MRX_DEBUG_SKIP"))
        p.debugSkip = (uint32_t)std::atoi(dbg);
    p.debugStamps = nullptr;
    if (const char *dbg = std::getenv("MRX_DEBUG_STAMPS"))
        if (std::atoi(dbg) != 0) {
            // (x4: the BVH kernel's 32x32 tile shape launches four workgroups per 64x64 tile)
            const size_t n = ((size_t)nviews * p.tilesFast * p.tilesSlow * 4 + 64) * 4 * 8;
            MRX_HIP(r.stamps.alloc(n));
            MRX_HIP(hipMemset(r.stamps.ptr, 0, n * sizeof(unsigned long long)));
            p.debugStamps = r.stamps.ptr;
        }
    p.writeThrough = 1;
    if (const char *dbg = std::getenv("MRX_WRITE_THROUGH"))
        p.writeThrough = std::atoi(dbg) != 0;
    p.grpViews = p.grpChunkTiles = p.grpPerView = 0;
    p.grpViewsWanted = p.grpTilesWanted = 0;
    if (const char *dbg = std::getenv("MRX_GROUP_VIEWS"))
        p.grpViewsWanted = std::atoi(dbg);
    if (const char *dbg = std::getenv("MRX_GROUP_TILES"))
        p.grpTilesWanted = std::atoi(dbg);
    p.xcdRotate = 0;
    p.xcdRotateWanted = -1;
    if (const char *dbg = std::getenv("MRX_XCD_ROTATE"))
        p.xcdRotateWanted = std::atoi(dbg);
    p.xccReport = nullptr;
    p.xcdPhase = 0;
    if (hipHostMalloc((void **)&r.xccHost, 64, hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer((void **)&r.xccDev, r.xccHost, 0) == hipSuccess) {
        *r.xccHost = 0;
        p.xccReport = r.xccDev;
    } else {
        (void)hipGetLastError();                  // no feedback: the split assumes an even start
        r.xccHost = nullptr;
    }
    if (const char *dbg = std::getenv("MRX_XCD_PHASE"))
        r.xcdPhaseForced = std::atoi(dbg) & 1;
    p.xcdSkew = 0;
    p.xcdSkewWanted = -1;
    if (const char *dbg = std::getenv("MRX_XCD_SKEW"))
        p.xcdSkewWanted = std::atoi(dbg);
    p.debugSlots = 0;
    if (const char *dbg = std::getenv("MRX_DEBUG_SLOTS"))
        p.debugSlots = std::atoi(dbg);
    p.bvhNodes = r.bvhNodes.ptr;
    p.bvhLeafTris = r.bvhLeafTris.ptr;
    p.numObjects = (uint32_t)r.objFirst.size();
    p.bvhUniInst = p.bvhUniCams = 0;
    if (cfg.num_worlds > 0) {
        bool uni = true;
        const uint32_t rows0 = worldInstStart[1] - worldInstStart[0];
        for (uint32_t w = 1; uni && w < cfg.num_worlds; ++w)
            uni = worldInstStart[w + 1] - worldInstStart[w] == rows0 &&
                  cfg.worlds[w].num_cameras == cfg.worlds[0].num_cameras;
        if (uni && rows0 > 0 && cfg.worlds[0].num_cameras > 0) {
            p.bvhUniInst = rows0;
            p.bvhUniCams = cfg.worlds[0].num_cameras;
        }
    }
    p.worldInstStart = r.worldInstStart.ptr;
    p.viewWorld = r.viewWorld.ptr;
    // TLAS records of up to 128 instances stay in LDS at once (two workgroups
    // per CU); larger worlds take several passes
    // (as many as the largest world has, in eights: what the records do not take leaves room for a second TLAS block)
    p.bvhPassInst = std::min<uint32_t>(128u, std::max<uint32_t>(8u, (maxWorldInst + 7u) / 8u * 8u));
    p.bvhTile = 0;
    if (const char *dbg = std::getenv("MRX_BVH_TILE"))
        p.bvhTile = std::max(0, std::min(2, std::atoi(dbg)));
    // scenes with meshes large enough for a BLAS get the instantiation that classifies the
    // listed large triangles per strip (bvh.hip, CLS): close-ups of such meshes are where
    // long lists of large triangles come from
    p.bvhClassify = blas.nodes.empty() ? 0 : 1;
    if (const char *dbg = std::getenv("MRX_BVH_CLASSIFY"))
        p.bvhClassify = std::atoi(dbg) != 0;
    // (swept on the round's final kernel, 16 ... 4096: a plateau from 192 to 1024 -- only triangles that cover a good
    // part of the tile are worth the shared list and its barrier; profiles/r02_bvh_perf.txt)
    p.bvhSmallArea = 256;
    if (const char *dbg = std::getenv("MRX_BVH_SMALL_AREA"))
        p.bvhSmallArea = std::max(0, std::min(4096, std::atoi(dbg)));
    if (const char *dbg = std::getenv("MRX_BVH_PASS_INST"))
        p.bvhPassInst = std::min<uint32_t>(512u, std::max<uint32_t>(8u, (uint32_t)std::atoi(dbg) / 8u * 8u));
    // Views of several tiles whose world fits one TLAS pass: a workgroup renders a run of the view's
    // tiles over one TLAS build (bvh.hip).  As long a run as leaves one full generation of resident
    // workgroups (two per CU: 512 on 256 CUs) -- measured, profiles/r03_bvh_group_tiles.txt: 512 views of 256x256
    // cube+plane 155 -> 119 us at 16 tiles per group, 1024 views of 128x128 73 -> 62 us at 4, and
    // every shape loses as soon as the groups no longer fill the chip.
    p.bvhGroupTiles = 1;
    if (maxWorldInst <= p.bvhPassInst) {
        const uint32_t tw = p.bvhTile == 2 ? 32 : 64, thh = p.bvhTile == 0 ? 64 : 32;
        const uint32_t tpv = ((nfast + tw - 1) / tw) * ((nslow + thh - 1) / thh);
        uint32_t g = 1;
        while (g < 16u && g * 2u <= tpv && (uint64_t)nviews * ((tpv + 2u * g - 1) / (2u * g)) >= 2u * p.numCUs)
            g *= 2;
        p.bvhGroupTiles = std::max(1u, g);
        if (const char *dbg = std::getenv("MRX_BVH_GROUP_TILES"))
            p.bvhGroupTiles = (uint32_t)std::max(1, std::min((int)tpv, std::atoi(dbg)));
    }
    r.bvhGroupTilesGeneral = p.bvhGroupTiles;         // (the flat kernel chooses its own: chooseBvhGroups)
    mrx_info_t &inf = r.info;
    inf.num_worlds = cfg.num_worlds;
    inf.num_views = nviews;
    inf.num_instances = (uint32_t)instObj.size();
    inf.num_objects = (uint32_t)r.objFirst.size();
    inf.num_triangles = (uint32_t)tris.size();
    inf.num_materials = (uint32_t)allMats.size();
    inf.num_textures = (uint32_t)texDescs.size();
    inf.storage_fast = nfast;
    inf.storage_slow = nslow;
    inf.device_id = r.device;
    inf.kernel_variant = r.variant;
    inf.bvh_nodes = (uint32_t)blas.nodes.size();
    inf.bvh_depth = blas.maxDepth;
    inf.max_world_instances = maxWorldInst;
    inf.num_shards = 1;
    inf.bytes_per_step = (uint64_t)px * (wantIds ? 12u : 8u) +
                         44ull * inf.num_instances + 28ull * nviews;
    return bindGeometry(r);
}

}  // namespace

extern "C" {

int mrx_abi_version(void) { return MRX_ABI_VERSION; }

const char *mrx_last_error(void) { return g_err.c_str(); }

int mrx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

// Where LARGE output tensors land in HBM matters on some boxes: with tensors of
// 256 MiB and more the same launch streams its stores ~20 % faster into some
// allocations than into others, steadily for the life of the allocation
// (profiles/r02_placement.txt).  Nothing visible from user space
// predicts which -- not the layout, not the distance between the tensors, not
// the address, not the XCD phase of the queue; some boxes offer no fast
// allocation at all -- so for such outputs a few candidates are allocated one
// after another (alternately all tensors in one block and one block per
// tensor), each timed with a few renders, and the fastest kept.  Bounded: at
// most two candidates (the best so far and the current one) plus one small
// spacer are alive at any time, two are tried by default (round 3: a pure
// two-tensor fill shows the same modes from allocation to allocation --
// profiles/r03_placement.txt -- so this is a property of the platform, not
// something a longer search could fix), and nothing is tried when two more
// copies of the outputs would not fit a quarter of the free memory.  Outputs below 256 MiB (every 64x64 batch up to 8192 views) are laid
// out deterministically in one block -- depth at phase 256 KiB of the 512 KiB
// period -- and need no search (what looked like a placement lottery for them
// in round 1 was the XCD phase of the stream: raster.hip, xcdPhase).
// MRX_PLACEMENT_TRIES=1 switches the search off.
static int choosePlacement(mrx_renderer *r)
{
    const size_t px = r->rgb.count;
    const bool wantIds = r->ids.ptr != nullptr;
    const size_t bytes = px * 4 * (wantIds ? 3 : 2);
    int maxTries = (bytes >= (256ull << 20) && bytes <= (16ull << 30)) ? 2 : 1;
    if (const char *dbg = std::getenv("MRX_PLACEMENT_TRIES"))
        maxTries = std::max(1, std::min(16, std::atoi(dbg)));
    if (maxTries > 1) {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess || 2 * bytes + (256ull << 20) > freeB / 4) {
            (void)hipGetLastError();
            maxTries = 1;
        }
    }
    if (maxTries <= 1)
        return MRX_OK;
    const bool trace = std::getenv("MRX_PLACEMENT_TRACE") != nullptr;
    struct Cand { DevBuf<uint32_t> rgb; DevBuf<float> depth; DevBuf<int32_t> ids; float us = 0.0f; };
    auto bind = [&](const Cand &c) {
        r->params.rgb = c.rgb.ptr;
        r->params.depth = c.depth.ptr;
        r->params.ids = wantIds ? c.ids.ptr : nullptr;
    };
    auto freeCand = [](Cand &c) { c.rgb.release(); c.depth.release(); c.ids.release(); };
    auto launch = [&]() { return r->launch(); };
    auto timeBatch = [&](int n, float &ms) -> hipError_t {
        hipError_t e = hipEventRecord(r->ev0, r->stream);
        for (int i = 0; i < n && e == hipSuccess; ++i)
            e = launch();
        if (e == hipSuccess) e = hipEventRecord(r->ev1, r->stream);
        if (e == hipSuccess) e = hipEventSynchronize(r->ev1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r->ev0, r->ev1);
        return e;
    };
    Cand best;
    best.rgb = r->rgb; best.depth = r->depth; best.ids = r->ids;   // what buildScene allocated
    // a render takes ~0.1 - 1 ms here: batches of ~0.5 ms, ~40 ms of warm-up (clocks)
    hipError_t st = hipSuccess;                   // first error; the cleanup below always runs
    bind(best);
    float one = 0.0f;
    st = launch();
    if (st == hipSuccess) st = timeBatch(1, one);
    one = std::max(one, 1e-3f);
    const int batch = std::max(3, std::min(64, (int)(0.5f / one)));
    float scratch = 0.0f;
    if (st == hipSuccess) st = timeBatch(std::max(8, std::min(4000, (int)(40.0f / one))), scratch);
    auto measure = [&](Cand &c) -> hipError_t {
        bind(c);
        hipError_t e = launch();
        float bestMs = 1e30f, ms = 0.0f;
        for (int rep = 0; rep < 3 && e == hipSuccess; ++rep) {
            e = timeBatch(batch, ms);
            bestMs = std::min(bestMs, ms);
        }
        c.us = bestMs / (float)batch * 1000.0f;
        return e;
    };
    if (st == hipSuccess) st = measure(best);
    float tmax = best.us;
    r->placementUs.push_back(best.us);
    std::string log;
    char buf[64];
    std::snprintf(buf, sizeof buf, " %.2f", best.us);
    log += buf;
    for (int k = 1; k < maxTries && st == hipSuccess; ++k) {
        // a spacer of 2 ... 128 MiB steps the next candidate on in the address space;
        // the best candidate so far stays allocated, so the new one cannot reuse its block
        DevBuf<uint8_t> sp;
        if (sp.alloc((size_t)(2 * ((k * 37) % 64 + 1)) << 20) != hipSuccess)
            (void)hipGetLastError();
        Cand c;
        const hipError_t ae = allocOutputs(px, wantIds, ((k & 1) == 0) == firstKindIsOneBlock(px),
                                           c.rgb, c.depth, c.ids);
        sp.release();
        if (ae != hipSuccess) {
            (void)hipGetLastError();                  // out of memory: make do with what there is
            break;
        }
        st = measure(c);
        if (st != hipSuccess) {
            freeCand(c);
            break;
        }
        std::snprintf(buf, sizeof buf, " %.2f", c.us);
        log += buf;
        r->placementUs.push_back(c.us);
        tmax = std::max(tmax, c.us);
        if (c.us < best.us) {
            freeCand(best);
            best = c;
        } else {
            freeCand(c);
        }
        // the two modes lie 5 - 7 % (outputs below 256 MiB) to ~20 % apart; candidates of
        // one mode scatter by +-0.5 % (small) to +-4 % (large)
        if (best.us <= (bytes < (256ull << 20) ? 0.965f : 0.88f) * tmax)
            break;                                    // a fast placement
    }
    if (trace)
        std::fprintf(stderr, "mrx: output placement, us/render:%s -> %.2f (rgb %p depth %p)\n", log.c_str(),
                     best.us, (void *)best.rgb.ptr, (void *)best.depth.ptr);
    r->rgb = best.rgb; r->depth = best.depth; r->ids = best.ids;
    r->placementKeptUs = best.us;
    bind(best);
    if (st != hipSuccess)
        return fail(MRX_E_HIP, std::string("output placement: ") + hipGetErrorString(st));
    return MRX_OK;
}

// one renderer on one device (mrx_create proper, or one shard of a multi-device renderer)
static int createOne(const mrx_config &cfg, mrx_renderer **out)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MRX_E_NO_DEVICE,
                    "no HIP device: this library renders on MI355X only (no CPU path)");
    if (cfg.gpu_id < 0 || cfg.gpu_id >= ndev)
        return fail(MRX_E_NO_DEVICE, "gpu_id " + std::to_string(cfg.gpu_id) +
                                         " out of range (" + std::to_string(ndev) + " devices)");
    MRX_HIP(hipSetDevice(cfg.gpu_id));

    mrx_renderer *r = new mrx_renderer();
    r->device = cfg.gpu_id;
    r->stream = (hipStream_t)cfg.stream;
    r->mode = cfg.render_mode;
    r->flags = cfg.flags;
    r->variant = cfg.kernel_variant;
    int rc = buildScene(cfg, *r);
    if (rc == MRX_OK) {
        hipError_t e = hipEventCreate(&r->ev0);
        if (e == hipSuccess) e = hipEventCreate(&r->ev1);
        if (e != hipSuccess)
            rc = fail(MRX_E_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
    }
    if (rc == MRX_OK)
        rc = choosePlacement(r);
    // the reference renders the first frame inside the constructor (mgr.cpp:524)
    if (rc == MRX_OK)
        rc = mrx_step(r);
    if (rc == MRX_OK)
        rc = mrx_sync(r);
    if (rc != MRX_OK) {
        delete r;
        return rc;
    }
    *out = r;
    return MRX_OK;
}

// contiguous world range of shard i of n: sizes differ by at most one (scenes.shard_range)
static uint32_t shardFirstWorld(uint32_t numWorlds, uint32_t i, uint32_t n)
{
    const uint32_t base = numWorlds / n, rem = numWorlds % n;
    return i * base + std::min(i, rem);
}

int mrx_create(const mrx_config *cfgIn, mrx_renderer **out)
{
    if (!cfgIn || !out)
        return fail(MRX_E_INVALID, "null argument");
    *out = nullptr;
    // ABI 2 callers pass the struct without its trailing ABI 3 fields: those read as zero
    if (cfgIn->struct_size != sizeof(mrx_config) && cfgIn->struct_size != MRX_CONFIG_V2_SIZE)
        return fail(MRX_E_INVALID, "mrx_config size mismatch (ABI)");
    mrx_config full;
    std::memset(&full, 0, sizeof full);
    std::memcpy(&full, cfgIn, cfgIn->struct_size);
    full.struct_size = sizeof(mrx_config);
    const mrx_config *cfg = &full;
    if (cfg->render_mode != MRX_MODE_RASTERIZER && cfg->render_mode != MRX_MODE_RAYTRACER)
        return fail(MRX_E_INVALID, "bad render_mode");
    if (cfg->view_width == 0 || cfg->view_height == 0 || cfg->view_width > 16384 ||
        cfg->view_height > 16384)
        return fail(MRX_E_INVALID, "bad view size");
    if (cfg->num_worlds && !cfg->worlds)
        return fail(MRX_E_INVALID, "worlds is null");
    if ((cfg->num_instances && !cfg->instances) || (cfg->num_cameras && !cfg->cameras) ||
        (cfg->num_asset_paths && !cfg->asset_paths) || (cfg->num_materials && !cfg->materials) ||
        (cfg->num_textures && !cfg->texture_paths))
        return fail(MRX_E_INVALID, "a table pointer is null while its count is not zero");
    if (cfg->geo.num_meshes &&
        (!cfg->geo.vertices || !cfg->geo.uvs || !cfg->geo.indices || !cfg->geo.mesh_vertex_offsets ||
         !cfg->geo.mesh_index_offsets || !cfg->geo.mesh_materials))
        return fail(MRX_E_INVALID, "raw geometry arrays are null while num_meshes is not zero");
    if (cfg->kernel_variant < 0 || cfg->kernel_variant >= mrx::kNumVariants)
        return fail(MRX_E_INVALID, "bad kernel_variant");
    if (cfg->max_instances_per_world > (1u << 20))
        return fail(MRX_E_INVALID, "max_instances_per_world out of range");
    if (cfg->num_devices > 1 && !cfg->device_ids)
        return fail(MRX_E_INVALID, "device_ids is null while num_devices is not zero");
    if (cfg->num_devices > 64)
        return fail(MRX_E_INVALID, "more than 64 devices");
    if (cfg->num_devices > 1 && cfg->num_devices > cfg->num_worlds)
        return fail(MRX_E_INVALID, "more devices (" + std::to_string(cfg->num_devices) + ") than worlds (" +
                                       std::to_string(cfg->num_worlds) + "): a shard would own no world");
    if (cfg->num_devices <= 1) {
        if (cfg->num_devices == 1 && cfg->device_ids)
            full.gpu_id = cfg->device_ids[0];
        full.num_devices = 0;
        full.device_ids = nullptr;
        return createOne(full, out);
    }
    // ---- one shard per listed device, each a renderer of its own over a contiguous world range
    if (cfg->stream)
        return fail(MRX_E_INVALID, "a renderer of several devices launches on each device's null stream: "
                                   "mrx_config.stream must be null (mrx_set_stream on a shard sets its stream)");
    mrx_renderer *top = new mrx_renderer();
    top->mode = cfg->render_mode;
    top->flags = cfg->flags;
    top->variant = cfg->kernel_variant;
    top->device = cfg->device_ids[0];
    const uint32_t n = cfg->num_devices;
    for (uint32_t i = 0; i < n; ++i) {
        mrx_config sub = full;
        const uint32_t lo = shardFirstWorld(cfg->num_worlds, i, n), hi = shardFirstWorld(cfg->num_worlds, i + 1, n);
        sub.gpu_id = cfg->device_ids[i];
        sub.worlds = cfg->worlds ? cfg->worlds + lo : nullptr;
        sub.num_worlds = hi - lo;
        sub.num_devices = 0;
        sub.device_ids = nullptr;
        mrx_renderer *sh = nullptr;
        const int rc = createOne(sub, &sh);
        if (rc != MRX_OK) {
            const std::string why = g_err;
            delete top;                               // (destroys the shards made so far)
            return fail(rc, "shard " + std::to_string(i) + " (device " + std::to_string(sub.gpu_id) + "): " + why);
        }
        sh->parent = top;
        top->shards.push_back(sh);
        top->shardFirstWorld.push_back(lo);
    }
    top->shardFirstWorld.push_back(cfg->num_worlds);
    startShardWorkers(top);
    *out = top;
    return MRX_OK;
}

void mrx_destroy(mrx_renderer *r)
{
    if (!r)
        return;
    stopShardWorkers(r);
    for (mrx_renderer *sh : r->shards) {
        (void)hipSetDevice(sh->device);
        (void)hipStreamSynchronize(sh->stream);
    }
    if (r->shards.empty()) {
        (void)hipSetDevice(r->device);
        (void)hipStreamSynchronize(r->stream);
    }
    delete r;
}

int mrx_render(mrx_renderer *r)
{
    if (r && r->parent) {                             // (a shard stepped on its own: behind what its renderer posted)
        const int src = settle(r);
        if (src != MRX_OK)
            return src;
    }
    if (!r)
        return fail(MRX_E_INVALID, "null renderer");
    // several devices: one launch each, enqueued concurrently by the shards' host threads
    // (shardsRun); on return every device has its kernel queued, nothing waits for a render
    if (!r->shards.empty())
        return shardsRun(r, kCmdRender);
    MRX_HIP(hipSetDevice(r->device));
    MRX_HIP(r->launch());
    return MRX_OK;
}

int mrx_step(mrx_renderer *r)
{
    // Manager::step = Step graph + Render graphs (mgr.cpp:177-185).  The Step
    // graph only advances a per-world clock nothing reads (sim.cpp:73-77) and
    // re-sorts static entities; what remains observable is the render.
    if (!r)
        return fail(MRX_E_INVALID, "null renderer");
    r->stepCount++;
    return mrx_render(r);
}

int mrx_sync(mrx_renderer *r)
{
    if (r && r->parent) {
        const int src = settle(r);
        if (src != MRX_OK)
            return src;
    }
    if (!r)
        return fail(MRX_E_INVALID, "null renderer");
    if (!r->shards.empty())
        return shardsRun(r, kCmdSync);
    MRX_HIP(hipSetDevice(r->device));
    MRX_HIP(hipStreamSynchronize(r->stream));
    return MRX_OK;
}

void *mrx_stream(mrx_renderer *r) { return r ? (void *)r->stream : nullptr; }

static int wantsShard(const char *what)
{
    return fail(MRX_E_UNSUPPORTED, std::string(what) + ": this renderer spans several devices -- "
                                   "address one shard (mrx_shard / mrx_buffer_shard)");
}

int mrx_num_shards(mrx_renderer *r) { return !r ? 0 : r->shards.empty() ? 1 : (int)r->shards.size(); }

mrx_renderer *mrx_shard(mrx_renderer *r, int shard)
{
    if (!r || shard < 0 || shard >= mrx_num_shards(r)) {
        fail(MRX_E_INVALID, "no such shard");
        return nullptr;
    }
    return r->shards.empty() ? r : r->shards[shard];
}

void *mrx_buffer_shard(mrx_renderer *r, int shard, int which, int64_t dims[4], int *ndim, int *dtype,
                       int *device)
{
    mrx_renderer *sh = mrx_shard(r, shard);
    return sh ? mrx_buffer(sh, which, dims, ndim, dtype, device) : nullptr;
}

int64_t mrx_shard_split(uint32_t num_worlds, uint32_t shard, uint32_t num_shards)
{
    if (num_shards == 0 || shard > num_shards)
        return fail(MRX_E_INVALID, "bad shard");
    return (int64_t)shardFirstWorld(num_worlds, shard, num_shards);
}

int64_t mrx_shard_first_world(mrx_renderer *r, int shard)
{
    if (!r || shard < 0 || shard > mrx_num_shards(r))
        return fail(MRX_E_INVALID, "no such shard");
    if (r->shards.empty())
        return shard == 0 ? 0 : (int64_t)r->info.num_worlds;
    return (int64_t)r->shardFirstWorld[shard];
}

int mrx_refresh_objects(mrx_renderer *r)
{
    {
        const int src = settle(r);
        if (src != MRX_OK)
            return src;
    }
    if (!r)
        return fail(MRX_E_INVALID, "null renderer");
    for (mrx_renderer *sh : r->shards) {
        const int rc = mrx_refresh_objects(sh);
        if (rc != MRX_OK)
            return rc;
    }
    if (!r->shards.empty())
        return MRX_OK;
    MRX_HIP(hipSetDevice(r->device));
    // renders in flight read the tables that are about to change
    MRX_HIP(hipStreamSynchronize(r->stream));
    std::vector<int32_t> live(r->boundObj.size());
    if (!live.empty())
        MRX_HIP(hipMemcpy(live.data(), r->instObj.ptr, live.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    // an id that names no object cannot be bound: refuse it before anything changes
    for (size_t i = 0; i < live.size(); ++i)
        if (live[i] >= 0 && (size_t)live[i] >= r->objFirst.size())
            return fail(MRX_E_INVALID, "instance row " + std::to_string(i) + " holds object id " +
                                           std::to_string(live[i]) + ", the scene has " +
                                           std::to_string(r->objFirst.size()) + " objects");
    bool changed = false;
    for (size_t i = 0; i < live.size(); ++i)
        if (live[i] >= 0 && live[i] != r->boundObj[i]) {
            r->boundObj[i] = live[i];
            changed = true;
        }
    if (!changed)
        return MRX_OK;
    return bindGeometry(*r);
}

int mrx_set_stream(mrx_renderer *r, void *stream)
{
    {
        const int src = settle(r);
        if (src != MRX_OK)
            return src;
    }
    if (!r)
        return fail(MRX_E_INVALID, "null renderer");
    if (!r->shards.empty())
        return wantsShard("mrx_set_stream");
    MRX_HIP(hipSetDevice(r->device));
    if ((hipStream_t)stream == r->stream)
        return MRX_OK;
    // renders already enqueued on the old stream finish before anything that
    // is enqueued on the new one can start
    MRX_HIP(hipStreamSynchronize(r->stream));
    r->stream = (hipStream_t)stream;
    return MRX_OK;
}

void *mrx_buffer(mrx_renderer *r, int which, int64_t dims[4], int *ndim, int *dtype,
                 int *device)
{
    if (settle(r) != MRX_OK)
        return nullptr;
    if (!r || !dims || !ndim || !dtype) {
        fail(MRX_E_INVALID, "null argument");
        return nullptr;
    }
    if (!r->shards.empty()) {
        wantsShard("mrx_buffer");
        return nullptr;
    }
    const bool rt = r->mode == MRX_MODE_RAYTRACER;
    const int64_t V = r->info.num_views, I = r->info.num_instances;
    const int64_t S = r->info.storage_slow, F = r->info.storage_fast;
    void *ptr = nullptr;
    if (device)
        *device = r->device;
    switch (which) {
    case MRX_BUF_RGB:       // mgr.cpp:547-568
        dims[0] = V; dims[1] = S; dims[2] = F; dims[3] = 4;
        *ndim = 4; *dtype = MRX_DTYPE_U8; ptr = r->rgb.ptr;
        break;
    case MRX_BUF_DEPTH:     // mgr.cpp:570-590
        dims[0] = V; dims[1] = S; dims[2] = F; dims[3] = 1;
        *ndim = rt ? 3 : 4; *dtype = MRX_DTYPE_F32; ptr = r->depth.ptr;
        break;
    case MRX_BUF_SEGMASK:   // mgr.cpp:592-605
        if (!rt) {
            fail(MRX_E_UNSUPPORTED, "Segmask not implemented for rasterizer");
            return nullptr;
        }
        if (!r->params.idsAreSegmask) {
            fail(MRX_E_UNSUPPORTED, "ids buffer holds visibility ids (MRX_FLAG_VISIBILITY_IDS)");
            return nullptr;
        }
        dims[0] = V; dims[1] = S; dims[2] = F;
        *ndim = 3; *dtype = MRX_DTYPE_I32; ptr = r->ids.ptr;
        break;
    case MRX_BUF_VISIBILITY:
        if (!r->ids.ptr || r->params.idsAreSegmask) {
            fail(MRX_E_UNSUPPORTED, "visibility ids need MRX_FLAG_VISIBILITY_IDS");
            return nullptr;
        }
        dims[0] = V; dims[1] = S; dims[2] = F;
        *ndim = 3; *dtype = MRX_DTYPE_I32; ptr = r->ids.ptr;
        break;
    case MRX_BUF_INSTANCE_POSITION:   // mgr.cpp:627-635
        dims[0] = I; dims[1] = 3; *ndim = 2; *dtype = MRX_DTYPE_F32; ptr = r->instPos.ptr;
        break;
    case MRX_BUF_INSTANCE_ROTATION:   // mgr.cpp:637-645
        dims[0] = I; dims[1] = 4; *ndim = 2; *dtype = MRX_DTYPE_F32; ptr = r->instRot.ptr;
        break;
    case MRX_BUF_INSTANCE_SCALE:
        dims[0] = I; dims[1] = 3; *ndim = 2; *dtype = MRX_DTYPE_F32; ptr = r->instScale.ptr;
        break;
    case MRX_BUF_INSTANCE_OBJECT:     // ObjectID column (sim.cpp:152-156); negative = hidden
        dims[0] = I; *ndim = 1; *dtype = MRX_DTYPE_I32; ptr = r->instObj.ptr;
        break;
    // The reference sizes the camera tensors with totalNumInstances
    // (mgr.cpp:652,662); the rows that exist are one per camera, exported so.
    case MRX_BUF_CAMERA_POSITION:
        dims[0] = V; dims[1] = 3; *ndim = 2; *dtype = MRX_DTYPE_F32; ptr = r->camPos.ptr;
        break;
    case MRX_BUF_CAMERA_ROTATION:
        dims[0] = V; dims[1] = 4; *ndim = 2; *dtype = MRX_DTYPE_F32; ptr = r->camRot.ptr;
        break;
    default:
        fail(MRX_E_INVALID, "unknown buffer id");
        return nullptr;
    }
    return ptr;
}

int mrx_copy_to_host(mrx_renderer *r, int which, void *dst, uint64_t bytes)
{
    {
        const int src = settle(r);
        if (src != MRX_OK)
            return src;
    }
    if (!r || !dst)
        return fail(MRX_E_INVALID, "null argument");
    if (!r->shards.empty())
        return wantsShard("mrx_copy_to_host");
    int64_t dims[4] = { 1, 1, 1, 1 };
    int nd = 0, dt = 0, dev = 0;
    void *src = mrx_buffer(r, which, dims, &nd, &dt, &dev);
    if (!src)
        return MRX_E_UNSUPPORTED;
    uint64_t total = dt == MRX_DTYPE_U8 ? 1 : 4;
    for (int i = 0; i < nd; ++i)
        total *= (uint64_t)dims[i];
    if (bytes > total)
        return fail(MRX_E_INVALID, "readback larger than the buffer");
    MRX_HIP(hipSetDevice(r->device));
    MRX_HIP(hipStreamSynchronize(r->stream));
    MRX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return MRX_OK;
}

static int infoFull(mrx_renderer *r, mrx_info_t *out)
{
    if (!r || !out)
        return fail(MRX_E_INVALID, "null argument");
    if (!r->shards.empty()) {
        // the shards share the scene: counts of what is per world add up, the rest is shard 0's
        mrx_info_t sum = r->shards[0]->info;
        for (size_t i = 1; i < r->shards.size(); ++i) {
            const mrx_info_t &o = r->shards[i]->info;
            sum.num_worlds += o.num_worlds;
            sum.num_views += o.num_views;
            sum.num_instances += o.num_instances;
            sum.bytes_per_step += o.bytes_per_step;
            sum.max_world_triangles = std::max(sum.max_world_triangles, o.max_world_triangles);
            sum.max_world_instances = std::max(sum.max_world_instances, o.max_world_instances);
            sum.render_path = std::max(sum.render_path, o.render_path);
        }
        sum.num_shards = (uint32_t)r->shards.size();
        *out = sum;
        return MRX_OK;
    }
    *out = r->info;
    return MRX_OK;
}

int mrx_info(mrx_renderer *r, mrx_info_t *out)
{
    // the ABI-2 entry point: its callers allocated the struct as it was then
    return mrx_info_sized(r, out, MRX_INFO_V2_SIZE);
}

int mrx_info_sized(mrx_renderer *r, void *out, size_t size)
{
    if (!r || !out)
        return fail(MRX_E_INVALID, "null argument");
    if (size < MRX_INFO_V2_SIZE)
        return fail(MRX_E_INVALID, "mrx_info_t size mismatch (ABI)");
    mrx_info_t full;
    const int rc = infoFull(r, &full);
    if (rc != MRX_OK)
        return rc;
    std::memcpy(out, &full, std::min(size, sizeof full));
    return MRX_OK;
}

int mrx_time_steps_host(mrx_renderer *r, int steps, double *us_per_step)
{
    if (!r || !us_per_step || steps <= 0)
        return fail(MRX_E_INVALID, "bad argument");
    g_trace = ShardTrace {};
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) {
        const int rc = mrx_step(r);
        if (rc != MRX_OK)
            return rc;
    }
    const auto t1 = std::chrono::steady_clock::now();
    *us_per_step = std::chrono::duration<double, std::micro>(t1 - t0).count() / (double)steps;
    if (shardTrace() && g_trace.n) {
        const double n = (double)g_trace.n * 1000.0;
        std::fprintf(stderr, "mrx shard trace (%lld steps, us): wake %.2f  worker launch %.2f  own launch %.2f  all enqueued %.2f\n",
                     (long long)g_trace.n, g_trace.wake / n, g_trace.launch / n, g_trace.own / n, g_trace.total / n);
        g_trace = ShardTrace {};
    }
    return mrx_sync(r);
}

int mrx_time_renders(mrx_renderer *r, int steps, float *ms_total)
{
    if (r && r->parent) {
        const int src = settle(r);
        if (src != MRX_OK)
            return src;
    }
    if (!r || !ms_total || steps < 0)
        return fail(MRX_E_INVALID, "bad argument");
    if (!r->shards.empty()) {
        // every device runs its `steps` launches between its own two events, all enqueued
        // before anything is waited for: the job takes as long as the slowest device
        {
            const int rc = shardsRun(r, kCmdTimed, steps);    // every device's launches from its own thread
            if (rc != MRX_OK)
                return rc;
        }
        *ms_total = 0.0f;
        for (mrx_renderer *sh : r->shards) {
            float ms = 0.0f;
            MRX_HIP(hipSetDevice(sh->device));
            MRX_HIP(hipEventSynchronize(sh->ev1));
            MRX_HIP(hipEventElapsedTime(&ms, sh->ev0, sh->ev1));
            *ms_total = std::max(*ms_total, ms);
        }
        return MRX_OK;
    }
    MRX_HIP(hipSetDevice(r->device));
    MRX_HIP(hipEventRecord(r->ev0, r->stream));
    for (int i = 0; i < steps; ++i)
        MRX_HIP(r->launch());
    MRX_HIP(hipEventRecord(r->ev1, r->stream));
    MRX_HIP(hipEventSynchronize(r->ev1));
    MRX_HIP(hipEventElapsedTime(ms_total, r->ev0, r->ev1));
    return MRX_OK;
}

int64_t mrx_debug_stamps(mrx_renderer *r, uint64_t *dst, int64_t capacity)
{
    if (settle(r) != MRX_OK)
        return 0;
    if (r && !r->shards.empty())
        r = r->shards[0];
    if (!r || !dst || !r->stamps.ptr)
        return 0;
    const int64_t n = (int64_t)r->stamps.count < capacity ? (int64_t)r->stamps.count : capacity;
    if (hipSetDevice(r->device) != hipSuccess || hipStreamSynchronize(r->stream) != hipSuccess ||
        hipMemcpy(dst, r->stamps.ptr, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return 0;
    return n;
}

int mrx_mark(mrx_renderer *r, int which)
{
    {
        const int src = settle(r);
        if (src != MRX_OK)
            return src;
    }
    if (!r || (which != 0 && which != 1))
        return fail(MRX_E_INVALID, "bad argument");
    for (mrx_renderer *sh : r->shards) {
        const int rc = mrx_mark(sh, which);
        if (rc != MRX_OK)
            return rc;
    }
    if (!r->shards.empty())
        return MRX_OK;
    MRX_HIP(hipSetDevice(r->device));
    MRX_HIP(hipEventRecord(which ? r->ev1 : r->ev0, r->stream));
    return MRX_OK;
}

int mrx_elapsed_ms(mrx_renderer *r, float *ms)
{
    {
        const int src = settle(r);
        if (src != MRX_OK)
            return src;
    }
    if (!r || !ms)
        return fail(MRX_E_INVALID, "null argument");
    if (!r->shards.empty()) {                     // the slowest device
        *ms = 0.0f;
        for (mrx_renderer *sh : r->shards) {
            float one = 0.0f;
            const int rc = mrx_elapsed_ms(sh, &one);
            if (rc != MRX_OK)
                return rc;
            *ms = std::max(*ms, one);
        }
        return MRX_OK;
    }
    MRX_HIP(hipSetDevice(r->device));
    MRX_HIP(hipEventSynchronize(r->ev1));
    MRX_HIP(hipEventElapsedTime(ms, r->ev0, r->ev1));
    return MRX_OK;
}

int mrx_placement(mrx_renderer *r, float *cand_us, int capacity, float *kept_us)
{
    if (r && !r->shards.empty())
        r = r->shards[0];
    if (!r || (!cand_us && capacity > 0))
        return fail(MRX_E_INVALID, "bad argument");
    for (size_t i = 0; i < r->placementUs.size() && (int)i < capacity; ++i)
        cand_us[i] = r->placementUs[i];
    if (kept_us)
        *kept_us = r->placementKeptUs;
    return (int)r->placementUs.size();
}

int mrx_copy_triangles(mrx_renderer *r, float *tri_pos, float *tri_uv, int32_t *tri_mat,
                       int32_t *obj_first, int32_t *obj_count)
{
    if (r && !r->shards.empty())
        r = r->shards[0];                         // every shard holds the whole object pool
    if (!r)
        return fail(MRX_E_INVALID, "null renderer");
    for (size_t t = 0; t < r->hostTris.size(); ++t) {
        if (tri_pos) std::memcpy(tri_pos + 9 * t, r->hostTris[t].p, 36);
        if (tri_uv) std::memcpy(tri_uv + 6 * t, r->hostTris[t].uv, 24);
        if (tri_mat) tri_mat[t] = r->hostTris[t].mat;
    }
    for (size_t o = 0; o < r->objFirst.size(); ++o) {
        if (obj_first) obj_first[o] = r->objFirst[o];
        if (obj_count) obj_count[o] = r->objCount[o];
    }
    return MRX_OK;
}

int mrx_load_obj(const char *path, float **tri_pos, float **tri_uv, uint32_t *num_tris)
{
    if (!path || !tri_pos || !tri_uv || !num_tris)
        return fail(MRX_E_INVALID, "null argument");
    mrx::TriSoup soup;
    std::string err;
    if (!mrx::loadOBJ(path, soup, err))
        return fail(MRX_E_ASSET, err);
    *num_tris = soup.numTris();
    *tri_pos = (float *)std::malloc(soup.pos.size() * sizeof(float) + 4);
    *tri_uv = (float *)std::malloc(soup.uv.size() * sizeof(float) + 4);
    std::memcpy(*tri_pos, soup.pos.data(), soup.pos.size() * sizeof(float));
    std::memcpy(*tri_uv, soup.uv.data(), soup.uv.size() * sizeof(float));
    return MRX_OK;
}

int mrx_obj_objects(const char *path, uint32_t *first_tri, uint32_t capacity)
{
    if (!path || (!first_tri && capacity))
        return fail(MRX_E_INVALID, "null argument");
    mrx::TriSoup soup;
    std::string err;
    if (!mrx::loadOBJ(path, soup, err))
        return fail(MRX_E_ASSET, err);
    for (size_t o = 0; o < soup.objStart.size() && o < capacity; ++o)
        first_tri[o] = soup.objStart[o];
    return (int)soup.objStart.size();
}

int mrx_decode_png(const char *path, uint8_t **rgba, uint32_t *width, uint32_t *height)
{
    if (!path || !rgba || !width || !height)
        return fail(MRX_E_INVALID, "null argument");
    mrx::Image img;
    std::string err;
    if (!mrx::decodePNG(path, img, err))
        return fail(MRX_E_ASSET, err);
    *width = img.width;
    *height = img.height;
    *rgba = (uint8_t *)std::malloc(img.rgba.size() + 4);
    std::memcpy(*rgba, img.rgba.data(), img.rgba.size());
    return MRX_OK;
}

int mrx_decode_texture(const char *path, uint8_t **rgba, uint32_t *width, uint32_t *height)
{
    if (!path || !rgba || !width || !height)
        return fail(MRX_E_INVALID, "null argument");
    mrx::Image img;
    std::string err;
    if (!mrx::decodeTexture(path, img, err))
        return fail(MRX_E_ASSET, err);
    *width = img.width;
    *height = img.height;
    *rgba = (uint8_t *)std::malloc(img.rgba.size() + 4);
    std::memcpy(*rgba, img.rgba.data(), img.rgba.size());
    return MRX_OK;
}

int mrx_decode_bc7(const uint8_t *blocks, uint32_t num_blocks, uint8_t *rgba)
{
    if ((!blocks || !rgba) && num_blocks)
        return fail(MRX_E_INVALID, "null argument");
    for (uint32_t b = 0; b < num_blocks; ++b)
        mrx::decodeBC7Block(blocks + 16 * (size_t)b, reinterpret_cast<uint8_t (*)[4]>(rgba + 64 * (size_t)b));
    return MRX_OK;
}

int64_t mrx_describe_obj_materials(const char *path, char *json, uint64_t capacity)
{
    if (!path || !json)
        return fail(MRX_E_INVALID, "null argument");
    mrx::TriSoup soup;
    std::string err;
    if (!mrx::loadOBJ(path, soup, err))
        return fail(MRX_E_ASSET, err);
    auto quote = [](const std::string &s) {
        std::string o = "\"";
        for (char c : s) {
            if (c == '"' || c == '\\') o += '\\';
            o += c;
        }
        return o + "\"";
    };
    std::string out = "{\"num_tris\":" + std::to_string(soup.numTris()) + ",\"tri_mtl\":[";
    for (size_t i = 0; i < soup.triMtl.size(); ++i)
        out += (i ? "," : "") + std::to_string(soup.triMtl[i]);
    out += "],\"names\":[";
    for (size_t i = 0; i < soup.mtlNames.size(); ++i)
        out += (i ? "," : "") + quote(soup.mtlNames[i]);
    out += "],\"libs\":[";
    for (size_t i = 0; i < soup.mtlLibs.size(); ++i)
        out += (i ? "," : "") + quote(soup.mtlLibs[i]);
    out += "],\"materials\":[";
    std::vector<mrx::MtlMaterial> lib;
    for (const std::string &ml : soup.mtlLibs) {
        std::string merr;
        (void)mrx::loadMTL(ml, lib, merr);
    }
    for (size_t i = 0; i < lib.size(); ++i) {
        char kd[96];
        std::snprintf(kd, sizeof kd, "[%.9g,%.9g,%.9g]", lib[i].kd[0], lib[i].kd[1], lib[i].kd[2]);
        out += std::string(i ? "," : "") + "{\"name\":" + quote(lib[i].name) + ",\"kd\":" + kd +
               ",\"map_kd\":" + quote(lib[i].mapKd) + "}";
    }
    out += "]}";
    if (out.size() + 1 > capacity)
        return fail(MRX_E_INVALID, "buffer too small");
    std::memcpy(json, out.c_str(), out.size() + 1);
    return (int64_t)out.size();
}

uint32_t mrx_dispatch_min_tris(uint32_t base, uint32_t num_views, int textured, uint32_t width, uint32_t height,
                               uint32_t num_cus)
{
    return mrx::bvhDispatchMinTris(base ? base : mrx::kBvhMinTris, num_views, textured != 0, width, height, num_cus);
}

uint32_t mrx_group_fill(uint32_t num_cus) { return mrx::groupFill(num_cus); }

int mrx_dispatch_flat(int raytracer, uint32_t num_views, uint32_t width, uint32_t height, uint32_t max_world_triangles,
                      uint32_t max_world_instances, uint32_t num_cus)
{
    // (Raytracer storage is square: res = width)
    return mrx::bvhDispatchFlat(raytracer != 0, num_views, width, raytracer ? width : height, max_world_triangles,
                                max_world_instances, num_cus) ? 1 : 0;
}

int mrx_blas_check(const float *tri_pos, uint32_t num_tris, uint32_t *num_nodes, uint32_t *depth,
                   uint32_t *num_leaves)
{
    if ((!tri_pos && num_tris) || !num_nodes || !depth || !num_leaves)
        return fail(MRX_E_INVALID, "null argument");
    using namespace mrx;
    std::vector<ObjTri> tris(num_tris);
    for (uint32_t t = 0; t < num_tris; ++t)
        std::memcpy(tris[t].p, tri_pos + 9 * (size_t)t, 36);
    BlasSet b;
    buildBlas(tris.data(), { 0 }, { (int32_t)num_tris }, b);
    *num_nodes = (uint32_t)b.nodes.size();
    *depth = b.maxDepth;
    *num_leaves = 0;
    const ObjInfo &o = b.objects[0];
    if (o.root < 0)
        return num_tris <= kBvhFlatMax && b.nodes.empty() ? MRX_OK
                                                          : fail(MRX_E_INVALID, "large object without a hierarchy");
    if (1 + 7 * b.maxDepth > kBvhStackCap)
        return fail(MRX_E_INVALID, "hierarchy too deep for the traversal stack");
    std::vector<uint32_t> seen(num_tris, 0);
    // (node, box that must contain everything below it)
    struct Item { uint32_t node; float lo[3], hi[3]; };
    std::vector<Item> todo;
    {
        Item root {};
        root.node = (uint32_t)o.root;
        std::memcpy(root.lo, o.bbMin, 12);
        std::memcpy(root.hi, o.bbMax, 12);
        todo.push_back(root);
    }
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.node >= b.nodes.size())
            return fail(MRX_E_INVALID, "child index out of range");
        const BvhNode &n = b.nodes[it.node];
        for (uint32_t c = 0; c < kBvhWidth; ++c) {
            const uint32_t ref = n.child[c];
            if (ref == kBvhEmpty)
                continue;
            for (int a = 0; a < 3; ++a)
                if (n.bmin[c][a] < it.lo[a] || n.bmax[c][a] > it.hi[a])
                    return fail(MRX_E_INVALID, "child box sticks out of its parent's");
            if (ref & kBvhLeafBit) {
                const uint32_t cnt = ((ref >> kBvhLeafStartBits) & 15u) + 1u;
                const uint32_t start = ref & ((1u << kBvhLeafStartBits) - 1u);
                if (cnt > kBvhLeafMax || start + cnt > b.leafTris.size())
                    return fail(MRX_E_INVALID, "bad leaf");
                ++*num_leaves;
                for (uint32_t i = 0; i < cnt; ++i) {
                    const uint32_t t = b.leafTris[start + i];
                    if (t >= num_tris || seen[t]++)
                        return fail(MRX_E_INVALID, "triangle missing from or repeated in the leaves");
                    for (int v = 0; v < 3; ++v)
                        for (int a = 0; a < 3; ++a) {
                            const float x = tris[t].p[3 * v + a];
                            if (x < n.bmin[c][a] || x > n.bmax[c][a])
                                return fail(MRX_E_INVALID, "triangle outside its leaf's box");
                        }
                }
            } else {
                Item ch {};
                ch.node = ref;
                std::memcpy(ch.lo, n.bmin[c], 12);
                std::memcpy(ch.hi, n.bmax[c], 12);
                todo.push_back(ch);
            }
        }
    }
    for (uint32_t t = 0; t < num_tris; ++t)
        if (seen[t] != 1)
            return fail(MRX_E_INVALID, "triangle missing from the leaves");
    return MRX_OK;
}

void mrx_free(void *p) { std::free(p); }

}  // extern "C"
