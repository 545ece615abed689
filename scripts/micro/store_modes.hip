// Calibration: cache-policy bits on the streaming stores of a tile-shaped fill
// (one workgroup = one contiguous 32 KiB tile, like the renderer) and what the
// launch-to-launch gap costs: per-launch time back to back (eager and inside a
// graph) next to the in-kernel span (first entry to last exit, s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ void store16(u32x4 *p, u32x4 v)
{
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256) void fillTiles(u32x4 *dst, uint32_t v, unsigned long long *stamps)
{
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    u32x4 *base = dst + (size_t)blockIdx.x * 2048;   // 32 KiB per WG
    const u32x4 val = { v, v + 1, v + 2, v + 3 };
#pragma unroll
    for (int i = 0; i < 8; ++i)
        store16<MODE>(base + i * 256 + threadIdx.x, val);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamps && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int MODE>
static void run(u32x4 *d, unsigned long long *stamps, hipStream_t s, const char *name)
{
    const size_t bytes = 128ull << 20;
    const int reps = 50;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) fillTiles<MODE><<<4096, 256, 0, s>>>(d, w, nullptr);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) fillTiles<MODE><<<4096, 256, 0, s>>>(d, r, nullptr);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const float eager = ms / reps * 1e3f;
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int r = 0; r < reps; ++r) fillTiles<MODE><<<4096, 256, 0, s>>>(d, r, nullptr);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const float graph = ms / reps * 1e3f;
    // in-kernel span of the last of a few stamped launches
    for (int r = 0; r < 3; ++r) fillTiles<MODE><<<4096, 256, 0, s>>>(d, r, stamps);
    hipStreamSynchronize(s);
    std::vector<unsigned long long> h(8192);
    hipMemcpy(h.data(), stamps, 8192 * 8, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (int i = 0; i < 4096; ++i) { lo = std::min(lo, h[2 * i]); hi = std::max(hi, h[2 * i + 1]); }
    printf("%-12s eager %.2f us/launch (%.2f TB/s)  graph %.2f us/launch (%.2f TB/s)  in-kernel span %.2f us\n",
           name, eager, bytes / (eager * 1e-6) / 1e12, graph, bytes / (graph * 1e-6) / 1e12, (hi - lo) / 100.0);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
}

int main()
{
    u32x4 *d; unsigned long long *stamps;
    hipMalloc(&d, 128ull << 20);
    hipMalloc(&stamps, 8192 * 8);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    run<0>(d, stamps, s, "plain");
    run<1>(d, stamps, s, "nt");
    run<2>(d, stamps, s, "sc0");
    run<3>(d, stamps, s, "sc1");
    run<4>(d, stamps, s, "sc0 sc1");
    run<5>(d, stamps, s, "sc0 sc1 nt");
    run<0>(d, stamps, s, "plain");
    return 0;
}
