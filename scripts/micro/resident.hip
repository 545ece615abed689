// Experiment: does a RESIDENT kernel -- workgroups that stay on the chip between steps and
// wait for a doorbell -- beat a kernel launch per step for small batches?  The step must stay stream-ordered
// (pose writes before it, readers after it, on the caller's stream), so the doorbell is rung by
// hipStreamWriteValue32 on that stream and completion is ordered by hipStreamWaitValue32 on a done word.
// This measures the mechanism alone, with the store work of BASELINE configs[1] (1024 views x 64x64 x 8 B =
// 32 MiB per step) and with almost none:
//   launch      one kernel launch per step (what mrx_step does), K steps back to back on one stream
//   resident    WriteValue32(step) -> resident kernel sees it, works, last workgroup writes done -> WaitValue32(done)
// Every spin has a watchdog (s_memrealtime, 2 s): a lost doorbell ends the kernel instead of hanging the device.
#include <hip/hip_runtime.h>
#ifndef FENCE_MODE
#define FENCE_MODE 1
#endif
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void fill(uint4 *out, unsigned per, unsigned step)
{
    // `per` 16-byte stores per thread, coalesced, write-through like the renderer's
    uint4 *dst = out + (size_t)blockIdx.x * blockDim.x * per + threadIdx.x;
    for (unsigned i = 0; i < per; ++i) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = { step, i, blockIdx.x, threadIdx.x };
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst + (size_t)i * blockDim.x), "v"(v) : "memory");
    }
}

__global__ __launch_bounds__(256) void stepKernel(uint4 *out, unsigned per, unsigned step) { fill(out, per, step); }

// ctl[0] = step doorbell (written by the stream), ctl[1] = exit flag, ctl[2] = workgroups done (device counter)
#define DBG(i, v) do { if (dbg && blockIdx.x == 0 && lane == 0) __hip_atomic_store(dbg + (i), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
// Control flow is WAVE-uniform (scalar branches on readfirstlane'd values), as in the renderer's kernels: the first
// version had `if (threadIdx.x == 0) { spin }` between __syncthreads() -- lane 0 against lanes 1..63 of ONE wave -- and the
// compiler's linearisation let lanes 1..63 run ahead through the barriers (s_barrier counts waves, not lanes) while lane
// 0's continuation was deferred behind a loop that never ended: the kernel re-ran step 1 for ever.
__global__ __launch_bounds__(256) void residentKernel(uint4 *out, unsigned per, volatile unsigned *ctl, unsigned *doneCount,
                                                       volatile unsigned *doneSignal, unsigned *dbg = nullptr)
{
    __shared__ unsigned cur;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    unsigned seen = 0;
    for (;;) {
        if (wave == 0) {
            // every lane of wave 0 polls the same word (one request per wave: the loads coalesce)
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned s;
            for (;;) {
                s = __builtin_amdgcn_readfirstlane(__hip_atomic_load((unsigned *)ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                const unsigned ex = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load((unsigned *)ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                if (ex)
                    s = 0xFFFFFFFFu;
                if (s != seen)
                    break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {   // 2 s at 100 MHz: watchdog
                    s = 0xFFFFFFFFu;
                    break;
                }
                // (~0.4 us between polls: 512 workgroups polling one word)
                __builtin_amdgcn_s_sleep(16);
            }
            if (lane == 0)
                cur = s;
            DBG(0, 1u); DBG(1, s);
        }
        __syncthreads();
        const unsigned s = __builtin_amdgcn_readfirstlane(cur);
        __syncthreads();
        if (s == 0xFFFFFFFFu)
            return;
        seen = s;
        fill(out, per, s);
        // the stores above are write-through (sc1): once acknowledged they are where every agent reads them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wave == 0) {
            unsigned n = 0;
            if (lane == 0)
                n = atomicAdd(doneCount, 1u) + 1u;
            n = __builtin_amdgcn_readfirstlane(n);
            DBG(0, 5u); DBG(2, n);
            if (n == gridDim.x * s && lane == 0)                     // the step's last workgroup
                __hip_atomic_store((unsigned *)doneSignal, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static double nowUs()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Wait for stream `s` for at most `seconds`; false = still busy.
static bool waitBounded(hipStream_t s, double seconds)
{
    const double t0 = nowUs();
    while (hipStreamQuery(s) == hipErrorNotReady)
        if (nowUs() - t0 > seconds * 1e6)
            return false;
    return true;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const unsigned groups = 512, threads = 256;
    hipStream_t S, P;
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&P, hipStreamNonBlocking));
    unsigned *ctl = nullptr, *doneCount = nullptr, *doneSignal = nullptr;
    CK(hipExtMallocWithFlags((void **)&ctl, 64, hipDeviceMallocUncached));
    CK(hipMalloc(&doneCount, 4));
    hipError_t se = hipExtMallocWithFlags((void **)&doneSignal, 8, hipMallocSignalMemory);
    if (se != hipSuccess) { printf("signal memory: %s\n", hipGetErrorString(se)); return 1; }
    uint4 *out = nullptr;
    CK(hipMalloc(&out, (size_t)groups * threads * 32 * 16));
    auto reset = [&]() -> int {
        CK(hipMemset(ctl, 0, 64)); CK(hipMemset(doneCount, 0, 4)); CK(hipMemset(doneSignal, 0, 8));
        CK(hipDeviceSynchronize());
        return 0;
    };
    // ---- stage A: the watchdog alone ends an un-rung resident kernel
    if (reset()) return 1;
    double t0 = nowUs();
    residentKernel<<<groups, threads, 0, P>>>(out, 1, ctl, doneCount, doneSignal);
    CK(hipGetLastError());
    const bool endedA = waitBounded(P, 6.0);
    printf("stage A (watchdog, no doorbell): resident kernel %s after %.2f s\n", endedA ? "ended" : "STILL RUNNING", (nowUs() - t0) / 1e6);
    if (!endedA) _exit(4);
    // ---- stage B: doorbell, done word and exit flag in host-mapped memory, written / read by the CPU directly
    {
        unsigned *hctl = nullptr, *hdone = nullptr;
        CK(hipHostMalloc((void **)&hctl, 64, hipHostMallocMapped));
        CK(hipHostMalloc((void **)&hdone, 64, hipHostMallocMapped));
        hctl[0] = hctl[1] = 0; hdone[0] = 0;
        for (int i = 8; i < 16; ++i) hdone[i] = 0;
        if (reset()) return 1;
        residentKernel<<<groups, threads, 0, P>>>(out, 1, hctl, doneCount, hdone, hdone + 8);
        CK(hipGetLastError());
        for (unsigned step = 1; step <= 3; ++step) {
            t0 = nowUs();
            __atomic_store_n(&hctl[0], step, __ATOMIC_RELEASE);
            while (__atomic_load_n(&hdone[0], __ATOMIC_ACQUIRE) != step && nowUs() - t0 < 1e6) {}
            printf("stage B (host-mapped doorbell): step %u done word %u after %.1f us; workgroup 0 progress %u (doorbell it saw %u, its count %u)\n",
                   step, hdone[0], nowUs() - t0, hdone[8], hdone[9], hdone[10]);
        }
        __atomic_store_n(&hctl[1], 1u, __ATOMIC_RELEASE);
        const bool endedB = waitBounded(P, 6.0);
        printf("stage B: resident kernel %s after the exit flag\n", endedB ? "ended" : "STILL RUNNING");
        if (!endedB) _exit(4);
        // ---- stage B2: the same through hipMemcpy into device memory
        if (reset()) return 1;
        residentKernel<<<groups, threads, 0, P>>>(out, 1, ctl, doneCount, doneSignal);
        CK(hipGetLastError());
        const unsigned one = 1u;
        unsigned ds = 0;
        t0 = nowUs();
        CK(hipMemcpy(ctl, &one, 4, hipMemcpyHostToDevice));
        printf("stage B2: hipMemcpy of the doorbell returned after %.0f us\n", nowUs() - t0);
        while (nowUs() - t0 < 1e6) {
            CK(hipMemcpy(&ds, doneSignal, 4, hipMemcpyDeviceToHost));
            if (ds == 1u) break;
        }
        printf("stage B2 (hipMemcpy doorbell into device memory): done signal %u after %.0f us\n", ds, nowUs() - t0);
        CK(hipMemcpy(ctl + 1, &one, 4, hipMemcpyHostToDevice));
        const bool endedB2 = waitBounded(P, 6.0);
        printf("stage B2: resident kernel %s after the exit flag\n", endedB2 ? "ended" : "STILL RUNNING");
        if (!endedB2) _exit(4);
    }
    // ---- stage C: stream-ordered doorbell, against a launch per step
    for (unsigned per : {1u, 16u, 32u}) {                // 2, 32, 64 MiB per step
        const int K = 2000;
        for (int i = 0; i < 200; ++i) stepKernel<<<groups, threads, 0, S>>>(out, per, i);
        CK(hipStreamSynchronize(S));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, S));
        t0 = nowUs();
        for (int i = 0; i < K; ++i) stepKernel<<<groups, threads, 0, S>>>(out, per, i);
        CK(hipEventRecord(e1, S));
        CK(hipStreamSynchronize(S));
        double wallLaunch = (nowUs() - t0) / K;
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%3u MiB/step  launch per step:   %.2f us/step device (events), %.2f us/step wall\n",
               groups * threads * per * 16 >> 20, ms * 1000 / K, wallLaunch);
        if (reset()) return 1;
        residentKernel<<<groups, threads, 0, P>>>(out, per, ctl, doneCount, doneSignal);
        CK(hipGetLastError());
        unsigned step = 0;
        auto stepOnce = [&]() -> hipError_t {
            ++step;
            hipError_t e = hipStreamWriteValue32(S, ctl, step, 0);
            if (e != hipSuccess) return e;
            return hipStreamWaitValue32(S, doneSignal, step, hipStreamWaitValueGte, 0xFFFFFFFFu);
        };
        auto giveUp = [&](const char *what) {
            unsigned h[3] = {}, dc = 0, ds = 0;
            (void)hipMemcpy(h, ctl, 12, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&dc, doneCount, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(&ds, doneSignal, 4, hipMemcpyDeviceToHost);
            printf("stage C: %s: doorbell %u exit %u, workgroup-steps done %u, done signal %u, step %u, resident kernel %s\n", what,
                   h[0], h[1], dc, ds, step, hipStreamQuery(P) == hipErrorNotReady ? "running" : "ended");
            const unsigned one = 1u, big = 0x7FFFFFFFu;
            (void)hipMemcpy(ctl + 1, &one, 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(doneSignal, &big, 4, hipMemcpyHostToDevice);
            printf("released: S %s, P %s\n", waitBounded(S, 5.0) ? "idle" : "STILL BUSY", waitBounded(P, 5.0) ? "idle" : "STILL BUSY");
            _exit(5);
        };
        hipError_t e = stepOnce();
        if (e != hipSuccess) { printf("stream memory op: %s\n", hipGetErrorString(e)); giveUp("stream memory op failed"); }
        if (!waitBounded(S, 3.0)) giveUp("step 1 did not complete in 3 s");
        for (int i = 0; i < 200 && e == hipSuccess; ++i) e = stepOnce();
        if (!waitBounded(S, 5.0)) giveUp("warm-up steps hang");
        CK(hipEventRecord(e0, S));
        t0 = nowUs();
        for (int i = 0; i < K && e == hipSuccess; ++i) e = stepOnce();
        CK(hipEventRecord(e1, S));
        if (!waitBounded(S, 10.0)) giveUp("timed steps hang");
        const double wall = (nowUs() - t0) / K;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%3u MiB/step  resident kernel:   %.2f us/step device (events), %.2f us/step wall\n",
               groups * threads * per * 16 >> 20, ms * 1000 / K, wall);
        CK(hipStreamWriteValue32(S, ctl + 1, 1u, 0));
        if (!waitBounded(S, 5.0) || !waitBounded(P, 5.0)) giveUp("exit flag not seen");
        unsigned dc = 0;
        CK(hipMemcpy(&dc, doneCount, 4, hipMemcpyDeviceToHost));
        printf("              resident kernel ended; workgroup-steps counted %u (expected %u)\n", dc, groups * step);
    }
    return 0;
}
