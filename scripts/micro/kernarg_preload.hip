// Does preloading kernel arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=N: the command processor writes the
// first N dwords of the kernel-argument segment into user SGPRs at wave launch) shorten a launch whose first instruction
// needs an argument?  A kernel of 1024 workgroups that does one dependent load through a pointer argument and one store,
// timed back to back, built twice:
//   hipcc --offload-arch=gfx950 -O3 -o kp0 kernarg_preload.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 -o kp1 kernarg_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>

__global__ __launch_bounds__(512) void touch(const uint32_t *in, uint32_t *out, uint32_t n, uint32_t salt)
{
    // a chain of dependent loads (each thread chases indices through `in`): long enough that the device, not the
    // host's launch rate, sets the pace; the first load's address comes straight from the arguments
    uint32_t i = blockIdx.x * 512u + threadIdx.x;
    uint32_t v = i;
#pragma unroll 1
    for (int hop = 0; hop < 12; ++hop)
        v = in[(v + salt) % n];
    out[i] = v;
}

int main()
{
    const uint32_t n = 1024 * 512;
    uint32_t *in, *out;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, n * 4);
    hipMemset(in, 0, n * 4);     // every hop lands on index salt % n ... (v = 0 after the first hop): L2 hits, fixed latency
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        const int K = 2000;
        hipEventRecord(e0, 0);
        for (int i = 0; i < K; ++i)
            touch<<<1024, 512>>>(in, out, n, (uint32_t)i);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep)
            best = std::min(best, ms / K * 1e3f);
    }
    printf("%.3f us per launch (1024 workgroups x 512, a chain of 12 dependent loads, back to back)\n", best);
    return 0;
}
