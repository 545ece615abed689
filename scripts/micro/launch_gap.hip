// Calibration: back-to-back launch cost of trivial kernels: null stream vs a
// created (non-blocking) stream vs a captured graph of the same launches.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void nop(int *p) { if (p && threadIdx.x == 9999) *p = 1; }

static float timeLaunches(hipStream_t s, int blocks, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 20; ++w) nop<<<blocks, 256, 0, s>>>(nullptr);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; ++r) nop<<<blocks, 256, 0, s>>>(nullptr);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main()
{
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int blocks : {1, 1024}) {
        printf("nop<<<%4d,256>>>: null %.2f us  created %.2f us  non-blocking %.2f us per launch\n", blocks,
               timeLaunches(nullptr, blocks, 200), timeLaunches(s1, blocks, 200), timeLaunches(s2, blocks, 200));
    }
    // graph of 100 launches
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s2, hipStreamCaptureModeGlobal);
    for (int r = 0; r < 100; ++r) nop<<<1024, 256, 0, s2>>>(nullptr);
    hipStreamEndCapture(s2, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s2); hipStreamSynchronize(s2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s2);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, s2);
    hipEventRecord(e1, s2); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("graph of 100 x nop<<<1024,256>>>: %.2f us per kernel\n", ms / 500 * 1e3f);
    return 0;
}
