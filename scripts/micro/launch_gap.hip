// Calibration: back-to-back launch cost of trivial kernels on the null stream.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void nop(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
int main()
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 256, 1024, 4096}) {
        for (int w = 0; w < 20; ++w) nop<<<blocks, 256>>>(nullptr);
        hipDeviceSynchronize();
        const int reps = 200;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) nop<<<blocks, 256>>>(nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("nop<<<%4d,256>>>: %.2f us per launch\n", blocks, ms / reps * 1e3);
    }
    return 0;
}
