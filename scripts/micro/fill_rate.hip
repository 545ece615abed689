// Calibration: what a pure streaming-store kernel reaches on this box for the
// headline output size (128 MiB) -- the practical ceiling for the renderer.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NT>
__global__ __launch_bounds__(256) void fill(uint4 *dst, size_t n, uint32_t v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const uint4 val = make_uint4(v, v + 1, v + 2, v + 3);
    for (; i < n; i += stride) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 vv = { val.x, val.y, val.z, val.w };
        if (NT) __builtin_nontemporal_store(vv, reinterpret_cast<u32x4 *>(dst + i));
        else dst[i] = val;
    }
}

// tile-shaped: one workgroup writes a 32 KiB contiguous tile (like the renderer)
__global__ __launch_bounds__(256) void fillTiles(uint4 *dst, uint32_t v)
{
    uint4 *base = dst + (size_t)blockIdx.x * 2048;   // 32 KiB per WG
    const uint4 val = make_uint4(v, v + 1, v + 2, v + 3);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        base[i * 256 + threadIdx.x] = val;
}

int main()
{
    const size_t bytes = 128ull << 20, n = bytes / 16;
    uint4 *d;
    hipMalloc(&d, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int blocks : {2048, 4096, 16384}) {
            if (mode == 2 && blocks != 4096) continue;
            const int reps = 50;
            for (int w = 0; w < 5; ++w) fill<0><<<blocks, 256>>>(d, n, w);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                if (mode == 0) fill<0><<<blocks, 256>>>(d, n, r);
                else if (mode == 1) fill<1><<<blocks, 256>>>(d, n, r);
                else fillTiles<<<4096, 256>>>(d, r);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s blocks=%5d: %.2f us/launch  %.2f TB/s\n",
                   mode == 0 ? "plain" : mode == 1 ? "nt   " : "tiles", blocks, ms / reps * 1e3,
                   bytes / (ms / reps * 1e-3) / 1e12);
        }
    return 0;
}
