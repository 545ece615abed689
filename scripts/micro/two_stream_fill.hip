// The last placement experiment: is the 20 % slow / fast mode of
// outputs of 256 MiB and more a property of the renderer or of the platform?  A pure fill
// with the renderer's exact store pattern -- per lane one `global_store_dwordx4 ... sc1`
// into the rgb tensor and one into the depth tensor, four consecutive pixels of one row
// of a 32x8 region, eight waves per 64x64 tile -- and no other work, over a sequence of
// freshly allocated tensor pairs.  If the pure fill shows the same two modes from
// allocation to allocation, the modes come from where the allocator puts the pages.
//
//   hipcc --offload-arch=gfx950 -O3 -o two_stream_fill two_stream_fill.hip
//   ./two_stream_fill [MiB per tensor = 256] [candidates = 12] [row pixels = 128]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store16(void *p, u32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
}

// one workgroup = one 64x64 tile of a view whose rows are `nfast` pixels long;
// wave = 64x8 strip, lane = four pixels of a row in each 32-pixel half
__global__ __launch_bounds__(512) void fillTwo(uint32_t *rgb, uint32_t *depth, uint32_t nfast, uint32_t tilesFast,
                                               uint32_t tilesPerView, uint32_t nslow, uint32_t v)
{
    const uint32_t view = blockIdx.x / tilesPerView, tile = blockIdx.x % tilesPerView;
    const uint32_t tx = (tile % tilesFast) * 64, ty = (tile / tilesFast) * 64;
    const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64, lx = lane & 7, ly = lane >> 3;
    const size_t base = ((size_t)view * nslow + ty + 8 * wave + ly) * nfast + tx + 4 * lx;
    const u32x4 c = { v, v + 1, v + 2, v + 3 };
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        store16(rgb + base + 32 * hf, c);
        store16(depth + base + 32 * hf, c);
    }
}

// one tensor only (the same bytes in total when launched for both): control
__global__ __launch_bounds__(512) void fillOne(uint32_t *t, uint32_t nfast, uint32_t tilesFast, uint32_t tilesPerView,
                                               uint32_t nslow, uint32_t v)
{
    const uint32_t view = blockIdx.x / tilesPerView, tile = blockIdx.x % tilesPerView;
    const uint32_t tx = (tile % tilesFast) * 64, ty = (tile / tilesFast) * 64;
    const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64, lx = lane & 7, ly = lane >> 3;
    const size_t base = ((size_t)view * nslow + ty + 8 * wave + ly) * nfast + tx + 4 * lx;
    const u32x4 c = { v, v + 1, v + 2, v + 3 };
    store16(t + base, c);
    store16(t + base + 32, c);
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? std::atoll(argv[1]) : 256;
    const int cands = argc > 2 ? std::atoi(argv[2]) : 12;
    const uint32_t nfast = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 128u;
    const size_t bytes = mib << 20, px = bytes / 4;
    const uint32_t nslow = nfast, tilesFast = nfast / 64, tilesPerView = tilesFast * tilesFast;
    const uint32_t views = (uint32_t)(px / ((size_t)nfast * nslow));
    const uint32_t grid = views * tilesPerView;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("two-stream fill: %zu MiB + %zu MiB, %u views of %ux%u, %u workgroups of 512\n", mib, mib, views, nfast, nslow, grid);
    // the renderer's layout for outputs of this size: one block per tensor, depth at phase 256 KiB
    // (mode 's'), and all in one block with the same phase (mode 'o'), alternating
    std::vector<float> both, single;
    void *held = nullptr;                             // the previous candidate stays alive: the next cannot reuse its block
    for (int k = 0; k < cands; ++k) {
        void *blockA = nullptr, *blockB = nullptr;
        uint32_t *rgb, *depth;
        const bool one = (k & 1) != 0;
        if (one) {
            if (hipMalloc(&blockA, 2 * bytes + (1u << 20)) != hipSuccess) break;
            rgb = (uint32_t *)blockA;
            depth = (uint32_t *)((char *)blockA + bytes + (256u << 10));
        } else {
            if (hipMalloc(&blockA, bytes) != hipSuccess) break;
            if (hipMalloc(&blockB, bytes + (256u << 10)) != hipSuccess) break;
            rgb = (uint32_t *)blockA;
            depth = (uint32_t *)((char *)blockB + (256u << 10));
        }
        auto timeIt = [&](int which) {
            const int reps = 40;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0, 0);
                for (int r = 0; r < reps; ++r) {
                    if (which == 0) fillTwo<<<grid, 512>>>(rgb, depth, nfast, tilesFast, tilesPerView, nslow, r);
                    else {
                        fillOne<<<grid, 512>>>(rgb, nfast, tilesFast, tilesPerView, nslow, r);
                        fillOne<<<grid, 512>>>(depth, nfast, tilesFast, tilesPerView, nslow, r);
                    }
                }
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) best = std::min(best, ms / reps * 1e3f);     // (first repeat: warm-up)
            }
            return best;
        };
        if (k == 0) { timeIt(0); timeIt(0); timeIt(0); }                 // clocks
        const float tb = timeIt(0), ts = timeIt(1);
        both.push_back(tb); single.push_back(ts);
        printf("candidate %2d (%s)  rgb %p depth %p   both tensors per launch %7.2f us (%.2f TB/s)   one tensor per launch, two launches %7.2f us\n",
               k, one ? "one block " : "two blocks", (void *)rgb, (void *)depth, tb, 2.0 * bytes / (tb * 1e-6) / 1e12, ts);
        fflush(stdout);
        if (held) hipFree(held);
        held = nullptr;
        if (blockB) { hipFree(blockA); held = blockB; } else held = blockA;
    }
    if (held) hipFree(held);
    if (argc > 4) {
        // scan mode (argv[4] = steps of 64 KiB): ONE pair of blocks, kept; the depth tensor slides through its block
        // in 64 KiB steps.  Does the mode follow the offset inside fixed physical blocks, or only the blocks?
        const int steps = std::atoi(argv[4]);
        void *a = nullptr, *b = nullptr;
        if (hipMalloc(&a, bytes) == hipSuccess && hipMalloc(&b, bytes + ((size_t)steps << 16) + (1u << 20)) == hipSuccess) {
            uint32_t *rgb = (uint32_t *)a;
            printf("scan: rgb %p, depth block %p\n", a, b);
            for (int st = 0; st < steps; ++st) {
                uint32_t *depth = (uint32_t *)((char *)b + ((size_t)st << 16));
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0, 0);
                    for (int r = 0; r < 20; ++r)
                        fillTwo<<<grid, 512>>>(rgb, depth, nfast, tilesFast, tilesPerView, nslow, r);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep) best = std::min(best, ms / 20 * 1e3f);
                }
                if (st % 16 == 0) printf("\n  depth offset %5d KiB:", st * 64);
                printf("%6.1f", best);
            }
            printf("\n");
        }
        if (a) hipFree(a);
        if (b) hipFree(b);
    }
    if (!both.empty()) {
        const float lo = *std::min_element(both.begin(), both.end()), hi = *std::max_element(both.begin(), both.end());
        const float slo = *std::min_element(single.begin(), single.end()), shi = *std::max_element(single.begin(), single.end());
        printf("both tensors per launch: min %.2f max %.2f us (max/min %.3f)   single-tensor launches: min %.2f max %.2f (%.3f)\n",
               lo, hi, hi / lo, slo, shi, shi / slo);
    }
    return 0;
}
