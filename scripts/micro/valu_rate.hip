// Calibration: VALU issue rate on the box (FMA chains, readlane, cmp/cndmask).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {            // 8 independent FMA chains
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b);
            x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b);
            x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (MODE == 1) {     // readlane -> fma with sgpr
            int l = i & 63;
            float s0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x0), l));
            float s1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x1), l));
            float s2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x2), l));
            float s3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x3), l));
            x4 = __builtin_fmaf(s0, x4, b); x5 = __builtin_fmaf(s1, x5, b);
            x6 = __builtin_fmaf(s2, x6, b); x7 = __builtin_fmaf(s3, x7, b);
        } else if (MODE == 2) {     // cmp + cndmask pairs
            x0 = x0 > x4 ? x1 : x0; x1 = x1 > x5 ? x2 : x1;
            x2 = x2 > x6 ? x3 : x2; x3 = x3 > x7 ? x0 : x3;
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b);
            x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (MODE == 4) {     // packed fma: 8 FMAs in 4 instructions
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
            const f2 aa = {a, a}, bb = {b, b};
            p0 = __builtin_elementwise_fma(p0, aa, bb); p1 = __builtin_elementwise_fma(p1, aa, bb);
            p2 = __builtin_elementwise_fma(p2, aa, bb); p3 = __builtin_elementwise_fma(p3, aa, bb);
            x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
        } else if (MODE == 5) {     // LDS broadcast b128 x3 + 4 fma
            __shared__ float4 tab[192];
            if (i == 0) { tab[threadIdx.x % 192] = make_float4(x0, x1, x2, x3); __syncthreads(); }
            const int l = (i * 7) & 63;
            const float4 u = tab[3 * l], v = tab[3 * l + 1], w = tab[3 * l + 2];
            x4 = __builtin_fmaf(u.x, x4, v.x); x5 = __builtin_fmaf(u.y, x5, v.y);
            x6 = __builtin_fmaf(u.z, x6, w.x); x7 = __builtin_fmaf(u.w, x7, w.y);
        } else if (MODE == 6) {     // mul / add / max (non-FMA f32)
            x0 = x0 * a; x1 = x1 + b; x2 = fmaxf(x2, x3); x3 = x3 * a;
            x4 = x4 + b; x5 = fminf(x5, x6); x6 = x6 * a; x7 = x7 + b;
        } else if (MODE == 3) {     // min3 + fma
            x0 = fminf(fminf(x0, x1), x2) + x3;
            x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b);
            x3 = __builtin_fmaf(x3, a, b); x4 = fminf(fminf(x4, x5), x6) + x7;
            x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b);
            x7 = __builtin_fmaf(x7, a, b);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
void run(const char *name, int instrPerIter, int blocks)
{
    float *d;
    hipMalloc(&d, sizeof(float) * 256 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<blocks, 256>>>(d, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waveInstr = (double)blocks * 4 * iters * instrPerIter;
    printf("%-22s blocks=%5d  %.3f ms  %.1f G wave-instr/s  (%.2f cycles/instr/SIMD @2.4GHz)\n", name,
           blocks, ms, waveInstr / ms / 1e6, 1024.0 * 2.4e9 / (waveInstr / (ms * 1e-3)));
    hipFree(d);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz memclk=%d kHz L2=%d regs/blk=%d\n", p.name, p.multiProcessorCount,
           p.clockRate, p.memoryClockRate, p.l2CacheSize, p.regsPerBlock);
    for (int blocks : {2048, 4096}) {
        run<0>("fma x8", 8, blocks);
        run<1>("readlane x4 + fma x4", 8, blocks);
        run<2>("cmp+cndmask x4 + fma x4", 12, blocks);
        run<3>("min3/add + fma", 10, blocks);
        run<4>("pk_fma x4 (8 fma)", 4, blocks);
        run<5>("3x ds_read_b128 + 4 fma", 7, blocks);
        run<6>("mul/add/max x8", 8, blocks);
    }
    return 0;
}
