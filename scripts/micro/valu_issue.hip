// Calibration: what does one wave64 VALU instruction cost a SIMD, by waves per
// SIMD?  MI355X_MICROARCH.md says 2 cycles with several waves per SIMD (4 for one wave alone); the
// round-1 valu_rate.hip derived 4.3 "cycles" from wall time at an ASSUMED 2.4 GHz.  Here the
// kernel reads the shader clock itself (s_memtime = shader cycles per the guide) around a block of
// independent FMAs, so the figure does not depend on what clock the chip held, and the wall time
// of the same launch gives that clock.
//   W waves per SIMD: 256 CUs x W workgroups of 256 threads (4 waves = one per SIMD).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float a, float b)
{
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    float x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == 0) {            // 8 independent v_fma_f32
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b);
                x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b);
                x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            } else if (MODE == 1) {     // 4 v_pk_fma_f32 (8 FMAs)
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
                const f2 aa = {a, a}, bb = {b, b};
                p0 = __builtin_elementwise_fma(p0, aa, bb); p1 = __builtin_elementwise_fma(p1, aa, bb);
                p2 = __builtin_elementwise_fma(p2, aa, bb); p3 = __builtin_elementwise_fma(p3, aa, bb);
                x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
            } else if (MODE == 2) {     // integer / compare mix as in the BVH kernel: add, and, cmp+cndmask, max
                unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
                u0 = u0 + u1; u1 = u1 & 0x7fffffffu;
                x0 = __uint_as_float(u0); x1 = __uint_as_float(u1);
                x2 = x2 > x6 ? x3 : x2; x3 = fmaxf(x3, x7);
                x4 = x4 * a; x5 = x5 + b;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if ((threadIdx.x & 63) == 0)
        cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
int run(const char *name, int instrPerUnroll, int wavesPerSimd, int cus)
{
    const int blocks = cus * wavesPerSimd, iters = 4000;
    float *d;
    unsigned long long *c;
    CK(hipMalloc(&d, sizeof(float) * 256 * blocks));
    CK(hipMalloc(&c, 8 * 4 * blocks));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w)
        k<MODE><<<blocks, 256>>>(d, c, iters, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<MODE><<<blocks, 256>>>(d, c, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(4 * blocks);
    CK(hipMemcpy(h.data(), c, 8 * h.size(), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double instr = (double)iters * 4 * instrPerUnroll;           // per wave
    // a SIMD serves its W waves together for (about) the median wave's span
    const double cycPerInstrPerSimd = med / (instr * wavesPerSimd);
    const double wallRate = (double)blocks * 4 * instr / (ms * 1e-3);  // wave-instr/s, whole chip
    printf("%-28s W=%d  wave span %9.0f ticks (median)  -> %.2f ticks per wave64 instr per SIMD;  wall %.3f ms -> "
           "%.1f G wave-instr/s chip-wide, tick rate %.2f GHz if ticks are shader cycles\n",
           name, wavesPerSimd, med, cycPerInstrPerSimd, ms, wallRate / 1e9, med / (ms * 1e-3) / 1e9);
    (void)hipFree(d); (void)hipFree(c);
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s CUs=%d clockRate=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    for (int w : {1, 2, 4, 8}) {
        if (run<0>("v_fma_f32 x8", 8, w, p.multiProcessorCount)) return 1;
        if (run<1>("v_pk_fma_f32 x4", 4, w, p.multiProcessorCount)) return 1;
        if (run<2>("add/and/cmp+cndmask/max/mul", 7, w, p.multiProcessorCount)) return 1;
    }
    return 0;
}
