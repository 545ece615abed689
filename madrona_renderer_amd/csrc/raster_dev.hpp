// Device-side pieces shared by the raster kernels (raster.hip) and the BVH
// kernel (bvh.hip): the spec arithmetic of DESIGN.md section 3 (S1-S8), the
// packed-FMA plane records and the streaming stores.  Everything here is
// __forceinline__ device code in an anonymous namespace: each translation unit
// gets its own copy.  Compile with -ffp-contract=off: fused multiply-adds
// appear only where the spec writes fma().
#pragma once

#include <hip/hip_runtime.h>

#include "raster.hpp"

namespace mrx {
namespace {

constexpr int kWave = 64;
constexpr int kCold = 12;        // dwords: u/v planes, lit colour
constexpr int kRegionBlocks = 4; // a lane owns 4 consecutive pixels of a 32x8 region

// S2: every 3-term dot product is one rounded product and two fused steps
__device__ __forceinline__ float dot3(float ax, float ay, float az,
                                      float bx, float by, float bz)
{
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}

// S1
__device__ __forceinline__ void quatToMat(float w, float x, float y, float z,
                                          float R[3][3])
{
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, yy = y * y2, zz = z * z2;
    float xy = x * y2, xz = x * z2, yz = y * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    R[0][0] = 1.0f - (yy + zz); R[0][1] = xy - wz;          R[0][2] = xz + wy;
    R[1][0] = xy + wz;          R[1][1] = 1.0f - (xx + zz); R[1][2] = yz - wx;
    R[2][0] = xz - wy;          R[2][1] = yz + wx;          R[2][2] = 1.0f - (xx + yy);
}

__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ __forceinline__ uint32_t toU8(float c)
{
    c = fminf(fmaxf(c, 0.0f), 1.0f);
    return (uint32_t)__builtin_fmaf(c, 255.0f, 0.5f);
}

struct ViewConst {
    float Rc[3][3];
    float c[3];
    float lv[3];
};

// Edge planes (inside <=> all >= 0) and the 1/depth plane of one triangle, as
// functions of the storage pixel: value = fl(A*x + fl(B*y + C)).
struct TriPlanes {
    float A0, B0, C0, A1, B1, C1, A2, B2, C2, Dx, Dy, Dc;
    // conservative bounds of the covered storage pixels (+-inf when a vertex
    // is not safely in front of the eye); binning aid only, never decides a pixel
    float bbX0, bbX1, bbY0, bbY1;
};

// S2/S3 of one instance under one view, and the S6b quantities that depend on
// the pair only: model -> view transform, the eye in the instance's unscaled
// frame, the scale and its handedness.
struct InstXform {
    float MV[3][3], tv[3];
    float qo[3];         // q = Ri^T (c - t)
    float sc[3];
    float det;           // (s0 * s1) * s2: mirroring flips the winding
};

template <typename PARAMS>
__device__ __forceinline__ void instanceTransform(const PARAMS &p, const ViewConst &vc,
                                                  uint32_t i, InstXform &x)
{
    const float tx = p.instPos[3 * i + 0], ty = p.instPos[3 * i + 1],
                tz = p.instPos[3 * i + 2];
    const float4 q = *reinterpret_cast<const float4 *>(p.instRot + 4 * i);
    const float s0 = p.instScale[3 * i + 0], s1 = p.instScale[3 * i + 1],
                s2 = p.instScale[3 * i + 2];
    float Ri[3][3], M[3][3];
    quatToMat(q.x, q.y, q.z, q.w, Ri);
    x.sc[0] = s0; x.sc[1] = s1; x.sc[2] = s2;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            M[r][c] = Ri[r][c] * x.sc[c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            x.MV[r][c] = dot3(vc.Rc[0][r], vc.Rc[1][r], vc.Rc[2][r],
                              M[0][c], M[1][c], M[2][c]);
    const float dt[3] = { tx - vc.c[0], ty - vc.c[1], tz - vc.c[2] };
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        x.tv[r] = dot3(vc.Rc[0][r], vc.Rc[1][r], vc.Rc[2][r], dt[0], dt[1], dt[2]);
        x.qo[r] = -dot3(Ri[0][r], Ri[1][r], Ri[2][r], dt[0], dt[1], dt[2]);
    }
    x.det = (s0 * s1) * s2;
}

// S3-S7 for object triangle `tri` drawn by an instance whose transform is `x`.
// Planes are returned in registers; the shading record goes to `shade` (rgba,
// texture, objectID, world-local index) and `cold` (u/v planes [0..5] -- only
// written for textured triangles -- and lit colour [6..8]).  Returns validity.
// UVPLANES = false (BVH kernel): the u/v planes are left out and cold[0] holds
// |1/d| instead -- the caller derives them with uvPlanes() once it knows the
// triangle needs a record, instead of carrying twelve registers through its
// slot allocation.
// PARAMS: RasterParams, or any struct with the members read here (tris, triMats, s6bPad, sx, sz,
// ox, oz, transposed, diffuse, ambient) -- the BVH kernel passes a copy it reads from the
// kernel-argument segment batch by batch instead of holding the values in scalar registers.
template <bool UVPLANES = true, typename PARAMS = RasterParams>
__device__ __forceinline__ bool setupTriangleCore(const PARAMS &p, const float (&lv)[3],
                                                  const InstXform &x, uint32_t tri, int32_t obj,
                                                  int32_t kWorld, TriPlanes &out,
                                                  float *shade, float *cold)
{
    const float (&MV)[3][3] = x.MV;
    const float (&tv)[3] = x.tv;
    const float4 *src = reinterpret_cast<const float4 *>(p.tris + tri);
    const float4 t0 = src[0], t1 = src[1], t2 = src[2], t3 = src[3];
    const float4 *msrc = reinterpret_cast<const float4 *>(p.triMats + tri);
    const float4 mc = msrc[0], m1 = msrc[1], m2 = msrc[2];
    const int32_t tex = __float_as_int(m1.x);
    // S6b: is the eye outside the (padded) bounding box of the triangle's shell,
    // by more than the reach of the near plane?  The eye in the instance's
    // unscaled frame, q = Ri^T (c - t), against the box scaled by s (no division).
    bool cullBack = false, cullFront = false;
    {
        const float orient = m1.y;
        const float bmin[3] = { m1.z, m1.w, m2.x }, bmax[3] = { m2.y, m2.z, m2.w };
        bool outside = false;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float qo = x.qo[r];
            const float b0 = bmin[r] * x.sc[r], b1 = bmax[r] * x.sc[r];
            outside = outside || qo < fminf(b0, b1) - p.s6bPad || qo > fmaxf(b0, b1) + p.s6bPad;
        }
        const float handed = orient * x.det;   // mirroring flips the winding
        cullBack = outside && handed > 0.0f;
        cullFront = outside && handed < 0.0f;
    }
    const float op[9] = { t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x };
    const float uv[6] = { t2.y, t2.z, t2.w, t3.x, t3.y, t3.z };

    float P[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            P[j][r] = __builtin_fmaf(MV[r][2], op[3 * j + 2],
                                     __builtin_fmaf(MV[r][1], op[3 * j + 1],
                                                    __builtin_fmaf(MV[r][0], op[3 * j], tv[r])));

    float N[3][3], e1[3], e2[3], nn[3];
    cross3(P[1], P[2], N[0]);
    cross3(P[2], P[0], N[1]);
    cross3(P[0], P[1], N[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        e1[r] = P[1][r] - P[0][r];
        e2[r] = P[2][r] - P[0][r];
    }
    cross3(e1, e2, nn);
    const float d = dot3(nn[0], nn[1], nn[2], P[0][0], P[0][1], P[0][2]);
    // S6: degenerate / edge-on triangles; S6b: faces of a closed object turned
    // away from an eye outside the object can never be the nearest hit
    // (an instance whose ObjectID is negative this step is hidden: MRX_BUF_INSTANCE_OBJECT)
    const bool valid = obj >= 0 && fabsf(d) > 0.0f && !(cullBack && d > 0.0f) && !(cullFront && d < 0.0f);

    // Binning aid: pixel-space bounding box of the projected vertices, padded
    // by a pixel plus a relative margin that swallows the rounding of the
    // approximate reciprocals here and of the plane evaluation.
    {
        const float wmin = fminf(fminf(P[0][1], P[1][1]), P[2][1]);
        const float isx = __builtin_amdgcn_rcpf(p.sx), isz = __builtin_amdgcn_rcpf(p.sz);
        float fx[3], fz[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float iw = __builtin_amdgcn_rcpf(P[j][1]);
            fx[j] = (P[j][0] * iw - p.ox) * isx;   // image x in pixels
            fz[j] = (P[j][2] * iw - p.oz) * isz;   // image y in pixels
        }
        float x0 = fminf(fminf(fx[0], fx[1]), fx[2]), x1 = fmaxf(fmaxf(fx[0], fx[1]), fx[2]);
        float z0 = fminf(fminf(fz[0], fz[1]), fz[2]), z1 = fmaxf(fmaxf(fz[0], fz[1]), fz[2]);
        const float mx = 1.0f + 8e-6f * fmaxf(fabsf(x0), fabsf(x1));
        const float mz = 1.0f + 8e-6f * fmaxf(fabsf(z0), fabsf(z1));
        x0 -= mx; x1 += mx; z0 -= mz; z1 += mz;
        const bool ok = wmin > 1e-6f && (x1 - x0) < 3.0e38f && (z1 - z0) < 3.0e38f;
        const float inf = __builtin_inff();
        const bool trs = p.transposed != 0;
        out.bbX0 = ok ? (trs ? z0 : x0) : -inf;
        out.bbX1 = ok ? (trs ? z1 : x1) : inf;
        out.bbY0 = ok ? (trs ? x0 : z0) : -inf;
        out.bbY1 = ok ? (trs ? x1 : z1) : inf;
    }
    const float flip = d < 0.0f ? -1.0f : 1.0f;
    const bool tr = p.transposed != 0;

    float A[3], B[3], C[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float ax = N[e][0] * p.sx;
        const float az = N[e][2] * p.sz;
        const float cc = __builtin_fmaf(N[e][2], p.oz, __builtin_fmaf(N[e][0], p.ox, N[e][1]));
        A[e] = (tr ? az : ax) * flip;
        B[e] = (tr ? ax : az) * flip;
        C[e] = cc * flip;
    }
    out.A0 = A[0]; out.B0 = B[0]; out.C0 = C[0];
    out.A1 = A[1]; out.B1 = B[1]; out.C1 = C[1];
    out.A2 = A[2]; out.B2 = B[2]; out.C2 = C[2];
    const float rd = 1.0f / d;
    {
        const float ax = (nn[0] * p.sx) * rd;
        const float az = (nn[2] * p.sz) * rd;
        out.Dx = tr ? az : ax;
        out.Dy = tr ? ax : az;
        out.Dc = __builtin_fmaf(nn[2], p.oz, __builtin_fmaf(nn[0], p.ox, nn[1])) * rd;
    }
    // u/v planes (S8) are only ever read for textured triangles
    if (!UVPLANES) {
        cold[0] = fabsf(rd);
    } else if (tex >= 0) {
        const float rad = fabsf(rd);
        cold[0] = __builtin_fmaf(uv[4], A[2], __builtin_fmaf(uv[2], A[1], uv[0] * A[0])) * rad;
        cold[1] = __builtin_fmaf(uv[4], B[2], __builtin_fmaf(uv[2], B[1], uv[0] * B[0])) * rad;
        cold[2] = __builtin_fmaf(uv[4], C[2], __builtin_fmaf(uv[2], C[1], uv[0] * C[0])) * rad;
        cold[3] = __builtin_fmaf(uv[5], A[2], __builtin_fmaf(uv[3], A[1], uv[1] * A[0])) * rad;
        cold[4] = __builtin_fmaf(uv[5], B[2], __builtin_fmaf(uv[3], B[1], uv[1] * B[0])) * rad;
        cold[5] = __builtin_fmaf(uv[5], C[2], __builtin_fmaf(uv[3], C[1], uv[1] * C[0])) * rad;
    }

    // S7: flat two-sided Lambert
    const float len = sqrtf(dot3(nn[0], nn[1], nn[2], nn[0], nn[1], nn[2]));
    float ndl = dot3(nn[0], nn[1], nn[2], lv[0], lv[1], lv[2]) / len;
    if (d > 0.0f)
        ndl = -ndl;
    const float lit = __builtin_fmaf(p.diffuse, fmaxf(ndl, 0.0f), p.ambient);
    const float l0 = lit * mc.x, l1 = lit * mc.y, l2 = lit * mc.z;
    cold[6] = l0; cold[7] = l1; cold[8] = l2;
    const uint32_t rgba = toU8(l0) | (toU8(l1) << 8) | (toU8(l2) << 16) | 0xFF000000u;
    shade[0] = __uint_as_float(rgba);
    shade[1] = __int_as_float(tex);
    shade[2] = obj >= 0 ? mc.w : __int_as_float(obj);   // the triangle's own object id (TriMat alpha slot)
    shade[3] = __int_as_float(kWorld);
    return valid;
}

// S8 u/v planes of a triangle whose edge planes are `c` (as setupTriangleCore left them:
// transposed and flipped), `rad` = |1/d|, uv = the six texture coordinates of ObjTri:
// the same operations in the same order as in setupTriangleCore<true>.
__device__ __forceinline__ void uvPlanes(const TriPlanes &c, float rad, const float (&uv)[6], float *cold)
{
    cold[0] = __builtin_fmaf(uv[4], c.A2, __builtin_fmaf(uv[2], c.A1, uv[0] * c.A0)) * rad;
    cold[1] = __builtin_fmaf(uv[4], c.B2, __builtin_fmaf(uv[2], c.B1, uv[0] * c.B0)) * rad;
    cold[2] = __builtin_fmaf(uv[4], c.C2, __builtin_fmaf(uv[2], c.C1, uv[0] * c.C0)) * rad;
    cold[3] = __builtin_fmaf(uv[5], c.A2, __builtin_fmaf(uv[3], c.A1, uv[1] * c.A0)) * rad;
    cold[4] = __builtin_fmaf(uv[5], c.B2, __builtin_fmaf(uv[3], c.B1, uv[1] * c.B0)) * rad;
    cold[5] = __builtin_fmaf(uv[5], c.C2, __builtin_fmaf(uv[3], c.C1, uv[1] * c.C0)) * rad;
}

// The same for a draw-list entry: transform of the instance, then the triangle.
__device__ __forceinline__ bool setupTriangle(const RasterParams &p,
                                              const ViewConst &vc,
                                              WorldTri wt, int32_t kWorld,
                                              TriPlanes &out, float *shade, float *cold)
{
    InstXform x;
    instanceTransform(p, vc, wt.inst, x);
    return setupTriangleCore(p, vc.lv, x, wt.tri, p.instObj[wt.inst], kWorld, out, shade, cold);
}

// S8: nearest texel, repeat addressing, v up.
__device__ __forceinline__ uint32_t shadeTextured(const RasterParams &p,
                                                  const float *cold, int32_t tex,
                                                  float px, float py, float tt)
{
    const float u = __builtin_fmaf(cold[0], px, __builtin_fmaf(cold[1], py, cold[2])) * tt;
    const float v = __builtin_fmaf(cold[3], px, __builtin_fmaf(cold[4], py, cold[5])) * tt;
    const TexDesc td = p.textures[tex];
    const int tw = (int)td.width, th = (int)td.height;
    float uf = u - floorf(u);
    float vf = v - floorf(v);
    vf = 1.0f - vf;
    int tx = (int)(uf * (float)tw);
    int ty = (int)(vf * (float)th);
    tx = tx > tw - 1 ? tw - 1 : tx;
    ty = ty > th - 1 ? th - 1 : ty;
    tx = tx < 0 ? 0 : tx;
    ty = ty < 0 ? 0 : ty;
    const uint32_t texel = p.texels[td.offset + (uint32_t)ty * (uint32_t)tw + (uint32_t)tx];
    const uint32_t r = toU8(((float)(texel & 255u) * (1.0f / 255.0f)) * cold[6]);
    const uint32_t g = toU8(((float)((texel >> 8) & 255u) * (1.0f / 255.0f)) * cold[7]);
    const uint32_t b = toU8(((float)((texel >> 16) & 255u) * (1.0f / 255.0f)) * cold[8]);
    return r | (g << 8) | (b << 16) | 0xFF000000u;
}

// The kernel-argument block (RasterParams by value, 408 bytes + the hidden arguments: seven 64-byte lines) is a
// fresh copy for every launch, so every CU's first read of each of its lines misses the scalar cache -- and the
// compiler reads the arguments lazily, a few at a time, where the control flow first needs them: four or five
// s_load + s_waitcnt rounds in a row ahead of a kernel's first pose load, each a miss on a line not touched before.
// One dword of every line, requested together and waited for once at kernel entry, brings the whole block into
// the scalar cache up front; the lazy reads then hit.  Measured (profiles/r03_kernarg.txt): headline 22.56 -> 22.14
// us, 1024 worlds 9.46 -> 9.35, 2048 worlds 13.94 -> 13.60.
// BYTES = size of the explicit arguments: only lines that hold some of them are touched (a kernel that uses no hidden
// argument has none, and a read past the end of the block may leave the pool it was carved from).
template <size_t BYTES = sizeof(RasterParams)>
__device__ __forceinline__ void touchKernelArguments()
{
    static_assert(BYTES > 0x180 && BYTES <= 0x240, "seven to nine lines: adjust the offsets below");
    const __attribute__((address_space(4))) char *ka =
        (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7, t8;
    if (BYTES > 0x200)
        asm volatile("s_load_dword %0, %9, 0x0\n\ts_load_dword %1, %9, 0x40\n\ts_load_dword %2, %9, 0x80\n\t"
                     "s_load_dword %3, %9, 0xc0\n\ts_load_dword %4, %9, 0x100\n\ts_load_dword %5, %9, 0x140\n\t"
                     "s_load_dword %6, %9, 0x180\n\ts_load_dword %7, %9, 0x1c0\n\ts_load_dword %8, %9, 0x200\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8)
                     : "s"(ka));
    else if (BYTES > 0x1c0)
        asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\t"
                     "s_load_dword %3, %8, 0xc0\n\ts_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\t"
                     "s_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7) : "s"(ka));
    else
        asm volatile("s_load_dword %0, %7, 0x0\n\ts_load_dword %1, %7, 0x40\n\ts_load_dword %2, %7, 0x80\n\t"
                     "s_load_dword %3, %7, 0xc0\n\ts_load_dword %4, %7, 0x100\n\ts_load_dword %5, %7, 0x140\n\t"
                     "s_load_dword %6, %7, 0x180\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6) : "s"(ka));
}

__device__ __forceinline__ void waveLdsSync()
{
    // LDS hand-off between lanes of ONE wave: order the accesses, no s_barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed f32 FMA: two independent, individually rounded fmaf()s in one
// v_pk_fma_f32 (plain v_fma_f32 issues at half the packed rate on gfx950).
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c)
{
    return __builtin_elementwise_fma(a, b, c);
}

// Planes of one triangle as the raster loop wants them: pairs that share a
// multiplier sit in adjacent registers.  LDS layout of TileLds::planes[k]:
//   [0..3] A0 A1 A2 Dx   [4..7] B0 B1 B2 Dy   [8..11] C0 C1 C2 Dc   [12] mask
struct PlanePairs {
    f32x2 A01, A2D, B01, B2D, C01, C2D;
};

__device__ __forceinline__ PlanePairs loadPlanes(const float (*planes)[16], int k)
{
    const float4 *src = reinterpret_cast<const float4 *>(planes[k]);
    const float4 a = src[0], b = src[1], c = src[2];
    PlanePairs q;
    q.A01 = f32x2{ a.x, a.y }; q.A2D = f32x2{ a.z, a.w };
    q.B01 = f32x2{ b.x, b.y }; q.B2D = f32x2{ b.z, b.w };
    q.C01 = f32x2{ c.x, c.y }; q.C2D = f32x2{ c.z, c.w };
    return q;
}

// Output stores are write-through (agent scope, `sc1`): the images are written
// once and never read back by this kernel, and lines left dirty in the XCDs'
// L2s would have to be written back at the end of the kernel, where nothing
// overlaps it (scripts/micro/store_modes.hip: 3.1 us -> 0.9 us between
// back-to-back 128 MiB launches).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void streamStore16(uint32_t writeThrough, void *dst,
                                              uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    const u32x4 v = { a, b, c, d };
    // s_nop 1: on gfx940+ a VALU write to the data registers of a >64-bit store
    // needs two wait states after it; the compiler cannot see into the asm to
    // insert them (one is not enough: dword 2 of the data was overwritten)
    if (writeThrough)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
    else
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
}
// Reads of output pixels this same lane stored earlier in the kernel (the BVH kernel's
// multi-round resolve): agent scope, so they are served by the L2 the write-through
// stores went to, and waited for here (the compiler does not track asm loads).
__device__ __forceinline__ u32x4 streamLoad16(const void *src)
{
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t streamLoad4(const void *src)
{
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src) : "memory");
    return v;
}
__device__ __forceinline__ void streamStore4(uint32_t writeThrough, void *dst, uint32_t a)
{
    if (writeThrough)
        asm volatile("global_store_dword %0, %1, off sc1" :: "v"(dst), "v"(a) : "memory");
    else
        asm volatile("global_store_dword %0, %1, off" :: "v"(dst), "v"(a) : "memory");
}

}  // namespace
}  // namespace mrx
