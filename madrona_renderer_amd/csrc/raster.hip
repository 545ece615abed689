// Tiled software rasterizer / primary-ray caster for gfx950 (MI355X).
//
// One wavefront renders one 64x64 tile of one view, start to finish:
//   S  setup     lane = triangle: pose -> view-space vertices -> edge / depth /
//                attribute planes and the flat-shaded colour, into LDS
//   R  raster    lane = pixel of an 8x8 block; the wave walks the 16 blocks of
//                a 64x16 band per triangle (triangle coefficients broadcast
//                from LDS), z / winner kept in VGPRs
//   O  output    shade the winner, 1/depth -> depth, store
// There is no inter-wave communication, so no workgroup barrier anywhere; a
// workgroup is just four independent waves sharing an LDS allocation.
//
// The arithmetic restates DESIGN.md section 3 (S0-S9) op for op; the file is
// compiled with -ffp-contract=off and fmaf appears only where the spec says so.
// The CPU oracle (oracle/raster_oracle.c) is a separate restatement of the
// same spec and is never linked here.
#include <hip/hip_runtime.h>

#include "raster.hpp"

namespace mrx {
namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kChunk = 64;       // triangles set up per pass (one per lane)
constexpr int kHot = 16;         // dwords: edges, depth plane, rgba, tex, seg, k
constexpr int kCold = 12;        // dwords: u/v planes, lit colour
constexpr int kBandRows = 16;    // a band is 64 x 16 pixels = 16 blocks of 8x8
constexpr int kBlocksPerBand = 16;

struct WaveLds {
    float hot[kChunk][kHot];
    float cold[kChunk][kCold];
};

__device__ __forceinline__ float dot3(float ax, float ay, float az,
                                      float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
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
    return (uint32_t)(c * 255.0f + 0.5f);
}

struct ViewConst {
    float Rc[3][3];
    float c[3];
    float lv[3];
};

// S3-S7 for one world-triangle.  Writes the LDS record, returns validity.
__device__ __forceinline__ bool setupTriangle(const RasterParams &p,
                                              const ViewConst &vc,
                                              WorldTri wt, int32_t kWorld,
                                              float *hot, float *cold)
{
    const uint32_t i = wt.inst;
    const float tx = p.instPos[3 * i + 0], ty = p.instPos[3 * i + 1],
                tz = p.instPos[3 * i + 2];
    const float4 q = *reinterpret_cast<const float4 *>(p.instRot + 4 * i);
    const float s0 = p.instScale[3 * i + 0], s1 = p.instScale[3 * i + 1],
                s2 = p.instScale[3 * i + 2];
    const int32_t obj = p.instObj[i];

    float Ri[3][3], M[3][3], MV[3][3], tv[3];
    quatToMat(q.x, q.y, q.z, q.w, Ri);
    const float sc[3] = { s0, s1, s2 };
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            M[r][c] = Ri[r][c] * sc[c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            MV[r][c] = dot3(vc.Rc[0][r], vc.Rc[1][r], vc.Rc[2][r],
                            M[0][c], M[1][c], M[2][c]);
    const float dt[3] = { tx - vc.c[0], ty - vc.c[1], tz - vc.c[2] };
#pragma unroll
    for (int r = 0; r < 3; ++r)
        tv[r] = dot3(vc.Rc[0][r], vc.Rc[1][r], vc.Rc[2][r], dt[0], dt[1], dt[2]);

    const float4 *src = reinterpret_cast<const float4 *>(p.tris + wt.tri);
    const float4 t0 = src[0], t1 = src[1], t2 = src[2], t3 = src[3];
    const float op[9] = { t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x };
    const float uv[6] = { t2.y, t2.z, t2.w, t3.x, t3.y, t3.z };
    const int32_t mat = __float_as_int(t3.w);

    float P[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 3; ++r)
            P[j][r] = dot3(MV[r][0], MV[r][1], MV[r][2],
                           op[3 * j], op[3 * j + 1], op[3 * j + 2]) + tv[r];

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
    const bool valid = fabsf(d) > 0.0f;                       // S6
    const float flip = d < 0.0f ? -1.0f : 1.0f;
    const bool tr = p.transposed != 0;

    float A[3], B[3], C[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const float ax = N[e][0] * p.sx;
        const float az = N[e][2] * p.sz;
        const float cc = (N[e][0] * p.ox + N[e][1]) + N[e][2] * p.oz;
        A[e] = (tr ? az : ax) * flip;
        B[e] = (tr ? ax : az) * flip;
        C[e] = cc * flip;
        hot[3 * e + 0] = A[e];
        hot[3 * e + 1] = B[e];
        hot[3 * e + 2] = C[e];
    }
    const float rd = 1.0f / d;
    {
        const float ax = (nn[0] * p.sx) * rd;
        const float az = (nn[2] * p.sz) * rd;
        hot[9] = tr ? az : ax;
        hot[10] = tr ? ax : az;
        hot[11] = ((nn[0] * p.ox + nn[1]) + nn[2] * p.oz) * rd;
    }
    const float rad = fabsf(rd);
    cold[0] = ((uv[0] * A[0] + uv[2] * A[1]) + uv[4] * A[2]) * rad;
    cold[1] = ((uv[0] * B[0] + uv[2] * B[1]) + uv[4] * B[2]) * rad;
    cold[2] = ((uv[0] * C[0] + uv[2] * C[1]) + uv[4] * C[2]) * rad;
    cold[3] = ((uv[1] * A[0] + uv[3] * A[1]) + uv[5] * A[2]) * rad;
    cold[4] = ((uv[1] * B[0] + uv[3] * B[1]) + uv[5] * B[2]) * rad;
    cold[5] = ((uv[1] * C[0] + uv[3] * C[1]) + uv[5] * C[2]) * rad;

    // S7: flat two-sided Lambert
    const float len = sqrtf(dot3(nn[0], nn[1], nn[2], nn[0], nn[1], nn[2]));
    float ndl = dot3(nn[0], nn[1], nn[2], vc.lv[0], vc.lv[1], vc.lv[2]) / len;
    if (d > 0.0f)
        ndl = -ndl;
    const float lit = p.ambient + p.diffuse * fmaxf(ndl, 0.0f);
    float col[3] = { p.defaultColor[0], p.defaultColor[1], p.defaultColor[2] };
    int32_t tex = -1;
    if (mat >= 0 && (uint32_t)mat < p.numMaterials) {
        const float4 mc = *reinterpret_cast<const float4 *>(p.materials[mat].color);
        col[0] = mc.x; col[1] = mc.y; col[2] = mc.z;
        tex = p.materials[mat].tex;
    }
    if (tex < 0 || (uint32_t)tex >= p.numTextures)
        tex = -1;
    const float l0 = lit * col[0], l1 = lit * col[1], l2 = lit * col[2];
    cold[6] = l0; cold[7] = l1; cold[8] = l2;
    const uint32_t rgba = toU8(l0) | (toU8(l1) << 8) | (toU8(l2) << 16) | 0xFF000000u;
    hot[12] = __uint_as_float(rgba);
    hot[13] = __int_as_float(tex);
    hot[14] = __int_as_float(obj);
    hot[15] = __int_as_float(kWorld);
    return valid;
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

// ---------------------------------------------------------------------------
// Shared pieces of the per-wave tile loop
// ---------------------------------------------------------------------------
__device__ __forceinline__ void waveLdsSync()
{
    // LDS hand-off between lanes of ONE wave: order the accesses, no s_barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct TileCtx {
    uint32_t view, tileX0, tileY0;
    uint32_t triBegin, numTris;
    int lx, ly;
};

__device__ __forceinline__ bool tileSetup(const RasterParams &p, int wave, int lane,
                                          TileCtx &t, ViewConst &vc)
{
    const uint32_t tilesPerView = p.tilesFast * p.tilesSlow;
    const uint32_t item = blockIdx.x * kWavesPerBlock + wave;
    if (item >= p.numViews * tilesPerView)
        return false;
    t.view = item / tilesPerView;
    const uint32_t tile = item % tilesPerView;
    t.tileX0 = (tile % p.tilesFast) * 64u;
    t.tileY0 = (tile / p.tilesFast) * 64u;
    const float4 q = *reinterpret_cast<const float4 *>(p.camRot + 4 * t.view);
    quatToMat(q.x, q.y, q.z, q.w, vc.Rc);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        vc.c[r] = p.camPos[3 * t.view + r];
        vc.lv[r] = dot3(vc.Rc[0][r], vc.Rc[1][r], vc.Rc[2][r],
                        p.toLight[0], p.toLight[1], p.toLight[2]);
    }
    const uint32_t world = p.viewWorld[t.view];
    t.triBegin = p.worldTriStart[world];
    t.numTris = p.worldTriStart[world + 1] - t.triBegin;
    t.lx = lane & 7;
    t.ly = lane >> 3;
    return true;
}

// S for one chunk of up to 64 world-triangles; returns the valid-lane mask.
__device__ __forceinline__ uint64_t setupChunk(const RasterParams &p, const ViewConst &vc,
                                               const TileCtx &t, uint32_t chunk, int lane,
                                               WaveLds &L)
{
    bool valid = false;
    const uint32_t k = chunk + lane;
    if (k < t.numTris) {
        const WorldTri wt = p.worldTris[t.triBegin + k];
        valid = setupTriangle(p, vc, wt, (int32_t)k, L.hot[lane], L.cold[lane]);
    }
    const uint64_t mask = __ballot(valid);
    waveLdsSync();
    return mask;
}

// Winner lookup + shading of one pixel (lane) of block b.
template <bool IDS>
__device__ __forceinline__ void resolvePixel(const RasterParams &p, const WaveLds &L,
                                             int32_t w, float bestInv, float px, float py,
                                             uint32_t &rgba, int32_t &id)
{
    const float *h = L.hot[w];
    rgba = __float_as_uint(h[12]);
    const int32_t tex = __float_as_int(h[13]);
    if (tex >= 0) {
        const float tt = 1.0f / bestInv;
        rgba = shadeTextured(p, L.cold[w], tex, px, py, tt);
    }
    if (IDS)
        id = __float_as_int(p.idsAreSegmask ? h[14] : h[15]);
}

// ---------------------------------------------------------------------------
// Variant 1 ("brute"): every valid triangle is tested at every pixel.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void rasterBandBrute(const WaveLds &L, uint64_t validMask,
                                                const float (&pxf)[8], float py0, float py1,
                                                float invNear,
                                                float (&best)[kBlocksPerBand],
                                                int32_t (&bid)[kBlocksPerBand])
{
    for (uint64_t m = validMask; m != 0; m &= m - 1) {
        const int k = __builtin_ctzll(m);
        const float *h = L.hot[k];
        const float A0 = h[0], B0 = h[1], C0 = h[2];
        const float A1 = h[3], B1 = h[4], C1 = h[5];
        const float A2 = h[6], B2 = h[7], C2 = h[8];
        const float Dx = h[9], Dy = h[10], Dc = h[11];
        float r0[2], r1[2], r2[2], rd[2];
        r0[0] = __builtin_fmaf(B0, py0, C0); r0[1] = __builtin_fmaf(B0, py1, C0);
        r1[0] = __builtin_fmaf(B1, py0, C1); r1[1] = __builtin_fmaf(B1, py1, C1);
        r2[0] = __builtin_fmaf(B2, py0, C2); r2[1] = __builtin_fmaf(B2, py1, C2);
        rd[0] = __builtin_fmaf(Dy, py0, Dc); rd[1] = __builtin_fmaf(Dy, py1, Dc);
#pragma unroll
        for (int b = 0; b < kBlocksPerBand; ++b) {
            const int r = b >> 3, bx = b & 7;
            const float e0 = __builtin_fmaf(A0, pxf[bx], r0[r]);
            const float e1 = __builtin_fmaf(A1, pxf[bx], r1[r]);
            const float e2 = __builtin_fmaf(A2, pxf[bx], r2[r]);
            const float it = __builtin_fmaf(Dx, pxf[bx], rd[r]);
            const bool in = (fminf(fminf(e0, e1), e2) >= 0.0f) &&
                            (it > best[b]) && (it <= invNear);
            best[b] = in ? it : best[b];
            bid[b] = in ? k : bid[b];
        }
    }
}

template <bool IDS, bool MULTI>
__global__ __launch_bounds__(kWave *kWavesPerBlock)
void rasterBruteKernel(const RasterParams p)
{
    __shared__ WaveLds lds[kWavesPerBlock];
    const int wave = threadIdx.x / kWave;
    const int lane = threadIdx.x % kWave;
    TileCtx t;
    ViewConst vc;
    if (!tileSetup(p, wave, lane, t, vc))
        return;
    WaveLds &L = lds[wave];

    float pxf[8];
#pragma unroll
    for (int bx = 0; bx < 8; ++bx)
        pxf[bx] = (float)(t.tileX0 + bx * 8 + t.lx);
    const float invNear = p.invNear;

    uint64_t mask0 = 0;
    if (!MULTI)
        mask0 = setupChunk(p, vc, t, 0, lane, L);

    for (int band = 0; band < 4; ++band) {
        float best[kBlocksPerBand];
        int32_t bid[kBlocksPerBand];
        uint32_t outRgba[kBlocksPerBand];
        int32_t outId[kBlocksPerBand];
#pragma unroll
        for (int b = 0; b < kBlocksPerBand; ++b) {
            best[b] = p.invFar;
            bid[b] = -1;
            outRgba[b] = 0xFF000000u;
            outId[b] = -1;
        }
        const float py0 = (float)(t.tileY0 + band * kBandRows + t.ly);
        const float py1 = (float)(t.tileY0 + band * kBandRows + 8 + t.ly);

        if (!MULTI) {
            rasterBandBrute(L, mask0, pxf, py0, py1, invNear, best, bid);
#pragma unroll
            for (int b = 0; b < kBlocksPerBand; ++b)
                if (bid[b] >= 0)
                    resolvePixel<IDS>(p, L, bid[b], best[b], pxf[b & 7],
                                      (b >> 3) ? py1 : py0, outRgba[b], outId[b]);
        } else {
            for (uint32_t chunk = 0; chunk < t.numTris; chunk += kChunk) {
                const uint64_t mask = setupChunk(p, vc, t, chunk, lane, L);
                rasterBandBrute(L, mask, pxf, py0, py1, invNear, best, bid);
                // resolve this chunk's winners before its records are replaced
#pragma unroll
                for (int b = 0; b < kBlocksPerBand; ++b) {
                    if (bid[b] >= 0)
                        resolvePixel<IDS>(p, L, bid[b], best[b], pxf[b & 7],
                                          (b >> 3) ? py1 : py0, outRgba[b], outId[b]);
                    bid[b] = -1;
                }
                waveLdsSync();
            }
        }

        // ---- O: output the band
#pragma unroll
        for (int b = 0; b < kBlocksPerBand; ++b) {
            const int r = b >> 3, bx = b & 7;
            const uint32_t fx = t.tileX0 + bx * 8 + t.lx;
            const uint32_t fy = t.tileY0 + band * kBandRows + r * 8 + t.ly;
            if (fx < p.nfast && fy < p.nslow) {
                const bool hit = best[b] > p.invFar;
                const float dep = hit ? 1.0f / best[b] : 0.0f;
                const size_t o = ((size_t)t.view * p.nslow + fy) * p.nfast + fx;
                p.rgb[o] = outRgba[b];
                p.depth[o] = dep;
                if (IDS)
                    p.ids[o] = outId[b];
            }
        }
    }
}

}  // namespace

hipError_t launchRaster(const RasterParams &p, uint32_t maxWorldTris,
                        int32_t variant, hipStream_t stream)
{
    (void)variant;
    const uint32_t items = p.numViews * p.tilesFast * p.tilesSlow;
    if (items == 0)
        return hipSuccess;
    const dim3 grid((items + kWavesPerBlock - 1) / kWavesPerBlock);
    const dim3 block(kWave * kWavesPerBlock);
    const bool ids = p.ids != nullptr;
    const bool multi = maxWorldTris > (uint32_t)kChunk;
    if (ids) {
        if (multi) rasterBruteKernel<true, true><<<grid, block, 0, stream>>>(p);
        else       rasterBruteKernel<true, false><<<grid, block, 0, stream>>>(p);
    } else {
        if (multi) rasterBruteKernel<false, true><<<grid, block, 0, stream>>>(p);
        else       rasterBruteKernel<false, false><<<grid, block, 0, stream>>>(p);
    }
    return hipGetLastError();
}

}  // namespace mrx
