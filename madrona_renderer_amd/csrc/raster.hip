// Tiled software rasterizer / primary-ray caster for gfx950 (MI355X).
//
// Three kernels share the spec arithmetic below:
//   rasterGroupKernel    the production path for worlds of <= 256 triangles:
//                        one workgroup per group of views / tiles, setup per
//                        view, register/LDS binning, packed-FMA raster,
//                        16-byte write-through stores
//   rasterChunkedKernel  worlds of more than 256 triangles: one workgroup per
//                        tile, triangles through LDS in passes of 256
//   rasterBruteKernel    the round's first kernel (every triangle at every
//                        pixel), kept as an on-device cross-check (variant 1)
// Phases: S setup (lane = triangle: pose -> view-space vertices -> edge /
// 1-over-depth / attribute planes, flat colour), binning (lane = triangle,
// regions of the tile), R raster (lane = pixels, per-pixel z and winner in
// VGPRs, triangle planes broadcast from LDS), O output (shade, 1/depth, store).
//
// The file is compiled with -ffp-contract=off; fused multiply-adds appear only
// where the spec writes fma().  The CPU oracle (oracle/raster_oracle.c) is a
// separate restatement of the same spec and is never linked here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>

#include "raster.hpp"
#include "raster_dev.hpp"

namespace mrx {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kChunk = 64;       // triangles set up per pass (one per lane)
constexpr int kHot = 16;         // dwords: edges, depth plane, rgba, tex, seg, k
constexpr int kBandRows = 16;    // a band is 64 x 16 pixels = 16 blocks of 8x8
constexpr int kBlocksPerBand = 16;

// Per-wave LDS: the shading record of each triangle of the chunk.  The brute
// variant also parks the planes here (hot[0..11]); the strip variant keeps
// them in registers and uses the compact layout.
struct WaveLds {
    float hot[kChunk][kHot];     // [0..11] planes (brute only), [12..15] shade
    float cold[kChunk][kCold];
};

// ---------------------------------------------------------------------------
// Shared pieces of the per-wave tile loop
// ---------------------------------------------------------------------------
struct TileCtx {
    uint32_t view, tileX0, tileY0;
    uint32_t triBegin, numTris;
    int lx, ly;
};

__device__ __forceinline__ bool tileSetup(const RasterParams &p, uint32_t item, int lane,
                                          TileCtx &t, ViewConst &vc)
{
    const uint32_t tilesPerView = p.tilesFast * p.tilesSlow;
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
    // wave-uniform: park the view constants in SGPRs
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            vc.Rc[r][cc] = __uint_as_float(
                __builtin_amdgcn_readfirstlane(__float_as_uint(vc.Rc[r][cc])));
        vc.c[r] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(vc.c[r])));
        vc.lv[r] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(vc.lv[r])));
    }
    t.triBegin = t.view * p.viewTriStride;
    t.numTris = p.viewTriCount[t.view];
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
        const WorldTri wt = p.viewTris[t.triBegin + k];
        TriPlanes c;
        float *h = L.hot[lane];
        valid = setupTriangle(p, vc, wt, (int32_t)k, c, h + 12, L.cold[lane]);
        h[0] = c.A0; h[1] = c.B0; h[2] = c.C0;
        h[3] = c.A1; h[4] = c.B1; h[5] = c.C1;
        h[6] = c.A2; h[7] = c.B2; h[8] = c.C2;
        h[9] = c.Dx; h[10] = c.Dy; h[11] = c.Dc;
    }
    const uint64_t mask = __ballot(valid);
    waveLdsSync();
    return mask;
}

// Winner lookup + shading of one pixel (lane) of block b.
__device__ __forceinline__ const float *shadeRec(const WaveLds &L, int32_t w) { return &L.hot[w][12]; }

template <bool IDS, typename LDS>
__device__ __forceinline__ void resolvePixel(const RasterParams &p, const LDS &L,
                                             int32_t w, float bestInv, float px, float py,
                                             uint32_t &rgba, int32_t &id)
{
    const float *h = shadeRec(L, w);
    rgba = __float_as_uint(h[0]);
    const int32_t tex = __float_as_int(h[1]);
    if (tex >= 0) {
        const float tt = 1.0f / bestInv;
        rgba = shadeTextured(p, L.cold[w], tex, px, py, tt);
    }
    if (IDS)
        id = __float_as_int(p.idsAreSegmask ? h[2] : h[3]);
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
    touchKernelArguments();
    const int wave = threadIdx.x / kWave;
    const int lane = threadIdx.x % kWave;
    TileCtx t;
    ViewConst vc;
    if (!tileSetup(p, blockIdx.x * kWavesPerBlock + wave, lane, t, vc))
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

// Bits 0..15: the tile's 32x8 regions (bit 2*strip + half) the lane's triangle
// can touch.  Bit 16: the 1/depth plane stays <= invNear over the whole tile,
// so the per-pixel near test can be dropped for this triangle.
constexpr uint32_t kNearFree = 1u << 16;

// ---------------------------------------------------------------------------
// Wave-level binning.  The lane that set a triangle up classifies the tile's
// sixteen 32x8-pixel regions against it: a region is dropped when one edge
// plane is negative, or the 1/depth plane is outside (invFar, invNear], at the
// region pixel where that plane is largest / smallest, or when it misses the
// triangle's conservative bounding box.  The planes are evaluated as
// fl(A*x + fl(B*y + C)), monotone in x and in y, so the extreme over a region
// is the value at one of its corner pixels and dropping a region cannot change
// any pixel: binning only skips work.
// ---------------------------------------------------------------------------
// Region bits of strips [S0, S1) for the lane's triangle; `nearOk` reports
// whether the 1/depth plane stays <= invNear over those strips.
template <int S0, int S1>
__device__ __forceinline__ uint32_t classifyStrips(const TriPlanes &c, uint32_t tileX0,
                                                   uint32_t tileY0, float invNear, float invFar,
                                                   bool &nearOk)
{
    const float X0 = (float)tileX0, Y0 = (float)tileY0;
    // x / y of the region pixel where each plane is largest (depth: also smallest)
    const float xa0 = c.A0 < 0.0f ? X0 : X0 + 31.0f;
    const float xa1 = c.A1 < 0.0f ? X0 : X0 + 31.0f;
    const float xa2 = c.A2 < 0.0f ? X0 : X0 + 31.0f;
    const float xdh = c.Dx < 0.0f ? X0 : X0 + 31.0f;
    const float xdl = c.Dx < 0.0f ? X0 + 31.0f : X0;
    const float yb0 = c.B0 < 0.0f ? Y0 : Y0 + 7.0f;
    const float yb1 = c.B1 < 0.0f ? Y0 : Y0 + 7.0f;
    const float yb2 = c.B2 < 0.0f ? Y0 : Y0 + 7.0f;
    const float ydh = c.Dy < 0.0f ? Y0 : Y0 + 7.0f;
    const float ydl = c.Dy < 0.0f ? Y0 + 7.0f : Y0;
    uint32_t mask = 0;
    float dmax = -__builtin_inff();
#pragma unroll
    for (int s = S0; s < S1; ++s) {
        const float dy = (float)(8 * s);
        const float r0 = __builtin_fmaf(c.B0, yb0 + dy, c.C0);
        const float r1 = __builtin_fmaf(c.B1, yb1 + dy, c.C1);
        const float r2 = __builtin_fmaf(c.B2, yb2 + dy, c.C2);
        const float rh = __builtin_fmaf(c.Dy, ydh + dy, c.Dc);
        const float rl = __builtin_fmaf(c.Dy, ydl + dy, c.Dc);
        const bool yin = (Y0 + dy + 7.0f >= c.bbY0) && (Y0 + dy <= c.bbY1);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const float dx = (float)(32 * hf);
            const float e0 = __builtin_fmaf(c.A0, xa0 + dx, r0);
            const float e1 = __builtin_fmaf(c.A1, xa1 + dx, r1);
            const float e2 = __builtin_fmaf(c.A2, xa2 + dx, r2);
            const float dh = __builtin_fmaf(c.Dx, xdh + dx, rh);
            const float dl = __builtin_fmaf(c.Dx, xdl + dx, rl);
            const bool xin = (X0 + dx + 31.0f >= c.bbX0) && (X0 + dx <= c.bbX1);
            const bool possible = (fminf(fminf(e0, e1), e2) >= 0.0f) &&
                                  (dh > invFar) && (dl <= invNear) && xin && yin;
            mask |= possible ? (1u << (2 * s + hf)) : 0u;
            dmax = fmaxf(dmax, dh);
        }
    }
    nearOk = dmax <= invNear;
    return mask;
}

__device__ __forceinline__ uint32_t classifyRegions(const TriPlanes &c, const TileCtx &t,
                                                    float invNear, float invFar)
{
    bool nearOk;
    uint32_t mask = classifyStrips<0, 8>(c, t.tileX0, t.tileY0, invNear, invFar, nearOk);
    if (nearOk)
        mask |= kNearFree;
    return mask;
}


// One pixel of the lane against one triangle.  The three lane predicates are
// combined on the scalar unit (s_and_b64), which runs beside the vector ALU
// that bounds this kernel.
template <bool NEAR>
__device__ __forceinline__ void pixelTest(const PlanePairs &q, f32x2 r01, f32x2 r2d, float px,
                                          float invNear, int32_t kv, float &best, int32_t &bid)
{
    const f32x2 pp = { px, px };
    const f32x2 e01 = fma2(q.A01, pp, r01);       // e0, e1
    const f32x2 e2d = fma2(q.A2D, pp, r2d);       // e2, 1/depth
    bool in = (fminf(fminf(e01.x, e01.y), e2d.x) >= 0.0f) & (e2d.y > best);
    if (NEAR)
        in = in & (e2d.y <= invNear);
    best = in ? e2d.y : best;
    bid = in ? kv : bid;
}

// Rasterise one 32x8 region against the triangles in `act` (bit k = triangle
// k of the chunk survives classification for this region), in triangle order.
// Triangle planes are broadcast from LDS (all lanes read the same 48 bytes).
template <bool NEAR, int IDSHIFT>
__device__ __forceinline__ void rasterRegion(const float (*planes)[16], uint64_t act, int recBase,
                                             const float (&px)[kRegionBlocks], float py,
                                             float invNear,
                                             float (&best)[kRegionBlocks],
                                             int32_t (&bid)[kRegionBlocks])
{
    const f32x2 yy = { py, py };
    for (; act != 0; act &= act - 1) {
        const int k = recBase + __builtin_ctzll(act);   // record index in the LDS tables
        const PlanePairs q = loadPlanes(planes, k);
        const f32x2 r01 = fma2(q.B01, yy, q.C01);
        const f32x2 r2d = fma2(q.B2D, yy, q.C2D);
        // the winner is tracked as k << IDSHIFT (group kernel: the byte offset
        // of its shading record)
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b)
            pixelTest<NEAR>(q, r01, r2d, px[b], invNear, k << IDSHIFT, best[b], bid[b]);
    }
}

// Store a region whose pixels are already shaded (chunked kernel).  A lane owns
// four consecutive pixels of one row (pixel b of the lane is x = fx0 + b), so
// the common case is one 16-byte store per output tensor per lane.
template <bool IDS>
__device__ __forceinline__ void outputRegion(const RasterParams &p, const TileCtx &t,
                                             uint32_t fx0, uint32_t fy, float invFar,
                                             const float (&best)[kRegionBlocks],
                                             const uint32_t (&rgba)[kRegionBlocks],
                                             const int32_t (&id)[kRegionBlocks])
{
    float dep[kRegionBlocks];
#pragma unroll
    for (int b = 0; b < kRegionBlocks; ++b)
        dep[b] = best[b] > invFar ? __builtin_amdgcn_rcpf(best[b]) : 0.0f;
    if (fy >= p.nslow || (p.debugSkip & 1u))
        return;
    const size_t o = ((size_t)t.view * p.nslow + fy) * p.nfast + fx0;
    if ((p.nfast & 3u) == 0 && fx0 + 3 < p.nfast) {
        streamStore16(p.writeThrough, p.rgb + o, rgba[0], rgba[1], rgba[2], rgba[3]);
        streamStore16(p.writeThrough, p.depth + o, __float_as_uint(dep[0]), __float_as_uint(dep[1]),
                      __float_as_uint(dep[2]), __float_as_uint(dep[3]));
        if (IDS)
            streamStore16(p.writeThrough, p.ids + o, (uint32_t)id[0], (uint32_t)id[1], (uint32_t)id[2], (uint32_t)id[3]);
    } else {
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b) {
            if (fx0 + b < p.nfast) {
                streamStore4(p.writeThrough, p.rgb + o + b, rgba[b]);
                streamStore4(p.writeThrough, p.depth + o + b, __float_as_uint(dep[b]));
                if (IDS)
                    streamStore4(p.writeThrough, p.ids + o + b, (uint32_t)id[b]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Worlds with more than 64 triangles: one workgroup = one 64x64 tile, the
// triangles go through LDS in passes of 256.  In a pass every wave sets up and
// classifies 64 triangles (lane = triangle) and publishes planes + region
// mask; after the barrier every wave rasterises its own two 64x8 strips
// against the 256 records and shades the pass's winners before the next pass
// replaces the records.  Large meshes proper are BVH territory.
// ---------------------------------------------------------------------------
constexpr int kPass = kChunk * kWavesPerBlock;    // triangles per pass

struct PassRecs {
    float shade[kPass][4];       // rgba, texture, objectID, world-local index
    float cold[kPass][kCold];    // u/v planes, lit colour
};
struct TileLds {
    float planes[kPass][16];     // A0 A1 A2 Dx | B0 B1 B2 Dy | C0 C1 C2 Dc | mask
    PassRecs rec;                // shading records
};
__device__ __forceinline__ const float *shadeRec(const PassRecs &L, int32_t w) { return L.shade[w]; }

template <bool IDS>
__global__ __launch_bounds__(kWave *kWavesPerBlock, 3)
void rasterChunkedKernel(const RasterParams p)
{
    __shared__ TileLds lds;
    touchKernelArguments();
    const int wave = threadIdx.x / kWave;
    const int lane = threadIdx.x % kWave;
    TileCtx t;
    ViewConst vc;
    if (!tileSetup(p, blockIdx.x, lane, t, vc))
        return;                                   // whole workgroup leaves together
    const float invNear = p.invNear, invFar = p.invFar;

    // per-wave pixel state: strips 2*wave and 2*wave+1, two regions each
    float best[4][kRegionBlocks];
    int32_t bid[4][kRegionBlocks];
    uint32_t outRgba[4][kRegionBlocks];
    int32_t outId[4][kRegionBlocks];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b) {
            best[g][b] = invFar;
            bid[g][b] = -1;
            outRgba[g][b] = 0xFF000000u;
            outId[g][b] = -1;
        }

    for (uint32_t pass = 0; pass < t.numTris; pass += kPass) {
        if (pass != 0)
            __syncthreads();                      // previous pass fully consumed
        {
            const int rec = wave * kWave + lane;
            const uint32_t k = pass + (uint32_t)rec;
            TriPlanes c;
            c.A0 = c.B0 = c.C0 = c.A1 = c.B1 = c.C1 = 0.0f;
            c.A2 = c.B2 = c.C2 = c.Dx = c.Dy = c.Dc = 0.0f;
            c.bbX0 = c.bbX1 = c.bbY0 = c.bbY1 = 0.0f;
            uint32_t mask = 0;
            // whole waves past the end of the list skip the setup code
            if (pass + (uint32_t)(wave * kWave) < t.numTris) {
                bool valid = false;
                if (k < t.numTris && !(p.debugSkip & 8u)) {
                    const WorldTri wt = p.viewTris[t.triBegin + k];
                    valid = setupTriangle(p, vc, wt, (int32_t)k, c, lds.rec.shade[rec], lds.rec.cold[rec]);
                }
                if (valid && !(p.debugSkip & 4u))
                    mask = classifyRegions(c, t, invNear, invFar);
            }
            float4 *dst = reinterpret_cast<float4 *>(lds.planes[rec]);
            dst[0] = make_float4(c.A0, c.A1, c.A2, c.Dx);
            dst[1] = make_float4(c.B0, c.B1, c.B2, c.Dy);
            dst[2] = make_float4(c.C0, c.C1, c.C2, c.Dc);
            dst[3] = make_float4(__uint_as_float(mask), 0.f, 0.f, 0.f);
        }
        __syncthreads();

        const uint32_t left = t.numTris - pass;
        const int subs = left >= (uint32_t)kPass ? kWavesPerBlock : (int)((left + kWave - 1) / kWave);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int strip = 2 * wave + (g >> 1), hf = g & 1;
            const uint32_t fy = t.tileY0 + strip * 8 + t.ly;
            const float py = (float)fy;
            const uint32_t fx0 = t.tileX0 + hf * 32 + 4 * t.lx;
            float px[kRegionBlocks];
#pragma unroll
            for (int b = 0; b < kRegionBlocks; ++b)
                px[b] = (float)(fx0 + b);
            // lane k looks at the region mask of records k, 64 + k, ...; draw
            // order is record order, so the sub-chunks are visited in order
            for (int sub = 0; sub < subs; ++sub) {
                const uint32_t mask = __float_as_uint(lds.planes[sub * kWave + lane][12]);
                const uint64_t act = __ballot((mask >> (2 * strip + hf)) & 1u);
                if (!(p.debugSkip & 2u))
                    rasterRegion<true, 0>(lds.planes, act, sub * kWave, px, py, invNear, best[g], bid[g]);
            }
            // shade this pass's winners before its records are replaced
#pragma unroll
            for (int b = 0; b < kRegionBlocks; ++b) {
                if (bid[g][b] >= 0)
                    resolvePixel<IDS>(p, lds.rec, bid[g][b], best[g][b], px[b], py,
                                      outRgba[g][b], outId[g][b]);
                bid[g][b] = -1;
            }
        }
    }

#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int strip = 2 * wave + (g >> 1), hf = g & 1;
        const uint32_t fy = t.tileY0 + strip * 8 + t.ly;
        const uint32_t fx0 = t.tileX0 + hf * 32 + 4 * t.lx;
        outputRegion<IDS>(p, t, fx0, fy, invFar, best[g], outRgba[g], outId[g]);
    }
}

// ---------------------------------------------------------------------------
// Worlds of at most 256 triangles (every BASELINE scene): one workgroup renders
// a group of views (all their tiles) or a chunk of the tiles of one view.
//   S1  one lane per (view of the group, triangle slot): the setup waves
//       publish planes, bounding boxes and shading records in LDS (dense
//       lanes: four 14-triangle views fill 56 of wave 0's 64 lanes); the last
//       wave notes where the group's tiles lie and the order of the work items;
//   S2  one lane per (tile, triangle slot): wave w classifies four of the
//       tile's sixteen regions (strips 2(w&3), 2(w&3)+1), OR-ing bits into LDS;
//   R+O all waves pull (tile, strip) items off an LDS counter, rasterise the
//       strip's two regions and store them.
// ---------------------------------------------------------------------------
// A group is up to 16 tiles drawn from one or more views whose triangles fit
// the records together (setup is per view, shared by the view's tiles).  Up to
// 64 slots per view the group holds kChunk records (with the XCD-aware split
// a workgroup on a fast XCD carries one 16-slot view more); views of 128 / 256
// slots are one view per group.
constexpr int kGroupTilesMax = 16;
constexpr int kGroupSlotsMax = 256;
constexpr int groupTilesMax(int slots) { return slots <= 128 ? kGroupTilesMax : 8; }

template <int SLOTS>
struct GroupLds {
    static constexpr int kRecs = SLOTS <= kChunk ? kChunk + 16 : SLOTS;
    static constexpr int kBackground = kRecs;   // record index of "nothing hit"
    float planes[kRecs][16];            // A0 A1 A2 Dx | B0 B1 B2 Dy | C0 C1 C2 Dc | bbox
    uint32_t live[kRecs];               // record holds a triangle that can be visible
    float shade[kRecs + 1][4];          // rgba, texture, objectID, world-local index
    float cold[kRecs][kCold];           // u/v planes, lit colour
    // per (tile, slot of the tile's view): region bits 0..15, near-free bits 16..19
    uint32_t masks[groupTilesMax(SLOTS) * (SLOTS <= kChunk ? kChunk : SLOTS)];
    uint32_t tileInfo[kGroupTilesMax][4];   // view, x0, y0, flags | first record << 8
    uint32_t nextItem;                  // (tile, strip) work counter of phase R
    uint8_t itemOrder[kGroupTilesMax * 8];  // work item -> tile * 8 + strip
};
constexpr uint32_t kTileValid = 4u;


// Shade + store one region of a tile (group kernel).  `bid` is the byte offset
// of the winner's shading record (the background has its own record, so the
// lookup is unconditional); base pointers are wave-uniform.
template <bool IDS, bool FULL, bool TEX, typename LDS>
__device__ __forceinline__ void storeRegion(const RasterParams &p, const LDS &L,
                                            uint32_t *rgbTile, float *depthTile, int32_t *idsTile,
                                            uint32_t pixOff, uint32_t fx0, uint32_t fy,
                                            bool anyTex, const float (&px)[kRegionBlocks], float py,
                                            const float (&best)[kRegionBlocks],
                                            const int32_t (&bid)[kRegionBlocks])
{
    uint32_t rgba[kRegionBlocks];
    int32_t id[kRegionBlocks];
    float dep[kRegionBlocks];
    const char *shadeBase = reinterpret_cast<const char *>(&L.shade[0][0]);
#pragma unroll
    for (int b = 0; b < kRegionBlocks; ++b) {
        const float *h = reinterpret_cast<const float *>(shadeBase + bid[b]);
        rgba[b] = __float_as_uint(h[0]);
        if (IDS)
            id[b] = __float_as_int(p.idsAreSegmask ? h[2] : h[3]);
        // depth = 1/best: v_rcp_f32 (<= 1 ulp); textured colour below uses the
        // correctly rounded quotient because texel choice depends on it
        dep[b] = bid[b] != LDS::kBackground * 16 ? __builtin_amdgcn_rcpf(best[b]) : 0.0f;
    }
    // TEX is a kernel-level switch: texel loads inside the work loop make the
    // compiler drain vmcnt at every loop header, which would also wait for the
    // previous strip's stores
    if (TEX && anyTex) {
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b) {
            const int32_t rec = bid[b] >> 4;
            const int32_t tex = __float_as_int(L.shade[rec][1]);
            if (tex >= 0)
                rgba[b] = shadeTextured(p, L.cold[rec], tex, px[b], py, 1.0f / best[b]);
        }
    }
    if (p.debugSkip & 1u)
        return;
    if (FULL) {
        streamStore16(p.writeThrough, rgbTile + pixOff, rgba[0], rgba[1], rgba[2], rgba[3]);
        streamStore16(p.writeThrough, depthTile + pixOff, __float_as_uint(dep[0]), __float_as_uint(dep[1]),
                      __float_as_uint(dep[2]), __float_as_uint(dep[3]));
        if (IDS)
            streamStore16(p.writeThrough, idsTile + pixOff, (uint32_t)id[0], (uint32_t)id[1], (uint32_t)id[2], (uint32_t)id[3]);
    } else if (fy < p.nslow) {
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b) {
            if (fx0 + b < p.nfast) {
                streamStore4(p.writeThrough, rgbTile + pixOff + b, rgba[b]);
                streamStore4(p.writeThrough, depthTile + pixOff + b, __float_as_uint(dep[b]));
                if (IDS)
                    streamStore4(p.writeThrough, idsTile + pixOff + b, (uint32_t)id[b]);
            }
        }
    }
}

// A region no triangle can touch: background everywhere, no per-pixel work.
template <bool IDS, bool FULL>
__device__ __forceinline__ void storeBackground(const RasterParams &p, uint32_t *rgbTile,
                                                float *depthTile, int32_t *idsTile,
                                                uint32_t pixOff, uint32_t fx0, uint32_t fy)
{
    if (p.debugSkip & 1u)
        return;
    const uint32_t bg = 0xFF000000u;
    if (FULL) {
        streamStore16(p.writeThrough, rgbTile + pixOff, bg, bg, bg, bg);
        streamStore16(p.writeThrough, depthTile + pixOff, 0u, 0u, 0u, 0u);
        if (IDS)
            streamStore16(p.writeThrough, idsTile + pixOff, ~0u, ~0u, ~0u, ~0u);
    } else if (fy < p.nslow) {
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b) {
            if (fx0 + b < p.nfast) {
                streamStore4(p.writeThrough, rgbTile + pixOff + b, bg);
                streamStore4(p.writeThrough, depthTile + pixOff + b, 0u);
                if (IDS)
                    streamStore4(p.writeThrough, idsTile + pixOff + b, ~0u);
            }
        }
    }
}

// What the group kernel's set-up reads per triangle (instanceTransform / setupTriangleCore are templates over
// the parameter type): pointers from the preloaded header or from RasterParams, scalars from RasterParams.
struct GroupSetupArgs {
    const ObjTri *tris;
    const TriMat *triMats;
    const float *instPos, *instRot, *instScale;
    float sx, ox, sz, oz, s6bPad, ambient, diffuse;
    int32_t transposed;
};

// Waves per workgroup of the group kernel: eight for untextured scenes (more
// waves in flight absorb the stalls of a saturated store path), four for the
// textured variant (its texel loads cost registers and vmcnt drains).
constexpr int groupWaves(bool tex) { return tex ? 4 : 8; }

// XMODE (16-slot instantiations only): bit 0 = the workgroups of every pair trade places
// (XCD phase, below), bit 1 = workgroup 0 reports the XCD it runs on.  Separate
// instantiations, chosen per launch by the host, because any extra state in this kernel's
// work loop costs more than the split gains (59 of 64 VGPRs, SGPRs spilled): XMODE 0 is
// the kernel as it always was.
// FAST (16-slot instantiations; chosen by the host for uniform worlds of one-tile views without diagnostics): the
// set-up waves' path to their first pose loads reads nothing but the twelve leading header arguments (GroupHeader,
// raster.hpp), which the command processor preloads into SGPRs (-mllvm -amdgpu-kernarg-preload-count=12): the loads
// go out without a round trip to the argument block; `p` is read for what comes after.
template <bool IDS, int SLOTS, bool TEX, int XMODE, bool FAST>
__device__ __forceinline__ void groupKernelBody(const char *hPose, const char *hGeom, uint32_t hViews, uint32_t hInstances,
                                                uint32_t hPool, uint32_t hShape, uint32_t hGroups, uint32_t hPrefix,
                                                uint32_t hFirst01, uint32_t hFirst23, const RasterParams p)
{
    __shared__ GroupLds<SLOTS> lds;
    constexpr int kBackground = GroupLds<SLOTS>::kBackground;
    // readfirstlane: the compiler cannot see that threadIdx.x / 64 is
    // wave-uniform and would predicate every `wave` branch instead of jumping
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x % kWave;
    // (FAST: the set-up waves -- 0 and 1 -- have what they need in SGPRs and must not wait for the argument block;
    // the other waves touch it for the whole CU, whose scalar cache they share)
    if (!FAST || wave >= 2)
        touchKernelArguments<(FAST ? 48 : 0) + sizeof(RasterParams)>();
    // what the prologue reads of the arguments: from the preloaded header (FAST) or from `p`
    const uint32_t aNumViews = FAST ? hViews : p.numViews;
    const uint32_t aGrpViews = FAST ? (hShape & 255u) : p.grpViews;
    const uint32_t aXcdSkew = FAST ? ((hShape >> 8) & 15u) : p.xcdSkew;
    const uint32_t aXcdRotate = FAST ? ((hShape >> 12) & 1u) : p.xcdRotate;
    const uint32_t aGrpPerView = FAST ? 1u : p.grpPerView;
    const uint32_t aGrid = FAST ? hGroups : gridDim.x;
    const uint32_t aUniInstances = FAST ? ((hShape >> 13) & 7u) : p.uniInstances;
    const uint32_t aUniCams = FAST ? ((hShape >> 16) & 255u) : p.uniCamsPerWorld;
    const uint32_t dskip = FAST ? 0u : p.debugSkip;           // (the host never picks FAST with diagnostics on)
    const PoseLayout lay = poseLayout(hViews, hInstances);
    const float *aCamRot = FAST ? reinterpret_cast<const float *>(hPose + lay.camRot) : p.camRot;
    const float *aCamPos = FAST ? reinterpret_cast<const float *>(hPose + lay.camPos) : p.camPos;
    const int32_t *aInstObj = FAST ? reinterpret_cast<const int32_t *>(hPose + lay.instObj) : p.instObj;
    const uint32_t tilesPerView = FAST ? 1u : p.tilesFast * p.tilesSlow;
    // ---- which views / tiles this workgroup owns (launchRaster):
    //  A  grpPerView == 1: grpViews whole views (all their tiles);
    //  B  otherwise: grpChunkTiles tiles of one view.
    uint32_t firstView, groupViews, firstTile, groupTiles;
    // Workgroup b runs on XCD b % 8.  With xcdRotate the group it renders moves
    // two places on within its round of eight, round after round, so that an
    // XCD does not keep writing the same eighth of every 512 KiB of the output
    // (parity is kept: the XCD-aware split below pairs even with odd groups).
    // Which XCD a launch's workgroup 0 lands on depends on the hardware queue and what
    // ran on it before (measured: XCC 0, 6 or 7 for different streams of one process,
    // stable from launch to launch; profiles/r02_xcd_phase.txt).  On an odd start the
    // XCD-aware split below would move its strips the wrong way (24.9 instead of 22.5
    // us), so the two workgroups of every pair trade places then (XMODE bit 0): the host
    // picks the instantiation by the parity workgroup 0 reported in an earlier launch
    // of this renderer -- every workgroup of a launch runs the same code, so the trade
    // is consistent whatever the hardware does; a stale value costs speed, never pixels.
    const uint32_t blk = ((XMODE & 1) && SLOTS == 16 && aXcdSkew && (blockIdx.x ^ 1u) < aGrid)
                             ? blockIdx.x ^ 1u : blockIdx.x;
    // workgroup 0 reports where it runs: a host-mapped word, written by the last wave,
    // which issues no loads during set-up -- the slow write sits ahead of nothing
    if ((XMODE & 2) && blockIdx.x == 0 && threadIdx.x == (groupWaves(TEX) - 1) * kWave && p.xccReport)
        *p.xccReport = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID[3:0]
    uint32_t bid = blk;
    if (aXcdRotate && (bid | 7u) < aGrid)
        bid = (bid & ~7u) | ((bid + 2u * (bid >> 3)) & 7u);
    if (aGrpPerView == 1) {
        firstView = bid * aGrpViews;
        groupViews = aGrpViews;
        firstTile = 0;
        groupTiles = groupViews * tilesPerView;
    } else {
        firstView = bid / aGrpPerView;
        groupViews = 1;
        firstTile = (bid - firstView * aGrpPerView) * p.grpChunkTiles;
        groupTiles = min(p.grpChunkTiles, tilesPerView - firstTile);
    }
    // XCD-aware split (see launchRaster; one-tile views, four per group):
    // consecutive workgroups run on consecutive XCDs and the odd XCD of each pair
    // drains its stores more slowly.  Workgroups 2m and 2m+1 (after the trade above:
    // even = on an even XCD) share the view between their two runs of four: the odd
    // one leaves its first p.xcdSkew strips to the even one (both set the view up).
    uint32_t firstStrip = 0, numStrips = groupTiles * 8;
    if (SLOTS == 16 && aXcdSkew) {
        if (blk & 1u) {
            firstStrip = aXcdSkew;
            numStrips -= aXcdSkew;
        } else {
            groupViews += 1;
            groupTiles += 1;
            numStrips += aXcdSkew;
        }
    }
    const int groupRecs = (int)groupViews * SLOTS;
    const int numPairs = (int)groupTiles * SLOTS;
    const float invNear = p.invNear, invFar = p.invFar;
    if (dskip & 16u)
        return;                                   // timing aid: bare launch
    unsigned long long *stamps = (!FAST && p.debugStamps && wave < 4)
        ? p.debugStamps + ((size_t)blockIdx.x * 4 + wave) * 8 : nullptr;
#define MRX_STAMP(i)                                                           \
    do {                                                                       \
        if (stamps && lane == 0)                                               \
            stamps[i] = __builtin_amdgcn_s_memrealtime();                      \
    } while (0)
    MRX_STAMP(0);
    for (int i = threadIdx.x; i < numPairs; i += groupWaves(TEX) * kWave)
        lds.masks[i] = 0u;

    // ---- S1: one lane per (view vi of the group, triangle slot k): setup.
    //      Wave 0 covers the first 64 records, wave 1 a fifth 16-slot view or
    //      the next 64 slots of a large view, ...  Meanwhile the last wave
    //      notes where each tile of the group lies.
    if (wave == groupWaves(TEX) - 1 && lane < (int)groupTiles) {
        uint32_t vi = (uint32_t)lane, tile = firstTile;
        if (tilesPerView != 1) {
            vi = groupViews == 1 ? 0u : (uint32_t)lane / tilesPerView;
            tile = firstTile + (uint32_t)lane - vi * tilesPerView;
        }
        const uint32_t ty = tilesPerView != 1 ? tile / p.tilesFast : 0u;
        lds.tileInfo[lane][0] = firstView + vi;
        lds.tileInfo[lane][1] = (tile - ty * p.tilesFast) * 64u;
        lds.tileInfo[lane][2] = ty * 64u;
        lds.tileInfo[lane][3] = (firstView + vi < aNumViews ? kTileValid : 0u) | ((vi * SLOTS) << 8);
    }
    // ... and the order of the work items.  One-tile views: strip by strip
    // across the views, top strips first -- strips above the horizon cost
    // nothing, so every wave has stores in flight right after the barrier.
    // Larger views: tile by tile (switching tiles per item costs more there).
    if (wave == groupWaves(TEX) - 1) {
        uint32_t base = 0;
        for (uint32_t c0 = 0; c0 < groupTiles * 8u; c0 += kWave) {
            const uint32_t c = c0 + (uint32_t)lane;
            uint32_t strip = c / groupTiles, tile = c - strip * groupTiles;
            if (tilesPerView != 1) {
                tile = c >> 3;
                strip = c & 7u;
            }
            const uint32_t g = tile * 8u + strip;
            const bool ok = strip < 8u && g >= firstStrip && g < firstStrip + numStrips;
            const uint64_t m = __ballot(ok);
            if (ok)
                lds.itemOrder[base + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint8_t)g;
            base += (uint32_t)__builtin_popcountll(m);
        }
    }
    if (wave * kWave < groupRecs) {
        const int rec = wave * kWave + lane;
        const int vi = rec / SLOTS, k = rec % SLOTS;
        const bool recOk = rec < groupRecs;
        const bool viewOk = recOk && firstView + vi < aNumViews;
        const uint32_t view = viewOk ? firstView + vi : 0u;
        // everything addressed by the view index is requested up front; the
        // pose / geometry rows one level down follow as soon as wt arrives --
        // or at once, when the draw list is arithmetic (uniform worlds)
        WorldTri wt;
        uint32_t numTris;
        if (aUniInstances) {
            const uint32_t kk = (uint32_t)k;
            // (FAST: the table rides in the header, eight / sixteen bits per entry)
            const uint32_t pre1 = FAST ? (hPrefix & 255u) : p.uniPrefix[1], pre2 = FAST ? ((hPrefix >> 8) & 255u) : p.uniPrefix[2];
            const uint32_t pre3 = FAST ? ((hPrefix >> 16) & 255u) : p.uniPrefix[3], pre4 = FAST ? (hPrefix >> 24) : p.uniPrefix[4];
            const uint32_t ft0 = FAST ? (hFirst01 & 0xFFFFu) : p.uniFirstTri[0], ft1 = FAST ? (hFirst01 >> 16) : p.uniFirstTri[1];
            const uint32_t ft2 = FAST ? (hFirst23 & 0xFFFFu) : p.uniFirstTri[2], ft3 = FAST ? (hFirst23 >> 16) : p.uniFirstTri[3];
            const uint32_t li = (kk >= pre1 ? 1u : 0u) + (kk >= pre2 ? 1u : 0u) + (kk >= pre3 ? 1u : 0u);
            const uint32_t pre = li == 0 ? 0u : li == 1 ? pre1 : li == 2 ? pre2 : pre3;
            const uint32_t first = li == 0 ? ft0 : li == 1 ? ft1 : li == 2 ? ft2 : ft3;
            // integer divisions cost ~25 VALU each: a real (scalar) branch
            // around this one for the common one-camera-per-world case
            uint32_t world = view;
            if (aUniCams != 1)
                world = view / aUniCams;
            wt.inst = world * aUniInstances + li;
            wt.tri = first + (kk - pre);
            numTris = viewOk ? pre4 : 0u;
            if (kk >= pre4) {                     // idle slot: keep the loads in range
                wt.inst = 0;
                wt.tri = 0;
            }
        } else {
            // slots past the view's row are idle: keep their load inside the row
            wt = p.viewTris[view * p.viewTriStride + ((uint32_t)k < p.viewTriStride ? (uint32_t)k : 0u)];
            numTris = viewOk ? p.viewTriCount[view] : 0u;
        }
        ViewConst vc;
        {
            const float4 q = *reinterpret_cast<const float4 *>(aCamRot + 4 * view);
            quatToMat(q.x, q.y, q.z, q.w, vc.Rc);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                vc.c[r] = aCamPos[3 * view + r];
                vc.lv[r] = dot3(vc.Rc[0][r], vc.Rc[1][r], vc.Rc[2][r],
                                p.toLight[0], p.toLight[1], p.toLight[2]);
            }
        }
        TriPlanes c;
        c.A0 = c.B0 = c.C0 = c.A1 = c.B1 = c.C1 = 0.0f;
        c.A2 = c.B2 = c.C2 = c.Dx = c.Dy = c.Dc = 0.0f;
        c.bbX0 = c.bbX1 = c.bbY0 = c.bbY1 = 0.0f;
        bool valid = false;
        MRX_STAMP(1);
        if (recOk) {
            lds.shade[rec][1] = __int_as_float(-1);
            if ((uint32_t)k < numTris && !(dskip & 8u)) {
                // the pose / geometry rows through pointers that come from the header (FAST) or from `p`
                const GroupSetupArgs sa = {
                    FAST ? reinterpret_cast<const ObjTri *>(hGeom) : p.tris,
                    FAST ? reinterpret_cast<const TriMat *>(hGeom + geomMatsOffset(hPool)) : p.triMats,
                    FAST ? reinterpret_cast<const float *>(hPose + lay.instPos) : p.instPos,
                    FAST ? reinterpret_cast<const float *>(hPose + lay.instRot) : p.instRot,
                    FAST ? reinterpret_cast<const float *>(hPose + lay.instScale) : p.instScale,
                    p.sx, p.ox, p.sz, p.oz, p.s6bPad, p.ambient, p.diffuse, p.transposed };
                InstXform x;
                instanceTransform(sa, vc, wt.inst, x);
                valid = setupTriangleCore<true>(sa, vc.lv, x, wt.tri, aInstObj[wt.inst], k, c, lds.shade[rec], lds.cold[rec]);
            }
            MRX_STAMP(2);
            float4 *dst = reinterpret_cast<float4 *>(lds.planes[rec]);
            dst[0] = make_float4(c.A0, c.A1, c.A2, c.Dx);
            dst[1] = make_float4(c.B0, c.B1, c.B2, c.Dy);
            dst[2] = make_float4(c.C0, c.C1, c.C2, c.Dc);
            dst[3] = make_float4(c.bbX0, c.bbX1, c.bbY0, c.bbY1);
            lds.live[rec] = (valid && !(dskip & 4u)) ? 1u : 0u;
        }
        if (rec == 0) {
            lds.nextItem = 0;
            lds.shade[kBackground][0] = __uint_as_float(0xFF000000u);
            lds.shade[kBackground][1] = __int_as_float(-1);
            lds.shade[kBackground][2] = __int_as_float(-1);
            lds.shade[kBackground][3] = __int_as_float(-1);
        }
    }
    __syncthreads();
    MRX_STAMP(3);

    // ---- S2: classification of every (tile, triangle slot) pair against the
    //      tile's sixteen 32x8 regions.  Wave w takes strips 2(w&3), 2(w&3)+1
    //      (four regions) of the pairs (w>>2)*64 + lane, + 64 * (waves / 4), ...;
    //      results are OR-ed in LDS: bits 0..15 regions, bits 16..19
    //      "near-free over this wave's strips".
    constexpr int kPairStride = (groupWaves(TEX) / 4) * kWave;
    for (int pair = (wave >> 2) * kWave + lane; pair - lane < numPairs; pair += kPairStride) {
        if (pair >= numPairs)
            continue;
        const int j = pair / SLOTS, k = pair % SLOTS;
        const uint32_t info = lds.tileInfo[j][3];
        const int rec = (int)(info >> 8) + k;
        if (!(info & kTileValid) || !lds.live[rec])
            continue;
        const float4 *src = reinterpret_cast<const float4 *>(lds.planes[rec]);
        const float4 a = src[0], b = src[1], cc = src[2], bb = src[3];
        TriPlanes c;
        c.A0 = a.x; c.A1 = a.y; c.A2 = a.z; c.Dx = a.w;
        c.B0 = b.x; c.B1 = b.y; c.B2 = b.z; c.Dy = b.w;
        c.C0 = cc.x; c.C1 = cc.y; c.C2 = cc.z; c.Dc = cc.w;
        c.bbX0 = bb.x; c.bbX1 = bb.y; c.bbY0 = bb.z; c.bbY1 = bb.w;
        const uint32_t tx0 = lds.tileInfo[j][1], ty0 = lds.tileInfo[j][2];
        bool nearOk = false;
        uint32_t m;
        switch (wave & 3) {
        case 0: m = classifyStrips<0, 2>(c, tx0, ty0, invNear, invFar, nearOk); break;
        case 1: m = classifyStrips<2, 4>(c, tx0, ty0, invNear, invFar, nearOk); break;
        case 2: m = classifyStrips<4, 6>(c, tx0, ty0, invNear, invFar, nearOk); break;
        default: m = classifyStrips<6, 8>(c, tx0, ty0, invNear, invFar, nearOk); break;
        }
        if (nearOk)
            m |= 1u << (16 + (wave & 3));
        if (m)
            atomicOr(&lds.masks[pair], m);
    }
    __syncthreads();
    MRX_STAMP(4);

    // ---- R + O: the four waves pull (tile, strip) work items off an LDS
    //      counter -- strips differ a lot in cost (sky vs. ground vs. objects),
    //      a static split leaves waves idle
    // lane -> four consecutive pixels of one row of a 32x8 region (a 64x4
    // region with fully linear 1 KiB stores was measured slower: no x culling)
    const int lx = lane & 7, ly = lane >> 3;
    const uint32_t laneOff = (uint32_t)ly * p.nfast + 4u * lx;
    int cachedTile = -1, recBase = 0;
    constexpr int SUBS = (SLOTS + kWave - 1) / kWave;     // 64-slot sub-chunks of a view
    constexpr int LANES = SLOTS < kWave ? SLOTS : kWave;
    uint32_t view = 0, tileX0 = 0, tileY0 = 0, mask[SUBS] = {};
    float pxTile[2][kRegionBlocks] = {};
    bool anyTex = false, nearFree = false, full = false;
    uint32_t *rgbTile = nullptr;
    float *depthTile = nullptr;
    int32_t *idsTile = nullptr;
    for (;;) {
        uint32_t item = 0;
        if (lane == 0)
            item = atomicAdd(&lds.nextItem, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= numStrips)
            break;
        item = __builtin_amdgcn_readfirstlane((uint32_t)lds.itemOrder[item]);
        const int j = (int)(item >> 3), strip = (int)(item & 7u);
        if (j != cachedTile) {
            const uint32_t info = __builtin_amdgcn_readfirstlane(lds.tileInfo[j][3]);
            if (!(info & kTileValid))
                continue;                           // a view past the end of the batch
            cachedTile = j;
            recBase = (int)(info >> 8);
            view = __builtin_amdgcn_readfirstlane(lds.tileInfo[j][0]);
            tileX0 = __builtin_amdgcn_readfirstlane(lds.tileInfo[j][1]);
            tileY0 = __builtin_amdgcn_readfirstlane(lds.tileInfo[j][2]);
            const size_t tileBase = ((size_t)view * p.nslow + tileY0) * p.nfast + tileX0;
            rgbTile = p.rgb + tileBase;
            depthTile = p.depth + tileBase;
            idsTile = IDS ? p.ids + tileBase : nullptr;
            full = (p.nfast & 3u) == 0 && tileX0 + 64u <= p.nfast && tileY0 + 64u <= p.nslow;
            // lane k looks at the region masks of triangle slots k, 64 + k, ... of tile j
            anyTex = false;
            nearFree = true;
#pragma unroll
            for (int sub = 0; sub < SUBS; ++sub) {
                const int slot = sub * kWave + lane;
                mask[sub] = lane < LANES ? lds.masks[j * SLOTS + slot] : 0u;
                const int32_t tex = lane < LANES ? __float_as_int(lds.shade[recBase + slot][1]) : -1;
                anyTex = anyTex || __ballot((mask[sub] & 0xFFFFu) != 0 && tex >= 0) != 0;
                // every surviving triangle stays behind the near plane over the
                // whole tile: the per-pixel near test is dropped for the tile
                nearFree = nearFree &&
                           __ballot((mask[sub] & 0xFFFFu) != 0 && ((mask[sub] >> 16) & 0xFu) != 0xFu) == 0;
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int b = 0; b < kRegionBlocks; ++b)
                    pxTile[hf][b] = (float)(tileX0 + hf * 32 + 4 * lx + b);
        }
        const uint32_t fy = tileY0 + strip * 8 + ly;
        const float py = (float)fy;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const uint32_t fx0 = tileX0 + hf * 32 + 4 * lx;
            const uint32_t pixOff = (uint32_t)(strip * 8) * p.nfast + hf * 32 + laneOff;
            // bit k of act[sub]: triangle slot 64 sub + k of the tile's view survives here
            uint64_t act[SUBS];
            bool any = false;
#pragma unroll
            for (int sub = 0; sub < SUBS; ++sub) {
                act[sub] = __ballot((mask[sub] >> (2 * strip + hf)) & 1u);
                any = any || act[sub] != 0;
            }
            if (!any) {
                if (full)
                    storeBackground<IDS, true>(p, rgbTile, depthTile, idsTile, pixOff, fx0, fy);
                else
                    storeBackground<IDS, false>(p, rgbTile, depthTile, idsTile, pixOff, fx0, fy);
                continue;
            }
            const float (&px)[kRegionBlocks] = pxTile[hf];
            float best[kRegionBlocks];
            int32_t bid[kRegionBlocks];
#pragma unroll
            for (int b = 0; b < kRegionBlocks; ++b) {
                best[b] = invFar;
                bid[b] = kBackground * 16;
            }
            if (!(dskip & 2u)) {
#pragma unroll
                for (int sub = 0; sub < SUBS; ++sub) {
                    if (SUBS > 1 && act[sub] == 0)
                        continue;
                    if (nearFree)
                        rasterRegion<false, 4>(lds.planes, act[sub], recBase + sub * kWave, px, py, invNear, best, bid);
                    else
                        rasterRegion<true, 4>(lds.planes, act[sub], recBase + sub * kWave, px, py, invNear, best, bid);
                }
            }
            if (full)
                storeRegion<IDS, true, TEX>(p, lds, rgbTile, depthTile, idsTile, pixOff, fx0, fy,
                                            anyTex, px, py, best, bid);
            else
                storeRegion<IDS, false, TEX>(p, lds, rgbTile, depthTile, idsTile, pixOff, fx0, fy,
                                             anyTex, px, py, best, bid);
        }
    }
    MRX_STAMP(5);
    MRX_STAMP(6);
    if (stamps && lane == 0)       // where this wave ran: HW_ID (reg 4) and XCC_ID (reg 20)
        stamps[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                    (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#undef MRX_STAMP
}

// The two entry points of the body: the plain one (arguments = RasterParams, as every other kernel here) and the
// FAST one with the preloaded header in front.
template <bool IDS, int SLOTS, bool TEX, int XMODE = 0>
// (the second bound is waves per SIMD; 256-slot groups are LDS-limited to 3 per CU)
__global__ __launch_bounds__(kWave *groupWaves(TEX), TEX ? (SLOTS > 128 ? 3 : 4) : (SLOTS > 128 ? 6 : 8))
void rasterGroupKernel(const RasterParams p)
{
    groupKernelBody<IDS, SLOTS, TEX, XMODE, false>(nullptr, nullptr, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, p);
}

template <bool IDS, bool TEX, int XMODE>
__global__ __launch_bounds__(kWave *groupWaves(TEX), TEX ? 4 : 8)
void rasterGroupKernelFast(const char *hPose, const char *hGeom, uint32_t hViews, uint32_t hInstances, uint32_t hPool,
                           uint32_t hShape, uint32_t hGroups, uint32_t hPrefix, uint32_t hFirst01, uint32_t hFirst23,
                           const RasterParams p)
{
    groupKernelBody<IDS, 16, TEX, XMODE, true>(hPose, hGeom, hViews, hInstances, hPool, hShape, hGroups, hPrefix, hFirst01,
                                               hFirst23, p);
}

}  // namespace

hipError_t launchRaster(const RasterParams &p, uint32_t maxWorldTris,
                        int32_t variant, hipStream_t stream)
{
    const uint32_t items = p.numViews * p.tilesFast * p.tilesSlow;
    if (items == 0)
        return hipSuccess;
    const dim3 block(kWave * kWavesPerBlock);
    const bool ids = p.ids != nullptr;
    const bool multi = maxWorldTris > (uint32_t)kChunk;       // brute variant: chunk loop
    const bool large = maxWorldTris > (uint32_t)kGroupSlotsMax;
    if (variant == kVariantBrute) {
        // v1 reference: one wave per tile, every triangle at every pixel
        const dim3 grid((items + kWavesPerBlock - 1) / kWavesPerBlock);
        if (ids) {
            if (multi) rasterBruteKernel<true, true><<<grid, block, 0, stream>>>(p);
            else       rasterBruteKernel<true, false><<<grid, block, 0, stream>>>(p);
        } else {
            if (multi) rasterBruteKernel<false, true><<<grid, block, 0, stream>>>(p);
            else       rasterBruteKernel<false, false><<<grid, block, 0, stream>>>(p);
        }
    } else if (large) {
        // more triangles per world than the group kernel holds: one workgroup per tile
        if (ids) rasterChunkedKernel<true><<<dim3(items), block, 0, stream>>>(p);
        else     rasterChunkedKernel<false><<<dim3(items), block, 0, stream>>>(p);
    } else {
        // triangle slots per view: the smallest of 16 ... 256 that holds a world
        int slots = 16;
        while ((uint32_t)slots < maxWorldTris)
            slots *= 2;
        if (p.debugSlots >= slots && p.debugSlots <= kGroupSlotsMax && (p.debugSlots & (p.debugSlots - 1)) == 0)
            slots = p.debugSlots;                 // tuning / test aid (MRX_DEBUG_SLOTS)
        // Workgroup shape.  Setup is per view and the view's tiles share it, so
        // a workgroup takes as many whole views as fit 64 records and 16 tiles
        // (8 for views of 256 slots); a view of more tiles is cut into chunks.
        // Small batches shrink the workgroups again until there are ~4 per CU.
        const uint32_t tpv = p.tilesFast * p.tilesSlow;
        const uint32_t maxTiles = (uint32_t)groupTilesMax(slots);
        const uint32_t kFill = groupFill(p.numCUs);    // 4 resident workgroups per CU: 1024 on 256 CUs
        RasterParams q = p;
        uint32_t vg = 1, ct = tpv, gpv = 1;
        if (tpv == 1) {
            // one-tile views: as many as fit 64 records, all workgroups resident
            // (textured: two -- its workgroups have four waves and longer tiles)
            vg = std::max<uint32_t>(1u, (uint32_t)(kChunk / slots));
            if (p.anyTextured)
                vg = std::min<uint32_t>(vg, 2u);
            // ... until ~4 workgroups per CU remain; two views per workgroup are kept
            // down to 2 per CU (1024 worlds: 9.9 -> 9.3 us)
            while (vg > 1 && (p.numViews + vg - 1) / vg < (vg == 2 ? kFill / 2 : kFill))
                vg /= 2;
        } else if (p.anyTextured) {
            // textured, several tiles per view: a quarter of a view per workgroup, at most four
            // tiles (4 waves per workgroup and per-pixel texturing make a tile long enough for
            // its own setup; measured on 128^2 and 256^2: profiles/r01_group_shapes.txt)
            ct = std::max<uint32_t>(1u, std::min<uint32_t>(4u, tpv / 4u));
        } else {
            // untextured, several tiles per view: one view per workgroup, at most eight tiles
            ct = std::min<uint32_t>(tpv, std::min<uint32_t>(maxTiles, 8u));
        }
        if (tpv != 1)
            while (ct > 1 && (uint64_t)p.numViews * ((tpv + ct - 1) / ct) < kFill)
                ct = (ct + 1) / 2;
        if (p.grpViewsWanted > 0 && tpv * (uint32_t)p.grpViewsWanted <= maxTiles &&
            p.grpViewsWanted * slots <= kChunk) {
            vg = (uint32_t)p.grpViewsWanted;
            ct = tpv;
        }
        if (p.grpTilesWanted > 0 && (uint32_t)p.grpTilesWanted <= maxTiles) {
            vg = 1;
            ct = std::min<uint32_t>((uint32_t)p.grpTilesWanted, tpv);
        }
        gpv = (tpv + ct - 1) / ct;
        q.grpViews = vg;
        q.grpChunkTiles = ct;
        q.grpPerView = gpv;
        const uint32_t numGroups = gpv == 1 ? (p.numViews + vg - 1) / vg : p.numViews * gpv;
        // XCD-aware split.  Workgroups go round-robin to the eight XCDs and the
        // odd XCD of every pair drains its stores ~15 % more slowly (measured,
        // profiles/r01_xcd.txt): once the batch fills the chip, xcdSkew strips
        // (eighths of a tile) per workgroup pair move from the odd to the even
        // XCD (see the kernel prologue).
        const bool skewable = slots == 16 && tpv == 1 && (vg == 4 || vg == 2);
        q.xcdSkew = (skewable && numGroups >= kFill) ? (vg == 4 ? 3u : 1u) : 0u;
        if (p.xcdSkewWanted >= 0)
            q.xcdSkew = skewable ? (uint32_t)(p.xcdSkewWanted < 8 ? p.xcdSkewWanted : 7) : 0u;
        q.xcdRotate = (tpv == 1 && numGroups >= kFill) ? 1u : 0u;
        if (p.xcdRotateWanted >= 0)
            q.xcdRotate = p.xcdRotateWanted ? 1u : 0u;
        const dim3 grid(numGroups);
        const dim3 gblock(kWave * groupWaves(p.anyTextured != 0));
        // The argument header (GroupHeader, raster.hpp) goes along with every launch; the FAST instantiations --
        // 16 slots, one-tile views, uniform worlds whose table fits the header, no diagnostics -- read nothing else
        // ahead of their pose loads.
        GroupHeader h {};
        h.pose = p.poseBlock;
        h.geom = p.geomBlock;
        h.views = p.numViews;
        h.instances = p.numInstances;
        h.poolTris = p.poolTris;
        h.groups = numGroups;
        bool fast = slots == 16 && tpv == 1 && gpv == 1 && p.uniInstances != 0 && p.uniCamsPerWorld < 256 && vg < 256 &&
                    p.poseBlock && p.geomBlock && p.debugSkip == 0 && p.debugStamps == nullptr && p.uniPrefix[4] < 256;
        for (int i = 0; i < 4 && fast; ++i)
            fast = p.uniFirstTri[i] < 65536u;
        if (const char *dbg = std::getenv("MRX_GROUP_FAST"))      // 0: never (A/B and tests)
            fast = fast && std::atoi(dbg) != 0;
        if (fast) {
            h.shape = vg | (q.xcdSkew << 8) | (q.xcdRotate << 12) | (p.uniInstances << 13) | (p.uniCamsPerWorld << 16) | (1u << 31);
            h.prefix = p.uniPrefix[1] | (p.uniPrefix[2] << 8) | (p.uniPrefix[3] << 16) | (p.uniPrefix[4] << 24);
            h.first01 = p.uniFirstTri[0] | (p.uniFirstTri[1] << 16);
            h.first23 = p.uniFirstTri[2] | (p.uniFirstTri[3] << 16);
        }
#define MRX_GROUP_ARGS h.pose, h.geom, h.views, h.instances, h.poolTris, h.shape, h.groups, h.prefix, h.first01, h.first23, q
#define MRX_GROUP_X(S, X)                                                      \
    do {                                                                       \
        if (fast && S == 16) {                                                 \
            if (p.anyTextured) {                                               \
                if (ids) rasterGroupKernelFast<true, true, X><<<grid, gblock, 0, stream>>>(MRX_GROUP_ARGS);   \
                else     rasterGroupKernelFast<false, true, X><<<grid, gblock, 0, stream>>>(MRX_GROUP_ARGS);  \
            } else {                                                           \
                if (ids) rasterGroupKernelFast<true, false, X><<<grid, gblock, 0, stream>>>(MRX_GROUP_ARGS);  \
                else     rasterGroupKernelFast<false, false, X><<<grid, gblock, 0, stream>>>(MRX_GROUP_ARGS); \
            }                                                                  \
        } else if (p.anyTextured) {                                            \
            if (ids) rasterGroupKernel<true, S, true, X><<<grid, gblock, 0, stream>>>(q);   \
            else     rasterGroupKernel<false, S, true, X><<<grid, gblock, 0, stream>>>(q);  \
        } else {                                                               \
            if (ids) rasterGroupKernel<true, S, false, X><<<grid, gblock, 0, stream>>>(q);  \
            else     rasterGroupKernel<false, S, false, X><<<grid, gblock, 0, stream>>>(q); \
        }                                                                      \
    } while (0)
#define MRX_GROUP(S) MRX_GROUP_X(S, 0)
        if (slots == 16) {
            // XCD phase: trade places within the pairs on an odd start, ask for a report now and then
            const int xmode = ((q.xcdSkew && (p.xcdPhase & 1u)) ? 1 : 0) | (p.xccReport ? 2 : 0);
            if (xmode == 0) MRX_GROUP_X(16, 0);
            else if (xmode == 1) MRX_GROUP_X(16, 1);
            else if (xmode == 2) MRX_GROUP_X(16, 2);
            else MRX_GROUP_X(16, 3);
        }
        else if (slots == 32) MRX_GROUP(32);
        else if (slots == 64) MRX_GROUP(64);
        else if (slots == 128) MRX_GROUP(128);
        else MRX_GROUP(256);
#undef MRX_GROUP_ARGS
#undef MRX_GROUP_X
#undef MRX_GROUP
    }
    return hipGetLastError();
}

}  // namespace mrx
