// BVH ray-trace path for gfx950 (MI355X): primary rays through a two-level
// bounding volume hierarchy -- the counterpart of the reference's Raytracer
// render graph (/root/reference/src/mgr.cpp:443-492: per-world TLAS over the
// instances, per-object BLAS from AssetProcessor::makeBVHData, closest hit per
// pixel).  Used for worlds too large for the group kernel's triangle slots
// (raster.hip), in both render modes: visibility is defined once (DESIGN.md
// section 3) and this kernel computes the same function of the scene.
//
// Wave64 design: the 64 primary rays... are not traced one per lane.  A wave
// owns a 64x8-pixel strip of a tile (8 pixels per lane, as in the raster
// kernels) and walks the hierarchy ONCE for the whole strip -- a packet
// traversal whose control flow is wave-uniform:
//   TLAS  one lane per instance (64-wide nodes): the instance's object box is
//         carried to view space and projected; the lanes whose screen
//         rectangle meets the strip form a ballot mask.  Built per step in LDS
//         from the live pose tensors (phase I).
//   BLAS  8-wide nodes x 8 box corners = 64 lanes: each lane projects one
//         corner of one child box, an 8-lane reduction gives the child's
//         rectangle; hit children go on a per-wave stack in LDS.
//   leaf  candidate triangles are queued and set up 64 at a time (lane =
//         triangle): the S6 edge / 1-over-depth planes of the spec, exactly as
//         the raster kernels and the oracle compute them.  The strip's pixels
//         are then tested against the surviving triangles with the planes
//         broadcast from LDS (packed FMAs).
// A box test only ever skips work: rectangles are padded so that no triangle
// that could own a pixel of the strip is dropped, and the pixel test itself is
// the spec's.  Traversal order is not draw order, so the winner is chosen by
// (1/depth, then lower world-local triangle index) -- the total order the
// oracle's in-order strict '>' scan induces.
//
// Compiled with -ffp-contract=off like raster.hip.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "bvh.hpp"
#include "raster_dev.hpp"

namespace mrx {
namespace {

constexpr int kBvhWaves = 8;            // one wave per 64x8 strip of the tile
constexpr int kQueueCap = 64 + 32;      // a flush takes 64; one append adds <= 32
constexpr int kInstRecDw = 24;          // MV[9] tv[3] qo[3] det sc[3] obj kBase firstTri numTris root

struct WaveScratch {
    float planes[kWave][16];            // A0 A1 A2 Dx | B0 B1 B2 Dy | C0 C1 C2 Dc | k - - -
    uint2 queue[kQueueCap];             // (instance of the pass, object triangle)
    uint32_t stack[kBvhStackCap];
};
static_assert(sizeof(WaveScratch) % 16 == 0, "WaveScratch alignment");

struct Rect { float x0, x1, y0, y1; };  // storage pixels (fast, slow), inclusive

__device__ __forceinline__ bool overlaps(const Rect &r, float X0, float X1, float Y0, float Y1)
{
    return r.x1 >= X0 && r.x0 <= X1 && r.y1 >= Y0 && r.y0 <= Y1;
}

__device__ __forceinline__ float rfl(float v)
{
    return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}
__device__ __forceinline__ uint32_t rflu(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// One corner of a box under MV / tv: its image position in pixels, and whether it
// lies safely in front of the eye plane (far enough, relative to the size of
// the terms it is summed from, for the quotient to be meaningful).
__device__ __forceinline__ void projectCorner(const RasterParams &p, const float (&MV)[3][3],
                                              const float (&tv)[3], float cx, float cy, float cz,
                                              float isx, float isz, float &fx, float &fz, bool &front)
{
    float P[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        P[r] = __builtin_fmaf(MV[r][2], cz, __builtin_fmaf(MV[r][1], cy, __builtin_fmaf(MV[r][0], cx, tv[r])));
    const float scale = fabsf(MV[1][0] * cx) + fabsf(MV[1][1] * cy) + fabsf(MV[1][2] * cz) + fabsf(tv[1]);
    front = P[1] > 1e-3f * scale && P[1] > 1e-6f;
    const float iw = __builtin_amdgcn_rcpf(P[1]);
    fx = (P[0] * iw - p.ox) * isx;
    fz = (P[2] * iw - p.oz) * isz;
}

// Image bounds -> padded storage rectangle.  Not `front`: the box reaches the
// eye plane, its image is unbounded -- always visit.
__device__ __forceinline__ Rect finishRect(const RasterParams &p, float x0, float x1, float z0,
                                           float z1, bool front)
{
    const float mx = 1.0f + 4e-3f * fmaxf(fabsf(x0), fabsf(x1));
    const float mz = 1.0f + 4e-3f * fmaxf(fabsf(z0), fabsf(z1));
    x0 -= mx; x1 += mx; z0 -= mz; z1 += mz;
    const bool ok = front && (x1 - x0) < 3.0e38f && (z1 - z0) < 3.0e38f;
    const float inf = __builtin_inff();
    const bool trs = p.transposed != 0;
    Rect r;
    r.x0 = ok ? (trs ? z0 : x0) : -inf;
    r.x1 = ok ? (trs ? z1 : x1) : inf;
    r.y0 = ok ? (trs ? x0 : z0) : -inf;
    r.y1 = ok ? (trs ? x1 : z1) : inf;
    return r;
}

// One pixel of the lane against one triangle; ties in 1/depth go to the lower
// world-local triangle index (what the oracle's in-order scan with a strict
// '>' does), so the result does not depend on the traversal order.  `key` is
// (triangle index << 6 | slot of the batch): one register carries both the
// tie-break and where the winner's shading data sits; `changed` collects one
// bit per pixel of the lane whose winner is of the current batch.
constexpr int kSlotBits = 6;
__device__ __forceinline__ void pixelTestTie(const PlanePairs &q, f32x2 r01, f32x2 r2d, float px,
                                             float invNear, int32_t keyv, uint32_t bit, float &best,
                                             int32_t &key, uint32_t &changed)
{
    const f32x2 pp = { px, px };
    const f32x2 e01 = fma2(q.A01, pp, r01);       // e0, e1
    const f32x2 e2d = fma2(q.A2D, pp, r2d);       // e2, 1/depth
    const float it = e2d.y;
    const bool closer = (it > best) | ((it == best) & (keyv < key));
    const bool in = (fminf(fminf(e01.x, e01.y), e2d.x) >= 0.0f) & closer & (it <= invNear);
    best = in ? it : best;
    key = in ? keyv : key;
    changed |= in ? bit : 0u;
}

// IDS: 0 = no id tensor, 1 = visibility ids (world-local triangle index),
// 2 = segmask (objectID of the winner's instance)
template <int IDS, bool TEX>
__global__ __launch_bounds__(kWave *kBvhWaves, 4)
void bvhTraceKernel(const RasterParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x % kWave;
    const uint32_t tilesPerView = p.tilesFast * p.tilesSlow;
    const uint32_t view = blockIdx.x / tilesPerView;
    const uint32_t tile = blockIdx.x - view * tilesPerView;
    const uint32_t tileX0 = (tile % p.tilesFast) * 64u, tileY0 = (tile / p.tilesFast) * 64u;
    const uint32_t passInst = p.bvhPassInst;

    // ---- LDS: TLAS of the pass (instance records, rectangles, 64-instance node
    //      rectangles), then the per-wave scratch
    float *instRec = reinterpret_cast<float *>(smem);
    float4 *instRect = reinterpret_cast<float4 *>(instRec + (size_t)passInst * kInstRecDw);
    float4 *chunkRect = instRect + passInst;
    WaveScratch *ws = reinterpret_cast<WaveScratch *>(chunkRect + passInst / kWave) + wave;
    float (*coldLds)[kCold] = nullptr;
    if (TEX)
        coldLds = reinterpret_cast<float (*)[kCold]>(
                      reinterpret_cast<WaveScratch *>(chunkRect + passInst / kWave) + kBvhWaves) +
                  wave * kWave;

    // ---- view constants (wave-uniform)
    ViewConst vc;
    {
        const float4 q = *reinterpret_cast<const float4 *>(p.camRot + 4 * view);
        quatToMat(q.x, q.y, q.z, q.w, vc.Rc);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            vc.c[r] = p.camPos[3 * view + r];
            vc.lv[r] = dot3(vc.Rc[0][r], vc.Rc[1][r], vc.Rc[2][r],
                            p.toLight[0], p.toLight[1], p.toLight[2]);
        }
    }
    const uint32_t world = p.viewWorld[view];
    const uint32_t i0 = p.worldInstStart[world], i1 = p.worldInstStart[world + 1];
    const float isx = __builtin_amdgcn_rcpf(p.sx), isz = __builtin_amdgcn_rcpf(p.sz);
    const float invNear = p.invNear, invFar = p.invFar;

    // ---- the wave's strip and the lane's pixels: four consecutive pixels of one
    //      row in each 32-pixel half (one 16-byte store per tensor and half)
    const int lx = lane & 7, ly = lane >> 3;
    const float SX0 = (float)tileX0, SX1 = (float)(tileX0 + 63u);
    const float SY0 = (float)(tileY0 + 8u * wave), SY1 = (float)(tileY0 + 8u * wave + 7u);
    const float py = (float)(tileY0 + 8u * wave + ly);
    float best[2][kRegionBlocks];
    int32_t key[2][kRegionBlocks], seg[2][kRegionBlocks];
    uint32_t rgba[2][kRegionBlocks];
    uint32_t changed = 0;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b) {
            best[hf][b] = invFar;
            key[hf][b] = -1;
            seg[hf][b] = -1;
            rgba[hf][b] = 0xFF000000u;
        }

    for (uint32_t passBase = i0; passBase < i1; passBase += passInst) {
        const uint32_t n = min(passInst, i1 - passBase);
        if (passBase != i0)
            __syncthreads();                          // previous pass's TLAS fully consumed
        // ---- phase I: the TLAS of this pass.  Lane = instance: transform (S2/S3),
        //      S6b quantities, projected object box; 64 instances form one node.
        for (uint32_t ch = (uint32_t)wave; ch * kWave < n; ch += kBvhWaves) {
            const uint32_t li = ch * kWave + lane;
            const bool has = li < n;
            const uint32_t row = passBase + (has ? li : 0u);
            const int32_t obj = p.instObj[row];
            const bool okObj = has && obj >= 0 && (uint32_t)obj < p.numObjects;
            const float4 *oi = reinterpret_cast<const float4 *>(p.objInfo + (okObj ? obj : 0));
            const float4 o0 = oi[0], omin = oi[1], omax = oi[2];
            InstXform x;
            instanceTransform(p, vc, row, x);
            float x0 = __builtin_inff(), x1 = -__builtin_inff(), z0 = x0, z1 = x1;
            bool front = true;
#pragma nounroll
            for (int corner = 0; corner < 8; ++corner) {
                float fx, fz;
                bool f;
                projectCorner(p, x.MV, x.tv, (corner & 1) ? omax.x : omin.x, (corner & 2) ? omax.y : omin.y,
                              (corner & 4) ? omax.z : omin.z, isx, isz, fx, fz, f);
                front = front && f;
                x0 = fminf(x0, fx); x1 = fmaxf(x1, fx);
                z0 = fminf(z0, fz); z1 = fmaxf(z1, fz);
            }
            Rect r = finishRect(p, x0, x1, z0, z1, front);
            const uint32_t numTris = __float_as_uint(o0.y);
            if (!okObj || numTris == 0u) {            // nothing to draw: a rectangle nothing meets
                r.x0 = r.y0 = __builtin_inff();
                r.x1 = r.y1 = -__builtin_inff();
            }
            if (has) {
                float4 *dst = reinterpret_cast<float4 *>(instRec + (size_t)li * kInstRecDw);
                dst[0] = make_float4(x.MV[0][0], x.MV[0][1], x.MV[0][2], x.MV[1][0]);
                dst[1] = make_float4(x.MV[1][1], x.MV[1][2], x.MV[2][0], x.MV[2][1]);
                dst[2] = make_float4(x.MV[2][2], x.tv[0], x.tv[1], x.tv[2]);
                dst[3] = make_float4(x.qo[0], x.qo[1], x.qo[2], x.det);
                dst[4] = make_float4(x.sc[0], x.sc[1], x.sc[2], __int_as_float(obj));
                dst[5] = make_float4(__uint_as_float(p.instKBase[row]), o0.x, o0.y, o0.z);
                instRect[li] = make_float4(r.x0, r.x1, r.y0, r.y1);
            }
            // the node over these 64 instances: union of their rectangles
            float nx0 = r.x0, nx1 = r.x1, ny0 = r.y0, ny1 = r.y1;
            if (!has) {
                nx0 = ny0 = __builtin_inff();
                nx1 = ny1 = -__builtin_inff();
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                nx0 = fminf(nx0, __shfl_xor(nx0, m));
                nx1 = fmaxf(nx1, __shfl_xor(nx1, m));
                ny0 = fminf(ny0, __shfl_xor(ny0, m));
                ny1 = fmaxf(ny1, __shfl_xor(ny1, m));
            }
            if (lane == 0)
                chunkRect[ch] = make_float4(nx0, nx1, ny0, ny1);
        }
        __syncthreads();

        // ---- phase II: the wave walks TLAS and BLAS for its strip.  All control
        //      flow below is wave-uniform.
        const uint32_t numChunks = (n + kWave - 1) / kWave;
        uint32_t qCount = 0, sp = 0, chunk = 0, curInst = 0;
        uint64_t instMask = 0;
        float sMV[3][3] = {}, sTv[3] = {};
        bool done = false;
        for (;;) {
            while (qCount < (uint32_t)kWave && !done) {
                if (sp > 0) {
                    --sp;
                    const uint32_t ref = rflu(ws->stack[sp]);
                    if (ref & kBvhLeafBit) {
                        // ---- leaf: its triangles join the queue
                        const uint32_t cnt = ((ref >> kBvhLeafStartBits) & 15u) + 1u;
                        const uint32_t start = ref & ((1u << kBvhLeafStartBits) - 1u);
                        if ((uint32_t)lane < cnt)
                            ws->queue[qCount + lane] = make_uint2(curInst, p.bvhLeafTris[start + lane]);
                        qCount += cnt;
                    } else {
                        // ---- inner node: lane = (child, box corner)
                        const BvhNode *nd = p.bvhNodes + ref;
                        const int c = lane >> 3, corner = lane & 7;
                        const uint32_t cref = nd->child[c];
                        const float cx = (corner & 1) ? nd->bmax[c][0] : nd->bmin[c][0];
                        const float cy = (corner & 2) ? nd->bmax[c][1] : nd->bmin[c][1];
                        const float cz = (corner & 4) ? nd->bmax[c][2] : nd->bmin[c][2];
                        float fx, fz;
                        bool f;
                        projectCorner(p, sMV, sTv, cx, cy, cz, isx, isz, fx, fz, f);
                        float x0 = fx, x1 = fx, z0 = fz, z1 = fz;
                        int fr = f ? 1 : 0;
#pragma unroll
                        for (int m = 1; m < 8; m <<= 1) {
                            x0 = fminf(x0, __shfl_xor(x0, m));
                            x1 = fmaxf(x1, __shfl_xor(x1, m));
                            z0 = fminf(z0, __shfl_xor(z0, m));
                            z1 = fmaxf(z1, __shfl_xor(z1, m));
                            fr &= __shfl_xor(fr, m);
                        }
                        const Rect r = finishRect(p, x0, x1, z0, z1, fr != 0);
                        const bool hit = cref != kBvhEmpty && corner == 0 && overlaps(r, SX0, SX1, SY0, SY1);
                        uint64_t hm = __ballot(hit);
                        while (hm) {
                            const int l = __builtin_ctzll(hm);
                            hm &= hm - 1;
                            ws->stack[sp++] = (uint32_t)__builtin_amdgcn_readlane((int)cref, l);
                        }
                    }
                } else if (instMask) {
                    // ---- next instance of the TLAS node whose rectangle meets the strip
                    const int b = __builtin_ctzll(instMask);
                    instMask &= instMask - 1;
                    curInst = (chunk - 1u) * kWave + (uint32_t)b;
                    const float *rec = instRec + (size_t)curInst * kInstRecDw;
                    const uint32_t first = rflu(__float_as_uint(rec[21]));
                    const uint32_t num = rflu(__float_as_uint(rec[22]));
                    const int32_t root = (int32_t)rflu(__float_as_uint(rec[23]));
                    if (root < 0) {
                        // flat object (<= kBvhFlatMax triangles): all of them
                        if ((uint32_t)lane < num)
                            ws->queue[qCount + lane] = make_uint2(curInst, first + lane);
                        qCount += num;
                    } else {
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
#pragma unroll
                            for (int cc = 0; cc < 3; ++cc)
                                sMV[r][cc] = rfl(rec[3 * r + cc]);
                            sTv[r] = rfl(rec[9 + r]);
                        }
                        ws->stack[sp++] = (uint32_t)root;
                    }
                } else if (chunk < numChunks) {
                    // ---- next TLAS node (64 instances): lane = instance
                    const float4 cr = chunkRect[chunk];
                    Rect nr;
                    nr.x0 = rfl(cr.x); nr.x1 = rfl(cr.y); nr.y0 = rfl(cr.z); nr.y1 = rfl(cr.w);
                    if (overlaps(nr, SX0, SX1, SY0, SY1)) {
                        const uint32_t li = chunk * kWave + lane;
                        const float4 ir = instRect[li < n ? li : 0u];
                        Rect r;
                        r.x0 = ir.x; r.x1 = ir.y; r.y0 = ir.z; r.y1 = ir.w;
                        instMask = __ballot(li < n && overlaps(r, SX0, SX1, SY0, SY1));
                    }
                    ++chunk;
                } else {
                    done = true;
                }
                waveLdsSync();
            }
            if (qCount == 0)
                break;

            // ---- leaf test, 64 candidates at a time.  Lane = triangle: S3-S7 with
            //      the instance's transform from the TLAS record.
            const uint32_t nb = qCount < (uint32_t)kWave ? qCount : (uint32_t)kWave;
            bool live = false;
            uint32_t rgbaL = 0;
            int32_t texL = -1, objL = -1;
            float bbX0 = 0.f, bbX1 = 0.f;
            if ((uint32_t)lane < nb) {
                const uint2 e = ws->queue[lane];
                const float4 *rec = reinterpret_cast<const float4 *>(instRec + (size_t)e.x * kInstRecDw);
                const float4 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4], a5 = rec[5];
                InstXform x;
                x.MV[0][0] = a0.x; x.MV[0][1] = a0.y; x.MV[0][2] = a0.z; x.MV[1][0] = a0.w;
                x.MV[1][1] = a1.x; x.MV[1][2] = a1.y; x.MV[2][0] = a1.z; x.MV[2][1] = a1.w;
                x.MV[2][2] = a2.x; x.tv[0] = a2.y; x.tv[1] = a2.z; x.tv[2] = a2.w;
                x.qo[0] = a3.x; x.qo[1] = a3.y; x.qo[2] = a3.z; x.det = a3.w;
                x.sc[0] = a4.x; x.sc[1] = a4.y; x.sc[2] = a4.z;
                const int32_t obj = __float_as_int(a4.w);
                const int32_t k = (int32_t)(__float_as_uint(a5.x) + (e.y - __float_as_uint(a5.y)));
                TriPlanes c;
                float shade[4], cold[kCold];
                const bool valid = setupTriangleCore(p, vc.lv, x, e.y, obj, k, c, shade, cold);
                live = valid && c.bbX1 >= SX0 && c.bbX0 <= SX1 && c.bbY1 >= SY0 && c.bbY0 <= SY1;
                float4 *dst = reinterpret_cast<float4 *>(ws->planes[lane]);
                dst[0] = make_float4(c.A0, c.A1, c.A2, c.Dx);
                dst[1] = make_float4(c.B0, c.B1, c.B2, c.Dy);
                dst[2] = make_float4(c.C0, c.C1, c.C2, c.Dc);
                dst[3] = make_float4(__int_as_float((k << kSlotBits) | lane), 0.f, 0.f, 0.f);
                rgbaL = __float_as_uint(shade[0]);
                texL = __float_as_int(shade[1]);
                objL = obj;
                bbX0 = c.bbX0;
                bbX1 = c.bbX1;
                if (TEX && texL >= 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i)
                        coldLds[lane][i] = cold[i];
                }
            }
            waveLdsSync();
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float HX0 = SX0 + 32.0f * hf;
                uint64_t act = __ballot(live && bbX1 >= HX0 && bbX0 <= HX0 + 31.0f);
                const f32x2 yy = { py, py };
                for (; act != 0; act &= act - 1) {
                    const int slot = __builtin_ctzll(act);
                    const PlanePairs q = loadPlanes(ws->planes, slot);
                    const int32_t keyv = (int32_t)rflu(__float_as_uint(ws->planes[slot][12]));
                    const f32x2 r01 = fma2(q.B01, yy, q.C01);
                    const f32x2 r2d = fma2(q.B2D, yy, q.C2D);
#pragma unroll
                    for (int b = 0; b < kRegionBlocks; ++b)
                        pixelTestTie(q, r01, r2d, (float)(tileX0 + hf * 32 + 4 * lx + b), invNear, keyv,
                                     1u << (hf * kRegionBlocks + b), best[hf][b], key[hf][b], changed);
                }
            }
            // ---- shade this batch's winners before its records are replaced: colour
            //      and objectID come from the lane that set the winner up
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int b = 0; b < kRegionBlocks; ++b) {
                    const bool has = (changed >> (hf * kRegionBlocks + b)) & 1u;
                    const int32_t slot = key[hf][b] & ((1 << kSlotBits) - 1);
                    const int src = (has ? slot : 0) << 2;
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)rgbaL);
                    rgba[hf][b] = has ? c0 : rgba[hf][b];
                    if (IDS == 2) {
                        const int32_t s0 = __builtin_amdgcn_ds_bpermute(src, objL);
                        seg[hf][b] = has ? s0 : seg[hf][b];
                    }
                    if (TEX) {
                        const int32_t tex = __builtin_amdgcn_ds_bpermute(src, texL);
                        if (has && tex >= 0)
                            rgba[hf][b] = shadeTextured(p, coldLds[slot], tex,
                                                        (float)(tileX0 + hf * 32 + 4 * lx + b), py,
                                                        1.0f / best[hf][b]);
                    }
                }
            changed = 0;
            waveLdsSync();
            // ---- what did not fit the batch moves to the front of the queue
            if (qCount > (uint32_t)kWave) {
                const uint32_t rem = qCount - kWave;
                uint2 e = make_uint2(0u, 0u);
                if ((uint32_t)lane < rem)
                    e = ws->queue[kWave + lane];
                waveLdsSync();
                if ((uint32_t)lane < rem)
                    ws->queue[lane] = e;
                waveLdsSync();
                qCount = rem;
            } else {
                qCount = 0;
            }
            if (done && qCount == 0)
                break;
        }
    }

    // ---- output: depth = 1/best (v_rcp_f32, <= 1 ulp), one 16-byte store per
    //      tensor and half
    if (view >= p.numViews || (p.debugSkip & 1u))
        return;
    const size_t tileBase = ((size_t)view * p.nslow + tileY0) * p.nfast + tileX0;
    const bool full = (p.nfast & 3u) == 0 && tileX0 + 64u <= p.nfast && tileY0 + 64u <= p.nslow;
    const uint32_t fy = tileY0 + 8u * wave + ly;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const uint32_t fx0 = tileX0 + hf * 32 + 4 * lx;
        const size_t o = tileBase + (size_t)(8u * wave + ly) * p.nfast + hf * 32 + 4 * lx;
        uint32_t dep[kRegionBlocks], id[kRegionBlocks];
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b) {
            dep[b] = key[hf][b] >= 0 ? __float_as_uint(__builtin_amdgcn_rcpf(best[hf][b])) : 0u;
            id[b] = (uint32_t)(IDS == 2 ? seg[hf][b] : key[hf][b] >> kSlotBits);
        }
        if (full) {
            streamStore16(p.writeThrough, p.rgb + o, rgba[hf][0], rgba[hf][1], rgba[hf][2], rgba[hf][3]);
            streamStore16(p.writeThrough, p.depth + o, dep[0], dep[1], dep[2], dep[3]);
            if (IDS)
                streamStore16(p.writeThrough, p.ids + o, id[0], id[1], id[2], id[3]);
        } else if (fy < p.nslow) {
#pragma unroll
            for (int b = 0; b < kRegionBlocks; ++b)
                if (fx0 + b < p.nfast) {
                    streamStore4(p.writeThrough, p.rgb + o + b, rgba[hf][b]);
                    streamStore4(p.writeThrough, p.depth + o + b, dep[b]);
                    if (IDS)
                        streamStore4(p.writeThrough, p.ids + o + b, id[b]);
                }
        }
    }
}

}  // namespace

size_t bvhLdsBytes(uint32_t passInst, bool textured)
{
    return (size_t)passInst * kInstRecDw * 4 + (size_t)passInst * 16 + (size_t)(passInst / kWave) * 16 +
           sizeof(WaveScratch) * kBvhWaves + (textured ? (size_t)kBvhWaves * kWave * kCold * 4 : 0);
}

hipError_t launchBvh(const RasterParams &p, hipStream_t stream)
{
    const uint32_t items = p.numViews * p.tilesFast * p.tilesSlow;
    if (items == 0)
        return hipSuccess;
    const int ids = p.ids == nullptr ? 0 : p.idsAreSegmask ? 2 : 1;
    const bool tex = p.anyTextured != 0;
    const size_t lds = bvhLdsBytes(p.bvhPassInst, tex);
    const dim3 grid(items), block(kWave * kBvhWaves);
#define MRX_BVH(I, T)                                                                          \
    do {                                                                                       \
        static size_t allowed = 0;                                                             \
        if (lds > allowed) {                                                                   \
            const hipError_t e = hipFuncSetAttribute((const void *)bvhTraceKernel<I, T>,       \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess)                                                               \
                return e;                                                                      \
            allowed = lds;                                                                     \
        }                                                                                      \
        bvhTraceKernel<I, T><<<grid, block, lds, stream>>>(p);                                 \
    } while (0)
    if (ids == 2) {
        if (tex) MRX_BVH(2, true); else MRX_BVH(2, false);
    } else if (ids == 1) {
        if (tex) MRX_BVH(1, true); else MRX_BVH(1, false);
    } else {
        if (tex) MRX_BVH(0, true); else MRX_BVH(0, false);
    }
#undef MRX_BVH
    return hipGetLastError();
}

}  // namespace mrx
