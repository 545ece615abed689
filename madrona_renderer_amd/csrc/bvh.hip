// BVH ray-trace path for gfx950 (MI355X): primary rays through a two-level
// bounding volume hierarchy -- the counterpart of the reference's Raytracer
// render graph (/root/reference/src/mgr.cpp:443-492: per-world TLAS over the
// instances, per-object BLAS from AssetProcessor::makeBVHData, closest hit per
// pixel).  Used for worlds too large for the group kernel's triangle slots
// (raster.hip), in both render modes: visibility is defined once (DESIGN.md
// section 3) and this kernel computes the same function of the scene.
//
// Wave64 design: the primary rays are not traced one per lane.  One workgroup
// (one wave per 64x8 strip) owns a tile; its rays are coherent, so the
// hierarchy is walked ONCE per tile with wave-uniform control flow and the 64
// lanes are used across the geometry:
//   TLAS  lane = instance: transform, S6b quantities and the padded screen
//         rectangle of the bounding sphere of the object's box, in LDS, rebuilt
//         every step from the live pose tensors (phase I); 64 rectangles are
//         tested against the tile per instruction.
//   BLAS  8-wide nodes x 8 box corners = 64 lanes: each lane projects one
//         corner of one child box, an 8-lane reduction gives the child's
//         rectangle; hit children go on a per-wave stack in LDS.  The waves
//         share the work by instance / by child of the root.
//   leaf  candidate triangles are queued and set up 64 at a time (lane =
//         triangle): the S6 edge / 1-over-depth planes of the spec, exactly as
//         the raster kernels and the oracle compute them, once per tile.
//   pixels  the tile's depth buffer is 64-bit words in LDS merged with
//         ds_max_u64; small triangles are walked by (triangle, row) items dealt
//         over the lanes, large ones go on a shared list every wave rasterises
//         over its own strip after a barrier; winners are shaded from a table
//         of records in LDS.
//   launch  one workgroup per tile, per run of a view's tiles over one TLAS build, or -- one-tile views --
//         per pair of views whose TLASes two waves build side by side (MULTI); where all workgroups are
//         resident at once, the one dispatched second to a CU starts at wave priority 1, because the
//         arbiters serve the older workgroup first (kernel head).
// A box test only ever skips work: rectangles are padded so that no triangle
// that could own a pixel is dropped, and the pixel test itself is the spec's.
// Traversal order is not draw order, so the winner is chosen by (1/depth, then
// lower world-local triangle index) -- the total order the oracle's in-order
// strict '>' scan induces.  DESIGN.md section 4.2 has the measurements.
//
// Compiled with -ffp-contract=off like raster.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>

#include "bvh.hpp"
#include "raster_dev.hpp"

namespace mrx {
namespace {

constexpr int kQueueCap = 64 + 32;      // a flush takes 64; one append adds <= 32
constexpr int kInstRecDw = 24;          // MV[9] tv[3] qo[3] det sc[3] obj kBase firstTri numTris root

constexpr uint32_t kRootFlag = 0x40000000u;  // stack entry: the root node of an instance's BLAS

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

// Inclusive prefix sum over the 64 lanes in six DPP adds: shifts by 1, 2, 4, 8 within
// the rows of 16, then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2, 3
// (row_bcast exists on gfx9 / CDNA).  Lanes a step does not reach add the `old` value 0.
__device__ __forceinline__ int waveInclusiveSum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Camera rotation / position of a view and the light direction in its frame (S2, S4).
template <typename PARAMS>
__device__ __forceinline__ void loadViewConst(const PARAMS &p, uint32_t view, ViewConst &vc)
{
    const float4 q = *reinterpret_cast<const float4 *>(p.camRot + 4 * view);
    quatToMat(q.x, q.y, q.z, q.w, vc.Rc);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        vc.c[r] = p.camPos[3 * view + r];
        vc.lv[r] = dot3(vc.Rc[0][r], vc.Rc[1][r], vc.Rc[2][r], p.toLight[0], p.toLight[1], p.toLight[2]);
    }
}

// One corner of a box under MV / tv: its image position in pixels, whether it
// lies safely in front of the eye plane (far enough, relative to the size of
// the terms it is summed from, for the quotient to be meaningful), and whether
// it lies safely behind it.
__device__ __forceinline__ void projectCorner(const RasterParams &p, const float (&MV)[3][3],
                                              const float (&tv)[3], float cx, float cy, float cz,
                                              float isx, float isz, float &fx, float &fz, bool &front,
                                              bool &behind)
{
    float P[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        P[r] = __builtin_fmaf(MV[r][2], cz, __builtin_fmaf(MV[r][1], cy, __builtin_fmaf(MV[r][0], cx, tv[r])));
    const float scale = fabsf(MV[1][0] * cx) + fabsf(MV[1][1] * cy) + fabsf(MV[1][2] * cz) + fabsf(tv[1]);
    front = P[1] > 1e-3f * scale && P[1] > 1e-6f;
    // safely behind the eye plane: a box whose eight corners all are holds nothing visible
    behind = P[1] < -1e-3f * scale;
    const float iw = __builtin_amdgcn_rcpf(P[1]);
    fx = (P[0] * iw - p.ox) * isx;
    fz = (P[2] * iw - p.oz) * isz;
}

// Image bounds -> padded storage rectangle.  Not `front`: the box reaches the
// eye plane, its image is unbounded -- always visit.
template <typename PARAMS>
__device__ __forceinline__ Rect finishRect(const PARAMS &p, float x0, float x1, float z0,
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

// Padded storage rectangle of the bounding sphere of an instance's object box: the box
// (omin, omax) under M = R diag(s) lies within |diag(s) h| of its centre, h the half
// extents.  Conservative like the corner projection: centre and radius are widened by the
// rounding of the sums they come from, the quotients use the near or far side of the
// sphere whichever widens the interval, and a sphere that reaches the eye plane gives the
// unbounded rectangle.
template <typename PARAMS>
__device__ __forceinline__ Rect sphereRect(const PARAMS &p, const InstXform &x, float4 omin, float4 omax,
                                           float isx, float isz)
{
    const float c[3] = { 0.5f * (omin.x + omax.x), 0.5f * (omin.y + omax.y), 0.5f * (omin.z + omax.z) };
    const float h[3] = { (omax.x - c[0]) * x.sc[0], (omax.y - c[1]) * x.sc[1], (omax.z - c[2]) * x.sc[2] };
    float P[3], pad[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        P[r] = __builtin_fmaf(x.MV[r][2], c[2], __builtin_fmaf(x.MV[r][1], c[1], __builtin_fmaf(x.MV[r][0], c[0], x.tv[r])));
        pad[r] = 2e-6f * (fabsf(x.MV[r][0] * c[0]) + fabsf(x.MV[r][1] * c[1]) + fabsf(x.MV[r][2] * c[2]) + fabsf(x.tv[r]));
    }
    // (the half extents are those of the box about the ROUNDED centre: max(omax - c, c - omin) differ by an ulp)
    const float R = sqrtf(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]) * 1.002f + 1e-30f;
    const float yn = P[1] - R - pad[1], yf = P[1] + R + pad[1];
    const bool front = yn > 1e-3f * (pad[1] * 5e5f) && yn > 1e-6f;
    const float in = __builtin_amdgcn_rcpf(yn), ifa = __builtin_amdgcn_rcpf(yf);
    const float xl = P[0] - R - pad[0], xh = P[0] + R + pad[0];
    const float zl = P[2] - R - pad[2], zh = P[2] + R + pad[2];
    const float qxl = xl * (xl >= 0.0f ? ifa : in), qxh = xh * (xh >= 0.0f ? in : ifa);
    const float qzl = zl * (zl >= 0.0f ? ifa : in), qzh = zh * (zh >= 0.0f ? in : ifa);
    const float fa = (qxl - p.ox) * isx, fb = (qxh - p.ox) * isx;
    const float ga = (qzl - p.oz) * isz, gb = (qzh - p.oz) * isz;
    Rect r = finishRect(p, fminf(fa, fb), fmaxf(fa, fb), fminf(ga, gb), fmaxf(ga, gb), front);
    if (yf < 0.0f) {                                  // the whole sphere behind the eye plane: nothing visible
        r.x0 = r.y0 = __builtin_inff();
        r.x1 = r.y1 = -__builtin_inff();
    }
    return r;
}

// ---------------------------------------------------------------------------
// The tile's depth buffer lives in LDS: one 64-bit word per pixel,
//   high 32 bits  1/depth of the nearest hit so far (a positive float: its bit
//                 pattern orders like the value)
//   low 32 bits   (~k & 0x1FFFFF) << 10 | slot   -- k the world-local triangle
//                 index, slot where the winner's shading record sits (never 0
//                 and never 0xFFFFFFFF: worlds hold fewer than 2^21 - 1 triangles)
// and every path merges into it with one ds_max_u64: a larger 1/depth wins, and
// among equal 1/depth the LOWER triangle index -- the total order the oracle's
// in-order scan with a strict '>' induces, so the result does not depend on
// the order the hierarchy is walked in.  A low word of 0 means "no hit".
// ---------------------------------------------------------------------------
#ifndef MRX_BVH_DIAG
#define MRX_BVH_DIAG 0
#endif
constexpr int kSlotBits = 10;
constexpr uint32_t kKeyMask = 0x1FFFFFu;        // 2M triangles per world
// The slot field's top value marks a pixel whose winner was resolved in an earlier round of
// the tile (the record table is reused from round to round): its colour / segmask label
// already sit in the output tensors.  Being the largest slot value it also wins the 64-bit
// maximum against the same triangle taken again with a fresh slot.
constexpr uint32_t kStashed = (1u << kSlotBits) - 1u;


// Tile shapes (the LDS tile-size sweep of BASELINE configs[2]): TW x TH pixels per workgroup, one wave per
// TW x 8 strip, so TH / 8 waves; the depth buffer takes TW * TH * 8 bytes of LDS.
// (64x64 tiles, untextured: 768 records and 64 list entries cost cube fields of up to 4994 triangles nothing against
// 1024 and 96 -- profiles/r03_bvh_priority.txt -- and leave room for two TLAS blocks of 104 instances beside two
// workgroups per CU; worlds of BLAS meshes, the CLS instantiations, lose 16 % with them and keep the larger ones)
// Textured instantiations (round 4): ONE table of 48-byte records -- u/v planes (or, for an untextured triangle of a
// textured scene, the packed colour), lit colour, texel offset, width | height << 16 (0 = untextured), object id --
// instead of a 16-byte record plus a 48-byte one, and as many of them as the launch's LDS has room for (p.bvhTexCap, chosen
// by the host: 256 records now take 12 KB, which lets worlds of up to 104 instances render two views per workgroup; worlds
// that do not get pairs take up to 512).  A round that fills the table stashes its winners and starts over, so every
// record more is a later round boundary (profiles/r04_bvh_textured.txt).
#ifndef MRX_TEX_BIGCAP
#define MRX_TEX_BIGCAP 64
#endif
constexpr int tabCap(bool tex, int tw, int th, bool cls) { return tex ? 0 : (tw * th >= 4096 ? (cls ? 1024 : 768) : 512); }
constexpr size_t tabBytes(bool tex, int cap) { return tex ? (size_t)cap * 48u : (size_t)cap * 16u; }
constexpr int tabUsable(int cap) { return cap < (int)kStashed ? cap : (int)kStashed; }   // slot kStashed is the marker
constexpr int bigCap(int tw, int th, bool cls, bool tex = false) { return tw * th >= 4096 && cls ? 96 : (tex ? MRX_TEX_BIGCAP : 64); }   // (>= 64: one batch always fits an empty list)

// Row stride of the tile's depth buffer = tile width + kZPad pixels (8 bytes each).  At a stride of 64 pixels =
// 128 dwords, rows fall on the same LDS banks: the (triangle, row) items of the small-triangle walk are consecutive
// rows of one triangle at the same x, lane by lane -- an up to 16-way conflict on every ds_max_u64 -- and the eight
// rows of a strip pass collide four ways (profiles/r03_bvh482_pmc_sq.txt: conflict replays >= useful LDS cycles).
#ifndef MRX_BVH_ZPAD
#define MRX_BVH_ZPAD 1
#endif
constexpr int kZPad = MRX_BVH_ZPAD;

// width | height << 16 of a texture descriptor (offset, width, height) as the unified record holds it;
// the host refuses textures beyond 65535 texels a side
__device__ __forceinline__ float packTexDims(float4 texDesc)
{
    return __uint_as_float((__float_as_uint(texDesc.y) & 0xFFFFu) | (__float_as_uint(texDesc.z) << 16));
}

struct WaveScratch {
    // (instance of the pass, object triangle)
    uint2 queue[kQueueCap];
    uint32_t stack[kBvhStackCap];
};
static_assert(sizeof(WaveScratch) % 16 == 0, "WaveScratch alignment");

__device__ __forceinline__ unsigned long long packHit(float it, uint32_t low)
{
    return ((unsigned long long)__float_as_uint(it) << 32) | low;
}


// ---------------------------------------------------------------------------
// Resolve of a wave's strip: every lane looks up the winners of its pixels (four consecutive
// pixels of one row in each 32-pixel half) in the round's record table and shades them.
//
// FINAL = false -- the end of a round that is not the tile's last (the record table is about
// to be reused): winners of this round are written to the output tensors pixel by pixel
// (colour, and the segmask label under IDS = 2) and their depth-buffer words are marked
// kStashed.  Nothing is carried in registers from round to round: the kernel used to hold
// the resolved colours and labels of its eight pixels in sixteen registers through every
// loop of the traversal, which the Raytracer-mode instantiations paid for in scratch
// (44 - 62 spilled VGPRs with textures: 2.2x the algorithmic HBM traffic on C5).
// FINAL = true -- the tile's last round: resolve and output in one pass, one 16-byte store
// per tensor and half; pixels marked kStashed take the colour / label this lane stored in
// an earlier round back from the tensor (only lanes that have such pixels load anything;
// tiles that resolve in a single round -- the common case -- never do).
// Depth (1/best, v_rcp_f32) and visibility ids come from the depth buffer itself.
// ---------------------------------------------------------------------------
// What resolveStrip reads of the kernel's parameters.  The in-loop (FINAL = false) call fills
// it from the kernel-argument segment through a laundered pointer, right where it is needed:
// taken from `p` these eight dwords would sit in scalar registers through the whole traversal
// (the kernel has none to spare: every scalar spilled costs a v_writelane / v_readlane pair
// in some loop).
struct ResolveArgs {
    uint32_t *rgb;
    float *depth;
    int32_t *ids;
    const uint32_t *texels;
    uint32_t nfast, nslow, writeThrough;
};
typedef const __attribute__((address_space(4))) RasterParams *KernargParams;
// the same for the triangle set-up of a batch (setupTriangleCore)
struct SetupArgs {
    const ObjTri *tris;
    const TriMat *triMats;
    float sx, ox, sz, oz, s6bPad, ambient, diffuse;
    int32_t transposed;
};

template <int IDS, bool TEX, int TW, int TH, bool FINAL, int ZS = TW>
__device__ __forceinline__ void resolveStrip(const ResolveArgs p, unsigned long long *zbuf, const float4 *shadeTab,
                                             const float (*coldTab)[kCold], uint32_t view, uint32_t tileX0,
                                             uint32_t tileY0, int wave, int lane)
{
    constexpr int kHalves = TW / 32;
    {
        // Both instantiations sit inside the kernel's loops (rounds, tiles of the group), and most
        // of what they compute is loop-invariant: left alone the compiler hoists the pixel addresses
        // of both halves out of the loops and carries them through the traversal in registers it does
        // not have.  Laundering the two values they all derive from keeps the arithmetic in here.
        asm volatile("" : "+v"(lane));
        asm volatile("" : "+s"(view));
    }
    const int lx = lane & 7, ly = lane >> 3;
    const size_t tileBase = ((size_t)view * p.nslow + tileY0) * p.nfast + tileX0;
    const bool full = (p.nfast & 3u) == 0 && tileX0 + TW <= p.nfast && tileY0 + TH <= p.nslow;
    const uint32_t fy = tileY0 + 8u * wave + ly;
#pragma unroll
    for (int hf = 0; hf < kHalves; ++hf) {
        unsigned long long *zrow = zbuf + (8 * wave + ly) * ZS + 32 * hf + 4 * lx;
        const uint32_t fx0 = tileX0 + hf * 32 + 4 * lx;
        const size_t o = tileBase + (size_t)(8u * wave + ly) * p.nfast + hf * 32 + 4 * lx;
        uint32_t rgba[kRegionBlocks], low[kRegionBlocks], itBits[kRegionBlocks], texSlot[kRegionBlocks];
        int32_t seg[kRegionBlocks];
        bool mine[kRegionBlocks], texOn[kRegionBlocks], anyTexOn = false, anyStashed = false;
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b) {
            const unsigned long long z = zrow[b];
            low[b] = (uint32_t)z;
            itBits[b] = (uint32_t)(z >> 32);
            const uint32_t slot = low[b] & kStashed;
            // a hit that is not marked belongs to this round: every earlier round ended with
            // its winners marked, and slots are handed out before any pixel is written
            mine[b] = low[b] != 0u && slot != kStashed;
            anyStashed = anyStashed || (low[b] != 0u && slot == kStashed);
            if (TEX) {
                // the unified record: [0] packed colour of an untextured triangle, [8..11] = lit.b, texel offset,
                // width | height << 16 (0: untextured), object id
                const float *recT = coldTab[mine[b] ? slot : 0u];
                const float4 tail = reinterpret_cast<const float4 *>(recT)[2];
                rgba[b] = mine[b] ? __float_as_uint(recT[0]) : 0xFF000000u;
                seg[b] = mine[b] ? __float_as_int(tail.w) : -1;
                texOn[b] = mine[b] && __float_as_uint(tail.z) != 0u;
            } else {
                const float4 rec = shadeTab[mine[b] ? slot : 0u];
                rgba[b] = mine[b] ? __float_as_uint(rec.x) : 0xFF000000u;
                seg[b] = mine[b] ? __float_as_int(rec.z) : -1;
                texOn[b] = false;
            }
            texSlot[b] = texOn[b] ? slot : 0u;
            anyTexOn = anyTexOn || texOn[b];
        }
        // Textured winners: the texel loads of the half's four pixels are all issued before any
        // is used (addresses of untextured pixels point at texel 0) -- under per-pixel branches
        // every load waited for the one before it.
        if (TEX && __ballot(anyTexOn) != 0) {
            uint32_t texAddr[kRegionBlocks], texel[kRegionBlocks];
#pragma unroll
            for (int b = 0; b < kRegionBlocks; ++b) {
                const float *cold = coldTab[texSlot[b]];
                const float it = __uint_as_float(itBits[b]);
                // S8, as shadeTextured() of raster_dev.hpp (same operations in the same order)
                const float px = (float)(fx0 + b), py = (float)fy;
                const float tt = 1.0f / it;
                const float u = __builtin_fmaf(cold[0], px, __builtin_fmaf(cold[1], py, cold[2])) * tt;
                const float v = __builtin_fmaf(cold[3], px, __builtin_fmaf(cold[4], py, cold[5])) * tt;
                const int tw = (int)(__float_as_uint(cold[10]) & 0xFFFFu), th = (int)(__float_as_uint(cold[10]) >> 16);
                const float uf = u - floorf(u);
                float vf = v - floorf(v);
                vf = 1.0f - vf;
                int tx = (int)(uf * (float)tw);
                int ty = (int)(vf * (float)th);
                tx = tx > tw - 1 ? tw - 1 : tx;
                ty = ty > th - 1 ? th - 1 : ty;
                tx = tx < 0 ? 0 : tx;
                ty = ty < 0 ? 0 : ty;
                texAddr[b] = texOn[b] ? __float_as_uint(cold[9]) + (uint32_t)ty * (uint32_t)tw + (uint32_t)tx : 0u;
            }
#pragma unroll
            for (int b = 0; b < kRegionBlocks; ++b)
                texel[b] = p.texels[texAddr[b]];
#pragma unroll
            for (int b = 0; b < kRegionBlocks; ++b)
                if (texOn[b]) {
                    const float *cold = coldTab[texSlot[b]];
                    const uint32_t r8 = toU8(((float)(texel[b] & 255u) * (1.0f / 255.0f)) * cold[6]);
                    const uint32_t g8 = toU8(((float)((texel[b] >> 8) & 255u) * (1.0f / 255.0f)) * cold[7]);
                    const uint32_t b8 = toU8(((float)((texel[b] >> 16) & 255u) * (1.0f / 255.0f)) * cold[8]);
                    rgba[b] = r8 | (g8 << 8) | (b8 << 16) | 0xFF000000u;
                }
        }
        if (!FINAL) {
            // stash this round's winners in the tensors and mark them (only the low word:
            // between the barrier ahead of the large pass and the one that opens the next
            // round nobody else touches the pixels of this strip)
            const bool anyMine = mine[0] || mine[1] || mine[2] || mine[3];
            if (full && !anyStashed) {
                // no pixel of the four holds an earlier round's colour: one 16-byte store
                // (pixels without a winner yet get the background, which whoever wins them
                // later -- or the tile's last resolve -- overwrites)
                if (anyMine) {
                    streamStore16(p.writeThrough, p.rgb + o, rgba[0], rgba[1], rgba[2], rgba[3]);
                    if (IDS == 2)
                        streamStore16(p.writeThrough, p.ids + o, (uint32_t)seg[0], (uint32_t)seg[1], (uint32_t)seg[2],
                                      (uint32_t)seg[3]);
                }
            } else {
#pragma unroll
                for (int b = 0; b < kRegionBlocks; ++b)
                    if (mine[b] && fx0 + b < p.nfast && fy < p.nslow) {
                        streamStore4(p.writeThrough, p.rgb + o + b, rgba[b]);
                        if (IDS == 2)
                            streamStore4(p.writeThrough, p.ids + o + b, (uint32_t)seg[b]);
                    }
            }
#pragma unroll
            for (int b = 0; b < kRegionBlocks; ++b)
                if (mine[b])
                    reinterpret_cast<uint32_t *>(zrow + b)[0] = low[b] | kStashed;
            continue;
        }
        // pixels resolved in an earlier round: their colour / label come back from the tensors
        if (__ballot(anyStashed) != 0) {
            if (anyStashed && full) {
                const u32x4 prgb = streamLoad16(p.rgb + o);
                u32x4 pseg = { 0u, 0u, 0u, 0u };
                if (IDS == 2)
                    pseg = streamLoad16(p.ids + o);
#pragma unroll
                for (int b = 0; b < kRegionBlocks; ++b)
                    if (low[b] != 0u && (low[b] & kStashed) == kStashed) {
                        rgba[b] = prgb[b];
                        seg[b] = (int32_t)pseg[b];
                    }
            } else if (anyStashed) {
#pragma unroll
                for (int b = 0; b < kRegionBlocks; ++b)
                    if (low[b] != 0u && (low[b] & kStashed) == kStashed && fx0 + b < p.nfast && fy < p.nslow) {
                        rgba[b] = streamLoad4(p.rgb + o + b);
                        if (IDS == 2)
                            seg[b] = (int32_t)streamLoad4(p.ids + o + b);
                    }
            }
        }
        uint32_t dep[kRegionBlocks], id[kRegionBlocks];
#pragma unroll
        for (int b = 0; b < kRegionBlocks; ++b) {
            dep[b] = low[b] != 0u ? __float_as_uint(__builtin_amdgcn_rcpf(__uint_as_float(itBits[b]))) : 0u;
            id[b] = IDS == 2 ? (uint32_t)seg[b]
                             : (low[b] != 0u ? (~(low[b] >> kSlotBits) & kKeyMask) : 0xFFFFFFFFu);
        }
        if (full) {
            streamStore16(p.writeThrough, p.rgb + o, rgba[0], rgba[1], rgba[2], rgba[3]);
            streamStore16(p.writeThrough, p.depth + o, dep[0], dep[1], dep[2], dep[3]);
            if (IDS)
                streamStore16(p.writeThrough, p.ids + o, id[0], id[1], id[2], id[3]);
        } else if (fy < p.nslow) {
#pragma unroll
            for (int b = 0; b < kRegionBlocks; ++b)
                if (fx0 + b < p.nfast) {
                    streamStore4(p.writeThrough, p.rgb + o + b, rgba[b]);
                    streamStore4(p.writeThrough, p.depth + o + b, dep[b]);
                    if (IDS)
                        streamStore4(p.writeThrough, p.ids + o + b, id[b]);
                }
        }
    }
}

// The instance rows [i0, i1) of a view's world
template <typename PARAMS>
__device__ __forceinline__ void viewInstances(const PARAMS &p, uint32_t view, uint32_t &i0, uint32_t &i1)
{
    // (uniform worlds: arithmetic instead of two dependent loads)
    if (p.bvhUniInst) {
        uint32_t world = view;
        if (p.bvhUniCams != 1)
            world = view / p.bvhUniCams;
        i0 = world * p.bvhUniInst;
        i1 = i0 + p.bvhUniInst;
    } else {
        const uint32_t world = p.viewWorld[view];
        i0 = p.worldInstStart[world];
        i1 = p.worldInstStart[world + 1];
    }
}

// Phase I for chunks chFirst, chFirst + chStep, ... of the nI instances of a pass (rows from passBase on),
// lane = instance: the TLAS records and rectangles of a view, into the block at `rec` ([passInst][24] records,
// then [passInst] rectangles).
template <typename PARAMS>
__device__ __forceinline__ void tlasChunks(const PARAMS &p, const ViewConst &vc, uint32_t passBase, uint32_t nI,
                                           uint32_t chFirst, uint32_t chStep, float *rec, uint32_t passInst, float isx,
                                           float isz, int lane)
{
    float4 *const rects = reinterpret_cast<float4 *>(rec + (size_t)passInst * kInstRecDw);
    for (uint32_t ch = chFirst; ch * kWave < nI; ch += chStep) {
        const uint32_t li = ch * kWave + (uint32_t)lane;
        const bool has = li < nI;
        const uint32_t row = passBase + (has ? li : 0u);
        const int32_t obj = p.instObj[row];
        const float4 *oi = reinterpret_cast<const float4 *>(p.instInfo + row);
        const float4 o0 = oi[0], omin = oi[1], omax = oi[2];
        const uint32_t kBase = p.instKBase[row];
        InstXform x;
        instanceTransform(p, vc, row, x);
        Rect r = sphereRect(p, x, omin, omax, isx, isz);
        if (!(obj >= 0) || __float_as_uint(o0.y) == 0u) {    // nothing to draw: a rectangle nothing meets
            r.x0 = r.y0 = __builtin_inff();
            r.x1 = r.y1 = -__builtin_inff();
        }
        if (has) {
            float4 *dst = reinterpret_cast<float4 *>(rec + (size_t)li * kInstRecDw);
            dst[0] = make_float4(x.MV[0][0], x.MV[0][1], x.MV[0][2], x.MV[1][0]);
            dst[1] = make_float4(x.MV[1][1], x.MV[1][2], x.MV[2][0], x.MV[2][1]);
            dst[2] = make_float4(x.MV[2][2], x.tv[0], x.tv[1], x.tv[2]);
            dst[3] = make_float4(x.qo[0], x.qo[1], x.qo[2], x.det);
            dst[4] = make_float4(x.sc[0], x.sc[1], x.sc[2], __int_as_float(obj));
            dst[5] = make_float4(__uint_as_float(kBase), o0.x, o0.y, o0.z);
            rects[li] = make_float4(r.x0, r.x1, r.y0, r.y1);
        }
    }
}

// IDS: 0 = no id tensor, 1 = visibility ids (world-local triangle index),
// 2 = segmask (objectID of the winner's instance)
// CLS: exact per-strip classification of the listed large triangles (64x64 tiles only)
// MULTI: groups of one-tile views whose worlds fit one TLAS pass (p.bvhGroupViews > 1; 64x64 tiles only).  A
// workgroup renders p.bvhGroupViews consecutive VIEWS: their TLASes are built side by side, by different waves,
// in one phase I -- a phase most waves of a workgroup sit out (one wave's worth of instances) and whose length
// is the latency of its pose loads, not their number -- then one tile after the other.  A separate
// instantiation: it has no pass loop (fewer scalar registers live through the traversal), and the plain
// kernel has none to spare for the group's state (482-triangle worlds, one view per workgroup: 25.4 us
// without it, 26.5 us with the state compiled in).
template <int IDS, bool TEX, int TW, int TH, bool CLS = false, bool MULTI = false>
__global__ __launch_bounds__(kWave *(TH / 8), 4)
void bvhTileKernel(const RasterParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // (touchKernelArguments(), which buys the raster kernels 0.1 - 0.4 us per launch, measured nothing here --
    // 482-triangle worlds 25.4 -> 26.0 us, textured 35.6 -> 35.4: this kernel's first phase waits for its pose
    // loads and a barrier either way, and the seven dwords cost the untextured instantiation three VGPRs)
    constexpr int kBvhWaves = TH / 8;             // one wave per TW x 8 strip of the tile
    constexpr int ZS = TW + kZPad;                // row stride of the depth buffer, in pixels (kZPad above)
    constexpr int kHalves = TW / 32;              // 32-pixel halves of a strip: 4 pixels of a lane each
    // records a round can hold: a constant of the untextured instantiations; the textured ones take it from the host
    // (p.bvhTexCap: as many 48-byte records as fit beside the launch's TLAS blocks, mrx_api.cpp chooseBvhGroups)
    constexpr int kCapC = tabCap(false, TW, TH, CLS);
    const uint32_t kCap = TEX ? p.bvhTexCap : (uint32_t)kCapC;
    const uint32_t kUsable = TEX ? min(kCap, kStashed) : (uint32_t)tabUsable(kCapC);
    constexpr bool kPartial = TEX || CLS;
    constexpr int kBigCap = bigCap(TW, TH, CLS, TEX);
    static_assert((TW == 64 || TW == 32) && (TH == 64 || TH == 32), "tile shapes of the sweep");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x % kWave;
    const uint32_t tilesFast = (p.nfast + TW - 1) / TW, tilesSlow = (p.nslow + TH - 1) / TH;
    const uint32_t tilesPerView = tilesFast * tilesSlow;
    // XCD-aware item order for views of several tiles: consecutive workgroups run on consecutive
    // XCDs (eight, round-robin), each with its own L2, so with item = blockIdx the tiles of a view
    // would be spread over all of them and every XCD would fetch the view's camera, poses and
    // texels from HBM for itself.  When the grid divides by eight, XCD x takes the x-th eighth of
    // the items instead: the tiles of a view follow each other on one XCD.
    uint32_t item = blockIdx.x;
    if ((tilesPerView > 1 || p.bvhGroupTiles > 1) && (gridDim.x & 7u) == 0)
        item = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // A workgroup renders p.bvhGroupTiles consecutive tiles of one view, one after the other,
    // over ONE build of the world's TLAS (worlds that fit a single TLAS pass; the launcher gives
    // one tile per workgroup otherwise): the instance transforms, their screen rectangles and
    // the view constants are per view, not per tile.
    // (MULTI: one-tile views, workgroup b renders views b groupViews, b groupViews + 1, ...)
    const uint32_t groupTiles = MULTI ? 1u : p.bvhGroupTiles;
    const uint32_t groupViews = MULTI ? (p.bvhGroupViews & 0xFFFFu) : 1u;
    const uint32_t groupsPerView = (tilesPerView + groupTiles - 1) / groupTiles;
    uint32_t view = MULTI ? blockIdx.x * groupViews : item / groupsPerView;
    uint32_t tile = MULTI ? 0u : (item - view * groupsPerView) * groupTiles;
    // tiles this workgroup has left to render
    uint32_t left = MULTI ? min(groupViews, p.numViews - view) : min(groupTiles, tilesPerView - tile);
    if (MULTI && groupViews == 2) {
        // Pairs and single views in one launch: with g workgroups for n views (g <= n <= 2 g) the first n - g
        // workgroups render two views each, the others one -- all pairs when g = ceil(n / 2); between one and two
        // views per resident workgroup the host launches as many workgroups as the chip holds, so that 513 ... 1023
        // views run as one generation (the pairs on the workgroups dispatched first, which the arbiters favour).
        const uint32_t pairs = p.numViews - gridDim.x;
        view = blockIdx.x < pairs ? 2u * blockIdx.x : blockIdx.x + pairs;
        left = blockIdx.x < pairs ? 2u : 1u;
    }
    uint32_t tileX0 = (tile % tilesFast) * TW, tileY0 = (tile / tilesFast) * TH;
    //   Wave priority against the age order (MULTI, launches whose groups all run at once).  A CU holds two of
    // these workgroups, and its instruction arbiters serve the older wave first: the workgroup dispatched second to a
    // CU -- index >= the number of CUs: the dispatcher gives every CU one workgroup before any gets its second --
    // runs in the gaps the first one leaves.  Measured on 512 workgroups of two 482-triangle views each (in-kernel
    // clock per view, profiles/r03_bvh_priority.txt): workgroups 0..255 take 15.3 us for their two views,
    // workgroups 256..511 17.9 us -- a step at 256 that follows the index whatever views are dealt where -- and the
    // launch ends with the slow half.  The younger workgroup therefore runs its first view at wave priority 1 and
    // its second at 0 (bits 17..19 of p.bvhGroupViews = 2; bits 20..31 = the first young index): 16.5 / 16.9 us,
    // the launch 24.2 -> 21.8 us.  (1 = priority 1 throughout: the step turns round, 18.0 / 15.3; 3 = and the older
    // workgroup at priority 1 in its second view: as 2.  Launches of several generations lose with any of them --
    // every later workgroup would be `young` -- and keep the hardware's order.)
    //   One view (or one run of tiles) per workgroup, all workgroups resident at once: the same, with the younger
    // workgroup at priority 1 up to the barrier that closes its first produce phase (512 views of 482 triangles
    // 13.6 -> 13.2 us, of 1202 triangles 24.5 -> 22.7).
    const uint32_t prioMode = (p.bvhGroupViews >> 17) & 7u;
    const bool young = blockIdx.x >= (p.bvhGroupViews >> 20);
    if (prioMode && young)
        __builtin_amdgcn_s_setprio(1);
    const uint32_t passInst = p.bvhPassInst;
    const uint32_t dskip = MRX_BVH_DIAG ? p.debugSkip : 0u;
    if (dskip & 16u)
        return;                                       // timing aid: bare launch
    // Diagnostics (in-kernel stamps, MRX_DEBUG_STAMPS=1; phases switched off, MRX_DEBUG_SKIP)
    // exist only in a build with -DMRX_BVH_DIAG=1 (scripts/ab_build.sh diag -DMRX_BVH_DIAG=1):
    // the flags and the stamp pointer cost scalar registers in loops that have none to spare.
    unsigned long long *stamps = (MRX_BVH_DIAG && p.debugStamps && wave < 4)
        ? p.debugStamps + ((size_t)blockIdx.x * 4 + wave) * 8 : nullptr;
#define MRX_STAMP(i)                                                           \
    do {                                                                       \
        if (MRX_BVH_DIAG && stamps && lane == 0)                               \
            stamps[i] = __builtin_amdgcn_s_memrealtime();                      \
    } while (0)
    MRX_STAMP(0);

    // ---- LDS: depth buffer of the tile, shading records of the pass, control
    //      words, the TLAS of the pass, per-wave scratch
    unsigned long long *zbuf = reinterpret_cast<unsigned long long *>(smem);           // [TH][TW]
    // the record table: [kCap] float4 (rgba, texture, object id, k) -- or, TEX, [kCap][12] unified records (tabCap above)
    float4 *shadeTab = reinterpret_cast<float4 *>(zbuf + ZS * TH);
    float (*coldTab)[kCold] = reinterpret_cast<float (*)[kCold]>(shadeTab);
    uint32_t *ctrl = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(shadeTab) +
                                                  (TEX ? (size_t)kCap * 48u : (size_t)kCapC * 16u));   // [16]: counters
    float (*bigList)[16] = reinterpret_cast<float (*)[16]>(ctrl + 16);                  // [kBigCap] planes, key, box
    WaveScratch *ws = reinterpret_cast<WaveScratch *>(bigList + kBigCap) + wave;      // (fixed offsets first)
    // the TLAS of a view: the light direction in the view's frame and the number of instances (four
    // dwords), [passInst][24] records, [passInst] rectangles; p.bvhGroupViews of them
    // (every address below is the current view's block + a multiple of passInst: one loop-carried scalar)
    float *instRec = reinterpret_cast<float *>(ws - wave + kBvhWaves) + 4;
#define MRX_TLAS_DW (passInst * (kInstRecDw + 4u) + 4u)
#define MRX_INST_RECT(rec) reinterpret_cast<float4 *>((rec) + (size_t)passInst * kInstRecDw)
#define MRX_TLAS_HDR(rec) ((rec) - 4)

    // ---- view constants (wave-uniform).  Two placements, picked by what measured faster:
    //      the plain untextured instantiations work them out in every wave, here; the others
    //      only in the waves that transform instances, inside the pass loop, with the light
    //      direction handed to the set-up of all waves through LDS.
    constexpr bool kLvInLds = TEX || MULTI;
    ViewConst vcAll = {};
    if (!kLvInLds)
        loadViewConst(p, view, vcAll);
    // ---- which TLAS this wave helps to build in phase I: wave w works on view w % groupViews of the
    //      group (groupViews is a power of two), chunks w / groupViews, + 8 / groupViews, ... of its
    //      instances; the instance rows of that view (none: a view past the end of the group)
    const uint32_t vShift = MULTI ? (uint32_t)__builtin_ctz(groupViews) : 0u;
    const uint32_t tI = (uint32_t)wave & (groupViews - 1u), ch0 = (uint32_t)wave >> vShift;
    const uint32_t chStride = (uint32_t)kBvhWaves >> vShift;
    const uint32_t myView = view + tI;
    uint32_t i0 = 0, i1 = 0;
    if (tI < left)
        viewInstances(p, myView, i0, i1);
    const float isx = __builtin_amdgcn_rcpf(p.sx), isz = __builtin_amdgcn_rcpf(p.sz);
    const float invNear = p.invNear, invFar = p.invFar;
    float TX0 = (float)tileX0, TX1 = (float)(tileX0 + TW - 1);
    float TY0 = (float)tileY0, TY1 = (float)(tileY0 + TH - 1);
    const int smallArea = p.bvhSmallArea;

    {
        // the waves that have instances to transform in the first pass go straight to their
        // pose loads: the others clear the depth buffer for them
        // (groups of views: the waves that may have instances to transform, by the largest world)
        const uint32_t n0 = MULTI ? (p.bvhUniInst ? min(passInst, p.bvhUniInst) : passInst)
                                  : (i1 > i0 ? min(passInst, i1 - i0) : 0u);
        const uint32_t busy = min(((n0 + kWave - 1u) / kWave) << vShift, (uint32_t)kBvhWaves);
        if (busy >= (uint32_t)kBvhWaves) {
            for (int i = threadIdx.x; i < ZS * TH; i += kWave * kBvhWaves)
                zbuf[i] = packHit(invFar, 0u);
        } else if ((uint32_t)wave >= busy) {
            for (uint32_t i = ((uint32_t)wave - busy) * kWave + (uint32_t)lane; i < (uint32_t)(ZS * TH);
                 i += ((uint32_t)kBvhWaves - busy) * kWave)
                zbuf[i] = packHit(invFar, 0u);
        }
    }
    if (threadIdx.x == 0)
        ctrl[0] = ctrl[1] = ctrl[2] = ctrl[3] = ctrl[4] = ctrl[6] = 0u;   // records / done waves / large triangles;
                                                      // [4], [6]: records / large triangles of odd rounds;
                                                      // [3]: done waves of the group's odd tiles
    uint32_t doneIdx = 1;

    // the lane's pixels in the large pass and at resolve / output time: strip = wave, four
    // consecutive pixels of one row in each 32-pixel half
    const int lx = lane & 7, ly = lane >> 3;

    // the record and large-triangle counters alternate between two sets from
    // round to round, so the idle set can be prepared while the other is read
    uint32_t par = 0;
    uint32_t passBase = i0;
    do {                                              // (an empty world: one pass over no instances)
        // (groups of views: one pass, nI the instances of the view whose TLAS this wave works on)
        const uint32_t nI = i1 > passBase ? min(passInst, i1 - passBase) : 0u;
        uint32_t n = nI;
        const bool lastPass = passBase + passInst >= i1;
        if (passBase != i0) {
            // (the first pass has nothing to wait for: the barrier that closes phase I also
            // orders the clearing of the depth buffer above ahead of its first use)
            __syncthreads();                          // previous TLAS consumed
            if (threadIdx.x == 0)
                ctrl[doneIdx] = 0u;  // done waves (every wave has read the last pass's count by now); the record and
                                  // large-triangle counters of the coming round were set at the end of the last one
        }
        // ---- phase I: the TLAS of this pass, in LDS.  Lane = instance: its transform
        //      (S2/S3), the S6b quantities, and the padded screen rectangle of the bounding
        //      SPHERE of its object's box -- one wave's worth of work for 64 instances.
        //      (Projecting the eight box corners, lane = (instance, corner), gives a
        //      tighter rectangle for eight times the work, and the burst matters: every
        //      workgroup of a generation runs this phase at the same moment, so the vector
        //      units are saturated during it whatever their average load.  Measured on
        //      64^2 ... 256^2 views, 482 ... 4994 triangles: the sphere wins by 3-5 % everywhere.)
        //      The object's range, root and box were copied per instance at load: no load
        //      depends on another here (an instance whose object id is negative this step is hidden).
        //      (only the waves that have instances to transform work out the view constants; the wave
        //      with the view's first chunk always does, and leaves the light direction in LDS)
        ViewConst vc = vcAll;
        float *const myRec = instRec + (size_t)tI * MRX_TLAS_DW;
        if (kLvInLds && ch0 * kWave < nI)
            loadViewConst(p, myView, vc);
        tlasChunks(p, vc, passBase, nI, ch0, chStride, myRec, passInst, isx, isz, lane);
        // (the header after the records: ahead of them its write would wait for the camera loads before the
        // instance loads are even issued)
        if (kLvInLds && ch0 == 0 && lane == 0) {
            if (nI != 0)
                *reinterpret_cast<float4 *>(MRX_TLAS_HDR(myRec)) =
                    make_float4(vc.lv[0], vc.lv[1], vc.lv[2], __uint_as_float(nI));
            else
                MRX_TLAS_HDR(myRec)[3] = __uint_as_float(0u);   // (no instances: only the count is read)
        }
        __syncthreads();
        MRX_STAMP(1);

        for (;;) {                                    // the tiles of the group (one, unless the world fits one pass)
        if (MULTI)
            n = rflu(__float_as_uint(MRX_TLAS_HDR(instRec)[3]));   // (the instances of this tile's view)
        // ---- phase II: geometry.  The waves split the work by instance: flat
        //      objects round-robin, the eight children of a BLAS root one per
        //      wave.  All control flow of a wave is wave-uniform.
        uint32_t qCount = 0, sp = 0, chunk = 0, curInst = 0;
        const uint32_t numChunks = (n + kWave - 1) / kWave;
        uint64_t instMask = 0;
        uint32_t vFirst = 0, vNum = 0;                // of the lane's instance of the current TLAS chunk
        int32_t vRoot = -1;
        bool done = (dskip & 4u) != 0;          // timing aid: no traversal at all
        bool reported = false;
        for (;; par ^= 4u) {
            // -- produce until this wave's share is exhausted or the record table is full
            bool tableFull = false;
            while (!tableFull) {
                while (qCount < (uint32_t)kWave && !done) {
                    if (sp > 0) {
                        --sp;
                        const uint32_t ref = rflu(ws->stack[sp]);
                        if (ref & kBvhLeafBit) {
                            // leaf: its triangles join the queue
                            const uint32_t cnt = ((ref >> kBvhLeafStartBits) & 15u) + 1u;
                            const uint32_t start = ref & ((1u << kBvhLeafStartBits) - 1u);
                            if ((uint32_t)lane < cnt)
                                ws->queue[qCount + lane] = make_uint2(curInst, p.bvhLeafTris[start + lane]);
                            qCount += cnt;
                        } else {
                            // inner node: lane = (child, box corner)
                            const bool isRoot = (ref & kRootFlag) != 0;
                            const BvhNode *nd = p.bvhNodes + (ref & ~kRootFlag);
                            const int c = lane >> 3, corner = lane & 7;
                            const uint32_t cref = nd->child[c];
                            const float cx = (corner & 1) ? nd->bmax[c][0] : nd->bmin[c][0];
                            const float cy = (corner & 2) ? nd->bmax[c][1] : nd->bmin[c][1];
                            const float cz = (corner & 4) ? nd->bmax[c][2] : nd->bmin[c][2];
                            float fx, fz;
                            bool f, bh;
                            // (the instance's transform comes from its TLAS record per visit: kept in
                            // scalar registers across the loops it cost twelve of them, spilled)
                            const float4 *irec = reinterpret_cast<const float4 *>(instRec + (size_t)curInst * kInstRecDw);
                            const float4 a0 = irec[0], a1 = irec[1], a2 = irec[2];
                            const float nMV[3][3] = { { a0.x, a0.y, a0.z }, { a0.w, a1.x, a1.y }, { a1.z, a1.w, a2.x } };
                            const float nTv[3] = { a2.y, a2.z, a2.w };
                            projectCorner(p, nMV, nTv, cx, cy, cz, isx, isz, fx, fz, f, bh);
                            float x0 = fx, x1 = fx, z0 = fz, z1 = fz;
                            int fr = (f ? 1 : 0) | (bh ? 2 : 0);
#pragma unroll
                            for (int m = 1; m < 8; m <<= 1) {
                                x0 = fminf(x0, __shfl_xor(x0, m));
                                x1 = fmaxf(x1, __shfl_xor(x1, m));
                                z0 = fminf(z0, __shfl_xor(z0, m));
                                z1 = fmaxf(z1, __shfl_xor(z1, m));
                                fr &= __shfl_xor(fr, m);
                            }
                            const Rect r = finishRect(p, x0, x1, z0, z1, (fr & 1) != 0);
                            // the root's children are dealt one per wave
                            // (a child entirely behind the eye plane is dropped -- otherwise its
                            // unbounded rectangle would send all its triangles to the leaf test)
                            const bool hit = cref != kBvhEmpty && corner == 0 && (!isRoot || (c & (kBvhWaves - 1)) == wave) &&
                                             !(fr & 2) && overlaps(r, TX0, TX1, TY0, TY1);
                            uint64_t hm = __ballot(hit);
                            while (hm) {
                                const int l = __builtin_ctzll(hm);
                                hm &= hm - 1;
                                ws->stack[sp++] = (uint32_t)__builtin_amdgcn_readlane((int)cref, l);
                            }
                            waveLdsSync();
                        }
                    } else if (instMask) {
                        // next instance of this wave's share whose rectangle meets the tile:
                        // its range and root sit in the registers of the lane that tested it
                        const int b = __builtin_ctzll(instMask);
                        instMask &= instMask - 1;
                        curInst = (chunk - 1u) * kWave + (uint32_t)b;
                        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)vFirst, b);
                        const uint32_t num = (uint32_t)__builtin_amdgcn_readlane((int)vNum, b);
                        const int32_t root = __builtin_amdgcn_readlane(vRoot, b);
                        if (root < 0) {
                            // flat object (<= kBvhFlatMax triangles): all of them
                            if ((uint32_t)lane < num)
                                ws->queue[qCount + lane] = make_uint2(curInst, first + lane);
                            qCount += num;
                        } else {
                            ws->stack[sp++] = (uint32_t)root | kRootFlag;
                            waveLdsSync();
                        }
                    } else if (chunk < numChunks) {
                        // next 64 instances of the TLAS: lane = instance
                        const uint32_t li = chunk * kWave + lane;
                        const float4 ir = MRX_INST_RECT(instRec)[li < n ? li : 0u];
                        const float4 oi = *reinterpret_cast<const float4 *>(
                            instRec + (size_t)(li < n ? li : 0u) * kInstRecDw + 20);
                        Rect r;
                        r.x0 = ir.x; r.x1 = ir.y; r.y0 = ir.z; r.y1 = ir.w;
                        vFirst = __float_as_uint(oi.y);
                        vNum = __float_as_uint(oi.z);
                        vRoot = __float_as_int(oi.w);
                        const bool hitI = li < n && overlaps(r, TX0, TX1, TY0, TY1);
                        instMask = __ballot(hitI && (vRoot >= 0 || (lane & (kBvhWaves - 1)) == wave));
                        ++chunk;
                    } else {
                        done = true;
                    }
                }
                if (qCount == 0)
                    break;
                // (the record table of the round is full already: no point in setting the batch up)
                if (kPartial && rflu(ctrl[0 + par]) >= kUsable) {
                    tableFull = true;
                    break;
                }
                waveLdsSync();                        // the queue entries written above are read below

                // -- a batch of up to 64 candidates.  Leaf test setup, lane = triangle:
                //    S3-S7 with the instance's transform from the TLAS record.
                const uint32_t nb = qCount < (uint32_t)kWave ? qCount : (uint32_t)kWave;
                if (dskip & 128u) MRX_STAMP(2);
                bool live = false;
                TriPlanes c;
                c.A0 = c.B0 = c.C0 = c.A1 = c.B1 = c.C1 = 0.0f;
                c.A2 = c.B2 = c.C2 = c.Dx = c.Dy = c.Dc = 0.0f;
                c.bbX0 = c.bbX1 = c.bbY0 = c.bbY1 = 0.0f;
                uint32_t kTri = 0;
                int32_t objL = -1;
                // cold: [0] = |1/d| (the u/v planes are derived once the triangle has a slot), [6..8] lit colour
                float shade[4] = { 0.f, 0.f, 0.f, 0.f }, cold[kCold];
                uint32_t triL = 0;
                if ((uint32_t)lane < nb && !(dskip & 8u)) {
                    const uint2 e = ws->queue[lane];
                    triL = e.y;
                    const float4 *rec = reinterpret_cast<const float4 *>(instRec + (size_t)e.x * kInstRecDw);
                    const float4 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4], a5 = rec[5];
                    InstXform x;
                    x.MV[0][0] = a0.x; x.MV[0][1] = a0.y; x.MV[0][2] = a0.z; x.MV[1][0] = a0.w;
                    x.MV[1][1] = a1.x; x.MV[1][2] = a1.y; x.MV[2][0] = a1.z; x.MV[2][1] = a1.w;
                    x.MV[2][2] = a2.x; x.tv[0] = a2.y; x.tv[1] = a2.z; x.tv[2] = a2.w;
                    x.qo[0] = a3.x; x.qo[1] = a3.y; x.qo[2] = a3.z; x.det = a3.w;
                    x.sc[0] = a4.x; x.sc[1] = a4.y; x.sc[2] = a4.z;
                    objL = __float_as_int(a4.w);
                    kTri = __float_as_uint(a5.x) + (e.y - __float_as_uint(a5.y));
                    // (the light direction comes from LDS batch by batch: held in registers it
                    // costs scalar spills in every loop below)
                    float4 lv4 = make_float4(vcAll.lv[0], vcAll.lv[1], vcAll.lv[2], 0.0f);
                    if (kLvInLds)
                        lv4 = *reinterpret_cast<const float4 *>(MRX_TLAS_HDR(instRec));
                    const float lv[3] = { lv4.x, lv4.y, lv4.z };
                    // (what the set-up reads of the kernel's parameters, fetched per batch from the
                    // kernel-argument segment instead of living in scalar registers: see ResolveArgs)
                    KernargParams pk = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(pk));
                    const SetupArgs sa = { pk->tris, pk->triMats, pk->sx, pk->ox, pk->sz, pk->oz, pk->s6bPad,
                                           pk->ambient, pk->diffuse, pk->transposed };
                    const bool valid = setupTriangleCore<false>(sa, lv, x, e.y, objL, (int32_t)kTri, c, shade, cold);
                    live = valid && c.bbX1 >= TX0 && c.bbX0 <= TX1 && c.bbY1 >= TY0 && c.bbY0 <= TY1;
                    // The planes at the tile's corners: fl(A x + fl(B y + C)) is monotone in x and in
                    // y, so its extreme over the tile's pixels is taken at a corner pixel, and a
                    // triangle with an edge plane negative, or 1/depth outside (1/zfar, 1/znear], at
                    // the most favourable corner owns no pixel of the tile -- exactly, not
                    // approximately.  This is what drops triangles behind the eye (their box is
                    // unbounded: they would all count as large) and boxes that only graze the tile.
                    const float eMax0 = __builtin_fmaf(c.A0, c.A0 >= 0.0f ? TX1 : TX0, __builtin_fmaf(c.B0, c.B0 >= 0.0f ? TY1 : TY0, c.C0));
                    const float eMax1 = __builtin_fmaf(c.A1, c.A1 >= 0.0f ? TX1 : TX0, __builtin_fmaf(c.B1, c.B1 >= 0.0f ? TY1 : TY0, c.C1));
                    const float eMax2 = __builtin_fmaf(c.A2, c.A2 >= 0.0f ? TX1 : TX0, __builtin_fmaf(c.B2, c.B2 >= 0.0f ? TY1 : TY0, c.C2));
                    const float itMax = __builtin_fmaf(c.Dx, c.Dx >= 0.0f ? TX1 : TX0, __builtin_fmaf(c.Dy, c.Dy >= 0.0f ? TY1 : TY0, c.Dc));
                    const float itMin = __builtin_fmaf(c.Dx, c.Dx >= 0.0f ? TX0 : TX1, __builtin_fmaf(c.Dy, c.Dy >= 0.0f ? TY0 : TY1, c.Dc));
                    live = live && fminf(fminf(eMax0, eMax1), eMax2) >= 0.0f && itMax > invFar && itMin <= invNear;
                }
                // -- shading records only for triangles that can own a pixel of the tile
                //    (about half the candidates are back faces): slots by rank among them
                const uint64_t liveMask = __ballot(live);
                const uint32_t numLive = (uint32_t)__builtin_popcountll(liveMask);
                uint32_t slotBase = 0;
                if (lane == 0 && numLive)
                    slotBase = atomicAdd(&ctrl[0 + par], numLive);
                slotBase = rflu(slotBase);
                // The table fills up in the middle of the batch.  Instantiations whose tables do
                // overflow in practice (textured: 256 records; CLS: mesh worlds) process the
                // triangles that still fit and leave the others in the queue for the next round
                // (kPartial; the round ends for this wave after the batch) -- 100 textured cubes
                // 101 -> 93 us, mesh worlds 230 -> 218.  The plain untextured kernel, whose 1024
                // records rarely run out, keeps the shorter code (the extra state costs it 1 %):
                // it drops the batch and takes it again after the resolve.
                const uint32_t liveRank = (uint32_t)__builtin_popcountll(liveMask & ((1ull << lane) - 1ull));
                uint64_t defMask = 0;
                if (slotBase + numLive > kUsable) {
                    tableFull = true;
                    if (!kPartial)
                        break;                        // wait for the resolve, then take the batch again
                    const uint32_t fit = slotBase < kUsable ? kUsable - slotBase : 0u;
                    defMask = __ballot(live && liveRank >= fit);
                    live = live && liveRank < fit;
                }
                const uint32_t slot = slotBase + liveRank;
                const uint32_t lowKey = ((~kTri & kKeyMask) << kSlotBits) | slot;
                if (live) {
                    if (!TEX)
                        shadeTab[slot] = make_float4(shade[0], shade[1], shade[2], __uint_as_float(kTri));   // [2]: the triangle's object id
                    else if (__float_as_int(shade[1]) < 0) {
                        // an untextured triangle of a textured scene: packed colour, "no texture", object id
                        float *dst = coldTab[slot];
                        dst[0] = shade[0];
                        *reinterpret_cast<float2 *>(dst + 10) = make_float2(__uint_as_float(0u), shade[2]);
                    }
                    if (TEX && __float_as_int(shade[1]) >= 0) {
                        // the u/v planes from the edge planes, the texture coordinates (read again:
                        // L1 / L2 hits) and |1/d|; the texture's descriptor rides with the material
                        KernargParams pk = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
                        asm volatile("" : "+s"(pk));
                        const float4 *tsrc = reinterpret_cast<const float4 *>(pk->tris + triL);
                        const float4 t2 = tsrc[2], t3 = tsrc[3];
                        const float4 texDesc = reinterpret_cast<const float4 *>(pk->triMats + triL)[3];
                        const float uv[6] = { t2.y, t2.z, t2.w, t3.x, t3.y, t3.z };
                        float uvp[6];
                        uvPlanes(c, cold[0], uv, uvp);
                        float4 *cdst = reinterpret_cast<float4 *>(coldTab[slot]);
                        cdst[0] = make_float4(uvp[0], uvp[1], uvp[2], uvp[3]);
                        cdst[1] = make_float4(uvp[4], uvp[5], cold[6], cold[7]);
                        cdst[2] = make_float4(cold[8], texDesc.x, packTexDims(texDesc), shade[2]);
                    }
                }
                if (dskip & 128u) MRX_STAMP(3);
                // pixel range of the triangle inside the tile: the conservative box
                // of setup, less all but 1/32 of its one-pixel margin
                const float kTrim = 0.96875f;
                const float fx0 = ceilf(fmaxf(c.bbX0 + kTrim, TX0)), fx1 = floorf(fminf(c.bbX1 - kTrim, TX1));
                const float fy0 = ceilf(fmaxf(c.bbY0 + kTrim, TY0)), fy1 = floorf(fminf(c.bbY1 - kTrim, TY1));
                live = live && fx0 <= fx1 && fy0 <= fy1;
                const int ix0 = live ? (int)fx0 : 0, iy0 = live ? (int)fy0 : 0;
                const int bw = live ? (int)fx1 - ix0 + 1 : 0, bh = live ? (int)fy1 - iy0 + 1 : 0;
                const int area = bw * bh;
                const bool small = live && area <= smallArea;
                const bool big = live && !small;
                bool listFull = false;
                if (!(dskip & 2u)) {
                    // -- small triangles, by (triangle, row of its box): the rows of all the
                    //    batch's small triangles are numbered through (prefix sum over the
                    //    lanes) and dealt 64 at a time, a lane fetches the planes of its row's
                    //    triangle from the lane that set it up (ds_bpermute) and walks the row
                    //    four pixels per step -- every lane has a row, however uneven the boxes
                    {
                        const int rowsMine = (small && !(dskip & 32u)) ? bh : 0;
                        const int incl = waveInclusiveSum(rowsMine);
                        const int total = __builtin_amdgcn_readlane(incl, kWave - 1);
                        const int packedBox = ix0 | (bw << 16);        // (both < 2^15)
                        for (int item0 = 0; item0 < total; item0 += kWave) {
                            const int j = item0 + lane;
                            const bool act = j < total;
                            // the lane t whose rows hold item j: smallest t with incl[t] > j
                            int t = 0;
#pragma unroll
                            for (int step = kWave / 2; step >= 1; step >>= 1) {
                                const int probe = __builtin_amdgcn_ds_bpermute((t + step - 1) << 2, incl);
                                t += probe <= j ? step : 0;
                            }
                            t = act ? t : 0;
                            const int src = t << 2;
                            const int inclT = __builtin_amdgcn_ds_bpermute(src, incl);
                            const int rowsT = __builtin_amdgcn_ds_bpermute(src, rowsMine);
                            const int boxT = __builtin_amdgcn_ds_bpermute(src, packedBox);
                            const int y0T = __builtin_amdgcn_ds_bpermute(src, iy0);
                            const uint32_t lowT = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)lowKey);
#define MRX_GATHER(v) __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v)))
                            const f32x2 A01 = { MRX_GATHER(c.A0), MRX_GATHER(c.A1) }, A2D = { MRX_GATHER(c.A2), MRX_GATHER(c.Dx) };
                            const f32x2 B01 = { MRX_GATHER(c.B0), MRX_GATHER(c.B1) }, B2D = { MRX_GATHER(c.B2), MRX_GATHER(c.Dy) };
                            const f32x2 C01 = { MRX_GATHER(c.C0), MRX_GATHER(c.C1) }, C2D = { MRX_GATHER(c.C2), MRX_GATHER(c.Dc) };
#undef MRX_GATHER
                            const int xBeg = boxT & 0xFFFF, xEnd = xBeg + (boxT >> 16);
                            const int sy = y0T + (j - (inclT - rowsT));
                            const float py = (float)sy;
                            const f32x2 yy = { py, py };
                            const f32x2 r01 = fma2(B01, yy, C01);
                            const f32x2 r2d = fma2(B2D, yy, C2D);
                            unsigned long long *zline = zbuf + (sy - (int)tileY0) * ZS - (int)tileX0;
                            for (int sx = xBeg; __ballot(act && sx < xEnd) != 0; sx += 4) {
                                if (act && sx < xEnd) {
#pragma unroll
                                    for (int q4 = 0; q4 < 4; ++q4) {
                                        const float px = (float)(sx + q4);
                                        const f32x2 pp = { px, px };
                                        const f32x2 e01 = fma2(A01, pp, r01);
                                        const f32x2 e2d = fma2(A2D, pp, r2d);
                                        if (sx + q4 < xEnd && fminf(fminf(e01.x, e01.y), e2d.x) >= 0.0f &&
                                            e2d.y > invFar && e2d.y <= invNear)
                                            atomicMax(zline + sx + q4, packHit(e2d.y, lowT));
                                    }
                                }
                            }
                        }
                    }
                    if (dskip & 128u) MRX_STAMP(4);
                    // -- large triangles go on the tile's shared list: after the barrier all
                    //    eight waves rasterise them, each its own strip
                    const uint64_t bigMask = __ballot(big && !(dskip & 64u));
                    const int numBig = __builtin_popcountll(bigMask);
                    const int rank = __builtin_popcountll(bigMask & ((1ull << lane) - 1ull));
                    const uint32_t boxXY = (uint32_t)(ix0 - (int)tileX0) | ((uint32_t)(bw - 1) << 8) |
                                           ((uint32_t)(iy0 - (int)tileY0) << 16) | ((uint32_t)(bh - 1) << 24);
                    if (numBig) {
                        uint32_t bigBase = 0;
                        if (lane == 0)
                            bigBase = atomicAdd(&ctrl[2 + par], (uint32_t)numBig);
                        bigBase = rflu(bigBase);
                        if (bigBase + (uint32_t)numBig <= (uint32_t)kBigCap) {
                            if ((bigMask >> lane) & 1ull) {
                                float4 *dst = reinterpret_cast<float4 *>(bigList[bigBase + rank]);
                                dst[0] = make_float4(c.A0, c.A1, c.A2, c.Dx);
                                dst[1] = make_float4(c.B0, c.B1, c.B2, c.Dy);
                                dst[2] = make_float4(c.C0, c.C1, c.C2, c.Dc);
                                dst[3] = make_float4(__uint_as_float(lowKey), __uint_as_float(boxXY), 0.f, 0.f);
                            }
                        } else {
                            // The list is full (a close-up: most triangles are large).  The round ends
                            // here for this wave and the batch is taken again in the next one, with an
                            // empty list: everything done for it so far is idempotent (the depth buffer
                            // takes maxima, records are a function of the triangle), and whoever
                            // reserves first in a round always fits, so every round makes progress.
                            // The part of the reservation inside the list is blanked -- the pass reads
                            // every entry below the count.
                            if (bigBase < (uint32_t)kBigCap && (uint32_t)lane < (uint32_t)kBigCap - bigBase)
                                reinterpret_cast<float4 *>(bigList[bigBase + lane])[3] =
                                    make_float4(0.f, __uint_as_float(0x00FF0000u), 0.f, 0.f);
                            listFull = true;
                        }
                    }
                    waveLdsSync();
                }
                if (listFull) {
                    tableFull = true;
                    break;
                }
                // -- what did not fit the batch (or the record table) moves to the front of the queue
                if (qCount > (uint32_t)kWave || (kPartial && defMask != 0)) {
                    const uint32_t rem = qCount > (uint32_t)kWave ? qCount - kWave : 0u;
                    const uint32_t numDef = (uint32_t)__builtin_popcountll(defMask);
                    const bool deferred = ((defMask >> lane) & 1ull) != 0;
                    uint2 e = make_uint2(0u, 0u), own = make_uint2(0u, 0u);
                    if ((uint32_t)lane < rem)
                        e = ws->queue[kWave + lane];
                    if (deferred)
                        own = ws->queue[lane];
                    waveLdsSync();
                    if (deferred)
                        ws->queue[__builtin_popcountll(defMask & ((1ull << lane) - 1ull))] = own;
                    if ((uint32_t)lane < rem)
                        ws->queue[numDef + lane] = e;
                    waveLdsSync();
                    qCount = numDef + rem;
                } else {
                    qCount = 0;
                }
            }
            if (done && qCount == 0 && !reported) {
                reported = true;
                if (lane == 0)
                    atomicAdd(&ctrl[doneIdx], 1u);
            }
            if (dskip & 128u) MRX_STAMP(5); else MRX_STAMP(2);
            __syncthreads();
            if (!(dskip & 128u)) MRX_STAMP(3);
            if (!MULTI || left == 1) {
                // (the priority of a younger workgroup with one view ends here -- of one with two, where it turns to its
                // second; read from the argument block: no register held for it)
                KernargParams pk = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(pk));
                if (((pk->bvhGroupViews >> 17) & 7u) >= 2u)
                    __builtin_amdgcn_s_setprio(0);
            }
            // -- the round's large triangles: wave = strip, lane = entry for the box
            //    test, then four pixels of the lane per 32-pixel half.  From here to the
            //    end of the round a wave touches only the pixels of its own strip (the
            //    small-triangle walks ended at the barrier), so no barrier separates this
            //    pass from the resolve below.
            const bool allDone = rflu(ctrl[doneIdx]) == (uint32_t)kBvhWaves;
            // The record table lives on from round to round and from pass to pass until it is
            // full (records are per triangle, not per TLAS pass): only then are the round's
            // winners resolved and stashed and the table started afresh.  A round that ended on
            // a full large-triangle list, or a pass that ended with room to spare, carries the
            // count over.  (A reservation that did not fit has pushed the count past kUsable; a
            // round in which every wave finished had none.)
            // The end of a pass is a free place to start afresh (no wave holds a batch it would have to
            // set up again), so a table more than half full is not carried into the next pass: it would
            // fill up in the middle of it (4994-triangle worlds at 64x64: 61 us this way, 66 us carrying
            // everything; 256x256 views of the same worlds, whose tiles see a fraction each: 134 against 150).
            const uint32_t recCount = rflu(ctrl[0 + par]);
            const bool tableReset = recCount >= kUsable || (allDone && recCount > kUsable / 2u);
            if (threadIdx.x == 0) {                                   // the next round's counters
                // (the tile's last round: the next tile of the group starts with an empty table and
                // counts its finished waves in the other counter, cleared here, two barriers ahead of its use)
                ctrl[0 + (par ^ 4u)] = (tableReset || (allDone && lastPass)) ? 0u : recCount;
                ctrl[2 + (par ^ 4u)] = 0u;
                if (allDone && lastPass)
                    ctrl[doneIdx ^ 2u] = 0u;
            }
            {
                const uint32_t listed = min(rflu(ctrl[2 + par]), (uint32_t)kBigCap);
                for (uint32_t e0 = 0; e0 < listed; e0 += kWave) {
                    const uint32_t ent = e0 + (uint32_t)lane;
                    const uint32_t box = ent < listed ? __float_as_uint(bigList[ent][13]) : 0u;
                    const int bx0 = (int)(box & 255u), bx1 = bx0 + (int)((box >> 8) & 255u);
                    const int by0 = (int)((box >> 16) & 255u), by1 = by0 + (int)(box >> 24);
                    const bool rows = ent < listed && by0 <= 8 * wave + 7 && by1 >= 8 * wave;
                    // Can the lane's entry touch this wave's strip at all?  Its planes at the most
                    // favourable corner pixel of each 32x8 half (exact, by monotonicity -- as for the
                    // whole tile at the leaf test): a triangle that crosses the eye plane has an
                    // unbounded box but touches few regions, and the two halves of a ground quad
                    // split the tile between them.
                    // CLS is a kernel-level switch like TEX: the untextured kernel has no register
                    // to spare (cube fields lose 2-3 % with the test compiled in, mesh worlds gain
                    // 27 %), so the host picks the instantiation by whether the scene has meshes
                    // large enough for a BLAS.
                    bool reg[kHalves];
#pragma unroll
                    for (int hf = 0; hf < kHalves; ++hf)
                        reg[hf] = true;
                    if (CLS) {
                        const float4 *src = reinterpret_cast<const float4 *>(bigList[ent < listed ? ent : 0u]);
                        const float4 pa = src[0], pb = src[1], pc = src[2];
                        const float y0 = TY0 + (float)(8 * wave), y1 = y0 + 7.0f;
                        const float r0 = __builtin_fmaf(pb.x, pb.x >= 0.0f ? y1 : y0, pc.x);
                        const float r1 = __builtin_fmaf(pb.y, pb.y >= 0.0f ? y1 : y0, pc.y);
                        const float r2 = __builtin_fmaf(pb.z, pb.z >= 0.0f ? y1 : y0, pc.z);
                        const float dMax = __builtin_fmaf(pb.w, pb.w >= 0.0f ? y1 : y0, pc.w);
                        const float dMin = __builtin_fmaf(pb.w, pb.w >= 0.0f ? y0 : y1, pc.w);
#pragma unroll
                        for (int hf = 0; hf < kHalves; ++hf) {
                            const float x0 = TX0 + (float)(32 * hf), x1 = x0 + 31.0f;
                            const float e0 = __builtin_fmaf(pa.x, pa.x >= 0.0f ? x1 : x0, r0);
                            const float e1 = __builtin_fmaf(pa.y, pa.y >= 0.0f ? x1 : x0, r1);
                            const float e2 = __builtin_fmaf(pa.z, pa.z >= 0.0f ? x1 : x0, r2);
                            const float iMax = __builtin_fmaf(pa.w, pa.w >= 0.0f ? x1 : x0, dMax);
                            const float iMin = __builtin_fmaf(pa.w, pa.w >= 0.0f ? x0 : x1, dMin);
                            reg[hf] = fminf(fminf(e0, e1), e2) >= 0.0f && iMax > invFar && iMin <= invNear;
                        }
                    }
#pragma unroll
                    for (int hf = 0; hf < kHalves; ++hf) {
                        uint64_t act = __ballot(rows && reg[hf] && bx0 <= 32 * hf + 31 && bx1 >= 32 * hf);
                        if (act == 0)
                            continue;
                        // every covered pixel goes straight to the depth buffer: the 64-bit
                        // maximum orders (1/depth, then lower index) by itself, which a
                        // register copy of the running best would need two more compares
                        // and two selects per test to imitate
                        const float py = (float)(tileY0 + 8u * wave + ly);
                        const f32x2 yy = { py, py };
                        unsigned long long *zrow = zbuf + (8 * wave + ly) * ZS + 32 * hf + 4 * lx;
                        for (; act != 0; act &= act - 1) {
                            const int l = (int)e0 + __builtin_ctzll(act);
                            const PlanePairs q = loadPlanes(bigList, l);
                            const uint32_t lowv = rflu(__float_as_uint(bigList[l][12]));
                            const f32x2 r01 = fma2(q.B01, yy, q.C01);
                            const f32x2 r2d = fma2(q.B2D, yy, q.C2D);
#pragma unroll
                            for (int b = 0; b < kRegionBlocks; ++b) {
                                const float px = (float)(tileX0 + 32 * hf + 4 * lx + b);
                                const f32x2 pp = { px, px };
                                const f32x2 e01 = fma2(q.A01, pp, r01);       // e0, e1
                                const f32x2 e2d = fma2(q.A2D, pp, r2d);       // e2, 1/depth
                                if (fminf(fminf(e01.x, e01.y), e2d.x) >= 0.0f && e2d.y > invFar && e2d.y <= invNear)
                                    atomicMax(zrow + b, packHit(e2d.y, lowv));
                            }
                        }
                    }
                }
            }
            if (!(dskip & 128u)) MRX_STAMP(4);
            // -- resolve (resolveStrip above).  The tile's last round -- this wave and all others
            //    done, no further pass -- leaves the loops and resolves + outputs in one go below;
            //    a round that filled the record table stashes its winners in the tensors, because
            //    the table is reused from here on.  Nothing resolved is carried in registers.
            if (allDone && lastPass)
                break;
            if (tableReset) {
                // (the kernel's only argument sits at offset 0 of the kernel-argument segment)
                KernargParams pk = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(pk));
                const ResolveArgs ra = { pk->rgb, pk->depth, pk->ids, pk->texels, pk->nfast, pk->nslow, pk->writeThrough };
                resolveStrip<IDS, TEX, TW, TH, false, ZS>(ra, zbuf, shadeTab, coldTab, view, tileX0, tileY0, wave, lane);
            }
            if (!(dskip & 128u)) MRX_STAMP(5);
            if (allDone) {
                par ^= 4u;                            // (the counters prepared above are the next pass's)
                break;                                // (the next pass opens with a barrier)
            }
            __syncthreads();
        }
        if (!lastPass)
            break;                                    // on to the next TLAS pass of this (only) tile
        // ---- the tile's last resolve + output: depth = 1/best (v_rcp_f32, <= 1 ulp), one
        //      16-byte store per tensor and half
        if (!(dskip & 1u)) {
            if (!(dskip & 128u)) MRX_STAMP(5);
            KernargParams pk = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(pk));
            const ResolveArgs ra = { pk->rgb, pk->depth, pk->ids, pk->texels, pk->nfast, pk->nslow, pk->writeThrough };
            resolveStrip<IDS, TEX, TW, TH, true, ZS>(ra, zbuf, shadeTab, coldTab, view, tileX0, tileY0, wave, lane);
        }
        if (--left == 0)
            break;
        // ---- the next tile of the group: this wave's strip of the depth buffer is cleared (nobody else
        //      touches it between the barrier ahead of the large pass and the one below), the rectangle
        //      moves on, the traversal state starts over; the TLAS, the view constants and the light
        //      direction stay.  One barrier: every strip is clear before anyone merges into it.
        for (int i = lane; i < ZS * 8; i += kWave)
            zbuf[8 * wave * ZS + i] = packHit(invFar, 0u);
        if (MULTI) {
            // the next view of the group: its TLAS is the next block
            ++view;
            instRec += MRX_TLAS_DW;
            if (prioMode >= 2) {
                if (young)
                    __builtin_amdgcn_s_setprio(0);
                else if (prioMode == 3)
                    __builtin_amdgcn_s_setprio(1);
            }
        } else {
            ++tile;
            tileX0 = (tile % tilesFast) * TW;
            tileY0 = (tile / tilesFast) * TH;
            TX0 = (float)tileX0; TX1 = (float)(tileX0 + TW - 1);
            TY0 = (float)tileY0; TY1 = (float)(tileY0 + TH - 1);
        }
        par ^= 4u;
        doneIdx ^= 2u;
        __syncthreads();
        }
        if (MULTI)
            break;                                    // (groups of views: every world fits the one pass)
        passBase += passInst;
    } while (passBase < i1);
    MRX_STAMP(6);
    if (MRX_BVH_DIAG && stamps && lane == 0)     // where this wave ran: HW_ID (reg 4) and XCC_ID (reg 20)
        stamps[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                    (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#undef MRX_STAMP
#undef MRX_TLAS_DW
#undef MRX_INST_RECT
#undef MRX_TLAS_HDR
}


// ---------------------------------------------------------------------------
// Worlds of at most 64 triangles in at most 64 instance rows (every BASELINE scene: cube + plane
// is 14 triangles in 2 rows; configs[4] names this path for them): the leaf set-up of the WHOLE
// world is one 64-lane batch, so the machinery of the general kernel above -- instance
// rectangles, per-wave queues, record slots handed out by atomics, a shared list of large
// triangles behind a barrier -- only costs.  profiles/r04_c5bvh_pmc_sq.txt: on configs[4] a wave
// of that kernel spends 58 % of its life at barriers; per tile one wave sets the cube's twelve
// triangles up again and walks them while six of the eight wait.  Here:
//   once per VIEW   wave 0: lane = instance -> TLAS records in LDS (phase I as above), then
//                   lane = world-local triangle k -> S3-S7 set-up (setupTriangleCore, the same
//                   call), planes + box into triRec[k], the shading record into slot k + 1 -- the
//                   record table is the world, nothing is allocated and nothing can overflow.
//   once per view   (cont.) every lane keeps its triangle's planes in registers for the whole view, and the
//                   classification of the 64 triangles against EVERY tile of the group (box, then the planes at
//                   the tile's corner pixels, exact by monotonicity) is worked out once, wave w for tiles w, w + 8:
//                   a dword per (tile, triangle) in the LDS the TLAS records occupied (configs[4] 521 -> 496 us;
//                   before, every wave repeated ~70 dependent vector instructions at the head of every tile)
//   per tile        every wave reads the tile's 64 dwords, so all eight hold the same masks and prefix sums
//                   without talking;
//                   the (triangle, row) items of the small triangles are dealt 64 at a time round
//                   robin over the WAVES (the general kernel walks a batch in the wave that set it
//                   up); large triangles are rasterised by every wave over its own strip straight
//                   from triRec (after the exact per-strip test of the CLS instantiations).  One
//                   barrier -- walks done -- then resolve + output per strip as above.
//   depth buffer    one (46 KB of LDS in all: three workgroups per CU): a wave clears its strip after its
//                   resolve and a second barrier closes the tile.  The first form had two buffers -- tile i
//                   merged into buffer i & 1, whose next use lay behind the barrier of tile i + 1: one barrier
//                   per tile, 80 KB, two workgroups per CU -- and lost to this one (MRX_FLAT_ZBUFS above).
// The pixel test, the (1/depth, lower index) order and the records are those of the general
// kernel, so the output is the same bit for bit.  MRX_BVH_FLAT=0 keeps such worlds on it.
// (Measured and not kept, profiles/r04_flat_whole_ab.txt: large triangles that cover a 32x8 half entirely --
// all three edge planes >= 0 at the least favourable corner, the ground quad on most tiles -- walked with the
// 1/depth plane alone: configs[4] 519.4 -> 516.2 us, one-tile views 1.3 % slower.  The vector instructions of
// the large pass are not what a tile waits for.  Wave priority 1 for the resolve + stores of a strip: within noise.)
// ---------------------------------------------------------------------------
constexpr int kFlatTris = 64;                      // triangles and instance rows per world, at most
// Row stride of the flat kernel's depth buffers, in pixels: 64 + 1.  A lane of a strip pass owns pixels
// (4 lx + b, ly); with a stride of 64 pixels = 128 dwords the eight rows of a strip fall on the same banks
// (ds_read_b64: 32 lanes per LDS cycle, bank = dword mod 64 -- a 4-way conflict on every access of the large
// pass, the resolve and the clear; profiles/r04_c5bvh_pmc_sq.txt: conflict cycles = half of all LDS cycles).
// 130 dwords = 2 mod 64 puts the four rows of a lane group on four different bank pairs -- and measures 1.5 %
// SLOWER on every shape (configs[4] 520-527 -> 535-538 us, 4096 x 128^2 149.3 -> 151.4: rows that alternate
// between 8- and 16-byte alignment cost the resolve and the clear their 16-byte LDS accesses, and the LDS pipe
// is not what these tiles wait for).  Kept as a build switch; 64 is the default.
#ifndef MRX_FLAT_ZS
#define MRX_FLAT_ZS 64
#endif
constexpr int kFlatZS = MRX_FLAT_ZS;
// One depth buffer and two barriers per tile (46 KB of LDS: THREE workgroups per CU), or two buffers and one barrier
// (80 KB: two per CU -- the first form of this kernel).  Occupancy is worth more than the barrier
// (profiles/r04_flat_zbufs_ab.txt): 4096 x 128^2 + wall 145 -> 115 us, one-tile views 14.7 -> 13.9, configs[4] level in
// either placement mode (495 / 498 us in the fast one).
#ifndef MRX_FLAT_ZBUFS
#define MRX_FLAT_ZBUFS 1
#endif
constexpr uint32_t kFlatZBufs = MRX_FLAT_ZBUFS;
constexpr size_t kFlatZBytes = (size_t)kFlatZBufs * 64u * kFlatZS * 8u;
constexpr size_t flatLdsBytes(bool tex)
{
    return kFlatZBytes + kFlatTris * 64u + tabBytes(tex, kFlatTris + 2) + 16u +
           (size_t)kFlatTris * kInstRecDw * 4u;
}

template <int IDS, bool TEX>
__global__ __launch_bounds__(kWave * 8, MRX_FLAT_ZBUFS == 2 ? 4 : 6)
void bvhFlatKernel(const RasterParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int TW = 64, TH = 64, kWaves = 8, kHalves = 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int lane = threadIdx.x % kWave;
    const uint32_t tilesFast = (p.nfast + TW - 1) / TW, tilesSlow = (p.nslow + TH - 1) / TH;
    const uint32_t tilesPerView = tilesFast * tilesSlow;
    // (XCD-aware item order and runs of a view's tiles per workgroup: as in bvhTileKernel)
    uint32_t item = blockIdx.x;
    if ((tilesPerView > 1 || p.bvhGroupTiles > 1) && (gridDim.x & 7u) == 0)
        item = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t groupTiles = p.bvhGroupTiles;
    const uint32_t groupsPerView = (tilesPerView + groupTiles - 1) / groupTiles;
    const uint32_t view = item / groupsPerView;
    uint32_t tile = (item - view * groupsPerView) * groupTiles;
    uint32_t left = min(groupTiles, tilesPerView - tile);

    unsigned long long *zbuf = reinterpret_cast<unsigned long long *>(smem);                   // [2][TH][TW]
    float (*triRec)[16] = reinterpret_cast<float (*)[16]>(smem + kFlatZBytes);                 // [64] planes, box
    float4 *shadeTab = reinterpret_cast<float4 *>(triRec + kFlatTris);                          // [66]: slot k + 1 (TEX: the
    float (*coldTab)[kCold] = reinterpret_cast<float (*)[kCold]>(shadeTab);                     // unified 48-byte records)
    float *hdr = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(shadeTab) + tabBytes(TEX, kFlatTris + 2));
    float *instRec = hdr + 4;                                                                    // [64][24]

    unsigned long long *stamps = (MRX_BVH_DIAG && p.debugStamps && wave < 4)
        ? p.debugStamps + ((size_t)blockIdx.x * 4 + wave) * 8 : nullptr;
#define MRX_STAMP(i)                                                           \
    do {                                                                       \
        if (MRX_BVH_DIAG && stamps && lane == 0)                               \
            stamps[i] = __builtin_amdgcn_s_memrealtime();                      \
    } while (0)
    MRX_STAMP(0);
    const float invNear = p.invNear, invFar = p.invFar;
    if (wave != 0) {
        // both depth buffers, by the seven waves that have nothing to load
        for (uint32_t i = (uint32_t)(wave - 1) * kWave + (uint32_t)lane; i < kFlatZBufs * kFlatZS * TH; i += (kWaves - 1) * kWave)
            zbuf[i] = packHit(invFar, 0u);
    } else {
        ViewConst vc;
        loadViewConst(p, view, vc);
        InstXform y;
        int32_t objL = -1;
        uint32_t myTri = 0;
        bool hasT = false;
        if (p.uniInstances != 0) {
            // ---- uniform worlds (every world the same <= 4 objects in the same order: every BASELINE scene): which row
            //      and which object triangle lane k sets up is arithmetic on kernel arguments (raster.hpp), so its
            //      pose, camera and geometry loads are all requested at once -- one load level, no TLAS records in LDS
            const uint32_t world = p.uniCamsPerWorld > 1u ? view / p.uniCamsPerWorld : view;
            const uint32_t k = (uint32_t)lane;
            uint32_t i = (k >= p.uniPrefix[1] ? 1u : 0u) + (k >= p.uniPrefix[2] ? 1u : 0u) + (k >= p.uniPrefix[3] ? 1u : 0u);
            i = min(i, p.uniInstances - 1u);
            hasT = k < p.uniPrefix[4];
            const uint32_t pre = i == 0u ? 0u : i == 1u ? p.uniPrefix[1] : i == 2u ? p.uniPrefix[2] : p.uniPrefix[3];
            const uint32_t ft = i == 0u ? p.uniFirstTri[0] : i == 1u ? p.uniFirstTri[1] : i == 2u ? p.uniFirstTri[2] : p.uniFirstTri[3];
            myTri = hasT ? ft + (k - pre) : 0u;
            const uint32_t row = world * p.uniInstances + i;
            objL = p.instObj[row];
            instanceTransform(p, vc, row, y);
        } else {
        // ---- phase I, lane = instance row of the view's world
        uint32_t i0, i1;
        viewInstances(p, view, i0, i1);
        const uint32_t nI = min(i1 - i0, (uint32_t)kFlatTris);
        const bool hasI = (uint32_t)lane < nI;
        uint32_t kBase = 0, firstI = 0, numI = 0;
        if (nI != 0) {                                // (an empty world has no row to read)
            const uint32_t row = i0 + (hasI ? (uint32_t)lane : 0u);
            const int32_t obj = p.instObj[row];
            const float4 o0 = reinterpret_cast<const float4 *>(p.instInfo + row)[0];   // first, count, root of the bound object
            kBase = p.instKBase[row];
            InstXform x;
            instanceTransform(p, vc, row, x);
            if (hasI) {
                float4 *dst = reinterpret_cast<float4 *>(instRec + (size_t)lane * kInstRecDw);
                dst[0] = make_float4(x.MV[0][0], x.MV[0][1], x.MV[0][2], x.MV[1][0]);
                dst[1] = make_float4(x.MV[1][1], x.MV[1][2], x.MV[2][0], x.MV[2][1]);
                dst[2] = make_float4(x.MV[2][2], x.tv[0], x.tv[1], x.tv[2]);
                dst[3] = make_float4(x.qo[0], x.qo[1], x.qo[2], x.det);
                dst[4] = make_float4(x.sc[0], x.sc[1], x.sc[2], __int_as_float(obj));
                firstI = __float_as_uint(o0.x);
                numI = __float_as_uint(o0.y);
            }
        }
        // ---- lane = world-local triangle k: the instance that draws it is the row whose
        //      [kBase, kBase + count) holds k (rows are in index order; a hidden or unbound row
        //      keeps its range and fails the validity test of the set-up)
        uint32_t myInst = 0;
        for (uint32_t j = 0; j < nI; ++j) {
            const uint32_t kb = (uint32_t)__builtin_amdgcn_readlane((int)kBase, (int)j);
            const uint32_t nt = (uint32_t)__builtin_amdgcn_readlane((int)numI, (int)j);
            const uint32_t ft = (uint32_t)__builtin_amdgcn_readlane((int)firstI, (int)j);
            const bool in = (uint32_t)lane >= kb && (uint32_t)lane - kb < nt;
            myInst = in ? j : myInst;
            myTri = in ? ft + ((uint32_t)lane - kb) : myTri;
            hasT = hasT || in;
        }
        waveLdsSync();                                // the records written above are read below
        if (hasT) {
            const float4 *rec = reinterpret_cast<const float4 *>(instRec + (size_t)myInst * kInstRecDw);
            const float4 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3], a4 = rec[4];
            y.MV[0][0] = a0.x; y.MV[0][1] = a0.y; y.MV[0][2] = a0.z; y.MV[1][0] = a0.w;
            y.MV[1][1] = a1.x; y.MV[1][2] = a1.y; y.MV[2][0] = a1.z; y.MV[2][1] = a1.w;
            y.MV[2][2] = a2.x; y.tv[0] = a2.y; y.tv[1] = a2.z; y.tv[2] = a2.w;
            y.qo[0] = a3.x; y.qo[1] = a3.y; y.qo[2] = a3.z; y.det = a3.w;
            y.sc[0] = a4.x; y.sc[1] = a4.y; y.sc[2] = a4.z;
            objL = __float_as_int(a4.w);
        }
        }
        TriPlanes c;
        c.A0 = c.B0 = c.C0 = c.A1 = c.B1 = c.C1 = 0.0f;
        c.A2 = c.B2 = c.C2 = c.Dx = c.Dy = c.Dc = 0.0f;
        bool valid = false;
        if (hasT) {
            float shade[4] = { 0.f, 0.f, 0.f, 0.f }, cold[kCold];
            valid = setupTriangleCore<false>(p, vc.lv, y, myTri, objL, (int32_t)lane, c, shade, cold);
            if (!TEX)
                shadeTab[lane + 1] = make_float4(shade[0], shade[1], shade[2], __int_as_float(lane));
            else if (!(valid && __float_as_int(shade[1]) >= 0)) {
                float *dst = coldTab[lane + 1];
                dst[0] = shade[0];
                *reinterpret_cast<float2 *>(dst + 10) = make_float2(__uint_as_float(0u), shade[2]);
            }
            if (TEX && valid && __float_as_int(shade[1]) >= 0) {
                // (u/v planes as the general kernel derives them: uvPlanes() from the edge planes and |1/d|)
                const float4 *tsrc = reinterpret_cast<const float4 *>(p.tris + myTri);
                const float4 t2 = tsrc[2], t3 = tsrc[3];
                const float4 texDesc = reinterpret_cast<const float4 *>(p.triMats + myTri)[3];
                const float uv[6] = { t2.y, t2.z, t2.w, t3.x, t3.y, t3.z };
                float uvp[6];
                uvPlanes(c, cold[0], uv, uvp);
                float4 *cdst = reinterpret_cast<float4 *>(coldTab[lane + 1]);
                cdst[0] = make_float4(uvp[0], uvp[1], uvp[2], uvp[3]);
                cdst[1] = make_float4(uvp[4], uvp[5], cold[6], cold[7]);
                cdst[2] = make_float4(cold[8], texDesc.x, packTexDims(texDesc), shade[2]);
            }
        }
        {
            // a triangle that cannot own a pixel gets a box nothing meets
            const float inf = __builtin_inff();
            float4 *dst = reinterpret_cast<float4 *>(triRec[lane]);
            dst[0] = make_float4(c.A0, c.A1, c.A2, c.Dx);
            dst[1] = make_float4(c.B0, c.B1, c.B2, c.Dy);
            dst[2] = make_float4(c.C0, c.C1, c.C2, c.Dc);
            dst[3] = valid ? make_float4(c.bbX0, c.bbX1, c.bbY0, c.bbY1) : make_float4(inf, -inf, inf, -inf);
        }
    }
    MRX_STAMP(7);
    __syncthreads();
    MRX_STAMP(1);

    const int lx = lane & 7, ly = lane >> 3;
    const int smallArea = p.bvhSmallArea;
    const uint32_t lowKey = ((~(uint32_t)lane & kKeyMask) << kSlotBits) | ((uint32_t)lane + 1u);
    // ---- the lane's triangle: its planes and box stay in registers for the whole view ...
    TriPlanes c;
    {
        const float4 *src = reinterpret_cast<const float4 *>(triRec[lane]);
        const float4 pa = src[0], pb = src[1], pc = src[2], bb = src[3];
        c.A0 = pa.x; c.A1 = pa.y; c.A2 = pa.z; c.Dx = pa.w;
        c.B0 = pb.x; c.B1 = pb.y; c.B2 = pb.z; c.Dy = pb.w;
        c.C0 = pc.x; c.C1 = pc.y; c.C2 = pc.z; c.Dc = pc.w;
        c.bbX0 = bb.x; c.bbX1 = bb.y; c.bbY0 = bb.z; c.bbY1 = bb.w;
    }
    // ---- ... and its classification against every tile of the group is worked out ONCE, wave w for tiles w, w + 8
    //      (lane = triangle: box, then the planes at the tile's corner pixels, exact by monotonicity): a dword per
    //      (tile, triangle) in the LDS the TLAS records occupied -- box origin and size inside the tile, small / live --
    //      instead of every wave repeating ~70 dependent vector instructions at the head of every tile
    //      (a group of ONE tile classifies in place: the table and its barrier cost one-tile views 2 %)
    uint32_t *clsTab = reinterpret_cast<uint32_t *>(instRec);                  // [16][64]
    auto classify = [&](uint32_t tg) -> uint32_t {
        const uint32_t gx0 = (tg % tilesFast) * TW, gy0 = (tg / tilesFast) * TH;
        const float TX0 = (float)gx0, TX1 = (float)(gx0 + TW - 1);
        const float TY0 = (float)gy0, TY1 = (float)(gy0 + TH - 1);
        bool live = c.bbX1 >= TX0 && c.bbX0 <= TX1 && c.bbY1 >= TY0 && c.bbY0 <= TY1;
        {
            const float eMax0 = __builtin_fmaf(c.A0, c.A0 >= 0.0f ? TX1 : TX0, __builtin_fmaf(c.B0, c.B0 >= 0.0f ? TY1 : TY0, c.C0));
            const float eMax1 = __builtin_fmaf(c.A1, c.A1 >= 0.0f ? TX1 : TX0, __builtin_fmaf(c.B1, c.B1 >= 0.0f ? TY1 : TY0, c.C1));
            const float eMax2 = __builtin_fmaf(c.A2, c.A2 >= 0.0f ? TX1 : TX0, __builtin_fmaf(c.B2, c.B2 >= 0.0f ? TY1 : TY0, c.C2));
            const float itMax = __builtin_fmaf(c.Dx, c.Dx >= 0.0f ? TX1 : TX0, __builtin_fmaf(c.Dy, c.Dy >= 0.0f ? TY1 : TY0, c.Dc));
            const float itMin = __builtin_fmaf(c.Dx, c.Dx >= 0.0f ? TX0 : TX1, __builtin_fmaf(c.Dy, c.Dy >= 0.0f ? TY0 : TY1, c.Dc));
            live = live && fminf(fminf(eMax0, eMax1), eMax2) >= 0.0f && itMax > invFar && itMin <= invNear;
        }
        const float kTrim = 0.96875f;
        const float fx0 = ceilf(fmaxf(c.bbX0 + kTrim, TX0)), fx1 = floorf(fminf(c.bbX1 - kTrim, TX1));
        const float fy0 = ceilf(fmaxf(c.bbY0 + kTrim, TY0)), fy1 = floorf(fminf(c.bbY1 - kTrim, TY1));
        live = live && fx0 <= fx1 && fy0 <= fy1;
        const int jx0 = live ? (int)fx0 - (int)gx0 : 0, jy0 = live ? (int)fy0 - (int)gy0 : 0;
        const int jw = live ? (int)fx1 - (int)fx0 : 0, jh = live ? (int)fy1 - (int)fy0 : 0;       // size - 1
        const bool sm = (jw + 1) * (jh + 1) <= smallArea;
        return live ? ((uint32_t)jx0 | ((uint32_t)jw << 6) | ((uint32_t)jy0 << 12) | ((uint32_t)jh << 18) |
                       (sm ? 1u << 24 : 0u) | (1u << 25)) : 0u;
    };
    const bool table = left > 1u;
    if (table) {
        for (uint32_t g = (uint32_t)wave; g < left; g += kWaves)
            clsTab[g * kWave + (uint32_t)lane] = classify(tile + g);
        __syncthreads();
    }
    uint32_t buf = 0, g = 0;
    for (;;) {
        const uint32_t tileX0 = (tile % tilesFast) * TW, tileY0 = (tile / tilesFast) * TH;
        const float TX0 = (float)tileX0;
        const float TY0 = (float)tileY0;
        unsigned long long *zb = zbuf + (size_t)buf * (kFlatZS * TH);
        const uint32_t cls = table ? clsTab[g * kWave + (uint32_t)lane] : classify(tile);
        const bool live = (cls >> 25) != 0u;
        const int ix0 = (int)tileX0 + (int)(cls & 63u), iy0 = (int)tileY0 + (int)((cls >> 12) & 63u);
        const int bw = live ? (int)((cls >> 6) & 63u) + 1 : 0, bh = live ? (int)((cls >> 18) & 63u) + 1 : 0;
        const bool small = live && ((cls >> 24) & 1u) != 0u;
        const bool big = live && !small;
        // ---- small triangles by (triangle, row of its box), the items dealt over waves and lanes
        {
            const int rowsMine = small ? bh : 0;
            const int incl = waveInclusiveSum(rowsMine);
            const int total = __builtin_amdgcn_readlane(incl, kWave - 1);
            const int packedBox = ix0 | (bw << 16);
            for (int item0 = wave * kWave; item0 < total; item0 += kWaves * kWave) {
                const int j = item0 + lane;
                const bool act = j < total;
                int t = 0;
#pragma unroll
                for (int step = kWave / 2; step >= 1; step >>= 1) {
                    const int probe = __builtin_amdgcn_ds_bpermute((t + step - 1) << 2, incl);
                    t += probe <= j ? step : 0;
                }
                t = act ? t : 0;
                const int src = t << 2;
                const int inclT = __builtin_amdgcn_ds_bpermute(src, incl);
                const int rowsT = __builtin_amdgcn_ds_bpermute(src, rowsMine);
                const int boxT = __builtin_amdgcn_ds_bpermute(src, packedBox);
                const int y0T = __builtin_amdgcn_ds_bpermute(src, iy0);
                const uint32_t lowT = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)lowKey);
#define MRX_GATHER(v) __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(v)))
                const f32x2 A01 = { MRX_GATHER(c.A0), MRX_GATHER(c.A1) }, A2D = { MRX_GATHER(c.A2), MRX_GATHER(c.Dx) };
                const f32x2 B01 = { MRX_GATHER(c.B0), MRX_GATHER(c.B1) }, B2D = { MRX_GATHER(c.B2), MRX_GATHER(c.Dy) };
                const f32x2 C01 = { MRX_GATHER(c.C0), MRX_GATHER(c.C1) }, C2D = { MRX_GATHER(c.C2), MRX_GATHER(c.Dc) };
#undef MRX_GATHER
                const int xBeg = boxT & 0xFFFF, xEnd = xBeg + (boxT >> 16);
                const int sy = y0T + (j - (inclT - rowsT));
                const float py = (float)sy;
                const f32x2 yy = { py, py };
                const f32x2 r01 = fma2(B01, yy, C01);
                const f32x2 r2d = fma2(B2D, yy, C2D);
                unsigned long long *zline = zb + (sy - (int)tileY0) * kFlatZS - (int)tileX0;
                for (int sx = xBeg; __ballot(act && sx < xEnd) != 0; sx += 4) {
                    if (act && sx < xEnd) {
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const float px = (float)(sx + q4);
                            const f32x2 pp = { px, px };
                            const f32x2 e01 = fma2(A01, pp, r01);
                            const f32x2 e2d = fma2(A2D, pp, r2d);
                            if (sx + q4 < xEnd && fminf(fminf(e01.x, e01.y), e2d.x) >= 0.0f &&
                                e2d.y > invFar && e2d.y <= invNear)
                                atomicMax(zline + sx + q4, packHit(e2d.y, lowT));
                        }
                    }
                }
            }
        }
        MRX_STAMP(2);
        // ---- large triangles: this wave's strip, each 32x8 half after the exact test of the
        //      triangle's planes at the half's most favourable corner pixel (lane = triangle)
        {
            const int bx0 = ix0 - (int)tileX0, bx1 = bx0 + bw - 1;
            const int by0 = iy0 - (int)tileY0, by1 = by0 + bh - 1;
            const bool rows = big && by0 <= 8 * wave + 7 && by1 >= 8 * wave;
            const float y0 = TY0 + (float)(8 * wave), y1 = y0 + 7.0f;
            const float r0 = __builtin_fmaf(c.B0, c.B0 >= 0.0f ? y1 : y0, c.C0);
            const float r1 = __builtin_fmaf(c.B1, c.B1 >= 0.0f ? y1 : y0, c.C1);
            const float r2 = __builtin_fmaf(c.B2, c.B2 >= 0.0f ? y1 : y0, c.C2);
            const float dMax = __builtin_fmaf(c.Dy, c.Dy >= 0.0f ? y1 : y0, c.Dc);
            const float dMin = __builtin_fmaf(c.Dy, c.Dy >= 0.0f ? y0 : y1, c.Dc);
#pragma unroll
            for (int hf = 0; hf < kHalves; ++hf) {
                const float x0 = TX0 + (float)(32 * hf), x1 = x0 + 31.0f;
                const float e0 = __builtin_fmaf(c.A0, c.A0 >= 0.0f ? x1 : x0, r0);
                const float e1 = __builtin_fmaf(c.A1, c.A1 >= 0.0f ? x1 : x0, r1);
                const float e2 = __builtin_fmaf(c.A2, c.A2 >= 0.0f ? x1 : x0, r2);
                const float iMax = __builtin_fmaf(c.Dx, c.Dx >= 0.0f ? x1 : x0, dMax);
                const float iMin = __builtin_fmaf(c.Dx, c.Dx >= 0.0f ? x0 : x1, dMin);
                const bool reg = fminf(fminf(e0, e1), e2) >= 0.0f && iMax > invFar && iMin <= invNear;
                uint64_t act = __ballot(rows && reg && bx0 <= 32 * hf + 31 && bx1 >= 32 * hf);
                if (act == 0)
                    continue;
                const float py = (float)(tileY0 + 8u * wave + ly);
                const f32x2 yy = { py, py };
                unsigned long long *zrow = zb + (8 * wave + ly) * kFlatZS + 32 * hf + 4 * lx;
                for (; act != 0; act &= act - 1) {
                    const int l = __builtin_ctzll(act);
                    const PlanePairs q = loadPlanes(triRec, l);
                    const uint32_t lowv = ((~(uint32_t)l & kKeyMask) << kSlotBits) | ((uint32_t)l + 1u);
                    const f32x2 r01 = fma2(q.B01, yy, q.C01);
                    const f32x2 r2d = fma2(q.B2D, yy, q.C2D);
#pragma unroll
                    for (int b = 0; b < kRegionBlocks; ++b) {
                        const float px = (float)(tileX0 + 32 * hf + 4 * lx + b);
                        const f32x2 pp = { px, px };
                        const f32x2 e01 = fma2(q.A01, pp, r01);
                        const f32x2 e2d = fma2(q.A2D, pp, r2d);
                        if (fminf(fminf(e01.x, e01.y), e2d.x) >= 0.0f && e2d.y > invFar && e2d.y <= invNear)
                            atomicMax(zrow + b, packHit(e2d.y, lowv));
                    }
                }
            }
        }
        MRX_STAMP(3);
        __syncthreads();                              // every walk into this tile's buffer is done
        MRX_STAMP(4);
        {
            KernargParams pk = (KernargParams)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(pk));
            const ResolveArgs ra = { pk->rgb, pk->depth, pk->ids, pk->texels, pk->nfast, pk->nslow, pk->writeThrough };
            resolveStrip<IDS, TEX, TW, TH, true, kFlatZS>(ra, zb, shadeTab, coldTab, view, tileX0, tileY0, wave, lane);
        }
        MRX_STAMP(5);
        if (--left == 0)
            break;
        // this wave's strip of the buffer, for the tile after the next (behind the next tile's barrier)
        for (int i = lane; i < kFlatZS * 8; i += kWave)
            zb[8 * wave * kFlatZS + i] = packHit(invFar, 0u);
        if (kFlatZBufs == 2)
            buf ^= 1u;
        else
            __syncthreads();                          // (one buffer: every strip clear before anyone merges into it)
        ++tile;
        ++g;
    }
    MRX_STAMP(6);
#undef MRX_STAMP
}

}  // namespace

namespace {
constexpr int kMaxDevices = 64;
std::mutex attrMutex;
// LDS bytes of one workgroup for a tile shape
size_t ldsFor(uint32_t passInst, bool textured, int tw, int th, bool cls, uint32_t tlasBlocks, uint32_t texCap)
{
    const size_t cap = textured ? (size_t)texCap : (size_t)tabCap(false, tw, th, cls);
    return (size_t)(tw + kZPad) * th * 8 + tabBytes(textured, (int)cap) + 64 + (size_t)bigCap(tw, th, cls, textured) * 64 +
           ((size_t)passInst * (kInstRecDw + 4) * 4 + 16) * tlasBlocks + sizeof(WaveScratch) * (size_t)(th / 8);
}
}  // namespace

size_t bvhLdsBytes(uint32_t passInst, bool textured, bool classify, uint32_t groupViews, uint32_t texCap)
{
    return ldsFor(passInst, textured, 64, 64, classify, groupViews, texCap);
}

hipError_t launchBvh(const RasterParams &p, hipStream_t stream)
{
    if (p.numViews == 0)
        return hipSuccess;
    const int ids = p.ids == nullptr ? 0 : p.idsAreSegmask ? 2 : 1;
    const bool tex = p.anyTextured != 0;
    if (p.bvhFlat) {
        // worlds of at most 64 triangles in at most 64 rows: one set-up per view (bvhFlatKernel)
        const uint32_t tpv = ((p.nfast + 63u) / 64u) * ((p.nslow + 63u) / 64u);
        const uint32_t gt = std::max<uint32_t>(1u, std::min<uint32_t>(p.bvhGroupTiles, tpv));
        const dim3 grid(p.numViews * ((tpv + gt - 1) / gt)), block(kWave * 8);
        int dev = 0;
        const hipError_t ge = hipGetDevice(&dev);
        if (ge != hipSuccess)
            return ge;
        if (dev < 0 || dev >= kMaxDevices)
            return hipErrorInvalidDevice;
#define MRX_FLAT(I, T)                                                                          \
    do {                                                                                       \
        static bool allowed[kMaxDevices] = {};                                                 \
        {                                                                                      \
            std::lock_guard<std::mutex> guard(attrMutex);                                      \
            if (!allowed[dev]) {                                                               \
                const hipError_t e = hipFuncSetAttribute((const void *)bvhFlatKernel<I, T>,   \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)flatLdsBytes(T)); \
                if (e != hipSuccess)                                                           \
                    return e;                                                                  \
                allowed[dev] = true;                                                           \
            }                                                                                  \
        }                                                                                      \
        bvhFlatKernel<I, T><<<grid, block, flatLdsBytes(T), stream>>>(p);                      \
    } while (0)
        if (ids == 2) { if (tex) MRX_FLAT(2, true); else MRX_FLAT(2, false); }
        else if (ids == 1) { if (tex) MRX_FLAT(1, true); else MRX_FLAT(1, false); }
        else { if (tex) MRX_FLAT(0, true); else MRX_FLAT(0, false); }
#undef MRX_FLAT
        return hipGetLastError();
    }
    // tile shape: p.bvhTile = 0 (64x64), 1 (64x32: TW 64, TH 32), 2 (32x32)
    const int tw = p.bvhTile == 2 ? 32 : 64, th = p.bvhTile == 0 ? 64 : 32;
    const uint32_t tilesPerView = ((p.nfast + tw - 1) / tw) * ((p.nslow + th - 1) / th);
    const uint32_t groupTiles = std::max<uint32_t>(1u, std::min<uint32_t>(p.bvhGroupTiles, tilesPerView));
    // groups of views (MULTI): one-tile views, 64x64 tiles, every world in one TLAS pass (the host's
    // business), a power of two, at most one view per wave
    const uint32_t groupViews = p.bvhGroupViews & 0xFFFFu;
    const bool multi = groupViews > 1;
    if (groupViews == 0 || (groupViews & (groupViews - 1)) != 0 || groupViews > (uint32_t)(th / 8) ||
        (multi && (tilesPerView != 1 || p.bvhTile != 0)))
        return hipErrorInvalidValue;
    // (bit 16 with two views per workgroup: as many workgroups as the chip holds -- twice the count in bits 20..31 --
    // the first numViews - that many with two views, the others with one; bvh.hip)
    uint32_t items = multi ? (p.numViews + groupViews - 1) / groupViews
                           : p.numViews * ((tilesPerView + groupTiles - 1) / groupTiles);
    if (groupViews == 2 && (p.bvhGroupViews & 0x10000u)) {
        const uint32_t resident = 2u * (p.bvhGroupViews >> 20);
        if (resident >= items && resident <= p.numViews)
            items = resident;
    }
    if (tex && (p.bvhTexCap < 64u || p.bvhTexCap > 1023u))
        return hipErrorInvalidValue;
    const size_t lds = ldsFor(p.bvhPassInst, tex, tw, th, p.bvhTile == 0 && p.bvhClassify, groupViews, p.bvhTexCap);
    const dim3 grid(items), block(kWave * (th / 8));
    // The kernel needs more dynamic LDS than the 64 KB a launch may ask for by default.  The
    // opt-in is a property of (function, device) -- a renderer per device in one process
    // (mrx_config.device_ids, renderer_headless --gpus N) launches the same instantiation on
    // several devices, from several host threads -- so what has been granted is remembered
    // per device, under a lock.
    int dev = 0;
    {
        const hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess)
            return e;
        if (dev < 0 || dev >= kMaxDevices)
            return hipErrorInvalidDevice;
    }
#define MRX_BVH(I, T, W, H, C, M)                                                               \
    do {                                                                                       \
        static size_t allowed[kMaxDevices] = {};                                               \
        {                                                                                      \
            std::lock_guard<std::mutex> guard(attrMutex);                                      \
            if (lds > allowed[dev]) {                                                          \
                const hipError_t e = hipFuncSetAttribute((const void *)bvhTileKernel<I, T, W, H, C, M>, \
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                if (e != hipSuccess)                                                           \
                    return e;                                                                  \
                allowed[dev] = lds;                                                            \
            }                                                                                  \
        }                                                                                      \
        bvhTileKernel<I, T, W, H, C, M><<<grid, block, lds, stream>>>(p);                      \
    } while (0)
#define MRX_BVH_SHAPE(I, T)                                                                    \
    do {                                                                                       \
        if (p.bvhTile == 0 && multi && p.bvhClassify) MRX_BVH(I, T, 64, 64, true, true);       \
        else if (p.bvhTile == 0 && multi) MRX_BVH(I, T, 64, 64, false, true);                  \
        else if (p.bvhTile == 0 && p.bvhClassify) MRX_BVH(I, T, 64, 64, true, false);          \
        else if (p.bvhTile == 0) MRX_BVH(I, T, 64, 64, false, false);                          \
        else if (p.bvhTile == 1) MRX_BVH(I, T, 64, 32, false, false);                          \
        else MRX_BVH(I, T, 32, 32, false, false);                                              \
    } while (0)
    if (ids == 2) {
        if (tex) MRX_BVH_SHAPE(2, true); else MRX_BVH_SHAPE(2, false);
    } else if (ids == 1) {
        if (tex) MRX_BVH_SHAPE(1, true); else MRX_BVH_SHAPE(1, false);
    } else {
        if (tex) MRX_BVH_SHAPE(0, true); else MRX_BVH_SHAPE(0, false);
    }
#undef MRX_BVH_SHAPE
#undef MRX_BVH
    return hipGetLastError();
}

}  // namespace mrx
