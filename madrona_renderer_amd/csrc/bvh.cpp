// Host-side BLAS builder (see bvh.hpp).
#include "bvh.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace mrx {

namespace {

struct Box {
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    float hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    void grow(const float *p)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box &b)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    double area() const
    {
        const double d[3] = { (double)hi[0] - lo[0], (double)hi[1] - lo[1], (double)hi[2] - lo[2] };
        if (d[0] < 0.0)
            return 0.0;
        return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};

struct Prim {
    uint32_t tri;     // index into the ObjTri pool
    Box box;
    float c[3];       // centroid of the box
};

// binary tree built over a permutation of the object's primitives
struct BinNode {
    Box box;
    int32_t left = -1, right = -1;    // children, -1: leaf
    uint32_t first = 0, count = 0;    // range of prims (leaves)
};

struct Builder {
    std::vector<Prim> prims;
    std::vector<BinNode> bin;
    bool median = false;

    // SAH splits can be arbitrarily lopsided (a few primitives peeled off per level on
    // adversarial meshes): beyond kSahDepth levels the subtree is built with median splits,
    // which halve the range, so the recursion depth is bounded by kSahDepth + log2(count)
    static constexpr uint32_t kSahDepth = 48;
    int32_t build(uint32_t first, uint32_t count, uint32_t level = 0)
    {
        BinNode n;
        n.first = first;
        n.count = count;
        Box cb;
        for (uint32_t i = first; i < first + count; ++i) {
            n.box.grow(prims[i].box);
            cb.grow(prims[i].c);
        }
        const int32_t me = (int32_t)bin.size();
        bin.push_back(n);
        if (count <= kBvhLeafMax)
            return me;
        // split axis / position: binned SAH over the centroid bounds
        int axis = 0;
        float ext = -1.0f;
        for (int a = 0; a < 3; ++a)
            if (cb.hi[a] - cb.lo[a] > ext) {
                ext = cb.hi[a] - cb.lo[a];
                axis = a;
            }
        uint32_t mid = first + count / 2;
        bool split = false;
        if (!median && level < kSahDepth && ext > 0.0f) {
            constexpr int kBins = 16;
            double bestCost = DBL_MAX;
            int bestAxis = -1, bestBin = -1;
            for (int a = 0; a < 3; ++a) {
                const float lo = cb.lo[a], w = cb.hi[a] - cb.lo[a];
                if (!(w > 0.0f))
                    continue;
                Box bb[kBins];
                uint32_t cnt[kBins] = {};
                for (uint32_t i = first; i < first + count; ++i) {
                    int b = (int)((prims[i].c[a] - lo) / w * kBins);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bb[b].grow(prims[i].box);
                    cnt[b]++;
                }
                double rightArea[kBins];
                uint32_t rightCnt[kBins];
                Box acc;
                uint32_t c = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bb[b]);
                    c += cnt[b];
                    rightArea[b] = acc.area();
                    rightCnt[b] = c;
                }
                Box accL;
                uint32_t cl = 0;
                for (int b = 0; b + 1 < kBins; ++b) {
                    accL.grow(bb[b]);
                    cl += cnt[b];
                    if (cl == 0 || rightCnt[b + 1] == 0)
                        continue;
                    const double cost = accL.area() * cl + rightArea[b + 1] * rightCnt[b + 1];
                    if (cost < bestCost) {
                        bestCost = cost;
                        bestAxis = a;
                        bestBin = b;
                    }
                }
            }
            if (bestAxis >= 0) {
                const float lo = cb.lo[bestAxis], w = cb.hi[bestAxis] - cb.lo[bestAxis];
                auto it = std::partition(prims.begin() + first, prims.begin() + first + count,
                                         [&](const Prim &p) {
                                             int b = (int)((p.c[bestAxis] - lo) / w * 16);
                                             b = std::min(std::max(b, 0), 15);
                                             return b <= bestBin;
                                         });
                mid = (uint32_t)(it - prims.begin());
                split = mid > first && mid < first + count;
            }
        }
        if (!split) {
            // balanced: median of the centroids along the widest axis (index order
            // when every centroid coincides)
            mid = first + count / 2;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                             [&](const Prim &a, const Prim &b) { return a.c[axis] < b.c[axis]; });
        }
        const int32_t l = build(first, mid - first, level + 1);
        const int32_t r = build(mid, first + count - mid, level + 1);
        bin[me].left = l;
        bin[me].right = r;
        return me;
    }
};

// Collapse binary node `b` into an 8-wide node; returns its index in out.nodes.
// `balanced`: open the child with the most primitives instead of the largest area -- over a
// median-split tree that gives wide nodes of near-equal subtrees, i.e. a depth of about a third
// of the binary tree's (the largest-area rule carries no such guarantee).
uint32_t collapse(const Builder &bl, int32_t b, BlasSet &out, uint32_t depth, uint32_t &maxDepth, bool balanced)
{
    maxDepth = std::max(maxDepth, depth);
    std::vector<int32_t> kids = { bl.bin[b].left, bl.bin[b].right };
    // open the inner child of largest surface area until eight children
    while (kids.size() < kBvhWidth) {
        int best = -1;
        double bestArea = -1.0;
        for (size_t i = 0; i < kids.size(); ++i) {
            const BinNode &k = bl.bin[kids[i]];
            const double key = balanced ? (double)k.count : k.box.area();
            if (k.left >= 0 && key > bestArea) {
                bestArea = key;
                best = (int)i;
            }
        }
        if (best < 0)
            break;
        const BinNode &k = bl.bin[kids[best]];
        kids[best] = k.left;
        kids.push_back(k.right);
    }
    const uint32_t me = (uint32_t)out.nodes.size();
    out.nodes.emplace_back();
    {
        BvhNode &n = out.nodes[me];
        std::memset(&n, 0, sizeof n);
        for (uint32_t c = 0; c < kBvhWidth; ++c)
            n.child[c] = kBvhEmpty;
    }
    for (size_t c = 0; c < kids.size(); ++c) {
        const BinNode &k = bl.bin[kids[c]];
        uint32_t ref;
        if (k.left < 0) {
            const uint32_t start = (uint32_t)out.leafTris.size();
            for (uint32_t i = 0; i < k.count; ++i)
                out.leafTris.push_back(bl.prims[k.first + i].tri);
            ref = kBvhLeafBit | ((k.count - 1) << kBvhLeafStartBits) | start;
        } else {
            ref = collapse(bl, kids[c], out, depth + 1, maxDepth, balanced);
        }
        BvhNode &n = out.nodes[me];      // (re-fetch: the vector may have grown)
        std::memcpy(n.bmin[c], k.box.lo, 12);
        std::memcpy(n.bmax[c], k.box.hi, 12);
        n.child[c] = ref;
    }
    return me;
}

}  // namespace

void buildBlas(const ObjTri *tris, const std::vector<int32_t> &objFirst,
               const std::vector<int32_t> &objCount, BlasSet &out)
{
    out.nodes.clear();
    out.leafTris.clear();
    out.objects.clear();
    out.maxDepth = 0;
    for (size_t o = 0; o < objFirst.size(); ++o) {
        const uint32_t first = (uint32_t)objFirst[o], count = (uint32_t)objCount[o];
        ObjInfo info {};
        info.firstTri = first;
        info.numTris = count;
        info.root = -1;
        Box all;
        Builder bl;
        bl.prims.resize(count);
        for (uint32_t t = 0; t < count; ++t) {
            Prim &p = bl.prims[t];
            p.tri = first + t;
            for (int c = 0; c < 3; ++c)
                p.box.grow(tris[first + t].p + 3 * c);
            for (int a = 0; a < 3; ++a)
                p.c[a] = 0.5f * p.box.lo[a] + 0.5f * p.box.hi[a];
            all.grow(p.box);
        }
        for (int a = 0; a < 3; ++a) {
            info.bbMin[a] = count ? all.lo[a] : 0.0f;
            info.bbMax[a] = count ? all.hi[a] : 0.0f;
        }
        if (count > kBvhFlatMax) {
            for (int attempt = 0; attempt < 2; ++attempt) {
                bl.bin.clear();
                bl.median = attempt == 1;
                const int32_t rootBin = bl.build(0, count);
                const size_t nodes0 = out.nodes.size(), leaves0 = out.leafTris.size();
                uint32_t depth = 0;
                const uint32_t root = collapse(bl, rootBin, out, 1, depth, attempt == 1);
                // a pop pushes at most eight entries: 1 + 7 per level bounds the stack.  The
                // balanced rebuild is accepted as it comes: buildScene (mrx_api.cpp) refuses a
                // scene whose deepest BLAS still exceeds the bound
                if (1 + 7 * depth <= kBvhStackCap || attempt == 1) {
                    info.root = (int32_t)root;
                    out.maxDepth = std::max(out.maxDepth, depth);
                    break;
                }
                out.nodes.resize(nodes0);
                out.leafTris.resize(leaves0);
            }
        }
        out.objects.push_back(info);
    }
}

}  // namespace mrx
