// hrt_treelets.hpp -- treelets of big triangle-mesh BLASes: what the LDS-staged walker (hrt_walker_tl.hpp) walks.
//
// In walk order a subtree is an index range of the node array and (for the reference's builder, which appends a subtree's
// leaves together, Scene.cs:439-440) a range of the leaf-slot records.  A TREELET is a maximal subtree whose nodes and
// triangle records fit a workgroup's LDS budget: two contiguous copies stage it.  Everything above the treelets is copied
// into a small REDUCED tree in which a treelet's root is a PORTAL: a node with the root's box whose hit suspends the ray
// into the treelet's queue instead of descending.  A walk over reduced tree + treelets visits the nodes of the uploaded
// BLAS in the uploaded order with the same box tests (the portal's test IS the root's test), so results and tie-breaks
// are the reference's (SceneDeviceViews.cs:173-237, :270-327); only where a node is read from changes.
//
// Reduced-tree records are NodeQ with explicit links into the reduced array (entries are ordered by subtree size, largest
// first, so that a prefix of the array is the top of every tree and can live in LDS):
//   inner   lo.w = left child (reduced index)         hi.w = skip (reduced index or kEnd)
//   leaf    lo.w = first leaf slot (global)            hi.w = skip | count << 28     (small subtrees that are no treelet stay in place)
//   portal  lo.w = kPortalBit | treelet id             hi.w = skip                   (count 0)
#pragma once
#include "hrt_trace_packed.hpp"
#include <algorithm>
#include <cstring>
#include <vector>

namespace hrt {

constexpr int kPortalBit = 0x40000000;
#ifndef HRT_TL_BYTES
#define HRT_TL_BYTES 55296                       // LDS budget of one treelet: 54 KiB (config 4: 511 nodes + 784 triangle records = 53 984 B)
#endif
constexpr int kTlBytes = HRT_TL_BYTES;
constexpr int kTlMinNodes = 31;                  // smaller maximal subtrees stay in the reduced tree (walked in place)
constexpr int kTlMinBlasNodes = 4096;            // BLASes below this are walked as before
struct TreeletLimits { int bytes = kTlBytes, minNodes = kTlMinNodes, minBlasNodes = kTlMinBlasNodes; };      // the shipped values; tests lower them to reach the code with small scenes
constexpr int kTlRedLdsMax = 256;                // reduced-tree records a workgroup keeps in LDS (8 KiB)
constexpr int kTlHistLds = 2048;                 // treelet queues counted in LDS (more: global atomics)

struct Treelet { int nodeLo, nodeHi, triLo, triHi, exitRed, pad0, pad1, pad2; };     // [nodeLo, nodeHi) of DPacked::blas, [triLo, triHi) of DPacked::ftri; exitRed: reduced index after the root (its skip)

struct DTreelets {
    const NodeQ* red;            // reduced trees of every treelet-enabled BLAS, one array
    const Treelet* tl;
    const int* redOfRoot;        // [n blas nodes] reduced index of the root of the BLAS that starts at this node, -1: BLAS without treelets
    int nTl, nRed, redLds, tlBytesMax;
};

struct TreeletsHost {
    std::vector<NodeQ> red;
    std::vector<Treelet> tl;
    std::vector<int32_t> redOfRoot;
    std::vector<int32_t> redOrig;      // packed index of the node a reduced record copies (tests)
    int tlBytesMax = 0;
    bool any() const { return !tl.empty(); }
};

// blas: packed nodes, every range [lo, hi) in `ranges` in walk order (root first, left child = index + 1, subtree of i = [i, subend[i])).
inline void build_treelets(const std::vector<NodeQ>& blas, const std::vector<int32_t>& subend, const std::vector<std::pair<int64_t, int64_t>>& ranges, const TreeletLimits& lim, TreeletsHost& out)
{
    out = TreeletsHost{};
    out.redOfRoot.assign(blas.size(), -1);
    auto cnt_of = [&](int64_t i) { return (int)((unsigned)__builtin_bit_cast(int, blas[(size_t)i].hi.w) >> 28); };
    auto left_of = [&](int64_t i) { return (int64_t)(__builtin_bit_cast(int, blas[(size_t)i].lo.w) & kEnd); };
    auto first_of = [&](int64_t i) { return (int64_t)__builtin_bit_cast(int, blas[(size_t)i].lo.w); };
    struct Rec { NodeQ q; int64_t orig; int64_t bytes; int64_t skipOrig; bool portal; };
    std::vector<Rec> recs;                     // reduced records of all meshes, walk order per mesh
    std::vector<std::pair<size_t, int64_t>> roots;      // (index into recs, blas root)
    for (const auto& r : ranges)
    {
        const int64_t lo = r.first, hi = r.second, n = hi - lo;
        if (n < lim.minBlasNodes) continue;
        bool ok = true;
        std::vector<int32_t> ntri((size_t)n, 0); std::vector<int64_t> tmin((size_t)n, 0), tmax((size_t)n, 0);
        for (int64_t i = hi - 1; i >= lo && ok; i--)
        {
            const size_t k = (size_t)(i - lo);
            const int64_t e = subend[(size_t)i];
            if (e <= i || e > hi) { ok = false; break; }
            if (cnt_of(i) > 0) { ntri[k] = cnt_of(i); tmin[k] = first_of(i); tmax[k] = first_of(i) + cnt_of(i); if (e != i + 1) ok = false; continue; }
            if (left_of(i) != i + 1 || e == i + 1) { ok = false; break; }                  // walk order: the hit child is the next record; an inner node has children
            int64_t c = i + 1; int32_t s = 0; int64_t mn = INT64_MAX, mx = -1;
            while (c < e) { const size_t kc = (size_t)(c - lo); s += ntri[kc]; mn = std::min(mn, tmin[kc]); mx = std::max(mx, tmax[kc]); c = subend[(size_t)c]; }
            if (c != e) { ok = false; break; }
            ntri[k] = s; tmin[k] = mn; tmax[k] = mx;
        }
        if (!ok) continue;
        auto bytes_of = [&](int64_t i) { return (int64_t)(subend[(size_t)i] - i) * (int64_t)sizeof(NodeQ) + (int64_t)ntri[(size_t)(i - lo)] * (int64_t)sizeof(FTri); };
        auto fits = [&](int64_t i) {
            const size_t k = (size_t)(i - lo);
            return cnt_of(i) == 0 && bytes_of(i) <= lim.bytes && subend[(size_t)i] - i >= lim.minNodes && tmax[k] - tmin[k] == ntri[k];
        };
        if (fits(lo)) continue;                                    // the whole BLAS is one treelet: nothing to queue for
        const size_t recs0 = recs.size(), tl0 = out.tl.size();
        std::vector<int64_t> rid((size_t)n, -1);
        for (int64_t i = lo; i < hi;)
        {
            Rec rc; rc.q = blas[(size_t)i]; rc.orig = i; rc.bytes = bytes_of(i); rc.portal = false;
            const int sk = __builtin_bit_cast(int, blas[(size_t)i].hi.w) & kEnd;
            rc.skipOrig = sk == kEnd ? -1 : sk;
            rid[(size_t)(i - lo)] = (int64_t)recs.size();
            if (fits(i))
            {
                Treelet t{}; t.nodeLo = (int)i; t.nodeHi = subend[(size_t)i]; t.triLo = (int)tmin[(size_t)(i - lo)]; t.triHi = (int)tmax[(size_t)(i - lo)]; t.exitRed = kEnd;
                rc.portal = true;
                rc.q.lo.w = __builtin_bit_cast(float, (int)(kPortalBit | (int)out.tl.size()));
                out.tl.push_back(t);
                out.tlBytesMax = std::max<int>(out.tlBytesMax, (int)bytes_of(i));
                recs.push_back(rc);
                i = subend[(size_t)i];
            }
            else { recs.push_back(rc); i++; }
        }
        if (out.tl.size() - tl0 < 2) { recs.resize(recs0); out.tl.resize(tl0); continue; }
        // links inside this mesh, still in recs numbering (remapped to the final order below)
        for (size_t k = recs0; k < recs.size(); k++)
        {
            Rec& rc = recs[k];
            const int64_t sk = rc.skipOrig < 0 ? -1 : rid[(size_t)(rc.skipOrig - lo)];
            const int cnt = rc.portal ? 0 : cnt_of(rc.orig);
            rc.skipOrig = sk;                                       // now: recs index or -1
            if (!rc.portal && cnt == 0) rc.q.lo.w = __builtin_bit_cast(float, (int)rid[(size_t)(rc.orig + 1 - lo)]);     // left child (recs index)
            rc.q.hi.w = __builtin_bit_cast(float, (int)((unsigned)cnt << 28));
        }
        roots.emplace_back(recs0, lo);
    }
    if (out.tl.empty()) { out.redOfRoot.clear(); return; }
    // final order: largest subtrees first (ties: walk order), so a prefix of the array is the top of every tree
    std::vector<size_t> order(recs.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return recs[a].bytes > recs[b].bytes; });
    std::vector<int32_t> pos(recs.size());
    for (size_t k = 0; k < order.size(); k++) pos[order[k]] = (int32_t)k;
    out.red.resize(recs.size());
    out.redOrig.resize(recs.size());
    for (size_t k = 0; k < recs.size(); k++)
    {
        const Rec& rc = recs[k];
        NodeQ q = rc.q;
        const int cntBits = __builtin_bit_cast(int, q.hi.w);
        const int sk = rc.skipOrig < 0 ? kEnd : pos[(size_t)rc.skipOrig];
        q.hi.w = __builtin_bit_cast(float, sk | cntBits);
        if (rc.portal) out.tl[(size_t)(__builtin_bit_cast(int, q.lo.w) & kEnd)].exitRed = sk;
        else if (((unsigned)cntBits >> 28) == 0) q.lo.w = __builtin_bit_cast(float, (int)pos[(size_t)__builtin_bit_cast(int, q.lo.w)]);
        out.red[(size_t)pos[k]] = q;
        out.redOrig[(size_t)pos[k]] = (int32_t)rc.orig;
    }
    for (const auto& rt : roots) out.redOfRoot[(size_t)rt.second] = pos[rt.first];
}

} // namespace hrt
