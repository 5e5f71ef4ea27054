// hrt_walker_wide.hpp -- the persistent-wave walker over the 4-wide collapse of the trees (WNode, hrt_trace_packed.hpp).
//
// The binary walk is a chain of dependent fetches: ~30-100 node records per ray, one memory round trip each (two with the
// lookahead), and a wave spends two thirds of its time waiting for them (profiles/r01_pmc_config4_chained_walker.txt).
// Inner nodes only accelerate (a leaf's own exact box test, taken with the closest t of that moment,
// decides its entry; leaves must be met in walk order), so the tree above the leaves may be regrouped: a WNode holds the
// exact boxes of up to four grandchildren in walk order in one 128-byte line.  One fetch tests four boxes and the ray
// descends two binary levels: 3-4x fewer dependent rounds for about the same bytes and box tests.
//
// Each lane keeps a stack of pending child references (4 bytes each): the first kWStackLds levels in LDS ([level][lane],
// conflict-free), deeper ones in a global overflow area; hrt_scene_upload only enables the walker when 3 x (wide depth) + 2
// fits both.  A step pops one reference: a WNode (test 4 boxes, push the hit children so that the first in walk order is
// popped first) or a leaf record of the binary array (exact test with the current closest t, then the same leaf code as the
// binary walker).  Rays whose origin or 1/d is not finite -- where the box test is not monotone -- are walked by the
// per-lane tree walk of TracerPackedT instead, world rays at fetch and object-space rays at instance entry.
#pragma once
#include "hrt_walker.hpp"

namespace hrt {

enum { W_IDLE = 0, W_WIDE = 1, W_TLEAF = 2, W_BLEAF = 3, W_DONE = 4 };
#ifndef HRT_WSTACK_LDS
#define HRT_WSTACK_LDS 12
#endif
constexpr int kWStackLds = HRT_WSTACK_LDS;      // stack levels per lane kept in LDS (1 KiB each per 256-thread workgroup)
constexpr int kWStackOvf = 52;      // further levels per lane in global memory
#ifndef HRT_WIDE_LEAFLOOP
#define HRT_WIDE_LEAFLOOP 1
#endif
#ifndef HRT_WIDE_BURST
#define HRT_WIDE_BURST 4
#endif
constexpr int kWideBurst = HRT_WIDE_BURST;
constexpr int kWideMaxDepth = (kWStackLds + kWStackOvf - 2) / 3;


HRT_D bool hit_box6(const Ray& r, float lx, float ly, float lz, float hx, float hy, float hz, float tMin, float tMax)   // hit_box on split coordinates
{
    float4 lo, hi; lo.x = lx; lo.y = ly; lo.z = lz; lo.w = 0.f; hi.x = hx; hi.y = hy; hi.z = hz; hi.w = 0.f;
    return hit_box(r, lo, hi, tMin, tMax);
}

template <int FEAT, bool ANY, class NextSeg, class Fetch, class Done>
HRT_D void walk_queue_wide(const TracerPackedT<FEAT>& tr, int* ovf, NextSeg nextSeg, Fetch fetch, Done done)
{
    constexpr bool kGeneral = (FEAT & 1) != 0;
    constexpr bool kAlpha = (FEAT & 2) != 0;
    __shared__ float park_mem[kGeneral ? 9 : 1][256];
    __shared__ int wst[kWStackLds][256];
    RayPark park; park.sh = park_mem;
    const DPacked& P = tr.P;
    const DScene& S = tr.S;
    Tex tex(S);
    Cnt<false> C;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int* myOvf = ovf + ((size_t)blockIdx.x * 256 + tid) * kWStackOvf;

    int sp = 0, wbase = 0;           // per lane: pending references [wbase, sp) belong to the tree being walked
    auto push = [&](int v) { if (sp < kWStackLds) wst[sp][tid] = v; else myOvf[sp - kWStackLds] = v; sp++; };
    auto pop = [&]() -> int { sp--; return sp < kWStackLds ? wst[sp][tid] : myOvf[sp - kWStackLds]; };

    int segBase = 0, segN = 0, segCur = 0;
    bool more = true;
    int mode = W_IDLE, rayIdx = -1;
    Ray w; w.o = w.d = w.inv = mk3(0.f, 0.f, 0.f);
    float tMaxW = 0.f;
    float bestT = 1e30f, bestTObj = 0.f; int bestSlot = -1, bestPrim = -1;
    bool occl = false;
    bool inBlas = false;
    int li = 0, lend = 0, bj = 0, bend = 0;
    int iflags = 0, islot = 0; float iscale = 1.f, tObj = 1e30f; int iprim = -1;

    for (;;)
    {
        // ---------------- refill idle lanes from the chain
        {
            unsigned long long idle = __ballot(mode == W_IDLE);
            int nIdle = __popcll(idle);
            if (more && (nIdle >= kRefillMin || nIdle == 64))
            {
                while (nIdle > 0)
                {
                    if (segCur >= segN)
                    {
                        more = nextSeg(segBase, segN);
                        segCur = 0;
                        if (!more) { segN = 0; break; }
                        continue;
                    }
                    const int avail = segN - segCur;
                    const int rank = __popcll(idle & lt);
                    if (mode == W_IDLE && rank < avail)
                    {
                        rayIdx = segBase + segCur + rank;
                        bestT = 1e30f; bestTObj = 0.f; bestSlot = -1; bestPrim = -1; occl = false;
                        inBlas = false; sp = 0; wbase = 0;
                        if (!fetch(rayIdx, w, tMaxW)) mode = W_DONE;                  // queue entry without a ray
                        else if (!finite_ray(w))
                        {   // the regrouped tree is only equivalent under finite slab arithmetic: per-lane tree walk
                            if (ANY) occl = tr.template occluded_ext<false, true>(w, tMaxW, C, park_mem);
                            else tr.template closest_raw<false, true>(w, bestT, bestTObj, bestSlot, bestPrim, C, park_mem);
                            mode = W_DONE;
                        }
                        else { push(P.wideTlasRoot); mode = W_WIDE; }
                    }
                    segCur += nIdle < avail ? nIdle : avail;
                    idle = __ballot(mode == W_IDLE);
                    nIdle = __popcll(idle);
                }
            }
            if (!more && __popcll(__ballot(mode == W_IDLE)) == 64) break;
        }

        // ---------------- pops: a wide node (4 box tests) or a leaf record (exact test at entry); several per iteration
        // while most lanes are still descending
        for (int burst = 0; burst < kWideBurst; burst++)
        {
        const int nWide = __popcll(__ballot(mode == W_WIDE));
        if (nWide == 0 || (burst > 0 && nWide < 24)) break;
        if (mode == W_WIDE)
        {
            if (sp == wbase)
            {   // nothing pending in this tree
                if (kGeneral && inBlas)
                {   // BLAS exhausted: fold the instance result into the world result (:65-77), back to the TLAS leaf
                    if (!ANY && tObj < 1e29f)
                    {
                        float tWorld = tObj / iscale;
                        if (tWorld < bestT) { bestT = tWorld; bestTObj = tObj; bestSlot = islot; bestPrim = iprim; }
                    }
                    w = park.get();
                    inBlas = false; wbase = 0;
                    mode = (li == lend) ? W_WIDE : W_TLEAF;
                }
                else mode = W_DONE;
            }
            else
            {
                const int ref = pop();
                const bool top = !kGeneral || !inBlas;
                const float lim = top ? (ANY ? tMaxW : bestT) : (ANY ? tMaxW * iscale : tObj);
                if (ref < 0)
                {
                    if (ref != kWNone)
                    {
                        const NodeQ nd = (top ? P.tlas : P.blas)[~ref];
                        if (hit_box(w, nd.lo, nd.hi, 0.001f, lim))
                        {
                            const int cnt = (int)((unsigned)wbits(nd.hi) >> 28);
                            if (top) { li = wbits(nd.lo); lend = li + cnt; mode = W_TLEAF; }
                            else     { bj = wbits(nd.lo); bend = bj + cnt; mode = W_BLEAF; }
                        }
                    }
                }
                else
                {
                    const WNode wn = P.wide[ref];
                    // children 3..0: the first in walk order is pushed last, i.e. popped first
                    if (wn.ref.w != kWNone && hit_box6(w, wn.lox.w, wn.loy.w, wn.loz.w, wn.hix.w, wn.hiy.w, wn.hiz.w, 0.001f, lim)) push(wn.ref.w);
                    if (wn.ref.z != kWNone && hit_box6(w, wn.lox.z, wn.loy.z, wn.loz.z, wn.hix.z, wn.hiy.z, wn.hiz.z, 0.001f, lim)) push(wn.ref.z);
                    if (wn.ref.y != kWNone && hit_box6(w, wn.lox.y, wn.loy.y, wn.loz.y, wn.hix.y, wn.hiy.y, wn.hiz.y, 0.001f, lim)) push(wn.ref.y);
                    if (wn.ref.x != kWNone && hit_box6(w, wn.lox.x, wn.loy.x, wn.loz.x, wn.hix.x, wn.hiy.x, wn.hiz.x, 0.001f, lim)) push(wn.ref.x);
                }
            }
        }
        }

        // ---------------- one TLAS leaf entry
        if (mode == W_TLEAF)
        {
            FInst f = P.finst[li];
            const int flags = wbits(f.a);
            if (!kGeneral || (flags & FI_FAST_SPHERE))
            {
                const float lim = ANY ? tMaxW : 1e30f;
                if (hit_box(w, f.a, f.b, 0.001f, lim))
                {
                    float t;
                    if (hit_sphere_t(w, xyz(f.c), f.c.w, t) && t > 0.001f && t < lim)
                    {
                        if (ANY) { occl = true; mode = W_DONE; }
                        else if (t < 1e29f && t < bestT) { bestT = t; bestTObj = t; bestSlot = li; bestPrim = wbits(f.b); }
                    }
                }
                li++;
            }
            else
            {   // general instance: park the world ray, walk its BLAS with the object-space ray
                islot = li; iflags = flags; iscale = f.c.z;
                const int blasStart = __float_as_int(f.c.x), blasEnd = __float_as_int(f.c.y);
                tObj = 1e30f; iprim = -1;
                const Ray iray = tr.object_ray(w, flags, wbits(f.b));
                li++;
                if (blasStart < blasEnd)
                {
                    if (!finite_ray(iray))
                    {   // per-lane BLAS walk (see header): result folded at once, the world ray never left its registers
                        if (ANY)
                        {
                            const bool blocked = (flags & FI_SPHERESET) ? tr.template blas_spheres_any<false>(iray, blasStart, blasEnd, tMaxW * iscale, C)
                                                                        : tr.template blas_tris_any<false>(iray, blasStart, blasEnd, tMaxW * iscale, C);
                            if (blocked) { occl = true; mode = W_DONE; }
                        }
                        else
                        {
                            if (flags & FI_SPHERESET) tr.template blas_spheres_closest<false>(iray, blasStart, blasEnd, tObj, iprim, C);
                            else                      tr.template blas_tris_closest<false>(iray, blasStart, blasEnd, tObj, iprim, C);
                            if (tObj < 1e29f)
                            {
                                float tWorld = tObj / iscale;
                                if (tWorld < bestT) { bestT = tWorld; bestTObj = tObj; bestSlot = islot; bestPrim = iprim; }
                            }
                        }
                    }
                    else
                    {
                        park.put(w);
                        w = iray;
                        inBlas = true; wbase = sp;
                        push(__float_as_int(f.c.w));                   // wide root of this BLAS
                        mode = W_WIDE;
                    }
                }
            }
            if (mode == W_TLEAF && li == lend) mode = W_WIDE;
        }

        // ---------------- one BLAS leaf entry
#if HRT_WIDE_LEAFLOOP
        while (kGeneral && mode == W_BLEAF)          // the whole leaf (<= 4 primitives) in one go: the lane is back among the node steps next iteration
#else
        if (kGeneral && mode == W_BLEAF)
#endif
        {
            const float lim = ANY ? tMaxW * iscale : tObj;
            if (iflags & FI_SPHERESET)
            {
                int p = S.spherePrimIdx[bj];
                const hrt_sphere* sp1 = &S.spheres[p];
                float t;
                if (hit_sphere_t(w, cv3(sp1->center), sp1->radius, t) && t > 0.001f && t < lim)
                {
                    if (ANY) { occl = true; mode = W_DONE; }
                    else { tObj = t; iprim = p; }
                }
            }
            else
            {
                FTri trr = P.ftri[bj];
                float t, bu, bv;
                if (hit_tri_t(w, xyz(trr.v0), xyz(trr.v1), xyz(trr.v2), t, bu, bv))
                {
                    if (!ANY)
                    {   // TraverseBLAS_Tri_Textured :196-227
                        if (t > 0.001f && t < tObj)
                        {
                            bool accept = true;
                            if (kAlpha && (wbits(trr.v2) & FT_TEXTURED))
                            {
                                const hrt_material* mat = &S.materials[wbits(trr.v1)];
                                int ati = mat->AlphaTexIndex;
                                float alpha = 1.f;
                                if (mat->HasAlphaMap != 0 && ati >= 0 && ati < S.n_texInfos)
                                {
                                    float uu, vv;
                                    tr.tri_uv(wbits(trr.v0), bu, bv, uu, vv);
                                    alpha = tex.mask_linear(S.texInfos[ati], uu, vv);
                                }
                                accept = !(alpha < mat->AlphaCutoff);
                            }
                            if (accept) { tObj = t; iprim = bj; }
                        }
                    }
                    else if (!(t <= 0.001f || t >= lim))
                    {   // AnyHit_Tri_Textured :292-317
                        bool blocked = true;
                        if (kAlpha && (wbits(trr.v2) & FT_TEXTURED))
                        {
                            const hrt_material* mat = &S.materials[wbits(trr.v1)];
                            int ati = mat->AlphaTexIndex;
                            if (mat->HasAlphaMap != 0 && ati >= 0 && ati < S.n_texInfos)
                            {
                                float uu, vv;
                                tr.tri_uv(wbits(trr.v0), bu, bv, uu, vv);
                                hrt_tex_info ainfo = S.texInfos[ati];
                                float aPoint = tex.mask_point(ainfo, uu, vv);
                                float cutoff = mat->AlphaCutoff;
                                if (aPoint < cutoff - 0.10f) blocked = false;
                                else if (aPoint >= cutoff + 0.10f) blocked = true;
                                else blocked = !(tex.mask_linear(ainfo, uu, vv) < cutoff);
                            }
                        }
                        if (blocked) { occl = true; mode = W_DONE; }
                    }
                }
            }
            bj++;
            if (mode == W_BLEAF && bj == bend) mode = W_WIDE;
        }

        // ---------------- retire finished rays
        if (mode == W_DONE)
        {
            WalkResult r; r.t = bestT; r.tObj = bestTObj; r.slot = bestSlot; r.prim = bestPrim; r.occluded = occl;
            done(rayIdx, r);
            mode = W_IDLE;
        }
    }
}

} // namespace hrt
