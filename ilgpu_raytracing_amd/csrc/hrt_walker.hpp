// hrt_walker.hpp -- persistent-wave BVH walker with ballot/prefix refill of idle lanes.
//
// Measured on the one-ray-per-lane walk (profiles/r01_pmc_config3_*): the traversal kernels are
// VALU-bound but only ~16 % of the lanes of an executed instruction are live -- a wave runs
// until its LONGEST ray is done (sky rays leave after a few nodes, grazing rays visit
// hundreds).  Here a wave owns a queue of rays -- one path range, or a chain of ranges it pulls
// from the launch-wide hand-out (RangeGrab, hrt_wavefront.hpp) -- and keeps its 64 lanes full:
// each lane carries a resumable walk state (which tree, which node, which leaf entry), and
// whenever >= kRefillMin lanes have finished, one __ballot + popcount prefix hands each of them
// the next unfetched ray of the queue.  Every iteration all walking lanes take a node step together (TLAS and BLAS
// nodes share the code: same 32-byte NodeQ, same box test), then lanes standing on a leaf entry
// take one primitive step.  A ray's sequence of node visits, box tests and primitive tests is
// exactly the reference's (SceneDeviceViews.cs:30-327), so hits, tie-breaks and work counters
// are unchanged; only the interleaving across rays differs.
//
// The walker produces raw winners (t, leaf slot, primitive) or an occlusion bit; shading the
// winner (TracerPackedT::finish_hit) runs afterwards at full lane occupancy.
#pragma once
#include "hrt_trace_packed.hpp"

namespace hrt {

enum { M_IDLE = 0, M_TLAS = 1, M_TLEAF = 2, M_BLAS = 3, M_BLEAF = 4, M_DONE = 5 };
#ifndef HRT_REFILL_MIN
#define HRT_REFILL_MIN 24
#endif
constexpr int kRefillMin = HRT_REFILL_MIN;      // refill when at least this many lanes are idle (or none is active); 8 / 16 / 24 / 32 measured, DESIGN.md 8
constexpr int kNodeBurst = 6;       // max node steps per iteration while most lanes are still walking
#ifndef HRT_LOOKAHEAD
#define HRT_LOOKAHEAD 2             // records fetched per node step (1 = no lookahead)
#endif
constexpr int kLook = HRT_LOOKAHEAD;

struct WalkResult { float t, tObj; int slot, prim; bool occluded; };

// Tuning aid (variant build -DHRT_WALK_STATS, tools/walk_stats.py): how often each phase of the loop runs and with how
// many live lanes.  [0] iterations [1] node steps [2] lanes in node steps [3] TLEAF runs [4] lanes [5] BLEAF runs
// [6] lanes [7] retire runs [8] lanes [9] idle lanes summed over iterations [10] segments pulled [11] waves
#ifdef HRT_WALK_STATS
__device__ unsigned long long g_walk_stats[2][24];
#define WSTAT(i, v) ws[i] += (unsigned long long)(v)
#define WTIME(i) { const long long wnow_ = (long long)__builtin_readcyclecounter(); ws[i] += (unsigned long long)(wnow_ - wt_); wt_ = wnow_; }
#else
#define WSTAT(i, v)
#define WTIME(i)
#endif

// FEAT as in TracerPackedT.  ANY = shadow rays (any hit, tMax) vs closest hit.
// The queue is a chain of segments: nextSeg(base, n) (wave-uniform) hands out the next run of n entries starting at
// global index base, or returns false when the chain is exhausted.  A wave keeps pulling segments while it has idle
// lanes, so its lanes stay full until the whole chain has been handed out: a path range that holds only a few live
// rays (every range does beyond the first bounce) no longer costs a wave of its own.
// fetch(i, ray, tMax) loads entry i (false: entry carries no ray); done(i, result) consumes its result.
// EXISTS (closest-hit walks of the LAST bounce, production frames): only whether TraceClosest finds a hit is used there
// (a miss adds the sky, a hit ends the path: RTRay.cs:241-243, 298-305), so the walk stops at the first hit it accepts -- the
// same tests with the same limits in the same order up to that point, hence the same answer to "is best.t < 1e29".
// ALT (any-hit and hit-or-miss walks of scenes made of fast-sphere instances, production frames): `tr` walks ANOTHER tree over the
// same instances (the device-built TLAS, hrt_bvh.hpp) than the one the scene was uploaded with.  Exact: such a query is an OR over
// the instances whose sphere test the walk reaches; the walk reaches an instance iff the instance's OWN box test passes (limits
// are fixed, every box above it contains it exactly, and enlarging a box can only turn a miss into a hit: DESIGN.md 4), so the
// tree above the instances decides nothing.  The argument needs finite slab arithmetic: a ray with a non-finite origin or 1/d
// walks the uploaded tree (`exact`) when it is fetched.
// ALT on a closest-hit walk that is not EXISTS (kTies): see "Closest-hit walks over the SECOND tree" in hrt_trace_packed.hpp --
// node tests take closest * kSecondLimit over inflated boxes, so the walk finds the least t over all candidates; a candidate at
// exactly the closest t so far, or a winner whose own slab entry exceeds its hit distance, sends the ray to the uploaded tree at
// retirement (none in practice).  Results carry leaf slots of `tr`'s tree (DPacked::slotMap translates the uploaded tree's), so
// the shading that follows reads `tr`'s instance records.
// LT: triangle records fetched per leaf step (2 or 3; chosen per scene at upload, DPacked::leafTris).
template <int FEAT, bool ANY, bool COUNT, bool EXISTS, bool ALT, int LT, class NextSeg, class Fetch, class Done>
HRT_D void walk_queue(const TracerPackedT<FEAT>& tr, const TracerPackedT<FEAT>& exact, NextSeg nextSeg, Fetch fetch, Done done, Cnt<COUNT>& C)
{
    static_assert(!(EXISTS && (ANY || COUNT)), "EXISTS is a closest-hit walk of a production frame");
    static_assert(!ALT || (!COUNT && FEAT == 0), "another tree only for production frames of fast-sphere instances");
    constexpr bool kTies = ALT && !ANY && !EXISTS;          // closest-hit walk over the other tree: watch for equal-t candidates
    constexpr bool kGeneral = (FEAT & 1) != 0;
    constexpr bool kAlpha = (FEAT & 2) != 0;
    // sphere-instance scenes: instance records are inlined into the node stream (DPacked::tlasX); a leaf hit just walks on
    // into them, an instance-record hit leaves ONE sphere test pending for the leaf step
    const bool inl = !kGeneral && tr.P.tlasX != nullptr;
    // the second tree in several child orders (DPacked::tlasXO): a ray walks the numbering of its direction's signs, in which the nearer
    // child of an inner node comes first -- a fixed-order walk then prunes the farther one against what it found in the nearer
    // (profiles/r02_walk_experiments.txt (14), (15): config 3's path stage -7.5 % with four numberings).  Which numbering a ray
    // walks changes no result: the walks over the second tree do not depend on the order (see above and hrt_trace_packed.hpp).
    // Only closest-hit walks that run to the end: a walk that stops at its first hit (any-hit, EXISTS) gains nothing from meeting the
    // nearer child first and would only share the L2 with three more copies.
#ifdef HRT_ORD_ALL                 // A/B
    const bool ord = ALT && inl && tr.P.tlasXO != nullptr;
#else
    const bool ord = kTies && inl && tr.P.tlasXO != nullptr;
#endif
    __shared__ float park_mem[kGeneral ? 9 : 1][256];
    RayPark park; park.sh = park_mem;
    const DPacked& P = tr.P;
    const DScene& S = tr.S;
    Tex tex(S);
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;

#ifdef HRT_WALK_STATS
    unsigned long long ws[20] = {};     // [12..16] shader-clock cycles in refill / node steps / TLEAF / BLEAF / retire
    long long wt_ = (long long)__builtin_readcyclecounter();
#endif
    int segBase = 0, segN = 0, segCur = 0;      // wave-uniform: current segment and its first unfetched entry
    bool more = true;                            // wave-uniform: the chain may still hold segments
    int mode = M_IDLE, rayIdx = -1;
    Ray w;                           // ray in use: world ray (TLAS modes) or object ray (BLAS modes; world ray parked in LDS)
    w.o = w.d = w.inv = mk3(0.f, 0.f, 0.f);
    float tMaxW = 0.f;               // ANY: world tMax
    float bestT = 1e30f, bestTObj = 0.f; int bestSlot = -1, bestPrim = -1;     // closest
    bool occl = false, tie = false, anom = false;
    int cur = 0, li = 0, lend = 0, lskip = kEnd;          // TLAS walk / leaf iteration
    int xlast = tr.P.nTlasX - 1;                           // last record of the node array this lane walks (ord: of its octant's copy)
    int bj = 0, bend = 0, bskip = kEnd;                   // BLAS leaf iteration
    int blasEnd = 0, iflags = 0, islot = 0; float iscale = 1.f, tObj = 1e30f; int iprim = -1;   // instance being walked

    for (;;)
    {
        // ---------------- refill idle lanes from the chain
        {
            unsigned long long idle = __ballot(mode == M_IDLE);
            int nIdle = __popcll(idle);
            WSTAT(0, 1); WSTAT(9, nIdle);
            if (more && (nIdle >= kRefillMin || nIdle == 64))
            {
                while (nIdle > 0)
                {
                    if (segCur >= segN)
                    {
                        more = nextSeg(segBase, segN);
                        segCur = 0;
                        WSTAT(10, 1);
                        if (!more) { segN = 0; break; }
                        continue;
                    }
                    const int avail = segN - segCur;
                    const int rank = __popcll(idle & lt);
                    if (mode == M_IDLE && rank < avail)
                    {
                        rayIdx = segBase + segCur + rank;
                        bestT = 1e30f; bestTObj = 0.f; bestSlot = -1; bestPrim = -1; occl = false; tie = false; anom = false;
                        if (fetch(rayIdx, w, tMaxW))
                        {
                            C.inc(ANY ? C_RAYS_SHADOW : C_RAYS_CLOSEST); cur = 0; mode = M_TLAS;
                            if (ord)
                            {
                                cur = ord_copy(P.xAxes, w.d.x, w.d.y, w.d.z) * P.xStride;
                                xlast = cur + P.xStride - 1;
                            }
                            // the tree-independence argument needs finite slab arithmetic: this ray walks the uploaded tree (at retirement)
                            if (ALT && !finite_ray(w)) { tie = true; mode = M_DONE; }
                        }
                        else mode = M_DONE;                   // queue entry without a ray (path already ended)
                    }
                    segCur += nIdle < avail ? nIdle : avail;
                    idle = __ballot(mode == M_IDLE);
                    nIdle = __popcll(idle);
                }
            }
            if (!more && __popcll(__ballot(mode == M_IDLE)) == 64) break;         // chain drained and every lane finished
        }

        WTIME(12);
        // ---------------- node steps: TLAS and BLAS nodes alike
        for (int burst = 0; burst < kNodeBurst; burst++)
        {
            const bool walking = (mode == M_TLAS) || (kGeneral && mode == M_BLAS);
            const int nWalk = __popcll(__ballot(walking));
            if (nWalk == 0 || (burst > 0 && nWalk < 24)) break;
            WSTAT(1, 1); WSTAT(2, nWalk);
            if (walking)
            {
                const bool top = !kGeneral || mode == M_TLAS;
                const NodeQ* nodes = top ? (inl ? (ord ? P.tlasXO : P.tlasX) : P.tlas) : P.blas;
                // In walk order the node entered after a hit on an inner node is the next record, usually in the same
                // 128-byte line: it is fetched together with the node itself, so a hit costs no second memory round trip
                // (the walk is bound by the latency of dependent loads, not by their number).
                const int last = top ? (inl ? xlast : P.nTlas - 1) : blasEnd - 1;
                // kLook consecutive records leave together (one or two 128-byte lines)
                NodeQ nds[kLook];
#pragma unroll
                for (int k = 0; k < kLook; k++) nds[k] = nodes[cur + k <= last ? cur + k : last];
                __builtin_amdgcn_sched_barrier(0);           // all loads leave before the first box test waits on one of them
                const float lim = top ? (ANY ? tMaxW : (kTies ? bestT * kSecondLimit : bestT)) : (ANY ? tMaxW * iscale : tObj);
#pragma unroll
                for (int k = 0; k < kLook; k++)
                {
                    const NodeQ nd = nds[k];
                    C.inc(C_NODE_VISITS);
                    int sk = wbits(nd.hi);
                    const int cnt = (int)((unsigned)sk >> 28);
                    sk &= kEnd;
                    const int here = cur;
                    bool stay = true;                    // still walking nodes of the same tree after this node
                    const bool isInst = inl && cnt == 15;        // the one-node BLAS of an instance (its box test takes tMax, not the closest t)
                    if (isInst) C.inc(C_LEAF_INST);
                    if (!hit_box(w, nd.lo, nd.hi, 0.001f, isInst ? (ANY ? tMaxW : 1e30f) : lim)) cur = sk;
                    else if (isInst) { stay = false; li = wbits(nd.lo); lskip = sk; mode = M_TLEAF; }
                    else if (cnt > 0)
                    {
                        if (inl) cur = here + 1;         // its instance records follow
                        else
                        {
                            stay = false;
                            if (top) { li = wbits(nd.lo); lend = li + cnt; lskip = sk; mode = M_TLEAF; }
                            else     { bj = wbits(nd.lo); bend = bj + cnt; bskip = sk; mode = M_BLEAF; }
                        }
                    }
                    else cur = wbits(nd.lo) & kEnd;
                    // the record after `here` is the next node after a hit on an inner node AND after a missed leaf (a
                    // leaf's skip link is its successor in walk order): only a missed inner node jumps elsewhere
                    if (!(stay && cur == here + 1 && here < last)) break;
                }
            }
            // walk ends
            if (kGeneral && mode == M_BLAS && !(cur < blasEnd))
            {   // BLAS exhausted: fold the instance result into the world result (:65-77), back to the TLAS leaf
                if (!ANY && tObj < 1e29f)
                {
                    float tWorld = tObj / iscale;
                    if (tWorld < bestT) { bestT = tWorld; bestTObj = tObj; bestSlot = islot; bestPrim = iprim; }
                }
                w = park.get();
                mode = M_TLEAF;
            }
            if (!inl && mode == M_TLEAF && li == lend) { cur = lskip; mode = M_TLAS; }
            if (mode == M_TLAS && cur == kEnd) mode = M_DONE;
        }

        WTIME(13);
        // ---------------- one TLAS leaf entry
#ifdef HRT_WALK_STATS
        { const int a = __popcll(__ballot(mode == M_TLEAF)); if (a) { WSTAT(3, 1); WSTAT(4, a); }
          const int b = __popcll(__ballot(kGeneral && mode == M_BLEAF)); if (b) { WSTAT(5, 1); WSTAT(6, b); } }
#endif
        if (inl && mode == M_TLEAF)
        {   // the sphere of the instance record whose box was hit
            const float4 fb = P.finst[li].b, fc = P.finst[li].c;
            const float lim = ANY ? tMaxW : 1e30f;
            C.inc(C_SPHERE_TESTS);
            float t;
            if (hit_sphere_t(w, xyz(fc), fc.w, t) && t > 0.001f && t < lim)
            {
                if (ANY) { occl = true; mode = M_DONE; }
                else if (t < 1e29f && t < bestT)
                {
                    bestT = t; bestTObj = t; bestSlot = li; bestPrim = wbits(fb); if (EXISTS) mode = M_DONE;
                    if (kTies) anom = box_entry(w, P.finst[li].a, fb) > t;
                }
                else if (kTies && t == bestT) tie = true;
            }
            if (mode == M_TLEAF) { cur = lskip; mode = (cur == kEnd) ? M_DONE : M_TLAS; }
        }
        else if (mode == M_TLEAF)
        {
            FInst f = P.finst[li];
            C.inc(C_LEAF_INST);
            const int flags = wbits(f.a);
            if (!kGeneral || (flags & FI_FAST_SPHERE))
            {
                C.inc(C_NODE_VISITS);                        // the one-node BLAS of the instance
                const float lim = ANY ? tMaxW : 1e30f;
                if (hit_box(w, f.a, f.b, 0.001f, lim))
                {
                    C.inc(C_SPHERE_TESTS);
                    float t;
                    if (hit_sphere_t(w, xyz(f.c), f.c.w, t) && t > 0.001f && t < lim)
                    {
                        if (ANY) { occl = true; mode = M_DONE; }
                        else if (t < 1e29f && t < bestT)
                        {
                            bestT = t; bestTObj = t; bestSlot = li; bestPrim = wbits(f.b); if (EXISTS) mode = M_DONE;
                            if (kTies) anom = box_entry(w, f.a, f.b) > t;
                        }
                        else if (kTies && t == bestT) tie = true;
                    }
                }
                li++;
            }
            else
            {   // general instance: park the world ray, walk its BLAS with the object-space ray
                islot = li; iflags = flags; iscale = f.c.z;
                cur = __float_as_int(f.c.x); blasEnd = __float_as_int(f.c.y);
                tObj = 1e30f; iprim = -1;
                park.put(w);
                w = tr.object_ray(w, flags, wbits(f.b));
                li++;
                mode = M_BLAS;
                if (!(cur < blasEnd)) { w = park.get(); mode = M_TLEAF; }          // empty BLAS
            }
            if (mode == M_TLEAF && li == lend) { cur = lskip; mode = (cur == kEnd) ? M_DONE : M_TLAS; }
        }

        WTIME(14);
        // ---------------- one BLAS leaf entry
        if (kGeneral && mode == M_BLEAF)
        {
            const float lim = ANY ? tMaxW * iscale : tObj;
            if (iflags & FI_SPHERESET)
            {
                int p = S.spherePrimIdx[bj];
                const hrt_sphere* sp = &S.spheres[p];
                C.inc(C_SPHERE_TESTS);
                float t;
                if (hit_sphere_t(w, cv3(sp->center), sp->radius, t) && t > 0.001f && t < lim)
                {
                    if (ANY) { occl = true; mode = M_DONE; }
                    else { tObj = t; iprim = p; }
                    if (EXISTS && tObj < 1e29f && tObj / iscale < 1e29f && tObj / iscale < bestT) { bestT = tObj / iscale; bestTObj = tObj; bestSlot = islot; bestPrim = iprim; mode = M_DONE; }   // what the fold at the end of this BLAS would accept (:65-77) AND the caller reads as a hit (a world t in [1e29, 1e30) is a miss: the walk goes on)
                }
            }
            else
            {
                // the next one or two triangles of the leaf leave together with this one: a leaf costs one memory round trip per
                // two or three triangles instead of one per triangle (three where most leaves hold three: the median split of
                // config 4's mesh; two where they hold four: config 5's -- measured; LT follows DPacked::leafTris, chosen at upload).  The tests
                // stay sequential per lane, in leaf order.
                FTri trs[LT];
#pragma unroll
                for (int q = 0; q < LT; q++) trs[q] = P.ftri[bj + q < bend ? bj + q : bend - 1];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < LT; q++)
                {
                    if (q > 0) { if (!(mode == M_BLEAF && bj + 1 < bend)) break; bj++; }
                    const FTri trr = trs[q];
                    C.inc(C_TRI_TESTS);
                    float t, bu, bv;
                    if (hit_tri_t(w, xyz(trr.v0), xyz(trr.v1), xyz(trr.v2), t, bu, bv))
                    {
                        if (!ANY)
                        {   // TraverseBLAS_Tri_Textured :196-227
                            C.inc(C_TRI_MT_HITS);
                            if (t > 0.001f && t < tObj)
                            {
                                C.inc(C_TRI_ACCEPTED);
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
                                if (EXISTS && accept && tObj < 1e29f && tObj / iscale < 1e29f && tObj / iscale < bestT) { bestT = tObj / iscale; bestTObj = tObj; bestSlot = islot; bestPrim = iprim; mode = M_DONE; }
                            }
                        }
                        else if (!(t <= 0.001f || t >= lim))
                        {   // AnyHit_Tri_Textured :292-317
                            C.inc(C_TRI_MT_HITS);
                            bool blocked = true;
                            if (kAlpha && (wbits(trr.v2) & FT_TEXTURED))
                            {
                                const hrt_material* mat = &S.materials[wbits(trr.v1)];
                                int ati = mat->AlphaTexIndex;
                                if (mat->HasAlphaMap != 0 && ati >= 0 && ati < S.n_texInfos)
                                {
                                    C.inc(C_TRI_ACCEPTED);
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
                            if (blocked) { occl = true; mode = M_DONE; }
                        }
                    }
                }
            }
            bj++;
            if (mode == M_BLEAF && bj == bend)
            {
                cur = bskip; mode = M_BLAS;
                if (!(cur < blasEnd))
                {
                    if (!ANY && tObj < 1e29f)
                    {
                        float tWorld = tObj / iscale;
                        if (tWorld < bestT) { bestT = tWorld; bestTObj = tObj; bestSlot = islot; bestPrim = iprim; }
                    }
                    w = park.get();
                    mode = M_TLEAF;
                    if (li == lend) { cur = lskip; mode = (cur == kEnd) ? M_DONE : M_TLAS; }
                }
            }
        }

        WTIME(15);
        // ---------------- retire finished rays
#ifdef HRT_WALK_STATS
        { const int a = __popcll(__ballot(mode == M_DONE)); if (a) { WSTAT(7, 1); WSTAT(8, a); } }
#endif
        if (mode == M_DONE)
        {
            if (kTies && anom) tie = true;
            if (ALT && tie)
            {   // a non-finite ray, or (kTies) two instances at exactly the same distance: the uploaded tree decides
                tie = false;
                if (ANY) occl = exact.template occluded_ext<COUNT, true>(w, tMaxW, C, park_mem);
                else
                {
                    bestT = 1e30f; bestTObj = 0.f; bestSlot = -1; bestPrim = -1;
                    exact.template closest_raw<COUNT, true>(w, bestT, bestTObj, bestSlot, bestPrim, C, park_mem);
                    if (kTies && bestSlot >= 0) bestSlot = P.slotMap[bestSlot];
                }
            }
            WalkResult r; r.t = bestT; r.tObj = bestTObj; r.slot = bestSlot; r.prim = bestPrim; r.occluded = occl;
            done(rayIdx, r);
            mode = M_IDLE;
        }
        WTIME(16);
    }
#ifdef HRT_WALK_STATS
    ws[11] = 1;
    if (lane == 0) for (int i = 0; i < 17; i++) atomicAdd(&g_walk_stats[ANY ? 0 : 1][i], ws[i]);
#endif
}

} // namespace hrt
