// hrt_walker_tl.hpp -- the LDS-staged, treelet-queued BVH walker (production frames of scenes with big triangle meshes).
//
// The persistent-wave walker of hrt_walker.hpp reads every node and triangle record of a mesh BLAS from global memory: with
// incoherent rays almost every wave-step has a lane that misses the L2 (a 7 ... 65 MB tree against 4 MiB per XCD), so a step
// costs a trip to HBM (profiles/r02_walk_experiments.txt).  Here the lower part of the tree is read from LDS instead:
//
//   * hrt_treelets.hpp cuts every big mesh BLAS into treelets (maximal subtrees that fit a workgroup's LDS budget) under a
//     small reduced tree whose top records stay in LDS for the whole kernel;
//   * a ray that reaches a treelet (its root's box test passes: a PORTAL of the reduced tree) is SUSPENDED: its walk state
//     goes to the ray's own slot in HBM, the treelet id to a key plane, the lane is refilled;
//   * between rounds the suspended rays are binned by treelet (a histogram kept in LDS by the walk kernel, one scan, one
//     scatter of ray indices: no atomics on global memory per ray);
//   * in the next round a workgroup takes a span of the binned list, stages each treelet of its span ONCE with coalesced
//     global loads -> ds_write, and walks all rays queued for it against LDS; a ray that leaves the treelet goes on through
//     the reduced tree (LDS again) and the TLAS until it ends or meets the next portal.
//   PHASE 0: fresh rays (path ranges), no treelet staged.   PHASE 1: one round over the binned rays.
//   PHASE 2: clean-up -- whatever is still suspended after the last round is finished from global memory, without suspending.
//
// A ray's sequence of box and primitive tests, their limits and their order are the uploaded tree's
// (SceneDeviceViews.cs:30-327): a portal's test IS the test of the treelet's root node, the walk resumes at the root's hit
// child with the state it was suspended with, and the object-space ray is recomputed from the world ray by the same
// expression.  Rays are independent, so the order in which queues are served changes no result.
#pragma once
#include "hrt_walker.hpp"
#include "hrt_treelets.hpp"

namespace hrt {

enum { T_IDLE = 0, T_TLAS, T_TLEAF, T_BLAS, T_BLEAF, T_RED, T_TL, T_TLF, T_DONE, T_SUSP };
constexpr float kTMaxAny = 1e29f;          // tMax of a shadow ray (RTRay.cs:623)
constexpr int kTlLdsBurst = 4;             // LDS node steps per global node step of the same iteration
constexpr int kTlSpanMin = 512;            // rays per workgroup span of a round, at least

// Tuning aid (variant build -DHRT_TL_STATS, tools/tl_stats.py): per PHASE and walk kind
// [0] rays started [1] suspended [2] finished [3] iterations [4] lanes in LDS node steps [5] lanes in global node steps [6] lanes in leaf steps
// [7] idle lanes summed over iterations [8..12] shader-clock cycles in refill / node steps / instance + leaf steps / retire / staging [13] waves [14] LDS node-step rounds [15] global rounds
#ifdef HRT_TL_STATS
__device__ unsigned long long g_tl_stats[8][2][16];      // [0] PHASE 0, [1..6] rounds of PHASE 1, [7] PHASE 2
#define TSTAT(i, v) ts[i] += (unsigned long long)(v)
#define TTIME(i) { const long long tnow_ = (long long)__builtin_readcyclecounter(); ts[i] += (unsigned long long)(tnow_ - tt_); tt_ = tnow_; }
#else
#define TSTAT(i, v)
#define TTIME(i)
#endif

struct TlQueues {                // per batch lane and walk kind
    int* key;                    // [cap] treelet a suspended ray waits for; -1: not suspended
    float4* state;               // [cap * (closest ? 2 : 1)] suspended walk state
    int* sorted;                 // [cap] ray indices binned by treelet
    int* hist;                   // [nTl] rays suspended per treelet since the last scan; followed by misc[32]
    int* misc;                   // [0] rays in `sorted`; [8..15] range hand-out counters of the clean-up launch
    int* offs;                   // [nTl + 1] first position of each treelet in `sorted`
    int* curs;                   // [nTl] scatter cursors
};

// dynamic LDS of the walk kernels: [treelet][reduced-tree prefix][ray park][histogram][misc]
struct TlShared { float4* tl; const float4* red; float (*park)[256]; int* hist; int* misc; };
HRT_D TlShared tl_shared(int tlRegion, int redLds, int histBins)
{
    extern __shared__ float4 tl_smem[];
    TlShared s;
    char* p = reinterpret_cast<char*>(tl_smem);
    s.tl = reinterpret_cast<float4*>(p); p += tlRegion;
    s.red = reinterpret_cast<const float4*>(p); p += redLds * 32;
    s.park = reinterpret_cast<float (*)[256]>(p); p += 9 * 256 * 4;
    s.hist = reinterpret_cast<int*>(p); p += histBins * 4;
    s.misc = reinterpret_cast<int*>(p);
    return s;
}
inline size_t tl_shared_bytes(int tlRegion, int redLds, int histBins) { return (size_t)tlRegion + (size_t)redLds * 32 + 9 * 256 * 4 + (size_t)histBins * 4 + 64; }

// fetch(i, ray): world ray of entry i (false: the entry carries none); done(i, result) as in walk_queue.
// PHASE 0 / 2: nextSeg hands out runs of ENTRIES (path slots / shadow requests); PHASE 1: runs of POSITIONS in Q.sorted.
template <int FEAT, bool ANY, bool EXISTS, int LT, int PHASE, class NextSeg, class Fetch, class Done>
HRT_D void walk_tl(const TracerPackedT<FEAT>& tr, const DTreelets& T, const TlQueues& Q, const TlShared& sh, int histBins,
                   const Treelet tlc, NextSeg nextSeg, Fetch fetch, Done done, int statSlot = 0)
{
    static_assert((FEAT & 1) != 0, "treelets belong to general instances");
    static_assert(!(EXISTS && ANY), "");
    constexpr bool kAlpha = (FEAT & 2) != 0;
    const DPacked& P = tr.P;
    const DScene& S = tr.S;
    Tex tex(S);
    RayPark park; park.sh = sh.park;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float4* sTl = sh.tl;
    const float4* sRed = sh.red;
    const int redLds = T.redLds;
    const int tlLo = tlc.nodeLo, tlHi = tlc.nodeHi, triLo = tlc.triLo, exitRed = tlc.exitRed;      // PHASE 1: the staged treelet (wave-uniform)
    const int triBase = 2 * (tlHi - tlLo);                                                          // its triangle records follow its nodes

    int segBase = 0, segN = 0, segCur = 0;
    bool more = true;
    int mode = T_IDLE, rayIdx = -1;
    Ray w; w.o = w.d = w.inv = mk3(0.f, 0.f, 0.f);
    float bestT = 1e30f, bestTObj = 0.f; int bestSlot = -1, bestPrim = -1;
    bool occl = false;
    int cur = 0, li = 0, lend = 0, lskip = kEnd;
    int bj = 0, bend = 0, bskip = kEnd, leafRet = T_BLAS;
    int blasEnd = 0, after = -1, iflags = 0, islot = 0; float iscale = 1.f, tObj = 1e30f; int iprim = -1, skey = -1;
#ifdef HRT_TL_STATS
    unsigned long long ts[16] = {};
    long long tt_ = (long long)__builtin_readcyclecounter();
#endif

    // the BLAS of the current instance is exhausted: fold its result into the world result (:65-77), back to the TLAS leaf
    auto blas_done = [&]() {
        if (!ANY && tObj < 1e29f)
        {
            const float tWorld = tObj / iscale;
            if (tWorld < bestT) { bestT = tWorld; bestTObj = tObj; bestSlot = islot; bestPrim = iprim; }
        }
        w = park.get();
        mode = T_TLEAF;
        if (li == lend) { cur = lskip; mode = (cur == kEnd) ? T_DONE : T_TLAS; }
    };
    auto after_step = [&]() {
        if (PHASE == 1 && mode == T_TL && !(cur < tlHi)) { cur = exitRed; mode = T_RED; }
        if (mode == T_BLAS && !(cur < blasEnd))
        {
            if (PHASE == 2 && after != -1) { cur = after; after = -1; mode = T_RED; }      // left a treelet walked from global memory
            else blas_done();
        }
        if (mode == T_RED && cur == kEnd) blas_done();
        if (mode == T_TLAS && cur == kEnd) mode = T_DONE;
    };
    auto step = [&](const NodeQ& nd) {
        int sk = wbits(nd.hi);
        const int cnt = (int)((unsigned)sk >> 28);
        sk &= kEnd;
        const bool top = mode == T_TLAS;
        const float lim = top ? (ANY ? kTMaxAny : bestT) : (ANY ? kTMaxAny * iscale : tObj);
        if (!hit_box(w, nd.lo, nd.hi, 0.001f, lim)) cur = sk;
        else if (cnt > 0)
        {
            if (top) { li = wbits(nd.lo); lend = li + cnt; lskip = sk; mode = T_TLEAF; }
            else { bj = wbits(nd.lo); bend = bj + cnt; bskip = sk; leafRet = mode; mode = (PHASE == 1 && mode == T_TL) ? T_TLF : T_BLEAF; }
        }
        else
        {
            const int link = wbits(nd.lo);
            if (mode == T_RED && (link & kPortalBit))
            {   // the root of a treelet, hit
                const int t = link & kEnd;
                if (PHASE == 2) { const Treelet tt = T.tl[t]; cur = tt.nodeLo + 1; blasEnd = tt.nodeHi; after = tt.exitRed; mode = T_BLAS; }
                else { skey = t; mode = T_SUSP; }
            }
            else cur = link & kEnd;
        }
    };
    // walk state of a suspended ray back into the lane; the ray stands behind the (hit) root of treelet `t`
    auto resume = [&](int t) {
        fetch(rayIdx, w);
        const float4 s0 = Q.state[ANY ? rayIdx : 2 * rayIdx];
        const int a = __float_as_int(s0.x);
        li = a & kEnd; lend = li + (int)((unsigned)a >> 28); lskip = __float_as_int(s0.y);
        occl = false; tObj = 1e30f; iprim = -1; bestT = 1e30f; bestTObj = 0.f; bestSlot = -1; bestPrim = -1;
        if (!ANY)
        {
            const float4 s1 = Q.state[2 * rayIdx + 1];
            bestT = s0.z; bestTObj = s0.w;
            bestSlot = __float_as_int(s1.x); bestPrim = __float_as_int(s1.y); tObj = s1.z; iprim = __float_as_int(s1.w);
        }
        islot = li - 1;
        const FInst f = P.finst[islot];
        iflags = wbits(f.a); iscale = f.c.z;
        park.put(w);
        w = tr.object_ray(w, iflags, wbits(f.b));
        after = -1;
        if (PHASE == 1) { cur = tlLo + 1; mode = T_TL; }
        else { const Treelet tt = T.tl[t]; cur = tt.nodeLo + 1; blasEnd = tt.nodeHi; after = tt.exitRed; mode = T_BLAS; }
    };

    for (;;)
    {
        // ---------------- refill idle lanes
        {
            unsigned long long idle = __ballot(mode == T_IDLE);
            int nIdle = __popcll(idle);
            TSTAT(3, 1); TSTAT(7, nIdle);
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
                    if (mode == T_IDLE && rank < avail)
                    {
                        const int e = segBase + segCur + rank;
                        if (PHASE == 0)
                        {
                            rayIdx = e;
                            bestT = 1e30f; bestTObj = 0.f; bestSlot = -1; bestPrim = -1; occl = false;
                            if (fetch(rayIdx, w)) { cur = 0; mode = T_TLAS; TSTAT(0, 1); }
                            else mode = T_DONE;                     // entry without a ray (path already ended)
                        }
                        else if (PHASE == 1) { rayIdx = Q.sorted[e]; resume(0); TSTAT(0, 1); }
                        else
                        {
                            const int k = Q.key[e];
                            if (k >= 0) { rayIdx = e; resume(k); TSTAT(0, 1); }  // others stay idle: nothing of theirs is pending
                        }
                    }
                    segCur += nIdle < avail ? nIdle : avail;
                    idle = __ballot(mode == T_IDLE);
                    nIdle = __popcll(idle);
                }
            }
            if (!more && __popcll(__ballot(mode == T_IDLE)) == 64) break;
        }
        TTIME(8);

        // ---------------- node steps.  Lanes whose node is in global memory issue their load first; lanes whose node is in
        // LDS (staged treelet, top of the reduced tree) take up to kTlLdsBurst steps while it is in flight.
        for (int burst = 0; burst < kNodeBurst; burst++)
        {
            const bool walking = mode == T_TLAS || mode == T_BLAS || mode == T_RED || (PHASE == 1 && mode == T_TL);
            const int nWalk = __popcll(__ballot(walking));
            if (nWalk == 0 || (burst > 0 && nWalk < 24)) break;
            const bool inLds = walking && ((PHASE == 1 && mode == T_TL) || (mode == T_RED && cur < redLds));
            const bool glob = walking && !inLds;
#ifdef HRT_TL_STATS
            { const int a = __popcll(__ballot(inLds)), b = __popcll(__ballot(glob)); TSTAT(4, a); TSTAT(5, b); if (a) TSTAT(14, 1); if (b) TSTAT(15, 1); }
#endif
            // global lanes: kLook consecutive records leave together (walk order: the record after a hit inner node or a missed leaf is the
            // next one, usually in the same 128-byte line); the reduced tree has explicit links, one record there
            NodeQ ngs[kLook];
            const int m0 = mode;
            int last = 0;
            if (glob)
            {
                const NodeQ* nodes = mode == T_TLAS ? P.tlas : (mode == T_RED ? T.red : P.blas);
                last = mode == T_TLAS ? P.nTlas - 1 : (mode == T_RED ? cur : blasEnd - 1);
#pragma unroll
                for (int k = 0; k < kLook; k++) ngs[k] = nodes[cur + k <= last ? cur + k : last];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (inLds)
            {
                for (int k = 0; k < kTlLdsBurst; k++)
                {
                    NodeQ nd;
                    if (PHASE == 1 && mode == T_TL) { nd.lo = sTl[2 * (cur - tlLo)]; nd.hi = sTl[2 * (cur - tlLo) + 1]; }
                    else { nd.lo = sRed[2 * cur]; nd.hi = sRed[2 * cur + 1]; }
                    step(nd);
                    after_step();
                    if (!((PHASE == 1 && mode == T_TL) || (mode == T_RED && cur < redLds))) break;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (glob)
            {
#pragma unroll
                for (int k = 0; k < kLook; k++)
                {
                    const int here = cur;
                    step(ngs[k]);
                    after_step();
                    if (!(mode == m0 && cur == here + 1 && here < last)) break;
                }
            }
        }

        TTIME(9);
        // ---------------- one TLAS leaf entry (an instance)
#ifdef HRT_TL_STATS
        TSTAT(6, __popcll(__ballot(mode == T_TLEAF || mode == T_BLEAF || mode == T_TLF)));
#endif
        if (mode == T_TLEAF)
        {
            const FInst f = P.finst[li];
            const int flags = wbits(f.a);
            if (flags & FI_FAST_SPHERE)
            {
                const float lim = ANY ? kTMaxAny : 1e30f;
                if (hit_box(w, f.a, f.b, 0.001f, lim))
                {
                    float t;
                    if (hit_sphere_t(w, xyz(f.c), f.c.w, t) && t > 0.001f && t < lim)
                    {
                        if (ANY) { occl = true; mode = T_DONE; }
                        else if (t < 1e29f && t < bestT) { bestT = t; bestTObj = t; bestSlot = li; bestPrim = wbits(f.b); if (EXISTS) mode = T_DONE; }
                    }
                }
                li++;
            }
            else
            {   // general instance: park the world ray, walk its BLAS with the object-space ray
                islot = li; iflags = flags; iscale = f.c.z;
                const int root = __float_as_int(f.c.x), end = __float_as_int(f.c.y);
                tObj = 1e30f; iprim = -1; after = -1;
                park.put(w);
                w = tr.object_ray(w, flags, wbits(f.b));
                li++;
                const int rr = (!(flags & FI_SPHERESET) && root < end) ? T.redOfRoot[root] : -1;
                if (rr >= 0) { cur = rr; mode = T_RED; }
                else
                {
                    cur = root; blasEnd = end; mode = T_BLAS;
                    if (!(cur < blasEnd)) { w = park.get(); mode = T_TLEAF; }          // empty BLAS
                }
            }
            if (mode == T_TLEAF && li == lend) { cur = lskip; mode = (cur == kEnd) ? T_DONE : T_TLAS; }
        }

        // ---------------- one BLAS leaf step: records from global memory (T_BLEAF) or from the staged treelet (T_TLF)
        auto tri_step = [&](auto load) {
            const float lim = ANY ? kTMaxAny * iscale : tObj;
            FTri trs[LT];
#pragma unroll
            for (int q = 0; q < LT; q++) trs[q] = load(bj + q < bend ? bj + q : bend - 1);
            __builtin_amdgcn_sched_barrier(0);
            const int leafMode = mode;
#pragma unroll
            for (int q = 0; q < LT; q++)
            {
                if (q > 0) { if (!(mode == leafMode && bj + 1 < bend)) break; bj++; }
                const FTri trr = trs[q];
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
                                const int ati = mat->AlphaTexIndex;
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
                            if (EXISTS && accept && tObj < 1e29f && tObj / iscale < 1e29f && tObj / iscale < bestT)
                            { bestT = tObj / iscale; bestTObj = tObj; bestSlot = islot; bestPrim = iprim; mode = T_DONE; }
                        }
                    }
                    else if (!(t <= 0.001f || t >= lim))
                    {   // AnyHit_Tri_Textured :292-317
                        bool blocked = true;
                        if (kAlpha && (wbits(trr.v2) & FT_TEXTURED))
                        {
                            const hrt_material* mat = &S.materials[wbits(trr.v1)];
                            const int ati = mat->AlphaTexIndex;
                            if (mat->HasAlphaMap != 0 && ati >= 0 && ati < S.n_texInfos)
                            {
                                float uu, vv;
                                tr.tri_uv(wbits(trr.v0), bu, bv, uu, vv);
                                const hrt_tex_info ainfo = S.texInfos[ati];
                                const float aPoint = tex.mask_point(ainfo, uu, vv);
                                const float cutoff = mat->AlphaCutoff;
                                if (aPoint < cutoff - 0.10f) blocked = false;
                                else if (aPoint >= cutoff + 0.10f) blocked = true;
                                else blocked = !(tex.mask_linear(ainfo, uu, vv) < cutoff);
                            }
                        }
                        if (blocked) { occl = true; mode = T_DONE; }
                    }
                }
            }
            bj++;
            if (mode == leafMode && bj == bend) { cur = bskip; mode = leafRet; after_step(); }
        };
        if (mode == T_BLEAF)
        {
            if (iflags & FI_SPHERESET)
            {
                const float lim = ANY ? kTMaxAny * iscale : tObj;
                const int p = S.spherePrimIdx[bj];
                const hrt_sphere* sp = &S.spheres[p];
                float t;
                if (hit_sphere_t(w, cv3(sp->center), sp->radius, t) && t > 0.001f && t < lim)
                {
                    if (ANY) { occl = true; mode = T_DONE; }
                    else { tObj = t; iprim = p; }
                    if (EXISTS && tObj < 1e29f && tObj / iscale < 1e29f && tObj / iscale < bestT) { bestT = tObj / iscale; bestTObj = tObj; bestSlot = islot; bestPrim = iprim; mode = T_DONE; }
                }
                bj++;
                if (mode == T_BLEAF && bj == bend) { cur = bskip; mode = leafRet; after_step(); }
            }
            else tri_step([&](int j) { return P.ftri[j]; });
        }
        if (PHASE == 1 && mode == T_TLF)
            tri_step([&](int j) { FTri r; const int b = triBase + 3 * (j - triLo); r.v0 = sTl[b]; r.v1 = sTl[b + 1]; r.v2 = sTl[b + 2]; return r; });

        TTIME(10);
        // ---------------- retire: finished rays hand over their result, suspended ones their state
        if (mode == T_DONE)
        {
            TSTAT(2, 1);
            WalkResult r; r.t = bestT; r.tObj = bestTObj; r.slot = bestSlot; r.prim = bestPrim; r.occluded = occl;
            done(rayIdx, r);
            Q.key[rayIdx] = -1;
            mode = T_IDLE;
        }
        else if (mode == T_SUSP)
        {
            TSTAT(1, 1);
            const int a = li | (int)((unsigned)(lend - li) << 28);
            if (ANY) Q.state[rayIdx] = make_float4(__int_as_float(a), __int_as_float(lskip), 0.f, 0.f);
            else
            {
                Q.state[2 * rayIdx] = make_float4(__int_as_float(a), __int_as_float(lskip), bestT, bestTObj);
                Q.state[2 * rayIdx + 1] = make_float4(__int_as_float(bestSlot), __int_as_float(bestPrim), tObj, __int_as_float(iprim));
            }
            Q.key[rayIdx] = skey;
            if (histBins > 0) atomicAdd(&sh.hist[skey], 1); else atomicAdd(&Q.hist[skey], 1);
            mode = T_IDLE;
        }
        TTIME(11);
    }
#ifdef HRT_TL_STATS
    ts[13] = 1;
    for (int i = 0; i < 16; i++)
    {
        unsigned long long v = ts[i];
        if (i < 8 && i != 3) for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);      // per-lane counts: sum over the wave
        if (lane == 0 && v) atomicAdd(&g_tl_stats[statSlot][ANY ? 0 : 1][i], v);
    }
#endif
}

// ---- round bookkeeping kernels -----------------------------------------------------------------------------------
// exclusive scan of the per-treelet counts -> offsets and scatter cursors; the counts are cleared for the next round
__device__ __forceinline__ void tl_scan_block(const TlQueues& Q, int nTl)
{
    __shared__ int s_part[1024];
    __shared__ int s_run;
    const int tid = threadIdx.x;
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < nTl; base += 1024)
    {
        const int i = base + tid;
        const int v = i < nTl ? Q.hist[i] : 0;
        s_part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1)
        {
            const int add = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += add;
            __syncthreads();
        }
        const int incl = s_part[tid], run = s_run;
        if (i < nTl) { Q.offs[i] = run + incl - v; Q.curs[i] = run + incl - v; Q.hist[i] = 0; }
        __syncthreads();
        if (tid == 1023) s_run = run + incl;
        __syncthreads();
    }
    if (tid == 0) { Q.offs[nTl] = s_run; Q.misc[0] = s_run; }
}

// ray indices of the suspended rays of one workgroup's four ranges -> Q.sorted, binned by treelet.  cnt: entries per range.
__device__ __forceinline__ void tl_scatter_block(const TlQueues& Q, int nTl, const int* cnt, int nRanges, int range)
{
    __shared__ int s_hist[kTlHistLds];
    if (Q.misc[0] == 0) return;                                   // nothing was suspended (uniform: every workgroup leaves)
    const int tid = threadIdx.x, lane = tid & 63;
    const bool inLds = nTl <= kTlHistLds;
    if (inLds) for (int b = tid; b < nTl; b += 256) s_hist[b] = 0;
    __syncthreads();
    const int n = (range >= 0 && range < nRanges) ? cnt[range] : 0;
    const long long base = (long long)range * kRange;
    int keys[kRange / 64], ranks[kRange / 64];
#pragma unroll
    for (int it = 0; it < kRange / 64; it++)
    {
        const int i = it * 64 + lane;
        keys[it] = i < n ? Q.key[base + i] : -1;
        ranks[it] = 0;
        if (keys[it] >= 0) ranks[it] = inLds ? atomicAdd(&s_hist[keys[it]], 1) : atomicAdd(&Q.curs[keys[it]], 1);
    }
    __syncthreads();
    if (inLds)
    {
        for (int b = tid; b < nTl; b += 256) { const int c = s_hist[b]; if (c) s_hist[b] = atomicAdd(&Q.curs[b], c); }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < kRange / 64; it++)
        if (keys[it] >= 0) Q.sorted[(inLds ? s_hist[keys[it]] : 0) + ranks[it]] = (int)(base + it * 64 + lane);
}

} // namespace hrt
