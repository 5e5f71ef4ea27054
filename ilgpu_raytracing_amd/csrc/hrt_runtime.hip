// hrt_runtime.hip -- libhip_raytrace.so: kernels' entry points + the C ABI of
// include/hip_raytrace.h (context, scene upload, frame render, multi-device row tiling).
//
// Replaces the ILGPU accelerator / kernel-launch layer of the reference
// (Engine/RTRenderer.cs:66-68,85-86,118-120,152-153,164,181-205,233; Engine/Scene.cs:258-279,
//  370-377; Engine/Framebuffer.cs:60-97,127-146,228-253).  HIP runtime only: no torch, no
// CPU fallback -- every entry point fails with HRT_ERR_NO_DEVICE / HRT_ERR_HIP when no
// MI355X is usable.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <cmath>
#include <cfloat>
#include <utility>
#include "hrt_device.hpp"
#include "hrt_trace_packed.hpp"
#include "hrt_wavefront.hpp"
#include "hrt_walker_tl.hpp"
#include "hrt_bvh.hpp"
#include "hrt_post.hpp"
#include "../../include/hip_raytrace.h"
#ifdef HRT_TEST_HOOKS
#include "../../include/hrt_test_hooks.h"
#endif

using namespace hrt;


// ---------------------------------------------------------------------------------------
// Pixel <-> lane mapping.  A 256-thread workgroup shades a 32x8 pixel tile: each of its 4
// waves owns one 8x8 sub-tile (lane l -> (l&7, l>>3)), so the 64 rays of a wave leave the
// camera through a compact square and walk nearly the same BVH nodes.  Workgroup ids are
// dealt round-robin over the 8 XCDs by the dispatcher; remap() hands every XCD one
// contiguous band of tiles so each private 4 MiB L2 caches one region of the BVH instead
// of all of it (bijective form of the T1 remap, cdna_hip_programming.md).
// ---------------------------------------------------------------------------------------
struct TileMap { int tilesX, tilesY, nTiles, wpb; };      // wpb: waves (8x8 pixel tiles) per workgroup, side by side

__device__ __forceinline__ bool tile_pixel(const TileMap& tm, const FrameK& k, int& x, int& y, int orig)
{
    int q = tm.nTiles >> 3, r = tm.nTiles & 7;
    int xcd = orig & 7, seq = orig >> 3;
    int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
    int ty = tile / tm.tilesX, tx = tile - ty * tm.tilesX;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    x = (tx * tm.wpb + wave) * 8 + (lane & 7);
    y = k.row_begin + (ty * k.strip_n + k.strip_i) * 8 + (lane >> 3);   // 8-row strips dealt round-robin over tiles
    return x < k.width && y < k.row_end;
}

// TR = TracerPacked (fast, device-private layout) or TracerRef (the reference's array layout);
// COUNT = work-counter build.
template <class TR, bool COUNT>
__global__ void __launch_bounds__(256)
hrt_primary_kernel(TR tr, FrameK k, DGBuffer gb, TileMap tm, unsigned long long* counters)
{
    Cnt<COUNT> C;
    int x, y;
    if (tile_pixel(tm, k, x, y, blockIdx.x)) primary_pixel<TR, COUNT>(tr, k, gb, y * k.width + x, C);
    C.flush(counters);
}

#ifndef HRT_PT_WAVES
#define HRT_PT_WAVES 4   // 128-VGPR cap: 4 waves/SIMD hide the dependent node loads better than 2 at 204 VGPRs (measured)
#endif
// The leaf-sweep tracer is the exception: at 5 waves/SIMD (102 VGPRs, 112 bytes of scratch per lane) the fused kernel of
// config 2 is 2 % faster than at 4 (VALU-bound: one more wave to issue from is worth the spill traffic); 6 and 3 lose 12 %.
// Every other tracer spills two to four times as much there and keeps 4.
template <class TR> struct PtWaves { static constexpr int value = HRT_PT_WAVES; };
#ifndef HRT_PT_WAVES_FLAT
#define HRT_PT_WAVES_FLAT (HRT_PT_WAVES + 1)
#endif
template <> struct PtWaves<TracerFlat> { static constexpr int value = HRT_PT_WAVES_FLAT; };
template <class TR, bool COUNT, bool REUSE = true>
__global__ void __launch_bounds__(256, PtWaves<TR>::value)
hrt_path_trace_kernel(TR tr, FrameK k, DGBuffer gb, DFramebuffer fb, DReservoir resPrev, DReservoir resCur,
                      long long nPix, TileMap tm, unsigned long long* counters)
{
    Cnt<COUNT> C;
    int x, y;
    if (tile_pixel(tm, k, x, y, blockIdx.x)) path_trace_pixel<TR, COUNT, false, REUSE>(tr, k, gb, fb, resPrev, resCur, nPix, y * k.width + x, C);
    C.flush(counters);
}

// Small tiles (a rank's share of a frame tiled over 4 or 8 GPUs, small images): one wave per 8x8 pixels running all samples
// fills the machine for about one round, and the launch lasts as long as its slowest wave.  The samples of a pixel are
// independent up to the ordered sum and the last-writer reservoir, so the launch is cut into sample groups (workgroup =
// tile x group) and a resolve pass puts the pixel together in sample order: same values, several rounds of shorter waves.
template <class TR, bool REUSE = true>
__global__ void __launch_bounds__(256, PtWaves<TR>::value)
hrt_path_trace_split_kernel(TR tr, FrameK k, DGBuffer gb, DFramebuffer fb, DReservoir resPrev, DReservoir resCur,
                            long long nPix, TileMap tm, hrt_float3* li, float* stage, int nGroups, int perGroup)
{
    Cnt<false> C;
    const int g = blockIdx.x / tm.nTiles;
    SplitK sk;
    sk.li = li; sk.stage = stage; sk.group = g; sk.nGroups = nGroups;
    sk.sBegin = g * perGroup; sk.sEnd = min(sk.sBegin + perGroup, max(1, k.spp));
    const int tileBlock = blockIdx.x - g * tm.nTiles;
    sk.local = tileBlock * (int)blockDim.x + (int)threadIdx.x; sk.nLocal = tm.nTiles * (int)blockDim.x;
    int x, y;
    if (tile_pixel(tm, k, x, y, tileBlock)) path_trace_pixel<TR, false, true, REUSE>(tr, k, gb, fb, resPrev, resCur, nPix, y * k.width + x, C, &sk);
}

__global__ void __launch_bounds__(256)
hrt_split_resolve_kernel(FrameK k, DGBuffer gb, DFramebuffer fb, DReservoir resCur, long long nPix, TileMap tm, const hrt_float3* li, const float* stage, int nGroups)
{
    int x, y;
    if (tile_pixel(tm, k, x, y, blockIdx.x)) split_resolve_pixel(k, gb, fb, resCur, y * k.width + x, li, stage, nGroups,
                                                                 (int)(blockIdx.x * blockDim.x + threadIdx.x), tm.nTiles * (int)blockDim.x);
}

// ---------------------------------------------------------------------------------------
// Streamed path-trace stage (hrt_wavefront.hpp): one wave per range of kRange path slots.
// Ranges are handed to workgroups through the same per-XCD contiguous remap as pixel tiles,
// so neighbouring ranges (neighbouring screen regions) share an L2.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int wf_range(int nRanges)
{
    int nBlocks = gridDim.x, orig = blockIdx.x;
    int q = nBlocks >> 3, r = nBlocks & 7;
    int xcd = orig & 7, seq = orig >> 3;
    int blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + seq;
    int range = blk * 4 + (threadIdx.x >> 6);
    return range < nRanges ? range : -1;
}

// FIRST: depth 0, vertices straight from the G-buffer (every wave takes part in the workgroup's packing of the live paths)
template <bool COUNT, bool FIRST>
__global__ void __launch_bounds__(256)
hrt_wf_shade_kernel(FrameK k, WfGeom g, DGBuffer gb, DReservoir resPrev, long long nPix, WfBuffers W, int vsel, int depth, unsigned long long* counters)
{
    Cnt<COUNT> C;
    int range = wf_range(W.nRanges);
    if (FIRST || range >= 0) wf_shade_wave<COUNT, FIRST>(k, g, gb, resPrev, nPix, W, vsel ? W.B : W.A, depth, range, C);
    C.flush(counters);
}

#ifndef HRT_WALK_BLOCKS_PER_CU
#define HRT_WALK_BLOCKS_PER_CU 4               // persistent workgroups per CU of a chained walk launch: 16 of the 32 wave slots a CU has at <= 64 VGPRs.
                                               // Fewer rays in flight, but their tree nodes stay in L1 / L2: 3 / 4 / 5 / 6 / 8 / 12 measured
#endif
constexpr int kWalkBlocksPerCU = HRT_WALK_BLOCKS_PER_CU;
#ifndef HRT_WALK_BLOCKS_PER_CU_2LANES
#define HRT_WALK_BLOCKS_PER_CU_2LANES 2        // ... per lane when two sample batches are in flight: 2 / 3 / 4 / 6 measured on configs 4 / 5 at 64 / 256 spp:
                                               // 314 / 322 / 328 / 342 ms and 1287 / 1298 / 1318 / 1387 ms
#endif
constexpr int kWalkBlocksPerCU2 = HRT_WALK_BLOCKS_PER_CU_2LANES;
// Walk launches either give every wave one path range (static) or let persistent waves pull ranges until none is
// left (chained, RangeGrab).  Measured, path stage of configs 3 / 4 / 5: static 33.5 / 56.4 / 32.8 ms, chained
// 36.2 / 37.2 / 21.7 ms: the triangle scenes gain 1.5x from full lanes, the sphere-instance scene is bound by L1
// accesses (more live lanes do not help it) and loses the L1 sharing of four neighbouring ranges per workgroup.
#ifndef HRT_CHAIN_FEAT0
#define HRT_CHAIN_FEAT0 0
#endif
#ifndef HRT_WF_TRACE_WAVES
#define HRT_WF_TRACE_WAVES 4
#endif
template <class TR, bool COUNT>
__global__ void __launch_bounds__(256, HRT_WF_TRACE_WAVES)
hrt_wf_shadow_kernel(TR tr, WfBuffers W, int vsel, int depth, unsigned long long* counters)
{
    Cnt<COUNT> C;
    int range = wf_range(W.nRanges);
    if (range >= 0) wf_shadow_wave<TR, COUNT>(tr, W, vsel ? W.B : W.A, depth, range, C);
    C.flush(counters);
}

template <class TR, bool COUNT>
__global__ void __launch_bounds__(256, HRT_WF_TRACE_WAVES)
hrt_wf_closest_kernel(TR tr, FrameK k, WfBuffers W, int vsel, int depth, unsigned long long* counters)
{
    Cnt<COUNT> C;
    int range = wf_range(W.nRanges);
    if (range >= 0) wf_closest_wave<TR, COUNT>(tr, k, W, vsel ? W.B : W.A, vsel ? W.A : W.B, depth, range, C);
    C.flush(counters);
}

// persistent-wave walk kernels (packed layout only) + the finish kernel that shades the winners
// ALT: `tr` walks the device-built tree over the same fast-sphere instances (boolean queries do not depend on the tree,
// hrt_walker.hpp); `exact` is the uploaded tree, for rays whose slab arithmetic is not finite
template <int FEAT, bool COUNT, bool ALT = false, int LT = 2>
__global__ void __launch_bounds__(256, HRT_WF_TRACE_WAVES)
hrt_wf_walk_shadow_kernel(TracerPackedT<FEAT> tr, TracerPackedT<FEAT> exact, WfBuffers W, int vsel, int depth, int chained, unsigned long long* counters)
{
    Cnt<COUNT> C;
    int own = -1;
    if (!chained) { own = wf_range(W.nRanges); if (own < 0) own = W.nRanges; }
    W = W.at_depth(depth);
    wf_walk_shadow_wave<FEAT, COUNT, ALT, LT>(tr, exact, W, vsel ? W.B : W.A, depth, W.grab + (depth * 2 + 0) * 8, own, C);
    C.flush(counters);
}

// EXISTS: the closest-hit walk of the last bounce, where only hit-or-miss is used (hrt_walker.hpp)
template <int FEAT, bool COUNT, bool EXISTS = false, bool ALT = false, int LT = 2>
__global__ void __launch_bounds__(256, HRT_WF_TRACE_WAVES)
hrt_wf_walk_closest_kernel(TracerPackedT<FEAT> tr, TracerPackedT<FEAT> exact, WfBuffers W, int depth, int chained, unsigned long long* counters)
{
    Cnt<COUNT> C;
    int own = -1;
    if (!chained) { own = wf_range(W.nRanges); if (own < 0) own = W.nRanges; }
    W = W.at_depth(depth);
    wf_walk_closest_wave<FEAT, COUNT, EXISTS, ALT, LT>(tr, exact, W, depth, W.grab + (depth * 2 + 1) * 8, own, C);
    C.flush(counters);
}

// ---------------------------------------------------------------------------------------
// Treelet-queued walks (hrt_walker_tl.hpp): production frames of scenes with big triangle meshes.
// PHASE 0 fresh rays of the path ranges; PHASE 1 one round over the rays binned by treelet, a span of the binned list per
// workgroup, each treelet of the span staged once in LDS; PHASE 2 clean-up of what is still suspended, from global memory.
// ---------------------------------------------------------------------------------------
template <int FEAT, bool ANY, bool EXISTS, int LT, int PHASE>
__global__ void __launch_bounds__(256, PHASE == 1 ? 2 : 4)
hrt_tl_walk_kernel(TracerPackedT<FEAT> tr, DTreelets T, TlQueues Q, WfBuffers W, int depth, int histBins, int tlRegion, int round)
{
    const TlShared sh = tl_shared(tlRegion, T.redLds, histBins);
    W = W.at_depth(depth);
    {   // the top of the reduced tree stays in LDS for the whole kernel
        float4* red = const_cast<float4*>(sh.red);
        const float4* src = reinterpret_cast<const float4*>(T.red);
        for (int i = threadIdx.x; i < T.redLds * 2; i += 256) red[i] = src[i];
        for (int i = threadIdx.x; i < histBins; i += 256) sh.hist[i] = 0;
    }
    __syncthreads();
    auto fetch = [&](int i, Ray& r) {
        if (ANY)
        {
            const float4 qa = W.SQ.ld4(SQ_A, i), qb = W.SQ.ld4(SQ_B, i);
            r.o = mk3(qa.x, qa.y, qa.z); r.d = mk3(qa.w, qb.x, qb.y); r.inv = inv_dir(r.d);
            return true;
        }
        const float4 qa = W.R.ld4(RQ_A, i), qb = W.R.ld4(RQ_B, i);
        if (__float_as_int(qb.z) & RF_DEAD) return false;
        r.o = mk3(qa.x, qa.y, qa.z); r.d = mk3(qa.w, qb.x, qb.y); r.inv = inv_dir(r.d);
        return true;
    };
    auto done = [&](int i, const WalkResult& res) {
        if (ANY) { if (!res.occluded) W.SQ.sti(S_VIS, i, 1); }
        else W.R.st4(RQ_H, i, mkq(res.t, res.tObj, __int_as_float(res.slot), __int_as_float(res.prim)));
    };
    if (PHASE != 1)
    {
        RangeGrab G; G.init(PHASE == 0 ? W.grab + (depth * 2 + (ANY ? 0 : 1)) * 8 : Q.misc + 8, W.nRanges, -1);
        const int* cnt = (ANY ? W.cntS : W.cntA) + (size_t)depth * W.nRanges;
        walk_tl<FEAT, ANY, EXISTS, LT, PHASE>(tr, T, Q, sh, histBins, Treelet{},
            [&](int& base, int& n) { for (;;) { const int r = G.next(); if (r < 0) return false; n = cnt[r]; base = r * kRange; if (n > 0) return true; } },
            fetch, done, PHASE == 0 ? 0 : 7);
    }
    else
    {
        const int M = Q.misc[0];
        const int G = (int)gridDim.x;
        int span = (M + G - 1) / G;
        if (span < kTlSpanMin) span = kTlSpanMin;
        const long long p0 = (long long)blockIdx.x * span;
        const int p1 = (int)(p0 + span < (long long)M ? p0 + span : (long long)M);
        int p = p0 < M ? (int)p0 : M;
        while (p < p1)
        {
            int lo = 0, hi = T.nTl;                           // offs[lo] <= p < offs[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (Q.offs[mid] <= p) lo = mid; else hi = mid; }
            const Treelet tlc = T.tl[lo];
            const int segEnd = Q.offs[lo + 1] < p1 ? Q.offs[lo + 1] : p1;
            __syncthreads();                                   // every wave has left the treelet staged before
#ifdef HRT_TL_STATS
            const long long tStage0 = (long long)__builtin_readcyclecounter();
#endif
            {
                const int nN = 2 * (tlc.nodeHi - tlc.nodeLo), nT = 3 * (tlc.triHi - tlc.triLo);
                const float4* sn = reinterpret_cast<const float4*>(tr.P.blas + tlc.nodeLo);
                const float4* st = reinterpret_cast<const float4*>(tr.P.ftri + tlc.triLo);
                for (int i = threadIdx.x; i < nN; i += 256) sh.tl[i] = sn[i];
                for (int i = threadIdx.x; i < nT; i += 256) sh.tl[nN + i] = st[i];
                if (threadIdx.x == 0) sh.misc[0] = p;
            }
            __syncthreads();
#ifdef HRT_TL_STATS
            if ((threadIdx.x & 63) == 0) atomicAdd(&g_tl_stats[round < 6 ? round : 6][ANY ? 0 : 1][12], (unsigned long long)((long long)__builtin_readcyclecounter() - tStage0));
#endif
            walk_tl<FEAT, ANY, EXISTS, LT, 1>(tr, T, Q, sh, histBins, tlc,
                [&](int& base, int& n) {
                    int b = 0;
                    if ((threadIdx.x & 63) == 0) b = atomicAdd(&sh.misc[0], 64);
                    b = __builtin_amdgcn_readfirstlane(b);
                    if (b >= segEnd) return false;
                    base = b; n = segEnd - b < 64 ? segEnd - b : 64;
                    return true;
                },
                fetch, done, round < 6 ? round : 6);
            p = segEnd;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < histBins; b += 256) { const int v = sh.hist[b]; if (v) atomicAdd(&Q.hist[b], v); }
}

__global__ void __launch_bounds__(1024)
hrt_tl_scan_kernel(TlQueues Q, int nTl) { tl_scan_block(Q, nTl); }

__global__ void __launch_bounds__(256)
hrt_tl_scatter_kernel(TlQueues Q, int nTl, const int* cnt, int nRanges)
{
    tl_scatter_block(Q, nTl, cnt, nRanges, wf_range(nRanges));
}

template <int FEAT, bool COUNT>
__global__ void __launch_bounds__(256)
hrt_wf_finish_kernel(TracerPackedT<FEAT> tr, FrameK k, WfBuffers W, int vsel, int depth)
{
    int range = wf_range(W.nRanges);
    W = W.at_depth(depth);
    wf_finish_wave<FEAT, COUNT>(tr, k, W, vsel ? W.B : W.A, vsel ? W.A : W.B, depth, range);      // every wave: the four waves of a workgroup pack their survivors together
}

// finish of bounce `depth` + shade of bounce depth + 1 in one kernel (every bounce but the last): hrt_wavefront.hpp
template <int FEAT, bool COUNT>
__global__ void __launch_bounds__(256)
hrt_wf_finish_shade_kernel(TracerPackedT<FEAT> tr, FrameK k, WfGeom g, DGBuffer gb, DReservoir resPrev, long long nPix, WfBuffers W, int vsel, int depth, unsigned long long* counters)
{
    Cnt<COUNT> C;
    int range = wf_range(W.nRanges);
    W = W.at_depth(depth);
    wf_finish_shade_wave<FEAT, COUNT>(tr, k, g, gb, resPrev, nPix, W, vsel ? W.B : W.A, vsel ? W.A : W.B, depth, range, C);
    C.flush(counters);
}

__global__ void __launch_bounds__(256)
hrt_wf_resolve_kernel(FrameK k, WfGeom g, DGBuffer gb, DFramebuffer fb, DReservoir resCur, WfBuffers W)
{
    int ord = blockIdx.x * 256 + threadIdx.x;
    if (ord < g.nOrd) wf_resolve_pixel(k, g, gb, fb, resCur, W, ord);
}

// ---------------------------------------------------------------------------------------
// Presentation kernels (hrt_post.hpp)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
hrt_taa_resolve_kernel(TaaK p)
{
    __shared__ float lut[256];                       // sRGB byte -> linear, same expression as UnpackSRGB
    lut[threadIdx.x] = srgb_to_linear_byte((int)threadIdx.x);
    __syncthreads();
    const int total = p.outW * p.outH;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) taa_resolve_pixel(lut, p, idx);
}

__global__ void __launch_bounds__(256)
hrt_blit_kernel(const int32_t* src, long long srcLen, int32_t* dst, long long dstLen)     // RTRenderer.cs:281-285
{
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= dstLen || i >= srcLen) return;
    dst[i] = src[i];
}

__global__ void __launch_bounds__(256)
hrt_bilinear_upsample_kernel(const int32_t* src, int srcW, int srcH, int32_t* dst, int dstW, int dstH)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < dstW * dstH) bilinear_upsample_pixel(src, srcW, srcH, dst, dstW, dstH, i);
}

#ifdef HRT_TEST_HOOKS
// exactness probe: evaluates include/hrt_math.h on the device (tests compare with the oracle's bits)
__global__ void hrt_math_probe_kernel(int fn, int n, const float* x, const float* y, float* out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y ? y[i] : 0.f, r = 0.f;
    switch (fn) {
    case 0: r = hrt_sin(a); break;   case 1: r = hrt_cos(a); break;   case 2: r = hrt_tan(a); break;
    case 3: r = hrt_atan(a); break;  case 4: r = hrt_atan2(a, b); break; case 5: r = hrt_acos(a); break;
    case 6: r = hrt_asin(a); break;  case 7: r = hrt_rsqrt(a); break; case 8: r = hrt_sqrt(a); break;
    case 9: r = hrt_fmin(a, b); break; case 10: r = hrt_fmax(a, b); break;
    case 11: r = hrt_floor(a); break; case 12: r = hrt_round(a); break;
    case 13: { int v = hrt_f2i(a); r = __int_as_float(v); } break;
    case 14: r = 1.0f / a; break;    case 15: r = a / b; break;
    case 16: r = a * b + a; break;   // must NOT be contracted
    case 17: r = hrt_log(a); break;  case 18: r = hrt_exp(a); break;  case 19: r = hrt_pow(a, b); break;
    case 20: { float s, c; hrt_sincos(a, &s, &c); r = s; } break;
    case 21: { float s, c; hrt_sincos(a, &s, &c); r = c; } break;
    case 22: r = sqrt_normal_range(a); break;
    case 23: r = rsqrt_clamped(a); break;
    case 24: { float s, c; hrt_sincos_nonneg(a, &s, &c); r = s; } break;
    case 25: { float s, c; hrt_sincos_nonneg(a, &s, &c); r = c; } break;
    case 26: r = sqrt_normal_range<true>(a); break;
    case 27: r = rsqrt_clamped<true>(a); break;
    }
    out[i] = r;
}

// exhaustive device-side comparison of a trimmed function with its IEEE definition over every float of its domain:
// which 0: rsqrt_clamped(x) vs hrt_rsqrt(x) for x in [1e-20, +inf]; 1: sqrt_normal_range(x) vs hrt_sqrt(x) for x = +0, x in [2^-96, +inf]
__global__ void hrt_math_exhaustive_kernel(int which, unsigned long long* mismatches, unsigned* firstBad)
{
    const unsigned lo = which == 0 ? __float_as_uint(1e-20f) : __float_as_uint(0x1p-96f);
    const unsigned hi = 0x7F800000u;                              // +inf, inclusive
    unsigned long long bad = 0;
    for (unsigned long long u = (unsigned long long)lo + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; u <= hi; u += (unsigned long long)gridDim.x * blockDim.x)
    {
        const float x = __uint_as_float((unsigned)u);
        const float a = which == 0 ? rsqrt_clamped(x) : sqrt_normal_range(x);
        const float b = which == 0 ? hrt_rsqrt(x) : hrt_sqrt(x);
        if (__float_as_uint(a) != __float_as_uint(b)) { bad++; atomicMin(firstBad, (unsigned)u); }
    }
    if (which == 1 && blockIdx.x == 0 && threadIdx.x == 0 && __float_as_uint(sqrt_normal_range(0.f)) != 0u) bad++;
    if (bad) atomicAdd(mismatches, bad);
}

#endif // HRT_TEST_HOOKS

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
namespace {

thread_local std::string g_create_error;
TreeletLimits g_treelet_limits;              // shipped values unless a test lowered them (hrt_debug_set_treelet_limits)

#ifndef HRT_BATCH_LANES
#define HRT_BATCH_LANES 2
#endif
constexpr int kMaxLanes = 4, kBatchLanes = HRT_BATCH_LANES;      // sample batches in flight: 1 / 2 / 3 / 4 measured on configs 4 / 5 at 64 / 256 spp: 362 / 328 / 334 / 325 ms and 1412 / 1323 / 1376 / 1330 ms
static_assert(kBatchLanes >= 1 && kBatchLanes <= kMaxLanes, "");

struct DeviceState {
    int device_id = -1;
    int n_cu = 256;                            // compute units (MI355X: 256)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;             // shadow walks run beside the closest-hit walks of the same bounce
    static constexpr int kRing = 128;          // frames that may be in flight between two syncs
    hipEvent_t ev[kRing][4] = {};
    int ring_head = 0;                         // frames enqueued since the last synchronize
    bool ring_counts = false;
    std::vector<float> frame_ms[2];            // per-frame HIP-event times (launch 1, path-trace stage) of the frames the last hrt_synchronize collected
    // scene (15 arrays)
    void* scene[15] = {};
    DScene dscene{};
    void* packed[7] = {};                      // NodeQ tlas, FInst, NodeQ blas, FTri, NodeQ TLAS leaves in walk order (device-private repack)
    DPacked dpacked{};
    TlasDevice tl{};                           // device-side TLAS maintenance (hrt_bvh.hpp); aux arrays below
    void* tlaux[10] = {};                      // parent, nchild, arrive, scanIn, scanOut, sa, flags, cost, saBase, scanTmp + costPartial
    void* tlscratch = nullptr;                 // LBVH scratch, allocated on the first rebuild
    // a second tree over the same instances, built on the device at upload: what boolean queries of fast-sphere scenes walk
    TlasDevice tl2{};
    void* tl2mem[18] = {};          // [14] slot map, [15] renumbered copies, [16] what they permute, [17] scratch of the slot map
    bool any_built = false;                    // a second tree exists for this scene (any_ok: and it describes the scene as it is now)
    size_t ordX = 0, ordP = 0;                 // records in the renumbered copies of tlasX / of tlas
    DPacked dpackedAny{};
    bool any_ok = false;
    bool tlas_base_valid = false;              // saBase holds the node areas of the TLAS as it was last built
    bool tlas_lbvh = false;                    // the TLAS in use was BUILT on the device (Auto rebuilds an uploaded tree once: the LBVH walks faster)
    BlasDevice bl{};                           // triangle-mesh BLAS maintenance after vertex updates
    void* blaux[12] = {};                      // parent, nchild, subend, orig, arrive, ids of the TriMesh instances, kind, ids of the SphereSet instances, sa, saBase, growPartial, grow
    int n_mesh_inst = 0, n_sphere_inst = 0;
    // treelets of the big triangle-mesh BLASes (hrt_treelets.hpp) and the queues of the treelet walker, per batch lane and walk kind (0 shadow, 1 closest)
    void* tlmem[3] = {};                       // reduced trees, treelet table, BLAS root -> reduced root
    DTreelets dtl{};
    bool tl_ok = false;                        // the treelets describe the BLASes as they are now (a vertex update or BLAS rebuild drops them)
    void* tlq_mem[kMaxLanes] = {}; size_t tlq_bytes[kMaxLanes] = {};
    int max_lds = 65536;                       // LDS a workgroup may ask for
    bool blas_base_valid = false;              // saBase holds the node areas of the mesh BLASes as they were last built
    // presentation (TAAU history + display-size colour), device slot 0 only
    int32_t *present_color = nullptr, *taa_hist_color = nullptr, *taa_hist_obj = nullptr;
    int present_w = 0, present_h = 0; bool taa_history_valid = false;
    // streamed path-trace workspace
    // two sample batches are in flight at a time (lane 0 on stream / stream2, lane 1 on stream3 / stream4): each has its own workspace
    float* wf_mem[kMaxLanes] = {}; size_t wf_bytes[kMaxLanes] = {};
    int* wf_cnt[kMaxLanes] = {}; size_t wf_cnt_ints[kMaxLanes] = {};
    float* wf_accum = nullptr; size_t wf_accum_floats = 0;     // Lframe carried across the batches of a frame (one plane set, shared)
    hipStream_t laneStream[kMaxLanes][2] = {};  // lanes >= 1: main and side stream (lane 0 uses stream / stream2)
    hipEvent_t evLane[kMaxLanes][3] = {};      // per lane: fork, join, resolve done
    hipEvent_t evStage = nullptr;
    float* split_mem = nullptr; size_t split_floats = 0;    // fused kernel in sample groups: per-sample radiance + staged reservoirs
    // per-pixel buffers, full image size on every device (rows outside the tile stay untouched)
    int64_t nPix = 0;
    DGBuffer gb{};
    DFramebuffer fb{};
    DReservoir resA{}, resB{};
    unsigned long long* counters = nullptr;   // 2 x 10
    int row_begin = 0, row_end = 0;            // rows [row_begin,row_end) ...
    int strip_n = 1, strip_i = 0;              // ... of which this device owns 8-row strips s with s % strip_n == strip_i
    int n_strips = 0;
};

} // namespace

struct hrt_ctx {
    std::vector<DeviceState> dev;
    std::string err;
    bool scene_ready = false;
    bool packed_ok = false;                    // false: scene exceeds the packed layout's limits -> TracerRef
    int packed_feat = 3;                       // TracerPackedT<FEAT> variant of the committed scene
    int flat_leaves = 0;                       // > 0: TLAS leaves of a fast-sphere-only scene that fits TracerFlat
    bool own_in_world = false;                 // PackedHost::own_in_world of the uploaded scene
    bool small_scene = false;                  // <= kSmallSceneNodes BVH nodes: the walk is ALU-bound and L1-resident -> megakernel
    // state of hrt_scene_update_instances
    bool refit_ok = false;                     // every reachable TLAS node has one parent and <= 64 children
    bool feat_alpha = false;                   // the triangle half of packed_feat (does not change with the TLAS)
    int64_t n_inst = 0, n_tlas = 0, n_slots = 0, n_blas = 0;
    int tlas_leaves = 0;                       // reachable leaves of the TLAS in use
    bool tlas_on_device = false;               // the TLAS in use was refitted / rebuilt on the device (walk-order numbering)
    bool blas_refit_ok = false;                // hrt_scene_update_positions can refit every triangle-mesh BLAS
    bool blas_rebuild_ok = false;              // ... and rebuild it (HRT_REBUILD_BLAS)
    std::vector<MeshJob> mesh_jobs;
    int max_mesh_items = 0;
    int64_t n_positions = 0, n_spheres = 0;
    int64_t scene_count[15] = {};
    int width = 0, height = 0;
    long long max_resident_paths = 0;          // hrt_set_workspace_limit: 0 = kWfMaxPaths
    std::vector<std::pair<char*, size_t>> pinned;   // hrt_host_register: page-locked ranges of the caller (gather targets)
};

namespace {

const size_t kSceneElem[15] = {sizeof(hrt_bvh_node), 4, sizeof(hrt_instance), sizeof(hrt_bvh_node), 4, sizeof(hrt_sphere), 4,
                               sizeof(hrt_float3), sizeof(hrt_mesh_tri), sizeof(hrt_float2), sizeof(hrt_mesh_tri_uv), 4,
                               sizeof(hrt_material), sizeof(hrt_rgba32), sizeof(hrt_tex_info)};

int fail(hrt_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

// no exception crosses the C ABI: std::bad_alloc of the host-side vectors / strings -> HRT_ERR_OUT_OF_MEMORY, anything else -> HRT_ERR_HIP
int on_exception(hrt_ctx* c, const char* who) noexcept
{
    try
    {
        try { throw; }
        catch (const std::bad_alloc&) { return fail(c, HRT_ERR_OUT_OF_MEMORY, std::string(who) + ": out of host memory"); }
        catch (const std::exception& e) { return fail(c, HRT_ERR_HIP, std::string(who) + ": " + e.what()); }
        catch (...) { return fail(c, HRT_ERR_HIP, std::string(who) + ": unknown exception"); }
    }
    catch (...) { return HRT_ERR_OUT_OF_MEMORY; }      // not even the message could be stored
}

#define HIPCHK(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(ctx, e__ == hipErrorOutOfMemory ? HRT_ERR_OUT_OF_MEMORY : HRT_ERR_HIP,     \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                       \
    } while (0)

void free_pixels(DeviceState& d)
{
    void* ptrs[] = {d.gb.worldPos, d.gb.normalWS, d.gb.baseColor, d.gb.matId, d.gb.objId, d.gb.hitMask,
                    d.fb.color, d.fb.depth, d.fb.objectId, d.fb.cameraId, d.fb.radiance,
                    d.resA.L, d.resA.wi, d.resA.pdf, d.resA.w, d.resA.wSum, d.resA.m, d.resA.lightId,
                    d.resB.L, d.resB.wi, d.resB.pdf, d.resB.w, d.resB.wSum, d.resB.m, d.resB.lightId};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    d.gb = DGBuffer{}; d.fb = DFramebuffer{}; d.resA = DReservoir{}; d.resB = DReservoir{};
    d.nPix = 0;
}

// zero-fill is ordered on the device's own (non-blocking) stream: the null stream does not
// synchronise with it, so a hipMemset there could land after the first kernel's stores
template <class T> hipError_t dalloc(T*& p, int64_t n, hipStream_t stream)
{
    void* v = nullptr;
    hipError_t e = hipMalloc(&v, (size_t)n * sizeof(T));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(v, 0, (size_t)n * sizeof(T), stream);
    if (e != hipSuccess) return e;
    p = static_cast<T*>(v);
    return hipSuccess;
}

hipError_t alloc_res(DReservoir& r, int64_t n, hipStream_t st)
{
    hipError_t e;
    if ((e = dalloc(r.L, n, st)) != hipSuccess) return e;
    if ((e = dalloc(r.wi, n, st)) != hipSuccess) return e;
    if ((e = dalloc(r.pdf, n, st)) != hipSuccess) return e;
    if ((e = dalloc(r.w, n, st)) != hipSuccess) return e;
    if ((e = dalloc(r.wSum, n, st)) != hipSuccess) return e;
    if ((e = dalloc(r.m, n, st)) != hipSuccess) return e;
    return dalloc(r.lightId, n, st);
}

// GBuffer.EnsureLength / Framebuffer.EnsureLength / EnsureLowResBuffers: realloc on size change only
int ensure_pixels(hrt_ctx* c, DeviceState& d, int64_t nPix)
{
    if (d.nPix == nPix) return HRT_OK;
    HIPCHK(c, hipSetDevice(d.device_id));
    free_pixels(d);
    HIPCHK(c, dalloc(d.gb.worldPos, nPix, d.stream));
    HIPCHK(c, dalloc(d.gb.normalWS, nPix, d.stream));
    HIPCHK(c, dalloc(d.gb.baseColor, nPix, d.stream));
    HIPCHK(c, dalloc(d.gb.matId, nPix, d.stream));
    HIPCHK(c, dalloc(d.gb.objId, nPix, d.stream));
    HIPCHK(c, dalloc(d.gb.hitMask, nPix, d.stream));
    HIPCHK(c, dalloc(d.fb.color, nPix, d.stream));
    HIPCHK(c, dalloc(d.fb.depth, nPix, d.stream));
    HIPCHK(c, dalloc(d.fb.objectId, nPix, d.stream));
    HIPCHK(c, dalloc(d.fb.cameraId, 1, d.stream));
    HIPCHK(c, dalloc(d.fb.radiance, nPix, d.stream));
    HIPCHK(c, alloc_res(d.resA, nPix, d.stream));
    HIPCHK(c, alloc_res(d.resB, nPix, d.stream));
    d.nPix = nPix;
    return HRT_OK;
}

void free_present(DeviceState& d)
{
    if (d.present_color) (void)hipFree(d.present_color);
    if (d.taa_hist_color) (void)hipFree(d.taa_hist_color);
    if (d.taa_hist_obj) (void)hipFree(d.taa_hist_obj);
    d.present_color = d.taa_hist_color = d.taa_hist_obj = nullptr; d.present_w = d.present_h = 0; d.taa_history_valid = false;
}

void free_workspace(DeviceState& d)
{
    for (int j = 0; j < kMaxLanes; j++) { if (d.tlq_mem[j]) (void)hipFree(d.tlq_mem[j]); d.tlq_mem[j] = nullptr; d.tlq_bytes[j] = 0; }
    for (int j = 0; j < kMaxLanes; j++)
    {
        if (d.wf_mem[j]) (void)hipFree(d.wf_mem[j]);
        if (d.wf_cnt[j]) (void)hipFree(d.wf_cnt[j]);
        d.wf_mem[j] = nullptr; d.wf_cnt[j] = nullptr; d.wf_bytes[j] = 0; d.wf_cnt_ints[j] = 0;
    }
    if (d.wf_accum) (void)hipFree(d.wf_accum);
    d.wf_accum = nullptr; d.wf_accum_floats = 0;
    if (d.split_mem) (void)hipFree(d.split_mem);
    d.split_mem = nullptr; d.split_floats = 0;
}

void free_scene(DeviceState& d)
{
    for (int i = 0; i < 15; i++) { if (d.scene[i]) (void)hipFree(d.scene[i]); d.scene[i] = nullptr; }
    for (int i = 0; i < 7; i++) { if (d.packed[i]) (void)hipFree(d.packed[i]); d.packed[i] = nullptr; }
    for (int i = 0; i < 10; i++) { if (d.tlaux[i]) (void)hipFree(d.tlaux[i]); d.tlaux[i] = nullptr; }
    for (int i = 0; i < 18; i++) { if (d.tl2mem[i]) (void)hipFree(d.tl2mem[i]); d.tl2mem[i] = nullptr; }
    d.tl2 = TlasDevice{}; d.any_ok = false; d.any_built = false; d.ordX = d.ordP = 0;
    if (d.tlscratch) (void)hipFree(d.tlscratch);
    d.tlscratch = nullptr; d.tl = TlasDevice{}; d.tlas_base_valid = false; d.tlas_lbvh = false;
    for (int i = 0; i < 12; i++) { if (d.blaux[i]) (void)hipFree(d.blaux[i]); d.blaux[i] = nullptr; }
    d.bl = BlasDevice{}; d.n_mesh_inst = 0; d.n_sphere_inst = 0; d.blas_base_valid = false;
    for (int i = 0; i < 3; i++) { if (d.tlmem[i]) (void)hipFree(d.tlmem[i]); d.tlmem[i] = nullptr; }
    d.dtl = DTreelets{}; d.tl_ok = false;
}

// ---------------------------------------------------------------------------------------
// Scene validation + repack (host, once per commit).
// Validation: every index a kernel will dereference is range-checked and the node graphs are
// checked to be acyclic, so a malformed scene is an HRT_ERR_INVALID_ARG here instead of a GPU
// fault or a walk that never ends.  (The reference trusts its own builder and checks nothing.)
// ---------------------------------------------------------------------------------------
struct PackedHost {
    std::vector<NodeQ> tlas, blas, flat;     // flat: the TLAS leaves in walk order (TracerFlat)
    std::vector<FInst> finst;
    std::vector<FTri> ftri;
    std::vector<NodeQ> tlasX; // FEAT 0: TLAS with instance records inlined after their leaf (walker)
    int n_tlasX = 0;          // records in tlasX (0: not built)
    int n_flat = 0;           // leaves in `flat` (0: scene does not qualify)
    std::vector<int32_t> parent, nchild;     // TLAS, packed numbering: parent of a node (-1: none), children of an inner node
    std::vector<int32_t> bparent, bnchild, bsubend, borig;   // BLAS nodes of triangle meshes, packed numbering: parent (-1 root, -2 not maintained),
                                                             // children, end of the subtree's index range, index in the uploaded numbering
    std::vector<int32_t> bkind;                              // 0: node of no maintained BLAS, 1: triangle mesh, 2: sphere set
    int max_range[3] = {0, 0, 0};                            // largest node range of a maintained BLAS, per kind
    std::vector<int32_t> sphereInst;                         // ids of the SphereSet instances whose BLAS is maintained
    std::vector<int32_t> meshInst;                           // ids of the TriMesh instances whose BLAS is maintained
    std::vector<MeshJob> meshJobs;                           // the same, with what a device-side rebuild of the BLAS needs
    std::vector<std::pair<int64_t, int64_t>> meshRanges;     // node ranges of the maintained triangle-mesh BLASes (walk order): candidates for treelets
    bool blas_rebuild_ok = true;                             // every mesh's leaves list their triangles in one region of triPrimIdx
    bool blas_refit_ok = true;                               // every TriMesh BLAS can be refitted on the device
    bool refit_ok = true;     // the TLAS can be refitted bottom-up on the device (hrt_bvh.hpp)
    int reach_leaves = 0;     // reachable TLAS leaves
    bool nested = true;       // every reachable TLAS node's box lies inside its parent's, every fast-sphere instance's own box inside its leaf's:
                              // what "the boxes above only accelerate" (TracerFlat, the second tree) needs; the builders guarantee it, an uploaded tree may not
    bool own_in_world = true; // every fast-sphere instance's own box (its one-node BLAS) lies inside its worldBounds: a TLAS refitted or rebuilt on the
                              // device (leaf boxes = unions of worldBounds) is then nested like the builder's; false e.g. for an instance whose BLAS
                              // the position-indexed builder put over another sphere (Scene.cs:386-395)
    bool inst_once = false;   // the reachable TLAS leaves list every instance exactly once (a second tree over "the instances" answers the same queries)
    bool ok = true;           // false -> limits of the packed encoding exceeded (not an error)
    int feat = 0;             // TracerPackedT<FEAT> bits the committed scene needs
};

inline float bits_f(int v) { float f; std::memcpy(&f, &v, 4); return f; }
inline float4 mkf4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

bool is_identity(const hrt_affine3x4& m)
{
    return m.m00 == 1.f && m.m01 == 0.f && m.m02 == 0.f && m.m03 == 0.f && m.m10 == 0.f && m.m11 == 1.f && m.m12 == 0.f && m.m13 == 0.f &&
           m.m20 == 0.f && m.m21 == 0.f && m.m22 == 1.f && m.m23 == 0.f;
}

// nodes[lo,hi): every link in {-1} U [lo,hi) (TLAS: lo = 0), walk graph (left edge of inner nodes, skip edge of
// all nodes) acyclic from `root`.  Returns "" or an error text.
std::string check_nodes(const hrt_bvh_node* nodes, int64_t lo, int64_t hi, int64_t root, int64_t leafLimit, const char* what)
{
    if (hi <= lo) return "";
    for (int64_t i = lo; i < hi; i++)
    {
        const hrt_bvh_node& n = nodes[i];
        if (n.skipIndex < -1 || n.skipIndex >= hi) return std::string(what) + ": skipIndex out of range";   // skip may leave a BLAS range upward? no: reference skips are -1 or inside
        if (n.skipIndex != -1 && n.skipIndex < lo) return std::string(what) + ": skipIndex below its BLAS";
        if (n.count > 0) { if (n.first < 0 || (int64_t)n.first + n.count > leafLimit) return std::string(what) + ": leaf range outside the index list"; }
        else if (n.left < -1 || n.left >= hi || (n.left != -1 && n.left < lo)) return std::string(what) + ": left child out of range";
    }
    // iterative DFS, colours: 0 new, 1 on stack, 2 done
    std::vector<uint8_t> col((size_t)(hi - lo), 0);
    std::vector<std::pair<int64_t, int>> st;
    st.emplace_back(root, 0);
    col[(size_t)(root - lo)] = 1;
    while (!st.empty())
    {
        auto& top = st.back();
        const hrt_bvh_node& n = nodes[top.first];
        int64_t next = -2;
        if (top.second == 0) { top.second = 1; next = n.count > 0 ? -1 : n.left; }
        else if (top.second == 1) { top.second = 2; next = n.skipIndex; }
        else { col[(size_t)(top.first - lo)] = 2; st.pop_back(); continue; }
        if (next < 0) continue;
        uint8_t& c = col[(size_t)(next - lo)];
        if (c == 1) return std::string(what) + ": node links form a cycle";
        if (c == 0) { c = 1; st.emplace_back(next, 0); }
    }
    return "";
}

std::string validate_and_pack(const hrt_scene_desc* s, PackedHost& out)
{
    const int64_t nT = s->n_tlasNodes, nTI = s->n_tlasInstanceIndices, nI = s->n_instances, nB = s->n_blasNodes;
    const int64_t nSP = s->n_spherePrimIdx, nS = s->n_spheres, nTP = s->n_triPrimIdx, nPos = s->n_meshPositions, nTri = s->n_meshTris;
    const int64_t nTC = s->n_meshTexcoords, nTU = s->n_meshTriUVs, nTM = s->n_triMatIndex, nM = s->n_materials, nTx = s->n_texels, nTxI = s->n_texInfos;
    for (int64_t v : {nT, nTI, nI, nB, nSP, nS, nTP, nPos, nTri, nTC, nTU, nTM, nM, nTx, nTxI})
        if (v > 0x7FFFFFF0LL) return "array too long for 32-bit indices";
    for (int64_t i = 0; i < nTI; i++) if (s->tlasInstanceIndices[i] < 0 || s->tlasInstanceIndices[i] >= nI) return "tlasInstanceIndices entry out of range";
    for (int64_t i = 0; i < nSP; i++) if (s->spherePrimIdx[i] < 0 || s->spherePrimIdx[i] >= nS) return "spherePrimIdx entry out of range";
    for (int64_t i = 0; i < nTP; i++) if (s->triPrimIdx[i] < 0 || s->triPrimIdx[i] >= nTri) return "triPrimIdx entry out of range";
    if (nTri > 0 && (nTM < nTri || nTU < nTri)) return "triMatIndex / meshTriUVs shorter than meshTris";
    const int64_t limTC = nTC > 0 ? nTC : 1, limM = nM > 0 ? nM : 1;      // an empty list is one zeroed element (Scene.cs:370-377)
    for (int64_t i = 0; i < nTri; i++)
    {
        const hrt_mesh_tri& t = s->meshTris[i];
        if (t.i0 < 0 || t.i1 < 0 || t.i2 < 0 || t.i0 >= nPos || t.i1 >= nPos || t.i2 >= nPos) return "meshTris vertex index out of range";
        const hrt_mesh_tri_uv& u = s->meshTriUVs[i];
        if (u.t0 < 0 || u.t1 < 0 || u.t2 < 0 || u.t0 >= limTC || u.t1 >= limTC || u.t2 >= limTC) return "meshTriUVs index out of range";
        if (s->triMatIndex[i] < 0 || s->triMatIndex[i] >= limM) return "triMatIndex entry out of range";
    }
    for (int64_t i = 0; i < nTxI; i++)
    {
        const hrt_tex_info& ti = s->texInfos[i];
        if (ti.Width > 0 && ti.Height > 0 && (ti.Offset < 0 || (int64_t)ti.Offset + (int64_t)ti.Width * ti.Height > nTx)) return "texInfos entry outside texels";
    }
    if (nT > 0) { std::string e = check_nodes(s->tlasNodes, 0, nT, 0, nTI, "tlasNodes"); if (!e.empty()) return e; }
    for (int64_t i = 0; i < nI; i++)
    {
        const hrt_instance& in = s->instances[i];
        if (in.blasNodeCount < 0 || in.blasRoot < 0 || (int64_t)in.blasRoot + in.blasNodeCount > nB) return "instance BLAS range outside blasNodes";
        if (in.blasNodeCount == 0) continue;
        std::string e = check_nodes(s->blasNodes, in.blasRoot, (int64_t)in.blasRoot + in.blasNodeCount, in.blasRoot,
                                    in.type == HRT_BLAS_SPHERESET ? nSP : nTP, "blasNodes");
        if (!e.empty()) return e;
    }

    // ---- repack
    // Nodes are renumbered into walk order (depth-first, hit edge before skip edge): the child a ray enters after
    // a hit is the next node in memory, so a descent reads consecutive 32-byte records (4 per 128-byte line)
    // instead of jumping between the two halves of the builder's right-first numbering.  Pure permutation: every
    // walk visits the same nodes in the same order.  perm[old - lo] = new - lo.
    auto walk_order = [&](const hrt_bvh_node* src, int64_t lo, int64_t hi, int64_t root, std::vector<int32_t>& perm) -> int32_t {   // returns the number of reachable nodes
        const size_t n = (size_t)(hi - lo);
        if (HRT_ENV("HRT_BUILDER_ORDER")) { perm.resize(n); for (size_t i = 0; i < n; i++) perm[i] = (int32_t)i; return -1; }   // A/B knob
        perm.assign(n, -1);
        int32_t next = 0;
        std::vector<int64_t> st;
        st.push_back(root);
        while (!st.empty())
        {
            const int64_t i = st.back(); st.pop_back();
            if (i < lo || i >= hi || perm[(size_t)(i - lo)] >= 0) continue;
            perm[(size_t)(i - lo)] = next++;
            const hrt_bvh_node& b = src[i];
            st.push_back(b.skipIndex);
            if (b.count <= 0) st.push_back(b.left);
        }
        const int32_t reachable = next;
        for (size_t i = 0; i < n; i++) if (perm[i] < 0) perm[i] = next++;        // unreachable nodes keep a slot
        return reachable;
    };
    auto pack_range = [&](const hrt_bvh_node* src, int64_t lo, int64_t hi, const std::vector<int32_t>& perm, std::vector<NodeQ>& dst) {
        auto remap = [&](int32_t link) -> int { return (link < lo || link >= hi) ? kEnd : (int)(lo + perm[(size_t)(link - lo)]); };
        for (int64_t i = lo; i < hi; i++)
        {
            const hrt_bvh_node& b = src[i];
            int cnt = b.count > 0 ? b.count : 0;
            if (cnt > 15) out.ok = false;
            int link = cnt > 0 ? b.first : remap(b.left);
            int hiw = remap(b.skipIndex) | (int)((unsigned)(cnt & 15) << 28);
            NodeQ& q = dst[(size_t)(lo + perm[(size_t)(i - lo)])];
            q.lo = mkf4(b.boundsMin.X, b.boundsMin.Y, b.boundsMin.Z, bits_f(link));
            q.hi = mkf4(b.boundsMax.X, b.boundsMax.Y, b.boundsMax.Z, bits_f(hiw));
        }
    };
    auto alloc_nodes = [&](int64_t n, std::vector<NodeQ>& dst) {
        dst.resize((size_t)std::max<int64_t>(n, 1));
        std::memset(dst.data(), 0, dst.size() * sizeof(NodeQ));
        if (n == 0) { dst[0].lo.w = bits_f(kEnd); dst[0].hi.w = bits_f(kEnd); }    // the 1-element zero buffer: count 0, left 0 -> treat as end
        if (n >= kEnd) out.ok = false;
    };
    std::vector<int32_t> perm;
    alloc_nodes(nT, out.tlas);
    int32_t reachableT = -1;
    if (nT > 0) { reachableT = walk_order(s->tlasNodes, 0, nT, 0, perm); pack_range(s->tlasNodes, 0, nT, perm, out.tlas); }
    {   // parents and child counts for the device refit: the children of an inner node are the chain left, left.skip, ...
        // up to the node's own skip link (two nodes for both builders)
        const size_t n = out.tlas.size();
        out.parent.assign(n, -1); out.nchild.assign(n, 0);
        auto cntq = [&](size_t i) { return (int)((unsigned)__builtin_bit_cast(int, out.tlas[i].hi.w) >> 28); };
        auto skipq = [&](size_t i) { return __builtin_bit_cast(int, out.tlas[i].hi.w) & kEnd; };
        for (size_t i = 0; i < (size_t)nT; i++)
        {
            if (cntq(i) > 0) { if (reachableT < 0 || (int32_t)i < reachableT) out.reach_leaves++; continue; }
            int c = __builtin_bit_cast(int, out.tlas[i].lo.w) & kEnd;
            const int end = skipq(i);
            int steps = 0;
            while (c != kEnd && c != end)
            {
                if (c == 0 || out.parent[(size_t)c] != -1 || ++steps > 64) { out.refit_ok = false; break; }
                out.parent[(size_t)c] = (int32_t)i; out.nchild[i]++;
                c = skipq((size_t)c);
            }
        }
        if (nT == 0) out.refit_ok = false;
        // does the walk meet every instance exactly once?
        std::vector<uint8_t> seen((size_t)std::max<int64_t>(nI, 1), 0);
        bool once = nT > 0 && out.refit_ok;
        for (size_t i = 0; once && i < (size_t)nT; i++)
        {
            if (cntq(i) == 0 || !(reachableT < 0 || (int32_t)i < reachableT)) continue;
            const int first = __builtin_bit_cast(int, out.tlas[i].lo.w);
            for (int j = 0; j < cntq(i); j++)
            {
                const int64_t slot = (int64_t)first + j;
                if (slot < 0 || slot >= nTI) { once = false; break; }
                const int64_t ii = s->tlasInstanceIndices[slot];
                if (ii < 0 || ii >= nI || seen[(size_t)ii]++) { once = false; break; }
            }
        }
        for (int64_t ii = 0; once && ii < nI; ii++) if (!seen[(size_t)ii]) once = false;
        out.inst_once = once;
        auto inside = [](const NodeQ& c, const NodeQ& p) {
            // false with a NaN.  The child has to be a regular box (min <= max) too: the slab test reads an inverted box as its
            // mirror image, which these comparisons say nothing about
            return c.lo.x <= c.hi.x && c.lo.y <= c.hi.y && c.lo.z <= c.hi.z &&
                   c.lo.x >= p.lo.x && c.lo.y >= p.lo.y && c.lo.z >= p.lo.z && c.hi.x <= p.hi.x && c.hi.y <= p.hi.y && c.hi.z <= p.hi.z;
        };
        for (size_t i = 1; i < (size_t)nT; i++)
            if ((reachableT < 0 || (int32_t)i < reachableT) && out.parent[i] >= 0 && !inside(out.tlas[i], out.tlas[(size_t)out.parent[i]])) out.nested = false;
    }
    alloc_nodes(nB, out.blas);
    {
        // every instance owns the node range [blasRoot, blasRoot + blasNodeCount); each distinct range is renumbered
        // on its own (root stays first).  Ranges that overlap without being equal cannot all be in walk order:
        // the whole array then keeps the builder's numbering.
        std::vector<std::pair<int64_t, int64_t>> ranges;
        for (int64_t i = 0; i < nI; i++)
            if (s->instances[i].blasNodeCount > 0) ranges.emplace_back((int64_t)s->instances[i].blasRoot, (int64_t)s->instances[i].blasRoot + s->instances[i].blasNodeCount);
        std::sort(ranges.begin(), ranges.end());
        ranges.erase(std::unique(ranges.begin(), ranges.end()), ranges.end());
        bool disjoint = true;
        for (size_t i = 1; i < ranges.size(); i++) if (ranges[i].first < ranges[i - 1].second) disjoint = false;
        const size_t nBq = out.blas.size();
        out.bparent.assign(nBq, -2); out.bnchild.assign(nBq, 0); out.bsubend.assign(nBq, 0); out.borig.assign(nBq, 0); out.bkind.assign(nBq, 0);
        for (size_t j = 0; j < nBq; j++) out.borig[j] = (int32_t)j;
        if (!disjoint)
        {
            perm.resize((size_t)nB);
            for (size_t j = 0; j < perm.size(); j++) perm[j] = (int32_t)j;
            pack_range(s->blasNodes, 0, nB, perm, out.blas);
            out.blas_refit_ok = false;
        }
        else
        {
            // who owns each range: bit 0 a triangle mesh, bit 1 anything else
            std::vector<uint8_t> rangeKind(ranges.size(), 0);
            for (int64_t i = 0; i < nI; i++)
            {
                const hrt_instance& in = s->instances[i];
                if (in.blasNodeCount <= 0) continue;
                const auto it = std::lower_bound(ranges.begin(), ranges.end(), std::make_pair((int64_t)in.blasRoot, (int64_t)in.blasRoot + in.blasNodeCount));
                rangeKind[(size_t)(it - ranges.begin())] |= in.type == HRT_BLAS_TRIMESH ? 1 : (in.type == HRT_BLAS_SPHERESET ? 2 : 4);
            }
            int64_t at = 0;
            for (const auto& r : ranges)
            {
                for (; at < r.first; at++) { perm.assign(1, 0); pack_range(s->blasNodes, at, at + 1, perm, out.blas); }   // owned by no instance: never walked
                const int32_t reach = walk_order(s->blasNodes, r.first, r.second, r.first, perm);
                pack_range(s->blasNodes, r.first, r.second, perm, out.blas);
                at = r.second;
                // maintenance arrays for the BLAS of a triangle mesh (device refit after a vertex update, hrt_bvh.hpp)
                const uint8_t kind = rangeKind[(size_t)(&r - ranges.data())];
                if (kind != 1 && kind != 2) { if (kind != 0) out.blas_refit_ok = false; continue; }               // shared between a mesh and a sphere set, or of an unknown type
                if (reach != (int32_t)(r.second - r.first)) { out.blas_refit_ok = false; continue; }              // unreachable nodes or builder numbering
                for (int64_t k = r.first; k < r.second; k++) out.bkind[(size_t)k] = kind;
                if (kind == 1) out.meshRanges.push_back(r);
                out.max_range[kind] = std::max(out.max_range[kind], (int)(r.second - r.first));
                auto cntq = [&](int64_t i) { return (int)((unsigned)__builtin_bit_cast(int, out.blas[(size_t)i].hi.w) >> 28); };
                auto skipq = [&](int64_t i) { return __builtin_bit_cast(int, out.blas[(size_t)i].hi.w) & kEnd; };
                for (int64_t k = r.first; k < r.second; k++) out.borig[(size_t)(r.first + perm[(size_t)(k - r.first)])] = (int32_t)k;
                out.bparent[(size_t)r.first] = -1;
                for (int64_t i = r.first; i < r.second; i++)
                {
                    const int sk = skipq(i);
                    out.bsubend[(size_t)i] = (int32_t)(sk == kEnd ? r.second : sk);
                    if (cntq(i) > 0) continue;
                    int c = __builtin_bit_cast(int, out.blas[(size_t)i].lo.w) & kEnd;
                    int steps = 0;
                    while (c != kEnd && c != sk)
                    {
                        if (c <= i || c >= r.second || out.bparent[(size_t)c] != -2 || ++steps > 64) { out.blas_refit_ok = false; break; }
                        out.bparent[(size_t)c] = (int32_t)i; out.bnchild[(size_t)i]++;
                        c = skipq(c);
                    }
                }
            }
            for (; at < nB; at++) { perm.assign(1, 0); pack_range(s->blasNodes, at, at + 1, perm, out.blas); }
        }
        for (int64_t i = 0; i < nI; i++)
        {
            const hrt_instance& in = s->instances[i];
            if (in.type == HRT_BLAS_SPHERESET && in.blasNodeCount > 0) out.sphereInst.push_back((int32_t)i);
            if (in.type != HRT_BLAS_TRIMESH || in.blasNodeCount <= 0) continue;
            out.meshInst.push_back((int32_t)i);
            // region of triPrimIdx the leaves of this BLAS point into (the builder appends it behind the item list, Scene.cs:439-440)
            int64_t lo = INT64_MAX, hi = -1, sum = 0;
            for (int64_t k = in.blasRoot; k < (int64_t)in.blasRoot + in.blasNodeCount; k++)
            {
                const hrt_bvh_node& b = s->blasNodes[k];
                if (b.count <= 0) continue;
                lo = std::min<int64_t>(lo, b.first); hi = std::max<int64_t>(hi, (int64_t)b.first + b.count); sum += b.count;
            }
            const int64_t n = in.primIndexCount;
            MeshJob J; J.inst = (int)i; J.root = in.blasRoot; J.nodeCap = in.blasNodeCount; J.leafBase = (int)lo; J.n = (int)n; J.itemFirst = in.primIndexFirst;
            const bool items_ok = n > 0 && in.primIndexFirst >= 0 && (int64_t)in.primIndexFirst + n <= nTP;
            const bool region_ok = hi - lo == n && sum == n && (lo >= (int64_t)in.primIndexFirst + n || hi <= in.primIndexFirst);
            if (!items_ok || !region_ok || 2 * ((n + 13) / 14) - 1 > in.blasNodeCount) out.blas_rebuild_ok = false;
            out.meshJobs.push_back(J);
        }
        {   // two meshes must not share a node range or a leaf region
            std::vector<std::pair<int, int>> byRoot, byLeaf;
            for (const MeshJob& J : out.meshJobs) { byRoot.emplace_back(J.root, J.nodeCap); byLeaf.emplace_back(J.leafBase, J.n); }
            std::sort(byRoot.begin(), byRoot.end()); std::sort(byLeaf.begin(), byLeaf.end());
            for (size_t k = 1; k < byRoot.size(); k++)
                if (byRoot[k].first < byRoot[k - 1].first + byRoot[k - 1].second || byLeaf[k].first < byLeaf[k - 1].first + byLeaf[k - 1].second) out.blas_rebuild_ok = false;
        }
    }
    if (nT == 0)
    {   // reference semantics of the zeroed 1-element TLAS: node 0 has count 0, left 0 -> loops forever on a hit;
        // its bounds are all zero so only rays through the origin would.  We end the walk instead.
    }
    out.finst.resize((size_t)std::max<int64_t>(nTI, 1));
    std::memset(out.finst.data(), 0, out.finst.size() * sizeof(FInst));
    for (int64_t i = 0; i < nTI; i++)
    {
        int ii = s->tlasInstanceIndices[i];
        const hrt_instance& in = s->instances[ii];
        const bool ident = is_identity(in.objectToWorld) && is_identity(in.worldToObject) && in.uniformScale == 1.0f;
        const bool sph = in.type == HRT_BLAS_SPHERESET;
        FInst f;
        bool fast = false;
        if (sph && ident && in.blasNodeCount >= 1)
        {
            const hrt_bvh_node& root = s->blasNodes[in.blasRoot];
            int64_t end = (int64_t)in.blasRoot + in.blasNodeCount;
            if (root.count == 1 && (root.skipIndex == -1 || root.skipIndex >= end))
            {
                int sid = s->spherePrimIdx[root.first];
                const hrt_sphere& sp = s->spheres[sid];
                f.a = mkf4(root.boundsMin.X, root.boundsMin.Y, root.boundsMin.Z, bits_f(FI_FAST_SPHERE | FI_IDENTITY | FI_SPHERESET));
                f.b = mkf4(root.boundsMax.X, root.boundsMax.Y, root.boundsMax.Z, bits_f(sid));
                f.c = mkf4(sp.center.X, sp.center.Y, sp.center.Z, sp.radius);
                fast = true;
            }
        }
        if (!fast)
        {
            out.feat |= 1;
            float scale = in.uniformScale > 0.f ? in.uniformScale : 1.f;
            f.a = mkf4(0.f, 0.f, 0.f, bits_f((ident ? FI_IDENTITY : 0) | (sph ? FI_SPHERESET : 0)));
            f.b = mkf4(0.f, 0.f, 0.f, bits_f(ii));
            f.c = mkf4(bits_f(in.blasRoot), bits_f(in.blasRoot + in.blasNodeCount), scale, 0.f);
        }
        out.finst[(size_t)i] = f;
        // BOTH corners of the own box: a negative radius inverts it (max < min), and the slab test reads that as the mirror image
        auto within = [](float v, float lo, float hi) { return v >= lo && v <= hi; };      // false with a NaN
        if (fast && !(within(f.a.x, in.worldBoundsMin.X, in.worldBoundsMax.X) && within(f.b.x, in.worldBoundsMin.X, in.worldBoundsMax.X) &&
                      within(f.a.y, in.worldBoundsMin.Y, in.worldBoundsMax.Y) && within(f.b.y, in.worldBoundsMin.Y, in.worldBoundsMax.Y) &&
                      within(f.a.z, in.worldBoundsMin.Z, in.worldBoundsMax.Z) && within(f.b.z, in.worldBoundsMin.Z, in.worldBoundsMax.Z))) out.own_in_world = false;
    }
    out.ftri.resize((size_t)std::max<int64_t>(nTP, 1));
    std::memset(out.ftri.data(), 0, out.ftri.size() * sizeof(FTri));
    const int64_t texLen = nTxI > 0 ? nTxI : 1;
    for (int64_t j = 0; j < nTP; j++)
    {
        int ti = s->triPrimIdx[j];
        const hrt_mesh_tri& t = s->meshTris[ti];
        const hrt_float3 &a = s->meshPositions[t.i0], &b = s->meshPositions[t.i1], &c = s->meshPositions[t.i2];
        int mi = s->triMatIndex[ti];
        static const hrt_material kZeroMaterial = {};
        const hrt_material& m = nM > 0 ? s->materials[mi] : kZeroMaterial;
        bool dmap = m.HasDiffuseMap != 0 && m.DiffuseTexIndex >= 0 && m.DiffuseTexIndex < texLen;
        bool amap = m.HasAlphaMap != 0 && m.AlphaTexIndex >= 0 && m.AlphaTexIndex < texLen;
        bool rejects_opaque = 1.0f < m.AlphaCutoff;
        int fl = ((dmap || amap || rejects_opaque) ? FT_TEXTURED : 0) | (m.TwoSided != 0 ? FT_TWOSIDED : 0);
        if (amap || rejects_opaque) out.feat |= 2;          // the walk itself must evaluate alpha (diffuse-only maps are resolved after it)
        FTri& o = out.ftri[(size_t)j];
        o.v0 = mkf4(a.X, a.Y, a.Z, bits_f(ti));
        o.v1 = mkf4(b.X, b.Y, b.Z, bits_f(mi));
        o.v2 = mkf4(c.X, c.Y, c.Z, bits_f(fl));
    }
    // ---- sphere-instance scenes: instance records inlined into the TLAS node stream (hrt_walker.hpp).  A leaf is followed by
    // one record per instance holding the box of its one-node BLAS; the walker treats them as nodes (count field 15), so the
    // instance box tests ride the node steps and their lookahead instead of costing a leaf step each.
    out.tlasX.assign(1, NodeQ{});
    bool leavesFit = true;                   // count code 15 marks an instance record in this stream: a leaf of 15 instances cannot be told from one
    for (int64_t i = 0; i < nT; i++) if (((unsigned)__builtin_bit_cast(int, out.tlas[(size_t)i].hi.w) >> 28) > 14u) leavesFit = false;
    if (out.ok && out.feat == 0 && reachableT > 0 && nT + nTI < kEnd && leavesFit && !HRT_ENV("HRT_NO_INLINE_INSTANCES"))
    {
        std::vector<int32_t> nidx((size_t)nT);
        int32_t at = 0;
        auto cnt_of = [&](int64_t i) { return (int)((unsigned)__builtin_bit_cast(int, out.tlas[(size_t)i].hi.w) >> 28); };
        for (int64_t i = 0; i < nT; i++) { nidx[(size_t)i] = at; at += 1 + cnt_of(i); }
        auto remap = [&](int v) { return v == kEnd ? kEnd : (int)nidx[(size_t)v]; };
        out.tlasX.assign((size_t)at, NodeQ{});
        for (int64_t i = 0; i < nT; i++)
        {
            const NodeQ& q = out.tlas[(size_t)i];
            const int c = cnt_of(i), link = __builtin_bit_cast(int, q.lo.w), sk = remap(__builtin_bit_cast(int, q.hi.w) & kEnd);
            NodeQ& o = out.tlasX[(size_t)nidx[(size_t)i]];
            o = q;
            o.hi.w = bits_f(sk | (int)((unsigned)c << 28));
            if (c == 0) { o.lo.w = bits_f(remap(link & kEnd)); continue; }
            for (int j = 0; j < c; j++)
            {
                const FInst& f = out.finst[(size_t)(link + j)];
                NodeQ& r = out.tlasX[(size_t)(nidx[(size_t)i] + 1 + j)];
                r.lo = mkf4(f.a.x, f.a.y, f.a.z, bits_f(link + j));
                const int next = (j + 1 < c) ? nidx[(size_t)i] + 2 + j : sk;
                r.hi = mkf4(f.b.x, f.b.y, f.b.z, bits_f(next | (int)(15u << 28)));
            }
        }
        out.n_tlasX = at;
    }

    // fast-sphere instances: own box inside the box of the leaf that lists them
    for (int64_t i = 0; out.nested && i < nT; i++)
    {
        const NodeQ& q = out.tlas[(size_t)i];
        const int cnt = (int)((unsigned)__builtin_bit_cast(int, q.hi.w) >> 28), first = __builtin_bit_cast(int, q.lo.w);
        if (cnt == 0 || !(reachableT < 0 || (int32_t)i < reachableT)) continue;
        for (int j = 0; j < cnt; j++)
        {
            if ((int64_t)first + j < 0 || (int64_t)first + j >= nTI) { out.nested = false; break; }
            const FInst& f = out.finst[(size_t)(first + j)];
            if (!(__builtin_bit_cast(int, f.a.w) & FI_FAST_SPHERE)) continue;
            // both corners of the own box (a negative radius inverts it, and the slab test reads that as the mirror image)
            auto within = [](float v, float lo, float hi) { return v >= lo && v <= hi; };
            if (!(within(f.a.x, q.lo.x, q.hi.x) && within(f.b.x, q.lo.x, q.hi.x) && within(f.a.y, q.lo.y, q.hi.y) && within(f.b.y, q.lo.y, q.hi.y) &&
                  within(f.a.z, q.lo.z, q.hi.z) && within(f.b.z, q.lo.z, q.hi.z))) { out.nested = false; break; }
        }
    }
    if (!out.nested) out.inst_once = false;
    // TracerFlat: the reachable TLAS leaves in walk order, for scenes made of fast-sphere instances only
    out.flat.assign(1, NodeQ{});
    if (out.ok && out.feat == 0 && reachableT > 0 && out.nested)
    {
        std::vector<NodeQ> leaves;
        for (int32_t i = 0; i < reachableT; i++)
            if (((unsigned)__builtin_bit_cast(int, out.tlas[(size_t)i].hi.w) >> 28) != 0) leaves.push_back(out.tlas[(size_t)i]);
        if (!leaves.empty() && (int)leaves.size() <= kFlatMaxLeaves) out.flat = leaves;
        else out.flat.clear(), out.flat.assign(1, NodeQ{});
        out.n_flat = (!leaves.empty() && (int)leaves.size() <= kFlatMaxLeaves) ? (int)leaves.size() : 0;
    }
    return "";
}

// copies the strips `owner` renders (rows [row_begin,row_end), strips s % strip_n == strip_i) of one per-pixel array
// from src to dst (same global indexing on both sides) on `stream`: device -> host gather, or device -> device exchange
template <class T>
int copy_strips(hrt_ctx* c, const DeviceState& owner, T* dst, const T* src, int width, hipMemcpyKind kind, hipStream_t stream)
{
    if (!dst || owner.n_strips == 0) return HRT_OK;
    const size_t rowElems = (size_t)width;
    if (owner.strip_n == 1)
    {   // one contiguous row block
        size_t off = (size_t)owner.row_begin * rowElems;
        size_t cnt = (size_t)(owner.row_end - owner.row_begin) * rowElems;
        HIPCHK(c, hipMemcpyAsync(dst + off, src + off, cnt * sizeof(T), kind, stream));
        return HRT_OK;
    }
    // interleaved 8-row strips: one strided 2-D copy for the full strips, one plain copy for a ragged last strip
    int lastStrip = owner.strip_i + (owner.n_strips - 1) * owner.strip_n;
    int lastRows = std::min(8, (owner.row_end - owner.row_begin) - lastStrip * 8);
    int fullStrips = lastRows == 8 ? owner.n_strips : owner.n_strips - 1;
    size_t first = ((size_t)owner.row_begin + (size_t)owner.strip_i * 8) * rowElems;
    if (fullStrips > 0)
        HIPCHK(c, hipMemcpy2DAsync(dst + first, (size_t)8 * rowElems * owner.strip_n * sizeof(T), src + first, (size_t)8 * rowElems * owner.strip_n * sizeof(T),
                                   (size_t)8 * rowElems * sizeof(T), (size_t)fullStrips, kind, stream));
    if (lastRows < 8 && lastRows > 0)
    {
        size_t off = ((size_t)owner.row_begin + (size_t)lastStrip * 8) * rowElems;
        HIPCHK(c, hipMemcpyAsync(dst + off, src + off, (size_t)lastRows * rowElems * sizeof(T), kind, stream));
    }
    return HRT_OK;
}
template <class T>
int gather_rows(hrt_ctx* c, DeviceState& d, T* host, const T* devp, int width)
{
    return copy_strips(c, d, host, devp, width, hipMemcpyDeviceToHost, d.stream);
}

// ---------------------------------------------------------------------------------------
// The path-trace launch of one device, either as the one-pixel-per-lane megakernel or as the
// streamed pipeline of hrt_wavefront.hpp (default).
// ---------------------------------------------------------------------------------------
// Organisation of the path-trace launch when the caller does not force one: scenes whose whole BVH
// is a few cache lines (the reference's default scene, BASELINE config 2) spend their time in ReSTIR
// arithmetic, not in the walk -- streaming path state through HBM only adds traffic there (measured:
// 2.8 ms fused vs 5.3 ms streamed on config 2; 158 ms vs 33 ms on config 3).
constexpr long long kSmallSceneNodes = 256;
constexpr long long kWfMaxPaths = 1ll << 25;      // paths resident per sample batch (320 B of workspace each)

// workspace of batch lane `lane` (0 / 1); growing one drains the device first (frames of earlier calls may still use it)
int ensure_workspace(hrt_ctx* c, DeviceState& d, int lane, long long cap, int nOrd, int nRanges, int maxDepth, WfBuffers& W, bool treelets = false)
{
    const size_t planes = 2 * V_PLANES + R_PLANES + S_PLANES + 3 + G_PLANES;
    const size_t bytes = (size_t)planes * (size_t)cap * sizeof(float);
    const size_t ints = (size_t)(2 * maxDepth + 2) * (size_t)nRanges + (size_t)maxDepth * 16;
    if (bytes > d.wf_bytes[lane])
    {
        if (d.wf_mem[lane]) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(d.wf_mem[lane]); d.wf_mem[lane] = nullptr; d.wf_bytes[lane] = 0; }
        void* v = nullptr;
        HIPCHK(c, hipMalloc(&v, bytes));
        d.wf_mem[lane] = (float*)v; d.wf_bytes[lane] = bytes;
    }
    if (ints > d.wf_cnt_ints[lane])
    {
        if (d.wf_cnt[lane]) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(d.wf_cnt[lane]); d.wf_cnt[lane] = nullptr; d.wf_cnt_ints[lane] = 0; }
        void* v = nullptr;
        HIPCHK(c, hipMalloc(&v, ints * sizeof(int)));
        d.wf_cnt[lane] = (int*)v; d.wf_cnt_ints[lane] = ints;
    }
    if ((size_t)3 * (size_t)nOrd > d.wf_accum_floats)
    {
        if (d.wf_accum) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(d.wf_accum); d.wf_accum = nullptr; d.wf_accum_floats = 0; }
        void* v = nullptr;
        HIPCHK(c, hipMalloc(&v, (size_t)3 * (size_t)nOrd * sizeof(float)));
        d.wf_accum = (float*)v; d.wf_accum_floats = (size_t)3 * (size_t)nOrd;
    }
    if (d.tl_ok && treelets)
    {   // queues of the treelet walker (only for frames that ask for it: HRT_FLAG_TREELETS): per walk kind key / state / binned indices over the path slots, and the per-treelet counters
        const size_t nTl = (size_t)d.dtl.nTl;
        const size_t ints = ((nTl + 32 + nTl + 1 + nTl) + 63) & ~(size_t)63;
        const size_t per0 = (size_t)cap * (4 + 16 + 4) + ints * 4, per1 = (size_t)cap * (4 + 32 + 4) + ints * 4;
        const size_t need = ((per0 + 255) & ~(size_t)255) + per1;
        if (need > d.tlq_bytes[lane])
        {
            if (d.tlq_mem[lane]) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(d.tlq_mem[lane]); d.tlq_mem[lane] = nullptr; d.tlq_bytes[lane] = 0; }
            void* v = nullptr;
            HIPCHK(c, hipMalloc(&v, need));
            d.tlq_mem[lane] = v; d.tlq_bytes[lane] = need;
        }
    }
    float* m = d.wf_mem[lane];
    auto take = [&](int nplanes, long long stride) { Planes pl; pl.base = m; pl.stride = stride; m += (size_t)nplanes * (size_t)stride; return pl; };
    W.A = take(V_PLANES, cap); W.B = take(V_PLANES, cap); W.R = take(R_PLANES, cap); W.SQ = take(S_PLANES, cap);
    static_assert(V_POS == 0 && V_LI >= R_PLANES && V_PID >= R_PLANES && R_PLANES >= S_PLANES, "the second request set aliases the vertex planes below V_LI");
    W.Rn = W.A; W.SQn = W.B; W.pingpong = 0;
    W.sampleLi = take(3, cap); W.stage = take(G_PLANES, cap);
    W.accum.base = d.wf_accum; W.accum.stride = nOrd;
    W.cntA = d.wf_cnt[lane]; W.cntS = d.wf_cnt[lane] + (size_t)(maxDepth + 1) * (size_t)nRanges;
    W.grab = d.wf_cnt[lane] + (size_t)(2 * maxDepth + 2) * (size_t)nRanges;
    W.nRanges = nRanges;
    return HRT_OK;
}


// queues of the treelet walker for one batch lane and walk kind (0 shadow, 1 closest), carved from DeviceState::tlq_mem
TlQueues tl_queues(const DeviceState& d, int lane, int kind, long long cap)
{
    const size_t nTl = (size_t)d.dtl.nTl;
    const size_t ints = ((nTl + 32 + nTl + 1 + nTl) + 63) & ~(size_t)63;
    const size_t per0 = (size_t)cap * (4 + 16 + 4) + ints * 4;
    char* p = (char*)d.tlq_mem[lane] + (kind ? ((per0 + 255) & ~(size_t)255) : 0);
    TlQueues Q;
    Q.state = (float4*)p; p += (size_t)cap * (kind ? 32 : 16);
    Q.key = (int*)p; p += (size_t)cap * 4;
    Q.sorted = (int*)p; p += (size_t)cap * 4;
    Q.hist = (int*)p; Q.misc = Q.hist + nTl; Q.offs = Q.misc + 32; Q.curs = Q.offs + nTl + 1;
    return Q;
}

#ifndef HRT_TL_ROUNDS
#define HRT_TL_ROUNDS 3
#endif
// One walk launch of the streamed pipeline through the treelet walker: fresh rays, kTlRounds rounds over the binned rays, clean-up.
template <int F, bool ANY, bool EXISTS, int LT>
int launch_tl_walk(hrt_ctx* c, DeviceState& d, const TracerPackedT<F>& tr, const WfBuffers& W, int lane, long long cap, int depth, hipStream_t st, dim3 gridW, dim3 gridR)
{
    const DTreelets& T = d.dtl;
    const TlQueues Q = tl_queues(d, lane, ANY ? 0 : 1, cap);
    const int histBins = T.nTl <= kTlHistLds ? T.nTl : 0;
    const size_t lds0 = tl_shared_bytes(0, T.redLds, histBins), lds1 = tl_shared_bytes(T.tlBytesMax, T.redLds, histBins);
    static const int rounds = HRT_ENV("HRT_TL_ROUNDS") ? std::max(0, atoi(HRT_ENV("HRT_TL_ROUNDS"))) : HRT_TL_ROUNDS;
    if (lds1 > 65536) HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&hrt_tl_walk_kernel<F, ANY, EXISTS, LT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    const dim3 block(256), grid1((unsigned)(d.n_cu * 8));
    HIPCHK(c, hipMemsetAsync(Q.hist, 0, ((size_t)T.nTl + 32) * sizeof(int), st));
    const int* cnt = (ANY ? W.cntS : W.cntA) + (size_t)depth * W.nRanges;
    hipLaunchKernelGGL((hrt_tl_walk_kernel<F, ANY, EXISTS, LT, 0>), gridW, block, lds0, st, tr, T, Q, W, depth, histBins, 0, 0);
    for (int r = 0; r < rounds; r++)
    {
        hipLaunchKernelGGL(hrt_tl_scan_kernel, dim3(1), dim3(1024), 0, st, Q, T.nTl);
        hipLaunchKernelGGL(hrt_tl_scatter_kernel, gridR, block, 0, st, Q, T.nTl, cnt, W.nRanges);
        hipLaunchKernelGGL((hrt_tl_walk_kernel<F, ANY, EXISTS, LT, 1>), grid1, block, lds1, st, tr, T, Q, W, depth, histBins, T.tlBytesMax, r + 1);
    }
    hipLaunchKernelGGL((hrt_tl_walk_kernel<F, ANY, EXISTS, LT, 2>), gridW, block, lds0, st, tr, T, Q, W, depth, histBins, 0, 0);
    HIPCHK(c, hipGetLastError());
    return HRT_OK;
}

template <class TR> struct PackedFeat { static constexpr int value = -1; };
template <int F> struct PackedFeat<TracerPackedT<F>> { static constexpr int value = F; };

template <class TR>
int run_path_stage(hrt_ctx* c, DeviceState& d, const TR& tr, const FrameK& k, const TileMap& tm, int width,
                   const DReservoir& resPrev, const DReservoir& resCur, long long nPix, bool count, bool mega, bool treelets = false)
{
    unsigned long long* cnt1 = d.counters + 10;
    if (tm.nTiles <= 0) return HRT_OK;
    if (mega || k.maxDepth > 64)
    {
        const dim3 grid(tm.nTiles), block(64 * tm.wpb);
        // frames without ReSTIR reuse: the leaf-sweep tracer's kernels with the import code compiled out (hrt_device.hpp, REUSE)
        const bool noReuse = std::is_same<TR, TracerFlat>::value && !count && k.enableTemporal == 0 && k.enableSpatial == 0;
        // sample groups when the tile gives the machine less than ~5 rounds of waves
        static const int splitEnv = HRT_ENV("HRT_SPLIT") ? atoi(HRT_ENV("HRT_SPLIT")) : -1;          // A/B knob: 0 never, n > 0 force n groups
        const int sppN = k.spp > 1 ? k.spp : 1;
        const long long waves = (long long)tm.nTiles * tm.wpb, slots = (long long)d.n_cu * 4 * HRT_PT_WAVES;
        int nGroups = 1;
        if (!count && k.maxDepth <= 64 && sppN > 1 && waves > 0)
        {
            if (splitEnv > 0) nGroups = std::min(splitEnv, sppN);
            else if (splitEnv < 0 && waves < 5 * slots) nGroups = (int)std::min<long long>(std::min(sppN, 8), (16 * slots + waves - 1) / waves);   // config 2 over N = 2 / 4 / 8 ranks: 4 groups each (1.111 -> 1.079, 0.593 -> 0.555, 0.296 ms)
        }
        if (nGroups > 1)
        {
            const int perGroup = (sppN + nGroups - 1) / nGroups;
            nGroups = (sppN + perGroup - 1) / perGroup;
            // scratch planes over the lanes of THIS launch's tiles (a rank's share of the frame), not over the image
            const size_t nLocal = (size_t)tm.nTiles * 64 * (size_t)tm.wpb;
            const size_t need = ((size_t)sppN * 3 + (size_t)nGroups * 12) * nLocal;
            if (need > d.split_floats)
            {
                if (d.split_mem) { HIPCHK(c, hipStreamSynchronize(d.stream)); (void)hipFree(d.split_mem); d.split_mem = nullptr; d.split_floats = 0; }
                void* v = nullptr;
                HIPCHK(c, hipMalloc(&v, need * sizeof(float)));
                d.split_mem = (float*)v; d.split_floats = need;
            }
            hrt_float3* li = (hrt_float3*)d.split_mem;
            float* stage = d.split_mem + (size_t)sppN * 3 * nLocal;
            if (noReuse) { if constexpr (std::is_same<TR, TracerFlat>::value) hipLaunchKernelGGL((hrt_path_trace_split_kernel<TR, false>), dim3(tm.nTiles * nGroups), block, 0, d.stream, tr, k, d.gb, d.fb, resPrev, resCur, nPix, tm, li, stage, nGroups, perGroup); }
            else hipLaunchKernelGGL((hrt_path_trace_split_kernel<TR>), dim3(tm.nTiles * nGroups), block, 0, d.stream, tr, k, d.gb, d.fb, resPrev, resCur, nPix, tm, li, stage, nGroups, perGroup);
            hipLaunchKernelGGL(hrt_split_resolve_kernel, grid, block, 0, d.stream, k, d.gb, d.fb, resCur, nPix, tm, (const hrt_float3*)li, (const float*)stage, nGroups);
            HIPCHK(c, hipGetLastError());
            return HRT_OK;
        }
        if (count) hipLaunchKernelGGL((hrt_path_trace_kernel<TR, true>), grid, block, 0, d.stream, tr, k, d.gb, d.fb, resPrev, resCur, nPix, tm, cnt1);
        else if (noReuse) { if constexpr (std::is_same<TR, TracerFlat>::value) hipLaunchKernelGGL((hrt_path_trace_kernel<TR, false, false>), grid, block, 0, d.stream, tr, k, d.gb, d.fb, resPrev, resCur, nPix, tm, cnt1); }
        else       hipLaunchKernelGGL((hrt_path_trace_kernel<TR, false>), grid, block, 0, d.stream, tr, k, d.gb, d.fb, resPrev, resCur, nPix, tm, cnt1);
        HIPCHK(c, hipGetLastError());
        return HRT_OK;
    }
    WfGeom g;
    g.tilesX8 = (width + 7) / 8;
    g.nOrd = g.tilesX8 * d.n_strips * 64;
    const int spp = k.spp > 1 ? k.spp : 1;
    long long maxPaths = kWfMaxPaths;
    if (c->max_resident_paths > 0) maxPaths = c->max_resident_paths;              // hrt_set_workspace_limit
    long long sb = maxPaths / g.nOrd;
    if (sb < 1) sb = 1;
    if (sb > spp) sb = spp;
    // Two sample batches in flight: the walks of one are latency-bound and leave the vector units idle most of the time, the
    // shade / finish / resolve kernels of the other fill them (measured first as two processes sharing the card: configs 4 / 5
    // -11 % / -7 %).  A frame that fits one batch is cut into two halves; only the ordered steps -- the per-pixel sample sum
    // and the last-writer reservoir in wf_resolve -- are chained by events, in batch order.
#ifdef HRT_ONE_BATCH_LANE          // A/B
    const int nLanes = 1;
#else
    // ... a frame that fits one batch is cut in two only while each half still fills the machine (measured: halves of 16.6 M paths
    // config 3 -5.5 %, config 4 +-0; halves of 8.3 M paths config 5 +13 %)
    constexpr long long kMinHalfBatchPaths = 12000000;
    const bool severalBatches = sb < spp;
    const bool halves = !severalBatches && spp >= 2 && (long long)((spp + 1) / 2) * g.nOrd >= kMinHalfBatchPaths;
    const int nBatchesNatural = (int)((spp + sb - 1) / sb);
    const int nLanes = severalBatches ? std::min(kBatchLanes, nBatchesNatural) : (halves ? std::min(kBatchLanes, 2) : 1);
#endif
    if (nLanes >= 2 && !severalBatches && sb > (spp + 1) / 2) sb = (spp + 1) / 2;
    const long long batchPaths = sb * (long long)g.nOrd;
    const int nRanges = (int)((batchPaths + kRange - 1) / kRange);
    const long long cap = (long long)nRanges * kRange;
    WfBuffers Wl[kMaxLanes];
    for (int j = 0; j < nLanes; j++) { int rc = ensure_workspace(c, d, j, cap, g.nOrd, nRanges, k.maxDepth, Wl[j], treelets); if (rc != HRT_OK) return rc; }
    hipStream_t laneMain[kMaxLanes], laneSide[kMaxLanes];
    laneMain[0] = d.stream; laneSide[0] = d.stream2;
    for (int j = 1; j < kMaxLanes; j++) { laneMain[j] = d.laneStream[j][0]; laneSide[j] = d.laneStream[j][1]; }
    if (nLanes >= 2)
    {   // the other lanes start behind everything enqueued so far (this frame's primary launch, the previous frame)
        HIPCHK(c, hipEventRecord(d.evStage, d.stream));
        for (int j = 1; j < nLanes; j++) HIPCHK(c, hipStreamWaitEvent(laneMain[j], d.evStage, 0));
    }
    const dim3 block(256), gridR((nRanges + 3) / 4), gridP((g.nOrd + 255) / 256);
    // walk launches are persistent: enough workgroups to fill every wave slot, each wave pulls ranges until none is left
    const dim3 gridW((unsigned)std::min<long long>((nRanges + 3) / 4, (long long)d.n_cu * (nLanes >= 2 ? kWalkBlocksPerCU2 : kWalkBlocksPerCU)));
    static const bool forkShadow = HRT_ENV("HRT_NO_FORK") == nullptr;       // A/B knob
    static const bool forkStatic = HRT_ENV("HRT_FORK_STATIC") != nullptr;   // A/B knob
    // finish of a bounce and shade of the next one as ONE kernel (the vertex never round-trips through its 22 planes): sphere-instance scenes
    // (config 3: path stage -3.6 %, HBM traffic 13.6 -> 11.7 GB per frame).  Triangle scenes keep the two kernels: their walks leave the vector
    // units to the other sample batch's shade / finish kernels, and the fused kernel (111 registers through the fetch-bound half) fills them
    // worse (config 5 +3 %, config 4 +-0.5 %: profiles/EXPERIMENTS.md)
    static const int fuseEnv = HRT_ENV("HRT_FUSE") ? atoi(HRT_ENV("HRT_FUSE")) : -1;       // A/B knob
    const bool fuse = PackedFeat<TR>::value >= 0 && (fuseEnv >= 0 ? fuseEnv != 0 : PackedFeat<TR>::value == 0);
    for (int j = 0; j < nLanes; j++) Wl[j].pingpong = fuse ? 1 : 0;
    int batch = 0;
    for (int b0 = 0; b0 < spp; b0 += (int)sb, batch++)
    {
        const int lane = batch % nLanes;
        const WfBuffers& W = Wl[lane];
        const hipStream_t sMain = laneMain[lane], sSide = laneSide[lane];
        if (k.maxDepth > 0) HIPCHK(c, hipMemsetAsync(W.grab, 0, (size_t)k.maxDepth * 16 * sizeof(int), sMain));
        g.batchStart = b0;
        g.batchCount = (int)std::min<long long>(sb, spp - b0);
        g.lastBatch = (b0 + g.batchCount >= spp) ? 1 : 0;
        for (int depth = 0; depth < k.maxDepth; depth++)
        {
            const int vsel = depth & 1;
#ifdef HRT_NO_EXISTS            // A/B
            const bool lastBounce = false;
#else
            const bool lastBounce = depth + 1 >= k.maxDepth;        // its closest-hit walk only decides hit or miss
#endif
            const int chained = (PackedFeat<TR>::value > 0 || HRT_CHAIN_FEAT0) ? 1 : 0;
            if (depth == 0)
            {
                if (count) hipLaunchKernelGGL((hrt_wf_shade_kernel<true, true>), gridR, block, 0, sMain, k, g, d.gb, resPrev, nPix, W, vsel, depth, cnt1);
                else       hipLaunchKernelGGL((hrt_wf_shade_kernel<false, true>), gridR, block, 0, sMain, k, g, d.gb, resPrev, nPix, W, vsel, depth, cnt1);
            }
            else if (fuse) { }                              // the vertices of this bounce were shaded by the fused finish + shade kernel of the last one
            else if (count) hipLaunchKernelGGL((hrt_wf_shade_kernel<true, false>), gridR, block, 0, sMain, k, g, d.gb, resPrev, nPix, W, vsel, depth, cnt1);
            else            hipLaunchKernelGGL((hrt_wf_shade_kernel<false, false>), gridR, block, 0, sMain, k, g, d.gb, resPrev, nPix, W, vsel, depth, cnt1);
            if constexpr (PackedFeat<TR>::value >= 0)
            {   // packed layout: persistent-wave walks + finish
                constexpr int F = PackedFeat<TR>::value;
                // production walks.  Boolean queries (shadow rays, the last bounce's hit-or-miss) of fast-sphere scenes walk the
                // device-built tree over the same instances when one was made at upload (hrt_walker.hpp, ALT)
                // scenes with big triangle meshes: the treelet-queued walker (hrt_walker_tl.hpp)
                const bool useTl = F != 0 && treelets && d.tl_ok && !count;
                int tlRc = HRT_OK;
                auto launch_shadow = [&](hipStream_t st) {
                    if constexpr (F != 0)
                        if (useTl)
                        {
                            const int rcT = d.dpacked.leafTris == 3 ? launch_tl_walk<F, true, false, 3>(c, d, tr, W, lane, cap, depth, st, gridW, gridR)
                                                                    : launch_tl_walk<F, true, false, 2>(c, d, tr, W, lane, cap, depth, st, gridW, gridR);
                            if (rcT != HRT_OK) tlRc = rcT;
                            return;
                        }
                    if constexpr (F == 0)
                        if (d.any_ok)
                        {
                            TR trAny = tr; trAny.P = d.dpackedAny;
                            hipLaunchKernelGGL((hrt_wf_walk_shadow_kernel<F, false, true>), chained ? gridW : gridR, block, 0, st, trAny, tr, W, vsel, depth, chained, cnt1);
                            return;
                        }
                    if (F != 0 && d.dpacked.leafTris == 3)
                        hipLaunchKernelGGL((hrt_wf_walk_shadow_kernel<F, false, false, (F != 0 ? 3 : 2)>), chained ? gridW : gridR, block, 0, st, tr, tr, W, vsel, depth, chained, cnt1);
                    else
                        hipLaunchKernelGGL((hrt_wf_walk_shadow_kernel<F, false>), chained ? gridW : gridR, block, 0, st, tr, tr, W, vsel, depth, chained, cnt1);
                };
                // with a second tree every production walk of the frame uses it, and so does the shading of their winners (leaf
                // slots are the second tree's)
                bool second = false;
                TR trFin = tr;
                if constexpr (F == 0) { second = d.any_ok && !count; if (second) trFin.P = d.dpackedAny; }
#ifdef HRT_NO_ALT_CLOSEST        // A/B: closest-hit walks of the inner bounces on the uploaded tree
                const bool secondClosest = false;
                trFin = tr;
#else
                const bool secondClosest = second;
#endif
                auto launch_closest = [&](hipStream_t st) {
                    if constexpr (F != 0)
                        if (useTl)
                        {
                            const bool lt3 = d.dpacked.leafTris == 3;
                            const int rcT = lastBounce ? (lt3 ? launch_tl_walk<F, false, true, 3>(c, d, tr, W, lane, cap, depth, st, gridW, gridR)
                                                              : launch_tl_walk<F, false, true, 2>(c, d, tr, W, lane, cap, depth, st, gridW, gridR))
                                                       : (lt3 ? launch_tl_walk<F, false, false, 3>(c, d, tr, W, lane, cap, depth, st, gridW, gridR)
                                                              : launch_tl_walk<F, false, false, 2>(c, d, tr, W, lane, cap, depth, st, gridW, gridR));
                            if (rcT != HRT_OK) tlRc = rcT;
                            return;
                        }
                    if constexpr (F == 0)
                        if (lastBounce ? d.any_ok : secondClosest)
                        {
                            TR trAny = tr; trAny.P = d.dpackedAny;
                            if (lastBounce) hipLaunchKernelGGL((hrt_wf_walk_closest_kernel<F, false, true, true>), chained ? gridW : gridR, block, 0, st, trAny, tr, W, depth, chained, cnt1);
                            else            hipLaunchKernelGGL((hrt_wf_walk_closest_kernel<F, false, false, true>), chained ? gridW : gridR, block, 0, st, trAny, tr, W, depth, chained, cnt1);
                            return;
                        }
                    const bool lt3 = F != 0 && d.dpacked.leafTris == 3;
                    if (!lastBounce)
                    {
                        if (lt3) hipLaunchKernelGGL((hrt_wf_walk_closest_kernel<F, false, false, false, (F != 0 ? 3 : 2)>), chained ? gridW : gridR, block, 0, st, tr, tr, W, depth, chained, cnt1);
                        else     hipLaunchKernelGGL((hrt_wf_walk_closest_kernel<F, false>), chained ? gridW : gridR, block, 0, st, tr, tr, W, depth, chained, cnt1);
                        return;
                    }
                    if (lt3) hipLaunchKernelGGL((hrt_wf_walk_closest_kernel<F, false, true, false, (F != 0 ? 3 : 2)>), chained ? gridW : gridR, block, 0, st, tr, tr, W, depth, chained, cnt1);
                    else     hipLaunchKernelGGL((hrt_wf_walk_closest_kernel<F, false, true>), chained ? gridW : gridR, block, 0, st, tr, tr, W, depth, chained, cnt1);
                };
                // the winners' shading: on the last bounce the paths end (wf_finish); before it the next vertex is shaded in the same kernel
                const bool finalBounce = depth + 1 >= k.maxDepth;
                auto launch_finish = [&](const TR& trF) {
                    if (count)
                    {
                        if (finalBounce || !fuse) hipLaunchKernelGGL((hrt_wf_finish_kernel<F, true>), gridR, block, 0, sMain, trF, k, W, vsel, depth);
                        else hipLaunchKernelGGL((hrt_wf_finish_shade_kernel<F, true>), gridR, block, 0, sMain, trF, k, g, d.gb, resPrev, nPix, W, vsel, depth, cnt1);
                    }
                    else if (finalBounce || !fuse) hipLaunchKernelGGL((hrt_wf_finish_kernel<F, false>), gridR, block, 0, sMain, trF, k, W, vsel, depth);
                    else hipLaunchKernelGGL((hrt_wf_finish_shade_kernel<F, false>), gridR, block, 0, sMain, trF, k, g, d.gb, resPrev, nPix, W, vsel, depth, cnt1);
                };
                if (count)
                {
                    hipLaunchKernelGGL((hrt_wf_walk_shadow_kernel<F, true>), chained ? gridW : gridR, block, 0, sMain, tr, tr, W, vsel, depth, chained, cnt1);
                    hipLaunchKernelGGL((hrt_wf_walk_closest_kernel<F, true>), chained ? gridW : gridR, block, 0, sMain, tr, tr, W, depth, chained, cnt1);
                    launch_finish(tr);
                }
                else if ((chained || forkStatic) && forkShadow)
                {   // the two walks of a bounce are independent (shadow requests vs bounce rays) and both are persistent
                    // launches that end in a drain: on two streams the second one's workgroups move into the wave slots
                    // the first one's drain frees, instead of waiting for its last ray
                    HIPCHK(c, hipEventRecord(d.evLane[lane][0], sMain));
                    HIPCHK(c, hipStreamWaitEvent(sSide, d.evLane[lane][0], 0));
                    launch_closest(sMain);
                    launch_shadow(sSide);
                    HIPCHK(c, hipEventRecord(d.evLane[lane][1], sSide));
                    HIPCHK(c, hipStreamWaitEvent(sMain, d.evLane[lane][1], 0));
                    launch_finish(trFin);
                }
                else
                {
                    launch_shadow(sMain);
                    launch_closest(sMain);
                    launch_finish(trFin);
                }
                if (tlRc != HRT_OK) return tlRc;
            }
            else
            {   // reference layout: one-ray-per-lane walks
                if (count)
                {
                    hipLaunchKernelGGL((hrt_wf_shadow_kernel<TR, true>), gridR, block, 0, sMain, tr, W, vsel, depth, cnt1);
                    hipLaunchKernelGGL((hrt_wf_closest_kernel<TR, true>), gridR, block, 0, sMain, tr, k, W, vsel, depth, cnt1);
                }
                else
                {
                    hipLaunchKernelGGL((hrt_wf_shadow_kernel<TR, false>), gridR, block, 0, sMain, tr, W, vsel, depth, cnt1);
                    hipLaunchKernelGGL((hrt_wf_closest_kernel<TR, false>), gridR, block, 0, sMain, tr, k, W, vsel, depth, cnt1);
                }
            }
        }
        // ordered part: Lframe is summed in sample order and resCur keeps its last writer, so a batch resolves after its predecessor
        if (batch > 0 && nLanes >= 2) HIPCHK(c, hipStreamWaitEvent(sMain, d.evLane[(batch - 1) % nLanes][2], 0));
        hipLaunchKernelGGL(hrt_wf_resolve_kernel, gridP, block, 0, sMain, k, g, d.gb, d.fb, resCur, W);
        HIPCHK(c, hipGetLastError());
        if (nLanes >= 2) HIPCHK(c, hipEventRecord(d.evLane[lane][2], sMain));
    }
    // the frame ends on the device's main stream: the resolves form one chain, so its last link covers every batch of both lanes
    if (nLanes >= 2 && batch > 1 && (batch - 1) % nLanes != 0) HIPCHK(c, hipStreamWaitEvent(d.stream, d.evLane[(batch - 1) % nLanes][2], 0));
    return HRT_OK;
}

struct SahTopology { std::vector<int32_t> order; std::vector<NodeQ> nodes; std::vector<int> parent, nchild; int leaves = 0; };
void host_sah_topology(const std::vector<hrt_instance>& inst, SahTopology& out);
constexpr int64_t kAnyTreeMinInstances = 256;       // scenes of fewer instances keep the uploaded tree alone (second tree: see build_second_tree)
#ifndef HRT_SAH_MAX_LOG2            // A/B (300 001 instances: LBVH topology 12.3 ms per frame and 0.29 s per upload, SAH 11.5 ms and 0.38 s)
#define HRT_SAH_MAX_LOG2 21
#endif
constexpr int64_t kHostSahMaxInstances = (int64_t)1 << HRT_SAH_MAX_LOG2;
int build_second_tree(hrt_ctx* c, DeviceState& d, const int32_t* uploadedSlots, int64_t nSlots, bool instOnce, const SahTopology* pre = nullptr, const hrt_instance* hostInst = nullptr);       // defined with the scene-update code below

} // namespace

extern "C" {

const char* hrt_version(void)
{
#if defined(HRT_TUNING)
    return "hip_raytrace 0.4 (gfx950) tuning-build";
#elif defined(HRT_TEST_HOOKS)
    return "hip_raytrace 0.4 (gfx950) test-hooks";
#else
    return "hip_raytrace 0.4 (gfx950)";
#endif
}

int hrt_device_count(void)
try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}
catch (...) { return on_exception(nullptr, "hrt_device_count"); }

const char* hrt_last_error(hrt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hrt_create(const int* device_ids, int n_dev, hrt_ctx** out)
try {
    if (!out) return fail(nullptr, HRT_ERR_INVALID_ARG, "hrt_create: out is NULL");
    *out = nullptr;
    int avail = 0;
    hipError_t e = hipGetDeviceCount(&avail);
    if (e != hipSuccess || avail <= 0)
        return fail(nullptr, HRT_ERR_NO_DEVICE, std::string("hrt_create: no HIP device (") + hipGetErrorString(e) + "); this library has no CPU path");
    std::vector<int> ids;
    if (device_ids && n_dev > 0) ids.assign(device_ids, device_ids + n_dev);
    else ids.push_back(0);
    for (int id : ids)
        if (id < 0 || id >= avail) return fail(nullptr, HRT_ERR_INVALID_ARG, "hrt_create: device id out of range");
    struct CtxGuard { hrt_ctx* p; ~CtxGuard() { if (p) hrt_destroy(p); } } guard{new hrt_ctx()};      // an exception below must not leak the context
    hrt_ctx* c = guard.p;
    c->dev.resize(ids.size());
    for (size_t i = 0; i < ids.size(); i++)
    {
        DeviceState& d = c->dev[i];
        d.device_id = ids[i];
        hipError_t err = hipSetDevice(d.device_id);
        if (err == hipSuccess) { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, d.device_id) == hipSuccess && cu > 0) d.n_cu = cu; }
        if (err == hipSuccess) { int lds = 0; if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, d.device_id) == hipSuccess && lds > 0) d.max_lds = lds; }
        if (err == hipSuccess) err = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        if (err == hipSuccess) err = hipStreamCreateWithFlags(&d.stream2, hipStreamNonBlocking);
        for (int j = 1; j < kBatchLanes && err == hipSuccess; j++)
            for (int e = 0; e < 2 && err == hipSuccess; e++) err = hipStreamCreateWithFlags(&d.laneStream[j][e], hipStreamNonBlocking);
        for (int j = 0; j < kMaxLanes && err == hipSuccess; j++)
            for (int e = 0; e < 3 && err == hipSuccess; e++) err = hipEventCreateWithFlags(&d.evLane[j][e], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&d.evStage, hipEventDisableTiming);
        for (int f = 0; f < DeviceState::kRing && err == hipSuccess; f++)
            for (int k = 0; k < 4 && err == hipSuccess; k++) err = hipEventCreate(&d.ev[f][k]);
        if (err == hipSuccess) { void* p = nullptr; err = hipMalloc(&p, 20 * sizeof(unsigned long long)); d.counters = (unsigned long long*)p; }
        if (err != hipSuccess)
        {
            std::string m = std::string("hrt_create: ") + hipGetErrorString(err);
            return fail(nullptr, HRT_ERR_HIP, m);           // the guard destroys the half-made context
        }
    }
    for (size_t i = 0; i < ids.size(); i++)
        for (size_t j = 0; j < ids.size(); j++)
            if (ids[i] != ids[j])
            {   // tiles are exchanged device-to-device when ReSTIR reuse is on (xGMI peer copies)
                (void)hipSetDevice(ids[i]);
                hipError_t pe = hipDeviceEnablePeerAccess(ids[j], 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
    guard.p = nullptr;
    *out = c;
    return HRT_OK;
}
catch (...) { return on_exception(nullptr, "hrt_create"); }

void hrt_destroy(hrt_ctx* c)
{
    if (!c) return;
    for (DeviceState& d : c->dev) if (d.device_id >= 0 && d.stream) { (void)hipSetDevice(d.device_id); (void)hipStreamSynchronize(d.stream); }
    for (const auto& r : c->pinned) (void)hipHostUnregister(r.first);
    c->pinned.clear();
    for (DeviceState& d : c->dev)
    {
        if (d.device_id < 0) continue;
        (void)hipSetDevice(d.device_id);
        if (d.stream) (void)hipStreamSynchronize(d.stream);
        free_pixels(d);
        free_scene(d);
        free_workspace(d);
        free_present(d);
        if (d.counters) (void)hipFree(d.counters);
        for (int f = 0; f < DeviceState::kRing; f++)
            for (int k = 0; k < 4; k++) if (d.ev[f][k]) (void)hipEventDestroy(d.ev[f][k]);
        if (d.stream2) { (void)hipStreamSynchronize(d.stream2); (void)hipStreamDestroy(d.stream2); }
        for (int j = 1; j < kMaxLanes; j++) for (int e = 0; e < 2; e++) if (d.laneStream[j][e]) { (void)hipStreamSynchronize(d.laneStream[j][e]); (void)hipStreamDestroy(d.laneStream[j][e]); }
        for (int j = 0; j < kMaxLanes; j++) for (int e = 0; e < 3; e++) if (d.evLane[j][e]) (void)hipEventDestroy(d.evLane[j][e]);
        if (d.evStage) (void)hipEventDestroy(d.evStage);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    delete c;
}

int hrt_synchronize(hrt_ctx* c, hrt_stats* stats)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    hrt_stats st; std::memset(&st, 0, sizeof(st));
    st.n_devices = (int)c->dev.size();
    for (DeviceState& d : c->dev)
    {
        HIPCHK(c, hipSetDevice(d.device_id));
        HIPCHK(c, hipStreamSynchronize(d.stream));        // _cuda.Synchronize(), RTRenderer.cs:233
        double k0 = 0, k1 = 0, dh = 0;
        if (d.ring_head > 0) { d.frame_ms[0].clear(); d.frame_ms[1].clear(); }
        for (int f = 0; f < d.ring_head; f++)
        {
            float ms;
            HIPCHK(c, hipEventElapsedTime(&ms, d.ev[f][0], d.ev[f][1])); k0 += ms; d.frame_ms[0].push_back(ms);
            HIPCHK(c, hipEventElapsedTime(&ms, d.ev[f][1], d.ev[f][2])); k1 += ms; d.frame_ms[1].push_back(ms);
            HIPCHK(c, hipEventElapsedTime(&ms, d.ev[f][2], d.ev[f][3])); dh += ms;
        }
        st.kernel_ms[0] = std::max(st.kernel_ms[0], k0);
        st.kernel_ms[1] = std::max(st.kernel_ms[1], k1);
        st.d2h_ms = std::max(st.d2h_ms, dh);
        st.frames = std::max(st.frames, d.ring_head);
        if (d.ring_counts)
        {
            unsigned long long h[20];
            HIPCHK(c, hipMemcpy(h, d.counters, sizeof(h), hipMemcpyDeviceToHost));
            for (int kk = 0; kk < 2; kk++)
            {
                uint64_t* dst = reinterpret_cast<uint64_t*>(&st.k[kk]);
                for (int i = 0; i < 10; i++) dst[i] += h[kk * 10 + i];
            }
            st.counters_valid = 1;
        }
        d.ring_head = 0;
        d.ring_counts = false;
    }
    if (stats) *stats = st;
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_synchronize"); }

int hrt_scene_upload(hrt_ctx* c, const hrt_scene_desc* s)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (!s) return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_upload: scene is NULL");
    const void* src[15] = {s->tlasNodes, s->tlasInstanceIndices, s->instances, s->blasNodes, s->spherePrimIdx, s->spheres,
                           s->triPrimIdx, s->meshPositions, s->meshTris, s->meshTexcoords, s->meshTriUVs, s->triMatIndex,
                           s->materials, s->texels, s->texInfos};
    const int64_t cnt[15] = {s->n_tlasNodes, s->n_tlasInstanceIndices, s->n_instances, s->n_blasNodes, s->n_spherePrimIdx, s->n_spheres,
                             s->n_triPrimIdx, s->n_meshPositions, s->n_meshTris, s->n_meshTexcoords, s->n_meshTriUVs, s->n_triMatIndex,
                             s->n_materials, s->n_texels, s->n_texInfos};
    for (int i = 0; i < 15; i++)
        if (cnt[i] < 0 || (cnt[i] > 0 && !src[i])) return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_upload: array " + std::to_string(i) + " has a count but no pointer");
    PackedHost ph;
    {
        std::string verr = validate_and_pack(s, ph);
        if (!verr.empty()) return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_upload: " + verr);
    }
    int rc = hrt_synchronize(c, nullptr);
    if (rc != HRT_OK) return rc;
    c->scene_ready = false;
    c->packed_ok = ph.ok;
    c->packed_feat = (ph.feat & 2) ? 3 : (ph.feat & 1);
    c->small_scene = (s->n_tlasNodes + s->n_blasNodes) <= kSmallSceneNodes;
    c->flat_leaves = ph.n_flat;
    c->own_in_world = ph.own_in_world;
    c->refit_ok = ph.refit_ok && ph.ok;
    c->feat_alpha = (ph.feat & 2) != 0;
    c->n_inst = s->n_instances; c->n_tlas = s->n_tlasNodes; c->n_slots = s->n_tlasInstanceIndices; c->n_blas = s->n_blasNodes;
    c->tlas_leaves = ph.reach_leaves;
    c->tlas_on_device = false;
    c->blas_refit_ok = ph.blas_refit_ok && ph.ok;
    c->blas_rebuild_ok = c->blas_refit_ok && ph.blas_rebuild_ok;
    c->mesh_jobs = ph.meshJobs;
    c->max_mesh_items = 0;
    for (const MeshJob& J : ph.meshJobs) c->max_mesh_items = std::max(c->max_mesh_items, J.n);
    c->n_positions = s->n_meshPositions; c->n_spheres = s->n_spheres;
    for (int i = 0; i < 15; i++) c->scene_count[i] = cnt[i];
    // room for a TLAS rebuilt on the device over all instances (leaves of two: hrt_bvh.hpp)
    const int64_t capT = std::max<int64_t>(std::max<int64_t>(s->n_tlasNodes, 2 * s->n_instances - 1), 1);
    const int64_t capTI = std::max<int64_t>(std::max<int64_t>(s->n_tlasInstanceIndices, s->n_instances), 1);
    hrt_bvh_node emptyTlas; std::memset(&emptyTlas, 0, sizeof(emptyTlas));
    emptyTlas.left = emptyTlas.right = emptyTlas.first = emptyTlas.skipIndex = -1;   // an empty TLAS ends the walk at once
    // topology of the second tree (many-sphere scenes): a function of the instances alone, computed once for all devices
    SahTopology sahOnce; bool haveSah = false;
#ifndef HRT_NO_HOST_SAH
#ifndef HRT_NO_ANY_TREE
    if (ph.ok && ph.feat == 0 && s->n_instances >= kAnyTreeMinInstances && ph.n_tlasX > 0 && ph.inst_once && s->n_tlasInstanceIndices == s->n_instances &&
        ph.own_in_world && s->n_instances <= kHostSahMaxInstances && s->n_instances > 2)
    {
        const std::vector<hrt_instance> inst(s->instances, s->instances + s->n_instances);
        host_sah_topology(inst, sahOnce);
        haveSah = true;
    }
#endif
#endif
    TreeletsHost tlh;
    if (ph.ok && (ph.feat & 1) && ph.blas_refit_ok && !ph.meshRanges.empty()) build_treelets(ph.blas, ph.bsubend, ph.meshRanges, g_treelet_limits, tlh);
    for (DeviceState& d : c->dev)
    {
        HIPCHK(c, hipSetDevice(d.device_id));
        free_scene(d);                                  // UploadAll disposes + reallocates all 15 (Scene.cs:260-278)
        for (int i = 0; i < 15; i++)
        {
            int64_t n = cnt[i] > 0 ? cnt[i] : 1;       // AllocateOrEmpty: empty -> 1 zeroed element
            size_t bytes = (size_t)n * kSceneElem[i];
            const size_t room = i == 0 ? (size_t)capT * kSceneElem[0] : (i == 1 ? (size_t)capTI * kSceneElem[1] : bytes);
            HIPCHK(c, hipMalloc(&d.scene[i], std::max(bytes, room)));
            if (cnt[i] > 0) HIPCHK(c, hipMemcpyAsync(d.scene[i], src[i], bytes, hipMemcpyHostToDevice, d.stream));
            else if (i == 0) HIPCHK(c, hipMemcpyAsync(d.scene[i], &emptyTlas, bytes, hipMemcpyHostToDevice, d.stream));
            else HIPCHK(c, hipMemsetAsync(d.scene[i], 0, bytes, d.stream));
        }
        DScene& S = d.dscene;
        S.tlasNodes = (const hrt_bvh_node*)d.scene[0]; S.tlasInst = (const int32_t*)d.scene[1];
        S.instances = (const hrt_instance*)d.scene[2]; S.blasNodes = (const hrt_bvh_node*)d.scene[3];
        S.spherePrimIdx = (const int32_t*)d.scene[4]; S.spheres = (const hrt_sphere*)d.scene[5];
        S.triPrimIdx = (const int32_t*)d.scene[6]; S.meshPositions = (const hrt_float3*)d.scene[7];
        S.meshTris = (const hrt_mesh_tri*)d.scene[8]; S.meshTexcoords = (const hrt_float2*)d.scene[9];
        S.meshTriUVs = (const hrt_mesh_tri_uv*)d.scene[10]; S.triMatIndex = (const int32_t*)d.scene[11];
        S.materials = (const hrt_material*)d.scene[12]; S.texels = (const hrt_rgba32*)d.scene[13];
        S.texInfos = (const hrt_tex_info*)d.scene[14];
        S.n_texInfos = (int32_t)(cnt[14] > 0 ? cnt[14] : 1);
        // device-private repack (TracerPacked)
        const void* psrc[7] = {ph.tlas.data(), ph.finst.data(), ph.blas.data(), ph.ftri.data(), ph.flat.data(), nullptr, ph.tlasX.data()};       // slot 5 unused
        const size_t pbytes[7] = {ph.tlas.size() * sizeof(NodeQ), ph.finst.size() * sizeof(FInst), ph.blas.size() * sizeof(NodeQ), ph.ftri.size() * sizeof(FTri),
                                  ph.flat.size() * sizeof(NodeQ), 0, ph.tlasX.size() * sizeof(NodeQ)};
        const size_t proom[7] = {(size_t)capT * sizeof(NodeQ), (size_t)capTI * sizeof(FInst), 0, 0, (size_t)kFlatMaxLeaves * sizeof(NodeQ), 0,
                                 (size_t)(capT + capTI) * sizeof(NodeQ)};
        for (int i = 0; i < 7; i++)
        {
            if (!psrc[i]) continue;
            HIPCHK(c, hipMalloc(&d.packed[i], std::max(pbytes[i], proom[i])));
            HIPCHK(c, hipMemcpyAsync(d.packed[i], psrc[i], pbytes[i], hipMemcpyHostToDevice, d.stream));
        }
        {   // maintenance arrays of the device-side TLAS update
            const size_t scanTmp = (tlas_scan_temp_bytes((int)capT) + 255) & ~(size_t)255, nPart = (size_t)(capT + 255) / 256;
            const size_t ab[10] = {(size_t)capT * 4, (size_t)capT * 4, (size_t)capT * 4, (size_t)capT * 8, (size_t)capT * 8, (size_t)capT * 4, 16, 16, (size_t)capT * 4,
                                   scanTmp + 3 * nPart * 4};
            for (int i = 0; i < 10; i++) { HIPCHK(c, hipMalloc(&d.tlaux[i], ab[i])); HIPCHK(c, hipMemsetAsync(d.tlaux[i], 0, ab[i], d.stream)); }
            if (!ph.parent.empty())
            {
                HIPCHK(c, hipMemcpyAsync(d.tlaux[0], ph.parent.data(), std::min(ph.parent.size(), (size_t)capT) * 4, hipMemcpyHostToDevice, d.stream));
                HIPCHK(c, hipMemcpyAsync(d.tlaux[1], ph.nchild.data(), std::min(ph.nchild.size(), (size_t)capT) * 4, hipMemcpyHostToDevice, d.stream));
            }
            TlasDevice& T = d.tl;
            T = TlasDevice{};
            T.tlasNodes = (hrt_bvh_node*)d.scene[0]; T.tlasInst = (int32_t*)d.scene[1]; T.instances = (hrt_instance*)d.scene[2];
            T.blasNodes = (const hrt_bvh_node*)d.scene[3]; T.spherePrimIdx = (const int32_t*)d.scene[4]; T.spheres = (const hrt_sphere*)d.scene[5];
            T.tlas = (NodeQ*)d.packed[0]; T.finst = (FInst*)d.packed[1]; T.tlasX = (NodeQ*)d.packed[6]; T.flat = (NodeQ*)d.packed[4];
            T.parent = (int*)d.tlaux[0]; T.nchild = (int*)d.tlaux[1]; T.arrive = (int*)d.tlaux[2]; T.scanIn = (unsigned long long*)d.tlaux[3]; T.scanOut = (unsigned long long*)d.tlaux[4];
            T.scanTmp = d.tlaux[9]; T.scanTmpBytes = scanTmp; T.costPartial = (float*)((char*)d.tlaux[9] + scanTmp);
            T.directMax = (ph.refit_ok && !HRT_ENV("HRT_BUILDER_ORDER")) ? 63 : 1;
            T.sa = (float*)d.tlaux[5]; T.flags = (int*)d.tlaux[6]; T.cost = (float*)d.tlaux[7]; T.saBase = (float*)d.tlaux[8];
            T.nI = (int)s->n_instances; T.nT = (int)s->n_tlasNodes; T.nTI = (int)s->n_tlasInstanceIndices;
            if ((!ph.meshInst.empty() || !ph.sphereInst.empty()) && ph.blas_refit_ok)
            {
                const size_t nBq = ph.blas.size();
                static const int32_t none = 0;
                const void* bsrc[12] = {ph.bparent.data(), ph.bnchild.data(), ph.bsubend.data(), ph.borig.data(), nullptr,
                                        ph.meshInst.empty() ? &none : ph.meshInst.data(), ph.bkind.data(), ph.sphereInst.empty() ? &none : ph.sphereInst.data(),
                                        nullptr, nullptr, nullptr, nullptr};
                const size_t bb[12] = {nBq * 4, nBq * 4, nBq * 4, nBq * 4, nBq * 4, std::max<size_t>(ph.meshInst.size(), 1) * 4, nBq * 4, std::max<size_t>(ph.sphereInst.size(), 1) * 4,
                                       nBq * 4, nBq * 4, ((nBq + 255) / 256) * 8, 16};
                for (int i = 0; i < 12; i++)
                {
                    HIPCHK(c, hipMalloc(&d.blaux[i], bb[i]));
                    if (bsrc[i]) HIPCHK(c, hipMemcpyAsync(d.blaux[i], bsrc[i], bb[i], hipMemcpyHostToDevice, d.stream));
                    else HIPCHK(c, hipMemsetAsync(d.blaux[i], 0, bb[i], d.stream));
                }
                BlasDevice& B = d.bl;
                B.blasNodes = (hrt_bvh_node*)d.scene[3]; B.triPrimIdx = (const int32_t*)d.scene[6]; B.meshTris = (const hrt_mesh_tri*)d.scene[8];
                B.triPrimIdxW = (int32_t*)d.scene[6]; B.triMatIndex = (const int32_t*)d.scene[11]; B.materials = (const hrt_material*)d.scene[12];
                B.nMaterials = (int)s->n_materials; B.texLen = (int)(s->n_texInfos > 0 ? s->n_texInfos : 1);
                B.spherePrimIdx = (const int32_t*)d.scene[4]; B.spheres = (const hrt_sphere*)d.scene[5]; B.kind = (int*)d.blaux[6];
                B.sa = (float*)d.blaux[8]; B.saBase = (float*)d.blaux[9]; B.growPartial = (float*)d.blaux[10]; B.grow = (float*)d.blaux[11];
                B.positions = (hrt_float3*)d.scene[7]; B.blas = (NodeQ*)d.packed[2]; B.ftri = (FTri*)d.packed[3];
                B.parent = (int*)d.blaux[0]; B.nchild = (int*)d.blaux[1]; B.subend = (int*)d.blaux[2]; B.orig = (int*)d.blaux[3]; B.arrive = (int*)d.blaux[4];
                B.nB = (int)s->n_blasNodes; B.nSlots = (int)s->n_triPrimIdx; B.directMax = 7;   // leaves cost up to four triangle records each: 7 / 15 / 31 / 63 measured 0.47 / 0.50 / 0.52 / 0.56 ms for the refit of a 524 k-node BLAS
                B.maxRange[0] = 0; B.maxRange[1] = ph.max_range[1]; B.maxRange[2] = ph.max_range[2];
                d.n_sphere_inst = (int)ph.sphereInst.size();
                d.n_mesh_inst = (int)ph.meshInst.size();
            }
            T.capT = (int)capT; T.capTI = (int)capTI; T.flatMax = kFlatMaxLeaves;
        }
        d.dpacked.tlas = (const NodeQ*)d.packed[0]; d.dpacked.finst = (const FInst*)d.packed[1];
        d.dpacked.blas = (const NodeQ*)d.packed[2]; d.dpacked.ftri = (const FTri*)d.packed[3];
        d.dpacked.nTlas = (int)ph.tlas.size();
        d.dpacked.tlasX = ph.n_tlasX > 0 ? (const NodeQ*)d.packed[6] : nullptr; d.dpacked.nTlasX = ph.n_tlasX;
        {   // triangle records per leaf step of the walker: three where leaves of three outnumber the fuller ones, else two (hrt_walker.hpp)
            size_t n3 = 0, n4 = 0;
            for (const NodeQ& q : ph.blas)
            {
                const unsigned cnt = (unsigned)__builtin_bit_cast(int, q.hi.w) >> 28;
                if (cnt == 3) n3++; else if (cnt >= 4) n4++;
            }
            d.dpacked.leafTris = n3 > n4 ? 3 : 2;
        }
        // treelets of the big triangle-mesh BLASes: what the LDS-staged walker of production frames walks (hrt_walker_tl.hpp)
        if (ph.ok && (ph.feat & 1) && ph.blas_refit_ok && !ph.meshRanges.empty() && !tlh.tl.empty())
        {
            const void* tsrc[3] = {tlh.red.data(), tlh.tl.data(), tlh.redOfRoot.data()};
            const size_t tbytes[3] = {tlh.red.size() * sizeof(NodeQ), tlh.tl.size() * sizeof(Treelet), tlh.redOfRoot.size() * sizeof(int32_t)};
            for (int i = 0; i < 3; i++)
            {
                HIPCHK(c, hipMalloc(&d.tlmem[i], tbytes[i]));
                HIPCHK(c, hipMemcpyAsync(d.tlmem[i], tsrc[i], tbytes[i], hipMemcpyHostToDevice, d.stream));
            }
            d.dtl.red = (const NodeQ*)d.tlmem[0]; d.dtl.tl = (const Treelet*)d.tlmem[1]; d.dtl.redOfRoot = (const int*)d.tlmem[2];
            d.dtl.nTl = (int)tlh.tl.size(); d.dtl.nRed = (int)tlh.red.size();
            d.dtl.redLds = (int)std::min<size_t>(tlh.red.size(), (size_t)kTlRedLdsMax);
            d.dtl.tlBytesMax = (tlh.tlBytesMax + 15) & ~15;
            const int histBins = d.dtl.nTl <= kTlHistLds ? d.dtl.nTl : 0;
            d.tl_ok = tl_shared_bytes(d.dtl.tlBytesMax, d.dtl.redLds, histBins) <= (size_t)d.max_lds;
        }
        HIPCHK(c, hipStreamSynchronize(d.stream));      // host arrays are only borrowed for the duration of the call
        if (int rcB = build_second_tree(c, d, s->tlasInstanceIndices, s->n_tlasInstanceIndices, ph.inst_once, haveSah ? &sahOnce : nullptr, s->instances))
        {   // the second tree is an accelerator, not part of the scene: without memory for it the walks use the uploaded tree
            if (rcB != HRT_ERR_OUT_OF_MEMORY) return rcB;
            (void)hipGetLastError();
            for (int i = 0; i < 18; i++) { if (d.tl2mem[i]) (void)hipFree(d.tl2mem[i]); d.tl2mem[i] = nullptr; }
            d.tl2 = TlasDevice{}; d.any_ok = false; d.any_built = false; d.ordX = d.ordP = 0;
            c->err.clear();
        }
    }
    c->scene_ready = true;
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_scene_upload"); }

namespace {

constexpr float kAutoRebuildGrowth = 1.5f;     // HRT_REBUILD_AUTO: rebuild when the node boxes grew to this multiple of their built area (geometric mean)

int ensure_lbvh_scratch(hrt_ctx* c, DeviceState& d)
{
    if (d.tlscratch) return HRT_OK;
    TlasDevice& T = d.tl;
    const size_t n = (size_t)std::max(std::max(T.nI, c->max_mesh_items), 1);
    const size_t L = n + 1;                                                                                   // Karras' tree over the single items
    const size_t sortBytes = tlas_sort_temp_bytes((int)n), iscanBytes = tlas_iscan_temp_bytes((int)n + 1);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t total = 3 * up(n * 4) + 5 * up(L * 4) + up(6 * 4) + up(sortBytes) + 2 * up((n + 1) * 4) + up(16 * 4) + up(iscanBytes);
    HIPCHK(c, hipMalloc(&d.tlscratch, total));
    char* p = (char*)d.tlscratch;
    auto take = [&](size_t b) { char* r = p; p += up(b); return (void*)r; };
    T.keys = (unsigned*)take(n * 4); T.keysSorted = (unsigned*)take(n * 4); T.vals = (int*)take(n * 4);
    T.rngA = (int*)take(L * 4); T.rngB = (int*)take(L * 4); T.split = (int*)take(L * 4); T.parInt = (int*)take(L * 4);
    T.parLeaf = (int*)take(L * 4);
    T.cboundsKey = (unsigned*)take(6 * 4);
    T.lstart = (int*)take((n + 1) * 4); T.lsum = (int*)take((n + 1) * 4); T.leafCounts = (int*)take(16 * 4);
    T.iscanTmp = take(iscanBytes); T.iscanTmpBytes = iscanBytes;
    T.sortTmp = take(sortBytes); T.sortTmpBytes = sortBytes;
    return HRT_OK;
}

// Scenes made of many fast-sphere instances (identity transform, one sphere): a second TLAS over the same instances (topology from
// host_sah_topology below, or the LBVH of the scene updates for very many instances; everything else by the device kernels of the
// scene updates), for the walks of the streamed pipeline (hrt_walker.hpp, ALT) and launch 1.  Any-hit walks and
// the last bounce's hit-or-miss walk do not depend on the tree at all; a closest-hit walk depends on it only through the order
// in which instances at exactly the same distance are met, which the walker detects and resolves on the uploaded tree.  The
// reference's median split cuts such a scene into slabs when one instance dominates the bounds (the ground sphere of BASELINE
// config 3: 103 node visits per ray against 50).  Needs the uploaded tree to list every instance exactly once (the
// second tree is built over "the instances").  Scene updates refit it (refit_second_tree).

// Topology of the second tree built on the HOST with a binned surface-area heuristic (16 bins on each axis over the box centres of
// the range, the split of least area(left) * n(left) + area(right) * n(right); leaves of at most four instances; a range the bins cannot
// split is halved), in the numbering the walkers want (walk order: a node's first child follows it).
// Only WHICH instances share a subtree is decided here -- boxes, leaf-slot records, the inlined layout and the slack are the
// device's (tlas_finish, tlas_inflate), exactly as for the LBVH the scene updates build.  Against that LBVH: 6-10 % fewer node
// visits per ray on config 3 (tools/tree_order_model.py); the scene updates keep the LBVH, which is built in 0.3 ms, and so do
// scenes of more than two million instances.
constexpr int kSahLeaf = 4;       // instances per leaf at most (config 3, path stage + launch 1: 15.66 / 15.37 / 15.39 / 15.41 ms for 2 / 3 / 4 / 6)
void host_sah_topology(const std::vector<hrt_instance>& inst, SahTopology& out)
{
    const int n = (int)inst.size();
    constexpr int kBins = 16;
    std::vector<float> cx((size_t)n), cy((size_t)n), cz((size_t)n);
    for (int i = 0; i < n; i++)
    {
        cx[(size_t)i] = 0.5f * (inst[(size_t)i].worldBoundsMin.X + inst[(size_t)i].worldBoundsMax.X);
        cy[(size_t)i] = 0.5f * (inst[(size_t)i].worldBoundsMin.Y + inst[(size_t)i].worldBoundsMax.Y);
        cz[(size_t)i] = 0.5f * (inst[(size_t)i].worldBoundsMin.Z + inst[(size_t)i].worldBoundsMax.Z);
        // (an infinite box is legal here; its centre only has to be a number the binning can convert to an integer)
        if (!std::isfinite(cx[(size_t)i])) cx[(size_t)i] = 0.f;
        if (!std::isfinite(cy[(size_t)i])) cy[(size_t)i] = 0.f;
        if (!std::isfinite(cz[(size_t)i])) cz[(size_t)i] = 0.f;
    }
    const float* cen[3] = {cx.data(), cy.data(), cz.data()};
    struct Box { float lo[3], hi[3]; };
    auto grow = [&](Box& b, int i) {
        const hrt_instance& r = inst[(size_t)i];
        const float l[3] = {r.worldBoundsMin.X, r.worldBoundsMin.Y, r.worldBoundsMin.Z}, h[3] = {r.worldBoundsMax.X, r.worldBoundsMax.Y, r.worldBoundsMax.Z};
        for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], l[a]); b.hi[a] = std::max(b.hi[a], h[a]); }
    };
    auto unite = [](Box& b, const Box& o) { for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], o.lo[a]); b.hi[a] = std::max(b.hi[a], o.hi[a]); } };
    auto area = [](const Box& b) { const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2]; return x * y + y * z + z * x; };
    const Box empty = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    out.order.resize((size_t)n);
    for (int i = 0; i < n; i++) out.order[(size_t)i] = i;
    out.nodes.clear(); out.parent.clear(); out.nchild.clear(); out.leaves = 0;
    struct Job { int a, b, parent; };
    std::vector<Job> todo;
    todo.push_back({0, n, -1});
    while (!todo.empty())
    {
        const Job j = todo.back();
        todo.pop_back();
        const int idx = (int)out.nodes.size();
        NodeQ q{};
        out.parent.push_back(j.parent);
        const int m = j.b - j.a;
        if (m <= kSahLeaf)
        {
            q.lo.w = bits_f(j.a);
            q.hi.w = bits_f((int)((unsigned)m << 28));           // the skip link comes with the subtree sizes, below
            out.nodes.push_back(q); out.nchild.push_back(0); out.leaves++;
            continue;
        }
        int32_t* it = out.order.data() + j.a;
        int bestAxis = -1, bestK = 0; float bestCost = 0.f, bestLo = 0.f, bestScale = 0.f;
        for (int a = 0; a < 3; a++)
        {
            float lo = FLT_MAX, hi = -FLT_MAX;
            for (int i = 0; i < m; i++) { lo = std::min(lo, cen[a][it[i]]); hi = std::max(hi, cen[a][it[i]]); }
            if (!(hi > lo) || !std::isfinite(hi - lo)) continue;
            const float scale = (float)kBins / (hi - lo);
            Box bb[kBins]; int cnt[kBins];
            for (int k = 0; k < kBins; k++) { bb[k] = empty; cnt[k] = 0; }
            for (int i = 0; i < m; i++)
            {
                const int k = std::min(kBins - 1, std::max(0, (int)((cen[a][it[i]] - lo) * scale)));
                grow(bb[k], it[i]); cnt[k]++;
            }
            Box right[kBins]; int rcnt[kBins];
            Box acc = empty; int c = 0;
            for (int k = kBins - 1; k >= 1; k--) { unite(acc, bb[k]); c += cnt[k]; right[k] = acc; rcnt[k] = c; }
            acc = empty; c = 0;
            for (int k = 1; k < kBins; k++)
            {
                unite(acc, bb[k - 1]); c += cnt[k - 1];
                if (c == 0 || rcnt[k] == 0) continue;
                const float cost = area(acc) * (float)c + area(right[k]) * (float)rcnt[k];
                if (std::isfinite(cost) && (bestAxis < 0 || cost < bestCost)) { bestAxis = a; bestK = k; bestCost = cost; bestLo = lo; bestScale = scale; }
            }
        }
        int mid = m / 2;
        if (bestAxis >= 0)
        {
            const float* ca = cen[bestAxis];
            int32_t* p2 = std::partition(it, it + m, [&](int32_t i) { return std::min(kBins - 1, std::max(0, (int)((ca[i] - bestLo) * bestScale))) < bestK; });
            const int left = (int)(p2 - it);
            if (left > 0 && left < m) mid = left;
        }
        q.lo.w = bits_f(idx + 1);
        out.nodes.push_back(q); out.nchild.push_back(2);
        todo.push_back({j.a + mid, j.b, idx});       // popped second: the first child is the next node
        todo.push_back({j.a, j.a + mid, idx});
    }
    // skip link = index + size of the subtree (walk order: children have larger indices than their parent)
    const int nT = (int)out.nodes.size();
    std::vector<int> size((size_t)nT, 1);
    for (int i = nT - 1; i > 0; i--) size[(size_t)out.parent[(size_t)i]] += size[(size_t)i];
    for (int i = 0; i < nT; i++)
    {
        const int end = i + size[(size_t)i];
        const int w = __builtin_bit_cast(int, out.nodes[(size_t)i].hi.w);
        out.nodes[(size_t)i].hi.w = bits_f((w & ~kEnd) | (end >= nT ? kEnd : end));
    }
}

// The inlined second tree (TlasDevice::tlasX: nodes in walk order, every leaf followed by one record per instance) renumbered for the
// rays whose direction has the signs `sign` (+1 / -1 per axis, 0: not known): at every inner node the child whose box centre comes first along such a ray,
// on the axis that separates the two centres most, is walked first.  Same records, same subtree sizes; only the order of the two
// subtrees under a node, and with it every link, changes.  Links are written as indices into the array of all eight copies
// (`base` = where this copy starts); from[i] = the record of X that position i of the copy holds (a refit refreshes the boxes through it).
// false: the array is not the binary tree in walk order it should be (nothing is used then).
bool reorder_second_tree(const std::vector<NodeQ>& X, const int sign[3], int base, NodeQ* out, int* from, bool inlined)
{
    const int nX = (int)X.size();
    auto w_ = [](float f) { return __builtin_bit_cast(int, f); };
    auto f_ = [](int v) { return __builtin_bit_cast(float, v); };
    auto cnt = [&](int i) { return (int)((unsigned)w_(X[(size_t)i].hi.w) >> 28); };
    auto end = [&](int i) { const int sk = w_(X[(size_t)i].hi.w) & kEnd; return sk == kEnd ? nX : sk; };
    std::vector<std::pair<int, int>> todo;                  // (record in X, its index in this numbering)
    todo.emplace_back(0, 0);
    int placed = 0;
    while (!todo.empty())
    {
        const int src = todo.back().first, at = todo.back().second;
        todo.pop_back();
        if (src < 0 || src >= nX || at < 0 || at >= nX) return false;
        const int size = end(src) - src;
        if (size < 1 || at + size > nX) return false;
        const int skip = at + size == nX ? kEnd : base + at + size;
        const int c = cnt(src);
        NodeQ q = X[(size_t)src];
        if (c == 15) return false;                          // an instance record where a node should be
        if (c > 0)
        {
            if (size != (inlined ? 1 + c : 1)) return false;
            q.hi.w = f_(skip | (int)((unsigned)c << 28));
            out[at] = q; from[at] = src;
            placed += size;
            if (!inlined) continue;                         // the plain node array: a leaf names its slots, no records follow
            for (int j = 0; j < c; j++)
            {
                NodeQ r = X[(size_t)(src + 1 + j)];
                if (cnt(src + 1 + j) != 15) return false;
                r.hi.w = f_((j + 1 < c ? base + at + 2 + j : skip) | (int)(15u << 28));
                out[at + 1 + j] = r; from[at + 1 + j] = src + 1 + j;
            }
            continue;
        }
        const int l = w_(q.lo.w) & kEnd;
        if (l != src + 1 || l >= nX) return false;
        const int r = end(l);
        if (r >= nX || end(r) != end(src)) return false;    // exactly two children
        const NodeQ &L = X[(size_t)l], &R = X[(size_t)r];
        const float cl[3] = {0.5f * (L.lo.x + L.hi.x), 0.5f * (L.lo.y + L.hi.y), 0.5f * (L.lo.z + L.hi.z)};
        const float cr[3] = {0.5f * (R.lo.x + R.hi.x), 0.5f * (R.lo.y + R.hi.y), 0.5f * (R.lo.z + R.hi.z)};
        int ax = 0;
        for (int a = 1; a < 3; a++) if (std::fabs(cl[a] - cr[a]) > std::fabs(cl[ax] - cr[ax])) ax = a;
        // +1 / -1: the rays of this copy go that way along ax; 0: either way -- the builder's order stays (lower Morton code first)
        const bool leftFirst = sign[ax] > 0 ? cl[ax] <= cr[ax] : (sign[ax] < 0 ? cl[ax] >= cr[ax] : true);
        const int a = leftFirst ? l : r, b = leftFirst ? r : l;
        q.lo.w = f_(base + at + 1);
        q.hi.w = f_(skip);
        out[at] = q; from[at] = src;
        placed += 1;
        todo.emplace_back(b, at + 1 + (end(a) - a));
        todo.emplace_back(a, at + 1);
    }
    return placed == nX;
}
int build_second_tree(hrt_ctx* c, DeviceState& d, const int32_t* uploadedSlots, int64_t nSlots, bool instOnce, const SahTopology* pre, const hrt_instance* hostInst)
{
    d.any_ok = false; d.any_built = false;
#ifdef HRT_NO_ANY_TREE             // A/B
    return HRT_OK;
#endif
    // own_in_world: the second tree's leaf boxes are unions of the instances' worldBounds, and its exactness argument needs every instance's
    // own box inside them (an instance whose BLAS the position-indexed builder put over another sphere, Scene.cs:386-395, breaks that)
    if (!c->packed_ok || c->packed_feat != 0 || c->n_inst < kAnyTreeMinInstances || !d.dpacked.tlasX || !instOnce || nSlots != c->n_inst || !c->own_in_world) return HRT_OK;
    int rc = ensure_lbvh_scratch(c, d);
    if (rc != HRT_OK) return rc;
    TlasDevice T = d.tl;                                        // inputs, capacities, temporaries and LBVH scratch are shared; outputs are its own
    const size_t capT = (size_t)T.capT, capTI = (size_t)T.capTI;
    const size_t bytes[14] = {capT * sizeof(hrt_bvh_node), capTI * 4, capT * sizeof(NodeQ), capTI * sizeof(FInst), (capT + capTI) * sizeof(NodeQ),
                              (size_t)kFlatMaxLeaves * sizeof(NodeQ), capT * 4, capT * 4, capT * 4, capT * 8, capT * 8, capT * 4, capT * 4, 32};
    for (int i = 0; i < 14; i++)
    {
        if (d.tl2mem[i]) { (void)hipFree(d.tl2mem[i]); d.tl2mem[i] = nullptr; }
        HIPCHK(c, hipMalloc(&d.tl2mem[i], bytes[i]));
        HIPCHK(c, hipMemsetAsync(d.tl2mem[i], 0, bytes[i], d.stream));
    }
    T.tlasNodes = (hrt_bvh_node*)d.tl2mem[0]; T.tlasInst = (int32_t*)d.tl2mem[1]; T.tlas = (NodeQ*)d.tl2mem[2]; T.finst = (FInst*)d.tl2mem[3];
    T.tlasX = (NodeQ*)d.tl2mem[4]; T.flat = (NodeQ*)d.tl2mem[5]; T.parent = (int*)d.tl2mem[6]; T.nchild = (int*)d.tl2mem[7]; T.arrive = (int*)d.tl2mem[8];
    T.scanIn = (unsigned long long*)d.tl2mem[9]; T.scanOut = (unsigned long long*)d.tl2mem[10]; T.sa = (float*)d.tl2mem[11]; T.saBase = (float*)d.tl2mem[12];
    T.flags = (int*)d.tl2mem[13]; T.cost = (float*)((char*)d.tl2mem[13] + 16);
    int leaves = 0;
    // the instances as the host uploaded them (no copy back from the device when the caller still has them)
    std::vector<hrt_instance> inst;
    if (hostInst) inst.assign(hostInst, hostInst + c->n_inst);
    else
    {
        inst.resize((size_t)c->n_inst);
        HIPCHK(c, hipMemcpy(inst.data(), T.instances, inst.size() * sizeof(hrt_instance), hipMemcpyDeviceToHost));
    }
#ifndef HRT_NO_HOST_SAH            // A/B
    if (c->n_inst <= kHostSahMaxInstances && c->n_inst > 2)
    {
        // the topology depends on the instances alone: the upload computes it once and hands it to every device
        SahTopology own;
        if (!pre) host_sah_topology(inst, own);
        const SahTopology& sah = pre ? *pre : own;
        T.nT = (int)sah.nodes.size(); T.nTI = (int)c->n_inst; leaves = sah.leaves;
        if (T.nT > T.capT || T.nTI > T.capTI || T.nT != 2 * leaves - 1) return fail(c, HRT_ERR_HIP, "second tree: host topology does not fit");
        HIPCHK(c, hipMemcpyAsync(T.tlas, sah.nodes.data(), sah.nodes.size() * sizeof(NodeQ), hipMemcpyHostToDevice, d.stream));
        HIPCHK(c, hipMemcpyAsync(T.tlasInst, sah.order.data(), sah.order.size() * 4, hipMemcpyHostToDevice, d.stream));
        HIPCHK(c, hipMemcpyAsync(T.parent, sah.parent.data(), sah.parent.size() * 4, hipMemcpyHostToDevice, d.stream));
        HIPCHK(c, hipMemcpyAsync(T.nchild, sah.nchild.data(), sah.nchild.size() * 4, hipMemcpyHostToDevice, d.stream));
        HIPCHK(c, hipStreamSynchronize(d.stream));              // the vectors go out of scope
    }
    else
#endif
    HIPCHK(c, tlas_rebuild_topology(T, d.stream, &leaves));
    T.directMax = 63;                                           // emitted in walk order
    HIPCHK(c, tlas_finish(T, d.stream));
    HIPCHK(c, tlas_inflate(T, d.stream));
    int flags[4] = {1, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(flags, T.flags, sizeof(flags), hipMemcpyDeviceToHost, d.stream));
    HIPCHK(c, hipStreamSynchronize(d.stream));
    if (flags[0] != 0 || flags[1] != 0 || leaves <= 0 || (int64_t)T.nT + T.nTI >= kEnd) return HRT_OK;      // an instance that is not a fast sphere after all
    // leaf slot of the uploaded tree -> leaf slot of this one (both list every instance once)
    if (T.nTI != (int)nSlots) return HRT_OK;
    std::vector<int32_t> mine((size_t)nSlots), slotOfInst((size_t)c->n_inst, -1), map((size_t)nSlots);
    HIPCHK(c, hipMemcpyAsync(mine.data(), T.tlasInst, (size_t)nSlots * 4, hipMemcpyDeviceToHost, d.stream));
    HIPCHK(c, hipStreamSynchronize(d.stream));
    for (int64_t a = 0; a < nSlots; a++)
    {
        if (mine[(size_t)a] < 0 || mine[(size_t)a] >= c->n_inst || slotOfInst[(size_t)mine[(size_t)a]] >= 0) return HRT_OK;
        slotOfInst[(size_t)mine[(size_t)a]] = (int32_t)a;
    }
    for (int64_t o = 0; o < nSlots; o++) map[(size_t)o] = slotOfInst[(size_t)uploadedSlots[o]];
    if (d.tl2mem[14]) { (void)hipFree(d.tl2mem[14]); d.tl2mem[14] = nullptr; }
    HIPCHK(c, hipMalloc(&d.tl2mem[14], (size_t)nSlots * 4));
    HIPCHK(c, hipMemcpy(d.tl2mem[14], map.data(), (size_t)nSlots * 4, hipMemcpyHostToDevice));
    d.tl2 = T;
    d.dpackedAny = d.dpacked;
    d.dpackedAny.tlas = T.tlas; d.dpackedAny.finst = T.finst; d.dpackedAny.nTlas = T.nT;
    d.dpackedAny.tlasX = T.tlasX; d.dpackedAny.nTlasX = T.nT + T.nTI;
    d.dpackedAny.slotMap = (const int*)d.tl2mem[14];
    d.dpackedAny.tlasXO = nullptr; d.dpackedAny.xStride = 0; d.dpackedAny.xAxes = 0; d.dpackedAny.tlasO = nullptr; d.dpackedAny.oStride = 0;
    d.ordX = d.ordP = 0;
    if (d.tl2mem[17]) { (void)hipFree(d.tl2mem[17]); d.tl2mem[17] = nullptr; }
    HIPCHK(c, hipMalloc(&d.tl2mem[17], (size_t)nSlots * 4));
    if (d.tl2mem[15]) { (void)hipFree(d.tl2mem[15]); d.tl2mem[15] = nullptr; }
#ifndef HRT_NO_ORDERED_COPIES      // A/B
    // Which signs select a numbering: the two axes along which the instances are spread most (extent of the box centres between their
    // 5th and 95th percentile: one huge ground sphere must not count) -- measured on config 3 (22 k records, 0.7 MB a copy), every walk
    // ordered: x and z 16.2 ms, z 16.8, x 16.7, all three 18.1, none 17.5, y alone 18.9 (along y the builder's order, ground first, is the
    // better one: one sphere test bounds every ray that goes down).  The copies need not fit the L2: with only the closest-hit walks
    // on them, frames of 30 001 / 100 001 instances at 4 spp go 10.65 -> 9.4 / 14.9 -> 10.7 ms with four copies of 2.1 / 7.4 MB; the
    // budget only bounds the memory a huge scene may take.
    const int nX = T.nT + T.nTI;
#ifndef HRT_ORDERED_BUDGET_MB       // A/B
#define HRT_ORDERED_BUDGET_MB 1024
#endif
    constexpr size_t kOrderedBudget = (size_t)HRT_ORDERED_BUDGET_MB << 20;
    int axes = 0;
    {
        float ext[3];
        std::vector<float> v(inst.size());
        for (int a = 0; a < 3; a++)
        {
            for (size_t i = 0; i < inst.size(); i++)
                v[i] = a == 0 ? inst[i].worldBoundsMin.X + inst[i].worldBoundsMax.X : (a == 1 ? inst[i].worldBoundsMin.Y + inst[i].worldBoundsMax.Y : inst[i].worldBoundsMin.Z + inst[i].worldBoundsMax.Z);
            for (float& x : v) if (!std::isfinite(x)) x = 0.f;       // (an ordering for std::sort; infinite boxes are legal here)
            std::sort(v.begin(), v.end());
            ext[a] = v[v.size() - 1 - v.size() / 20] - v[v.size() / 20];
        }
        int order[3] = {0, 1, 2};
        std::sort(order, order + 3, [&](int p, int q) { return ext[p] > ext[q] || (ext[p] == ext[q] && p < q); });
        for (int k = 0; k < 2; k++)
            if (ext[order[k]] > 0.f && ext[order[k]] >= 0.25f * ext[order[0]] && (size_t)nX * sizeof(NodeQ) * (size_t)ord_copies(axes | (1 << order[k])) <= kOrderedBudget)
                axes |= 1 << order[k];
    }
    const int copies = ord_copies(axes);
    if (axes != 0 && (int64_t)nX * copies < kEnd)
    {
        std::vector<NodeQ> X((size_t)nX), all((size_t)nX * (size_t)copies);
        std::vector<int> from((size_t)(nX + T.nT) * (size_t)copies);
        HIPCHK(c, hipMemcpy(X.data(), T.tlasX, (size_t)nX * sizeof(NodeQ), hipMemcpyDeviceToHost));
        bool ok = true;
        for (int o = 0; ok && o < copies; o++)
        {
            int sign[3] = {0, 0, 0};
            for (int a = 0; a < 3; a++)      // the copy bit of axis a = the index of a direction that is positive along a only
                if (axes & (1 << a)) sign[a] = (ord_copy(axes, a == 0 ? 1.f : -1.f, a == 1 ? 1.f : -1.f, a == 2 ? 1.f : -1.f) & o) ? 1 : -1;
            ok = reorder_second_tree(X, sign, o * nX, all.data() + (size_t)o * (size_t)nX, from.data() + (size_t)o * (size_t)nX, true);
        }
        // ... and of the plain node array, for launch 1 (both in one allocation: the inlined copies first)
        const int nP = T.nT;
        std::vector<NodeQ> Pn((size_t)nP), allP((size_t)nP * (size_t)copies);
        HIPCHK(c, hipMemcpy(Pn.data(), T.tlas, (size_t)nP * sizeof(NodeQ), hipMemcpyDeviceToHost));
        for (int o = 0; ok && o < copies; o++)
        {
            int sign[3] = {0, 0, 0};
            for (int a = 0; a < 3; a++)
                if (axes & (1 << a)) sign[a] = (ord_copy(axes, a == 0 ? 1.f : -1.f, a == 1 ? 1.f : -1.f, a == 2 ? 1.f : -1.f) & o) ? 1 : -1;
            ok = reorder_second_tree(Pn, sign, o * nP, allP.data() + (size_t)o * (size_t)nP, from.data() + all.size() + (size_t)o * (size_t)nP, false);
        }
        if (ok)
        {
            HIPCHK(c, hipMalloc(&d.tl2mem[15], (all.size() + allP.size()) * sizeof(NodeQ)));
            HIPCHK(c, hipMemcpy(d.tl2mem[15], all.data(), all.size() * sizeof(NodeQ), hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy((NodeQ*)d.tl2mem[15] + all.size(), allP.data(), allP.size() * sizeof(NodeQ), hipMemcpyHostToDevice));
            d.dpackedAny.tlasXO = (const NodeQ*)d.tl2mem[15]; d.dpackedAny.xStride = nX; d.dpackedAny.xAxes = axes;
            d.dpackedAny.tlasO = (const NodeQ*)d.tl2mem[15] + all.size(); d.dpackedAny.oStride = nP;
            if (d.tl2mem[16]) { (void)hipFree(d.tl2mem[16]); d.tl2mem[16] = nullptr; }
            HIPCHK(c, hipMalloc(&d.tl2mem[16], from.size() * sizeof(int)));
            HIPCHK(c, hipMemcpy(d.tl2mem[16], from.data(), from.size() * sizeof(int), hipMemcpyHostToDevice));
            d.ordX = all.size(); d.ordP = allP.size();
        }
    }
#endif
    d.any_ok = true; d.any_built = true;
    return HRT_OK;
}

// After a scene update: the second tree keeps its topology and takes the new boxes (instance records and spheres are shared with the
// tree in use and already updated), as long as the scene is still what the second tree is exact for -- every instance a fast sphere
// with a regular box (the flags of its own finish pass), the tree in use a device refit / rebuild (unions of regular boxes: nested,
// every instance once).  A new topology of the tree in use needs a new slot map.  Otherwise the walks go back to the tree in use.
int refit_second_tree(hrt_ctx* c, DeviceState& d, bool newTopologyInUse, bool sceneStillFits)
{
    d.any_ok = false;
    if (!d.any_built || !sceneStillFits || !c->own_in_world) return HRT_OK;
    const TlasDevice& T2 = d.tl2;
    if (d.tl.nTI != T2.nTI) return HRT_OK;
    HIPCHK(c, tlas_finish(T2, d.stream));
    HIPCHK(c, tlas_inflate(T2, d.stream));
    int flags[4] = {1, 1, 0, 0};
    HIPCHK(c, hipMemcpyAsync(flags, T2.flags, sizeof(flags), hipMemcpyDeviceToHost, d.stream));
    if (newTopologyInUse) HIPCHK(c, tlas_slot_map(d.tl.tlasInst, T2.tlasInst, (int*)d.tl2mem[17], (int*)d.tl2mem[14], T2.nTI, d.stream));
    if (d.dpackedAny.tlasXO)
    {
        HIPCHK(c, tlas_refresh_copies((NodeQ*)d.tl2mem[15], T2.tlasX, (const int*)d.tl2mem[16], (int)d.ordX, d.stream));
        HIPCHK(c, tlas_refresh_copies((NodeQ*)d.tl2mem[15] + d.ordX, T2.tlas, (const int*)d.tl2mem[16] + d.ordX, (int)d.ordP, d.stream));
    }
    HIPCHK(c, hipStreamSynchronize(d.stream));
    d.any_ok = flags[0] == 0 && flags[1] == 0;
    return HRT_OK;
}

} // namespace

namespace {

// Shared tail of the scene updates: `mutate` enqueues what changes the instance records on one device (staging buffers it
// allocates go into the vector and are freed here), then the TLAS is refitted / rebuilt per `policy` and the walkers' view of
// the tree is refreshed.
int apply_update(hrt_ctx* c, int policy, const char* who, const std::function<int(DeviceState&, std::vector<void*>&)>& mutate, hrt_bvh_update_stats* st)
{
    if (policy != HRT_REBUILD_AUTO && policy != HRT_REBUILD_FORCE_REFIT && policy != HRT_REBUILD_FORCE_REBUILD)
        return fail(c, HRT_ERR_INVALID_ARG, std::string(who) + ": unknown policy");
    if (!c->packed_ok) return fail(c, HRT_ERR_INVALID_STATE, std::string(who) + ": the scene exceeds the limits of the packed layout");
    if (c->n_inst <= 0) return fail(c, HRT_ERR_INVALID_STATE, std::string(who) + ": the scene has no instances");
    if (policy != HRT_REBUILD_FORCE_REBUILD && !c->refit_ok && !c->tlas_on_device)
    {
        if (policy == HRT_REBUILD_FORCE_REFIT)
            return fail(c, HRT_ERR_INVALID_STATE, std::string(who) + ": this TLAS cannot be refitted (a node has several parents or more than 64 children); use HRT_REBUILD_FORCE_REBUILD");
        policy = HRT_REBUILD_FORCE_REBUILD;
    }
    int rc = hrt_synchronize(c, nullptr);
    if (rc != HRT_OK) return rc;
    hrt_bvh_update_stats out; std::memset(&out, 0, sizeof(out));
    bool first = true;
    for (DeviceState& d : c->dev)
    {
        HIPCHK(c, hipSetDevice(d.device_id));
        TlasDevice& T = d.tl;
        hipEvent_t e0 = d.ev[0][0], e1 = d.ev[0][1];
        HIPCHK(c, hipEventRecord(e0, d.stream));
        int h_flags[4]; float h_cost[2] = {1.f, 0.f};
        auto finish_and_read = [&]() -> int {
            HIPCHK(c, tlas_finish(T, d.stream));
            HIPCHK(c, hipMemcpyAsync(h_flags, T.flags, sizeof(h_flags), hipMemcpyDeviceToHost, d.stream));
            HIPCHK(c, hipMemcpyAsync(h_cost, T.cost, sizeof(h_cost), hipMemcpyDeviceToHost, d.stream));
            HIPCHK(c, hipStreamSynchronize(d.stream));
            return HRT_OK;
        };
        auto keep_as_base = [&]() -> int {
            HIPCHK(c, hipMemcpyAsync(T.saBase, T.sa, (size_t)T.nT * 4, hipMemcpyDeviceToDevice, d.stream));
            d.tlas_base_valid = true;
            return HRT_OK;
        };
        if (!d.tlas_base_valid && policy != HRT_REBUILD_FORCE_REBUILD)
        {   // node areas of the tree as it was built: taken once, before anything moves
            HIPCHK(c, tlas_finish(T, d.stream));
            if ((rc = keep_as_base()) != HRT_OK) return rc;
        }
        d.any_ok = false;                      // the second tree describes the scene as it was: refit_second_tree brings it back below
        std::vector<void*> staged;
        struct StagedGuard {               // staging buffers of `mutate` are freed on every way out (their copies are ordered on d.stream)
            std::vector<void*>& v; hipStream_t st;
            ~StagedGuard() { if (!v.empty()) { (void)hipStreamSynchronize(st); for (void* p : v) (void)hipFree(p); } }
        } stagedGuard{staged, d.stream};
        if ((rc = mutate(d, staged)) != HRT_OK) return rc;
        int action = policy == HRT_REBUILD_FORCE_REBUILD ? HRT_REBUILD_FORCE_REBUILD : HRT_REBUILD_FORCE_REFIT;
        if (policy == HRT_REBUILD_AUTO && !d.tlas_lbvh) action = HRT_REBUILD_FORCE_REBUILD;   // an uploaded tree: the device-built one costs as much as a refit and walks faster
        float growthRefit = 0.f;
        int rebuiltLeaves = 0;
        if (action == HRT_REBUILD_FORCE_REFIT)
        {
            if ((rc = finish_and_read()) != HRT_OK) return rc;
            growthRefit = h_cost[0];
            if (policy == HRT_REBUILD_AUTO && growthRefit > kAutoRebuildGrowth) action = HRT_REBUILD_FORCE_REBUILD;
        }
        if (action == HRT_REBUILD_FORCE_REBUILD)
        {
            if ((rc = ensure_lbvh_scratch(c, d)) != HRT_OK) return rc;
            HIPCHK(c, tlas_rebuild_topology(T, d.stream, &rebuiltLeaves));
            T.directMax = 63;                                           // emitted in walk order
            if ((rc = finish_and_read()) != HRT_OK) return rc;
            if ((rc = keep_as_base()) != HRT_OK) return rc;
            h_cost[0] = 1.f;                                            // as built
            d.tlas_lbvh = true;
        }
        HIPCHK(c, hipEventRecord(e1, d.stream));
        HIPCHK(c, hipEventSynchronize(e1));
        for (void* p : staged) (void)hipFree(p);
        staged.clear();
        // the walkers' view of the tree
        const bool general = h_flags[0] != 0;
        d.dpacked.nTlas = T.nT;
        const bool walkOrder = action == HRT_REBUILD_FORCE_REBUILD || c->tlas_on_device || !HRT_ENV("HRT_BUILDER_ORDER");
        const bool inl = !general && !c->feat_alpha && walkOrder && (int64_t)T.nT + T.nTI < kEnd && !HRT_ENV("HRT_NO_INLINE_INSTANCES");
        d.dpacked.tlasX = inl ? (const NodeQ*)d.packed[6] : nullptr;
        d.dpacked.nTlasX = inl ? T.nT + T.nTI : 0;
        // the second tree follows the scene (same topology, new boxes) or stands down
        if ((rc = refit_second_tree(c, d, action == HRT_REBUILD_FORCE_REBUILD, !general && !c->feat_alpha && h_flags[1] == 0)) != HRT_OK) return rc;
        if (first)
        {
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
            out.action = action; out.tlas_nodes = T.nT; out.tlas_slots = T.nTI; out.general_instances = general ? 1 : 0;
            out.growth_refit = growthRefit; out.growth_final = h_cost[0]; out.sah_cost = h_cost[1]; out.device_ms = ms;
            if (action == HRT_REBUILD_FORCE_REBUILD) { c->tlas_leaves = rebuiltLeaves; c->refit_ok = true; }
            c->packed_feat = c->feat_alpha ? 3 : (general ? 1 : 0);
            // the leaf sweep skips box tests the reference makes, which is only sound over nested boxes: the device's trees are unions of
            // the instances' worldBounds, so it takes every fast-sphere instance's own box to lie inside its worldBounds
            // ... and every worldBounds to be a regular box (no NaN bound, min <= max: h_flags[1]), or the unions are not nested
            c->flat_leaves = (!general && !c->feat_alpha && walkOrder && c->own_in_world && h_flags[1] == 0 && c->tlas_leaves > 0 && c->tlas_leaves <= kFlatMaxLeaves) ? c->tlas_leaves : 0;
            c->n_tlas = T.nT; c->n_slots = T.nTI;
            c->small_scene = (c->n_tlas + c->n_blas) <= kSmallSceneNodes;
            c->tlas_on_device = true;
            first = false;
        }
    }
    if (st) *st = out;
    return HRT_OK;
}

} // namespace

int hrt_scene_update_instances(hrt_ctx* c, const int32_t* ids, int32_t n, const hrt_affine3x4* xf, int32_t policy, hrt_bvh_update_stats* st)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (!c->scene_ready) return fail(c, HRT_ERR_INVALID_STATE, "hrt_scene_update_instances: no scene uploaded");
    if (n < 0 || (n > 0 && (!ids || !xf))) return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_update_instances: n instances need ids and transforms");
    {
        std::vector<uint8_t> seen((size_t)std::max<int64_t>(c->n_inst, 0), 0);
        for (int i = 0; i < n; i++)
        {
            if (ids[i] < 0 || ids[i] >= c->n_inst) return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_update_instances: instance id out of range");
            if (seen[(size_t)ids[i]]++) return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_update_instances: instance id listed twice");
        }
    }
    return apply_update(c, policy, "hrt_scene_update_instances", [&](DeviceState& d, std::vector<void*>& staged) -> int {
        if (n <= 0) return HRT_OK;
        const size_t idb = ((size_t)n * 4 + 63) & ~(size_t)63;
        void* buf = nullptr;
        HIPCHK(c, hipMalloc(&buf, idb + (size_t)n * sizeof(hrt_affine3x4)));
        staged.push_back(buf);
        HIPCHK(c, hipMemcpyAsync(buf, ids, (size_t)n * 4, hipMemcpyHostToDevice, d.stream));
        HIPCHK(c, hipMemcpyAsync((char*)buf + idb, xf, (size_t)n * sizeof(hrt_affine3x4), hipMemcpyHostToDevice, d.stream));
        HIPCHK(c, tlas_set_transforms(d.tl, (const int32_t*)buf, (const hrt_affine3x4*)((char*)buf + idb), n, d.stream));
        return HRT_OK;
    }, st);
}
catch (...) { return on_exception(c, "hrt_scene_update_instances"); }

int hrt_scene_update_positions(hrt_ctx* c, int64_t first, int64_t n, const hrt_float3* positions, int32_t policy, hrt_bvh_update_stats* st)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (!c->scene_ready) return fail(c, HRT_ERR_INVALID_STATE, "hrt_scene_update_positions: no scene uploaded");
    if (first < 0 || n < 0 || first + n > c->n_positions || (n > 0 && !positions))
        return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_update_positions: vertex range outside meshPositions");
    if (!c->blas_refit_ok)
        return fail(c, HRT_ERR_INVALID_STATE, "hrt_scene_update_positions: a triangle-mesh BLAS of this scene cannot be refitted (shared or overlapping node ranges, unreachable nodes)");
    const bool rebuildBlas = policy >= 0 && (policy & HRT_REBUILD_BLAS) != 0;
    if (policy >= 0) policy &= ~HRT_REBUILD_BLAS;
    if (rebuildBlas && !c->blas_rebuild_ok)
        return fail(c, HRT_ERR_INVALID_STATE, "hrt_scene_update_positions: a triangle-mesh BLAS of this scene cannot be rebuilt on the device (its leaves do not list their triangles in one region of triPrimIdx)");
    int blasAction = 0; float blasGrowth = 0.f;
    const int rc = apply_update(c, policy, "hrt_scene_update_positions", [&](DeviceState& d, std::vector<void*>&) -> int {
        const bool meshes = d.n_mesh_inst > 0;
        d.tl_ok = false;                                // the reduced trees hold copies of the boxes as uploaded: walks go back to the plain walker
        auto keep_base = [&]() -> int {
            HIPCHK(c, hipMemcpyAsync(d.bl.saBase, d.bl.sa, (size_t)d.bl.nB * 4, hipMemcpyDeviceToDevice, d.stream));
            d.blas_base_valid = true;
            return HRT_OK;
        };
        auto rebuild_all = [&]() -> int {
            int rc2 = ensure_lbvh_scratch(c, d);
            if (rc2 != HRT_OK) return rc2;
            for (const MeshJob& J : c->mesh_jobs) HIPCHK(c, blas_rebuild_mesh(d.tl, d.bl, J, d.stream, nullptr));
            return HRT_OK;
        };
        int rc2;
        if (meshes && !d.blas_base_valid && !rebuildBlas)
        {   // node areas of the BLASes as they were built: taken once, before the first vertex moves
            HIPCHK(c, blas_refit(d.bl, 1, d.stream));
            if ((rc2 = keep_base()) != HRT_OK) return rc2;
        }
        if (n > 0) HIPCHK(c, hipMemcpyAsync((hrt_float3*)d.scene[7] + first, positions, (size_t)n * sizeof(hrt_float3), hipMemcpyHostToDevice, d.stream));
        bool rebuilt = false;
        if (rebuildBlas && !c->mesh_jobs.empty()) { if ((rc2 = rebuild_all()) != HRT_OK) return rc2; rebuilt = true; }
        if (meshes) HIPCHK(c, blas_refit(d.bl, 1, d.stream));
        float growth = 0.f;
        if (meshes && !rebuilt)
        {
            HIPCHK(c, blas_growth(d.bl, 1, d.stream));
            HIPCHK(c, hipMemcpyAsync(&growth, d.bl.grow, 4, hipMemcpyDeviceToHost, d.stream));
            HIPCHK(c, hipStreamSynchronize(d.stream));
            if (policy == HRT_REBUILD_AUTO && growth > kAutoRebuildGrowth && c->blas_rebuild_ok && !c->mesh_jobs.empty())
            {
                if ((rc2 = rebuild_all()) != HRT_OK) return rc2;
                HIPCHK(c, blas_refit(d.bl, 1, d.stream));
                rebuilt = true;
            }
        }
        if (rebuilt && (rc2 = keep_base()) != HRT_OK) return rc2;
        if (&d == &c->dev[0]) { blasAction = meshes ? (rebuilt ? HRT_REBUILD_FORCE_REBUILD : HRT_REBUILD_FORCE_REFIT) : 0; blasGrowth = growth; }
        HIPCHK(c, tlas_rebound_instances(d.tl, (const int32_t*)d.blaux[5], d.n_mesh_inst, d.stream));
        return HRT_OK;
    }, st);
    if (rc == HRT_OK && st) { st->blas_action = blasAction; st->blas_growth = blasGrowth; }
    return rc;
}
catch (...) { return on_exception(c, "hrt_scene_update_positions"); }

int hrt_scene_update_spheres(hrt_ctx* c, int64_t first, int64_t n, const hrt_sphere* spheres, int32_t policy, hrt_bvh_update_stats* st)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (!c->scene_ready) return fail(c, HRT_ERR_INVALID_STATE, "hrt_scene_update_spheres: no scene uploaded");
    if (first < 0 || n < 0 || first + n > c->n_spheres || (n > 0 && !spheres))
        return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_update_spheres: range outside spheres");
    if (!c->blas_refit_ok)
        return fail(c, HRT_ERR_INVALID_STATE, "hrt_scene_update_spheres: a BLAS of this scene cannot be refitted (shared or overlapping node ranges, unreachable nodes)");
    return apply_update(c, policy, "hrt_scene_update_spheres", [&](DeviceState& d, std::vector<void*>&) -> int {
        if (n > 0) HIPCHK(c, hipMemcpyAsync((hrt_sphere*)d.scene[5] + first, spheres, (size_t)n * sizeof(hrt_sphere), hipMemcpyHostToDevice, d.stream));
        if (d.n_sphere_inst > 0) HIPCHK(c, blas_refit(d.bl, 2, d.stream));
        HIPCHK(c, tlas_rebound_instances(d.tl, (const int32_t*)d.blaux[7], d.n_sphere_inst, d.stream));
        return HRT_OK;
    }, st);
}
catch (...) { return on_exception(c, "hrt_scene_update_spheres"); }

int hrt_scene_download_array(hrt_ctx* c, int dev, int array, void* dst, int64_t cap, int64_t* count)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (!c->scene_ready) return fail(c, HRT_ERR_INVALID_STATE, "hrt_scene_download_array: no scene uploaded");
    if (dev < 0 || dev >= (int)c->dev.size() || array < 0 || array >= 15) return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_download_array: device slot or array index out of range");
    int rc = hrt_synchronize(c, nullptr);
    if (rc != HRT_OK) return rc;
    const int64_t have = array == 0 ? c->n_tlas : (array == 1 ? c->n_slots : c->scene_count[array]);
    if (count) *count = have;
    if (!dst) return HRT_OK;
    if (cap < have) return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_download_array: destination too small");
    DeviceState& d = c->dev[(size_t)dev];
    HIPCHK(c, hipSetDevice(d.device_id));
    if (have > 0) HIPCHK(c, hipMemcpyAsync(dst, d.scene[array], (size_t)have * kSceneElem[array], hipMemcpyDeviceToHost, d.stream));
    HIPCHK(c, hipStreamSynchronize(d.stream));
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_scene_download_array"); }

int hrt_scene_download_tlas(hrt_ctx* c, int dev, hrt_bvh_node* nodes, int64_t capN, int32_t* idx, int64_t capI, hrt_instance* inst, int64_t capInst, int64_t* counts)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (!c->scene_ready) return fail(c, HRT_ERR_INVALID_STATE, "hrt_scene_download_tlas: no scene uploaded");
    if (dev < 0 || dev >= (int)c->dev.size()) return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_download_tlas: device slot out of range");
    int rc = hrt_synchronize(c, nullptr);
    if (rc != HRT_OK) return rc;
    DeviceState& d = c->dev[(size_t)dev];
    const int64_t have[3] = {c->n_tlas, c->n_slots, c->n_inst};
    if (counts) { counts[0] = have[0]; counts[1] = have[1]; counts[2] = have[2]; }
    if ((nodes && capN < have[0]) || (idx && capI < have[1]) || (inst && capInst < have[2]))
        return fail(c, HRT_ERR_INVALID_ARG, "hrt_scene_download_tlas: destination too small");
    HIPCHK(c, hipSetDevice(d.device_id));
    if (nodes && have[0] > 0) HIPCHK(c, hipMemcpyAsync(nodes, d.scene[0], (size_t)have[0] * sizeof(hrt_bvh_node), hipMemcpyDeviceToHost, d.stream));
    if (idx && have[1] > 0) HIPCHK(c, hipMemcpyAsync(idx, d.scene[1], (size_t)have[1] * 4, hipMemcpyDeviceToHost, d.stream));
    if (inst && have[2] > 0) HIPCHK(c, hipMemcpyAsync(inst, d.scene[2], (size_t)have[2] * sizeof(hrt_instance), hipMemcpyDeviceToHost, d.stream));
    HIPCHK(c, hipStreamSynchronize(d.stream));
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_scene_download_tlas"); }

int hrt_reset_history(hrt_ctx* c)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    int rc = hrt_synchronize(c, nullptr);
    if (rc != HRT_OK) return rc;
    for (DeviceState& d : c->dev)
    {
        d.taa_history_valid = false;
        if (d.nPix == 0) continue;
        HIPCHK(c, hipSetDevice(d.device_id));
        DReservoir* rs[2] = {&d.resA, &d.resB};
        for (DReservoir* r : rs)
        {
            HIPCHK(c, hipMemsetAsync(r->L, 0, d.nPix * 12, d.stream)); HIPCHK(c, hipMemsetAsync(r->wi, 0, d.nPix * 12, d.stream));
            HIPCHK(c, hipMemsetAsync(r->pdf, 0, d.nPix * 4, d.stream)); HIPCHK(c, hipMemsetAsync(r->w, 0, d.nPix * 4, d.stream));
            HIPCHK(c, hipMemsetAsync(r->wSum, 0, d.nPix * 4, d.stream)); HIPCHK(c, hipMemsetAsync(r->m, 0, d.nPix * 4, d.stream));
            HIPCHK(c, hipMemsetAsync(r->lightId, 0, d.nPix * 4, d.stream));
        }
        HIPCHK(c, hipStreamSynchronize(d.stream));
    }
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_reset_history"); }

int hrt_render_frame(hrt_ctx* c, const hrt_frame_params* p, const hrt_render_opts* opts, const hrt_outputs* out, hrt_stats* stats)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (!p) return fail(c, HRT_ERR_INVALID_ARG, "hrt_render_frame: params is NULL");
    if (!c->scene_ready) return fail(c, HRT_ERR_INVALID_STATE, "hrt_render_frame: no scene uploaded (call hrt_scene_upload first)");
    if (p->width <= 0 || p->height <= 0) return fail(c, HRT_ERR_INVALID_ARG, "hrt_render_frame: width/height must be positive");
    if ((int64_t)p->width * p->height > 0x7FFFFFFFLL) return fail(c, HRT_ERR_INVALID_ARG, "hrt_render_frame: image too large for int pixel indices");
    if (p->maxDepth < 0) return fail(c, HRT_ERR_INVALID_ARG, "hrt_render_frame: maxDepth must be >= 0");
    const uint32_t flags = opts ? opts->flags : 0u;
    int rb = opts ? opts->row_begin : 0, re = opts ? opts->row_end : 0;
    int sn = opts && opts->strip_n > 0 ? opts->strip_n : 1, si = opts ? opts->strip_i : 0;
    if (rb == 0 && re == 0) re = p->height;
    if (rb < 0 || re > p->height || rb > re) return fail(c, HRT_ERR_INVALID_ARG, "hrt_render_frame: row range outside the image");
    if (si < 0 || si >= sn) return fail(c, HRT_ERR_INVALID_ARG, "hrt_render_frame: strip_i must be in [0, strip_n)");
    const bool nosync = (flags & HRT_FLAG_NO_SYNC) != 0;
    if (nosync && out) return fail(c, HRT_ERR_INVALID_ARG, "hrt_render_frame: HRT_FLAG_NO_SYNC frames cannot gather to host (outputs must be NULL)");
    const bool reuse = (p->enableTemporalReuse != 0 || p->enableSpatialReuse != 0);
    const int nd = (int)c->dev.size();
    const bool primaryOnly = (flags & HRT_FLAG_PRIMARY_ONLY) != 0;
    if (primaryOnly && (flags & HRT_FLAG_SKIP_PRIMARY)) return fail(c, HRT_ERR_INVALID_ARG, "hrt_render_frame: PRIMARY_ONLY and SKIP_PRIMARY exclude each other");
    if (reuse && !primaryOnly && !(flags & HRT_FLAG_EXCHANGED) && (sn > 1 || rb != 0 || re != p->height))
        return fail(c, HRT_ERR_INVALID_STATE, "hrt_render_frame: ReSTIR reuse needs every pixel's G-buffer and previous reservoir: render reuse frames as full images, "
                    "or exchange tiles between processes and say so (HRT_FLAG_PRIMARY_ONLY / HRT_FLAG_EXCHANGED; one ctx over several devices exchanges tiles itself)");
    if (reuse && nd > 1 && nosync) return fail(c, HRT_ERR_INVALID_ARG, "hrt_render_frame: multi-device reuse frames cannot be enqueued with HRT_FLAG_NO_SYNC");
    const int64_t nPix = (int64_t)p->width * p->height;
    const bool count = (flags & HRT_FLAG_COUNTERS) != 0;

    if (nPix != c->dev[0].nPix || c->dev[0].ring_head >= DeviceState::kRing)
    {   // resize (or a full event ring) drains the frames in flight first
        int rc = hrt_synchronize(c, nullptr);
        if (rc != HRT_OK) return rc;
    }
    for (DeviceState& d : c->dev)
    {
        int rc = ensure_pixels(c, d, nPix);
        if (rc != HRT_OK) return rc;
    }
    c->width = p->width; c->height = p->height;

    // 8-row strips of [rb,re) are dealt round-robin: this call owns strips s % sn == si, and
    // device i of the ctx takes every nd-th of those (sky rows are cheap, geometry rows are
    // expensive: interleaving balances the tiles without knowing the image)
    const int S = (re - rb + 7) / 8;
    for (int i = 0; i < nd; i++)
    {
        DeviceState& d = c->dev[i];
        d.row_begin = rb; d.row_end = re;
        d.strip_n = sn * nd; d.strip_i = si + sn * i;
        d.n_strips = d.strip_i < S ? (S - d.strip_i + d.strip_n - 1) / d.strip_n : 0;
    }

    // Framebuffer.GetReservoirPair: even frame -> prev = B, cur = A (Framebuffer.cs:132-145)
    const bool even = (p->frame & 1) == 0;
    const bool exchange = reuse && nd > 1;      // ReSTIR reuse reads other tiles' G-buffer and previous reservoirs
    auto frame_k = [&](const DeviceState& d) {
        FrameK k;
        k.width = p->width; k.height = p->height; k.frame = p->frame;
        k.row_begin = d.row_begin; k.row_end = d.row_end; k.strip_n = d.strip_n; k.strip_i = d.strip_i;
        k.cam = p->cam; k.prevCam = p->prevCam;
        k.dirLightDir = p->dirLightDir; k.dirLightRadiance = p->dirLightRadiance;
        {   // Float3.Normalize(k.dirLightDir) (RTRay.cs:464) is the same for every vertex of the frame: evaluated here, by the contract's
            // host definitions (IEEE sqrt and division, maxNum: include/hrt_math.h), instead of at every diffuse vertex on the device
            const float x = p->dirLightDir.X, y = p->dirLightDir.Y, z = p->dirLightDir.Z;
            const float inv = hrt_rsqrt(hrt_fmax(1e-20f, x * x + y * y + z * z));
            k.dirLightN.X = x * inv; k.dirLightN.Y = y * inv; k.dirLightN.Z = z * inv;
        }
        k.skyTop = p->skyTintTop; k.skyBottom = p->skyTintBottom;
        k.debugCamSeq = p->debugCamSeq; k.enableTemporal = p->enableTemporalReuse; k.enableSpatial = p->enableSpatialReuse;
        k.rngLockNoise = p->rngLockNoise; k.spp = p->spp; k.maxDepth = p->maxDepth;
        return k;
    };
    // pixel kernels: waves per workgroup.  One wave per workgroup gives the dispatcher the finest grain: the launch ends when
    // the last 8x8 tile ends instead of the last 32x8 tile (matters most when a rank renders 1/8 of the image)
    static const int ptWaves = HRT_ENV("HRT_PT_BLOCK") ? std::max(1, std::min(4, atoi(HRT_ENV("HRT_PT_BLOCK")) / 64)) : 4;
    auto tile_map = [&](const DeviceState& d) {
        TileMap tm;
        tm.wpb = ptWaves;
        tm.tilesX = (p->width + 8 * ptWaves - 1) / (8 * ptWaves);
        tm.tilesY = d.n_strips;
        tm.nTiles = tm.tilesX * tm.tilesY;
        return tm;
    };
    // tracer variant: the smallest packed walker that covers the committed scene, or the reference layout
    const bool usePacked = c->packed_ok && !(flags & HRT_FLAG_REFERENCE_LAYOUT);
    const int variant = usePacked ? c->packed_feat : -1;
    const bool mega = (flags & HRT_FLAG_MEGAKERNEL) ? true : ((flags & HRT_FLAG_STREAMED) ? false : c->small_scene);
    // production frames of a tiny fast-sphere scene in the fused kernel: wave-uniform sweep over the TLAS leaves
    static const bool noFlat = HRT_ENV("HRT_NO_FLAT") != nullptr;       // A/B knob
    const bool flat = variant == 0 && mega && !count && c->flat_leaves > 0 && !noFlat;
    auto with_tracer = [&](DeviceState& d, auto fn) -> int {
        if (flat) { TracerFlat t; t.tree.P = d.dpacked; t.tree.S = d.dscene; t.leaves = (const NodeQ*)d.packed[4]; t.nLeaves = c->flat_leaves; return fn(t); }
        if (variant == 0)      { TracerPackedT<0> t; t.P = d.dpacked; t.S = d.dscene; return fn(t); }
        else if (variant == 1) { TracerPackedT<1> t; t.P = d.dpacked; t.S = d.dscene; return fn(t); }
        else if (variant == 3) { TracerPackedT<3> t; t.P = d.dpacked; t.S = d.dscene; return fn(t); }
        TracerRef t; t.S = d.dscene; return fn(t);
    };
    // all-gather of per-pixel arrays between the devices of the ctx: every device receives the strips the others own.
    // Copies run on the RECEIVER's stream after it has waited for the owner's event, so no host synchronisation is needed.
    auto exchange_arrays = [&](int evIndex, auto get_arrays) -> int {
        for (int j = 0; j < nd; j++)
        {
            DeviceState& dst = c->dev[j];
            HIPCHK(c, hipSetDevice(dst.device_id));
            for (int i = 0; i < nd; i++)
            {
                if (i == j) continue;
                DeviceState& src = c->dev[i];
                HIPCHK(c, hipStreamWaitEvent(dst.stream, src.ev[src.ring_head][evIndex], 0));
                int rc = get_arrays(src, dst);
                if (rc != HRT_OK) return rc;
            }
        }
        return HRT_OK;
    };
    const int W = p->width;

    // ---- phase 1: primary visibility on every device
    for (DeviceState& d : c->dev)
    {
        HIPCHK(c, hipSetDevice(d.device_id));
        const FrameK k = frame_k(d);
        const TileMap tm = tile_map(d);
        if (count)
        {
            if (d.ring_head > 0 && !d.ring_counts) { int rc = hrt_synchronize(c, nullptr); if (rc != HRT_OK) return rc; HIPCHK(c, hipSetDevice(d.device_id)); }
            if (!d.ring_counts) HIPCHK(c, hipMemsetAsync(d.counters, 0, 20 * sizeof(unsigned long long), d.stream));
            d.ring_counts = true;
        }
        hipEvent_t* ev = d.ev[d.ring_head];
        HIPCHK(c, hipEventRecord(ev[0], d.stream));
        if (tm.nTiles > 0 && !(flags & HRT_FLAG_SKIP_PRIMARY))
        {
            const dim3 grid(tm.nTiles), block(64 * tm.wpb);
            int rcs = with_tracer(d, [&](auto tr) -> int {
                using TR = decltype(tr);
                bool second = false;
                if constexpr (std::is_same<TR, TracerPackedT<0>>::value) second = !count && d.any_ok;
                if (second)
                {
                    if constexpr (std::is_same<TR, TracerPackedT<0>>::value)
                    {
                        TracerSecond t2; t2.second = tr; t2.second.P = d.dpackedAny; t2.uploaded = tr;
                        hipLaunchKernelGGL((hrt_primary_kernel<TracerSecond, false>), grid, block, 0, d.stream, t2, k, d.gb, tm, d.counters);
                    }
                }
                else if (count) hipLaunchKernelGGL((hrt_primary_kernel<TR, true>), grid, block, 0, d.stream, tr, k, d.gb, tm, d.counters);
                else            hipLaunchKernelGGL((hrt_primary_kernel<TR, false>), grid, block, 0, d.stream, tr, k, d.gb, tm, d.counters);
                HIPCHK(c, hipGetLastError());
                return HRT_OK;
            });
            if (rcs != HRT_OK) return rcs;
        }
        HIPCHK(c, hipEventRecord(ev[1], d.stream));
    }
    if (exchange)
    {   // SpatialCompatible reads objId / normalWS / worldPos of the CURRENT frame at other pixels (RTRay.cs:363-374)
        int rc = exchange_arrays(1, [&](DeviceState& src, DeviceState& dst) -> int {
            int r;
            if ((r = copy_strips(c, src, dst.gb.worldPos, (const hrt_float3*)src.gb.worldPos, W, hipMemcpyDeviceToDevice, dst.stream)) != HRT_OK) return r;
            if ((r = copy_strips(c, src, dst.gb.normalWS, (const hrt_float3*)src.gb.normalWS, W, hipMemcpyDeviceToDevice, dst.stream)) != HRT_OK) return r;
            return copy_strips(c, src, dst.gb.objId, (const int32_t*)src.gb.objId, W, hipMemcpyDeviceToDevice, dst.stream);
        });
        if (rc != HRT_OK) return rc;
    }

    // ---- phase 2: path trace + gather on every device
    for (DeviceState& d : c->dev)
    {
        HIPCHK(c, hipSetDevice(d.device_id));
        const FrameK k = frame_k(d);
        const TileMap tm = tile_map(d);
        DReservoir resPrev = even ? d.resB : d.resA;
        DReservoir resCur = even ? d.resA : d.resB;
        hipEvent_t* ev = d.ev[d.ring_head];
        int rcs = primaryOnly ? HRT_OK : with_tracer(d, [&](auto tr) -> int {
            return run_path_stage(c, d, tr, k, tm, p->width, resPrev, resCur, (long long)nPix, count, mega, (flags & HRT_FLAG_TREELETS) != 0);
        });
        if (rcs != HRT_OK) return rcs;
        HIPCHK(c, hipEventRecord(ev[2], d.stream));
    }
    if (exchange && !primaryOnly)
    {   // next frame's resPrev must be complete on every device (temporal reprojection can land anywhere, RTRay.cs:339-360)
        int rc = exchange_arrays(2, [&](DeviceState& src, DeviceState& dst) -> int {
            const DReservoir& a = even ? src.resA : src.resB;
            const DReservoir& b = even ? dst.resA : dst.resB;
            int r;
            if ((r = copy_strips(c, src, b.L, (const hrt_float3*)a.L, W, hipMemcpyDeviceToDevice, dst.stream)) != HRT_OK) return r;
            if ((r = copy_strips(c, src, b.wi, (const hrt_float3*)a.wi, W, hipMemcpyDeviceToDevice, dst.stream)) != HRT_OK) return r;
            if ((r = copy_strips(c, src, b.pdf, (const float*)a.pdf, W, hipMemcpyDeviceToDevice, dst.stream)) != HRT_OK) return r;
            if ((r = copy_strips(c, src, b.w, (const float*)a.w, W, hipMemcpyDeviceToDevice, dst.stream)) != HRT_OK) return r;
            if ((r = copy_strips(c, src, b.wSum, (const float*)a.wSum, W, hipMemcpyDeviceToDevice, dst.stream)) != HRT_OK) return r;
            if ((r = copy_strips(c, src, b.m, (const int32_t*)a.m, W, hipMemcpyDeviceToDevice, dst.stream)) != HRT_OK) return r;
            return copy_strips(c, src, b.lightId, (const int32_t*)a.lightId, W, hipMemcpyDeviceToDevice, dst.stream);
        });
        if (rc != HRT_OK) return rc;
    }
    // ---- phase 3: per-tile gather into the caller's host framebuffer.  Each device's copies are issued by its own host
    // thread when the ctx spans several devices: a copy into pageable memory blocks its issuing thread, so one thread
    // would serialise the N gathers.  Into page-locked memory (hrt_host_register) the copies are asynchronous DMA anyway.
    auto gather_device = [&](DeviceState& d, hrt_ctx* ec) -> int {        // ec == nullptr: errors go to the calling thread's own slot
        HIPCHK(ec, hipSetDevice(d.device_id));
        DReservoir resCur = even ? d.resA : d.resB;
        hipEvent_t* ev = d.ev[d.ring_head];
        if (out)
        {
            int rc;
#define G(hostp, devp) if ((rc = gather_rows(ec, d, hostp, devp, W)) != HRT_OK) return rc
            G(out->color, d.fb.color); G(out->depth, d.fb.depth); G(out->objectId, d.fb.objectId);
            G(out->radiance, d.fb.radiance);
            G(out->gb_worldPos, d.gb.worldPos); G(out->gb_normalWS, d.gb.normalWS); G(out->gb_baseColor, d.gb.baseColor);
            G(out->gb_matId, d.gb.matId); G(out->gb_objId, d.gb.objId); G(out->gb_hitMask, d.gb.hitMask);
            G(out->res_L, resCur.L); G(out->res_wi, resCur.wi); G(out->res_pdf, resCur.pdf); G(out->res_w, resCur.w);
            G(out->res_wSum, resCur.wSum); G(out->res_m, resCur.m); G(out->res_lightId, resCur.lightId);
#undef G
            if (out->cameraId && d.row_begin == 0 && d.strip_i == 0 && d.n_strips > 0)
                HIPCHK(ec, hipMemcpyAsync(out->cameraId, d.fb.cameraId, 4, hipMemcpyDeviceToHost, d.stream));
        }
        HIPCHK(ec, hipEventRecord(ev[3], d.stream));
        return HRT_OK;
    };
    if (out && nd > 1)
    {
        std::vector<int> rcs((size_t)nd, HRT_OK);
        std::vector<std::string> errs((size_t)nd);
        std::vector<std::thread> workers;
        for (int i = 0; i < nd; i++)
            workers.emplace_back([&, i]() {
                try { rcs[(size_t)i] = gather_device(c->dev[(size_t)i], nullptr); if (rcs[(size_t)i] != HRT_OK) errs[(size_t)i] = g_create_error; }
                catch (...) { rcs[(size_t)i] = HRT_ERR_OUT_OF_MEMORY; }
            });
        for (std::thread& t : workers) t.join();
        for (int i = 0; i < nd; i++) if (rcs[(size_t)i] != HRT_OK) return fail(c, rcs[(size_t)i], "hrt_render_frame: gather of device slot " + std::to_string(i) + ": " + errs[(size_t)i]);
    }
    else
        for (DeviceState& d : c->dev) { int rc = gather_device(d, c); if (rc != HRT_OK) return rc; }
    for (DeviceState& d : c->dev) d.ring_head++;
    if (nosync) { if (stats) std::memset(stats, 0, sizeof(*stats)); return HRT_OK; }
    return hrt_synchronize(c, stats);
}
catch (...) { return on_exception(c, "hrt_render_frame"); }

int hrt_present(hrt_ctx* c, const hrt_present_params* pp, int32_t* out_color_host)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (!pp) return fail(c, HRT_ERR_INVALID_ARG, "hrt_present: params is NULL");
    if (pp->out_width <= 0 || pp->out_height <= 0 || (int64_t)pp->out_width * pp->out_height > 0x7FFFFFFFLL)
        return fail(c, HRT_ERR_INVALID_ARG, "hrt_present: output size must be positive");
    if (pp->mode != HRT_PRESENT_RESAMPLE && pp->mode != HRT_PRESENT_TAAU) return fail(c, HRT_ERR_INVALID_ARG, "hrt_present: unknown mode");
    const int nd = (int)c->dev.size();
    DeviceState& d = c->dev[0];
    if (d.nPix == 0 || c->width <= 0) return fail(c, HRT_ERR_INVALID_STATE, "hrt_present: no frame rendered yet");
    if (d.strip_n != nd || d.row_begin != 0 || d.row_end != c->height) return fail(c, HRT_ERR_INVALID_STATE, "hrt_present: the last frame was a partial tile");
    int rc = hrt_synchronize(c, nullptr);
    if (rc != HRT_OK) return rc;
    HIPCHK(c, hipSetDevice(d.device_id));
    for (int i = 1; i < nd; i++)
    {   // the resolve runs on device slot 0: bring the other devices' strips of colour and objectId over
        DeviceState& srcd = c->dev[i];
        if ((rc = copy_strips(c, srcd, d.fb.color, (const int32_t*)srcd.fb.color, c->width, hipMemcpyDeviceToDevice, d.stream)) != HRT_OK) return rc;
        if ((rc = copy_strips(c, srcd, d.fb.objectId, (const int32_t*)srcd.fb.objectId, c->width, hipMemcpyDeviceToDevice, d.stream)) != HRT_OK) return rc;
    }
    HIPCHK(c, hipSetDevice(d.device_id));
    const int outW = pp->out_width, outH = pp->out_height, inW = c->width, inH = c->height;
    const size_t outLen = (size_t)outW * outH;
    if (d.present_w != outW || d.present_h != outH)
    {   // RTTaa.Ensure (RTTaa.cs:34-47): new display size -> new history, invalid until written once
        free_present(d);
        HIPCHK(c, dalloc(d.present_color, (int64_t)outLen, d.stream));
        HIPCHK(c, dalloc(d.taa_hist_color, (int64_t)outLen, d.stream));
        HIPCHK(c, dalloc(d.taa_hist_obj, (int64_t)outLen, d.stream));
        d.present_w = outW; d.present_h = outH; d.taa_history_valid = false;
    }
    const int blocks = (int)((outLen + 255) / 256);
    if (pp->mode == HRT_PRESENT_TAAU)
    {
        TaaK k;
        k.outColor = d.present_color; k.inColorLow = d.fb.color; k.inObjIdLow = d.fb.objectId;
        k.historyColor = d.taa_hist_color; k.historyObjId = d.taa_hist_obj;
        k.outW = outW; k.outH = outH; k.inW = inW; k.inH = inH;
        k.feedback = pp->feedback <= 0.f ? 0.075f : pp->feedback;          // tunables of RTTaa.cs:77-79; "<= 0 selects the default" as the
        k.sharpness = pp->sharpness <= 0.f ? 0.10f : pp->sharpness;        // header says: a NaN is not <= 0 and goes through to the kernel,
        k.clampK = pp->clampK <= 0.f ? 1.25f : pp->clampK;                 // as a NaN written to the reference's public fields would
        k.isFirstFrame = d.taa_history_valid ? 0 : 1;
        hipLaunchKernelGGL(hrt_taa_resolve_kernel, dim3(std::min(blocks, 256 * 16)), dim3(256), 0, d.stream, k);
        d.taa_history_valid = true;
    }
    else if (inW == outW && inH == outH)
        hipLaunchKernelGGL(hrt_blit_kernel, dim3(blocks), dim3(256), 0, d.stream, (const int32_t*)d.fb.color, (long long)d.nPix, d.present_color, (long long)outLen);
    else
        hipLaunchKernelGGL(hrt_bilinear_upsample_kernel, dim3(blocks), dim3(256), 0, d.stream, (const int32_t*)d.fb.color, inW, inH, d.present_color, outW, outH);
    HIPCHK(c, hipGetLastError());
    if (out_color_host) HIPCHK(c, hipMemcpyAsync(out_color_host, d.present_color, outLen * 4, hipMemcpyDeviceToHost, d.stream));
    HIPCHK(c, hipStreamSynchronize(d.stream));
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_present"); }

int hrt_set_workspace_limit(hrt_ctx* c, int64_t max_resident_paths)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (max_resident_paths < 0) return fail(c, HRT_ERR_INVALID_ARG, "hrt_set_workspace_limit: the limit must be >= 0 (0 = default)");
    c->max_resident_paths = max_resident_paths;
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_set_workspace_limit"); }

int hrt_host_register(hrt_ctx* c, void* ptr, int64_t bytes)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (!ptr || bytes <= 0) return fail(c, HRT_ERR_INVALID_ARG, "hrt_host_register: needs a pointer and a positive size");
    for (const auto& r : c->pinned) if (r.first == (char*)ptr) return fail(c, HRT_ERR_INVALID_STATE, "hrt_host_register: this range is registered already");
    HIPCHK(c, hipSetDevice(c->dev[0].device_id));
    HIPCHK(c, hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable));
    c->pinned.emplace_back((char*)ptr, (size_t)bytes);
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_host_register"); }

int hrt_host_unregister(hrt_ctx* c, void* ptr)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    for (size_t i = 0; i < c->pinned.size(); i++)
        if (c->pinned[i].first == (char*)ptr)
        {
            int rc = hrt_synchronize(c, nullptr);                 // no copy into the range may still be in flight
            if (rc != HRT_OK) return rc;
            c->pinned.erase(c->pinned.begin() + (long)i);
            HIPCHK(c, hipHostUnregister(ptr));
            return HRT_OK;
        }
    return fail(c, HRT_ERR_INVALID_ARG, "hrt_host_unregister: range was not registered through this context");
}
catch (...) { return on_exception(c, "hrt_host_unregister"); }

int hrt_frame_times(hrt_ctx* c, int dev, int launch, float* ms, int cap, int* n)
try {
    if (!c) return HRT_ERR_INVALID_ARG;
    if (dev < 0 || dev >= (int)c->dev.size() || launch < 0 || launch > 1 || cap < 0) return fail(c, HRT_ERR_INVALID_ARG, "hrt_frame_times: device slot or launch out of range");
    const std::vector<float>& v = c->dev[(size_t)dev].frame_ms[launch];
    if (n) *n = (int)v.size();
    if (ms) for (int i = 0; i < cap && i < (int)v.size(); i++) ms[i] = v[(size_t)i];
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_frame_times"); }

int hrt_device_buffers(hrt_ctx* c, int dev, hrt_device_views* o)
try {
    if (!c || !o) return HRT_ERR_INVALID_ARG;
    if (dev < 0 || dev >= (int)c->dev.size()) return fail(c, HRT_ERR_INVALID_ARG, "hrt_device_buffers: device slot out of range");
    DeviceState& d = c->dev[dev];
    if (d.nPix == 0) return fail(c, HRT_ERR_INVALID_STATE, "hrt_device_buffers: no frame rendered yet");
    o->row_begin = d.row_begin; o->row_end = d.row_end; o->strip_n = d.strip_n; o->strip_i = d.strip_i;
    o->width = c->width; o->height = c->height; o->device_id = d.device_id; o->reserved = 0;
    o->color = d.fb.color; o->depth = d.fb.depth; o->objectId = d.fb.objectId; o->radiance = d.fb.radiance;
    o->gb_worldPos = d.gb.worldPos; o->gb_normalWS = d.gb.normalWS; o->gb_baseColor = d.gb.baseColor;
    o->gb_matId = d.gb.matId; o->gb_objId = d.gb.objId; o->gb_hitMask = d.gb.hitMask;
    o->present_color = d.present_color; o->present_width = d.present_w; o->present_height = d.present_h;
    const DReservoir* rs[2] = {&d.resA, &d.resB};
    void** dst[2] = {o->res_a, o->res_b};
    for (int i = 0; i < 2; i++)
    {
        dst[i][0] = rs[i]->L; dst[i][1] = rs[i]->wi; dst[i][2] = rs[i]->pdf; dst[i][3] = rs[i]->w;
        dst[i][4] = rs[i]->wSum; dst[i][5] = rs[i]->m; dst[i][6] = rs[i]->lightId;
    }
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_device_buffers"); }

#ifdef HRT_TEST_HOOKS
// test hook: evaluate hrt_math.h function `fn` on device 0 of ctx (see hrt_math_probe_kernel)
int hrt_math_probe(hrt_ctx* c, int fn, int n, const float* x, const float* y, float* out)
try {
    if (!c || !x || !out || n <= 0) return HRT_ERR_INVALID_ARG;
    DeviceState& d = c->dev[0];
    HIPCHK(c, hipSetDevice(d.device_id));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIPCHK(c, hipMalloc((void**)&dx, (size_t)n * 4));
    HIPCHK(c, hipMalloc((void**)&dout, (size_t)n * 4));
    HIPCHK(c, hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice));
    if (y) { HIPCHK(c, hipMalloc((void**)&dy, (size_t)n * 4)); HIPCHK(c, hipMemcpy(dy, y, (size_t)n * 4, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(hrt_math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, d.stream, fn, n, dx, dy, dout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(d.stream));
    HIPCHK(c, hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dout); if (dy) (void)hipFree(dy);
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_math_probe"); }

// test hook: compares a trimmed device function with its IEEE definition over EVERY float of its stated domain, on the device
// (which: 0 rsqrt_clamped, 1 sqrt_normal_range); *mismatches = number of differing bit patterns, *first_bad = the smallest one
int hrt_math_exhaustive(hrt_ctx* c, int which, uint64_t* mismatches, uint32_t* first_bad)
try {
    if (!c || !mismatches || which < 0 || which > 1) return HRT_ERR_INVALID_ARG;
    DeviceState& d = c->dev[0];
    HIPCHK(c, hipSetDevice(d.device_id));
    unsigned long long* dm = nullptr;
    HIPCHK(c, hipMalloc((void**)&dm, 16));
    const unsigned long long init[2] = {0ull, 0xFFFFFFFFull};
    HIPCHK(c, hipMemcpy(dm, init, 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(hrt_math_exhaustive_kernel, dim3(4096), dim3(256), 0, d.stream, which, dm, (unsigned*)(dm + 1));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(d.stream));
    unsigned long long h[2];
    HIPCHK(c, hipMemcpy(h, dm, 16, hipMemcpyDeviceToHost));
    (void)hipFree(dm);
    *mismatches = h[0];
    if (first_bad) *first_bad = (uint32_t)h[1];
    return HRT_OK;
}
catch (...) { return on_exception(c, "hrt_math_exhaustive"); }
#endif // HRT_TEST_HOOKS

} // extern "C"

#ifdef HRT_TEST_HOOKS
// test hooks of the treelet cut (include/hrt_test_hooks.h)
extern "C" int hrt_debug_set_treelet_limits(int bytes, int min_nodes, int min_blas_nodes)
{
    g_treelet_limits = TreeletLimits{};
    if (bytes > 0) { g_treelet_limits.bytes = bytes; g_treelet_limits.minNodes = min_nodes; g_treelet_limits.minBlasNodes = min_blas_nodes; }
    if (min_blas_nodes < 0) g_treelet_limits.minBlasNodes = 0x7FFFFFFF;
    return 0;
}

extern "C" int hrt_debug_treelet_count(hrt_ctx* c)
{
    if (!c || c->dev.empty()) return 0;
    return c->dev[0].tl_ok ? c->dev[0].dtl.nTl : 0;
}

extern "C" int hrt_debug_treelets(const hrt_scene_desc* s, int bytes, int min_nodes, int min_blas_nodes,
                                  float* blas, float* red, int32_t* red_orig, int32_t* treelets, int32_t* red_of_root, int64_t* counts)
try {
    if (!s || !counts) return HRT_ERR_INVALID_ARG;
    PackedHost ph;
    if (!validate_and_pack(s, ph).empty()) return HRT_ERR_INVALID_ARG;
    TreeletLimits lim;
    if (bytes > 0) { lim.bytes = bytes; lim.minNodes = min_nodes; lim.minBlasNodes = min_blas_nodes; }
    TreeletsHost th;
    if (ph.ok && (ph.feat & 1) && ph.blas_refit_ok && !ph.meshRanges.empty()) build_treelets(ph.blas, ph.bsubend, ph.meshRanges, lim, th);
    counts[0] = (int64_t)ph.blas.size(); counts[1] = (int64_t)th.red.size(); counts[2] = (int64_t)th.tl.size();
    if (blas) std::memcpy(blas, ph.blas.data(), ph.blas.size() * sizeof(NodeQ));
    if (red && !th.red.empty()) std::memcpy(red, th.red.data(), th.red.size() * sizeof(NodeQ));
    if (red_orig && !th.redOrig.empty()) std::memcpy(red_orig, th.redOrig.data(), th.redOrig.size() * sizeof(int32_t));
    if (treelets && !th.tl.empty()) std::memcpy(treelets, th.tl.data(), th.tl.size() * sizeof(Treelet));
    if (red_of_root && !th.redOfRoot.empty()) std::memcpy(red_of_root, th.redOfRoot.data(), th.redOfRoot.size() * sizeof(int32_t));
    return HRT_OK;
}
catch (...) { return on_exception(nullptr, "hrt_debug_treelets"); }
#endif // HRT_TEST_HOOKS

#ifdef HRT_TL_STATS
// variant builds only (tools/tl_stats.py): read and clear the treelet walker's statistics of the current device
extern "C" int hrt_debug_tl_stats(unsigned long long* out256)
{
    if (hipMemcpyFromSymbol(out256, HIP_SYMBOL(g_tl_stats), sizeof(g_tl_stats)) != hipSuccess) return -1;
    static const unsigned long long zero[8][2][16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tl_stats), zero, sizeof(g_tl_stats)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef HRT_PT_STATS
// variant builds only (tools/pt_stats.py): read and clear the fused kernel's section statistics of the current device
extern "C" int hrt_debug_pt_stats(unsigned long long* out64)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_pt_stats), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    unsigned long long z[64] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pt_stats), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif

#ifdef HRT_TEST_HOOKS
// ---- host-only test hooks of the second tree (include/hip_raytrace.h)
extern "C" {

int hrt_debug_second_tree_topology(const hrt_instance* instances, int32_t n, int32_t* order, int32_t* link, int32_t* skip, int32_t* count,
                                   int32_t* parent, int32_t* n_nodes)
try {
    if (!instances || n < 1 || !order || !link || !skip || !count || !parent || !n_nodes) return HRT_ERR_INVALID_ARG;
    std::vector<hrt_instance> inst(instances, instances + n);
    SahTopology t;
    host_sah_topology(inst, t);
    *n_nodes = (int32_t)t.nodes.size();
    std::memcpy(order, t.order.data(), (size_t)n * 4);
    for (size_t i = 0; i < t.nodes.size(); i++)
    {
        const int lo = __builtin_bit_cast(int, t.nodes[i].lo.w), hi = __builtin_bit_cast(int, t.nodes[i].hi.w);
        link[i] = lo; skip[i] = hi & kEnd; count[i] = (int32_t)((unsigned)hi >> 28); parent[i] = t.parent[i];
    }
    return HRT_OK;
}
catch (...) { return on_exception(nullptr, "hrt_debug_second_tree_topology"); }

int hrt_debug_second_tree_reorder(const float* records, int32_t n_records, const int32_t* sign, int32_t base, int32_t inlined, float* out_records, int32_t* from)
try {
    if (!records || n_records < 1 || !sign || !out_records || !from) return HRT_ERR_INVALID_ARG;
    std::vector<NodeQ> X((size_t)n_records), out((size_t)n_records);
    std::memcpy(X.data(), records, (size_t)n_records * sizeof(NodeQ));
    const int sg[3] = {sign[0], sign[1], sign[2]};
    if (!reorder_second_tree(X, sg, base, out.data(), from, inlined != 0)) return HRT_ERR_INVALID_ARG;
    std::memcpy(out_records, out.data(), (size_t)n_records * sizeof(NodeQ));
    return HRT_OK;
}
catch (...) { return on_exception(nullptr, "hrt_debug_second_tree_reorder"); }

} // extern "C"
#endif // HRT_TEST_HOOKS

#ifdef HRT_WALK_STATS
// variant builds only (tools/walk_stats.py): read and clear the walker's phase statistics of the current device
extern "C" int hrt_debug_walk_stats(unsigned long long* out48)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(hrt::g_walk_stats), sizeof(unsigned long long) * 48) != hipSuccess) return -1;
    unsigned long long z[48] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(hrt::g_walk_stats), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif
