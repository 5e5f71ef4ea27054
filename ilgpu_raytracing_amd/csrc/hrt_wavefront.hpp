// hrt_wavefront.hpp -- the path-trace launch (PathTraceKernel, Engine/RTRay.cs:203-325) as a
// STREAMED pipeline instead of one megakernel.
//
// The reference runs spp x maxDepth bounces of one pixel inside one thread.  On a 64-lane
// machine that couples three things that want different shapes: BVH walks (latency-bound
// pointer chasing: wants 8 waves/SIMD and ~40 registers), ReSTIR candidate generation (pure
// ALU: ~1.5k instructions per vertex, ~100 registers) and path bookkeeping.  Fused, the walk
// runs at the occupancy of the fattest stage and every lane whose path ended idles until the
// slowest path of its wave finishes all samples.
//
// Here every (pixel, sample) pair is an independent PATH whose state lives in HBM as
// structure-of-arrays planes (one float per plane per path, so a wave reads/writes 256
// contiguous bytes per plane), and one bounce is three small kernels:
//
//     wf_shade   (ALU)     vertex -> bounce ray (+ shadow-ray request for diffuse vertices)
//     wf_shadow  (walk)    any-hit; unoccluded -> Li += T * direct
//     wf_closest (walk)    closest hit -> next vertex, or Li += T * sky and the path ends
//
// between the depth-0 form of wf_shade (paths start from the G-buffer) and wf_resolve (ordered per-pixel sum over samples,
// reservoir hand-off, framebuffer store).  Samples of a pixel run concurrently; the only
// cross-sample orderings of the reference are reproduced explicitly: Lframe is summed in sample
// order by wf_resolve, and resCur receives the reservoir of the LAST sample that reached a
// diffuse vertex (RTRay.cs:231,292-296) via a per-path staging plane.
//
// Compaction without atomics: the path-id space is cut into RANGES of kRange slots, one wave
// per range.  A wave compacts its survivors with ballot/popcount prefixes into the front of
// its own range and records the count; the next kernel's wave of the same range reads only
// that many.  No global counter, no atomics, deterministic order, every plane access a
// contiguous 256-byte run per wave.  Dead paths cost nothing; a range whose paths all ended
// exits on its first scalar load.
//
// Arithmetic per path is exactly path_trace_pixel's (hrt_device.hpp): same functions, same
// order, RNG state carried in a plane, so results stay bit-identical to the oracle.
#pragma once
#include "hrt_device.hpp"
#include "hrt_walker.hpp"

namespace hrt {

#ifndef HRT_RANGE
#define HRT_RANGE 384
#endif
constexpr int kRange = HRT_RANGE;            // slots per range (one wave owns one range); 256 / 384 / 512 / 1024 measured, DESIGN.md 8
static_assert(kRange % 64 == 0 && 4 * kRange <= 65536, "a range is a whole number of waves; shade tags a vertex with its request's offset in 16 bits");

struct Planes {                              // plane p of slot i = base[p * stride + i]
    float* base; long long stride;
    HRT_D float ldf(int p, long long i) const { return base[p * stride + i]; }
    HRT_D int ldi(int p, long long i) const { return __float_as_int(base[p * stride + i]); }
    HRT_D void stf(int p, long long i, float v) const { base[p * stride + i] = v; }
    HRT_D void sti(int p, long long i, int v) const { base[p * stride + i] = __int_as_float(v); }
    HRT_D F3 ld3(int p, long long i) const { return mk3(ldf(p, i), ldf(p + 1, i), ldf(p + 2, i)); }
    HRT_D void st3(int p, long long i, F3 v) const { stf(p, i, v.x); stf(p + 1, i, v.y); stf(p + 2, i, v.z); }
    // A group of 4 planes starting at plane p can instead hold one 16-byte record per slot (record i at float 4*i of the
    // group): what a walker lane fetches for ONE ray at a time is then one dwordx4 load from one page instead of four
    // dword loads from four planes (strides are multiples of 256 floats, so records stay 16-byte aligned).
    HRT_D float4 ld4(int p, long long i) const { return *reinterpret_cast<const float4*>(base + p * stride + 4 * i); }
    HRT_D void st4(int p, long long i, float4 v) const { *reinterpret_cast<float4*>(base + p * stride + 4 * i) = v; }
};
HRT_D float4 mkq(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

// vertex state (buffers A / B, ping-pong per bounce)
enum { V_POS = 0, V_NRM = 3, V_ALB = 6, V_IDIR = 9, V_T = 12, V_LI = 15, V_RNG = 18, V_PID = 19, V_MAT = 20, V_IOR = 21, V_PLANES = 22 };
// ray request written by wf_shade for the same slot
// RQ_A = (o.xyz, d.x)  RQ_B = (d.yz, flags, rng)  RQ_H = raw winner of the closest-hit walk (t, tObj, leaf slot, primitive): 16-byte records
enum { RQ_A = 0, RQ_B = 4, R_T = 8, RQ_H = 11, R_PLANES = 15 };
enum { RF_DEAD = 1, RF_WROTE = 2, RF_SHADOW = 4 };   // flags of RQ_B.z (RF_WROTE also in V_MAT bit 17); RF_SHADOW: bits 8..23 = the vertex' shadow request, relative to its group of four ranges
// shadow request (compacted per range)
// SQ_A = (o.xyz, d.x)  SQ_B = (d.yz, slot in range, add.x): 16-byte records; add.yz in two planes
enum { SQ_A = 0, SQ_B = 4, S_ADDY = 8, S_ADDZ = 9, S_VIS = 10, S_PLANES = 11 };      // S_VIS: 1 once the walk found the request unoccluded
// reservoir staging per path id
enum { G_WI = 0, G_PDF = 3, G_W = 4, G_WSUM = 5, G_M = 6, G_LID = 7, G_FLAG = 8, G_PLANES = 9 };     // L is res_L(wi, w, lightId): not staged

struct WfBuffers {
    Planes A, B, R, SQ;          // stride = cap
    Planes Rn, SQn;              // packed pipeline: the ray / shadow requests of the NEXT bounce (the fused finish + shade kernel reads one set and
                                 // writes the other; they alias planes 0..14 of A and 0..10 of B, which that pipeline never stores vertices in)
    Planes sampleLi;             // 3 planes, indexed by path id
    Planes stage;                // G_PLANES planes, indexed by path id
    Planes accum;                // 3 planes over pixel ordinals (Lframe carried across sample batches)
    int* cntA;                   // [depth][range] live paths of a range at a depth
    int* cntS;                   // [depth][range] shadow requests
    int* grab;                   // [depth][2][8] range hand-out counters of the walk launches (zeroed per sample batch)
    int nRanges;
    int pingpong;                // 1: frames whose finish and next shade are one kernel (the two request sets swap with every bounce)
    // requests of bounce `depth` in R / SQ, those of the next one in Rn / SQn
    HRT_D WfBuffers at_depth(int depth) const
    {
        WfBuffers w = *this;
        if (pingpong && (depth & 1)) { w.R = Rn; w.SQ = SQn; w.Rn = R; w.SQn = SQ; }
        return w;
    }
};

struct WfGeom {
    int nOrd;                    // pixel ordinals of this tile (multiple of 64: 8x8 wave tiles)
    int tilesX8;                 // 8-pixel tiles per row
    int batchStart, batchCount;  // samples [batchStart, batchStart+batchCount) of spp
    int lastBatch;
};

// pixel ordinal -> global pixel; ordinals enumerate 8x8 tiles of the device's strips row-major
HRT_D bool ord_pixel(const WfGeom& g, const FrameK& k, int ord, int& x, int& y)
{
    int tile = ord >> 6, lane = ord & 63;
    int ts = tile / g.tilesX8, tx = tile - ts * g.tilesX8;
    x = tx * 8 + (lane & 7);
    y = k.row_begin + (ts * k.strip_n + k.strip_i) * 8 + (lane >> 3);
    return x < k.width && y < k.row_end;
}

// wave-local stable compaction: slot offset of this lane among the keepers, and their number
HRT_D int wave_prefix(bool keep, int& total)
{
    unsigned long long m = __ballot(keep);
    total = __popcll(m);
    return __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
}

// One vertex of the bounce loop (RTRay.cs:235-312): the bounce ray, the new throughput, the shadow request of a diffuse vertex and the
// staged reservoir of a sample's first diffuse vertex.  Shared by the depth-0 shade (vertices from the G-buffer), the stand-alone shade
// of the reference-layout pipeline (vertices from the V planes) and the fused finish + shade kernel (vertices straight from the winners).
// pid: path id; index0: pixel index (FIRST only: else derived from pid); poison: `Li += T * 0` of a non-finite throughput (:286/:291).
template <bool COUNT, bool FIRST>
HRT_D void shade_vertex(const FrameK& k, const WfGeom& g, const DGBuffer& gb, const DReservoir& resPrev, long long nPix, const WfBuffers& W, int depth,
                        F3 pos, F3 nrm, F3 alb, F3 I, int mat, float ior0, int pid, int index0,
                        F3& T, Rng& rng, int& flg, Ray& ray, bool& wantShadow, F3& so, F3& sd, F3& sadd, bool& poison, F3& poisonAdd, Cnt<COUNT>& C)
{
    const int shade = mat & 0xFFFF;
    flg = (mat >> 16) & RF_WROTE;
    if (shade == HRT_SHADING_MIRROR)
    {   // :235-244
        F3 dirR = I - nrm * (2.f * dot(I, nrm));
        ray = ray_with_normal_offset(pos, nrm, dirR);
        T = T * alb;
    }
    else if (shade == HRT_SHADING_GLASS)
    {   // :246-275
        const float ior = ior0;
        F3 Nuse = nrm;
        bool outside = dot(I, nrm) < 0.f;
        if (!outside) Nuse = Nuse * -1.f;
        float iorUse = ior > 0.f ? ior : 1.5f;
        float etaI = outside ? 1.f : iorUse;
        float etaT = outside ? iorUse : 1.f;
        F3 dirR = I - Nuse * (2.f * dot(I, Nuse));
        float eta = etaI / etaT;
        float cosIr = -dot(I, Nuse);
        float kk = 1.f - eta * eta * (1.f - cosIr * cosIr);
        bool refrOk = !(kk < 0.f);
        F3 dirT = mk3(0.f, 0.f, 0.f);
        if (refrOk) dirT = normalize(I * eta + Nuse * (eta * cosIr - hrt_sqrt(kk)));
        float cosI = hrt_abs(dot(I, Nuse));
        float r0 = (etaI - etaT) / (etaI + etaT);
        r0 = r0 * r0;
        float om = 1.f - cosI;
        float om2 = om * om;
        float Fr = r0 + (1.f - r0) * (om2 * om2 * om);
        float xi = rng.next_f();
        bool reflect = (!refrOk || xi < Fr);
        ray = reflect ? ray_with_normal_offset(pos, Nuse, dirR) : ray_with_normal_offset(pos, -Nuse, dirT);
        if (refrOk && xi >= Fr)
        {
            F3 tint = (alb.x == 0.f && alb.y == 0.f && alb.z == 0.f) ? mk3(1.f, 1.f, 1.f) : alb;
            float etaScale = (etaI * etaI) / (etaT * etaT);
            T = T * tint * etaScale;
        }
    }
    else
    {   // :277-317
        int index = index0;
        if (!FIRST)
        {
            int x, y;
            ord_pixel(g, k, pid % g.nOrd, x, y);
            index = y * k.width + x;
        }
        Frame fr = make_frame(nrm);
        Res r = restir_candidates<COUNT>(k, gb, resPrev, nPix, index, !(flg & RF_WROTE), pos, fr, alb, rng, C);
        if (r.m > 0 && r.wSum > 0.f && r.w > 0.f)
        {
            F3 wiSel = r.wi;
            int lidSel = r.lightId == 2 ? 2 : 1;
            float nlSel = hrt_fmax(0.f, dot(nrm, wiSel));
            if (nlSel > 0.f)
            {
                Ray sray = ray_with_normal_offset(pos, nrm, wiSel);
                float pdfSel = (lidSel == 2) ? hrt_fmax(kEPS_MIN, 1.f / 9.f) : hrt_fmax(kEPS_MIN, cos_hemi_pdf(nrm, wiSel) * (8.f / 9.f));
                F3 LiSel = (lidSel == 2) ? cv3(k.dirLightRadiance) : sky(k, wiSel);
                F3 f_over_p = alb * LiSel * ((nlSel / pdfSel) * kINV_PI);
                float Wt = r.wSum / (float)hrt_imax(1, r.m) / hrt_fmax(kEPS_MIN, r.w);
                wantShadow = true;
                so = sray.o; sd = sray.d;
                sadd = T * (f_over_p * Wt);          // added to Li by wf_shadow iff unoccluded (:286/:291)
            }
        }
        // The reference adds T * contrib also when contrib is (0, 0, 0) -- no light selected, or occluded (:286/:291).  That is a
        // no-op (Li is never -0) unless a component of T is not finite (albedo products that overflowed): inf * 0 = NaN, and the
        // component stays non-finite to the end of the sample, where SafeColor zeroes it, whatever else is added -- so it is
        // poisoned here, before the visibility is known.  (T = 1 at a sample's first vertex.)
        if (!FIRST && !(hrt_isfinite(T.x) && hrt_isfinite(T.y) && hrt_isfinite(T.z))) { poison = true; poisonAdd = T * mk3(0.f, 0.f, 0.f); }
        if (!(flg & RF_WROTE))
        {   // first diffuse vertex of this sample: stage the reservoir (resCur.Write :292-296)
            W.stage.st3(G_WI, pid, r.wi); W.stage.stf(G_PDF, pid, res_pdf(nrm, r));
            W.stage.stf(G_W, pid, r.w); W.stage.stf(G_WSUM, pid, r.wSum); W.stage.sti(G_M, pid, r.m);
            W.stage.sti(G_LID, pid, r.lightId); W.stage.sti(G_FLAG, pid, 1);
            flg |= RF_WROTE;
        }
        F3 wi = sample_hemisphere_cosine(fr, rng);
        ray = ray_with_normal_offset(pos, nrm, wi);
        T = T * alb;
        if (depth >= 3)
        {   // :306-312
            float maxC = hrt_fmax(T.x, hrt_fmax(T.y, T.z));
            maxC = hrt_clamp(maxC, 0.05f, 0.98f);
            if (rng.next_f() > maxC)
            {   // throughput = 0; break  -> the path ends with its current Li (+ this vertex's direct light)
                flg |= RF_DEAD;
            }
            else T = T * (1.0f / maxC);
        }
    }
}

// ------------------------------------------------------------------ shade: vertex -> ray requests (RTRay.cs:235-312)
// FIRST (depth 0): the vertices are the G-buffer's (RTRay.cs:212-231) and are never written out as path state.  The four
// waves of a workgroup own four consecutive ranges; the live paths (pixels with a primary hit) of those 4 x kRange path ids are
// packed together into the front of the group, as wf_finish_wave packs survivors: pass 1 lists them in LDS by rank (ballot
// prefixes + one block-wide offset), pass 2 lets lane l of iteration t of wave j shade rank j kRange + 64 t + l at its own slot.
// Only what the later kernels read of a vertex that was never stored goes to the vertex planes: V_PID and V_LI = 0.
template <bool COUNT, bool FIRST>
HRT_D void wf_shade_wave(const FrameK& k, const WfGeom& g, const DGBuffer& gb, const DReservoir& resPrev, long long nPix,
                         const WfBuffers& W, const Planes& V, int depth, int range, Cnt<COUNT>& C)
{
    const int lane = threadIdx.x & 63;
    const long long base = (long long)range * kRange;
    __shared__ int s_first_cnt[FIRST ? 4 : 1];
    __shared__ int s_first_pid[FIRST ? 4 * kRange : 1];
    int n;
    if (FIRST)
    {
        const int wv = threadIdx.x >> 6;
        const long long nPaths = (long long)g.batchCount * g.nOrd;
        auto live_at = [&](long long pid) -> bool {
            if (pid >= nPaths) return false;
            const int p32 = (int)pid, s = p32 / g.nOrd;                   // a sample batch holds at most 2^25 paths
            int x, y;
            return ord_pixel(g, k, p32 - s * g.nOrd, x, y) && gb.hitMask[y * k.width + x] != 0;
        };
        int mine = 0;
        if (range >= 0)
            for (int it = 0; it < kRange / 64; it++) mine += __popcll(__ballot(live_at(base + it * 64 + lane)));
        if (lane == 0) s_first_cnt[wv] = mine;
        __syncthreads();
        int before = 0, groupTotal = 0;
        for (int j = 0; j < 4; j++) { const int cj = s_first_cnt[j]; if (j < wv) before += cj; groupTotal += cj; }
        if (range >= 0)
        {
            int at = before;
            for (int it = 0; it < kRange / 64; it++)
            {
                const long long pid = base + it * 64 + lane;
                const bool live = live_at(pid);
                int total;
                const int off = wave_prefix(live, total);
                if (live) s_first_pid[at + off] = (int)pid;
                at += total;
            }
        }
        __syncthreads();
        if (range < 0) return;
        n = max(0, min(kRange, groupTotal - (range & 3) * kRange));
        if (lane == 0) W.cntA[range] = n;
    }
    else n = W.cntA[depth * W.nRanges + range];
    int sqCount = 0;
    for (int it = 0; it * 64 < n; it++)
    {
        const int i = it * 64 + lane;
        const bool valid = i < n;
        const long long slot = base + i;
        bool wantShadow = false;
        F3 so = mk3(0.f, 0.f, 0.f), sd = so, sadd = so, T = so;
        Rng rng; rng.s = 0;
        int flg = 0;
        Ray ray; ray.o = so; ray.d = so;
        if (valid)
        {
            F3 pos, nrm, alb, I; int mat; int pid0 = 0, index0 = 0; float ior0 = 0.f;
            if (FIRST)
            {   // the G-buffer vertex every sample starts from (:221-230)
                pid0 = s_first_pid[(range & 3) * kRange + i];
                const int s0 = pid0 / g.nOrd;
                int x, y;
                ord_pixel(g, k, pid0 - s0 * g.nOrd, x, y);
                index0 = y * k.width + x;
                pos = ld3(&gb.worldPos[index0]);
                nrm = normalize(ld3(&gb.normalWS[index0]));                   // :222
                alb = ld3(&gb.baseColor[index0]);
                I = normalize(pos - cv3(k.cam.origin));                        // ViewDirFromCam :230
                T = mk3(1.f, 1.f, 1.f);
                const SeedBase sb = seed_base((uint32_t)(index0 % hrt_imax(1, k.width)), (uint32_t)(index0 / hrt_imax(1, k.width)), k.frame, 0xC0FFEEu, k.rngLockNoise);
                rng = rng_for_sample(sb, (uint32_t)(g.batchStart + s0));
                const int packedMat = gb.matId[index0];
                mat = packedMat & 0xFFFF;
                ior0 = (float)((packedMat >> 16) & 0xFFFF) / 1000.f;          // I16ToFloat :226
                V.sti(V_PID, slot, pid0);
                V.st3(V_LI, slot, mk3(0.f, 0.f, 0.f));
            }
            else
            {
                pos = V.ld3(V_POS, slot); nrm = V.ld3(V_NRM, slot); alb = V.ld3(V_ALB, slot); I = V.ld3(V_IDIR, slot);
                T = V.ld3(V_T, slot);
                rng.s = (uint32_t)V.ldi(V_RNG, slot);
                mat = V.ldi(V_MAT, slot);
            }
            bool poison = false; F3 poisonAdd = mk3(0.f, 0.f, 0.f);
            int pidv = pid0;
            if (!FIRST) { pidv = V.ldi(V_PID, slot); ior0 = V.ldf(V_IOR, slot); }
            shade_vertex<COUNT, FIRST>(k, g, gb, resPrev, nPix, W, depth, pos, nrm, alb, I, mat, ior0, pidv, index0, T, rng, flg, ray, wantShadow, so, sd, sadd, poison, poisonAdd, C);
            if (poison) V.st3(V_LI, slot, V.ld3(V_LI, slot) + poisonAdd);
        }
        int total;
        int off = wave_prefix(wantShadow, total);
        if (wantShadow)
        {
            // the request lives in the SHADING wave's own range; vertex and request refer to each other relative to the 4-range group
            // (the fused finish + shade kernel writes a vertex into another range of its group than the one its wave owns)
            const long long groupBase = (long long)(range & ~3) * kRange;
            long long q = base + sqCount + off;
            W.SQ.st4(SQ_A, q, mkq(so.x, so.y, so.z, sd.x));
            W.SQ.st4(SQ_B, q, mkq(sd.y, sd.z, __int_as_float((int)(slot - groupBase)), sadd.x));
            W.SQ.stf(S_ADDY, q, sadd.y); W.SQ.stf(S_ADDZ, q, sadd.z);
            W.SQ.sti(S_VIS, q, 0);
            flg |= RF_SHADOW | ((int)(q - groupBase) << 8);        // wf_finish adds the direct light if the walk leaves S_VIS set
        }
        if (FIRST && valid && !(flg & RF_WROTE)) W.stage.sti(G_FLAG, s_first_pid[(range & 3) * kRange + i], 0);     // no reservoir from this sample yet
        if (valid)
        {
            W.R.st4(RQ_A, slot, mkq(ray.o.x, ray.o.y, ray.o.z, ray.d.x));
            W.R.st4(RQ_B, slot, mkq(ray.d.y, ray.d.z, __int_as_float(flg), __int_as_float((int)rng.s)));
            W.R.st3(R_T, slot, T);
        }
        sqCount += total;
    }
    if (lane == 0) W.cntS[depth * W.nRanges + range] = sqCount;
}

// ------------------------------------------------------------------ shadow: Li += T * direct where visible (:518-539)
template <class TR, bool COUNT>
HRT_D void wf_shadow_wave(const TR& tr, const WfBuffers& W, const Planes& V, int depth, int range, Cnt<COUNT>& C)
{
    const int lane = threadIdx.x & 63;
    const long long base = (long long)range * kRange;
    const int n = W.cntS[depth * W.nRanges + range];
    for (int it = 0; it * 64 < n; it++)
    {
        const int j = it * 64 + lane;
        if (j < n)
        {
            const long long q = base + j;
            const float4 qa = W.SQ.ld4(SQ_A, q), qb = W.SQ.ld4(SQ_B, q);
            Ray r; r.o = mk3(qa.x, qa.y, qa.z); r.d = mk3(qa.w, qb.x, qb.y); r.inv = inv_dir(r.d);
            if (!tr.template occluded<COUNT>(r, 1e29f, C))
            {
                const long long slot = (long long)(range & ~3) * kRange + __float_as_int(qb.z);
                V.st3(V_LI, slot, V.ld3(V_LI, slot) + mk3(qb.w, W.SQ.ldf(S_ADDY, q), W.SQ.ldf(S_ADDZ, q)));
            }
        }
    }
}

// ------------------------------------------------------------------ closest: next vertex or sky (TraceNext :659-671, :241-243)
template <class TR, bool COUNT>
HRT_D void wf_closest_wave(const TR& tr, const FrameK& k, const WfBuffers& W, const Planes& V, const Planes& Vn, int depth, int range, Cnt<COUNT>& C)
{
    const int lane = threadIdx.x & 63;
    const long long base = (long long)range * kRange;
    const int n = W.cntA[depth * W.nRanges + range];
    const bool lastDepth = depth + 1 >= k.maxDepth;
    int outCount = 0;
    for (int it = 0; it * 64 < n; it++)
    {
        const int i = it * 64 + lane;
        const long long slot = base + i;
        // only the ray is live across the walk; throughput, radiance, RNG and ids are fetched after it
        bool survive = false, missed = false, dead = false;
        Hit h; Ray r;
        if (i < n)
        {
            const float4 qa = W.R.ld4(RQ_A, slot), qb = W.R.ld4(RQ_B, slot);
            dead = (__float_as_int(qb.z) & RF_DEAD) != 0;        // Russian roulette ended the path in wf_shade
            if (!dead)
            {
                r.o = mk3(qa.x, qa.y, qa.z); r.d = mk3(qa.w, qb.x, qb.y); r.inv = inv_dir(r.d);
                if (!tr.template closest<COUNT>(r, h, C)) missed = true;
                else survive = !lastDepth;
            }
            if (!survive)
            {   // the path ends here: final Li (+ T * sky on a miss, :241-243)
                const int pid = V.ldi(V_PID, slot);
                F3 Li = V.ld3(V_LI, slot);
                if (missed) Li = Li + W.R.ld3(R_T, slot) * sky(k, r.d);
                W.sampleLi.st3(0, pid, Li);
            }
        }
        int total;
        int off = wave_prefix(survive, total);
        if (survive)
        {
            long long o = base + outCount + off;
            Vn.st3(V_POS, o, r.o + r.d * h.t);
            Vn.st3(V_NRM, o, normalize(h.n));
            Vn.st3(V_ALB, o, h.albedo);
            Vn.st3(V_IDIR, o, r.d);
            Vn.st3(V_T, o, W.R.ld3(R_T, slot));
            Vn.st3(V_LI, o, V.ld3(V_LI, slot));
            const float4 qb2 = W.R.ld4(RQ_B, slot);
            Vn.sti(V_RNG, o, __float_as_int(qb2.w));
            Vn.sti(V_PID, o, V.ldi(V_PID, slot));
            Vn.sti(V_MAT, o, (h.shade & 0xFFFF) | ((__float_as_int(qb2.z) & RF_WROTE) << 16));
            Vn.stf(V_IOR, o, h.ior);
        }
        outCount += total;
    }
    if (lane == 0) W.cntA[(depth + 1) * W.nRanges + range] = outCount;
}

// ------------------------------------------------------------------ persistent-wave variants (packed layout): walk, then finish
// Hands the path ranges of one walk launch to persistent waves.  The ranges are cut into 8 contiguous partitions,
// one per XCD (workgroup i runs on XCD i & 7), so waves that share an L2 walk neighbouring screen regions; a wave
// whose partition is exhausted steals from the next ones.  One atomic per range per launch.
struct RangeGrab {
    int* ctr;            // 8 counters, zeroed before the launch
    int nRanges;
    int part, tried;     // wave-uniform
    int own;             // >= 0: static mode, the wave takes this one range and nothing else
    HRT_D void init(int* counters, int n, int ownRange) { ctr = counters; nRanges = n; part = blockIdx.x & 7; tried = 0; own = ownRange; }
    HRT_D int next()
    {
        if (own != -1) { const int r = own < nRanges ? own : -1; own = nRanges; return r; }
        int r = -1, t = tried;
        if ((threadIdx.x & 63) == 0)
        {
            while (t < 8)
            {
                const int p = (part + t) & 7;
                const int lo = (int)((long long)nRanges * p / 8), hi = (int)((long long)nRanges * (p + 1) / 8);
                const int i = (hi > lo) ? atomicAdd(&ctr[p], 1) : 0;
                if (i < hi - lo) { r = lo + i; break; }
                t++;
            }
        }
        tried = __builtin_amdgcn_readfirstlane(t);
        return __builtin_amdgcn_readfirstlane(r);
    }
};

template <int FEAT, bool COUNT, bool ALT, int LT>
HRT_D void wf_walk_shadow_wave(const TracerPackedT<FEAT>& tr, const TracerPackedT<FEAT>& exact, const WfBuffers& W, const Planes& V, int depth, int* grabCtr, int ownRange, Cnt<COUNT>& C)
{
    RangeGrab G; G.init(grabCtr, W.nRanges, ownRange);
    const int* cnt = W.cntS + (size_t)depth * W.nRanges;
    walk_queue<FEAT, true, COUNT, false, ALT, LT>(tr, exact,
        [&](int& base, int& n) { for (;;) { const int r = G.next(); if (r < 0) return false; n = cnt[r]; base = r * kRange; if (n > 0) return true; } },
        [&](int q, Ray& r, float& tMax) {
            const float4 qa = W.SQ.ld4(SQ_A, q), qb = W.SQ.ld4(SQ_B, q);
            r.o = mk3(qa.x, qa.y, qa.z); r.d = mk3(qa.w, qb.x, qb.y); r.inv = inv_dir(r.d); tMax = 1e29f; return true;
        },
        [&](int q, const WalkResult& res) {
            if (!res.occluded) W.SQ.sti(S_VIS, q, 1);         // wf_finish_wave adds the request's light to its vertex
        }, C);
}

template <int FEAT, bool COUNT, bool EXISTS, bool ALT, int LT>
HRT_D void wf_walk_closest_wave(const TracerPackedT<FEAT>& tr, const TracerPackedT<FEAT>& exact, const WfBuffers& W, int depth, int* grabCtr, int ownRange, Cnt<COUNT>& C)
{
    RangeGrab G; G.init(grabCtr, W.nRanges, ownRange);
    const int* cnt = W.cntA + (size_t)depth * W.nRanges;
    walk_queue<FEAT, false, COUNT, EXISTS, ALT, LT>(tr, exact,
        [&](int& base, int& n) { for (;;) { const int r = G.next(); if (r < 0) return false; n = cnt[r]; base = r * kRange; if (n > 0) return true; } },
        [&](int slot, Ray& r, float& tMax) {
            tMax = 1e30f;
            const float4 qa = W.R.ld4(RQ_A, slot), qb = W.R.ld4(RQ_B, slot);
            if (__float_as_int(qb.z) & RF_DEAD) return false;
            r.o = mk3(qa.x, qa.y, qa.z); r.d = mk3(qa.w, qb.x, qb.y); r.inv = inv_dir(r.d);
            return true;
        },
        [&](int slot, const WalkResult& res) {
            W.R.st4(RQ_H, slot, mkq(res.t, res.tObj, __int_as_float(res.slot), __int_as_float(res.prim)));
        }, C);
}

// next vertex or end of path from the raw winners (TraceNext :659-671, :241-243), with segmented compaction
// The four waves of a workgroup own four consecutive ranges.  Their survivors are packed TOGETHER into the front of that
// 1024-slot group (a path may change range: everything it owns travels by path id), so the next bounce sees a few full
// ranges and empty ones instead of four ranges with a few dozen live paths each -- fuller waves in wf_shade, wf_finish and
// the static walk launches.  Pass 1 counts each wave's survivors (flags and hit distance only), pass 2 is the real work
// with the block-wide offset.  range < 0: a wave past the end that still takes part in the barrier.
template <int FEAT, bool COUNT>
HRT_D void wf_finish_wave(const TracerPackedT<FEAT>& tr, const FrameK& k, const WfBuffers& W, const Planes& V, const Planes& Vn, int depth, int range)
{
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long base = (long long)range * kRange;
    const int n = range >= 0 ? W.cntA[depth * W.nRanges + range] : 0;
    const bool lastDepth = depth + 1 >= k.maxDepth;
    int mine = 0;
    if (!lastDepth)
        for (int it = 0; it * 64 < n; it++)
        {
            const int i = it * 64 + lane;
            bool sv = false;
            if (i < n)
            {
                const float4 qb = W.R.ld4(RQ_B, base + i);
                if (!(__float_as_int(qb.z) & RF_DEAD)) sv = W.R.ld4(RQ_H, base + i).x < 1e29f;
            }
            mine += __popcll(__ballot(sv));
        }
    if (lane == 0) s_cnt[wv] = mine;
    __syncthreads();
    int before = 0, groupTotal = 0;
    for (int j = 0; j < 4; j++) { const int cj = s_cnt[j]; if (j < wv) before += cj; groupTotal += cj; }
    if (range < 0) return;
    const int groupRange = range & ~3;
    const long long groupBase = (long long)groupRange * kRange;
    int outCount = before;
    for (int it = 0; it * 64 < n; it++)
    {
        const int i = it * 64 + lane;
        const long long slot = base + i;
        bool survive = false;
        Hit h; Ray r;
        F3 Li = mk3(0.f, 0.f, 0.f);
        if (i < n)
        {
            const float4 qa = W.R.ld4(RQ_A, slot), qb = W.R.ld4(RQ_B, slot);
            const int flg = __float_as_int(qb.z);
            const bool dead = (flg & RF_DEAD) != 0;
            bool missed = false;
            Li = V.ld3(V_LI, slot);
            if (flg & RF_SHADOW)
            {   // direct light of this vertex, if its shadow walk found the light unoccluded (:286/:291)
                const long long q = (long long)(range & ~3) * kRange + ((flg >> 8) & 0xFFFF);
                if (W.SQ.ldi(S_VIS, q)) Li = Li + mk3(W.SQ.ld4(SQ_B, q).w, W.SQ.ldf(S_ADDY, q), W.SQ.ldf(S_ADDZ, q));
            }
            if (!dead)
            {
                r.o = mk3(qa.x, qa.y, qa.z); r.d = mk3(qa.w, qb.x, qb.y); r.inv = mk3(0.f, 0.f, 0.f);
                const float4 qh = W.R.ld4(RQ_H, slot);
                const float ht = qh.x;
                if (!(ht < 1e29f)) missed = true;
                else if (!lastDepth)
                {
                    survive = true;
                    tr.finish_hit(r, ht, qh.y, __float_as_int(qh.z), __float_as_int(qh.w), h);
                }
            }
            if (!survive)
            {
                const int pid = V.ldi(V_PID, slot);
                if (missed) Li = Li + W.R.ld3(R_T, slot) * sky(k, r.d);
                W.sampleLi.st3(0, pid, Li);
            }
        }
        int total;
        int off = wave_prefix(survive, total);
        if (survive)
        {
            long long o = groupBase + outCount + off;
            Vn.st3(V_POS, o, r.o + r.d * h.t);
            Vn.st3(V_NRM, o, normalize(h.n));
            Vn.st3(V_ALB, o, h.albedo);
            Vn.st3(V_IDIR, o, r.d);
            Vn.st3(V_T, o, W.R.ld3(R_T, slot));
            Vn.st3(V_LI, o, Li);
            const float4 qb2 = W.R.ld4(RQ_B, slot);
            Vn.sti(V_RNG, o, __float_as_int(qb2.w));
            Vn.sti(V_PID, o, V.ldi(V_PID, slot));
            Vn.sti(V_MAT, o, (h.shade & 0xFFFF) | ((__float_as_int(qb2.z) & RF_WROTE) << 16));
            Vn.stf(V_IOR, o, h.ior);
        }
        outCount += total;
    }
    // range j of the group now holds survivors [256 j, 256 j + 256) of the group
    if (lane == 0) W.cntA[(depth + 1) * W.nRanges + range] = max(0, min(kRange, groupTotal - (range - groupRange) * kRange));
}

// finish of bounce `depth` FUSED with the shade of bounce depth + 1 (packed layout, every bounce but the last): the winner's shading
// feeds the next vertex in registers -- position, normal, albedo, incoming direction, throughput, RNG and material never round-trip
// through the 22 vertex planes (88 B written and read back per surviving path and bounce before); only Li and the path id are kept
// for the finish that follows.  Same arithmetic in the same order as wf_finish_wave + wf_shade_wave, so the same bits.
template <int FEAT, bool COUNT>
HRT_D void wf_finish_shade_wave(const TracerPackedT<FEAT>& tr, const FrameK& k, const WfGeom& g, const DGBuffer& gb, const DReservoir& resPrev, long long nPix,
                                const WfBuffers& W, const Planes& V, const Planes& Vn, int depth, int range, Cnt<COUNT>& C)
{
    __shared__ int s_cnt[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long base = (long long)range * kRange;
    const int n = range >= 0 ? W.cntA[depth * W.nRanges + range] : 0;
    int mine = 0;
    for (int it = 0; it * 64 < n; it++)
    {
        const int i = it * 64 + lane;
        bool sv = false;
        if (i < n)
        {
            const float4 qb = W.R.ld4(RQ_B, base + i);
            if (!(__float_as_int(qb.z) & RF_DEAD)) sv = W.R.ld4(RQ_H, base + i).x < 1e29f;
        }
        mine += __popcll(__ballot(sv));
    }
    if (lane == 0) s_cnt[wv] = mine;
    __syncthreads();
    int before = 0, groupTotal = 0;
    for (int j = 0; j < 4; j++) { const int cj = s_cnt[j]; if (j < wv) before += cj; groupTotal += cj; }
    if (range < 0) return;
    const int groupRange = range & ~3;
    const long long groupBase = (long long)groupRange * kRange;
    int outCount = before, sqCount = 0;
    // The winners are shaded 64 source entries at a time (dependent random fetches: instance, triangle, material records), but only
    // the survivors go on to the expensive half (eight ReSTIR candidates per diffuse vertex): they wait in a wave-local LDS list until
    // 64 of them fill a wave, so that half runs with full lanes whatever the survival rate.  An entry is what the fetches produced
    // (normal, albedo, material word, ior, hit distance), the radiance so far and the source slot; ray, throughput, RNG state and path
    // id are read again from the source slot.  The requests of bounce depth + 1 go to the OTHER set (Rn / SQn): other waves of the
    // group may still be reading this wave's range as bounce `depth`.
    constexpr int kPend = 128;
    __shared__ float s_pend[4][13][kPend];
    float (*pend)[kPend] = s_pend[wv];
    int nPend = 0;
    auto shade_batch = [&](int cnt) {        // the first cnt <= 64 entries of the list become vertices of bounce depth + 1
        const bool act = lane < cnt;
        bool wantShadow = false;
        F3 Li = mk3(0.f, 0.f, 0.f), T = Li, so = Li, sd = Li, sadd = Li;
        Rng rng; rng.s = 0;
        int flgNext = 0, pid = 0;
        Ray ray; ray.o = Li; ray.d = Li;
        if (act)
        {
            const long long slot = base + __float_as_int(pend[0][lane]);
            const F3 nrm = mk3(pend[1][lane], pend[2][lane], pend[3][lane]), alb = mk3(pend[4][lane], pend[5][lane], pend[6][lane]);
            const int mat = __float_as_int(pend[7][lane]);
            const float ior = pend[8][lane], ht = pend[9][lane];
            Li = mk3(pend[10][lane], pend[11][lane], pend[12][lane]);
            const float4 qa = W.R.ld4(RQ_A, slot), qb = W.R.ld4(RQ_B, slot);
            const F3 ro = mk3(qa.x, qa.y, qa.z), rd = mk3(qa.w, qb.x, qb.y);
            pid = V.ldi(V_PID, slot);
            T = W.R.ld3(R_T, slot);
            rng.s = (uint32_t)__float_as_int(qb.w);
            bool poison = false; F3 poisonAdd = mk3(0.f, 0.f, 0.f);
            shade_vertex<COUNT, false>(k, g, gb, resPrev, nPix, W, depth + 1, ro + rd * ht, nrm, alb, rd, mat, ior, pid, 0, T, rng, flgNext, ray, wantShadow, so, sd, sadd, poison, poisonAdd, C);
            if (poison) Li = Li + poisonAdd;
        }
        int totalS;
        const int offS = wave_prefix(wantShadow, totalS);
        if (act)
        {
            const long long o = groupBase + outCount + lane;
            if (wantShadow)
            {
                const long long q = base + sqCount + offS;
                W.SQn.st4(SQ_A, q, mkq(so.x, so.y, so.z, sd.x));
                W.SQn.st4(SQ_B, q, mkq(sd.y, sd.z, __int_as_float((int)(o - groupBase)), sadd.x));
                W.SQn.stf(S_ADDY, q, sadd.y); W.SQn.stf(S_ADDZ, q, sadd.z);
                W.SQn.sti(S_VIS, q, 0);
                flgNext |= RF_SHADOW | ((int)(q - groupBase) << 8);
            }
            Vn.st3(V_LI, o, Li);
            Vn.sti(V_PID, o, pid);
            W.Rn.st4(RQ_A, o, mkq(ray.o.x, ray.o.y, ray.o.z, ray.d.x));
            W.Rn.st4(RQ_B, o, mkq(ray.d.y, ray.d.z, __int_as_float(flgNext), __int_as_float((int)rng.s)));
            W.Rn.st3(R_T, o, T);
        }
        outCount += cnt;
        sqCount += totalS;
        // what is left moves to the front of the list
        __builtin_amdgcn_wave_barrier();
        float keep[13];
        const bool mv = lane + cnt < nPend;
        if (mv) for (int c = 0; c < 13; c++) keep[c] = pend[c][lane + cnt];
        __builtin_amdgcn_wave_barrier();
        if (mv) for (int c = 0; c < 13; c++) pend[c][lane] = keep[c];
        __builtin_amdgcn_wave_barrier();
        nPend -= cnt;
    };
    for (int it = 0; it * 64 < n; it++)
    {
        const int i = it * 64 + lane;
        const long long slot = base + i;
        bool survive = false;
        Hit h;
        F3 Li = mk3(0.f, 0.f, 0.f);
        int matNext = 0;
        if (i < n)
        {
            const float4 qa = W.R.ld4(RQ_A, slot), qb = W.R.ld4(RQ_B, slot);
            const int flg = __float_as_int(qb.z);
            const bool dead = (flg & RF_DEAD) != 0;
            bool missed = false;
            Ray r; r.o = mk3(qa.x, qa.y, qa.z); r.d = mk3(qa.w, qb.x, qb.y); r.inv = mk3(0.f, 0.f, 0.f);
            Li = V.ld3(V_LI, slot);
            if (flg & RF_SHADOW)
            {
                const long long q = groupBase + ((flg >> 8) & 0xFFFF);
                if (W.SQ.ldi(S_VIS, q)) Li = Li + mk3(W.SQ.ld4(SQ_B, q).w, W.SQ.ldf(S_ADDY, q), W.SQ.ldf(S_ADDZ, q));
            }
            if (!dead)
            {
                const float4 qh = W.R.ld4(RQ_H, slot);
                const float ht = qh.x;
                if (!(ht < 1e29f)) missed = true;
                else { survive = true; tr.finish_hit(r, ht, qh.y, __float_as_int(qh.z), __float_as_int(qh.w), h); matNext = (h.shade & 0xFFFF) | ((flg & RF_WROTE) << 16); }
            }
            if (!survive)
            {
                if (missed) Li = Li + W.R.ld3(R_T, slot) * sky(k, r.d);
                W.sampleLi.st3(0, V.ldi(V_PID, slot), Li);
            }
        }
        int total;
        const int off = wave_prefix(survive, total);
        if (survive)
        {
            const int e = nPend + off;
            const F3 nn = normalize(h.n);
            pend[0][e] = __int_as_float(i);
            pend[1][e] = nn.x; pend[2][e] = nn.y; pend[3][e] = nn.z;
            pend[4][e] = h.albedo.x; pend[5][e] = h.albedo.y; pend[6][e] = h.albedo.z;
            pend[7][e] = __int_as_float(matNext); pend[8][e] = h.ior; pend[9][e] = h.t;
            pend[10][e] = Li.x; pend[11][e] = Li.y; pend[12][e] = Li.z;
        }
        nPend += total;
        __builtin_amdgcn_wave_barrier();
        if (nPend >= 64) shade_batch(64);
    }
    if (nPend > 0) shade_batch(nPend);
    if (lane == 0)
    {
        W.cntA[(depth + 1) * W.nRanges + range] = max(0, min(kRange, groupTotal - (range - groupRange) * kRange));
        W.cntS[(depth + 1) * W.nRanges + range] = sqCount;
    }
}

// ------------------------------------------------------------------ resolve: ordered sample sum, reservoir hand-off, framebuffer store (:320-324)
HRT_D void wf_resolve_pixel(const FrameK& k, const WfGeom& g, const DGBuffer& gb, const DFramebuffer& fb, const DReservoir& resCur,
                            const WfBuffers& W, int ord)
{
    int x, y;
    if (!ord_pixel(g, k, ord, x, y)) return;
    const int index = y * k.width + x;
    if (index == 0 && fb.cameraId && g.batchStart == 0) fb.cameraId[0] = k.debugCamSeq;
    F3 Lframe = g.batchStart == 0 ? mk3(0.f, 0.f, 0.f) : W.accum.ld3(0, ord);
    const int hitMask = gb.hitMask[index];
    if (hitMask == 0)
    {
        F3 c = safe_color(sky(k, primary_ray(k, x, y).d));                    // :214-219
        for (int s = 0; s < g.batchCount; s++) Lframe = Lframe + c;
    }
    else
    {
        int winner = -1;
        for (int s = 0; s < g.batchCount; s++)
        {
            long long pid = (long long)s * g.nOrd + ord;
            if (k.maxDepth <= 0) continue;                                     // no bounce loop: Li = 0 (:228), nothing was staged
            Lframe = Lframe + safe_color(W.sampleLi.ld3(0, pid));             // :320, in sample order; every live path stores its Li exactly once, where it ends
            if (W.stage.ldi(G_FLAG, pid)) winner = s;                          // last sample that reached a diffuse vertex
        }
        if (winner >= 0)
        {
            long long pid = (long long)winner * g.nOrd + ord;
            const F3 wi = W.stage.ld3(G_WI, pid);
            const float w = W.stage.ldf(G_W, pid);
            const int lid = W.stage.ldi(G_LID, pid);
            resCur.L[index] = to3(res_L(k, wi, w, lid)); resCur.wi[index] = to3(wi);
            resCur.pdf[index] = W.stage.ldf(G_PDF, pid); resCur.w[index] = w;
            resCur.wSum[index] = W.stage.ldf(G_WSUM, pid); resCur.lightId[index] = lid;
            resCur.m[index] = W.stage.ldi(G_M, pid);
        }
    }
    if (!g.lastBatch) { W.accum.st3(0, ord, Lframe); return; }
    F3 Lout = Lframe * (1.0f / (float)hrt_imax(1, k.spp));
    if (fb.radiance) fb.radiance[index] = to3(Lout);
    fb.color[index] = pack_rgba8(Lout);
    fb.depth[index] = cam_distance(k, ld3(&gb.worldPos[index]));
    fb.objectId[index] = gb.objId[index];
}

} // namespace hrt
