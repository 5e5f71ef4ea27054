// hrt_device.hpp -- gfx950 device code of the render hot path (general kernels).
//
// Re-implements, as hand-written HIP, what the reference's two ILGPU kernels compute:
//   PrimaryVisibilityKernel   Engine/RTRay.cs:188-201
//   PathTraceKernel           Engine/RTRay.cs:203-325   (+ ReSTIR-DI :330-543, BSDF :548-671)
//   TraceClosest / ShadowOcclusion and the BLAS walkers, intersectors, texture samplers
//                             Engine/SceneDeviceViews.cs:30-558
//   Ray / RNG                 Engine/RTUtils.cs:6-138,  Float3 Engine/Float3.cs
// (paths relative to /root/reference/ILGPU_Raytracing/).
//
// Not a translation: one pixel per lane of a 64-wide wave, but the bounce loop is rebuilt
// so that every lane of the wave reaches ONE closest-hit traversal site and ONE any-hit
// site per bounce (the reference inlines TraceClosest at three sites and ReSTIR at two,
// which on a 64-lane SIMT machine serialises mirror / glass / diffuse lanes through three
// copies of the traversal loop).  Hit normals and sphere texture lookups are deferred to
// the end of a traversal (pure functions of the winning hit), the two-level walk keeps
// only scalars live across the inner loop, and all per-pixel arrays are addressed with
// 12-byte-per-lane contiguous accesses.  Arithmetic follows include/hrt_math.h exactly
// (binary32, no contraction, IEEE div/sqrt), so results are bit-identical to the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hrt_types.h"
#include "../../include/hrt_math.h"

#define HRT_D __device__ __forceinline__

// Tuning aid (variant build -DHRT_PT_STATS, tools/pt_stats.py): how often each section of the fused kernel runs and with how
// many live lanes.  g_pt_stats[2 i] = executions of section i (per wave), [2 i + 1] = live lanes summed.
#ifdef HRT_PT_STATS
__device__ unsigned long long g_pt_stats[64];
#define PSTAT(i) { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true); \
                   if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) { atomicAdd(&g_pt_stats[2 * (i)], 1ull); atomicAdd(&g_pt_stats[2 * (i) + 1], (unsigned long long)__popcll(m_)); } }
#else
#define PSTAT(i)
#endif

namespace hrt {

// ------------------------------------------------------------------ small vector type
struct F3 { float x, y, z; };
HRT_D F3 mk3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
HRT_D F3 ld3(const hrt_float3* p) { hrt_float3 v = *p; return mk3(v.X, v.Y, v.Z); }
HRT_D F3 cv3(const hrt_float3& v) { return mk3(v.X, v.Y, v.Z); }
HRT_D hrt_float3 to3(F3 v) { hrt_float3 r; r.X = v.x; r.Y = v.y; r.Z = v.z; return r; }
HRT_D F3 operator+(F3 a, F3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
HRT_D F3 operator-(F3 a, F3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
HRT_D F3 operator*(F3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
HRT_D F3 operator*(F3 a, F3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
HRT_D F3 operator-(F3 a) { return mk3(-a.x, -a.y, -a.z); }
HRT_D float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HRT_D F3 cross(F3 a, F3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// IEEE square root for arguments that are +0, +inf or normal numbers >= 2^-96: the same v_sqrt_f32 + two one-ulp
// corrections hipcc emits for sqrtf (-fhip-fp32-correctly-rounded-divide-sqrt), minus the rescaling of tiny arguments and
// the class test for infinities -- 7 of its 17 instructions.  tests/test_math_gpu.py compares it with hrt_sqrt over every
// argument the sampler below can produce and over [1e-20, 1e20].
template <bool NONZERO = false>       // NONZERO: the caller knows x != 0 (the zero select is dropped)
HRT_D float sqrt_normal_range(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sDn = __int_as_float(__float_as_int(s) - 1), sUp = __int_as_float(__float_as_int(s) + 1);
    const float rDn = __builtin_fmaf(-sDn, s, x), rUp = __builtin_fmaf(-sUp, s, x);
    float r = (rDn <= 0.f) ? sDn : s;
    r = (rUp > 0.f) ? sUp : r;
    return (!NONZERO && x == 0.f) ? x : r;
}
// hrt_rsqrt(x) = 1 / sqrt(x), both IEEE, for x = max(1e-20, .): x in [2^-67, +inf], never NaN -- the argument of every
// Normalize (Float3.cs:91-95).  sqrt: the trimmed form above.  1 / s with s in [2^-34, 2^64]: the division sequence hipcc
// emits for 1.0f / s minus what these operands cannot need: v_div_scale (no operand is scaled when the denominator and the
// quotient are normal numbers far from the exponent limits), the scaling half of v_div_fmas and the special-case v_div_fixup,
// of which only s = +inf (|v|^2 overflowed) remains and is handled by the select.  16 instructions instead of 28;
// tests/test_math_gpu.py compares it with hrt_rsqrt on the device for EVERY float in the domain.
template <bool FINITE = false>        // FINITE: the caller knows x < +inf (the select for s = +inf is dropped)
HRT_D float rsqrt_clamped(float x)
{
    const float s = sqrt_normal_range<true>(x);
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float e0 = __builtin_fmaf(-s, r0, 1.f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float e1 = __builtin_fmaf(-s, r1, 1.f);           // numerator 1: n * r1 == r1
    const float q1 = __builtin_fmaf(e1, r1, r1);
    const float e2 = __builtin_fmaf(-s, q1, 1.f);
    const float q = __builtin_fmaf(e2, r1, q1);
    return (!FINITE && s == __builtin_inff()) ? 0.f : q;
}
template <bool FINITE = false>
HRT_D F3 normalize(F3 v)   // Float3.cs:91-95
{
    float inv = rsqrt_clamped<FINITE>(hrt_fmax(1e-20f, v.x * v.x + v.y * v.y + v.z * v.z));
    return mk3(v.x * inv, v.y * inv, v.z * inv);
}
HRT_D F3 inv_dir(F3 d)     // RTRay.cs:548-549
{
    return mk3(1.f / (d.x != 0.f ? d.x : 1e-8f), 1.f / (d.y != 0.f ? d.y : 1e-8f), 1.f / (d.z != 0.f ? d.z : 1e-8f));
}

struct Ray { F3 o, d, inv; };

constexpr float kPI = 3.14159265358979323846f;
constexpr float kINV_PI = 0.31830988618379067154f;
constexpr float kEPS_N = 0.0025f;
constexpr float kEPS_MIN = 1e-6f;

// ------------------------------------------------------------------ RNG (RTUtils.cs:20-138)
struct Rng {
    uint32_t s;
    HRT_D uint32_t next_u()
    {
        uint32_t x = s;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        // The reference's "x != 0 ? x : 1" (RTUtils.cs:40) can never fire: each of the three steps is an invertible linear map of
        // GF(2)^32, so x == 0 only from s == 0, and s is never 0 (Create maps a zero seed to 1, the seed mixer ORs in 1, :28,:96).
        // tests/test_oracle_kat.py checks the rank of the map.
        s = x;
        return s;
    }
    HRT_D float next_f() { return (float)(next_u() & 0x00FFFFFFu) * (1.0f / 16777216.0f); }
};
HRT_D uint32_t rotl32(uint32_t v, int r) { return (v << (r & 31)) | (v >> ((32 - r) & 31)); }
HRT_D uint32_t splitmix32(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x ^= (x >> 31);
    return (uint32_t)(x ^ (x >> 32));
}
HRT_D uint32_t pcg_permute(uint32_t x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
HRT_D uint32_t hash32(uint32_t x)   // RTUtils.cs:77-84 == RTRay.cs:637-641
{
    x ^= x >> 17; x *= 0xED5AD4BBu; x ^= x >> 11; x *= 0xAC4C1B51u; x ^= x >> 15; x *= 0x31848BABu; x ^= x >> 14;
    return x;
}
// RTUtils.cs:116-137.  The (px,py,frame,lockNoise) part of the seed is sample-independent:
// seed_half0 is computed once per pixel, the sample only enters lane1a.
struct SeedBase { uint32_t s0; uint32_t lane1b; uint32_t rot_px; };
HRT_D SeedBase seed_base(uint32_t px, uint32_t py, int frame, uint32_t salt, int lockNoise)
{
    uint32_t f = (lockNoise != 0) ? 0u : (uint32_t)frame;
    uint32_t ln = (uint32_t)lockNoise;
    uint32_t lnMix0 = (lockNoise != 0) ? (hash32(ln) ^ (ln * 0x1B873593u)) : 0u;
    uint32_t lnMix1 = (lockNoise != 0) ? (rotl32(ln, 7) * 0x85EBCA6Bu) : 0u;
    uint32_t lane0a = px ^ 0xB5297A4Du;
    uint32_t lane0b = (py * 0x68E31DA4u) ^ (f * 0x9E3779B1u + 0x85EBCA6Bu) ^ lnMix0;
    SeedBase b;
    b.s0 = splitmix32((((uint64_t)lane0a << 32) | lane0b) ^ 0xD1B54A32D192ED03ULL);
    b.lane1b = ((salt ^ 0x27D4EB2Fu) + rotl32(py, 8)) ^ lnMix1;
    b.rot_px = rotl32(px, 16);
    return b;
}
HRT_D Rng rng_for_sample(const SeedBase& b, uint32_t sample)
{
    uint32_t lane1a = (sample ^ 0xC2B2AE35u) + b.rot_px;
    uint32_t s1 = splitmix32((((uint64_t)lane1a << 32) | b.lane1b) ^ 0x94D049BB133111EBULL);
    uint32_t s = pcg_permute(b.s0 ^ (rotl32(s1, 13) + 0x9E3779B1u));
    s |= 1u;
    Rng r; r.s = (s == 0u) ? 1u : s;
    return r;
}

// ------------------------------------------------------------------ device views
struct DScene {          // the 15 arrays of SceneDeviceViews.cs:13-27 (device pointers)
    const hrt_bvh_node* tlasNodes;
    const int32_t* tlasInst;
    const hrt_instance* instances;
    const hrt_bvh_node* blasNodes;
    const int32_t* spherePrimIdx;
    const hrt_sphere* spheres;
    const int32_t* triPrimIdx;
    const hrt_float3* meshPositions;
    const hrt_mesh_tri* meshTris;
    const hrt_float2* meshTexcoords;
    const hrt_mesh_tri_uv* meshTriUVs;
    const int32_t* triMatIndex;
    const hrt_material* materials;
    const hrt_rgba32* texels;
    const hrt_tex_info* texInfos;
    int32_t n_texInfos;  // texInfos.Length (>= 1, Scene.cs:370-377)
};

struct DGBuffer {        // RTRay.cs:80-87
    hrt_float3 *worldPos, *normalWS, *baseColor;
    int32_t *matId, *objId, *hitMask;
};
struct DReservoir {      // RTRay.cs:23-31
    hrt_float3 *L, *wi;
    float *pdf, *w, *wSum;
    int32_t *m, *lightId;
};
struct DFramebuffer {    // RTRay.cs:51-56 (+ radiance)
    int32_t* color; float* depth; int32_t* objectId; int32_t* cameraId; hrt_float3* radiance;
};

struct Counters {        // per-lane, 32 bit; wave-reduced at kernel exit
    uint32_t v[10];
};
enum { C_RAYS_CLOSEST, C_RAYS_SHADOW, C_NODE_VISITS, C_LEAF_INST, C_SPHERE_TESTS, C_TRI_TESTS, C_TRI_MT_HITS, C_TRI_ACCEPTED, C_REUSE_IMPORTS, C_DIFFUSE_VERTS };

template <bool COUNT> struct Cnt {
    Counters c;
    HRT_D Cnt()
    {
        if (COUNT) {
#pragma unroll
            for (int i = 0; i < 10; i++) c.v[i] = 0;
        }
    }
    HRT_D void inc(int i) { if (COUNT) c.v[i]++; }
    // one atomicAdd per counter per wave (Guideline 12)
    HRT_D void flush(unsigned long long* g)
    {
        if (!COUNT) return;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            uint32_t v = c.v[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&g[i], (unsigned long long)v);
        }
    }
};

// ------------------------------------------------------------------ intersectors
// SceneDeviceViews.cs:496-514
HRT_D bool hit_aabb(const Ray& r, const hrt_bvh_node* n, float tMin, float tMax)
{
    float t1 = (n->boundsMin.X - r.o.x) * r.inv.x;
    float t2 = (n->boundsMax.X - r.o.x) * r.inv.x;
    float tmin = hrt_fmin(t1, t2);
    float tmax = hrt_fmax(t1, t2);
    t1 = (n->boundsMin.Y - r.o.y) * r.inv.y;
    t2 = (n->boundsMax.Y - r.o.y) * r.inv.y;
    tmin = hrt_fmax(tmin, hrt_fmin(t1, t2));
    tmax = hrt_fmin(tmax, hrt_fmax(t1, t2));
    t1 = (n->boundsMin.Z - r.o.z) * r.inv.z;
    t2 = (n->boundsMax.Z - r.o.z) * r.inv.z;
    tmin = hrt_fmax(tmin, hrt_fmin(t1, t2));
    tmax = hrt_fmin(tmax, hrt_fmax(t1, t2));
    return tmax >= hrt_fmax(tmin, tMin) && tmin <= tMax;
}

// SceneDeviceViews.cs:517-533 without the normal (:534-535 is a pure function of t: deferred)
// a = dot(r.d, r.d) is a property of the ray: callers that test several spheres with one ray compute it once
HRT_D bool hit_sphere_ta(const Ray& r, float a, F3 c, float radius, float& t)
{
    F3 oc = r.o - c;
    float b = 2.f * dot(oc, r.d);
    float cc = dot(oc, oc) - radius * radius;
    float disc = b * b - 4.f * a * cc;
    if (disc < 0.f) return false;
    PSTAT(9);
    float sq = hrt_sqrt(disc);
    float tt = (-b - sq) / (2.f * a);
    if (tt < 0.001f) {
        PSTAT(10);
        tt = (-b + sq) / (2.f * a);
        if (tt < 0.001f) return false;
    }
    t = tt;
    return true;
}
HRT_D bool hit_sphere_t(const Ray& r, F3 c, float radius, float& t) { return hit_sphere_ta(r, dot(r.d, r.d), c, radius, t); }
HRT_D F3 sphere_normal(const Ray& r, F3 c, float t) { return normalize((r.o + r.d * t) - c); }   // :534-535

// SceneDeviceViews.cs:540-555 without the normal (:556 deferred)
HRT_D bool hit_tri_t(const Ray& r, F3 v0, F3 v1, F3 v2, float& t, float& bu, float& bv)
{
    F3 e1 = v1 - v0;
    F3 e2 = v2 - v0;
    F3 p = cross(r.d, e2);
    float det = dot(e1, p);
    if (hrt_abs(det) < 1e-8f) return false;
    float invDet = 1.f / det;
    F3 tv = r.o - v0;
    bu = dot(tv, p) * invDet;
    if (bu < 0.f || bu > 1.f) return false;
    F3 q = cross(tv, e1);
    bv = dot(r.d, q) * invDet;
    if (bv < 0.f || bu + bv > 1.f) return false;
    t = dot(e2, q) * invDet;
    if (t <= 0.f) return false;
    return true;
}

// SceneDeviceViews.cs:475-493
HRT_D F3 xform_point(const hrt_affine3x4& m, F3 p)
{
    return mk3(m.m00 * p.x + m.m01 * p.y + m.m02 * p.z + m.m03, m.m10 * p.x + m.m11 * p.y + m.m12 * p.z + m.m13, m.m20 * p.x + m.m21 * p.y + m.m22 * p.z + m.m23);
}
HRT_D F3 xform_vector(const hrt_affine3x4& m, F3 v)
{
    return mk3(m.m00 * v.x + m.m01 * v.y + m.m02 * v.z, m.m10 * v.x + m.m11 * v.y + m.m12 * v.z, m.m20 * v.x + m.m21 * v.y + m.m22 * v.z);
}

// ------------------------------------------------------------------ texture sampling (SceneDeviceViews.cs:330-472)
struct Tex {
    const DScene& S;
    HRT_D Tex(const DScene& s) : S(s) {}
    HRT_D hrt_rgba32 raw(const hrt_tex_info& info, int x, int y) const
    {
        int w = info.Width, h = info.Height;
        hrt_rgba32 z = {0, 0, 0, 0};
        if (w <= 0 || h <= 0) return z;
        int sx = hrt_imax(0, hrt_imin(w - 1, x));
        int sy = hrt_imax(0, hrt_imin(h - 1, y));
        return S.texels[info.Offset + sy * w + sx];
    }
    static HRT_D float luma(hrt_rgba32 p)
    {
        float r = p.R * (1.f / 255.f), g = p.G * (1.f / 255.f), b = p.B * (1.f / 255.f);
        return 0.2126f * r + 0.7152f * g + 0.0722f * b;
    }
    static HRT_D F3 rgb(hrt_rgba32 p) { return mk3(p.R * (1.f / 255.f), p.G * (1.f / 255.f), p.B * (1.f / 255.f)); }

    struct Tap { int x0, y0, x1, y1; float tx, ty; };
    static HRT_D Tap taps(const hrt_tex_info& info, float u, float v)
    {
        int w = info.Width, h = info.Height;
        float fu = u - hrt_floor(u);
        float fv = 1.f - (v - hrt_floor(v));
        float x = fu * (float)(w - 1);
        float y = fv * (float)(h - 1);
        Tap t;
        t.x0 = hrt_f2i(hrt_floor(x));
        t.y0 = hrt_f2i(hrt_floor(y));
        t.x1 = hrt_imin(w - 1, t.x0 + 1);
        t.y1 = hrt_imin(h - 1, t.y0 + 1);
        t.tx = x - (float)t.x0;
        t.ty = y - (float)t.y0;
        return t;
    }
    // :358-385 and the RGB part of :431-472 (same arithmetic)
    HRT_D F3 linear_rgb(const hrt_tex_info& info, float u, float v) const
    {
        if (info.Width <= 0 || info.Height <= 0) return mk3(1.f, 1.f, 1.f);
        Tap t = taps(info, u, v);
        F3 c00 = rgb(raw(info, t.x0, t.y0)), c10 = rgb(raw(info, t.x1, t.y0));
        F3 c01 = rgb(raw(info, t.x0, t.y1)), c11 = rgb(raw(info, t.x1, t.y1));
        F3 cx0 = c00 * (1.f - t.tx) + c10 * t.tx;
        F3 cx1 = c01 * (1.f - t.tx) + c11 * t.tx;
        return cx0 * (1.f - t.ty) + cx1 * t.ty;
    }
    // :388-415
    HRT_D float mask_linear(const hrt_tex_info& info, float u, float v) const
    {
        if (info.Width <= 0 || info.Height <= 0) return 1.f;
        Tap t = taps(info, u, v);
        float a00 = luma(raw(info, t.x0, t.y0)), a10 = luma(raw(info, t.x1, t.y0));
        float a01 = luma(raw(info, t.x0, t.y1)), a11 = luma(raw(info, t.x1, t.y1));
        float ax0 = a00 * (1.f - t.tx) + a10 * t.tx;
        float ax1 = a01 * (1.f - t.tx) + a11 * t.tx;
        return ax0 * (1.f - t.ty) + ax1 * t.ty;
    }
    // :418-428
    HRT_D float mask_point(const hrt_tex_info& info, float u, float v) const
    {
        int w = info.Width, h = info.Height;
        if (w <= 0 || h <= 0) return 1.f;
        float fu = u - hrt_floor(u);
        float fv = 1.f - (v - hrt_floor(v));
        int x = hrt_f2i(hrt_round(fu * (float)(w - 1)));
        int y = hrt_f2i(hrt_round(fv * (float)(h - 1)));
        return luma(raw(info, x, y));
    }
};

// ------------------------------------------------------------------ closest-hit record
struct Hit {
    float t;         // world t, 1e30 = miss
    F3 n;            // world normal (Normalize(objectToWorld * nObj))
    F3 albedo;
    int objId, shade;
    float ior;
};

// ------------------------------------------------------------------ tracer over the reference's own array layout
// (44-byte nodes, 144-byte instance records, index indirections).  Handles every scene the
// boundary accepts; TracerPacked (hrt_trace_packed.hpp) is the fast path for the same walk.
struct TracerRef {
    DScene S;

// TraceClosest + TraverseBLAS_* (SceneDeviceViews.cs:30-86,124-237)
template <bool COUNT>
HRT_D bool closest(const Ray& wray, Hit& best, Cnt<COUNT>& C) const
{
    C.inc(C_RAYS_CLOSEST);
    best.t = 1e30f; best.n = mk3(0.f, 0.f, 0.f); best.albedo = mk3(1.f, 1.f, 1.f); best.objId = -1; best.shade = 0; best.ior = 1.f;
    Tex tex(S);
    int cur = 0;
    while (cur != -1)
    {
        const hrt_bvh_node* n = &S.tlasNodes[cur];
        C.inc(C_NODE_VISITS);
        int skip = n->skipIndex;
        if (hit_aabb(wray, n, 0.001f, best.t))
        {
            int count = n->count;
            if (count > 0)
            {
                int first = n->first;
                for (int i = first; i < first + count; i++)
                {
                    const hrt_instance* inst = &S.instances[S.tlasInst[i]];
                    C.inc(C_LEAF_INST);
                    Ray iray;
                    iray.o = xform_point(inst->worldToObject, wray.o);
                    iray.d = xform_vector(inst->worldToObject, wray.d);
                    iray.inv = inv_dir(iray.d);
                    float us = inst->uniformScale;
                    float scale = us > 0.f ? us : 1.f;
                    int blasStart = inst->blasRoot;
                    int blasEnd = blasStart + inst->blasNodeCount;
                    bool isSphere = inst->type == HRT_BLAS_SPHERESET;

                    // ---- BLAS walk: only the winning primitive id and t stay live
                    float tObj = 1e30f;
                    int prim = -1;           // sphere index or triangle index of the accepted hit
                    float hbu = 0.f, hbv = 0.f;
                    F3 triAlbedo = mk3(0.85f, 0.85f, 0.85f);
                    bool triFlip = false;
                    int bcur = blasStart;
                    while (bcur != -1 && bcur < blasEnd)
                    {
                        const hrt_bvh_node* bn = &S.blasNodes[bcur];
                        C.inc(C_NODE_VISITS);
                        int bskip = bn->skipIndex;
                        if (hit_aabb(iray, bn, 0.001f, tObj))
                        {
                            int bcount = bn->count;
                            if (bcount > 0)
                            {
                                int bfirst = bn->first;
                                for (int j = bfirst; j < bfirst + bcount; j++)
                                {
                                    if (isSphere)
                                    {
                                        int p = S.spherePrimIdx[j];
                                        const hrt_sphere* sp = &S.spheres[p];
                                        C.inc(C_SPHERE_TESTS);
                                        float t;
                                        if (hit_sphere_t(iray, cv3(sp->center), sp->radius, t) && t > 0.001f && t < tObj) { tObj = t; prim = p; }
                                    }
                                    else
                                    {
                                        int ti = S.triPrimIdx[j];
                                        hrt_mesh_tri tri = S.meshTris[ti];
                                        F3 v0 = ld3(&S.meshPositions[tri.i0]), v1 = ld3(&S.meshPositions[tri.i1]), v2 = ld3(&S.meshPositions[tri.i2]);
                                        C.inc(C_TRI_TESTS);
                                        float t, bu, bv;
                                        if (hit_tri_t(iray, v0, v1, v2, t, bu, bv))
                                        {
                                            C.inc(C_TRI_MT_HITS);
                                            if (t > 0.001f && t < tObj)
                                            {
                                                C.inc(C_TRI_ACCEPTED);
                                                const hrt_material* mat = &S.materials[S.triMatIndex[ti]];
                                                hrt_mesh_tri_uv tuv = S.meshTriUVs[ti];
                                                hrt_float2 t0 = S.meshTexcoords[tuv.t0], t1 = S.meshTexcoords[tuv.t1], t2 = S.meshTexcoords[tuv.t2];
                                                float w = 1.f - bu - bv;
                                                float uu = t0.X * w + t1.X * bu + t2.X * bv;
                                                float vv = t0.Y * w + t1.Y * bu + t2.Y * bv;
                                                float alpha = 1.f;
                                                F3 kd = cv3(mat->Kd);
                                                int dti = mat->DiffuseTexIndex, ati = mat->AlphaTexIndex;
                                                if (mat->HasDiffuseMap != 0 && dti >= 0 && dti < S.n_texInfos) kd = tex.linear_rgb(S.texInfos[dti], uu, vv);
                                                if (mat->HasAlphaMap != 0 && ati >= 0 && ati < S.n_texInfos) alpha = tex.mask_linear(S.texInfos[ati], uu, vv);
                                                if (!(alpha < mat->AlphaCutoff))
                                                {
                                                    tObj = t; prim = ti; hbu = bu; hbv = bv; triAlbedo = kd;
                                                    // normal = Normalize(Cross(e1,e2)); flipped iff TwoSided && Dot(n, dir) > 0
                                                    F3 nn = normalize(cross(v1 - v0, v2 - v0));
                                                    triFlip = (mat->TwoSided != 0) && (dot(nn, iray.d) > 0.f);
                                                }
                                            }
                                        }
                                    }
                                }
                                bcur = bskip;
                            }
                            else bcur = bn->left;
                        }
                        else bcur = bskip;
                    }
                    (void)hbu; (void)hbv;

                    if (tObj < 1e29f)
                    {
                        float tWorld = tObj / scale;
                        if (tWorld < best.t)
                        {
                            F3 nObj, alb; int shade = 0; float ior = 1.f; int objId;
                            if (isSphere)
                            {
                                const hrt_sphere* sp = &S.spheres[prim];
                                nObj = sphere_normal(iray, cv3(sp->center), tObj);
                                F3 kd = cv3(sp->material.Kd);
                                alb = (kd.x == 0.f && kd.y == 0.f && kd.z == 0.f) ? cv3(sp->albedo) : kd;
                                int dti = sp->material.DiffuseTexIndex;
                                if (sp->material.HasDiffuseMap != 0 && dti >= 0 && dti < S.n_texInfos)
                                {
                                    float u = 0.5f + hrt_atan2(nObj.z, nObj.x) / (2.f * kPI);
                                    float v = hrt_acos(hrt_fmin(1.f, hrt_fmax(-1.f, nObj.y))) / kPI;
                                    alb = tex.linear_rgb(S.texInfos[dti], u, v);
                                }
                                shade = sp->shading;
                                float sior = sp->ior;
                                ior = sior > 0.f ? sior : 1.f;
                                objId = -1;
                            }
                            else
                            {
                                hrt_mesh_tri tri = S.meshTris[prim];
                                F3 v0 = ld3(&S.meshPositions[tri.i0]), v1 = ld3(&S.meshPositions[tri.i1]), v2 = ld3(&S.meshPositions[tri.i2]);
                                nObj = normalize(cross(v1 - v0, v2 - v0));
                                if (triFlip) nObj = nObj * -1.f;
                                alb = triAlbedo;
                                objId = prim;
                            }
                            best.t = tWorld;
                            best.n = normalize(xform_vector(inst->objectToWorld, nObj));
                            best.albedo = alb; best.objId = objId; best.shade = shade; best.ior = ior;
                        }
                    }
                }
                cur = skip;
            }
            else cur = n->left;
        }
        else cur = skip;
    }
    return best.t < 1e29f;
}

// ShadowOcclusion + AnyHit_* (SceneDeviceViews.cs:89-121,240-327)
template <bool COUNT>
HRT_D bool occluded(const Ray& wray, float tMaxWorld, Cnt<COUNT>& C) const
{
    C.inc(C_RAYS_SHADOW);
    Tex tex(S);
    int cur = 0;
    while (cur != -1)
    {
        const hrt_bvh_node* n = &S.tlasNodes[cur];
        C.inc(C_NODE_VISITS);
        int skip = n->skipIndex;
        if (hit_aabb(wray, n, 0.001f, tMaxWorld))
        {
            int count = n->count;
            if (count > 0)
            {
                int first = n->first;
                for (int i = first; i < first + count; i++)
                {
                    const hrt_instance* inst = &S.instances[S.tlasInst[i]];
                    C.inc(C_LEAF_INST);
                    Ray iray;
                    iray.o = xform_point(inst->worldToObject, wray.o);
                    iray.d = xform_vector(inst->worldToObject, wray.d);
                    iray.inv = inv_dir(iray.d);
                    float us = inst->uniformScale;
                    float scale = us > 0.f ? us : 1.f;
                    float tMaxObj = tMaxWorld * scale;
                    int blasStart = inst->blasRoot;
                    int blasEnd = blasStart + inst->blasNodeCount;
                    bool isSphere = inst->type == HRT_BLAS_SPHERESET;

                    int bcur = blasStart;
                    while (bcur != -1 && bcur < blasEnd)
                    {
                        const hrt_bvh_node* bn = &S.blasNodes[bcur];
                        C.inc(C_NODE_VISITS);
                        int bskip = bn->skipIndex;
                        if (hit_aabb(iray, bn, 0.001f, tMaxObj))
                        {
                            int bcount = bn->count;
                            if (bcount > 0)
                            {
                                int bfirst = bn->first;
                                for (int j = bfirst; j < bfirst + bcount; j++)
                                {
                                    if (isSphere)
                                    {
                                        const hrt_sphere* sp = &S.spheres[S.spherePrimIdx[j]];
                                        C.inc(C_SPHERE_TESTS);
                                        float t;
                                        if (hit_sphere_t(iray, cv3(sp->center), sp->radius, t) && t > 0.001f && t < tMaxObj) return true;
                                    }
                                    else
                                    {
                                        int ti = S.triPrimIdx[j];
                                        hrt_mesh_tri tri = S.meshTris[ti];
                                        F3 v0 = ld3(&S.meshPositions[tri.i0]), v1 = ld3(&S.meshPositions[tri.i1]), v2 = ld3(&S.meshPositions[tri.i2]);
                                        C.inc(C_TRI_TESTS);
                                        float t, bu, bv;
                                        if (hit_tri_t(iray, v0, v1, v2, t, bu, bv))
                                        {
                                            if (t <= 0.001f || t >= tMaxObj) continue;
                                            C.inc(C_TRI_MT_HITS);
                                            const hrt_material* mat = &S.materials[S.triMatIndex[ti]];
                                            int ati = mat->AlphaTexIndex;
                                            if (mat->HasAlphaMap != 0 && ati >= 0 && ati < S.n_texInfos)
                                            {
                                                C.inc(C_TRI_ACCEPTED);
                                                hrt_mesh_tri_uv tuv = S.meshTriUVs[ti];
                                                hrt_float2 t0 = S.meshTexcoords[tuv.t0], t1 = S.meshTexcoords[tuv.t1], t2 = S.meshTexcoords[tuv.t2];
                                                float w = 1.f - bu - bv;
                                                float uu = t0.X * w + t1.X * bu + t2.X * bv;
                                                float vv = t0.Y * w + t1.Y * bu + t2.Y * bv;
                                                hrt_tex_info ainfo = S.texInfos[ati];
                                                float aPoint = tex.mask_point(ainfo, uu, vv);
                                                float cutoff = mat->AlphaCutoff;
                                                const float Band = 0.10f;
                                                if (aPoint < cutoff - Band) continue;
                                                if (aPoint >= cutoff + Band) return true;
                                                float aLin = tex.mask_linear(ainfo, uu, vv);
                                                if (aLin < cutoff) continue;
                                            }
                                            return true;
                                        }
                                    }
                                }
                                bcur = bskip;
                            }
                            else bcur = bn->left;
                        }
                        else bcur = bskip;
                    }
                }
                cur = skip;
            }
            else cur = n->left;
        }
        else cur = skip;
    }
    return false;
}
};   // struct TracerRef

// ------------------------------------------------------------------ frame constants
struct FrameK {          // scalar part of GBufferParams / IntegratorParams (RTRay.cs:112-146)
    int32_t width, height, frame;
    int32_t row_begin, row_end, strip_n, strip_i;   // tile: 8-row strips s of [row_begin,row_end) with s % strip_n == strip_i
    hrt_camera cam, prevCam;
    hrt_float3 dirLightDir, dirLightRadiance, skyTop, skyBottom;
    hrt_float3 dirLightN;    // Normalize(dirLightDir), evaluated once per frame on the host (hrt_runtime.hip, frame_k)
    int32_t debugCamSeq, enableTemporal, enableSpatial, rngLockNoise, spp, maxDepth;
};

HRT_D Ray primary_ray(const FrameK& k, int x, int y)   // RTRay.cs:120-126 + RTUtils.cs:13-17
{
    float u = ((float)x + 0.5f) / (float)hrt_imax(1, k.width);
    float v = ((float)y + 0.5f) / (float)hrt_imax(1, k.height);
    F3 dir = normalize(cv3(k.cam.lowerLeft) + cv3(k.cam.horizontal) * u + cv3(k.cam.vertical) * v - cv3(k.cam.origin));
    Ray r; r.o = cv3(k.cam.origin); r.d = dir; r.inv = inv_dir(dir);
    return r;
}
HRT_D F3 sky(const FrameK& k, F3 dir)   // RTRay.cs:164-168
{
    float tbg = 0.5f * (dir.y + 1.0f);
    return cv3(k.skyBottom) * (1.f - tbg) + cv3(k.skyTop) * tbg;
}
HRT_D float cam_distance(const FrameK& k, F3 p)   // RTRay.cs:158-162
{
    F3 d = p - cv3(k.cam.origin);
    return hrt_sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
}
HRT_D int float_to_i16(float x)   // RTRay.cs:609-613
{
    float cl = hrt_fmax(0.f, hrt_fmin(65535.f, x * 1000.f));
    return hrt_f2i(cl) & 0xFFFF;
}
HRT_D int to_byte(float x) { return hrt_f2i(255.99f * hrt_fmin(1.f, hrt_fmax(0.f, x))); }   // RTRay.cs:72-76
HRT_D int pack_rgba8(F3 c)   // RTRay.cs:66-70
{
    return (int)((255u << 24) | ((uint32_t)to_byte(c.x) << 16) | ((uint32_t)to_byte(c.y) << 8) | (uint32_t)to_byte(c.z));
}
HRT_D F3 safe_color(F3 c)    // RTRay.cs:646-655
{
    float x = hrt_isfinite(c.x) ? c.x : 0.f, y = hrt_isfinite(c.y) ? c.y : 0.f, z = hrt_isfinite(c.z) ? c.z : 0.f;
    return mk3(hrt_fmin(1e6f, hrt_fmax(-1e6f, x)), hrt_fmin(1e6f, hrt_fmax(-1e6f, y)), hrt_fmin(1e6f, hrt_fmax(-1e6f, z)));
}
HRT_D Ray ray_with_normal_offset(F3 origin, F3 n, F3 dir)   // RTRay.cs:552-558
{
    F3 d = normalize(dir);
    float s = dot(n, d) >= 0.f ? 1.f : -1.f;
    Ray r; r.o = origin + n * (kEPS_N * s); r.d = d; r.inv = inv_dir(d);
    return r;
}
HRT_D float luminance(F3 c) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }   // RTRay.cs:627
HRT_D float cos_hemi_pdf(F3 n, F3 wi) { return hrt_fmax(0.f, dot(n, wi)) * kINV_PI; }     // RTRay.cs:630-634

// RTRay.cs:586-606.  The tangent frame depends only on n: callers that draw several
// directions around one normal build it once.
struct Frame { F3 t, b, n; };
HRT_D Frame make_frame(F3 n)
{
    F3 up = hrt_abs(n.y) < 0.999f ? mk3(0.f, 1.f, 0.f) : mk3(1.f, 0.f, 0.f);
    Frame f; f.n = n;
    f.t = normalize(cross(up, n));
    f.b = cross(n, f.t);
    return f;
}
HRT_D F3 sample_hemisphere_cosine(const Frame& f, Rng& rng)
{
    float r1 = rng.next_f(), r2 = rng.next_f();          // k / 2^24: r2 is 0 or >= 2^-24, 1 - r2 is >= 2^-24
    float phi = 2.f * kPI * r1;
    float cosTheta = sqrt_normal_range<true>(1.f - r2);
    float sinTheta = sqrt_normal_range(r2);
    float sn, cs;
    hrt_sincos_nonneg(phi, &sn, &cs);            // == hrt_sin(phi), hrt_cos(phi) for phi >= 0; both polynomials once, no divergent branch
    float x = cs * sinTheta;
    float y = sn * sinTheta;
    float z = cosTheta;
    F3 v = f.t * x + f.b * y + f.n * z;
    // |x|, |y|, z <= 1 and the frame vectors are outputs of normalize / a cross product of two of them (components <= 2, or NaN):
    // |v|^2 < 100 or NaN, and max(1e-20, NaN) = 1e-20 -- the sum of squares is never +inf
    return normalize<true>(v);
}

// ------------------------------------------------------------------ ReSTIR-DI (RTRay.cs:330-543)
// The reservoir's L and pdf fields are not carried through the candidate stream: every update site stores values that are
// functions of the accepted (wi, lightId) and the vertex normal alone --
//   L   = lightId == 2 ? dirLightRadiance : SkyWeighted(wi)                                    (:455,:467,:421)
//   pdf = lightId == 2 ? max(EPS, mixDelta) : max(EPS, CosHemispherePdf(n, wi) * mixLocal)     (:466,:424-426; for a local
//         candidate :453-454 writes max(EPS, max(EPS, c) * mixLocal), the same number: c >= EPS makes the two expressions
//         identical, c < EPS makes both EPS because mixLocal < 1)
// -- so res_L / res_pdf evaluate them once, where a reservoir is stored.  "Never accepted" (L = 0, pdf = 0, the defaults of
// :330-334) is w == 0: an accepted score is > 0 (acceptP = score / newSum has to exceed a draw >= 0 with newSum > 0).
struct Res { F3 wi; float w, wSum; int m, lightId; };

HRT_D F3 res_L(const FrameK& k, F3 wi, float w, int lightId)
{
    return w > 0.f ? (lightId == 2 ? cv3(k.dirLightRadiance) : sky(k, wi)) : mk3(0.f, 0.f, 0.f);
}
HRT_D float res_pdf(F3 n, const Res& r)
{
    return r.w > 0.f ? (r.lightId == 2 ? hrt_fmax(kEPS_MIN, 1.f / 9.f) : hrt_fmax(kEPS_MIN, cos_hemi_pdf(n, r.wi) * (8.f / 9.f))) : 0.f;
}

HRT_D void reservoir_update(Res& r, F3 wi, float score, int lightId, Rng& rng)   // :394-405 (multiplicity 1)
{
    float newSum = r.wSum + score;
    float acceptP = (newSum > 0.f) ? score / newSum : 0.f;
    if (rng.next_f() < acceptP) { r.wi = wi; r.w = score; r.lightId = lightId; }
    r.wSum = newSum;
    r.m = r.m + 1;
}

HRT_D int reproject_prev(const FrameK& k, F3 posWS)   // :339-360
{
    F3 p = posWS - cv3(k.prevCam.origin);
    float x = dot(p, cv3(k.prevCam.right));
    float y = dot(p, cv3(k.prevCam.up));
    float z = dot(p, cv3(k.prevCam.forward));
    if (z <= 1e-4f) return -1;
    float tanHalfFov = hrt_tan(0.5f * k.prevCam.fovYRadians);
    float ndcX = x / (z * tanHalfFov * k.prevCam.aspect);
    float ndcY = y / (z * tanHalfFov);
    float fx = 0.5f * (ndcX + 1.f) * (float)k.width;
    float fy = 0.5f * (ndcY + 1.f) * (float)k.height;
    int px = hrt_f2i(fx), py = hrt_f2i(fy);
    if ((uint32_t)px >= (uint32_t)k.width || (uint32_t)py >= (uint32_t)k.height) return -1;
    return py * k.width + px;
}

template <bool COUNT>
HRT_D void import_prev(const FrameK& k, const DGBuffer& gb, const DReservoir& prev, int64_t nPix, int prevIdx, int curIdx,
                       F3 n, F3 albedo, float mixLocal, float mixDelta, Rng& rng, Res& r, Cnt<COUNT>& C)   // :408-435
{
    if (prevIdx < 0 || nPix <= (int64_t)prevIdx) return;
    C.inc(C_REUSE_IMPORTS);
    // SpatialCompatible :363-374 (current-frame G-buffer at both indices)
    int objA = gb.objId[curIdx], objB = gb.objId[prevIdx];
    if (objA != objB)
    {
        F3 nB = normalize(ld3(&gb.normalWS[prevIdx]));
        if (dot(n, nB) < 0.85f) return;
        float zA = cam_distance(k, ld3(&gb.worldPos[curIdx]));
        float zB = cam_distance(k, ld3(&gb.worldPos[prevIdx]));
        float rel = hrt_abs(zA - zB) / hrt_fmax(1e-3f, zA);
        if (!(rel < 0.05f)) return;
    }
    int pm = prev.m[prevIdx];
    float pw = prev.w[prevIdx], pwSum = prev.wSum[prevIdx];
    if (!(pm > 0 && pw > 0.f && pwSum > 0.f)) return;
    F3 wi = ld3(&prev.wi[prevIdx]);
    int lid = prev.lightId[prevIdx] == 2 ? 2 : 1;
    F3 LiImp = (lid == 2) ? cv3(k.dirLightRadiance) : sky(k, wi);
    float nl = hrt_fmax(0.f, dot(n, wi));
    float pdfHere = (lid == 2) ? hrt_fmax(kEPS_MIN, mixDelta) : hrt_fmax(kEPS_MIN, cos_hemi_pdf(n, wi) * mixLocal);
    F3 f_over_p = albedo * LiImp * ((nl / pdfHere) * kINV_PI);
    float sHere = luminance(f_over_p);
    float Wsrc = pwSum / ((float)hrt_imax(1, pm) * hrt_fmax(kEPS_MIN, pw));
    reservoir_update(r, wi, sHere * Wsrc, lid, rng);
}

// candidate generation + reuse of ReSTIR_Direct (:449-516); the final visibility ray is
// traced by the caller at the wave-common shadow site.
template <bool COUNT>
HRT_D Res restir_candidates(const FrameK& k, const DGBuffer& gb, const DReservoir& prev, int64_t nPix, int index, bool allowReuse,
                            F3 pos, const Frame& fr, F3 albedo, Rng& rng, Cnt<COUNT>& C)
{
    C.inc(C_DIFFUSE_VERTS);
    const float mixLocal = 8.f / 9.f;     // (float)8/(float)9 , :446
    const float mixDelta = 1.f / 9.f;
    F3 n = fr.n;
    Res r; r.wi = mk3(0.f, 0.f, 0.f); r.w = 0.f; r.wSum = 0.f; r.m = 0; r.lightId = 0;
    // kept rolled: unrolled 8x the candidate body alone is ~3000 instructions (24 KB), a third of the instruction cache
    // two CUs share; the loop-carried state is a handful of registers
#pragma unroll 1
    for (int i = 0; i < 8; i++)
    {
        F3 wi = sample_hemisphere_cosine(fr, rng);
        float nl = hrt_fmax(0.f, dot(n, wi));
        float pdfLocal = hrt_fmax(kEPS_MIN, cos_hemi_pdf(n, wi));
        float pdfSel = hrt_fmax(kEPS_MIN, pdfLocal * mixLocal);
        F3 LiLoc = sky(k, wi);
        F3 f_over_p = albedo * LiLoc * ((nl / pdfSel) * kINV_PI);
        reservoir_update(r, wi, luminance(f_over_p), 1, rng);
    }
    {
        F3 wi = cv3(k.dirLightN);                  // Float3.Normalize(k.dirLightDir), :464
        float nl = hrt_fmax(0.f, dot(n, wi));
        float pdfSel = hrt_fmax(kEPS_MIN, mixDelta);
        F3 LiDir = cv3(k.dirLightRadiance);
        F3 f_over_p = albedo * LiDir * ((nl / pdfSel) * kINV_PI);
        reservoir_update(r, wi, luminance(f_over_p), 2, rng);
    }
    if (allowReuse && k.enableTemporal != 0)
    {
        int prevIdx = reproject_prev(k, pos);
        if (prevIdx >= 0) import_prev(k, gb, prev, nPix, prevIdx, index, n, albedo, mixLocal, mixDelta, rng, r, C);
    }
    if (allowReuse && k.enableSpatial != 0)
    {
        uint32_t h = hash32((uint32_t)index ^ hash32((uint32_t)k.frame ^ hash32(0xB31F5AB1u)));   // Hash3, :643
        int rot = (int)(h & 3u);
        int rad = 1 + (int)((h >> 2) & 1u);
        int x0 = index % k.width, y0 = index / k.width;
        // Neighbor8 (:377-391): offsets (-r,0)(r,0)(0,-r)(0,r)(-r,-r)(r,-r)(-r,r)(r,r) rotated by rot*90deg.
        // (bx+1) and (by+1) of neighbour j sit in nibble j of two constants, so the loop stays rolled.
        const uint32_t kBX = 0x20201120u;     // bx+1 = 0,2,1,1,0,2,0,2
        const uint32_t kBY = 0x22002011u;     // by+1 = 1,1,0,2,0,0,2,2
#pragma unroll 1
        for (int j = 0; j < 8; j++)
        {
            int ox = ((int)((kBX >> (4 * j)) & 15u) - 1) * rad, oy = ((int)((kBY >> (4 * j)) & 15u) - 1) * rad;
            int dx = rot == 0 ? ox : (rot == 1 ? -oy : (rot == 2 ? -ox : oy));
            int dy = rot == 0 ? oy : (rot == 1 ? ox : (rot == 2 ? -oy : -ox));
            int nb = ((uint32_t)(x0 + dx) < (uint32_t)k.width && (uint32_t)(y0 + dy) < (uint32_t)k.height) ? (y0 + dy) * k.width + (x0 + dx) : -1;
            import_prev(k, gb, prev, nPix, nb, index, n, albedo, mixLocal, mixDelta, rng, r, C);
        }
    }
    return r;
}

// ------------------------------------------------------------------ kernel bodies
// PrimaryVisibilityKernel (RTRay.cs:188-201), one pixel per lane.
template <class TR, bool COUNT>
HRT_D void primary_pixel(const TR& tr, const FrameK& k, const DGBuffer& gb, int index, Cnt<COUNT>& C)
{
    int x = index % k.width, y = index / k.width;
    Ray wray = primary_ray(k, x, y);
    Hit h;
    bool hit = tr.template closest<COUNT>(wray, h, C);
    if (!hit)
    {   // StoreMiss :100-108
        gb.hitMask[index] = 0;
        gb.worldPos[index] = to3(wray.o + wray.d * 1e6f);
        gb.normalWS[index] = to3(mk3(0.f, 1.f, 0.f));
        gb.baseColor[index] = to3(mk3(0.f, 0.f, 0.f));
        gb.matId[index] = -1;
        gb.objId[index] = -1;
        return;
    }
    F3 posWS = wray.o + wray.d * h.t;
    int packedMat = (h.shade & 0xFFFF) | (float_to_i16(h.ior) << 16);
    gb.hitMask[index] = 1;
    gb.worldPos[index] = to3(posWS);
    gb.normalWS[index] = to3(h.n);
    gb.baseColor[index] = to3(h.albedo);
    gb.matId[index] = packedMat;
    gb.objId[index] = h.objId;
}

// PathTraceKernel (RTRay.cs:203-325), one pixel per lane, restructured so that each bounce
// has one shadow-ray site and one closest-hit site shared by all material branches.
// SPLIT (small tiles, see SplitK): the lane runs only samples [sk.sBegin, sk.sEnd) of its pixel and hands per-sample radiance
// and its last reservoir to split_resolve_pixel instead of accumulating and storing itself.
struct SplitK {
    hrt_float3* li;          // [spp][nLocal]: SafeColor(Li) of every sample
    float* stage;            // [nGroups][12][nLocal]: L.xyz wi.xyz pdf w wSum m lightId flag of the group's last reservoir
    int sBegin, sEnd, group, nGroups;
    int local, nLocal;       // this lane's slot in the scratch planes and their length: lanes of the launch's tiles, not image pixels
};
// REUSE = false: a frame with both ReSTIR reuse switches off (the launch knows); the import code and the arguments only it reads
// (previous reservoirs, previous camera) are compiled out instead of being carried -- and spilled -- through the bounce loop.
template <class TR, bool COUNT, bool SPLIT = false, bool REUSE = true>
HRT_D void path_trace_pixel(const TR& tr, const FrameK& k, const DGBuffer& gb, const DFramebuffer& fb,
                            const DReservoir& resPrev, const DReservoir& resCur, int64_t nPix, int index, Cnt<COUNT>& C, const SplitK* sk = nullptr)
{
    if (index == 0 && fb.cameraId && (!SPLIT || sk->group == 0)) fb.cameraId[0] = k.debugCamSeq;

    int px = index % hrt_imax(1, k.width), py = index / hrt_imax(1, k.width);
    const int sppAll = hrt_imax(1, k.spp);
    const int sFirst = SPLIT ? sk->sBegin : 0;
    const int spp = SPLIT ? sk->sEnd : sppAll;           // end of this lane's sample range
    F3 Lframe = mk3(0.f, 0.f, 0.f);
    auto add_sample = [&](int sIdx, F3 c) {              // Lframe += SafeColor(Li), RTRay.cs:320 -- or hand the term to the resolve
        if (SPLIT) sk->li[(size_t)sIdx * (size_t)sk->nLocal + (size_t)sk->local] = to3(c);
        else Lframe = Lframe + c;
    };

    const int hitMask = gb.hitMask[index];

    if (hitMask == 0)
    {
        F3 c = safe_color(sky(k, primary_ray(k, px, py).d));
        for (int s = sFirst; s < spp; s++) add_sample(s, c);     // :214-219, same value every sample
        if (SPLIT) sk->stage[((size_t)sk->group * 12 + 11) * (size_t)sk->nLocal + (size_t)sk->local] = 0.f;      // no reservoir from this group
    }
    else
    {
        // Everything a sample start needs from the pixel (seed halves, G-buffer addresses) is derived again from the pixel index
        // behind an optimisation barrier, so that one register lives across the bounce loop instead of the ~17 the compiler would
        // otherwise keep -- and, at 96 registers, spill to scratch memory (68 B per lane of HBM traffic).
        auto fresh_index = [&]() { int i = index; asm volatile("" : "+v"(i)); return i; };
        // The G-buffer vertex every sample starts from (:221-230) is re-read from memory at each sample start (an L2
        // hit) instead of being held in 14 registers across the whole bounce loop.
        F3 pos, nrm, alb, I; int shade; float ior;
        Rng rng;
        auto start_sample = [&](int sIdx) {
            const int i = fresh_index();
            const int w = hrt_imax(1, k.width);
            const SeedBase sb = seed_base((uint32_t)(i % w), (uint32_t)(i / w), k.frame, 0xC0FFEEu, k.rngLockNoise);
            rng = rng_for_sample(sb, (uint32_t)sIdx);
            pos = ld3(&gb.worldPos[i]);
            nrm = normalize(ld3(&gb.normalWS[i]));
            alb = ld3(&gb.baseColor[i]);
            const int packedMat = gb.matId[i];
            shade = packedMat & 0xFFFF;
            ior = (float)((packedMat >> 16) & 0xFFFF) / 1000.f;
            I = normalize(pos - cv3(k.cam.origin));
        };
        // resCur.Write (:42-47): every sample's first diffuse vertex writes the same slot and only the last write survives.
        // The latest one waits in the lane's own LDS column (8 dwords, [field][thread]: conflict-free; L is a function of the
        // others, res_L) instead of registers held across the whole bounce loop, and is stored once after the sample loop
        // ((spp-1) x 44 B/pixel of HBM writes saved).
        __shared__ float s_res[8][256];
        bool haveRes = false;

        // Lanes do not wait for each other at sample boundaries: every lane runs its own (sample, depth)
        // cursor through one flat bounce loop, so a lane whose path ended starts its next sample while its
        // neighbours are still bouncing.  Per-pixel order is untouched (samples of a pixel stay sequential).
        int s = sFirst, depth = 0;
        bool wroteReservoir = false;            // per sample (RTRay.cs:231): every sample's first diffuse vertex writes resCur
        start_sample(sFirst);
        F3 Li = mk3(0.f, 0.f, 0.f), T = mk3(1.f, 1.f, 1.f);
        if (k.maxDepth <= 0) { for (; s < spp; s++) add_sample(s, safe_color(Li)); }

        while (s < spp)
        {
            PSTAT(0);
            bool ended = false;
            {
                Ray ray;
                bool terminated = false;
                if (shade == HRT_SHADING_MIRROR)
                {   // :235-244
                    PSTAT(1);
                    F3 dirR = I - nrm * (2.f * dot(I, nrm));
                    ray = ray_with_normal_offset(pos, nrm, dirR);
                    T = T * alb;
                }
                else if (shade == HRT_SHADING_GLASS)
                {   // :246-275
                    PSTAT(2);
                    F3 Nuse = nrm;
                    bool outside = dot(I, nrm) < 0.f;
                    if (!outside) Nuse = Nuse * -1.f;
                    float iorUse = ior > 0.f ? ior : 1.5f;
                    float etaI = outside ? 1.f : iorUse;
                    float etaT = outside ? iorUse : 1.f;
                    F3 dirR = I - Nuse * (2.f * dot(I, Nuse));
                    // Refract :564-572
                    float eta = etaI / etaT;
                    float cosIr = -dot(I, Nuse);
                    float kk = 1.f - eta * eta * (1.f - cosIr * cosIr);
                    bool refrOk = !(kk < 0.f);
                    F3 dirT = mk3(0.f, 0.f, 0.f);
                    if (refrOk) dirT = normalize(I * eta + Nuse * (eta * cosIr - hrt_sqrt(kk)));
                    float cosI = hrt_abs(dot(I, Nuse));
                    // SchlickFresnel :575-583
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
                    PSTAT(3);
                    Frame fr = make_frame(nrm);
                    Res r = restir_candidates<COUNT>(k, gb, resPrev, nPix, index, REUSE && !wroteReservoir, pos, fr, alb, rng, C);
                    // (5) final shading with one visibility ray :518-539
                    F3 contrib = mk3(0.f, 0.f, 0.f);
                    if (r.m > 0 && r.wSum > 0.f && r.w > 0.f)
                    {
                        F3 wiSel = r.wi;
                        int lidSel = r.lightId == 2 ? 2 : 1;
                        float nlSel = hrt_fmax(0.f, dot(nrm, wiSel));
                        // Visible(): nl <= 0 -> false (:620-621); nlSel > 0 implies it
                        if (nlSel > 0.f)
                        {
                            PSTAT(4);
                            Ray sray = ray_with_normal_offset(pos, nrm, wiSel);
                            if (!tr.template occluded<COUNT>(sray, 1e29f, C))
                            {
                                float pdfSel = (lidSel == 2) ? hrt_fmax(kEPS_MIN, 1.f / 9.f) : hrt_fmax(kEPS_MIN, cos_hemi_pdf(nrm, wiSel) * (8.f / 9.f));
                                F3 LiSel = (lidSel == 2) ? cv3(k.dirLightRadiance) : sky(k, wiSel);
                                F3 f_over_p = alb * LiSel * ((nlSel / pdfSel) * kINV_PI);
                                float W = r.wSum / (float)hrt_imax(1, r.m) / hrt_fmax(kEPS_MIN, r.w);
                                contrib = f_over_p * W;
                            }
                        }
                    }
                    Li = Li + T * contrib;       // :286/:291, also when contrib == 0
                    if (!wroteReservoir)
                    {
                        const int t = threadIdx.x;
                        s_res[0][t] = r.wi.x; s_res[1][t] = r.wi.y; s_res[2][t] = r.wi.z; s_res[3][t] = res_pdf(nrm, r);
                        s_res[4][t] = r.w; s_res[5][t] = r.wSum; s_res[6][t] = __int_as_float(r.m); s_res[7][t] = __int_as_float(r.lightId);
                        haveRes = true;
                        wroteReservoir = true;
                    }
                    F3 wi = sample_hemisphere_cosine(fr, rng);
                    ray = ray_with_normal_offset(pos, nrm, wi);
                    T = T * alb;
                    if (depth >= 3)
                    {   // :306-312
                        float maxC = hrt_fmax(T.x, hrt_fmax(T.y, T.z));
                        maxC = hrt_clamp(maxC, 0.05f, 0.98f);
                        if (rng.next_f() > maxC) { T = mk3(0.f, 0.f, 0.f); terminated = true; }
                        else T = T * (1.0f / maxC);
                    }
                }
                if (terminated) ended = true;
                else
                {   // TraceNext :659-671 -- the one closest-hit site of the bounce loop
                    PSTAT(5);
                    Hit h;
                    if (!tr.template closest<COUNT>(ray, h, C)) { Li = Li + T * sky(k, ray.d); ended = true; }
                    else
                    {
                        pos = ray.o + ray.d * h.t;
                        nrm = normalize(h.n);
                        alb = h.albedo; shade = h.shade; ior = h.ior;
                        I = ray.d;
                        depth++;
                        if (depth >= k.maxDepth) ended = true;
                    }
                }
            }
            if (ended)
            {
                PSTAT(6);
                add_sample(s, safe_color(Li));                     // :320
                s++;
                if (s < spp)
                {   // next sample starts again from the G-buffer vertex (:212-231)
                    wroteReservoir = false;
                    start_sample(s);
                    Li = mk3(0.f, 0.f, 0.f); T = mk3(1.f, 1.f, 1.f);
                    depth = 0;
                }
            }
        }
        if (SPLIT)
        {   // this group's last reservoir (or "none"): split_resolve_pixel keeps the one of the last group that has one
            float* st = sk->stage + (size_t)sk->group * 12 * (size_t)sk->nLocal + (size_t)sk->local;
            const size_t P = (size_t)sk->nLocal;
            st[11 * P] = haveRes ? 1.f : 0.f;
            if (haveRes)
            {
                const int t = threadIdx.x;
                const F3 wi = mk3(s_res[0][t], s_res[1][t], s_res[2][t]);
                const F3 L = res_L(k, wi, s_res[4][t], __float_as_int(s_res[7][t]));
                st[0] = L.x; st[P] = L.y; st[2 * P] = L.z; st[3 * P] = wi.x; st[4 * P] = wi.y; st[5 * P] = wi.z;
#pragma unroll
                for (int f = 3; f < 8; f++) st[(f + 3) * P] = s_res[f][t];       // pdf, w, wSum, m, lightId
            }
        }
        else if (haveRes)
        {
            const int t = threadIdx.x;
            const int i = fresh_index();
            const F3 wi = mk3(s_res[0][t], s_res[1][t], s_res[2][t]);
            resCur.L[i] = to3(res_L(k, wi, s_res[4][t], __float_as_int(s_res[7][t]))); resCur.wi[i] = to3(wi);
            resCur.pdf[i] = s_res[3][t]; resCur.w[i] = s_res[4][t]; resCur.wSum[i] = s_res[5][t];
            resCur.lightId[i] = __float_as_int(s_res[7][t]); resCur.m[i] = __float_as_int(s_res[6][t]);
        }
    }
    if (SPLIT) return;

    F3 Lout = Lframe * (1.0f / (float)sppAll);
    int io = index; asm volatile("" : "+v"(io));          // addresses of the stores derived here, not carried through the loop
    if (fb.radiance) fb.radiance[io] = to3(Lout);
    fb.color[io] = pack_rgba8(Lout);
    fb.depth[io] = cam_distance(k, ld3(&gb.worldPos[io]));
    fb.objectId[io] = gb.objId[io];
}

// second half of a SPLIT frame: ordered sample sum (:320-324), reservoir of the last sample that reached a diffuse vertex
HRT_D void split_resolve_pixel(const FrameK& k, const DGBuffer& gb, const DFramebuffer& fb, const DReservoir& resCur, int index,
                               const hrt_float3* li, const float* stage, int nGroups, int local, int nLocal)
{
    const int spp = hrt_imax(1, k.spp);
    F3 Lframe = mk3(0.f, 0.f, 0.f);
    for (int s = 0; s < spp; s++) Lframe = Lframe + ld3(&li[(size_t)s * (size_t)nLocal + (size_t)local]);
    const size_t P = (size_t)nLocal;
    for (int g = nGroups - 1; g >= 0; g--)
    {
        const float* st = stage + (size_t)g * 12 * P + (size_t)local;
        if (st[11 * P] != 0.f)
        {
            resCur.L[index] = to3(mk3(st[0], st[P], st[2 * P])); resCur.wi[index] = to3(mk3(st[3 * P], st[4 * P], st[5 * P]));
            resCur.pdf[index] = st[6 * P]; resCur.w[index] = st[7 * P]; resCur.wSum[index] = st[8 * P];
            resCur.lightId[index] = __float_as_int(st[10 * P]); resCur.m[index] = __float_as_int(st[9 * P]);
            break;
        }
    }
    F3 Lout = Lframe * (1.0f / (float)spp);
    if (fb.radiance) fb.radiance[index] = to3(Lout);
    fb.color[index] = pack_rgba8(Lout);
    fb.depth[index] = cam_distance(k, ld3(&gb.worldPos[index]));
    fb.objectId[index] = gb.objId[index];
}

} // namespace hrt
