/*
 * oracle/orc_kernels.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's device code for the render hot path:
 *   Engine/Float3.cs, Engine/RTUtils.cs, Engine/SceneDeviceViews.cs, Engine/RTRay.cs
 * (paths relative to /root/reference/ILGPU_Raytracing/).  One function per reference
 * function, same names, same statement order; every function cites the lines it follows.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for this
 * path and cannot be built here (C#/.NET 8 + NuGet ILGPU, no
 * toolchain, CUDA-only).  What pins this file: integer-exact RNG known answers
 * (SURVEY.md Appendix C), closed-form geometric KATs and BVH-vs-brute-force checks in
 * tests/.  XMath.* is evaluated through include/hrt_math.h (see its header).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 *
 * C# semantics fixed explicitly here: (int)float truncates (orc_f2i; out-of-range and
 * NaN give INT_MIN as cvttss2si does), uint arithmetic wraps, `+` binds tighter than
 * `^` (RTUtils.cs:131,133), struct assignment copies, compound `a += b` on Float3 is
 * `a = a + b`, float expressions evaluate left to right in binary32 with no contraction.
 */
#ifndef ORC_KERNELS_HPP
#define ORC_KERNELS_HPP

#include <cstdint>
#include <cstring>
#include "../include/hrt_types.h"
#include "../include/hrt_math.h"

namespace orc {

// C# (int)x for float x (total definition shared with the kernels: hrt_math.h)
static inline int f2i(float x) { return hrt_f2i(x); }

// ---------------------------------------------------------------- Float3.cs:6-114
struct Float3 {
    float X, Y, Z;
    Float3() : X(0.f), Y(0.f), Z(0.f) {}                       // `default`
    Float3(float x, float y, float z) : X(x), Y(y), Z(z) {}    // :12-15
    Float3(const hrt_float3& f) : X(f.X), Y(f.Y), Z(f.Z) {}
    operator hrt_float3() const { hrt_float3 r = {X, Y, Z}; return r; }
};
static inline Float3 operator+(Float3 a, Float3 b) { return Float3(a.X + b.X, a.Y + b.Y, a.Z + b.Z); }   // :18-21
static inline Float3 operator-(Float3 a, Float3 b) { return Float3(a.X - b.X, a.Y - b.Y, a.Z - b.Z); }   // :24-27
static inline Float3 operator*(Float3 a, float s)  { return Float3(a.X * s, a.Y * s, a.Z * s); }         // :30-33
static inline Float3 operator*(float s, Float3 a)  { return Float3(a.X * s, a.Y * s, a.Z * s); }         // :36-39
static inline Float3 operator*(Float3 a, Float3 b) { return Float3(a.X * b.X, a.Y * b.Y, a.Z * b.Z); }   // :42-45
static inline Float3 operator/(Float3 a, float s)  { float inv = 1.f / s; return Float3(a.X * inv, a.Y * inv, a.Z * inv); } // :48-52
static inline Float3 operator-(Float3 v)           { return Float3(-v.X, -v.Y, -v.Z); }                  // :61-64
static inline Float3 Min(Float3 a, Float3 b) { return Float3(hrt_fmin(a.X, b.X), hrt_fmin(a.Y, b.Y), hrt_fmin(a.Z, b.Z)); } // :67-70
static inline Float3 Max(Float3 a, Float3 b) { return Float3(hrt_fmax(a.X, b.X), hrt_fmax(a.Y, b.Y), hrt_fmax(a.Z, b.Z)); } // :73-76
static inline Float3 Cross(Float3 a, Float3 b)                                                            // :79-82
{
    return Float3(a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X);
}
static inline float Dot(Float3 a, Float3 b) { return a.X * b.X + a.Y * b.Y + a.Z * b.Z; }                 // :85-88
static inline Float3 Normalize(Float3 v)                                                                  // :91-95
{
    float inv = hrt_rsqrt(hrt_fmax(1e-20f, v.X * v.X + v.Y * v.Y + v.Z * v.Z));
    return Float3(v.X * inv, v.Y * inv, v.Z * inv);
}
static inline float Length(Float3 v) { return hrt_sqrt(v.X * v.X + v.Y * v.Y + v.Z * v.Z); }              // :104-107
static inline Float3 Center(Float3 a, Float3 b) { return Float3(0.5f * (a.X + b.X), 0.5f * (a.Y + b.Y), 0.5f * (a.Z + b.Z)); } // :110-113

// ---------------------------------------------------------------- RTRay.cs:183-186
static const float PI = 3.14159265358979323846f;
static const float INV_PI = 0.31830988618379067154f;
static const float EPS_N = 0.0025f;
static const float EPS_MIN = 1e-6f;

// RTRay.cs:548-549
static inline Float3 InvDir(Float3 d)
{
    return Float3(1.f / (d.X != 0.f ? d.X : 1e-8f), 1.f / (d.Y != 0.f ? d.Y : 1e-8f), 1.f / (d.Z != 0.f ? d.Z : 1e-8f));
}

// ---------------------------------------------------------------- RTUtils.cs:6-18
struct Ray {
    Float3 origin, dir, invDir;
};
static inline Ray GenerateRay(const hrt_camera& cam, float u, float v)                                   // :13-17
{
    Float3 dir = Normalize(Float3(cam.lowerLeft) + Float3(cam.horizontal) * u + Float3(cam.vertical) * v - Float3(cam.origin));
    Ray r;
    r.origin = cam.origin; r.dir = dir;
    r.invDir = Float3(1.f / (dir.X != 0.f ? dir.X : 1e-8f), 1.f / (dir.Y != 0.f ? dir.Y : 1e-8f), 1.f / (dir.Z != 0.f ? dir.Z : 1e-8f));
    return r;
}

// ---------------------------------------------------------------- RTUtils.cs:20-138
struct RNG {
    uint32_t state;

    static RNG Create(uint32_t seed) { RNG r; r.state = (seed == 0u) ? 1u : seed; return r; }            // :25-30

    uint32_t NextUInt()                                                                                  // :33-42
    {
        uint32_t x = state;
        x ^= x << 13;
        x ^= x >> 17;
        x ^= x << 5;
        state = (x != 0u) ? x : 1u;
        return state;
    }
    float NextFloat()                                                                                    // :45-49
    {
        uint32_t u = NextUInt();
        return (float)(u & 0x00FFFFFFu) * (1.0f / 16777216.0f);
    }
    static uint32_t SplitMix32(uint64_t x)                                                               // :54-62
    {
        x += 0x9E3779B97F4A7C15ULL;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
        x ^= (x >> 31);
        return (uint32_t)(x ^ (x >> 32));
    }
    static uint32_t PcgPermute(uint32_t x)                                                               // :65-74
    {
        x ^= x >> 16;
        x *= 0x7FEB352Du;
        x ^= x >> 15;
        x *= 0x846CA68Bu;
        x ^= x >> 16;
        return x;
    }
    static uint32_t Hash32(uint32_t x)                                                                   // :77-84
    {
        x ^= x >> 17; x *= 0xED5AD4BBu;
        x ^= x >> 11; x *= 0xAC4C1B51u;
        x ^= x >> 15; x *= 0x31848BABu;
        x ^= x >> 14;
        return x;
    }
    static uint32_t RotateLeft(uint32_t v, int r) { return (v << (r & 31)) | (v >> ((32 - r) & 31)); }   // :100-103
    static uint32_t MakeSeed32(uint32_t a, uint32_t b, uint32_t c, uint32_t d)                           // :87-97
    {
        uint64_t lane0 = ((uint64_t)a << 32) | b;
        uint64_t lane1 = ((uint64_t)c << 32) | d;
        uint32_t s0 = SplitMix32(lane0 ^ 0xD1B54A32D192ED03ULL);
        uint32_t s1 = SplitMix32(lane1 ^ 0x94D049BB133111EBULL);
        uint32_t s = PcgPermute(s0 ^ (RotateLeft(s1, 13) + 0x9E3779B1u));
        s |= 1u;
        return s;
    }
    static RNG CreateFromPixel(int px_, int py_, int frame, uint32_t sample, uint32_t salt, int lockNoise) // :116-137
    {
        uint32_t px = (uint32_t)px_;
        uint32_t py = (uint32_t)py_;
        uint32_t f = (lockNoise != 0) ? 0u : (uint32_t)frame;
        uint32_t ln = (uint32_t)lockNoise;
        uint32_t lnMix0 = (lockNoise != 0) ? (Hash32(ln) ^ (ln * 0x1B873593u)) : 0u;
        uint32_t lnMix1 = (lockNoise != 0) ? (RotateLeft(ln, 7) * 0x85EBCA6Bu) : 0u;
        uint32_t lane0a = px ^ 0xB5297A4Du;
        uint32_t lane0b = (py * 0x68E31DA4u) ^ (f * 0x9E3779B1u + 0x85EBCA6Bu) ^ lnMix0;
        uint32_t lane1a = (sample ^ 0xC2B2AE35u) + RotateLeft(px, 16);
        uint32_t lane1b = ((salt ^ 0x27D4EB2Fu) + RotateLeft(py, 8)) ^ lnMix1;   // C#: + before ^
        uint32_t seed = MakeSeed32(lane0a, lane0b, lane1a, lane1b);
        return Create(seed);
    }
    static RNG CreateFromIndex1D(int index, int width, int height, int frame, uint32_t sample, uint32_t salt, int lockNoise) // :108-113
    {
        (void)height;
        uint32_t x = (uint32_t)(index % hrt_imax(1, width));
        uint32_t y = (uint32_t)(index / hrt_imax(1, width));
        return CreateFromPixel((int)x, (int)y, frame, sample, salt, lockNoise);
    }
};

// ---------------------------------------------------------------- views
template <class T> struct ArrayView {      // ILGPU ArrayView<T>: pointer + Length
    T* p; int64_t Length;
    T& operator[](int64_t i) const { return p[i]; }
};

struct Counters : hrt_kernel_counters {
    Counters() { std::memset(static_cast<hrt_kernel_counters*>(this), 0, sizeof(hrt_kernel_counters)); }
};

// SceneDeviceViews.cs:11-27
struct SceneDeviceViews {
    ArrayView<const hrt_bvh_node> tlasNodes;
    ArrayView<const int32_t> tlasInstanceIndices;
    ArrayView<const hrt_instance> instances;
    ArrayView<const hrt_bvh_node> blasNodes;
    ArrayView<const int32_t> spherePrimIdx;
    ArrayView<const hrt_sphere> spheres;
    ArrayView<const int32_t> triPrimIdx;
    ArrayView<const hrt_float3> meshPositions;
    ArrayView<const hrt_mesh_tri> meshTris;
    ArrayView<const hrt_float2> meshTexcoords;
    ArrayView<const hrt_mesh_tri_uv> meshTriUVs;
    ArrayView<const int32_t> triMatIndex;
    ArrayView<const hrt_material> materials;
    ArrayView<const hrt_rgba32> texels;
    ArrayView<const hrt_tex_info> texInfos;
    Counters* C;   // oracle-only: work counters (not in the reference)

    // ---- :475-493
    static Float3 TransformPoint(const hrt_affine3x4& m, Float3 p)
    {
        return Float3(m.m00 * p.X + m.m01 * p.Y + m.m02 * p.Z + m.m03, m.m10 * p.X + m.m11 * p.Y + m.m12 * p.Z + m.m13, m.m20 * p.X + m.m21 * p.Y + m.m22 * p.Z + m.m23);
    }
    static Float3 TransformVector(const hrt_affine3x4& m, Float3 v)
    {
        return Float3(m.m00 * v.X + m.m01 * v.Y + m.m02 * v.Z, m.m10 * v.X + m.m11 * v.Y + m.m12 * v.Z, m.m20 * v.X + m.m21 * v.Y + m.m22 * v.Z);
    }
    static Ray TransformRay(const hrt_affine3x4& m, const Ray& w)
    {
        Ray r;
        r.origin = TransformPoint(m, w.origin);
        r.dir = TransformVector(m, w.dir);
        r.invDir = InvDir(r.dir);
        return r;
    }

    // ---- :496-514
    static bool IntersectAABB(const Ray& ray, Float3 bmin, Float3 bmax, float tMin, float tMax)
    {
        float t1 = (bmin.X - ray.origin.X) * ray.invDir.X;
        float t2 = (bmax.X - ray.origin.X) * ray.invDir.X;
        float tmin = hrt_fmin(t1, t2);
        float tmax = hrt_fmax(t1, t2);

        t1 = (bmin.Y - ray.origin.Y) * ray.invDir.Y;
        t2 = (bmax.Y - ray.origin.Y) * ray.invDir.Y;
        tmin = hrt_fmax(tmin, hrt_fmin(t1, t2));
        tmax = hrt_fmin(tmax, hrt_fmax(t1, t2));

        t1 = (bmin.Z - ray.origin.Z) * ray.invDir.Z;
        t2 = (bmax.Z - ray.origin.Z) * ray.invDir.Z;
        tmin = hrt_fmax(tmin, hrt_fmin(t1, t2));
        tmax = hrt_fmin(tmax, hrt_fmax(t1, t2));

        return tmax >= hrt_fmax(tmin, tMin) && tmin <= tMax;
    }

    // ---- :517-537
    static bool IntersectSphere(const Ray& ray, const hrt_sphere& s, float& t, Float3& n)
    {
        Float3 oc = ray.origin - Float3(s.center);
        float a = Dot(ray.dir, ray.dir);
        float b = 2.f * Dot(oc, ray.dir);
        float c = Dot(oc, oc) - s.radius * s.radius;
        float disc = b * b - 4.f * a * c;
        if (disc < 0.f) { t = 0.f; n = Float3(); return false; }
        float sqrtD = hrt_sqrt(disc);
        float t0 = (-b - sqrtD) / (2.f * a);
        float t1 = (-b + sqrtD) / (2.f * a);
        t = t0;
        if (t < 0.001f)
        {
            t = t1;
            if (t < 0.001f) { n = Float3(); return false; }
        }
        Float3 p = ray.origin + ray.dir * t;
        n = Normalize(p - Float3(s.center));
        return true;
    }

    // ---- :540-558
    static bool IntersectTriangleMT_Bary(const Ray& ray, Float3 v0, Float3 v1, Float3 v2, float& t, Float3& n, float& bu, float& bv)
    {
        Float3 e1 = v1 - v0;
        Float3 e2 = v2 - v0;
        Float3 p = Cross(ray.dir, e2);
        float det = Dot(e1, p);
        if (hrt_abs(det) < 1e-8f) { t = 0.f; n = Float3(); bu = 0.f; bv = 0.f; return false; }
        float invDet = 1.f / det;
        Float3 tv = ray.origin - v0;
        bu = Dot(tv, p) * invDet;
        if (bu < 0.f || bu > 1.f) { t = 0.f; n = Float3(); bv = 0.f; return false; }
        Float3 q = Cross(tv, e1);
        bv = Dot(ray.dir, q) * invDet;
        if (bv < 0.f || bu + bv > 1.f) { t = 0.f; n = Float3(); return false; }
        t = Dot(e2, q) * invDet;
        if (t <= 0.f) { n = Float3(); return false; }
        n = Normalize(Cross(e1, e2));
        return true;
    }

    // ---- :330-339
    hrt_rgba32 TexelRaw(const hrt_tex_info& info, int x, int y) const
    {
        int w = info.Width;
        int h = info.Height;
        hrt_rgba32 z = {0, 0, 0, 0};
        if (w <= 0 || h <= 0) return z;
        int sx = hrt_imax(0, hrt_imin(w - 1, x));
        int sy = hrt_imax(0, hrt_imin(h - 1, y));
        int idx = info.Offset + sy * w + sx;
        return texels[idx];
    }
    // ---- :342-348
    static float Luma01(hrt_rgba32 p)
    {
        float r = p.R * (1.f / 255.f);
        float g = p.G * (1.f / 255.f);
        float b = p.B * (1.f / 255.f);
        return 0.2126f * r + 0.7152f * g + 0.0722f * b;
    }
    // ---- :351-355
    Float3 TexelRGB(const hrt_tex_info& info, int x, int y) const
    {
        hrt_rgba32 p = TexelRaw(info, x, y);
        return Float3(p.R * (1.f / 255.f), p.G * (1.f / 255.f), p.B * (1.f / 255.f));
    }
    // ---- :358-385
    Float3 SampleTextureLinear(const hrt_tex_info& info, float u, float v) const
    {
        int w = info.Width, h = info.Height;
        if (w <= 0 || h <= 0) return Float3(1.f, 1.f, 1.f);

        float fu = u - hrt_floor(u);
        float fv = 1.f - (v - hrt_floor(v));

        float x = fu * (float)(w - 1);
        float y = fv * (float)(h - 1);

        int x0 = f2i(hrt_floor(x));
        int y0 = f2i(hrt_floor(y));
        int x1 = hrt_imin(w - 1, x0 + 1);
        int y1 = hrt_imin(h - 1, y0 + 1);

        float tx = x - (float)x0;
        float ty = y - (float)y0;

        Float3 c00 = TexelRGB(info, x0, y0);
        Float3 c10 = TexelRGB(info, x1, y0);
        Float3 c01 = TexelRGB(info, x0, y1);
        Float3 c11 = TexelRGB(info, x1, y1);

        Float3 cx0 = c00 * (1.f - tx) + c10 * tx;
        Float3 cx1 = c01 * (1.f - tx) + c11 * tx;
        return cx0 * (1.f - ty) + cx1 * ty;
    }
    // ---- :388-415
    float SampleMaskLinear(const hrt_tex_info& info, float u, float v) const
    {
        int w = info.Width, h = info.Height;
        if (w <= 0 || h <= 0) return 1.f;

        float fu = u - hrt_floor(u);
        float fv = 1.f - (v - hrt_floor(v));

        float x = fu * (float)(w - 1);
        float y = fv * (float)(h - 1);

        int x0 = f2i(hrt_floor(x));
        int y0 = f2i(hrt_floor(y));
        int x1 = hrt_imin(w - 1, x0 + 1);
        int y1 = hrt_imin(h - 1, y0 + 1);

        float tx = x - (float)x0;
        float ty = y - (float)y0;

        float a00 = Luma01(TexelRaw(info, x0, y0));
        float a10 = Luma01(TexelRaw(info, x1, y0));
        float a01 = Luma01(TexelRaw(info, x0, y1));
        float a11 = Luma01(TexelRaw(info, x1, y1));

        float ax0 = a00 * (1.f - tx) + a10 * tx;
        float ax1 = a01 * (1.f - tx) + a11 * tx;
        return ax0 * (1.f - ty) + ax1 * ty;
    }
    // ---- :418-428
    float SampleMaskPoint(const hrt_tex_info& info, float u, float v) const
    {
        int w = info.Width, h = info.Height;
        if (w <= 0 || h <= 0) return 1.f;

        float fu = u - hrt_floor(u);
        float fv = 1.f - (v - hrt_floor(v));
        int x = f2i(hrt_round(fu * (float)(w - 1)));
        int y = f2i(hrt_round(fv * (float)(h - 1)));
        return Luma01(TexelRaw(info, x, y));
    }
    // ---- :431-472
    Float3 SampleTextureLinearRGB_A(const hrt_tex_info& info, float u, float v, float& a) const
    {
        int w = info.Width, h = info.Height;
        if (w <= 0 || h <= 0) { a = 1.f; return Float3(1.f, 1.f, 1.f); }

        float fu = u - hrt_floor(u);
        float fv = 1.f - (v - hrt_floor(v));

        float x = fu * (float)(w - 1);
        float y = fv * (float)(h - 1);

        int x0 = f2i(hrt_floor(x));
        int y0 = f2i(hrt_floor(y));
        int x1 = hrt_imin(w - 1, x0 + 1);
        int y1 = hrt_imin(h - 1, y0 + 1);

        float tx = x - (float)x0;
        float ty = y - (float)y0;

        hrt_rgba32 p00 = TexelRaw(info, x0, y0);
        hrt_rgba32 p10 = TexelRaw(info, x1, y0);
        hrt_rgba32 p01 = TexelRaw(info, x0, y1);
        hrt_rgba32 p11 = TexelRaw(info, x1, y1);

        Float3 c00(p00.R * (1.f / 255.f), p00.G * (1.f / 255.f), p00.B * (1.f / 255.f));
        Float3 c10(p10.R * (1.f / 255.f), p10.G * (1.f / 255.f), p10.B * (1.f / 255.f));
        Float3 c01(p01.R * (1.f / 255.f), p01.G * (1.f / 255.f), p01.B * (1.f / 255.f));
        Float3 c11(p11.R * (1.f / 255.f), p11.G * (1.f / 255.f), p11.B * (1.f / 255.f));

        float a00 = p00.A * (1.f / 255.f);
        float a10 = p10.A * (1.f / 255.f);
        float a01 = p01.A * (1.f / 255.f);
        float a11 = p11.A * (1.f / 255.f);

        Float3 cx0 = c00 * (1.f - tx) + c10 * tx;
        Float3 cx1 = c01 * (1.f - tx) + c11 * tx;
        float ax0 = a00 * (1.f - tx) + a10 * tx;
        float ax1 = a01 * (1.f - tx) + a11 * tx;

        a = ax0 * (1.f - ty) + ax1 * ty;
        return cx0 * (1.f - ty) + cx1 * ty;
    }

    // ---- :124-170
    bool TraverseBLAS_Sphere(const Ray& rayObj, int blasStart, int blasEnd, float& tClosest, Float3& nObj, Float3& albedo, int& shading, float& ior) const
    {
        tClosest = 1e30f; nObj = Float3(); albedo = Float3(1.f, 1.f, 1.f); shading = 0; ior = 1.f;
        int cur = blasStart;
        while (cur != -1 && cur < blasEnd)
        {
            hrt_bvh_node n = blasNodes[cur];
            C->node_visits++;
            if (IntersectAABB(rayObj, n.boundsMin, n.boundsMax, 0.001f, tClosest))
            {
                if (n.count > 0)
                {
                    int end = n.first + n.count;
                    for (int i = n.first; i < end; i++)
                    {
                        int prim = spherePrimIdx[i];
                        float t; Float3 nn;
                        C->sphere_tests++;
                        if (IntersectSphere(rayObj, spheres[prim], t, nn))
                        {
                            if (t > 0.001f && t < tClosest)
                            {
                                tClosest = t;
                                nObj = nn;
                                hrt_sphere s = spheres[prim];
                                Float3 kd = s.material.Kd;
                                Float3 col = (kd.X == 0.f && kd.Y == 0.f && kd.Z == 0.f) ? Float3(s.albedo) : kd;
                                if (s.material.HasDiffuseMap != 0 && s.material.DiffuseTexIndex >= 0 && s.material.DiffuseTexIndex < texInfos.Length)
                                {
                                    const float PI_ = 3.14159265358979323846f;
                                    float u = 0.5f + hrt_atan2(nn.Z, nn.X) / (2.f * PI_);
                                    float v = hrt_acos(hrt_fmin(1.f, hrt_fmax(-1.f, nn.Y))) / PI_;
                                    float aTmp;
                                    col = SampleTextureLinearRGB_A(texInfos[s.material.DiffuseTexIndex], u, v, aTmp);
                                }
                                albedo = col;
                                shading = s.shading;
                                ior = s.ior > 0.f ? s.ior : 1.f;
                            }
                        }
                    }
                    cur = n.skipIndex;
                }
                else cur = n.left;
            }
            else cur = n.skipIndex;
        }
        return tClosest < 1e29f;
    }

    // ---- :173-237
    bool TraverseBLAS_Tri_Textured(const Ray& rayObj, int blasStart, int blasEnd, float& tClosest, Float3& nObj, Float3& albedo, int& triOut, float& buOut, float& bvOut) const
    {
        tClosest = 1e30f; nObj = Float3(); albedo = Float3(0.85f, 0.85f, 0.85f); triOut = -1; buOut = 0.f; bvOut = 0.f;
        int cur = blasStart;
        while (cur != -1 && cur < blasEnd)
        {
            hrt_bvh_node n = blasNodes[cur];
            C->node_visits++;
            if (IntersectAABB(rayObj, n.boundsMin, n.boundsMax, 0.001f, tClosest))
            {
                if (n.count > 0)
                {
                    int end = n.first + n.count;
                    for (int i = n.first; i < end; i++)
                    {
                        int triIndex = triPrimIdx[i];
                        hrt_mesh_tri tri = meshTris[triIndex];
                        Float3 v0 = meshPositions[tri.i0];
                        Float3 v1 = meshPositions[tri.i1];
                        Float3 v2 = meshPositions[tri.i2];

                        float t; Float3 nn; float bu; float bv;
                        C->tri_tests++;
                        if (IntersectTriangleMT_Bary(rayObj, v0, v1, v2, t, nn, bu, bv))
                        {
                            int midx = triMatIndex[triIndex];
                            hrt_material mat = materials[midx];
                            C->tri_mt_hits++;

                            if (t > 0.001f && t < tClosest)
                            {
                                C->tri_accepted++;
                                hrt_mesh_tri_uv tuv = meshTriUVs[triIndex];
                                hrt_float2 t0 = meshTexcoords[tuv.t0];
                                hrt_float2 t1 = meshTexcoords[tuv.t1];
                                hrt_float2 t2 = meshTexcoords[tuv.t2];
                                float w = 1.f - bu - bv;
                                float uu = t0.X * w + t1.X * bu + t2.X * bv;
                                float vv = t0.Y * w + t1.Y * bu + t2.Y * bv;

                                float alpha = 1.f;
                                Float3 kdCol = mat.Kd;

                                if (mat.HasDiffuseMap != 0 && mat.DiffuseTexIndex >= 0 && mat.DiffuseTexIndex < texInfos.Length)
                                    kdCol = SampleTextureLinear(texInfos[mat.DiffuseTexIndex], uu, vv);

                                if (mat.HasAlphaMap != 0 && mat.AlphaTexIndex >= 0 && mat.AlphaTexIndex < texInfos.Length)
                                    alpha = SampleMaskLinear(texInfos[mat.AlphaTexIndex], uu, vv);

                                if (alpha < mat.AlphaCutoff) { continue; }

                                tClosest = t;
                                nObj = nn;
                                if (mat.TwoSided != 0 && Dot(nObj, rayObj.dir) > 0.f) nObj = nObj * -1.f;
                                albedo = kdCol;
                                triOut = triIndex;
                                buOut = bu;
                                bvOut = bv;
                            }
                        }
                    }
                    cur = n.skipIndex;
                }
                else cur = n.left;
            }
            else cur = n.skipIndex;
        }
        return tClosest < 1e29f;
    }

    // ---- :240-267
    bool AnyHit_Sphere(const Ray& rayObj, int blasStart, int blasEnd, float tMaxObj) const
    {
        int cur = blasStart;
        while (cur != -1 && cur < blasEnd)
        {
            hrt_bvh_node n = blasNodes[cur];
            C->node_visits++;
            if (IntersectAABB(rayObj, n.boundsMin, n.boundsMax, 0.001f, tMaxObj))
            {
                if (n.count > 0)
                {
                    int end = n.first + n.count;
                    for (int i = n.first; i < end; i++)
                    {
                        int prim = spherePrimIdx[i];
                        float t; Float3 _n;
                        C->sphere_tests++;
                        if (IntersectSphere(rayObj, spheres[prim], t, _n))
                        {
                            if (t > 0.001f && t < tMaxObj) return true;
                        }
                    }
                    cur = n.skipIndex;
                }
                else cur = n.left;
            }
            else cur = n.skipIndex;
        }
        return false;
    }

    // ---- :270-327
    bool AnyHit_Tri_Textured(const Ray& rayObj, int blasStart, int blasEnd, float tMaxObj) const
    {
        int cur = blasStart;
        while (cur != -1 && cur < blasEnd)
        {
            hrt_bvh_node n = blasNodes[cur];
            C->node_visits++;
            if (IntersectAABB(rayObj, n.boundsMin, n.boundsMax, 0.001f, tMaxObj))
            {
                if (n.count > 0)
                {
                    int end = n.first + n.count;
                    for (int i = n.first; i < end; i++)
                    {
                        int triIndex = triPrimIdx[i];
                        hrt_mesh_tri tri = meshTris[triIndex];
                        Float3 v0 = meshPositions[tri.i0];
                        Float3 v1 = meshPositions[tri.i1];
                        Float3 v2 = meshPositions[tri.i2];

                        float t; Float3 nn; float bu; float bv;
                        C->tri_tests++;
                        if (IntersectTriangleMT_Bary(rayObj, v0, v1, v2, t, nn, bu, bv))
                        {
                            if (t <= 0.001f || t >= tMaxObj) continue;

                            int midx = triMatIndex[triIndex];
                            hrt_material mat = materials[midx];
                            C->tri_mt_hits++;

                            if (mat.HasAlphaMap != 0 && mat.AlphaTexIndex >= 0 && mat.AlphaTexIndex < texInfos.Length)
                            {
                                C->tri_accepted++;
                                hrt_mesh_tri_uv tuv = meshTriUVs[triIndex];
                                hrt_float2 t0 = meshTexcoords[tuv.t0];
                                hrt_float2 t1 = meshTexcoords[tuv.t1];
                                hrt_float2 t2 = meshTexcoords[tuv.t2];
                                float w = 1.f - bu - bv;
                                float uu = t0.X * w + t1.X * bu + t2.X * bv;
                                float vv = t0.Y * w + t1.Y * bu + t2.Y * bv;

                                float aPoint = SampleMaskPoint(texInfos[mat.AlphaTexIndex], uu, vv);
                                float cutoff = mat.AlphaCutoff;
                                const float Band = 0.10f;
                                if (aPoint < cutoff - Band) { continue; }
                                if (aPoint >= cutoff + Band) { return true; }

                                float aLin = SampleMaskLinear(texInfos[mat.AlphaTexIndex], uu, vv);
                                if (aLin < cutoff) { continue; }
                            }

                            return true;
                        }
                    }
                    cur = n.skipIndex;
                }
                else cur = n.left;
            }
            else cur = n.skipIndex;
        }
        return false;
    }

    // ---- :30-86
    bool TraceClosest(const Ray& wray, float& closestT, Float3& bestNormal, Float3& bestAlbedo, int& bestObjId, int& bestShade, float& bestIor) const
    {
        C->rays_closest++;
        closestT = 1e30f; bestNormal = Float3(); bestAlbedo = Float3(1.f, 1.f, 1.f); bestObjId = -1; bestShade = 0; bestIor = 1.f;
        int cur = 0;
        while (cur != -1)
        {
            hrt_bvh_node n = tlasNodes[cur];
            C->node_visits++;
            if (IntersectAABB(wray, n.boundsMin, n.boundsMax, 0.001f, closestT))
            {
                if (n.count > 0)
                {
                    int end = n.first + n.count;
                    for (int i = n.first; i < end; i++)
                    {
                        int instIndex = tlasInstanceIndices[i];
                        hrt_instance inst = instances[instIndex];
                        C->leaf_instances++;
                        Ray iray = TransformRay(inst.worldToObject, wray);
                        float scale = inst.uniformScale > 0.f ? inst.uniformScale : 1.f;

                        float tObjClosest; Float3 normalObj; Float3 albedo; int triLocal; float bu; float bv; int shade; float ior;
                        bool hit;
                        int blasStart = inst.blasRoot;
                        int blasEnd = blasStart + inst.blasNodeCount;

                        if (inst.type == HRT_BLAS_SPHERESET)
                        {
                            triLocal = -1; bu = 0.f; bv = 0.f; shade = 0; ior = 1.f;
                            hit = TraverseBLAS_Sphere(iray, blasStart, blasEnd, tObjClosest, normalObj, albedo, shade, ior);
                        }
                        else
                        {
                            shade = 0; ior = 1.f;
                            hit = TraverseBLAS_Tri_Textured(iray, blasStart, blasEnd, tObjClosest, normalObj, albedo, triLocal, bu, bv);
                        }

                        if (hit)
                        {
                            float tWorld = tObjClosest / scale;
                            if (tWorld < closestT)
                            {
                                closestT = tWorld;
                                bestNormal = Normalize(TransformVector(inst.objectToWorld, normalObj));
                                bestAlbedo = albedo;
                                bestObjId = triLocal;
                                bestShade = shade;
                                bestIor = ior;
                            }
                        }
                    }
                    cur = n.skipIndex;
                }
                else cur = n.left;
            }
            else cur = n.skipIndex;
        }
        return closestT < 1e29f;
    }

    // ---- :89-121
    bool ShadowOcclusion(const Ray& srayWorld, float tMaxWorld) const
    {
        C->rays_shadow++;
        int cur = 0;
        while (cur != -1)
        {
            hrt_bvh_node n = tlasNodes[cur];
            C->node_visits++;
            if (IntersectAABB(srayWorld, n.boundsMin, n.boundsMax, 0.001f, tMaxWorld))
            {
                if (n.count > 0)
                {
                    int end = n.first + n.count;
                    for (int i = n.first; i < end; i++)
                    {
                        int instIndex = tlasInstanceIndices[i];
                        hrt_instance inst = instances[instIndex];
                        C->leaf_instances++;

                        Ray srayObj = TransformRay(inst.worldToObject, srayWorld);
                        float scale = inst.uniformScale > 0.f ? inst.uniformScale : 1.f;
                        float tMaxObj = tMaxWorld * scale;

                        bool blocked = (inst.type == HRT_BLAS_SPHERESET)
                            ? AnyHit_Sphere(srayObj, inst.blasRoot, inst.blasRoot + inst.blasNodeCount, tMaxObj)
                            : AnyHit_Tri_Textured(srayObj, inst.blasRoot, inst.blasRoot + inst.blasNodeCount, tMaxObj);
                        if (blocked) return true;
                    }
                    cur = n.skipIndex;
                }
                else cur = n.left;
            }
            else cur = n.skipIndex;
        }
        return false;
    }
};

// ---------------------------------------------------------------- RTRay.cs:23-48
struct GpuReservoirSoA {
    ArrayView<hrt_float3> L, wi;
    ArrayView<float> pdf, w, wSum;
    ArrayView<int32_t> m, lightId;

    hrt_reservoir Read(int index) const                                          // :34-40
    {
        hrt_reservoir r;
        r.L = L[index]; r.wi = wi[index]; r.pdf = pdf[index];
        r.w = w[index]; r.wSum = wSum[index]; r.m = m[index]; r.lightId = lightId[index];
        return r;
    }
    void Write(int index, const hrt_reservoir& r) const                          // :42-47
    {
        L[index] = r.L; wi[index] = r.wi; pdf[index] = r.pdf;
        w[index] = r.w; wSum[index] = r.wSum; lightId[index] = r.lightId;
        m[index] = r.m;
    }
};

// ---------------------------------------------------------------- RTRay.cs:51-77
struct GpuFramebuffer {
    ArrayView<int32_t> color;
    ArrayView<float> depth;
    ArrayView<int32_t> objectId;
    ArrayView<int32_t> cameraId;
    ArrayView<hrt_float3> radiance;    // oracle/boundary extra: pre-pack Lout

    static int ToByte(float x)                                                   // :72-76
    {
        float c = hrt_fmin(1.f, hrt_fmax(0.f, x));
        return f2i(255.99f * c);
    }
    static int PackRGBA8(Float3 c)                                               // :66-70
    {
        int R = ToByte(c.X), G = ToByte(c.Y), B = ToByte(c.Z);
        return (int)((255u << 24) | ((uint32_t)R << 16) | ((uint32_t)G << 8) | (uint32_t)B);
    }
    void Store(int index, Float3 rgb, float z, int obj) const                    // :59-64
    {
        color[index] = PackRGBA8(rgb);
        depth[index] = z;
        objectId[index] = obj;
    }
};

// ---------------------------------------------------------------- RTRay.cs:80-109
struct GpuGBuffer {
    ArrayView<hrt_float3> worldPos, normalWS, baseColor;
    ArrayView<int32_t> matId, objId, hitMask;

    void StoreHit(int index, Float3 posWS, Float3 nWS, Float3 albedo, int packedMat, int oid) const   // :90-98
    {
        hitMask[index] = 1;
        worldPos[index] = posWS;
        normalWS[index] = nWS;
        baseColor[index] = albedo;
        matId[index] = packedMat;
        objId[index] = oid;
    }
    void StoreMiss(int index, const Ray& primary) const                                               // :100-108
    {
        hitMask[index] = 0;
        worldPos[index] = primary.origin + primary.dir * 1e6f;
        normalWS[index] = Float3(0.f, 1.f, 0.f);
        baseColor[index] = Float3(0.f, 0.f, 0.f);
        matId[index] = -1;
        objId[index] = -1;
    }
};

// ---------------------------------------------------------------- RTRay.cs:112-127
struct GBufferParams {
    int width, height, frame;
    hrt_camera cam;
    SceneDeviceViews views;
    GpuGBuffer gb;

    Ray PrimaryRay(int index) const                                              // :120-126
    {
        int x = index % width, y = index / width;
        float u = ((float)x + 0.5f) / (float)hrt_imax(1, width);
        float v = ((float)y + 0.5f) / (float)hrt_imax(1, height);
        return GenerateRay(cam, u, v);
    }
};

// ---------------------------------------------------------------- RTRay.cs:129-169
struct IntegratorParams {
    int width, height, frame;
    hrt_camera cam;
    hrt_camera prevCam;
    SceneDeviceViews views;
    GpuGBuffer gb;
    GpuFramebuffer fb;
    Float3 dirLightDir, dirLightRadiance;
    Float3 skyTintTop, skyTintBottom;
    int debugCamSeq;
    GpuReservoirSoA resPrev;
    GpuReservoirSoA resCur;
    int enableTemporalReuse, enableSpatialReuse, rngLockNoise;
    int spp;

    Float3 PrimaryRayDir(int index) const                                        // :148-154
    {
        int x = index % width, y = index / width;
        float u = ((float)x + 0.5f) / (float)hrt_imax(1, width);
        float v = ((float)y + 0.5f) / (float)hrt_imax(1, height);
        return GenerateRay(cam, u, v).dir;
    }
    Float3 ViewDirFromCam(Float3 posWS) const { return Normalize(posWS - Float3(cam.origin)); }   // :156
    float DistanceFromCamera(Float3 posWS) const                                 // :158-162
    {
        Float3 d = posWS - Float3(cam.origin);
        return hrt_sqrt(d.X * d.X + d.Y * d.Y + d.Z * d.Z);
    }
    Float3 SkyWeighted(Float3 dir) const                                         // :164-168
    {
        float tbg = 0.5f * (dir.Y + 1.0f);
        return skyTintBottom * (1.f - tbg) + skyTintTop * tbg;
    }
};

// ---------------------------------------------------------------- RTRay.cs:181-671
struct RTRay {
    // :609-615
    static int FloatToI16(float x)
    {
        float cl = hrt_fmax(0.f, hrt_fmin(65535.f, x * 1000.f));
        return f2i(cl) & 0xFFFF;
    }
    static float I16ToFloat(int v) { return (float)v / 1000.f; }

    // :188-201
    static void PrimaryVisibilityKernel(int index, const GBufferParams& p)
    {
        int64_t length = p.gb.worldPos.Length;
        if (index >= length) return;

        Ray wray = p.PrimaryRay(index);
        float t; Float3 n; Float3 albedo; int objId; int shade; float ior;
        bool hit = p.views.TraceClosest(wray, t, n, albedo, objId, shade, ior);

        if (!hit) { p.gb.StoreMiss(index, wray); return; }
        Float3 posWS = wray.origin + wray.dir * t;
        int packedMat = (shade & 0xFFFF) | (FloatToI16(ior) << 16);
        p.gb.StoreHit(index, posWS, n, albedo, packedMat, objId);
    }

    // :552-558
    static Ray MakeRayWithNormalOffset(Float3 origin, Float3 n, Float3 dir, float epsN)
    {
        Float3 d = Normalize(dir);
        float s = Dot(n, d) >= 0.f ? 1.f : -1.f;
        Float3 o = origin + n * (epsN * s);
        Ray r; r.origin = o; r.dir = d; r.invDir = InvDir(d);
        return r;
    }
    // :561
    static Float3 Reflect(Float3 I, Float3 N) { return I - N * (2.f * Dot(I, N)); }
    // :564-572
    static bool Refract(Float3 I, Float3 N, float etaI, float etaT, Float3& T)
    {
        float eta = etaI / etaT;
        float cosI = -Dot(I, N);
        float k = 1.f - eta * eta * (1.f - cosI * cosI);
        if (k < 0.f) { T = Float3(); return false; }
        T = Normalize(I * eta + N * (eta * cosI - hrt_sqrt(k)));
        return true;
    }
    // :575-583
    static float SchlickFresnel(float cos, float etaI, float etaT)
    {
        float r0 = (etaI - etaT) / (etaI + etaT);
        r0 = r0 * r0;
        float oneMinusCos = 1.f - cos;
        float oneMinusCos2 = oneMinusCos * oneMinusCos;
        float oneMinusCos5 = oneMinusCos2 * oneMinusCos2 * oneMinusCos;
        return r0 + (1.f - r0) * oneMinusCos5;
    }
    // :601-606
    static void OrthonormalBasis(Float3 n, Float3& t, Float3& b)
    {
        Float3 up = hrt_abs(n.Y) < 0.999f ? Float3(0.f, 1.f, 0.f) : Float3(1.f, 0.f, 0.f);
        t = Normalize(Cross(up, n));
        b = Cross(n, t);
    }
    // :586-598
    static Float3 SampleHemisphereCosine(Float3 n, RNG& rng)
    {
        float r1 = rng.NextFloat(), r2 = rng.NextFloat();
        float phi = 2.f * PI * r1;
        float cosTheta = hrt_sqrt(1.f - r2);
        float sinTheta = hrt_sqrt(r2);
        float x = hrt_cos(phi) * sinTheta;
        float y = hrt_sin(phi) * sinTheta;
        float z = cosTheta;
        Float3 t, b;
        OrthonormalBasis(n, t, b);
        Float3 v = t * x + b * y + n * z;
        return Normalize(v);
    }
    // :618-624
    static bool Visible(const IntegratorParams& k, Float3 origin, Float3 n, Float3 wi)
    {
        float nl = Dot(n, wi);
        if (nl <= 0.f) return false;
        Ray s = MakeRayWithNormalOffset(origin, n, wi, EPS_N);
        return !k.views.ShadowOcclusion(s, 1e29f);
    }
    // :627
    static float Luminance(Float3 c) { return 0.2126f * c.X + 0.7152f * c.Y + 0.0722f * c.Z; }
    // :630-634
    static float CosHemispherePdf(Float3 n, Float3 wi)
    {
        float nl = hrt_fmax(0.f, Dot(n, wi));
        return nl * INV_PI;
    }
    // :637-643
    static uint32_t Hash(uint32_t x)
    {
        x ^= x >> 17; x *= 0xed5ad4bbu; x ^= x >> 11; x *= 0xac4c1b51u; x ^= x >> 15; x *= 0x31848babu; x ^= x >> 14;
        return x;
    }
    static uint32_t Hash3(uint32_t a, uint32_t b, uint32_t c) { return Hash(a ^ Hash(b ^ Hash(c))); }
    // :646-655
    static Float3 SafeColor(Float3 c)
    {
        float x = hrt_isfinite(c.X) ? c.X : 0.f;
        float y = hrt_isfinite(c.Y) ? c.Y : 0.f;
        float z = hrt_isfinite(c.Z) ? c.Z : 0.f;
        x = hrt_fmin(1e6f, hrt_fmax(-1e6f, x));
        y = hrt_fmin(1e6f, hrt_fmax(-1e6f, y));
        z = hrt_fmin(1e6f, hrt_fmax(-1e6f, z));
        return Float3(x, y, z);
    }
    // :659-671
    static bool TraceNext(const IntegratorParams& k, const Ray& ray, Float3& pos, Float3& nrm, Float3& alb, int& shade, float& ior)
    {
        float t; Float3 n2; Float3 alb2; int obj2; int shade2; float ior2;
        bool hit2 = k.views.TraceClosest(ray, t, n2, alb2, obj2, shade2, ior2);
        if (!hit2) return false;
        pos = ray.origin + ray.dir * t;
        nrm = Normalize(n2);
        alb = alb2;
        shade = shade2;
        ior = ior2;
        return true;
    }

    // :330-335
    static hrt_reservoir NewReservoirDefault()
    {
        hrt_reservoir r;
        r.L = Float3(); r.wi = Float3(); r.pdf = 0.f; r.w = 0.f; r.wSum = 0.f; r.m = 0; r.lightId = 0;
        return r;
    }
    // :339-360
    static int ReprojectToPrevPixel(Float3 posWS, const IntegratorParams& k)
    {
        Float3 p = posWS - Float3(k.prevCam.origin);
        float x = Dot(p, k.prevCam.right);
        float y = Dot(p, k.prevCam.up);
        float z = Dot(p, k.prevCam.forward);

        if (z <= 1e-4f) return -1;

        float tanHalfFov = hrt_tan(0.5f * k.prevCam.fovYRadians);
        float ndcX = x / (z * tanHalfFov * k.prevCam.aspect);
        float ndcY = y / (z * tanHalfFov);

        float fx = 0.5f * (ndcX + 1.f) * (float)k.width;
        float fy = 0.5f * (ndcY + 1.f) * (float)k.height;
        int px = f2i(fx);
        int py = f2i(fy);
        if ((uint32_t)px >= (uint32_t)k.width || (uint32_t)py >= (uint32_t)k.height) return -1;
        return py * k.width + px;
    }
    // :363-374
    static bool SpatialCompatible(const IntegratorParams& k, int idxA, int idxB, Float3 nA)
    {
        int objA = k.gb.objId[idxA], objB = k.gb.objId[idxB];
        if (objA == objB) return true;
        Float3 nB = Normalize(k.gb.normalWS[idxB]);
        float ndot = Dot(nA, nB);
        if (ndot < 0.85f) return false;
        float zA = k.DistanceFromCamera(k.gb.worldPos[idxA]);
        float zB = k.DistanceFromCamera(k.gb.worldPos[idxB]);
        float rel = hrt_abs(zA - zB) / hrt_fmax(1e-3f, zA);
        return rel < 0.05f;
    }
    // :377-391
    static int RX(int x, int y, int R) { return R == 0 ? x : (R == 1 ? -y : (R == 2 ? -x : y)); }
    static int RY(int x, int y, int R) { return R == 0 ? y : (R == 1 ? x : (R == 2 ? -y : -x)); }
    static void Neighbor8(int rot, int radius, int dx[8], int dy[8])
    {
        int r = radius;
        dx[0] = RX(-r, 0, rot); dy[0] = RY(-r, 0, rot);
        dx[1] = RX(r, 0, rot); dy[1] = RY(r, 0, rot);
        dx[2] = RX(0, -r, rot); dy[2] = RY(0, -r, rot);
        dx[3] = RX(0, r, rot); dy[3] = RY(0, r, rot);
        dx[4] = RX(-r, -r, rot); dy[4] = RY(-r, -r, rot);
        dx[5] = RX(r, -r, rot); dy[5] = RY(r, -r, rot);
        dx[6] = RX(-r, r, rot); dy[6] = RY(-r, r, rot);
        dx[7] = RX(r, r, rot); dy[7] = RY(r, r, rot);
    }
    // :394-405
    static void ReservoirUpdate(hrt_reservoir& r, Float3 wi, float pdfSel, Float3 Li, float scoreS, int multiplicity, int lightId, RNG& rng)
    {
        float add = scoreS;
        float newSum = r.wSum + add;
        float acceptP = (newSum > 0.f) ? add / newSum : 0.f;
        if (rng.NextFloat() < acceptP)
        {
            r.wi = wi; r.pdf = pdfSel; r.L = Li; r.w = scoreS; r.lightId = lightId;
        }
        r.wSum = newSum;
        r.m = r.m + hrt_imax(1, multiplicity);
    }
    // :408-435
    static void ImportFromPrevReservoir(int prevIdx, int curIdx, const IntegratorParams& k, Float3 n, Float3 albedo, float mixLocal, float mixDelta, RNG& rng, hrt_reservoir& r)
    {
        if (prevIdx < 0 || k.resPrev.L.Length <= prevIdx) return;
        k.views.C->reuse_imports++;
        if (!SpatialCompatible(k, curIdx, prevIdx, n)) return;

        hrt_reservoir pr = k.resPrev.Read(prevIdx);
        if (!(pr.m > 0 && pr.w > 0.f && pr.wSum > 0.f)) return;

        Float3 wi = pr.wi;
        int lid = pr.lightId == 2 ? 2 : 1;
        Float3 LiImp = (lid == 2) ? k.dirLightRadiance : k.SkyWeighted(wi);

        float nl = hrt_fmax(0.f, Dot(n, wi));
        float pdfHere = (lid == 2)
            ? hrt_fmax(EPS_MIN, mixDelta)
            : hrt_fmax(EPS_MIN, CosHemispherePdf(n, wi) * mixLocal);

        Float3 f_over_p = albedo * LiImp * ((nl / pdfHere) * INV_PI);
        float sHere = Luminance(f_over_p);

        float Wsrc = pr.wSum / ((float)hrt_imax(1, pr.m) * hrt_fmax(EPS_MIN, pr.w));
        float eff = sHere * Wsrc;

        ReservoirUpdate(r, wi, pdfHere, LiImp, eff, 1, lid, rng);
    }
    // :438-543
    static Float3 ReSTIR_Direct(int index, const IntegratorParams& k, Float3 pos, Float3 n, Float3 albedo, RNG& rng, hrt_reservoir& outRes)
    {
        k.views.C->diffuse_vertices++;
        const int LocalCandidates = 8;
        const int DeltaCandidates = 1;
        const int TotalNew = LocalCandidates + DeltaCandidates;
        float mixLocal = (float)LocalCandidates / (float)TotalNew;
        float mixDelta = (float)DeltaCandidates / (float)TotalNew;

        hrt_reservoir r = NewReservoirDefault();

        for (int i = 0; i < LocalCandidates; i++)
        {
            Float3 wi = SampleHemisphereCosine(n, rng);
            float nl = hrt_fmax(0.f, Dot(n, wi));
            float pdfLocal = hrt_fmax(EPS_MIN, CosHemispherePdf(n, wi));
            float pdfSel = hrt_fmax(EPS_MIN, pdfLocal * mixLocal);
            Float3 LiLoc = k.SkyWeighted(wi);
            Float3 f_over_p = albedo * LiLoc * ((nl / pdfSel) * INV_PI);
            float s = Luminance(f_over_p);
            ReservoirUpdate(r, wi, pdfSel, LiLoc, s, 1, 1, rng);
        }

        {
            Float3 wi = Normalize(k.dirLightDir);
            float nl = hrt_fmax(0.f, Dot(n, wi));
            float pdfSel = hrt_fmax(EPS_MIN, mixDelta);
            Float3 LiDir = k.dirLightRadiance;
            Float3 f_over_p = albedo * LiDir * ((nl / pdfSel) * INV_PI);
            float s = Luminance(f_over_p);
            ReservoirUpdate(r, wi, pdfSel, LiDir, s, 1, 2, rng);
        }

        if (k.enableTemporalReuse != 0)
        {
            int prevIdx = ReprojectToPrevPixel(pos, k);
            if (prevIdx >= 0)
            {
                ImportFromPrevReservoir(prevIdx, index, k, n, albedo, mixLocal, mixDelta, rng, r);
            }
        }

        if (k.enableSpatialReuse != 0)
        {
            uint32_t h = Hash3((uint32_t)index, (uint32_t)k.frame, 0xB31F5AB1u);
            int rot = (int)(h & 3u);
            int radius = 1 + (int)((h >> 2) & 1u);
            int x0 = index % k.width, y0 = index / k.width;

            int dx[8], dy[8];
            Neighbor8(rot, radius, dx, dy);

            int nb[8];
            for (int j = 0; j < 8; j++)
                nb[j] = ((uint32_t)(x0 + dx[j]) < (uint32_t)k.width && (uint32_t)(y0 + dy[j]) < (uint32_t)k.height) ? (y0 + dy[j]) * k.width + (x0 + dx[j]) : -1;

            for (int j = 0; j < 8; j++)
                ImportFromPrevReservoir(nb[j], index, k, n, albedo, mixLocal, mixDelta, rng, r);
        }

        Float3 contrib(0.f, 0.f, 0.f);
        if (r.m > 0 && r.wSum > 0.f && r.w > 0.f)
        {
            Float3 wiSel = r.wi;
            int lidSel = r.lightId == 2 ? 2 : 1;

            float nlSel = hrt_fmax(0.f, Dot(n, wiSel));
            if (nlSel > 0.f && Visible(k, pos, n, wiSel))
            {
                float mixLocal2 = (float)LocalCandidates / (float)(LocalCandidates + DeltaCandidates);
                float mixDelta2 = (float)DeltaCandidates / (float)(LocalCandidates + DeltaCandidates);
                float pdfSel = (lidSel == 2)
                    ? hrt_fmax(EPS_MIN, mixDelta2)
                    : hrt_fmax(EPS_MIN, CosHemispherePdf(n, wiSel) * mixLocal2);

                Float3 LiSel = (lidSel == 2) ? k.dirLightRadiance : k.SkyWeighted(wiSel);
                Float3 f_over_p = albedo * LiSel * ((nlSel / pdfSel) * INV_PI);
                float W = r.wSum / (float)hrt_imax(1, r.m) / hrt_fmax(EPS_MIN, r.w);
                contrib = f_over_p * W;
            }
        }

        outRes = r;
        return contrib;
    }

    // :203-325
    static void PathTraceKernel(int index, const IntegratorParams& k, int MaxDepth)
    {
        if (index >= k.fb.color.Length) return;
        if (index == 0 && k.fb.cameraId.Length > 0) k.fb.cameraId[0] = k.debugCamSeq;

        Float3 Lframe;

        for (int s = 0; s < hrt_imax(1, k.spp); s++)
        {
            RNG rng = RNG::CreateFromIndex1D(index, k.width, k.height, k.frame, (uint32_t)s, 0xC0FFEEu, k.rngLockNoise);

            if (k.gb.hitMask[index] == 0)
            {
                Float3 vdir = k.PrimaryRayDir(index);
                Lframe = Lframe + SafeColor(k.SkyWeighted(vdir));
                continue;
            }

            Float3 pos = k.gb.worldPos[index];
            Float3 nrm = Normalize(k.gb.normalWS[index]);
            Float3 alb = k.gb.baseColor[index];
            int packedMat = k.gb.matId[index];
            int shade = packedMat & 0xFFFF;
            float ior = I16ToFloat((packedMat >> 16) & 0xFFFF);

            Float3 Li(0.f, 0.f, 0.f);
            Float3 throughput(1.f, 1.f, 1.f);
            Float3 I = k.ViewDirFromCam(pos);
            bool wroteReservoir = false;

            for (int depth = 0; depth < MaxDepth; depth++)
            {
                if (shade == HRT_SHADING_MIRROR)
                {
                    Float3 dirR = Reflect(I, nrm);
                    Ray ray = MakeRayWithNormalOffset(pos, nrm, dirR, EPS_N);
                    throughput = throughput * alb;

                    if (!TraceNext(k, ray, pos, nrm, alb, shade, ior))
                    { Li = Li + throughput * k.SkyWeighted(ray.dir); break; }
                    I = ray.dir; continue;
                }

                if (shade == HRT_SHADING_GLASS)
                {
                    Float3 Nuse = nrm;
                    bool outside = Dot(I, nrm) < 0.f;
                    if (!outside) Nuse = Nuse * -1.f;
                    float etaI = outside ? 1.f : (ior > 0.f ? ior : 1.5f);
                    float etaT = outside ? (ior > 0.f ? ior : 1.5f) : 1.f;

                    Float3 dirR = Reflect(I, Nuse);
                    Float3 dirT;
                    bool refrOk = Refract(I, Nuse, etaI, etaT, dirT);
                    float cosI = hrt_abs(Dot(I, Nuse));
                    float Fr = SchlickFresnel(cosI, etaI, etaT);
                    float xi = rng.NextFloat();

                    Ray ray = (!refrOk || xi < Fr)
                        ? MakeRayWithNormalOffset(pos, Nuse, dirR, EPS_N)
                        : MakeRayWithNormalOffset(pos, -Nuse, dirT, EPS_N);

                    if (refrOk && xi >= Fr)
                    {
                        Float3 transTint = (alb.X == 0.f && alb.Y == 0.f && alb.Z == 0.f) ? Float3(1.f, 1.f, 1.f) : alb;
                        float etaScale = (etaI * etaI) / (etaT * etaT);
                        throughput = throughput * transTint * etaScale;
                    }

                    if (!TraceNext(k, ray, pos, nrm, alb, shade, ior))
                    { Li = Li + throughput * k.SkyWeighted(ray.dir); break; }
                    I = ray.dir; continue;
                }

                {
                    hrt_reservoir outRes;
                    if (wroteReservoir)
                    {
                        IntegratorParams kLocal = k;
                        kLocal.enableTemporalReuse = 0;
                        kLocal.enableSpatialReuse = 0;
                        Float3 direct = ReSTIR_Direct(index, kLocal, pos, nrm, alb, rng, outRes);
                        Li = Li + throughput * direct;
                    }
                    else
                    {
                        Float3 direct = ReSTIR_Direct(index, k, pos, nrm, alb, rng, outRes);
                        Li = Li + throughput * direct;
                        if (k.resCur.L.Length > index)
                        {
                            k.resCur.Write(index, outRes);
                            wroteReservoir = true;
                        }
                    }
                }

                {
                    Float3 wi = SampleHemisphereCosine(nrm, rng);
                    Ray ray = MakeRayWithNormalOffset(pos, nrm, wi, EPS_N);
                    throughput = throughput * alb;

                    if (depth >= 3)
                    {
                        float maxC = hrt_fmax(throughput.X, hrt_fmax(throughput.Y, throughput.Z));
                        maxC = hrt_clamp(maxC, 0.05f, 0.98f);
                        if (rng.NextFloat() > maxC) { throughput = Float3(0.f, 0.f, 0.f); break; }
                        throughput = throughput * (1.0f / maxC);
                    }

                    if (!TraceNext(k, ray, pos, nrm, alb, shade, ior))
                    { Li = Li + throughput * k.SkyWeighted(ray.dir); break; }
                    I = ray.dir; continue;
                }
            }

            Lframe = Lframe + SafeColor(Li);
        }

        Float3 Lout = Lframe * (1.0f / (float)hrt_imax(1, k.spp));
        if (k.fb.radiance.p) k.fb.radiance[index] = Lout;
        k.fb.Store(index, Lout, k.DistanceFromCamera(k.gb.worldPos[index]), k.gb.objId[index]);
    }
};

} // namespace orc
#endif
