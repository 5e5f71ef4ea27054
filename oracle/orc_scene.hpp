/*
 * oracle/orc_scene.hpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Literal CPU restatement of the reference's HOST code that feeds the hot path:
 *   Engine/Scene.cs:83-142 (BuildDefaultScene, minus the Sponza probe), :144-256
 *   (LoadObjInstance, array-append half: the OBJ file parser is out of scope),
 *   :315-652 (AddSphere, BuildSphereInstance, BLAS/TLAS builders, comparators,
 *   TransformAABB, InvertRigidOrUniform, ComputeMeshBounds),
 *   Engine/Camera.cs:19-47,100-126,184-230, Engine/RTRenderer.cs:174-178,241-263,
 *   Engine/Framebuffer.cs:127-146.
 * List<T> -> std::vector<T>, recursion kept as recursion, same append order.
 *
 * Third-party algorithm restated here: System.Array.Sort(T[], int, int, IComparer<T>)
 * of .NET 8 (target framework of ILGPU_Raytracing.csproj:5), which is the unstable
 * introspective sort of System.Collections.Generic.ArraySortHelper<T>: insertion sort
 * for partitions <= 16 (2- and 3-element special cases), median-of-three pivot with the
 * pivot parked at hi-1, heapsort once the depth limit 2*(floor(log2 n)+1) is used up.
 * The BVH topology under centroid ties depends on it.  PARITY UNPINNED (no reference
 * fixture exercises it); tests check the output is a correct ordering and that closest
 * hits equal brute force.
 */
#ifndef ORC_SCENE_HPP
#define ORC_SCENE_HPP

#include <vector>
#include <cfloat>
#include "orc_kernels.hpp"

namespace orc {

// ------------------------------------------------------------------ host flavour of Float3.Min / Max / Normalize
// Everything in this file is code the reference runs on the HOST, where XMath.Min / Max are Math.Min / Max of .NET (a NaN
// operand is returned; include/hrt_math.h, hrt_host_fmin).  The kernels' Min / Max / Normalize of orc_kernels.hpp (minNum)
// are not used here.
static inline Float3 HMin(Float3 a, Float3 b) { return Float3(hrt_host_fmin(a.X, b.X), hrt_host_fmin(a.Y, b.Y), hrt_host_fmin(a.Z, b.Z)); } // Float3.cs:67-70
static inline Float3 HMax(Float3 a, Float3 b) { return Float3(hrt_host_fmax(a.X, b.X), hrt_host_fmax(a.Y, b.Y), hrt_host_fmax(a.Z, b.Z)); } // Float3.cs:73-76
static inline Float3 HNormalize(Float3 v)                                                                                                     // Float3.cs:91-95
{
    float inv = hrt_rsqrt(hrt_host_fmax(1e-20f, v.X * v.X + v.Y * v.Y + v.Z * v.Z));
    return Float3(v.X * inv, v.Y * inv, v.Z * inv);
}

// ------------------------------------------------------------------ .NET 8 ArraySortHelper<T>
template <class T, class Cmp>
struct DotnetSort {
    static void Swap(T* k, int i, int j) { T t = k[i]; k[i] = k[j]; k[j] = t; }
    static void SwapIfGreater(T* k, Cmp& c, int i, int j) { if (c(k[i], k[j]) > 0) Swap(k, i, j); }
    static int Log2(uint32_t v) { int r = 0; while (v >>= 1) r++; return r; }

    static void Sort(T* keys, int length, Cmp& c)
    {
        if (length < 2) return;
        IntroSort(keys, length, 2 * (Log2((uint32_t)length) + 1), c);
    }
    static void IntroSort(T* keys, int partitionSize, int depthLimit, Cmp& c)
    {
        while (partitionSize > 1)
        {
            if (partitionSize <= 16)
            {
                if (partitionSize == 2) { SwapIfGreater(keys, c, 0, 1); return; }
                if (partitionSize == 3)
                {
                    SwapIfGreater(keys, c, 0, 1);
                    SwapIfGreater(keys, c, 0, 2);
                    SwapIfGreater(keys, c, 1, 2);
                    return;
                }
                InsertionSort(keys, partitionSize, c);
                return;
            }
            if (depthLimit == 0) { HeapSort(keys, partitionSize, c); return; }
            depthLimit--;
            int p = PickPivotAndPartition(keys, partitionSize, c);
            IntroSort(keys + (p + 1), partitionSize - (p + 1), depthLimit, c);
            partitionSize = p;
        }
    }
    static int PickPivotAndPartition(T* keys, int length, Cmp& c)
    {
        int hi = length - 1;
        int middle = hi >> 1;
        SwapIfGreater(keys, c, 0, middle);
        SwapIfGreater(keys, c, 0, hi);
        SwapIfGreater(keys, c, middle, hi);
        T pivot = keys[middle];
        Swap(keys, middle, hi - 1);
        int left = 0, right = hi - 1;
        while (left < right)
        {
            while (c(keys[++left], pivot) < 0) ;
            while (c(pivot, keys[--right]) < 0) ;
            if (left >= right) break;
            Swap(keys, left, right);
        }
        if (left != hi - 1) Swap(keys, left, hi - 1);
        return left;
    }
    static void InsertionSort(T* keys, int length, Cmp& c)
    {
        for (int i = 0; i < length - 1; i++)
        {
            T t = keys[i + 1];
            int j = i;
            while (j >= 0 && c(t, keys[j]) < 0) { keys[j + 1] = keys[j]; j--; }
            keys[j + 1] = t;
        }
    }
    static void DownHeap(T* keys, int i, int n, Cmp& c)
    {
        T d = keys[i - 1];
        while (i <= n >> 1)
        {
            int child = 2 * i;
            if (child < n && c(keys[child - 1], keys[child]) < 0) child++;
            if (!(c(d, keys[child - 1]) < 0)) break;
            keys[i - 1] = keys[child - 1];
            i = child;
        }
        keys[i - 1] = d;
    }
    static void HeapSort(T* keys, int n, Cmp& c)
    {
        for (int i = n >> 1; i >= 1; i--) DownHeap(keys, i, n, c);
        for (int i = n; i > 1; i--) { Swap(keys, 0, i - 1); DownHeap(keys, 1, i - 1, c); }
    }
};
// Array.Sort(idx, start, count, comparer)
template <class Cmp> static void ArraySort(int* idx, int start, int count, Cmp c)
{
    DotnetSort<int, Cmp>::Sort(idx + start, count, c);
}

// ------------------------------------------------------------------ Camera.cs
struct CameraOps {
    static float DegToRad(float d) { return d * (3.14159274f / 180.f); }      // :218 (XMath.PI is a float)
    // :193-205
    static void OrthoBasis(Float3 forward, Float3 upHint, Float3& u, Float3& v, Float3& w)
    {
        Float3 f = HNormalize(forward);
        Float3 up = upHint;
        if (hrt_abs(Dot(f, up)) > 0.999f)
        {
            up = Float3(0.f, 1.f, 0.f);
            if (hrt_abs(Dot(f, up)) > 0.999f) up = Float3(1.f, 0.f, 0.f);
        }
        u = HNormalize(Cross(f, up));
        v = HNormalize(Cross(u, f));
        w = Float3(-f.X, -f.Y, -f.Z);
    }
    // :184-191
    static void UpdateDerived(hrt_camera& c, float aspectIn, float fovYRadIn)
    {
        Float3 forward = HNormalize((Float3(c.lowerLeft) + Float3(c.horizontal) * 0.5f + Float3(c.vertical) * 0.5f) - Float3(c.origin));
        Float3 up = HNormalize(c.vertical);
        Float3 right = HNormalize(Cross(forward, up));
        c.forward = forward; c.up = up; c.right = right;
        c.aspect = aspectIn;
        c.fovYRadians = fovYRadIn;
    }
    // :19-47
    static hrt_camera CreateCamera(int width, int height, float fovDegrees)
    {
        float aspect = (float)width / (float)hrt_imax(1, height);
        float theta = fovDegrees * (3.14159274f / 180.f);
        float halfHeight = hrt_tan(0.5f * theta);
        float halfWidth = aspect * halfHeight;

        Float3 origin(0.f, 1.f, 3.f);
        Float3 lookAt(0.f, 0.5f, 0.f);
        Float3 upHint(0.f, 1.f, 0.f);

        Float3 w = HNormalize(origin - lookAt);
        Float3 u = HNormalize(Cross(upHint, w));
        Float3 v = Cross(w, u);

        Float3 lowerLeft = origin - u * halfWidth - v * halfHeight - w;
        Float3 horizontal = u * (2.f * halfWidth);
        Float3 vertical = v * (2.f * halfHeight);

        hrt_camera cam;
        std::memset(&cam, 0, sizeof(cam));
        cam.origin = origin; cam.lowerLeft = lowerLeft; cam.horizontal = horizontal; cam.vertical = vertical;
        UpdateDerived(cam, aspect, theta);
        return cam;
    }
    // :100-119
    static hrt_camera LookAt(Float3 origin, Float3 lookAt, Float3 up, float vfovDegrees, float aspect, float focusDist)
    {
        float theta = DegToRad(vfovDegrees);
        float halfHeight = hrt_tan(0.5f * theta);
        float halfWidth = aspect * halfHeight;

        Float3 forward = HNormalize(lookAt - origin);
        Float3 u, v, w;
        OrthoBasis(forward, up, u, v, w);

        hrt_camera c;
        c.origin = origin;
        c.horizontal = u * (2.f * halfWidth);
        c.vertical = v * (2.f * halfHeight);
        c.lowerLeft = origin - u * halfWidth - v * halfHeight + forward * focusDist;

        Float3 fwd = HNormalize((Float3(c.lowerLeft) + Float3(c.horizontal) * 0.5f + Float3(c.vertical) * 0.5f) - origin);
        c.forward = fwd;
        c.right = HNormalize(Cross(fwd, v));
        c.up = HNormalize(v);
        c.aspect = aspect;
        c.fovYRadians = theta;
        return c;
    }
    // :121-126
    static void Translate(hrt_camera& c, Float3 delta)
    {
        c.origin = Float3(c.origin) + delta;
        c.lowerLeft = Float3(c.lowerLeft) + delta;
        UpdateDerived(c, c.aspect, c.fovYRadians);
    }
    // RTRenderer.cs:241-263
    static void BakeCameraDerived(hrt_camera& c, int pixelW, int pixelH)
    {
        Float3 center = Float3(c.lowerLeft) + Float3(c.horizontal) * 0.5f + Float3(c.vertical) * 0.5f;
        Float3 forward = HNormalize(center - Float3(c.origin));
        Float3 up = HNormalize(c.vertical);
        Float3 right = HNormalize(Cross(forward, up));

        float focusDist = Length(center - Float3(c.origin));
        float halfHeight = 0.5f * Length(c.vertical);
        float tanHalfFov = (focusDist > 1e-6f) ? (halfHeight / focusDist) : halfHeight;
        float fovY = 2.f * hrt_atan(tanHalfFov);
        float aspect = (Length(c.horizontal) > 1e-6f && Length(c.vertical) > 1e-6f)
            ? (Length(c.horizontal) / Length(c.vertical))
            : ((float)pixelW / (float)hrt_imax(1, pixelH));

        c.forward = forward;
        c.up = up;
        c.right = right;
        c.fovYRadians = fovY;
        c.aspect = aspect;
    }
    // RTRenderer.cs:174-178
    static Float3 SunDir(float azimuth, float elevation)
    {
        return HNormalize(Float3(hrt_cos(azimuth) * hrt_cos(elevation), hrt_sin(elevation), hrt_sin(azimuth) * hrt_cos(elevation)));
    }
};

// ------------------------------------------------------------------ MeshHost (MeshLoaderOBJ.cs:21-41)
struct TextureSrc { int Width, Height; std::vector<uint8_t> BGRA; };
struct MeshHost {
    std::vector<hrt_float3> Positions;
    std::vector<hrt_mesh_tri> Triangles;
    std::vector<hrt_float2> Texcoords;
    std::vector<hrt_mesh_tri_uv> TriUVs;
    std::vector<int> TriMaterialIndex;
    std::vector<hrt_material> Materials;
    std::vector<TextureSrc> Textures;
};

// ------------------------------------------------------------------ Scene.cs
struct Scene {
    std::vector<hrt_bvh_node> _hTLASNodes;
    std::vector<int> _hTLASInstanceIndices;
    std::vector<hrt_instance> _hInstances;
    std::vector<hrt_bvh_node> _hBLASNodes;
    std::vector<int> _hSpherePrimIndices;
    std::vector<hrt_sphere> _hSpheres;
    std::vector<int> _hTriPrimIndices;
    std::vector<hrt_float3> _hMeshPositions;
    std::vector<hrt_mesh_tri> _hMeshTris;
    std::vector<hrt_float2> _hMeshTexcoords;
    std::vector<hrt_mesh_tri_uv> _hMeshTriUVs;
    std::vector<int> _hTriMaterialIndex;
    std::vector<hrt_material> _hMaterials;
    std::vector<hrt_tex_info> _hTexInfos;
    std::vector<hrt_rgba32> _hTexels;

    static hrt_affine3x4 Identity() { hrt_affine3x4 a; std::memset(&a, 0, sizeof(a)); a.m00 = 1.f; a.m11 = 1.f; a.m22 = 1.f; return a; }

    // :640-652
    static Float3 TransformPoint(const hrt_affine3x4& m, Float3 p) { return SceneDeviceViews::TransformPoint(m, p); }
    static Float3 TransformVector(const hrt_affine3x4& m, Float3 v) { return SceneDeviceViews::TransformVector(m, v); }

    // :98-109 (local function of BuildDefaultScene)
    int AddCheckerTexture(int w, int h, int step, hrt_rgba32 c0, hrt_rgba32 c1)
    {
        int offset = (int)_hTexels.size();
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
            {
                bool a = (((x / step) + (y / step)) & 1) == 0;
                _hTexels.push_back(a ? c0 : c1);
            }
        hrt_tex_info ti = {offset, w, h};
        _hTexInfos.push_back(ti);
        return (int)_hTexInfos.size() - 1;
    }
    // generic form of the same append (arbitrary RGBA texels)
    int AddTexture(int w, int h, const hrt_rgba32* px)
    {
        int offset = (int)_hTexels.size();
        _hTexels.insert(_hTexels.end(), px, px + (size_t)w * h);
        hrt_tex_info ti = {offset, w, h};
        _hTexInfos.push_back(ti);
        return (int)_hTexInfos.size() - 1;
    }

    void Clear()
    {
        _hBLASNodes.clear(); _hSpherePrimIndices.clear(); _hSpheres.clear(); _hTriPrimIndices.clear();
        _hMeshPositions.clear(); _hMeshTris.clear(); _hMeshTexcoords.clear(); _hMeshTriUVs.clear();
        _hTriMaterialIndex.clear(); _hMaterials.clear(); _hTexInfos.clear(); _hTexels.clear();
    }

    static hrt_material Mat(Float3 kd, int hasMap, int tex)
    {
        hrt_material m; std::memset(&m, 0, sizeof(m));
        m.Kd = kd; m.HasDiffuseMap = hasMap; m.DiffuseTexIndex = tex; m.Shading = HRT_SHADING_LAMBERT; m.IOR = 1.f;
        m.HasAlphaMap = 0; m.AlphaTexIndex = -1; m.AlphaCutoff = 0.5f; m.TwoSided = 0;
        return m;
    }
    static hrt_sphere Sph(Float3 c, float r, Float3 alb, hrt_material m, int shading, float ior)
    {
        hrt_sphere s; s.center = c; s.radius = r; s.albedo = alb; s.material = m; s.shading = shading; s.ior = ior; return s;
    }

    // :83-142
    void BuildDefaultScene()
    {
        Clear();
        hrt_rgba32 w255 = {255, 255, 255, 255}, g20 = {20, 20, 20, 255}, b40 = {40, 40, 200, 255}, y200 = {200, 200, 40, 255};
        int checker0 = AddCheckerTexture(256, 256, 16, w255, g20);
        int checker1 = AddCheckerTexture(256, 256, 8, b40, y200);

        hrt_material matGround = Mat(Float3(1.f, 1.f, 1.f), 1, checker0);
        hrt_material matRed = Mat(Float3(0.8f, 0.3f, 0.3f), 0, -1);
        hrt_material matGreen = Mat(Float3(0.3f, 0.8f, 0.3f), 0, -1);
        hrt_material matTex = Mat(Float3(1.f, 1.f, 1.f), 1, checker1);
        hrt_material matWhite = Mat(Float3(1.f, 1.f, 1.f), 0, -1);

        int ground = AddSphere(Sph(Float3(0.f, -1000.5f, 0.f), 1000.f, Float3(1.f, 1.f, 1.f), matGround, HRT_SHADING_LAMBERT, 1.f));
        int s0 = AddSphere(Sph(Float3(-0.9f, 0.5f, -0.2f), 0.5f, Float3(0.8f, 0.3f, 0.3f), matRed, HRT_SHADING_LAMBERT, 1.f));
        int s1 = AddSphere(Sph(Float3(0.9f, 0.35f, 0.2f), 0.35f, Float3(0.3f, 0.8f, 0.3f), matGreen, HRT_SHADING_LAMBERT, 1.f));
        int s2 = AddSphere(Sph(Float3(0.0f, 0.75f, 0.6f), 0.75f, Float3(1.f, 1.f, 1.f), matTex, HRT_SHADING_LAMBERT, 1.f));
        int sMirror = AddSphere(Sph(Float3(-1.8f, 0.5f, 0.8f), 0.5f, Float3(1.f, 1.f, 1.f), matWhite, HRT_SHADING_MIRROR, 1.f));
        int sGlass = AddSphere(Sph(Float3(1.8f, 0.5f, -0.8f), 0.5f, Float3(1.f, 1.f, 1.f), matWhite, HRT_SHADING_GLASS, 1.5f));

        std::vector<hrt_instance> inst;
        int ids[6] = {ground, s0, s1, s2, sMirror, sGlass};
        for (int i = 0; i < 6; i++) inst.push_back(BuildSphereInstance(&ids[i], 1, Identity()));
        _hInstances = inst;
        // TryAddSponzaFromKnownLocations(): no asset in this build -> returns false (:654-674)
        RebuildTLAS();
    }

    // :315-321
    int AddSphere(const hrt_sphere& s)
    {
        int id = (int)_hSpheres.size();
        _hSpheres.push_back(s);
        _hSpherePrimIndices.push_back(id);
        return id;
    }

    // :323-356
    hrt_instance BuildSphereInstance(const int* sphereIds, int nIds, const hrt_affine3x4& objectToWorld)
    {
        Float3 bmin(FLT_MAX, FLT_MAX, FLT_MAX);
        Float3 bmax(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = 0; i < nIds; i++)
        {
            hrt_sphere s = _hSpheres[sphereIds[i]];
            bmin = HMin(bmin, Float3(s.center.X - s.radius, s.center.Y - s.radius, s.center.Z - s.radius));
            bmax = HMax(bmax, Float3(s.center.X + s.radius, s.center.Y + s.radius, s.center.Z + s.radius));
        }

        int primStart = sphereIds[0];
        int primCount = nIds;

        int blasStart = (int)_hBLASNodes.size();
        BuildBLAS_Spheres(_hBLASNodes, _hSpherePrimIndices, primStart, primCount, _hSpheres);
        int blasCount = (int)_hBLASNodes.size() - blasStart;

        Float3 wmin, wmax;
        TransformAABB(objectToWorld, bmin, bmax, wmin, wmax);
        float uniScale;
        hrt_affine3x4 worldToObject = InvertRigidOrUniform(objectToWorld, uniScale);

        hrt_instance inst; std::memset(&inst, 0, sizeof(inst));
        inst.type = HRT_BLAS_SPHERESET;
        inst.blasRoot = blasStart;
        inst.blasNodeCount = blasCount;
        inst.primIndexFirst = primStart;
        inst.primIndexCount = primCount;
        inst.objectToWorld = objectToWorld;
        inst.worldToObject = worldToObject;
        inst.uniformScale = uniScale;
        inst.worldBoundsMin = wmin;
        inst.worldBoundsMax = wmax;
        return inst;
    }

    // :144-256, from `int baseVertex = ...` on (the MeshLoaderOBJ.Load call is the caller's)
    void LoadMeshInstance(const MeshHost& mesh, const hrt_affine3x4& objectToWorld)
    {
        int baseVertex = (int)_hMeshPositions.size();
        int baseTri = (int)_hMeshTris.size();
        int baseUV = (int)_hMeshTexcoords.size();
        int baseMat = (int)_hMaterials.size();

        _hMeshPositions.insert(_hMeshPositions.end(), mesh.Positions.begin(), mesh.Positions.end());
        _hMeshTexcoords.insert(_hMeshTexcoords.end(), mesh.Texcoords.begin(), mesh.Texcoords.end());

        for (int i = 0; i < (int)mesh.Triangles.size(); i++)
        {
            hrt_mesh_tri t = mesh.Triangles[i];
            t.i0 += baseVertex; t.i1 += baseVertex; t.i2 += baseVertex;
            _hMeshTris.push_back(t);

            hrt_mesh_tri_uv tuv = mesh.TriUVs[i];
            tuv.t0 += baseUV; tuv.t1 += baseUV; tuv.t2 += baseUV;
            _hMeshTriUVs.push_back(tuv);

            int matIndexLocal = (i < (int)mesh.TriMaterialIndex.size()) ? mesh.TriMaterialIndex[i] : 0;
            int matIndexGlobal = baseMat + matIndexLocal;
            _hTriMaterialIndex.push_back(matIndexGlobal);

            _hTriPrimIndices.push_back(baseTri + i);
        }

        int localMatCount = (int)mesh.Materials.size();
        std::vector<hrt_material> matRemapped;
        for (int i = 0; i < localMatCount; i++)
        {
            hrt_material m = mesh.Materials[i];

            if (m.HasDiffuseMap != 0 && m.DiffuseTexIndex >= 0 && m.DiffuseTexIndex < (int)mesh.Textures.size())
            {
                const TextureSrc& src = mesh.Textures[m.DiffuseTexIndex];
                int start = (int)_hTexels.size();
                for (size_t p = 0; p < src.BGRA.size(); p += 4)
                {
                    hrt_rgba32 px; px.B = src.BGRA[p + 0]; px.G = src.BGRA[p + 1]; px.R = src.BGRA[p + 2]; px.A = src.BGRA[p + 3];
                    _hTexels.push_back(px);
                }
                int texIndexGlobal = (int)_hTexInfos.size();
                hrt_tex_info ti = {start, src.Width, src.Height};
                _hTexInfos.push_back(ti);
                m.DiffuseTexIndex = texIndexGlobal;
                m.HasDiffuseMap = 1;
            }
            else { m.HasDiffuseMap = 0; m.DiffuseTexIndex = -1; }

            if (m.HasAlphaMap != 0 && m.AlphaTexIndex >= 0 && m.AlphaTexIndex < (int)mesh.Textures.size())
            {
                const TextureSrc& srcA = mesh.Textures[m.AlphaTexIndex];
                int startA = (int)_hTexels.size();
                for (size_t p = 0; p < srcA.BGRA.size(); p += 4)
                {
                    hrt_rgba32 px; px.B = srcA.BGRA[p + 0]; px.G = srcA.BGRA[p + 1]; px.R = srcA.BGRA[p + 2]; px.A = srcA.BGRA[p + 3];
                    _hTexels.push_back(px);
                }
                int texIndexGlobalA = (int)_hTexInfos.size();
                hrt_tex_info ti = {startA, srcA.Width, srcA.Height};
                _hTexInfos.push_back(ti);
                m.AlphaTexIndex = texIndexGlobalA;
                m.HasAlphaMap = 1;
            }
            else { m.HasAlphaMap = 0; m.AlphaTexIndex = -1; }

            matRemapped.push_back(m);
        }
        _hMaterials.insert(_hMaterials.end(), matRemapped.begin(), matRemapped.end());

        // MeshGlobal.SetTris(_hMeshTris): builders below resolve triangles through _hMeshTris
        int blasStart = (int)_hBLASNodes.size();
        BuildBLAS_Triangles(_hBLASNodes, _hTriPrimIndices, baseTri, (int)mesh.Triangles.size(), _hMeshPositions);
        int blasCount = (int)_hBLASNodes.size() - blasStart;

        Float3 bmin, bmax, wmin, wmax;
        ComputeMeshBounds(mesh.Positions, mesh.Triangles, bmin, bmax);
        TransformAABB(objectToWorld, bmin, bmax, wmin, wmax);

        float uniScale;
        hrt_affine3x4 worldToObject = InvertRigidOrUniform(objectToWorld, uniScale);

        hrt_instance instRec; std::memset(&instRec, 0, sizeof(instRec));
        instRec.type = HRT_BLAS_TRIMESH;
        instRec.blasRoot = blasStart;
        instRec.blasNodeCount = blasCount;
        instRec.primIndexFirst = baseTri;
        instRec.primIndexCount = (int)mesh.Triangles.size();
        instRec.objectToWorld = objectToWorld;
        instRec.worldToObject = worldToObject;
        instRec.uniformScale = uniScale;
        instRec.worldBoundsMin = wmin;
        instRec.worldBoundsMax = wmax;

        _hInstances.push_back(instRec);
        RebuildTLAS();
    }

    // :358-368
    void RebuildTLAS()
    {
        int n = (int)_hInstances.size();
        std::vector<int> idx(n);
        for (int i = 0; i < n; i++) idx[i] = i;
        std::vector<hrt_bvh_node> outNodes;
        outNodes.reserve(2 * (size_t)n);
        if (n > 0) BuildTLASNodeRecursive(_hInstances, idx.data(), 0, n, outNodes, -1);
        else
        {
            // the reference would still emit one empty-bounds leaf for n == 0 (count 0 <= 2)
            BuildTLASNodeRecursive(_hInstances, idx.data(), 0, 0, outNodes, -1);
        }
        _hTLASNodes = outNodes;
        _hTLASInstanceIndices = idx;
    }

    // :381-396
    void BuildBLAS_Spheres(std::vector<hrt_bvh_node>& outBLAS, std::vector<int>& primIdx, int primStart, int primCount, const std::vector<hrt_sphere>& spheres)
    {
        std::vector<int> idx(primCount);
        for (int i = 0; i < primCount; i++) idx[i] = primStart + i;

        std::vector<Float3> bmin(primCount), bmax(primCount);
        for (int i = 0; i < primCount; i++)
        {
            hrt_sphere s = spheres[primIdx[primStart + i]];
            bmin[i] = Float3(s.center.X - s.radius, s.center.Y - s.radius, s.center.Z - s.radius);
            bmax[i] = Float3(s.center.X + s.radius, s.center.Y + s.radius, s.center.Z + s.radius);
        }
        BuildBLASNodeRecursive(outBLAS, primIdx, idx.data(), 0, primCount, bmin.data(), bmax.data(), -1, true);
    }
    // :398-403
    void BuildBLAS_Triangles(std::vector<hrt_bvh_node>& outBLAS, std::vector<int>& primIdx, int primStart, int primCount, const std::vector<hrt_float3>& positions)
    {
        (void)positions;
        std::vector<int> idx(primCount);
        for (int i = 0; i < primCount; i++) idx[i] = primStart + i;
        BuildBLASNodeRecursive(outBLAS, primIdx, idx.data(), 0, primCount, nullptr, nullptr, -1, false);
    }

    // :597-614
    void BoundsOfTriangle(int triIndex, Float3& mn, Float3& mx) const
    {
        hrt_mesh_tri tri = _hMeshTris[triIndex];
        Float3 v0 = _hMeshPositions[tri.i0], v1 = _hMeshPositions[tri.i1], v2 = _hMeshPositions[tri.i2];
        mn = HMin(v0, HMin(v1, v2));
        mx = HMax(v0, HMax(v1, v2));
    }
    Float3 CenterOfTriangle(int triIndex) const
    {
        hrt_mesh_tri tri = _hMeshTris[triIndex];
        Float3 v0 = _hMeshPositions[tri.i0], v1 = _hMeshPositions[tri.i1], v2 = _hMeshPositions[tri.i2];
        return Float3((v0.X + v1.X + v2.X) / 3.f, (v0.Y + v1.Y + v2.Y) / 3.f, (v0.Z + v1.Z + v2.Z) / 3.f);
    }

    // :512-543 comparators (return <0, 0, >0)
    struct CmpSpheres {
        int axis; const std::vector<int>* primIdx; const std::vector<hrt_sphere>* spheres;
        int operator()(int a, int b) const
        {
            int ia = (*primIdx)[a], ib = (*primIdx)[b];
            const hrt_sphere& sa = (*spheres)[ia]; const hrt_sphere& sb = (*spheres)[ib];
            float ca = axis == 0 ? sa.center.X : (axis == 1 ? sa.center.Y : sa.center.Z);
            float cb = axis == 0 ? sb.center.X : (axis == 1 ? sb.center.Y : sb.center.Z);
            if (ca < cb) return -1; if (ca > cb) return 1; return 0;
        }
    };
    struct CmpTris {
        int axis; const std::vector<int>* primIdx; const Scene* sc;
        int operator()(int a, int b) const
        {
            int ia = (*primIdx)[a], ib = (*primIdx)[b];
            Float3 ca = sc->CenterOfTriangle(ia);
            Float3 cb = sc->CenterOfTriangle(ib);
            float va = axis == 0 ? ca.X : (axis == 1 ? ca.Y : ca.Z);
            float vb = axis == 0 ? cb.X : (axis == 1 ? cb.Y : cb.Z);
            if (va < vb) return -1; if (va > vb) return 1; return 0;
        }
    };
    // :545-558
    struct CmpInst {
        int axis; const std::vector<hrt_instance>* inst;
        int operator()(int a, int b) const
        {
            Float3 ca = Center((*inst)[a].worldBoundsMin, (*inst)[a].worldBoundsMax);
            Float3 cb = Center((*inst)[b].worldBoundsMin, (*inst)[b].worldBoundsMax);
            float va = axis == 0 ? ca.X : (axis == 1 ? ca.Y : ca.Z);
            float vb = axis == 0 ? cb.X : (axis == 1 ? cb.Y : cb.Z);
            if (va < vb) return -1; if (va > vb) return 1; return 0;
        }
    };

    // :405-467
    int BuildBLASNodeRecursive(std::vector<hrt_bvh_node>& outBLAS, std::vector<int>& primIdx, int* idx, int start, int count, const Float3* bminPre, const Float3* bmaxPre, int parentSkip, bool spheres)
    {
        int nodeIndex = (int)outBLAS.size();
        hrt_bvh_node node; std::memset(&node, 0, sizeof(node));
        node.first = -1; node.count = 0; node.left = -1; node.right = -1; node.skipIndex = parentSkip;

        Float3 nbMin(FLT_MAX, FLT_MAX, FLT_MAX);
        Float3 nbMax(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        if (bminPre != nullptr)
        {
            for (int i = start; i < start + count; i++)
            {
                nbMin = HMin(nbMin, bminPre[i]);     // by POSITION, not through idx[] (reference quirk)
                nbMax = HMax(nbMax, bmaxPre[i]);
            }
        }
        else
        {
            for (int i = start; i < start + count; i++)
            {
                int triIndex = primIdx[idx[i]];
                Float3 mn, mx;
                BoundsOfTriangle(triIndex, mn, mx);
                nbMin = HMin(nbMin, mn);
                nbMax = HMax(nbMax, mx);
            }
        }

        node.boundsMin = nbMin;
        node.boundsMax = nbMax;
        outBLAS.push_back(node);

        const int LeafThreshold = 4;
        if (count <= LeafThreshold)
        {
            int leafStart = (int)primIdx.size();
            for (int i = start; i < start + count; i++) { int v = primIdx[idx[i]]; primIdx.push_back(v); }
            hrt_bvh_node leaf = outBLAS[nodeIndex];
            leaf.first = leafStart; leaf.count = count; leaf.skipIndex = parentSkip;
            outBLAS[nodeIndex] = leaf;
            return nodeIndex;
        }

        Float3 extent = nbMax - nbMin;
        int axis = 0;
        if (extent.Y > extent.X && extent.Y >= extent.Z) axis = 1;
        else if (extent.Z > extent.X && extent.Z >= extent.Y) axis = 2;

        if (spheres) { CmpSpheres c = {axis, &primIdx, &_hSpheres}; ArraySort(idx, start, count, c); }
        else         { CmpTris c = {axis, &primIdx, this};          ArraySort(idx, start, count, c); }

        int mid = start + (count >> 1);

        int rightRoot = BuildBLASNodeRecursive(outBLAS, primIdx, idx, mid, count - (mid - start), bminPre, bmaxPre, parentSkip, spheres);
        int leftRoot = BuildBLASNodeRecursive(outBLAS, primIdx, idx, start, mid - start, bminPre, bmaxPre, rightRoot, spheres);

        hrt_bvh_node inner = outBLAS[nodeIndex];
        inner.left = leftRoot; inner.right = rightRoot; inner.skipIndex = parentSkip;
        outBLAS[nodeIndex] = inner;
        return nodeIndex;
    }

    // :469-510
    int BuildTLASNodeRecursive(const std::vector<hrt_instance>& inst, int* idx, int start, int count, std::vector<hrt_bvh_node>& outNodes, int parentSkip)
    {
        int nodeIndex = (int)outNodes.size();
        hrt_bvh_node node; std::memset(&node, 0, sizeof(node));
        node.first = -1; node.count = 0; node.left = -1; node.right = -1; node.skipIndex = parentSkip;

        Float3 nbMin(FLT_MAX, FLT_MAX, FLT_MAX);
        Float3 nbMax(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = start; i < start + count; i++)
        {
            const hrt_instance& r = inst[idx[i]];
            nbMin = HMin(nbMin, r.worldBoundsMin);
            nbMax = HMax(nbMax, r.worldBoundsMax);
        }
        node.boundsMin = nbMin; node.boundsMax = nbMax;
        outNodes.push_back(node);

        const int LeafThreshold = 2;
        if (count <= LeafThreshold)
        {
            hrt_bvh_node leaf = outNodes[nodeIndex];
            leaf.first = start; leaf.count = count; leaf.skipIndex = parentSkip;
            outNodes[nodeIndex] = leaf;
            return nodeIndex;
        }

        Float3 extent = nbMax - nbMin;
        int axis = 0;
        if (extent.Y > extent.X && extent.Y >= extent.Z) axis = 1;
        else if (extent.Z > extent.X && extent.Z >= extent.Y) axis = 2;

        CmpInst c = {axis, &inst};
        ArraySort(idx, start, count, c);

        int mid = start + (count >> 1);
        int rightRoot = BuildTLASNodeRecursive(inst, idx, mid, count - (mid - start), outNodes, parentSkip);
        int leftRoot = BuildTLASNodeRecursive(inst, idx, start, mid - start, outNodes, rightRoot);

        hrt_bvh_node inner = outNodes[nodeIndex];
        inner.left = leftRoot; inner.right = rightRoot; inner.skipIndex = parentSkip;
        outNodes[nodeIndex] = inner;
        return nodeIndex;
    }

    // :560-580
    static void TransformAABB(const hrt_affine3x4& m, Float3 bmin, Float3 bmax, Float3& outMin, Float3& outMax)
    {
        Float3 c[8];
        c[0] = Float3(bmin.X, bmin.Y, bmin.Z);
        c[1] = Float3(bmax.X, bmin.Y, bmin.Z);
        c[2] = Float3(bmin.X, bmax.Y, bmin.Z);
        c[3] = Float3(bmin.X, bmin.Y, bmax.Z);
        c[4] = Float3(bmax.X, bmax.Y, bmin.Z);
        c[5] = Float3(bmin.X, bmax.Y, bmax.Z);
        c[6] = Float3(bmax.X, bmin.Y, bmax.Z);
        c[7] = Float3(bmax.X, bmax.Y, bmax.Z);
        Float3 mn(FLT_MAX, FLT_MAX, FLT_MAX);
        Float3 mx(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = 0; i < 8; i++)
        {
            Float3 w = TransformPoint(m, c[i]);
            mn = HMin(mn, w);
            mx = HMax(mx, w);
        }
        outMin = mn; outMax = mx;
    }
    // :582-595
    static void ComputeMeshBounds(const std::vector<hrt_float3>& pos, const std::vector<hrt_mesh_tri>& tris, Float3& bmin, Float3& bmax)
    {
        bmin = Float3(FLT_MAX, FLT_MAX, FLT_MAX);
        bmax = Float3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (size_t i = 0; i < tris.size(); i++)
        {
            hrt_mesh_tri t = tris[i];
            Float3 v0 = pos[t.i0], v1 = pos[t.i1], v2 = pos[t.i2];
            bmin = HMin(bmin, HMin(v0, HMin(v1, v2)));
            bmax = HMax(bmax, HMax(v0, HMax(v1, v2)));
        }
    }
    // :616-638
    static hrt_affine3x4 InvertRigidOrUniform(const hrt_affine3x4& m, float& uniformScale)
    {
        float sx = Length(Float3(m.m00, m.m10, m.m20));
        float sy = Length(Float3(m.m01, m.m11, m.m21));
        float sz = Length(Float3(m.m02, m.m12, m.m22));
        uniformScale = (sx + sy + sz) / 3.f;
        float inv = uniformScale > 0.f ? 1.f / uniformScale : 1.f;

        Float3 r0 = HNormalize(Float3(m.m00, m.m10, m.m20));
        Float3 r1 = HNormalize(Float3(m.m01, m.m11, m.m21));
        Float3 r2 = HNormalize(Float3(m.m02, m.m12, m.m22));

        hrt_affine3x4 invM; std::memset(&invM, 0, sizeof(invM));
        invM.m00 = r0.X * inv; invM.m01 = r1.X * inv; invM.m02 = r2.X * inv; invM.m03 = 0.f;
        invM.m10 = r0.Y * inv; invM.m11 = r1.Y * inv; invM.m12 = r2.Y * inv; invM.m13 = 0.f;
        invM.m20 = r0.Z * inv; invM.m21 = r1.Z * inv; invM.m22 = r2.Z * inv; invM.m23 = 0.f;

        Float3 t(m.m03, m.m13, m.m23);
        Float3 it = TransformVector(invM, t) * -1.f;
        invM.m03 = it.X; invM.m13 = it.Y; invM.m23 = it.Z;
        return invM;
    }

    // A moved instance: what BuildSphereInstance / LoadObjInstance would have written for this objectToWorld
    // (Scene.cs:395-402, :236-252), with the box of the instance's BLAS root node as the object-space bounds
    // (both builders make the root the union of all primitive boxes, which is what those two functions transform).
    void SetInstanceTransform(int id, const hrt_affine3x4& objectToWorld)
    {
        hrt_instance& inst = _hInstances[(size_t)id];
        Float3 bmin(0.f, 0.f, 0.f), bmax(0.f, 0.f, 0.f);
        if (inst.blasNodeCount > 0) { bmin = Float3(_hBLASNodes[(size_t)inst.blasRoot].boundsMin); bmax = Float3(_hBLASNodes[(size_t)inst.blasRoot].boundsMax); }
        Float3 wmin, wmax;
        TransformAABB(objectToWorld, bmin, bmax, wmin, wmax);
        float uniScale;
        hrt_affine3x4 worldToObject = InvertRigidOrUniform(objectToWorld, uniScale);
        inst.objectToWorld = objectToWorld;
        inst.worldToObject = worldToObject;
        inst.uniformScale = uniScale;
        inst.worldBoundsMin = wmin;
        inst.worldBoundsMax = wmax;
    }

    void GetDesc(hrt_scene_desc& d) const
    {
        std::memset(&d, 0, sizeof(d));
#define ORC_SET(field, vec) d.field = (vec).empty() ? nullptr : (vec).data(); d.n_##field = (int64_t)(vec).size()
        ORC_SET(tlasNodes, _hTLASNodes);
        ORC_SET(tlasInstanceIndices, _hTLASInstanceIndices);
        ORC_SET(instances, _hInstances);
        ORC_SET(blasNodes, _hBLASNodes);
        ORC_SET(spherePrimIdx, _hSpherePrimIndices);
        ORC_SET(spheres, _hSpheres);
        ORC_SET(triPrimIdx, _hTriPrimIndices);
        ORC_SET(meshPositions, _hMeshPositions);
        ORC_SET(meshTris, _hMeshTris);
        ORC_SET(meshTexcoords, _hMeshTexcoords);
        ORC_SET(meshTriUVs, _hMeshTriUVs);
        ORC_SET(triMatIndex, _hTriMaterialIndex);
        ORC_SET(materials, _hMaterials);
        ORC_SET(texels, _hTexels);
        ORC_SET(texInfos, _hTexInfos);
#undef ORC_SET
    }
};

} // namespace orc
#endif
