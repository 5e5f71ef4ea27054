// hrt_host.cpp -- host side ABOVE the C ABI, in C++ because the reference's own host
// language (C# / .NET 8) has no toolchain in this image.  Mirrors, for the render path
// only, the host classes a C# maintainer would keep unchanged:
//
//   hrth_scene_*    Engine/Scene.cs: AddSphere :315-321, BuildSphereInstance :323-356,
//                   LoadObjInstance (array-append half) :151-256, RebuildTLAS :358-368,
//                   BuildBLAS_* / Build*NodeRecursive :381-510, comparators :512-558,
//                   TransformAABB :560-580, ComputeMeshBounds :582-595,
//                   InvertRigidOrUniform :616-638, BuildDefaultScene :83-142
//   hrth_camera_*   Engine/Camera.cs: CreateCamera :19-47, look-at ctor :100-119,
//                   Translate :121-126, UpdateDerived :184-191;
//                   RTRenderer.BakeCameraDerived Engine/RTRenderer.cs:241-263
//   hrth_sun_dir    Engine/RTRenderer.cs:174-178
//
// Output arrays are byte-identical to what the reference's builders emit (same node
// order [node][right subtree][left subtree], same skip pointers, same leaf appends),
// but the build is organised differently: per-primitive bounds and centroid keys are
// computed once, the median split sorts plain (key,id) arrays, subtree sizes are known
// in closed form (the split is count>>1), so node and primitive-index slots are
// pre-assigned and the two children of the upper levels are built on separate threads.
//
// Array.Sort (unstable introsort of .NET 8's ArraySortHelper<T>) decides topology under
// centroid ties; it is restated in dotnet_introsort() below.
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <future>
#include <map>
#include <thread>
#include <vector>
#include "../../include/hrt_types.h"
#include "../../include/hrt_math.h"

namespace {

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 v3(const hrt_float3& f) { return v3(f.X, f.Y, f.Z); }
inline hrt_float3 f3(V3 v) { hrt_float3 r = {v.x, v.y, v.z}; return r; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
// Host code of the reference: Min / Max are .NET's Math.Min / Max (a NaN operand is returned), not the kernels' minNum.
inline V3 vmin(V3 a, V3 b) { return v3(hrt_host_fmin(a.x, b.x), hrt_host_fmin(a.y, b.y), hrt_host_fmin(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return v3(hrt_host_fmax(a.x, b.x), hrt_host_fmax(a.y, b.y), hrt_host_fmax(a.z, b.z)); }
inline float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 vcross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline V3 vnorm(V3 v) { float inv = hrt_rsqrt(hrt_host_fmax(1e-20f, v.x * v.x + v.y * v.y + v.z * v.z)); return v3(v.x * inv, v.y * inv, v.z * inv); }
inline float vlen(V3 v) { return hrt_sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
inline V3 xpoint(const hrt_affine3x4& m, V3 p)
{
    return v3(m.m00 * p.x + m.m01 * p.y + m.m02 * p.z + m.m03, m.m10 * p.x + m.m11 * p.y + m.m12 * p.z + m.m13, m.m20 * p.x + m.m21 * p.y + m.m22 * p.z + m.m23);
}
inline V3 xvector(const hrt_affine3x4& m, V3 v)
{
    return v3(m.m00 * v.x + m.m01 * v.y + m.m02 * v.z, m.m10 * v.x + m.m11 * v.y + m.m12 * v.z, m.m20 * v.x + m.m21 * v.y + m.m22 * v.z);
}

// ---- .NET 8 Array.Sort(int[], index, length, IComparer<int>) on ids ordered by key[id] ----
// (System.Collections.Generic.ArraySortHelper<T>: IntroSort, threshold 16, depth 2*(log2 n + 1))
struct KeyLess {
    const float* key;
    inline int cmp(int a, int b) const { float ka = key[a], kb = key[b]; return ka < kb ? -1 : (ka > kb ? 1 : 0); }
};
inline void swap_if_greater(int* k, const KeyLess& c, int i, int j) { if (c.cmp(k[i], k[j]) > 0) { int t = k[i]; k[i] = k[j]; k[j] = t; } }
void down_heap(int* k, int i, int n, const KeyLess& c)
{
    int d = k[i - 1];
    while (i <= (n >> 1)) {
        int child = 2 * i;
        if (child < n && c.cmp(k[child - 1], k[child]) < 0) child++;
        if (!(c.cmp(d, k[child - 1]) < 0)) break;
        k[i - 1] = k[child - 1];
        i = child;
    }
    k[i - 1] = d;
}
void intro(int* k, int n, int depth, const KeyLess& c)
{
    while (n > 1) {
        if (n <= 16) {
            if (n == 2) { swap_if_greater(k, c, 0, 1); return; }
            if (n == 3) { swap_if_greater(k, c, 0, 1); swap_if_greater(k, c, 0, 2); swap_if_greater(k, c, 1, 2); return; }
            for (int i = 0; i < n - 1; i++) {           // insertion sort
                int t = k[i + 1], j = i;
                while (j >= 0 && c.cmp(t, k[j]) < 0) { k[j + 1] = k[j]; j--; }
                k[j + 1] = t;
            }
            return;
        }
        if (depth == 0) {                               // heap sort
            for (int i = n >> 1; i >= 1; i--) down_heap(k, i, n, c);
            for (int i = n; i > 1; i--) { int t = k[0]; k[0] = k[i - 1]; k[i - 1] = t; down_heap(k, 1, i - 1, c); }
            return;
        }
        depth--;
        int hi = n - 1, mid = hi >> 1;                  // median of three, pivot parked at hi-1
        swap_if_greater(k, c, 0, mid); swap_if_greater(k, c, 0, hi); swap_if_greater(k, c, mid, hi);
        int pivot = k[mid];
        { int t = k[mid]; k[mid] = k[hi - 1]; k[hi - 1] = t; }
        int left = 0, right = hi - 1;
        while (left < right) {
            while (c.cmp(k[++left], pivot) < 0) {}
            while (c.cmp(pivot, k[--right]) < 0) {}
            if (left >= right) break;
            int t = k[left]; k[left] = k[right]; k[right] = t;
        }
        if (left != hi - 1) { int t = k[left]; k[left] = k[hi - 1]; k[hi - 1] = t; }
        intro(k + left + 1, n - (left + 1), depth, c);
        n = left;
    }
}
void dotnet_introsort(int* ids, int n, const float* key)
{
    if (n < 2) return;
    int lg = 0; for (uint32_t v = (uint32_t)n; v >>= 1;) lg++;
    KeyLess c = {key};
    intro(ids, n, 2 * (lg + 1), c);
}

// ---- skip-pointer median-split BVH (Scene.cs:405-510), slot-preassigned ----------------
struct BvhJob {
    int leafMax;                 // 4 for BLAS (:436), 2 for TLAS (:486)
    bool boundsByPosition;       // sphere BLAS quirk: bounds are read by array POSITION (:413-419), SURVEY F4
    const V3* bmin; const V3* bmax;        // per local item
    const float* key[3];         // centroid per axis, per local item
    int* ids;                    // permutation of local items (sorted in place)
    hrt_bvh_node* nodes;         // output slots of this tree
    int nodeBase;                // global index of nodes[0]
    // leaf payload: leafFirst[leaf slot] semantics differ (BLAS appends prim ids; TLAS points into ids)
    int* leafOut;                // BLAS: destination of appended primitive ids (count = n); TLAS: nullptr
    int leafBase;                // BLAS: global index of leafOut[0]
    const int* itemValue;        // BLAS: primIdx value of each local item
    std::map<int, int>* sizes;   // memo of node counts
};

int subtree_nodes(int count, int leafMax, std::map<int, int>& memo)
{
    if (count <= leafMax) return 1;
    auto it = memo.find(count);
    if (it != memo.end()) return it->second;
    int half = count >> 1;
    int r = 1 + subtree_nodes(count - half, leafMax, memo) + subtree_nodes(half, leafMax, memo);
    memo[count] = r;
    return r;
}

// builds the subtree over ids[start, start+count) into node slot `slot` (local), leaf slot `lslot`
void build_subtree(const BvhJob& J, int start, int count, int slot, int lslot, int parentSkip, int parallelDepth)
{
    hrt_bvh_node node;
    std::memset(&node, 0, sizeof(node));
    V3 mn = v3(FLT_MAX, FLT_MAX, FLT_MAX), mx = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = start; i < start + count; i++) {
        int it = J.boundsByPosition ? i : J.ids[i];
        mn = vmin(mn, J.bmin[it]);
        mx = vmax(mx, J.bmax[it]);
    }
    node.boundsMin = f3(mn); node.boundsMax = f3(mx);
    node.left = -1; node.right = -1; node.first = -1; node.count = 0; node.skipIndex = parentSkip;

    if (count <= J.leafMax) {
        if (J.leafOut) {
            for (int i = 0; i < count; i++) J.leafOut[lslot + i] = J.itemValue[J.ids[start + i]];
            node.first = J.leafBase + lslot;
        } else node.first = start;
        node.count = count;
        J.nodes[slot] = node;
        return;
    }
    V3 ext = mx - mn;
    int axis = 0;
    if (ext.y > ext.x && ext.y >= ext.z) axis = 1;
    else if (ext.z > ext.x && ext.z >= ext.y) axis = 2;
    dotnet_introsort(J.ids + start, count, J.key[axis]);

    int half = count >> 1;                       // mid = start + (count >> 1)
    int rightCount = count - half;
    int rightSlot = slot + 1;                    // right child is built (and numbered) first
    int leftSlot = rightSlot + subtree_nodes(rightCount, J.leafMax, *J.sizes);
    node.right = J.nodeBase + rightSlot;
    node.left = J.nodeBase + leftSlot;
    J.nodes[slot] = node;
    // right subtree: items [start+half, start+count), skip = parentSkip, leaves appended first
    // left subtree:  items [start, start+half),      skip = right root
    if (parallelDepth > 0 && count > 8192) {
        // sizes map is only read below this point for counts already memoised by the call above
        auto fut = std::async(std::launch::async, [&]() { build_subtree(J, start + half, rightCount, rightSlot, lslot, parentSkip, parallelDepth - 1); });
        build_subtree(J, start, half, leftSlot, lslot + rightCount, J.nodeBase + rightSlot, parallelDepth - 1);
        fut.get();
    } else {
        build_subtree(J, start + half, rightCount, rightSlot, lslot, parentSkip, 0);
        build_subtree(J, start, half, leftSlot, lslot + rightCount, J.nodeBase + rightSlot, 0);
    }
}

void warm_sizes(int count, int leafMax, std::map<int, int>& memo)
{   // memoise every subtree size that can occur so worker threads only read the map
    if (count <= leafMax) return;
    subtree_nodes(count, leafMax, memo);
    int half = count >> 1;
    if (memo.find(-count) != memo.end()) return;
    memo[-count] = 1;    // visited marker
    warm_sizes(count - half, leafMax, memo);
    warm_sizes(half, leafMax, memo);
}

struct HostScene {
    std::vector<hrt_bvh_node> tlasNodes;
    std::vector<int32_t> tlasInst;
    std::vector<hrt_instance> instances;
    std::vector<hrt_bvh_node> blasNodes;
    std::vector<int32_t> spherePrimIdx;
    std::vector<hrt_sphere> spheres;
    std::vector<int32_t> triPrimIdx;
    std::vector<hrt_float3> meshPositions;
    std::vector<hrt_mesh_tri> meshTris;
    std::vector<hrt_float2> meshTexcoords;
    std::vector<hrt_mesh_tri_uv> meshTriUVs;
    std::vector<int32_t> triMatIndex;
    std::vector<hrt_material> materials;
    std::vector<hrt_rgba32> texels;
    std::vector<hrt_tex_info> texInfos;

    void clear()
    {
        tlasNodes.clear(); tlasInst.clear(); instances.clear(); blasNodes.clear(); spherePrimIdx.clear(); spheres.clear();
        triPrimIdx.clear(); meshPositions.clear(); meshTris.clear(); meshTexcoords.clear(); meshTriUVs.clear();
        triMatIndex.clear(); materials.clear(); texels.clear(); texInfos.clear();
    }

    // builds one BLAS over n local items; appends nodes + leaf prim ids to the scene lists
    void build_blas(int n, const std::vector<V3>& bmin, const std::vector<V3>& bmax, const std::vector<float> key[3],
                    const std::vector<int>& itemValue, bool byPosition, std::vector<int32_t>& primIdx)
    {
        std::map<int, int> sizes;
        warm_sizes(n, 4, sizes);
        int nNodes = subtree_nodes(n, 4, sizes);
        size_t nodeBase = blasNodes.size(), leafBase = primIdx.size();
        blasNodes.resize(nodeBase + nNodes);
        primIdx.resize(leafBase + n);
        std::vector<int> ids(n);
        for (int i = 0; i < n; i++) ids[i] = i;
        BvhJob J;
        J.leafMax = 4; J.boundsByPosition = byPosition; J.bmin = bmin.data(); J.bmax = bmax.data();
        J.key[0] = key[0].data(); J.key[1] = key[1].data(); J.key[2] = key[2].data();
        J.ids = ids.data(); J.nodes = blasNodes.data() + nodeBase; J.nodeBase = (int)nodeBase;
        J.leafOut = primIdx.data() + leafBase; J.leafBase = (int)leafBase; J.itemValue = itemValue.data(); J.sizes = &sizes;
        int par = 0;
        unsigned hw = std::thread::hardware_concurrency();
        while ((1u << par) < hw && par < 6) par++;
        build_subtree(J, 0, n, 0, 0, -1, n > 8192 ? par : 0);
    }

    static void transform_aabb(const hrt_affine3x4& m, V3 bmin, V3 bmax, V3& omin, V3& omax)   // Scene.cs:560-580
    {
        V3 c[8] = {v3(bmin.x, bmin.y, bmin.z), v3(bmax.x, bmin.y, bmin.z), v3(bmin.x, bmax.y, bmin.z), v3(bmin.x, bmin.y, bmax.z),
                   v3(bmax.x, bmax.y, bmin.z), v3(bmin.x, bmax.y, bmax.z), v3(bmax.x, bmin.y, bmax.z), v3(bmax.x, bmax.y, bmax.z)};
        V3 mn = v3(FLT_MAX, FLT_MAX, FLT_MAX), mx = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = 0; i < 8; i++) { V3 w = xpoint(m, c[i]); mn = vmin(mn, w); mx = vmax(mx, w); }
        omin = mn; omax = mx;
    }
    static hrt_affine3x4 invert_rigid_or_uniform(const hrt_affine3x4& m, float& uniformScale)  // Scene.cs:616-638
    {
        float sx = vlen(v3(m.m00, m.m10, m.m20)), sy = vlen(v3(m.m01, m.m11, m.m21)), sz = vlen(v3(m.m02, m.m12, m.m22));
        uniformScale = (sx + sy + sz) / 3.f;
        float inv = uniformScale > 0.f ? 1.f / uniformScale : 1.f;
        V3 r0 = vnorm(v3(m.m00, m.m10, m.m20)), r1 = vnorm(v3(m.m01, m.m11, m.m21)), r2 = vnorm(v3(m.m02, m.m12, m.m22));
        hrt_affine3x4 im; std::memset(&im, 0, sizeof(im));
        im.m00 = r0.x * inv; im.m01 = r1.x * inv; im.m02 = r2.x * inv;
        im.m10 = r0.y * inv; im.m11 = r1.y * inv; im.m12 = r2.y * inv;
        im.m20 = r0.z * inv; im.m21 = r1.z * inv; im.m22 = r2.z * inv;
        V3 it = xvector(im, v3(m.m03, m.m13, m.m23)) * -1.f;
        im.m03 = it.x; im.m13 = it.y; im.m23 = it.z;
        return im;
    }

    int add_sphere(const hrt_sphere& s)
    {
        int id = (int)spheres.size();
        spheres.push_back(s);
        spherePrimIdx.push_back(id);
        return id;
    }
    int add_texture(int w, int h, const hrt_rgba32* px)
    {
        hrt_tex_info ti = {(int32_t)texels.size(), w, h};
        texels.insert(texels.end(), px, px + (size_t)w * h);
        texInfos.push_back(ti);
        return (int)texInfos.size() - 1;
    }

    int build_sphere_instance(const int* sphereIds, int nIds, const hrt_affine3x4& o2w)
    {
        V3 bmin = v3(FLT_MAX, FLT_MAX, FLT_MAX), bmax = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = 0; i < nIds; i++) {
            const hrt_sphere& s = spheres[sphereIds[i]];
            bmin = vmin(bmin, v3(s.center.X - s.radius, s.center.Y - s.radius, s.center.Z - s.radius));
            bmax = vmax(bmax, v3(s.center.X + s.radius, s.center.Y + s.radius, s.center.Z + s.radius));
        }
        int primStart = sphereIds[0], primCount = nIds;
        // local item i <-> spherePrimIdx[primStart + i]   (Scene.cs:383-393)
        std::vector<V3> lmin(primCount), lmax(primCount);
        std::vector<float> key[3];
        for (int a = 0; a < 3; a++) key[a].resize(primCount);
        std::vector<int> val(primCount);
        for (int i = 0; i < primCount; i++) {
            int sid = spherePrimIdx[primStart + i];
            const hrt_sphere& s = spheres[sid];
            lmin[i] = v3(s.center.X - s.radius, s.center.Y - s.radius, s.center.Z - s.radius);
            lmax[i] = v3(s.center.X + s.radius, s.center.Y + s.radius, s.center.Z + s.radius);
            key[0][i] = s.center.X; key[1][i] = s.center.Y; key[2][i] = s.center.Z;
            val[i] = sid;
        }
        int blasStart = (int)blasNodes.size();
        build_blas(primCount, lmin, lmax, key, val, /*byPosition=*/true, spherePrimIdx);
        int blasCount = (int)blasNodes.size() - blasStart;

        V3 wmin, wmax;
        transform_aabb(o2w, bmin, bmax, wmin, wmax);
        float uni;
        hrt_affine3x4 w2o = invert_rigid_or_uniform(o2w, uni);
        hrt_instance inst; std::memset(&inst, 0, sizeof(inst));
        inst.type = HRT_BLAS_SPHERESET; inst.blasRoot = blasStart; inst.blasNodeCount = blasCount;
        inst.primIndexFirst = primStart; inst.primIndexCount = primCount;
        inst.objectToWorld = o2w; inst.worldToObject = w2o; inst.uniformScale = uni;
        inst.worldBoundsMin = f3(wmin); inst.worldBoundsMax = f3(wmax);
        instances.push_back(inst);
        return (int)instances.size() - 1;
    }

    int load_mesh_instance(const hrt_float3* pos, int nPos, const hrt_mesh_tri* tris, int nTris, const hrt_float2* tex, int nTex,
                           const hrt_mesh_tri_uv* tuv, const int* triMat, int nTriMat, const hrt_material* mats, int nMats,
                           const int* texW, const int* texH, const uint8_t* texBGRA, int nTextures, const hrt_affine3x4& o2w)
    {
        int baseVertex = (int)meshPositions.size(), baseTri = (int)meshTris.size(), baseUV = (int)meshTexcoords.size(), baseMat = (int)materials.size();
        meshPositions.insert(meshPositions.end(), pos, pos + nPos);
        meshTexcoords.insert(meshTexcoords.end(), tex, tex + nTex);
        meshTris.reserve(meshTris.size() + nTris); meshTriUVs.reserve(meshTriUVs.size() + nTris);
        for (int i = 0; i < nTris; i++) {
            hrt_mesh_tri t = tris[i]; t.i0 += baseVertex; t.i1 += baseVertex; t.i2 += baseVertex;
            meshTris.push_back(t);
            hrt_mesh_tri_uv u = tuv[i]; u.t0 += baseUV; u.t1 += baseUV; u.t2 += baseUV;
            meshTriUVs.push_back(u);
            triMatIndex.push_back(baseMat + ((triMat && i < nTriMat) ? triMat[i] : 0));
            triPrimIdx.push_back(baseTri + i);
        }
        // texture flattening + material remap (Scene.cs:180-227); BGRA source -> RGBA32 texels
        std::vector<size_t> texOff(nTextures + 1, 0);
        for (int i = 0; i < nTextures; i++) texOff[i + 1] = texOff[i] + (size_t)texW[i] * texH[i] * 4;
        auto append_tex = [&](int ti) {
            hrt_tex_info info = {(int32_t)texels.size(), texW[ti], texH[ti]};
            const uint8_t* b = texBGRA + texOff[ti];
            size_t npx = (size_t)texW[ti] * texH[ti];
            for (size_t p = 0; p < npx; p++) { hrt_rgba32 px = {b[4 * p + 2], b[4 * p + 1], b[4 * p + 0], b[4 * p + 3]}; texels.push_back(px); }
            texInfos.push_back(info);
            return (int)texInfos.size() - 1;
        };
        for (int i = 0; i < nMats; i++) {
            hrt_material m = mats[i];
            if (m.HasDiffuseMap != 0 && m.DiffuseTexIndex >= 0 && m.DiffuseTexIndex < nTextures) { m.DiffuseTexIndex = append_tex(m.DiffuseTexIndex); m.HasDiffuseMap = 1; }
            else { m.HasDiffuseMap = 0; m.DiffuseTexIndex = -1; }
            if (m.HasAlphaMap != 0 && m.AlphaTexIndex >= 0 && m.AlphaTexIndex < nTextures) { m.AlphaTexIndex = append_tex(m.AlphaTexIndex); m.HasAlphaMap = 1; }
            else { m.HasAlphaMap = 0; m.AlphaTexIndex = -1; }
            materials.push_back(m);
        }
        // per-triangle bounds and centroid keys, once (BoundsOfTriangle / CenterOfTriangle, Scene.cs:597-614)
        std::vector<V3> lmin(nTris), lmax(nTris);
        std::vector<float> key[3];
        for (int a = 0; a < 3; a++) key[a].resize(nTris);
        std::vector<int> val(nTris);
        V3 mbmin = v3(FLT_MAX, FLT_MAX, FLT_MAX), mbmax = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        for (int i = 0; i < nTris; i++) {
            // BuildBLAS_Triangles hands the recursion idx[i] = baseTri + i and the recursion reads primIdx[idx[i]] (Scene.cs:398-403,
            // :428): the triangle list is looked up BY POSITION in the prim-index list, which holds this mesh's own triangles at
            // [baseTri, baseTri + n) only for the first mesh of a scene -- every earlier mesh has appended its leaf copies in between.
            // The BLAS of a later mesh is therefore built over whatever triangles that window names (entries below the list's
            // length at this point never change afterwards).  Restated literally; the instance bounds below are the mesh's own.
            const int triIndex = triPrimIdx[(size_t)baseTri + (size_t)i];
            const hrt_mesh_tri& t = meshTris[(size_t)triIndex];
            V3 a = v3(meshPositions[t.i0]), b = v3(meshPositions[t.i1]), c = v3(meshPositions[t.i2]);
            lmin[i] = vmin(a, vmin(b, c)); lmax[i] = vmax(a, vmax(b, c));
            key[0][i] = (a.x + b.x + c.x) / 3.f; key[1][i] = (a.y + b.y + c.y) / 3.f; key[2][i] = (a.z + b.z + c.z) / 3.f;
            val[i] = triIndex;
            const hrt_mesh_tri& own = meshTris[(size_t)baseTri + (size_t)i];                  // ComputeMeshBounds :582-595 (mesh.Positions / mesh.Triangles)
            mbmin = vmin(mbmin, vmin(v3(meshPositions[own.i0]), vmin(v3(meshPositions[own.i1]), v3(meshPositions[own.i2]))));
            mbmax = vmax(mbmax, vmax(v3(meshPositions[own.i0]), vmax(v3(meshPositions[own.i1]), v3(meshPositions[own.i2]))));
        }
        int blasStart = (int)blasNodes.size();
        build_blas(nTris, lmin, lmax, key, val, /*byPosition=*/false, triPrimIdx);
        int blasCount = (int)blasNodes.size() - blasStart;

        V3 wmin, wmax;
        transform_aabb(o2w, mbmin, mbmax, wmin, wmax);
        float uni;
        hrt_affine3x4 w2o = invert_rigid_or_uniform(o2w, uni);
        hrt_instance inst; std::memset(&inst, 0, sizeof(inst));
        inst.type = HRT_BLAS_TRIMESH; inst.blasRoot = blasStart; inst.blasNodeCount = blasCount;
        inst.primIndexFirst = baseTri; inst.primIndexCount = nTris;
        inst.objectToWorld = o2w; inst.worldToObject = w2o; inst.uniformScale = uni;
        inst.worldBoundsMin = f3(wmin); inst.worldBoundsMax = f3(wmax);
        instances.push_back(inst);
        rebuild_tlas();                         // LoadObjInstance ends with RebuildTLAS (:255)
        return (int)instances.size() - 1;
    }

    // a moved instance: the records BuildSphereInstance / LoadObjInstance derive from objectToWorld (Scene.cs:395-402,236-252);
    // object-space bounds = box of the BLAS root (the union of all primitive boxes for both builders)
    bool set_instance_transform(int id, const hrt_affine3x4& o2w)
    {
        if (id < 0 || id >= (int)instances.size()) return false;
        hrt_instance& inst = instances[(size_t)id];
        V3 bmin = v3(0.f, 0.f, 0.f), bmax = bmin;
        if (inst.blasNodeCount > 0) { bmin = v3(blasNodes[(size_t)inst.blasRoot].boundsMin); bmax = v3(blasNodes[(size_t)inst.blasRoot].boundsMax); }
        V3 wmin, wmax;
        transform_aabb(o2w, bmin, bmax, wmin, wmax);
        float uni;
        hrt_affine3x4 w2o = invert_rigid_or_uniform(o2w, uni);
        inst.objectToWorld = o2w; inst.worldToObject = w2o; inst.uniformScale = uni;
        inst.worldBoundsMin = f3(wmin); inst.worldBoundsMax = f3(wmax);
        return true;
    }

    void rebuild_tlas()                         // Scene.cs:358-368,469-510
    {
        int n = (int)instances.size();
        std::vector<V3> bmin(n), bmax(n);
        std::vector<float> key[3];
        for (int a = 0; a < 3; a++) key[a].resize(n);
        for (int i = 0; i < n; i++) {
            bmin[i] = v3(instances[i].worldBoundsMin); bmax[i] = v3(instances[i].worldBoundsMax);
            key[0][i] = 0.5f * (bmin[i].x + bmax[i].x); key[1][i] = 0.5f * (bmin[i].y + bmax[i].y); key[2][i] = 0.5f * (bmin[i].z + bmax[i].z);
        }
        std::map<int, int> sizes;
        warm_sizes(n, 2, sizes);
        int nNodes = subtree_nodes(n, 2, sizes);
        tlasNodes.assign(nNodes, hrt_bvh_node());
        tlasInst.resize(n);
        for (int i = 0; i < n; i++) tlasInst[i] = i;
        BvhJob J;
        J.leafMax = 2; J.boundsByPosition = false; J.bmin = bmin.data(); J.bmax = bmax.data();
        J.key[0] = key[0].data(); J.key[1] = key[1].data(); J.key[2] = key[2].data();
        J.ids = tlasInst.data(); J.nodes = tlasNodes.data(); J.nodeBase = 0;
        J.leafOut = nullptr; J.leafBase = 0; J.itemValue = nullptr; J.sizes = &sizes;
        build_subtree(J, 0, n, 0, 0, -1, 0);
    }

    static hrt_material lambert(float r, float g, float b, int hasMap, int tex)
    {
        hrt_material m; std::memset(&m, 0, sizeof(m));
        m.Kd.X = r; m.Kd.Y = g; m.Kd.Z = b; m.HasDiffuseMap = hasMap; m.DiffuseTexIndex = tex;
        m.Shading = HRT_SHADING_LAMBERT; m.IOR = 1.f; m.AlphaTexIndex = -1; m.AlphaCutoff = 0.5f;
        return m;
    }
    int checker(int w, int h, int step, hrt_rgba32 c0, hrt_rgba32 c1)     // Scene.cs:98-109
    {
        std::vector<hrt_rgba32> px((size_t)w * h);
        for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) px[(size_t)y * w + x] = ((((x / step) + (y / step)) & 1) == 0) ? c0 : c1;
        return add_texture(w, h, px.data());
    }
    void build_default_scene()                  // Scene.cs:83-142 (no Sponza asset in this build)
    {
        clear();
        hrt_rgba32 white = {255, 255, 255, 255}, dark = {20, 20, 20, 255}, blue = {40, 40, 200, 255}, yellow = {200, 200, 40, 255};
        int c0 = checker(256, 256, 16, white, dark), c1 = checker(256, 256, 8, blue, yellow);
        struct D { float cx, cy, cz, r, ar, ag, ab; hrt_material m; int shading; float ior; };
        D d[6] = {
            {0.f, -1000.5f, 0.f, 1000.f, 1.f, 1.f, 1.f, lambert(1.f, 1.f, 1.f, 1, c0), HRT_SHADING_LAMBERT, 1.f},
            {-0.9f, 0.5f, -0.2f, 0.5f, 0.8f, 0.3f, 0.3f, lambert(0.8f, 0.3f, 0.3f, 0, -1), HRT_SHADING_LAMBERT, 1.f},
            {0.9f, 0.35f, 0.2f, 0.35f, 0.3f, 0.8f, 0.3f, lambert(0.3f, 0.8f, 0.3f, 0, -1), HRT_SHADING_LAMBERT, 1.f},
            {0.0f, 0.75f, 0.6f, 0.75f, 1.f, 1.f, 1.f, lambert(1.f, 1.f, 1.f, 1, c1), HRT_SHADING_LAMBERT, 1.f},
            {-1.8f, 0.5f, 0.8f, 0.5f, 1.f, 1.f, 1.f, lambert(1.f, 1.f, 1.f, 0, -1), HRT_SHADING_MIRROR, 1.f},
            {1.8f, 0.5f, -0.8f, 0.5f, 1.f, 1.f, 1.f, lambert(1.f, 1.f, 1.f, 0, -1), HRT_SHADING_GLASS, 1.5f},
        };
        int ids[6];
        for (int i = 0; i < 6; i++) {
            hrt_sphere s; s.center.X = d[i].cx; s.center.Y = d[i].cy; s.center.Z = d[i].cz; s.radius = d[i].r;
            s.albedo.X = d[i].ar; s.albedo.Y = d[i].ag; s.albedo.Z = d[i].ab; s.material = d[i].m; s.shading = d[i].shading; s.ior = d[i].ior;
            ids[i] = add_sphere(s);
        }
        hrt_affine3x4 I; std::memset(&I, 0, sizeof(I)); I.m00 = I.m11 = I.m22 = 1.f;
        for (int i = 0; i < 6; i++) build_sphere_instance(&ids[i], 1, I);
        rebuild_tlas();
    }

    void get_desc(hrt_scene_desc& d) const
    {
        std::memset(&d, 0, sizeof(d));
#define HS(field, vec) d.field = (vec).empty() ? nullptr : (vec).data(); d.n_##field = (int64_t)(vec).size()
        HS(tlasNodes, tlasNodes); HS(tlasInstanceIndices, tlasInst); HS(instances, instances); HS(blasNodes, blasNodes);
        HS(spherePrimIdx, spherePrimIdx); HS(spheres, spheres); HS(triPrimIdx, triPrimIdx); HS(meshPositions, meshPositions);
        HS(meshTris, meshTris); HS(meshTexcoords, meshTexcoords); HS(meshTriUVs, meshTriUVs); HS(triMatIndex, triMatIndex);
        HS(materials, materials); HS(texels, texels); HS(texInfos, texInfos);
#undef HS
    }
};

// ---- camera (Camera.cs) ----------------------------------------------------------------
const float kPiF = 3.14159274f;    // XMath.PI as a float
void update_derived(hrt_camera& c, float aspect, float fovY)
{
    V3 fwd = vnorm((v3(c.lowerLeft) + v3(c.horizontal) * 0.5f + v3(c.vertical) * 0.5f) - v3(c.origin));
    V3 up = vnorm(v3(c.vertical));
    V3 right = vnorm(vcross(fwd, up));
    c.forward = f3(fwd); c.up = f3(up); c.right = f3(right); c.aspect = aspect; c.fovYRadians = fovY;
}

} // namespace

extern "C" {

void* hrth_scene_new(void) { return new HostScene(); }
void hrth_scene_free(void* s) { delete static_cast<HostScene*>(s); }
void hrth_scene_clear(void* s) { static_cast<HostScene*>(s)->clear(); }
void hrth_scene_build_default(void* s) { static_cast<HostScene*>(s)->build_default_scene(); }
int hrth_scene_add_texture(void* s, int w, int h, const hrt_rgba32* px) { if (!s || !px || w <= 0 || h <= 0) return -1; return static_cast<HostScene*>(s)->add_texture(w, h, px); }
int hrth_scene_add_sphere(void* s, const hrt_sphere* sp) { if (!s || !sp) return -1; return static_cast<HostScene*>(s)->add_sphere(*sp); }
int hrth_scene_build_sphere_instance(void* s_, const int* ids, int n, const hrt_affine3x4* m)
{
    HostScene* s = static_cast<HostScene*>(s_);
    if (!s || !ids || n <= 0 || !m) return -1;
    for (int i = 0; i < n; i++) if (ids[i] < 0 || ids[i] >= (int)s->spheres.size()) return -1;
    if (ids[0] + n > (int)s->spherePrimIdx.size()) return -1;
    return s->build_sphere_instance(ids, n, *m);
}
int hrth_scene_load_mesh_instance(void* s_, const hrt_float3* pos, int nPos, const hrt_mesh_tri* tris, int nTris,
                                  const hrt_float2* tex, int nTex, const hrt_mesh_tri_uv* tuv, const int* triMat, int nTriMat,
                                  const hrt_material* mats, int nMats, const int* texW, const int* texH, const uint8_t* texBGRA, int nTextures,
                                  const hrt_affine3x4* m)
{
    HostScene* s = static_cast<HostScene*>(s_);
    if (!s || !pos || !tris || !tuv || !m || nTris <= 0 || nPos <= 0 || nTex < 0 || nMats < 0 || nTextures < 0) return -1;
    if ((nTex > 0 && !tex) || (nMats > 0 && !mats) || (nTextures > 0 && (!texW || !texH || !texBGRA))) return -1;
    // an OBJ without vt / usemtl statements refers to texcoord 0 / material 0 of lists that are empty: the reference
    // uploads one zeroed element for an empty list (Scene.cs:370-377), so index 0 is the only one that stays valid
    const int limTex = nTex > 0 ? nTex : 1, limMat = nMats > 0 ? nMats : 1;
    for (int i = 0; i < nTris; i++) {
        const hrt_mesh_tri& t = tris[i]; const hrt_mesh_tri_uv& u = tuv[i];
        if (t.i0 < 0 || t.i1 < 0 || t.i2 < 0 || t.i0 >= nPos || t.i1 >= nPos || t.i2 >= nPos) return -1;
        if (u.t0 < 0 || u.t1 < 0 || u.t2 < 0 || u.t0 >= limTex || u.t1 >= limTex || u.t2 >= limTex) return -1;
        if (triMat && i < nTriMat && (triMat[i] < 0 || triMat[i] >= limMat)) return -1;
    }
    return s->load_mesh_instance(pos, nPos, tris, nTris, tex, nTex, tuv, triMat, nTriMat, mats, nMats, texW, texH, texBGRA, nTextures, *m);
}
void hrth_scene_rebuild_tlas(void* s) { static_cast<HostScene*>(s)->rebuild_tlas(); }
int hrth_scene_set_instance_transform(void* s, int id, const hrt_affine3x4* m)
{
    if (!s || !m) return -1;
    return static_cast<HostScene*>(s)->set_instance_transform(id, *m) ? 0 : -1;
}
void hrth_scene_get_desc(void* s, hrt_scene_desc* d) { static_cast<HostScene*>(s)->get_desc(*d); }

void hrth_camera_create(int width, int height, float fovDegrees, hrt_camera* out)      // Camera.cs:19-47
{
    float aspect = (float)width / (float)hrt_imax(1, height);
    float theta = fovDegrees * (kPiF / 180.f);
    float halfH = hrt_tan(0.5f * theta), halfW = aspect * halfH;
    V3 origin = v3(0.f, 1.f, 3.f), lookAt = v3(0.f, 0.5f, 0.f), upHint = v3(0.f, 1.f, 0.f);
    V3 w = vnorm(origin - lookAt), u = vnorm(vcross(upHint, w)), v = vcross(w, u);
    hrt_camera c; std::memset(&c, 0, sizeof(c));
    c.origin = f3(origin);
    c.lowerLeft = f3(origin - u * halfW - v * halfH - w);
    c.horizontal = f3(u * (2.f * halfW));
    c.vertical = f3(v * (2.f * halfH));
    update_derived(c, aspect, theta);
    *out = c;
}
void hrth_camera_lookat(const float* o, const float* l, const float* upv, float vfovDegrees, float aspect, float focusDist, hrt_camera* out) // :100-119
{
    V3 origin = v3(o[0], o[1], o[2]), lookAt = v3(l[0], l[1], l[2]), up = v3(upv[0], upv[1], upv[2]);
    float theta = vfovDegrees * (kPiF / 180.f);
    float halfH = hrt_tan(0.5f * theta), halfW = aspect * halfH;
    V3 f = vnorm(vnorm(lookAt - origin));             // OrthoBasis normalises its (already unit) argument again (:195)
    if (hrt_abs(vdot(f, up)) > 0.999f) { up = v3(0.f, 1.f, 0.f); if (hrt_abs(vdot(f, up)) > 0.999f) up = v3(1.f, 0.f, 0.f); }
    V3 u = vnorm(vcross(f, up)), v = vnorm(vcross(u, f));
    V3 forward = vnorm(lookAt - origin);
    hrt_camera c;
    c.origin = f3(origin);
    c.horizontal = f3(u * (2.f * halfW));
    c.vertical = f3(v * (2.f * halfH));
    c.lowerLeft = f3(origin - u * halfW - v * halfH + forward * focusDist);
    V3 fwd = vnorm((v3(c.lowerLeft) + v3(c.horizontal) * 0.5f + v3(c.vertical) * 0.5f) - origin);
    c.forward = f3(fwd); c.right = f3(vnorm(vcross(fwd, v))); c.up = f3(vnorm(v));
    c.aspect = aspect; c.fovYRadians = theta;
    *out = c;
}
void hrth_camera_translate(hrt_camera* c, const float* d)                               // :121-126
{
    V3 dv = v3(d[0], d[1], d[2]);
    c->origin = f3(v3(c->origin) + dv);
    c->lowerLeft = f3(v3(c->lowerLeft) + dv);
    update_derived(*c, c->aspect, c->fovYRadians);
}
void hrth_camera_bake(hrt_camera* c, int pixelW, int pixelH)                            // RTRenderer.cs:241-263
{
    V3 center = v3(c->lowerLeft) + v3(c->horizontal) * 0.5f + v3(c->vertical) * 0.5f;
    V3 fwd = vnorm(center - v3(c->origin)), up = vnorm(v3(c->vertical)), right = vnorm(vcross(fwd, up));
    float focus = vlen(center - v3(c->origin));
    float halfH = 0.5f * vlen(v3(c->vertical));
    float tanHalf = (focus > 1e-6f) ? (halfH / focus) : halfH;
    float fovY = 2.f * hrt_atan(tanHalf);
    float lh = vlen(v3(c->horizontal)), lv = vlen(v3(c->vertical));
    float aspect = (lh > 1e-6f && lv > 1e-6f) ? (lh / lv) : ((float)pixelW / (float)hrt_imax(1, pixelH));
    c->forward = f3(fwd); c->up = f3(up); c->right = f3(right); c->fovYRadians = fovY; c->aspect = aspect;
}
void hrth_sun_dir(float az, float el, float* out)                                       // RTRenderer.cs:174-178
{
    V3 s = vnorm(v3(hrt_cos(az) * hrt_cos(el), hrt_sin(el), hrt_sin(az) * hrt_cos(el)));
    out[0] = s.x; out[1] = s.y; out[2] = s.z;
}

} // extern "C"
