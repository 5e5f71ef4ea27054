// hrt_bvh.hip -- kernels behind hrt_scene_update_instances and hrt_scene_update_positions (see hrt_bvh.hpp for what they
// replace in the reference).
//
// All of it is small integer / min-max work over arrays of a few bytes per instance, triangle or node: one thread per item,
// coalesced where the numbering allows, no LDS tiling.  The only ordering problems are the bottom-up box pass (small
// subtrees straight from their leaves, one arrival counter per inner node above them, release/acquire at agent scope) and
// the sort of the Morton keys (hipCUB radix sort, stable, so equal keys keep item order and the trees are deterministic).
#include "hrt_bvh.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cfloat>
#include <cstdlib>

namespace hrt {
namespace {

constexpr int kBlock = 256;
inline int blocks_for(int n) { return (n + kBlock - 1) / kBlock; }

HRT_D int f2i(float f) { return __float_as_int(f); }
HRT_D float i2f(int i) { return __int_as_float(i); }
HRT_D int node_cnt(const NodeQ* n, int i) { return (int)((unsigned)f2i(n[i].hi.w) >> 28); }
HRT_D int node_skip(const NodeQ* n, int i) { return f2i(n[i].hi.w) & kEnd; }
HRT_D int node_link(const NodeQ* n, int i) { return f2i(n[i].lo.w); }

// Scene.cs:475-493 / host xpoint, xvector: same expression order
HRT_D F3 xf_point(const hrt_affine3x4& m, F3 p)
{
    return mk3(m.m00 * p.x + m.m01 * p.y + m.m02 * p.z + m.m03, m.m10 * p.x + m.m11 * p.y + m.m12 * p.z + m.m13, m.m20 * p.x + m.m21 * p.y + m.m22 * p.z + m.m23);
}
HRT_D F3 xf_vector(const hrt_affine3x4& m, F3 v)
{
    return mk3(m.m00 * v.x + m.m01 * v.y + m.m02 * v.z, m.m10 * v.x + m.m11 * v.y + m.m12 * v.z, m.m20 * v.x + m.m21 * v.y + m.m22 * v.z);
}
// These kernels stand in for HOST code of the reference (the builders of Scene.cs), so their Min / Max are .NET's Math.Min /
// Max -- a NaN operand is returned -- and not the v_min_f32 / v_max_f32 the render kernels use (include/hrt_math.h).
HRT_D F3 min3(F3 a, F3 b) { return mk3(hrt_host_fmin(a.x, b.x), hrt_host_fmin(a.y, b.y), hrt_host_fmin(a.z, b.z)); }
HRT_D F3 max3(F3 a, F3 b) { return mk3(hrt_host_fmax(a.x, b.x), hrt_host_fmax(a.y, b.y), hrt_host_fmax(a.z, b.z)); }
HRT_D F3 host_normalize(F3 v)
{
    const float inv = hrt_rsqrt(hrt_host_fmax(1e-20f, v.x * v.x + v.y * v.y + v.z * v.z));
    return mk3(v.x * inv, v.y * inv, v.z * inv);
}
HRT_D bool is_identity(const hrt_affine3x4& m)
{
    return m.m00 == 1.f && m.m01 == 0.f && m.m02 == 0.f && m.m03 == 0.f && m.m10 == 0.f && m.m11 == 1.f && m.m12 == 0.f && m.m13 == 0.f &&
           m.m20 == 0.f && m.m21 == 0.f && m.m22 == 1.f && m.m23 == 0.f;
}

// ------------------------------------------------------------------ instance records
// One thread per moved instance.  The object-space box is the box of the instance's BLAS root node: both builders of the
// reference make it the union of all primitive boxes, which is what BuildSphereInstance / LoadObjInstance transform
// (Scene.cs:395-399, ComputeMeshBounds :582-595).
__global__ void __launch_bounds__(kBlock) k_set_transforms(TlasDevice T, const int32_t* ids, const hrt_affine3x4* xf, int n)
{
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const int ii = ids[k];
    hrt_instance* in = T.instances + ii;
    const hrt_affine3x4 m = xf ? xf[k] : in->objectToWorld;
    F3 bmin = mk3(0.f, 0.f, 0.f), bmax = bmin;
    if (in->blasNodeCount > 0)
    {
        const hrt_bvh_node* root = T.blasNodes + in->blasRoot;
        bmin = cv3(root->boundsMin); bmax = cv3(root->boundsMax);
    }
    // TransformAABB, Scene.cs:560-580
    const F3 c[8] = {mk3(bmin.x, bmin.y, bmin.z), mk3(bmax.x, bmin.y, bmin.z), mk3(bmin.x, bmax.y, bmin.z), mk3(bmin.x, bmin.y, bmax.z),
                     mk3(bmax.x, bmax.y, bmin.z), mk3(bmin.x, bmax.y, bmax.z), mk3(bmax.x, bmin.y, bmax.z), mk3(bmax.x, bmax.y, bmax.z)};
    F3 mn = mk3(FLT_MAX, FLT_MAX, FLT_MAX), mx = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int i = 0; i < 8; i++) { const F3 w = xf_point(m, c[i]); mn = min3(mn, w); mx = max3(mx, w); }
    // InvertRigidOrUniform, Scene.cs:616-638
    const F3 c0 = mk3(m.m00, m.m10, m.m20), c1 = mk3(m.m01, m.m11, m.m21), c2 = mk3(m.m02, m.m12, m.m22);
    const float sx = hrt_sqrt(c0.x * c0.x + c0.y * c0.y + c0.z * c0.z), sy = hrt_sqrt(c1.x * c1.x + c1.y * c1.y + c1.z * c1.z),
                sz = hrt_sqrt(c2.x * c2.x + c2.y * c2.y + c2.z * c2.z);
    const float uni = (sx + sy + sz) / 3.f;
    const float inv = uni > 0.f ? 1.f / uni : 1.f;
    const F3 r0 = host_normalize(c0), r1 = host_normalize(c1), r2 = host_normalize(c2);
    hrt_affine3x4 im;
    im.m00 = r0.x * inv; im.m01 = r1.x * inv; im.m02 = r2.x * inv; im.m03 = 0.f;
    im.m10 = r0.y * inv; im.m11 = r1.y * inv; im.m12 = r2.y * inv; im.m13 = 0.f;
    im.m20 = r0.z * inv; im.m21 = r1.z * inv; im.m22 = r2.z * inv; im.m23 = 0.f;
    const F3 it = xf_vector(im, mk3(m.m03, m.m13, m.m23)) * -1.f;
    im.m03 = it.x; im.m13 = it.y; im.m23 = it.z;
    in->objectToWorld = m;
    in->worldToObject = im;
    in->uniformScale = uni;
    in->worldBoundsMin = to3(mn);
    in->worldBoundsMax = to3(mx);
}

// ------------------------------------------------------------------ leaf slots (the FInst half of validate_and_pack)
__global__ void __launch_bounds__(kBlock) k_leaf_slots(TlasDevice T)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= T.nTI) return;
    const int ii = T.tlasInst[i];
    const hrt_instance in = T.instances[ii];
    const bool ident = is_identity(in.objectToWorld) && is_identity(in.worldToObject) && in.uniformScale == 1.0f;
    const bool sph = in.type == HRT_BLAS_SPHERESET;
    FInst f;
    bool fast = false;
    if (sph && ident && in.blasNodeCount >= 1)
    {
        const hrt_bvh_node root = T.blasNodes[in.blasRoot];
        const int end = in.blasRoot + in.blasNodeCount;
        if (root.count == 1 && (root.skipIndex == -1 || root.skipIndex >= end))
        {
            const int sid = T.spherePrimIdx[root.first];
            const hrt_sphere* sp = T.spheres + sid;
            f.a = make_float4(root.boundsMin.X, root.boundsMin.Y, root.boundsMin.Z, i2f(FI_FAST_SPHERE | FI_IDENTITY | FI_SPHERESET));
            f.b = make_float4(root.boundsMax.X, root.boundsMax.Y, root.boundsMax.Z, i2f(sid));
            f.c = make_float4(sp->center.X, sp->center.Y, sp->center.Z, sp->radius);
            fast = true;
        }
    }
    if (!fast)
    {
        const float scale = in.uniformScale > 0.f ? in.uniformScale : 1.f;
        f.a = make_float4(0.f, 0.f, 0.f, i2f((ident ? FI_IDENTITY : 0) | (sph ? FI_SPHERESET : 0)));
        f.b = make_float4(0.f, 0.f, 0.f, i2f(ii));
        f.c = make_float4(i2f(in.blasRoot), i2f(in.blasRoot + in.blasNodeCount), scale, 0.f);
        atomicOr(T.flags, 1);
    }
    // A box with a NaN bound or with min > max: the union over it (k_refit, host Min / Max) need not contain, as far as the slab
    // test goes, the other boxes it unites -- the walks that skip inner-node tests must not run over such a tree.
    if (!(in.worldBoundsMin.X <= in.worldBoundsMax.X && in.worldBoundsMin.Y <= in.worldBoundsMax.Y && in.worldBoundsMin.Z <= in.worldBoundsMax.Z))
        atomicOr(T.flags + 1, 1);
    T.finst[i] = f;
}

// ------------------------------------------------------------------ boxes, bottom-up
// One thread per node.  In walk-order numbering the subtree of node i is the index range [i, skip(i)), so a node whose
// subtree has at most T.directMax nodes takes its box straight from the instances of the leaves in that range
// (BuildTLASNodeRecursive's own loop over its items, Scene.cs:472-480) -- no ordering between threads at all for the
// bottom five levels, which hold 31 of every 32 nodes; k_refit_level finishes the rest.  Only a tree that is not numbered
// in walk order (HRT_BUILDER_ORDER) climbs: leaves report to their parent through one arrival counter, the last child to
// arrive unites the children (the chain left, left.skip, ... up to the parent's own skip link) and climbs on; release /
// acquire at agent scope costs an L2 write-back per step on a multi-XCD part (300 us for a 131 k-node tree when every
// node climbed).  min / max are exact: no order can change a bit.
HRT_D int subtree_size(const TlasDevice& T, int i)
{
    const int sk = node_skip(T.tlas, i);
    return (sk == kEnd ? T.nT : sk) - i;
}

__global__ void __launch_bounds__(kBlock) k_refit(TlasDevice T)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= T.nT) return;
    NodeQ* nodes = T.tlas;
    const int cnt = node_cnt(nodes, i);
    const int size = cnt > 0 ? 1 : (T.directMax > 1 ? subtree_size(T, i) : T.nT + 1);
    if (size > T.directMax || size < 1) return;              // waits for its children (size < 1: not a walk-order subtree)
    F3 mn = mk3(FLT_MAX, FLT_MAX, FLT_MAX), mx = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int j = i; j < i + size; j++)
    {
        const int c = node_cnt(nodes, j);
        const int first = node_link(nodes, j);
        for (int k = 0; k < c; k++)
        {
            const hrt_instance* in = T.instances + T.tlasInst[first + k];
            mn = min3(mn, cv3(in->worldBoundsMin)); mx = max3(mx, cv3(in->worldBoundsMax));
        }
    }
    nodes[i].lo.x = mn.x; nodes[i].lo.y = mn.y; nodes[i].lo.z = mn.z;
    nodes[i].hi.x = mx.x; nodes[i].hi.y = mx.y; nodes[i].hi.z = mx.z;
    int cur = i;
    for (;;)
    {
        const int p = T.parent[cur];
        if (p < 0) break;
        if (T.directMax > 1) break;                            // walk-order tree: k_refit_level takes over above the direct subtrees
        const int old = __hip_atomic_fetch_add(T.arrive + p, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 < T.nchild[p]) break;
        const int pskip = node_skip(nodes, p);
        mn = mk3(FLT_MAX, FLT_MAX, FLT_MAX); mx = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        int c = node_link(nodes, p) & kEnd;
        for (int guard = 0; c != kEnd && c != pskip && guard < 64; guard++)
        {
            const float4 lo = nodes[c].lo, hi = nodes[c].hi;
            mn = min3(mn, mk3(lo.x, lo.y, lo.z)); mx = max3(mx, mk3(hi.x, hi.y, hi.z));
            c = f2i(hi.w) & kEnd;
        }
        nodes[p].lo.x = mn.x; nodes[p].lo.y = mn.y; nodes[p].lo.z = mn.z;
        nodes[p].hi.x = mx.x; nodes[p].hi.y = mx.y; nodes[p].hi.z = mx.z;
        cur = p;
    }
}

// Walk-order trees need no arrival counters at all: a node whose subtree has more than `lo` and at most `hi` nodes unites the
// boxes of the maximal subtrees of at most `lo` nodes inside its index range (finished by the previous launch), hopping over
// each of them with its skip link.  With lo = 63, 63 * 64, ... three or four launches reach the root of any tree, every
// thread reads at most a few hundred boxes, and a kernel boundary is the only synchronisation.
__global__ void __launch_bounds__(kBlock) k_refit_level(TlasDevice T, int lo, int hi)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= T.nT) return;
    NodeQ* nodes = T.tlas;
    if (node_cnt(nodes, i) > 0) return;
    const int size = subtree_size(T, i);
    if (size <= lo || size > hi) return;
    F3 mn = mk3(FLT_MAX, FLT_MAX, FLT_MAX), mx = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    const int end = i + size;
    for (int j = i + 1; j < end;)
    {
        const float4 qlo = nodes[j].lo, qhi = nodes[j].hi;
        const int sk = f2i(qhi.w) & kEnd;
        const int e = sk == kEnd ? T.nT : sk;
        if (e - j <= lo) { mn = min3(mn, mk3(qlo.x, qlo.y, qlo.z)); mx = max3(mx, mk3(qhi.x, qhi.y, qhi.z)); j = e; }
        else j++;
    }
    nodes[i].lo.x = mn.x; nodes[i].lo.y = mn.y; nodes[i].lo.z = mn.z;
    nodes[i].hi.x = mx.x; nodes[i].hi.y = mx.y; nodes[i].hi.z = mx.z;
}

// ------------------------------------------------------------------ exclusive scans over the node list
// One 64-bit scan carries both sums: low word = leaf slots before node i (i + that = its position in the TLAS with
// inlined instance records), high word = leaves before node i (its position in the flat leaf list).
__global__ void __launch_bounds__(kBlock) k_scan_input(TlasDevice T)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= T.nT) return;
    const unsigned long long c = (unsigned long long)node_cnt(T.tlas, i);
    T.scanIn[i] = c | ((c > 0 ? 1ull : 0ull) << 32);
}

// ------------------------------------------------------------------ everything that is a function of the packed TLAS
__global__ void __launch_bounds__(kBlock) k_derive(TlasDevice T)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= T.nT) return;
    const NodeQ q = T.tlas[i];
    const int cnt = (int)((unsigned)f2i(q.hi.w) >> 28), skip = f2i(q.hi.w) & kEnd, link = f2i(q.lo.w);
    // reference layout, in walk-order numbering (Scene.cs:705-714; leaf / inner conventions of BuildTLASNodeRecursive)
    hrt_bvh_node* r = T.tlasNodes + i;
    r->boundsMin.X = q.lo.x; r->boundsMin.Y = q.lo.y; r->boundsMin.Z = q.lo.z;
    r->boundsMax.X = q.hi.x; r->boundsMax.Y = q.hi.y; r->boundsMax.Z = q.hi.z;
    r->skipIndex = skip == kEnd ? -1 : skip;
    if (cnt > 0) { r->left = -1; r->right = -1; r->first = link; r->count = cnt; }
    else
    {
        const int l = link & kEnd;
        int rr = -1;
        if (l != kEnd) { const int s = node_skip(T.tlas, l); if (s != kEnd && s != skip) rr = s; }
        r->left = l == kEnd ? -1 : l; r->right = rr; r->first = -1; r->count = 0;
    }
    // surface area of the box (unlinked nodes of an uploaded tree do not count) and the node's term of the SAH estimate
    const float dx = q.hi.x - q.lo.x, dy = q.hi.y - q.lo.y, dz = q.hi.z - q.lo.z;
    const bool linked = i == 0 || T.parent[i] >= 0;
    T.sa[i] = linked ? 2.f * (dx * dy + dy * dz + dz * dx) : 0.f;
    T.arrive[i] = cnt > 0 ? cnt : 1;          // the refit is over: the counters now carry the SAH weights for k_cost
    // the TLAS with instance records inlined after their leaf (hrt_walker.hpp), as validate_and_pack lays it out
    const int at = i + (int)(unsigned)T.scanOut[i];
    const int skX = skip == kEnd ? kEnd : skip + (int)(unsigned)T.scanOut[skip];
    NodeQ o = q;
    o.hi.w = i2f(skX | (int)((unsigned)cnt << 28));
    if (cnt == 0) { const int l = link & kEnd; o.lo.w = i2f(l == kEnd ? kEnd : l + (int)(unsigned)T.scanOut[l]); }
    T.tlasX[at] = o;
    for (int j = 0; j < cnt; j++)
    {
        const FInst f = T.finst[link + j];
        NodeQ rec;
        rec.lo = make_float4(f.a.x, f.a.y, f.a.z, i2f(link + j));
        const int next = (j + 1 < cnt) ? at + 2 + j : skX;
        rec.hi = make_float4(f.b.x, f.b.y, f.b.z, i2f(next | (int)(15u << 28)));
        T.tlasX[at + 1 + j] = rec;
    }
    const int leavesBefore = (int)(T.scanOut[i] >> 32);
    if (cnt > 0 && leavesBefore < T.flatMax) T.flat[leavesBefore] = q;
}

// cost[0]: geometric mean over the nodes of area now / area when the tree was last built (1 = as built; a mean that one
// far-flung instance or one huge instance cannot dominate); cost[1]: the classic estimate sum(area x (leaf ? count : 1)) / area(root).
// Two stages with a fixed summation order: reproducible, so HRT_REBUILD_AUTO takes the same decision for the same moves.
__global__ void __launch_bounds__(kBlock) k_cost_partial(TlasDevice T)
{
    __shared__ float s[3][kBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float lg = 0.f, m = 0.f, sah = 0.f;
    if (i < T.nT)
    {
        const float a = T.sa[i], b = T.saBase[i];
        sah = a * (float)T.arrive[i];
        if (a > 0.f && b > 0.f) { lg = __logf(a / b); m = 1.f; }
    }
    s[0][threadIdx.x] = lg; s[1][threadIdx.x] = m; s[2][threadIdx.x] = sah;
    __syncthreads();
    for (int d = kBlock / 2; d > 0; d >>= 1)
    {
        if ((int)threadIdx.x < d) for (int k = 0; k < 3; k++) s[k][threadIdx.x] += s[k][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x < 3) T.costPartial[3 * blockIdx.x + threadIdx.x] = s[threadIdx.x][0];
}

__global__ void __launch_bounds__(1024) k_cost(TlasDevice T, int nPartials)
{
    __shared__ float s[3][1024];
    float acc[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nPartials; i += 1024) for (int k = 0; k < 3; k++) acc[k] += T.costPartial[3 * i + k];
    for (int k = 0; k < 3; k++) s[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1)
    {
        if ((int)threadIdx.x < d) for (int k = 0; k < 3; k++) s[k][threadIdx.x] += s[k][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        T.cost[0] = s[1][0] > 0.f ? __expf(s[0][0] / s[1][0]) : 1.f;
        const float root = T.sa[0];
        T.cost[1] = root > 0.f ? s[2][0] / root : 0.f;
    }
}

// ------------------------------------------------------------------ LBVH topology
HRT_D F3 inst_centroid(const hrt_instance* in)
{   // the builder's sort key (Scene.cs:487-489): 0.5 * (min + max)
    return mk3(0.5f * (in->worldBoundsMin.X + in->worldBoundsMax.X), 0.5f * (in->worldBoundsMin.Y + in->worldBoundsMax.Y),
               0.5f * (in->worldBoundsMin.Z + in->worldBoundsMax.Z));
}

// what the LBVH is built over: the instances of the scene (TLAS) or the triangles of one mesh instance (BLAS)
struct ItemSrc {
    const hrt_instance* instances;      // TLAS: item i = instance i
    const int32_t* triPrimIdx; const hrt_mesh_tri* tris; const hrt_float3* pos; int itemFirst;   // BLAS: item i = triangle triPrimIdx[itemFirst + i]
    int n;
};
HRT_D int item_value(const ItemSrc& S, int i) { return S.instances ? i : S.triPrimIdx[S.itemFirst + i]; }
HRT_D F3 item_centroid(const ItemSrc& S, int i)
{
    if (S.instances) return inst_centroid(S.instances + i);
    const hrt_mesh_tri t = S.tris[S.triPrimIdx[S.itemFirst + i]];         // CenterOfTriangle, Scene.cs:607-614
    const hrt_float3 a = S.pos[t.i0], b = S.pos[t.i1], c = S.pos[t.i2];
    return mk3((a.X + b.X + c.X) / 3.f, (a.Y + b.Y + c.Y) / 3.f, (a.Z + b.Z + c.Z) / 3.f);
}

// floats as unsigned keys of the same order, so the bounds of all blocks meet in six atomicMin / atomicMax words
HRT_D unsigned ord_key(float f) { const unsigned u = (unsigned)f2i(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
HRT_D float ord_float(unsigned k) { return i2f((int)((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k)); }

__global__ void __launch_bounds__(kBlock) k_centroid_bounds(TlasDevice T, ItemSrc S)
{
    __shared__ float s[6][kBlock / 64];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    F3 mn = mk3(FLT_MAX, FLT_MAX, FLT_MAX), mx = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    if (i < S.n) { const F3 c = item_centroid(S, i); mn = c; mx = c; }
    float v[6] = {mn.x, mn.y, mn.z, mx.x, mx.y, mx.z};
    for (int d = 32; d > 0; d >>= 1)
        for (int k = 0; k < 6; k++) { const float o = __shfl_xor(v[k], d); v[k] = k < 3 ? hrt_fmin(v[k], o) : hrt_fmax(v[k], o); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) for (int k = 0; k < 6; k++) s[k][wv] = v[k];
    __syncthreads();
    if (threadIdx.x < 6)
    {
        const int k = threadIdx.x;
        float r = s[k][0];
        for (int w = 1; w < kBlock / 64; w++) r = k < 3 ? hrt_fmin(r, s[k][w]) : hrt_fmax(r, s[k][w]);
        if (k < 3) atomicMin(T.cboundsKey + k, ord_key(r)); else atomicMax(T.cboundsKey + k, ord_key(r));
    }
}

HRT_D unsigned spread3(unsigned v)
{   // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
HRT_D unsigned quant10(float c, float lo, float hi)
{
    const float ext = hi - lo;
    const float n = ext > 0.f ? (c - lo) / ext : 0.f;
    if (!(n > 0.f)) return 0u;
    const float s = n * 1024.f;
    return s >= 1023.f ? 1023u : (unsigned)(int)s;
}

__global__ void __launch_bounds__(kBlock) k_morton(TlasDevice T, ItemSrc S)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= S.n) return;
    const F3 c = item_centroid(S, i);
    float cb[6];
    for (int k = 0; k < 6; k++) cb[k] = ord_float(T.cboundsKey[k]);
    // cubic cells: every axis is cut with the pitch of the longest one, so a flat scene (a terrain, a city) is not sliced
    // along its thin axis at the top of the tree
    const float ext = hrt_fmax(cb[3] - cb[0], hrt_fmax(cb[4] - cb[1], cb[5] - cb[2]));
    const unsigned x = quant10(c.x, cb[0], cb[0] + ext), y = quant10(c.y, cb[1], cb[1] + ext), z = quant10(c.z, cb[2], cb[2] + ext);
    T.keys[i] = (spread3(x) << 2) | (spread3(y) << 1) | spread3(z);
    T.vals[i] = item_value(S, i);
}

// leaf k = sorted slots [stride k, stride k + stride); its key is the key of its first slot; equal keys are told apart by k
HRT_D int lbvh_delta(const unsigned* keys, int L, int stride, int a, int b)
{
    if (b < 0 || b >= L) return -1;
    const unsigned ka = keys[stride * a], kb = keys[stride * b];
    return ka != kb ? __clz((int)(ka ^ kb)) : 32 + __clz(a ^ b);
}

// one thread per inner node j of the L - 1 (Karras 2012): range of leaves, split, children
__global__ void __launch_bounds__(kBlock) k_lbvh_inner(TlasDevice T, int L, int stride)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= L - 1) return;
    const unsigned* keys = T.keysSorted;
    const int d = lbvh_delta(keys, L, stride, j, j + 1) - lbvh_delta(keys, L, stride, j, j - 1) >= 0 ? 1 : -1;
    const int dmin = lbvh_delta(keys, L, stride, j, j - d);
    int lmax = 2;
    while (lbvh_delta(keys, L, stride, j, j + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (lbvh_delta(keys, L, stride, j, j + (l + t) * d) > dmin) l += t;
    const int e = j + l * d;
    const int dnode = lbvh_delta(keys, L, stride, j, e);
    int s = 0, t = l;
    do { t = (t + 1) >> 1; if (lbvh_delta(keys, L, stride, j, j + (s + t) * d) > dnode) s += t; } while (t > 1);
    const int g = j + s * d + min(d, 0);
    const int a = min(j, e), b = max(j, e);
    T.rngA[j] = a; T.rngB[j] = b; T.split[j] = g;
    if (a == g) T.parLeaf[g] = j; else T.parInt[g] = j;
    if (b == g + 1) T.parLeaf[g + 1] = j; else T.parInt[g + 1] = j;
    if (j == 0) T.parInt[0] = -1;
}

__global__ void k_single_leaf(TlasDevice T)
{   // <= 2 instances: the root is the only node
    T.tlas[0].lo.w = i2f(0);
    T.tlas[0].hi.w = i2f(kEnd | (int)((unsigned)T.nI << 28));
    T.parent[0] = -1; T.nchild[0] = 0;
    for (int i = 0; i < T.nI; i++) T.tlasInst[i] = i;
}

// ------------------------------------------------------------------ triangle-mesh BLASes after a vertex update
__global__ void __launch_bounds__(kBlock) k_copy_positions(hrt_float3* dst, const hrt_float3* src, int n)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// one thread per leaf slot: the three vertices of the slot's triangle (index, material and flags in the .w lanes stay)
__global__ void __launch_bounds__(kBlock) k_tri_records(BlasDevice B)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= B.nSlots) return;
    FTri* r = B.ftri + j;
    const hrt_mesh_tri t = B.meshTris[f2i(r->v0.w)];
    const hrt_float3 a = B.positions[t.i0], b = B.positions[t.i1], c = B.positions[t.i2];
    r->v0.x = a.X; r->v0.y = a.Y; r->v0.z = a.Z;
    r->v1.x = b.X; r->v1.y = b.Y; r->v1.z = b.Z;
    r->v2.x = c.X; r->v2.y = c.Y; r->v2.z = c.Z;
}

// Same scheme as k_refit: subtrees of up to directMax nodes straight from their triangles (BoundsOfTriangle over the
// node's items, Scene.cs:423-429,597-605), arrival counters above.
__global__ void __launch_bounds__(kBlock) k_blas_refit(BlasDevice B, int kind)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= B.nB || B.kind[i] != kind) return;
    NodeQ* nodes = B.blas;
    const int size = B.subend[i] - i;
    if (size > B.directMax || size < 1) return;
    F3 mn = mk3(FLT_MAX, FLT_MAX, FLT_MAX), mx = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int j = i; j < i + size; j++)
    {
        const int c = node_cnt(nodes, j);
        const int first = node_link(nodes, j);
        for (int k = 0; k < c; k++)
        {
            if (kind == 1)
            {
                const FTri t = B.ftri[first + k];
                const F3 a = mk3(t.v0.x, t.v0.y, t.v0.z), b = mk3(t.v1.x, t.v1.y, t.v1.z), cc = mk3(t.v2.x, t.v2.y, t.v2.z);
                mn = min3(mn, min3(a, min3(b, cc))); mx = max3(mx, max3(a, max3(b, cc)));
            }
            else
            {   // centre -+ radius per sphere of the leaf (BuildSphereInstance / BuildBLAS_Spheres, Scene.cs:331-336,386-390)
                const hrt_sphere* sp = B.spheres + B.spherePrimIdx[first + k];
                const float r = sp->radius;
                mn = min3(mn, mk3(sp->center.X - r, sp->center.Y - r, sp->center.Z - r));
                mx = max3(mx, mk3(sp->center.X + r, sp->center.Y + r, sp->center.Z + r));
            }
        }
    }
    nodes[i].lo.x = mn.x; nodes[i].lo.y = mn.y; nodes[i].lo.z = mn.z;
    nodes[i].hi.x = mx.x; nodes[i].hi.y = mx.y; nodes[i].hi.z = mx.z;
    int cur = i;
    for (;;)
    {
        const int p = B.parent[cur];
        if (p < 0) break;
        if (B.directMax > 1) break;                            // k_blas_refit_level takes over above the direct subtrees
        const int old = __hip_atomic_fetch_add(B.arrive + p, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 < B.nchild[p]) break;
        const int pskip = node_skip(nodes, p);
        mn = mk3(FLT_MAX, FLT_MAX, FLT_MAX); mx = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
        int c = node_link(nodes, p) & kEnd;
        for (int guard = 0; c != kEnd && c != pskip && guard < 64; guard++)
        {
            const float4 lo = nodes[c].lo, hi = nodes[c].hi;
            mn = min3(mn, mk3(lo.x, lo.y, lo.z)); mx = max3(mx, mk3(hi.x, hi.y, hi.z));
            c = f2i(hi.w) & kEnd;
        }
        nodes[p].lo.x = mn.x; nodes[p].lo.y = mn.y; nodes[p].lo.z = mn.z;
        nodes[p].hi.x = mx.x; nodes[p].hi.y = mx.y; nodes[p].hi.z = mx.z;
        cur = p;
    }
}

__global__ void __launch_bounds__(kBlock) k_blas_refit_level(BlasDevice B, int kind, int lo, int hi)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= B.nB || B.kind[i] != kind) return;
    NodeQ* nodes = B.blas;
    const int end = B.subend[i], size = end - i;
    if (size <= lo || size > hi) return;
    F3 mn = mk3(FLT_MAX, FLT_MAX, FLT_MAX), mx = mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int j = i + 1; j < end;)
    {
        const int e = B.subend[j];
        if (e - j <= lo)
        {
            const float4 qlo = nodes[j].lo, qhi = nodes[j].hi;
            mn = min3(mn, mk3(qlo.x, qlo.y, qlo.z)); mx = max3(mx, mk3(qhi.x, qhi.y, qhi.z));
            j = e;
        }
        else j++;
    }
    nodes[i].lo.x = mn.x; nodes[i].lo.y = mn.y; nodes[i].lo.z = mn.z;
    nodes[i].hi.x = mx.x; nodes[i].hi.y = mx.y; nodes[i].hi.z = mx.z;
}

__global__ void __launch_bounds__(kBlock) k_blas_derive(BlasDevice B, int kind)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= B.nB || B.kind[i] != kind) return;
    const NodeQ q = B.blas[i];
    hrt_bvh_node* r = B.blasNodes + B.orig[i];
    r->boundsMin.X = q.lo.x; r->boundsMin.Y = q.lo.y; r->boundsMin.Z = q.lo.z;
    r->boundsMax.X = q.hi.x; r->boundsMax.Y = q.hi.y; r->boundsMax.Z = q.hi.z;
    const float dx = q.hi.x - q.lo.x, dy = q.hi.y - q.lo.y, dz = q.hi.z - q.lo.z;
    B.sa[i] = 2.f * (dx * dy + dy * dz + dz * dx);
    // links, in the numbering the reference-layout array uses (the uploaded one, or walk order after a rebuild)
    const int cnt = (int)((unsigned)f2i(q.hi.w) >> 28), skip = f2i(q.hi.w) & kEnd, link = f2i(q.lo.w);
    r->skipIndex = skip == kEnd ? -1 : B.orig[skip];
    if (cnt > 0) { r->left = -1; r->right = -1; r->first = link; r->count = cnt; }
    else
    {
        const int l = link & kEnd;
        int rr = -1;
        if (l != kEnd) { const int sk = node_skip(B.blas, l); if (sk != kEnd && sk != skip) rr = B.orig[sk]; }
        r->left = l == kEnd ? -1 : B.orig[l]; r->right = rr; r->first = -1; r->count = 0;
    }
}

// growth of the node boxes of the mesh BLASes since their last build (same measure, same fixed summation order as k_cost)
__global__ void __launch_bounds__(kBlock) k_blas_growth_partial(BlasDevice B, int kind)
{
    __shared__ float s[2][kBlock];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    float lg = 0.f, m = 0.f;
    if (i < B.nB && B.kind[i] == kind)
    {
        const float a = B.sa[i], b = B.saBase[i];
        if (a > 0.f && b > 0.f) { lg = __logf(a / b); m = 1.f; }
    }
    s[0][threadIdx.x] = lg; s[1][threadIdx.x] = m;
    __syncthreads();
    for (int d = kBlock / 2; d > 0; d >>= 1)
    {
        if ((int)threadIdx.x < d) { s[0][threadIdx.x] += s[0][threadIdx.x + d]; s[1][threadIdx.x] += s[1][threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x < 2) B.growPartial[2 * blockIdx.x + threadIdx.x] = s[threadIdx.x][0];
}

__global__ void __launch_bounds__(1024) k_blas_growth(BlasDevice B, int nPartials)
{
    __shared__ float s[2][1024];
    float a0 = 0.f, a1 = 0.f;
    for (int i = threadIdx.x; i < nPartials; i += 1024) { a0 += B.growPartial[2 * i]; a1 += B.growPartial[2 * i + 1]; }
    s[0][threadIdx.x] = a0; s[1][threadIdx.x] = a1;
    __syncthreads();
    for (int d = 512; d > 0; d >>= 1)
    {
        if ((int)threadIdx.x < d) { s[0][threadIdx.x] += s[0][threadIdx.x + d]; s[1][threadIdx.x] += s[1][threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) B.grow[0] = s[1][0] > 0.f ? __expf(s[0][0] / s[1][0]) : 1.f;
}

// ------------------------------------------------------------------ BLAS rebuild of one mesh
// Karras' tree over the single triangles (n leaves, n - 1 inner nodes); a node is EMITTED as a leaf when its subtree holds at
// most `limit` triangles and its parent's more, as an inner node when it holds more.  Leaves therefore never straddle a
// split of the hierarchy (groups of four consecutive triangles in Morton order did, and the few that straddled a top-level
// split had boxes across the whole mesh).  Emitted leaves partition the sorted triangles into runs, so with
// S[i] = leaves starting before triangle i the emitted subtree of a node over [a, b] has 2 (S[b + 1] - S[a]) - 1 nodes:
// walk-order indices and skip links again come from the root path alone.
HRT_D int knode_size(const TlasDevice& T, int n, int v) { return v < n - 1 ? T.rngB[v] - T.rngA[v] + 1 : 1; }     // v >= n - 1: single triangle v - (n - 1)
HRT_D int knode_parent(const TlasDevice& T, int n, int v) { return v < n - 1 ? T.parInt[v] : T.parLeaf[v - (n - 1)]; }
HRT_D int knode_first(const TlasDevice& T, int n, int v) { return v < n - 1 ? T.rngA[v] : v - (n - 1); }

// leafCounts[t] = leaves of the tree under leaf-size limit t, t = 4..14
__global__ void __launch_bounds__(kBlock) k_leaf_counts(TlasDevice T, int n)
{
    __shared__ int s[16];
    if (threadIdx.x < 16) s[threadIdx.x] = 0;
    __syncthreads();
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v < 2 * n - 1)
    {
        const int size = knode_size(T, n, v);
        const int p = n == 1 ? -1 : knode_parent(T, n, v);
        const int psize = p < 0 ? 0x7FFFFFFF : knode_size(T, n, p);
        for (int t = max(size, 4); t <= 14 && t < psize; t++) atomicAdd(&s[t], 1);
    }
    __syncthreads();
    if (threadIdx.x >= 4 && threadIdx.x <= 14 && s[threadIdx.x]) atomicAdd(T.leafCounts + threadIdx.x, s[threadIdx.x]);
}

__global__ void __launch_bounds__(kBlock) k_mark_leaves(TlasDevice T, int n, int limit)
{
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= 2 * n - 1) return;
    const int size = knode_size(T, n, v);
    const int p = n == 1 ? -1 : knode_parent(T, n, v);
    if (size <= limit && (p < 0 || knode_size(T, n, p) > limit)) T.lstart[knode_first(T, n, v)] = 1;
}

__global__ void __launch_bounds__(kBlock) k_blas_emit(TlasDevice T, BlasDevice B, MeshJob J, int limit)
{
    const int n = J.n;
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= 2 * n - 1) return;
    const int size = knode_size(T, n, v);
    int p = n == 1 ? -1 : knode_parent(T, n, v);
    const bool leaf = size <= limit;
    if (leaf && p >= 0 && knode_size(T, n, p) <= limit) return;          // inside a collapsed subtree
    const int a = knode_first(T, n, v), b = a + size - 1;
    const int* S = T.lsum;
    // root path: +1 per left edge, +1 + (emitted nodes of the left sibling) per right edge
    int idx = 0, start = a, parentDelta = 0;
    for (int q = p; q >= 0; q = T.parInt[q])
    {
        const int g = T.split[q], qa = T.rngA[q];
        const int d = start == g + 1 ? 2 * (S[g + 1] - S[qa]) : 1;
        if (q == p) parentDelta = d;
        idx += d;
        start = qa;
    }
    const int total = 2 * S[n] - 1;
    const int emitted = 2 * (S[b + 1] - S[a]) - 1;
    const int g = J.root + idx;
    B.parent[g] = p < 0 ? -1 : g - parentDelta;
    B.subend[g] = g + emitted;
    B.orig[g] = g;
    B.kind[g] = 1;
    const int skip = idx + emitted >= total ? kEnd : g + emitted;
    NodeQ* q = B.blas + g;
    if (leaf)
    {
        q->lo.w = i2f(J.leafBase + a);
        q->hi.w = i2f(skip | (int)((unsigned)size << 28));
        B.nchild[g] = 0;
    }
    else
    {
        q->lo.w = i2f(g + 1);
        q->hi.w = i2f(skip);
        B.nchild[g] = 2;
    }
    if (p < 0) T.instances[J.inst].blasNodeCount = total;
}

// the rest of the range the host's tree needed: owned by nobody
__global__ void __launch_bounds__(kBlock) k_blas_tail(TlasDevice T, BlasDevice B, MeshJob J)
{
    const int total = 2 * T.lsum[J.n] - 1;
    const int v = total + blockIdx.x * kBlock + threadIdx.x;
    if (v >= J.nodeCap) return;
    const int g = J.root + v;
    B.parent[g] = -2; B.nchild[g] = 0; B.subend[g] = g + 1; B.orig[g] = g; B.kind[g] = 0;
    NodeQ z; z.lo = make_float4(0.f, 0.f, 0.f, i2f(kEnd)); z.hi = make_float4(0.f, 0.f, 0.f, i2f(kEnd));
    B.blas[g] = z;
    hrt_bvh_node* r = B.blasNodes + g;
    r->boundsMin.X = r->boundsMin.Y = r->boundsMin.Z = 0.f; r->boundsMax.X = r->boundsMax.Y = r->boundsMax.Z = 0.f;
    r->left = -1; r->right = -1; r->first = -1; r->count = 0; r->skipIndex = -1;
}

// the same emission for the TLAS (items = instances, leaf-size limit 2 like the reference's TLAS, Scene.cs:469-510)
__global__ void __launch_bounds__(kBlock) k_tlas_emit(TlasDevice T, int limit)
{
    const int n = T.nI;
    const int v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= 2 * n - 1) return;
    const int size = knode_size(T, n, v);
    const int p = knode_parent(T, n, v);
    const bool leaf = size <= limit;
    if (leaf && p >= 0 && knode_size(T, n, p) <= limit) return;
    const int a = knode_first(T, n, v), b = a + size - 1;
    const int* S = T.lsum;
    int idx = 0, start = a, parentDelta = 0;
    for (int q = p; q >= 0; q = T.parInt[q])
    {
        const int g = T.split[q], qa = T.rngA[q];
        const int d = start == g + 1 ? 2 * (S[g + 1] - S[qa]) : 1;
        if (q == p) parentDelta = d;
        idx += d;
        start = qa;
    }
    const int total = 2 * S[n] - 1;
    const int emitted = 2 * (S[b + 1] - S[a]) - 1;
    T.parent[idx] = p < 0 ? -1 : idx - parentDelta;
    const int skip = idx + emitted >= total ? kEnd : idx + emitted;
    NodeQ* q = T.tlas + idx;
    if (leaf)
    {
        q->lo.w = i2f(a);
        q->hi.w = i2f(skip | (int)((unsigned)size << 28));
        T.nchild[idx] = 0;
    }
    else
    {
        q->lo.w = i2f(idx + 1);
        q->hi.w = i2f(skip);
        T.nchild[idx] = 2;
    }
}

// every lane of the FTri records of a rebuilt leaf region (the FTri half of validate_and_pack)
__global__ void __launch_bounds__(kBlock) k_tri_records_full(BlasDevice B, int first, int n)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const int ti = B.triPrimIdx[first + j];
    const hrt_mesh_tri t = B.meshTris[ti];
    const hrt_float3 a = B.positions[t.i0], b = B.positions[t.i1], c = B.positions[t.i2];
    const int mi = B.triMatIndex[ti];
    hrt_material m;
    if (B.nMaterials > 0) m = B.materials[mi];
    else { m.Kd.X = m.Kd.Y = m.Kd.Z = 0.f; m.HasDiffuseMap = m.DiffuseTexIndex = m.Shading = 0; m.IOR = 0.f; m.HasAlphaMap = m.AlphaTexIndex = m.TwoSided = 0; m.AlphaCutoff = 0.f; }
    const bool dmap = m.HasDiffuseMap != 0 && m.DiffuseTexIndex >= 0 && m.DiffuseTexIndex < B.texLen;
    const bool amap = m.HasAlphaMap != 0 && m.AlphaTexIndex >= 0 && m.AlphaTexIndex < B.texLen;
    const bool rejects_opaque = 1.0f < m.AlphaCutoff;
    const int fl = ((dmap || amap || rejects_opaque) ? FT_TEXTURED : 0) | (m.TwoSided != 0 ? FT_TWOSIDED : 0);
    FTri o;
    o.v0 = make_float4(a.X, a.Y, a.Z, i2f(ti));
    o.v1 = make_float4(b.X, b.Y, b.Z, i2f(mi));
    o.v2 = make_float4(c.X, c.Y, c.Z, i2f(fl));
    B.ftri[first + j] = o;
}

} // namespace

hipError_t blas_set_positions(const BlasDevice& B, int first, int n, const hrt_float3* posDev, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    k_copy_positions<<<blocks_for(n), kBlock, 0, s>>>(B.positions + first, posDev, n);
    return hipGetLastError();
}

size_t tlas_iscan_temp_bytes(int n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int*)nullptr, (int*)nullptr, n, (hipStream_t) nullptr);
    return bytes;
}

hipError_t blas_rebuild_mesh(const TlasDevice& T, const BlasDevice& B, const MeshJob& J, hipStream_t s, int* leafLimitOut)
{
    const int n = J.n;
    if (n <= 0) return hipErrorInvalidValue;
    ItemSrc S{};
    S.triPrimIdx = B.triPrimIdx; S.tris = B.meshTris; S.pos = B.positions; S.itemFirst = J.itemFirst; S.n = n;
    hipError_t e;
    if ((e = hipMemsetAsync(T.cboundsKey, 0xFF, 3 * sizeof(unsigned), s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(T.cboundsKey + 3, 0, 3 * sizeof(unsigned), s)) != hipSuccess) return e;
    k_centroid_bounds<<<blocks_for(n), kBlock, 0, s>>>(T, S);
    k_morton<<<blocks_for(n), kBlock, 0, s>>>(T, S);
    size_t bytes = T.sortTmpBytes;
    e = hipcub::DeviceRadixSort::SortPairs(T.sortTmp, bytes, (const unsigned*)T.keys, T.keysSorted, (const int*)T.vals, (int*)(B.triPrimIdxW + J.leafBase), n, 0, 30, s);
    if (e != hipSuccess) return e;
    if (n > 1) k_lbvh_inner<<<blocks_for(n - 1), kBlock, 0, s>>>(T, n, 1);
    // the smallest leaf-size limit whose tree fits the node range of the mesh
    int counts[16];
    if ((e = hipMemsetAsync(T.leafCounts, 0, 16 * sizeof(int), s)) != hipSuccess) return e;
    k_leaf_counts<<<blocks_for(2 * n - 1), kBlock, 0, s>>>(T, n);
    if ((e = hipMemcpyAsync(counts, T.leafCounts, sizeof(counts), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    int limit = 4;                                   // the reference's BLAS leaf size; HRT_BLAS_LEAF_LIMIT = 4..14 for experiments
    if (const char* e2 = HRT_ENV("HRT_BLAS_LEAF_LIMIT")) { const int v = atoi(e2); if (v >= 4 && v <= 14) limit = v; }
    while (limit < 14 && 2 * counts[limit] - 1 > J.nodeCap) limit++;
    if (2 * counts[limit] - 1 > J.nodeCap) return hipErrorInvalidValue;
    if (leafLimitOut) *leafLimitOut = limit;
    if ((e = hipMemsetAsync(T.lstart, 0, (size_t)(n + 1) * sizeof(int), s)) != hipSuccess) return e;
    k_mark_leaves<<<blocks_for(2 * n - 1), kBlock, 0, s>>>(T, n, limit);
    bytes = T.iscanTmpBytes;
    if ((e = hipcub::DeviceScan::ExclusiveSum(T.iscanTmp, bytes, (const int*)T.lstart, T.lsum, n + 1, s)) != hipSuccess) return e;
    k_blas_emit<<<blocks_for(2 * n - 1), kBlock, 0, s>>>(T, B, J, limit);
    k_blas_tail<<<blocks_for(J.nodeCap), kBlock, 0, s>>>(T, B, J);
    k_tri_records_full<<<blocks_for(n), kBlock, 0, s>>>(B, J.leafBase, n);
    return hipGetLastError();
}

hipError_t blas_refit(const BlasDevice& B, int kind, hipStream_t s)
{
    hipError_t e;
    if (B.directMax <= 1 && (e = hipMemsetAsync(B.arrive, 0, (size_t)B.nB * sizeof(int), s)) != hipSuccess) return e;   // only the climb counts arrivals
    if (kind == 1 && B.nSlots > 0) k_tri_records<<<blocks_for(B.nSlots), kBlock, 0, s>>>(B);
    k_blas_refit<<<blocks_for(B.nB), kBlock, 0, s>>>(B, kind);
    if (B.directMax > 1)
        for (long long lo = B.directMax; lo < B.maxRange[kind]; lo *= 64)
            k_blas_refit_level<<<blocks_for(B.nB), kBlock, 0, s>>>(B, kind, (int)lo, (int)std::min<long long>(lo * 64, 0x7FFFFFFF));
    k_blas_derive<<<blocks_for(B.nB), kBlock, 0, s>>>(B, kind);
    return hipGetLastError();
}

hipError_t blas_growth(const BlasDevice& B, int kind, hipStream_t s)
{
    k_blas_growth_partial<<<blocks_for(B.nB), kBlock, 0, s>>>(B, kind);
    k_blas_growth<<<1, 1024, 0, s>>>(B, blocks_for(B.nB));
    return hipGetLastError();
}

hipError_t tlas_rebound_instances(const TlasDevice& T, const int32_t* idsDev, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    k_set_transforms<<<blocks_for(n), kBlock, 0, s>>>(T, idsDev, nullptr, n);
    return hipGetLastError();
}

size_t tlas_scan_temp_bytes(int n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, n, (hipStream_t) nullptr);
    return bytes;
}

size_t tlas_sort_temp_bytes(int n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, n, 0, 30, (hipStream_t) nullptr);
    return bytes;
}

hipError_t tlas_set_transforms(const TlasDevice& T, const int32_t* idsDev, const hrt_affine3x4* xfDev, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    k_set_transforms<<<blocks_for(n), kBlock, 0, s>>>(T, idsDev, xfDev, n);
    return hipGetLastError();
}

hipError_t tlas_rebuild_topology(TlasDevice& T, hipStream_t s, int* leavesOut)
{
    const int n = T.nI;
    if (n <= 0) return hipErrorInvalidValue;
    T.nTI = n;
    if (n > T.capTI) return hipErrorInvalidValue;
    if (n <= 2)
    {
        T.nT = 1;
        if (leavesOut) *leavesOut = 1;
        k_single_leaf<<<1, 1, 0, s>>>(T);
        return hipGetLastError();
    }
    hipError_t e;
    if ((e = hipMemsetAsync(T.cboundsKey, 0xFF, 3 * sizeof(unsigned), s)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(T.cboundsKey + 3, 0, 3 * sizeof(unsigned), s)) != hipSuccess) return e;
    ItemSrc S{};
    S.instances = T.instances; S.n = n;
    k_centroid_bounds<<<blocks_for(n), kBlock, 0, s>>>(T, S);
    k_morton<<<blocks_for(n), kBlock, 0, s>>>(T, S);
    size_t bytes = T.sortTmpBytes;
    e = hipcub::DeviceRadixSort::SortPairs(T.sortTmp, bytes, (const unsigned*)T.keys, T.keysSorted, (const int*)T.vals, (int*)T.tlasInst, n, 0, 30, s);
    if (e != hipSuccess) return e;
    // Karras over the single instances; subtrees of <= 2 instances become the leaves (see the BLAS rebuild)
    k_lbvh_inner<<<blocks_for(n - 1), kBlock, 0, s>>>(T, n, 1);
    if ((e = hipMemsetAsync(T.lstart, 0, (size_t)(n + 1) * sizeof(int), s)) != hipSuccess) return e;
    int limit = 2;                                   // the reference's TLAS leaf size; HRT_TLAS_LEAF_LIMIT = 1..14 for experiments
    if (const char* e2 = HRT_ENV("HRT_TLAS_LEAF_LIMIT")) { const int v = atoi(e2); if (v >= 1 && v <= 14) limit = v; }
    k_mark_leaves<<<blocks_for(2 * n - 1), kBlock, 0, s>>>(T, n, limit);
    bytes = T.iscanTmpBytes;
    if ((e = hipcub::DeviceScan::ExclusiveSum(T.iscanTmp, bytes, (const int*)T.lstart, T.lsum, n + 1, s)) != hipSuccess) return e;
    int leaves = 0;
    if ((e = hipMemcpyAsync(&leaves, T.lsum + n, sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return e;
    T.nT = 2 * leaves - 1;
    if (leaves < 1 || T.nT > T.capT) return hipErrorInvalidValue;
    if (leavesOut) *leavesOut = leaves;
    k_tlas_emit<<<blocks_for(2 * n - 1), kBlock, 0, s>>>(T, limit);
    return hipGetLastError();
}

// Second tree of a fast-sphere scene (hrt_trace_packed.hpp, "Closest-hit walks over the SECOND tree"): every node box -- not the
// instance records inlined into tlasX, whose own test must stay the reference's -- grows by a slack that covers how far the
// computed slab entry of an instance inside it can exceed the computed hit distance of its sphere:
//   4 sqrt(2 rho 2^-24 m) + 2^-7 rho + 2^-19 m,  rho = half the node's longest side (>= any radius inside), m = its largest |coordinate|
// (the cap of a sphere outside its rounded box, the error of a grazing hit's t in units of the radius, a few ulps of the coordinates so
// that a ray skimming a face within rounding of it is inside the slab; tests/test_second_tree_bound.py probes the sum).
__device__ void inflate_box(float4& lo, float4& hi)
{
    const float ex = hi.x - lo.x, ey = hi.y - lo.y, ez = hi.z - lo.z;
    if (!(ex >= 0.f && ey >= 0.f && ez >= 0.f)) return;                 // empty / unset box
    const float rho = 0.5f * fmaxf(ex, fmaxf(ey, ez));
    const float m = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y))), fmaxf(fabsf(lo.z), fabsf(hi.z)));
    const float s = 1.0625f * (4.f * sqrtf(2.f * rho * 0x1p-24f * m) + 0x1p-7f * rho + 0x1p-19f * m) + 1e-30f;
    lo.x = nextafterf(lo.x - s, -INFINITY); lo.y = nextafterf(lo.y - s, -INFINITY); lo.z = nextafterf(lo.z - s, -INFINITY);
    hi.x = nextafterf(hi.x + s, INFINITY);  hi.y = nextafterf(hi.y + s, INFINITY);  hi.z = nextafterf(hi.z + s, INFINITY);
}
__global__ void __launch_bounds__(kBlock) k_inflate(TlasDevice T)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < T.nT) { NodeQ q = T.tlas[i]; inflate_box(q.lo, q.hi); T.tlas[i] = q; }
    if (i < T.nT + T.nTI)
    {
        NodeQ q = T.tlasX[i];
        if (((unsigned)f2i(q.hi.w) >> 28) != 15u) { inflate_box(q.lo, q.hi); T.tlasX[i] = q; }
    }
}
hipError_t tlas_inflate(const TlasDevice& T, hipStream_t s)
{
    k_inflate<<<blocks_for(T.nT + T.nTI), kBlock, 0, s>>>(T);
    return hipGetLastError();
}

__global__ void __launch_bounds__(kBlock) k_refresh_copies(NodeQ* dst, const NodeQ* src, const int* map, int n)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const NodeQ q = src[map[i]];
    dst[i].lo.x = q.lo.x; dst[i].lo.y = q.lo.y; dst[i].lo.z = q.lo.z;
    dst[i].hi.x = q.hi.x; dst[i].hi.y = q.hi.y; dst[i].hi.z = q.hi.z;
}
hipError_t tlas_refresh_copies(NodeQ* dst, const NodeQ* src, const int* map, int n, hipStream_t s)
{
    if (n > 0) k_refresh_copies<<<blocks_for(n), kBlock, 0, s>>>(dst, src, map, n);
    return hipGetLastError();
}
__global__ void __launch_bounds__(kBlock) k_slot_of_inst(const int32_t* inst, int* slotOfInst, int n)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) slotOfInst[inst[i]] = i;
}
__global__ void __launch_bounds__(kBlock) k_slot_map(const int32_t* instInUse, const int* slotOfInst, int* map, int n)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) map[i] = slotOfInst[instInUse[i]];
}
hipError_t tlas_slot_map(const int32_t* instInUse, const int32_t* instSecond, int* slotOfInst, int* map, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    k_slot_of_inst<<<blocks_for(n), kBlock, 0, s>>>(instSecond, slotOfInst, n);
    k_slot_map<<<blocks_for(n), kBlock, 0, s>>>(instInUse, slotOfInst, map, n);
    return hipGetLastError();
}

hipError_t tlas_finish(const TlasDevice& T, hipStream_t s)
{
    hipError_t e;
    if ((e = hipMemsetAsync(T.flags, 0, 4 * sizeof(int), s)) != hipSuccess) return e;
    if (T.directMax <= 1 && (e = hipMemsetAsync(T.arrive, 0, (size_t)T.nT * sizeof(int), s)) != hipSuccess) return e;   // only the climb counts arrivals
    if (T.nTI > 0) k_leaf_slots<<<blocks_for(T.nTI), kBlock, 0, s>>>(T);
    k_refit<<<blocks_for(T.nT), kBlock, 0, s>>>(T);
    if (T.directMax > 1)
        for (long long lo = T.directMax; lo < T.nT; lo *= 64)
            k_refit_level<<<blocks_for(T.nT), kBlock, 0, s>>>(T, (int)lo, (int)std::min<long long>(lo * 64, 0x7FFFFFFF));
    k_scan_input<<<blocks_for(T.nT), kBlock, 0, s>>>(T);
    size_t bytes = T.scanTmpBytes;
    if ((e = hipcub::DeviceScan::ExclusiveSum(T.scanTmp, bytes, (const unsigned long long*)T.scanIn, T.scanOut, T.nT, s)) != hipSuccess) return e;
    k_derive<<<blocks_for(T.nT), kBlock, 0, s>>>(T);
    k_cost_partial<<<blocks_for(T.nT), kBlock, 0, s>>>(T);
    k_cost<<<1, 1024, 0, s>>>(T, blocks_for(T.nT));
    return hipGetLastError();
}

} // namespace hrt
