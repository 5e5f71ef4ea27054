// hrt_trace_packed.hpp -- the fast tracer: same walk, same arithmetic, same results as
// TracerRef (SceneDeviceViews.cs:30-327), over a device-private repack of the scene that
// hrt_scene_upload builds once per commit.
//
// Why a repack: the reference's arrays make every TLAS leaf a chain of six dependent global
// loads (tlasInstanceIndices[i] -> InstanceRecord 144 B -> BLASNode 44 B -> primIdx ->
// Sphere 80 B, or MeshTri 12 B -> 3 x Float3), each a fresh L1/L2 round trip on a machine
// whose rays are latency-bound, and its 44-byte nodes straddle 16-byte segments.  Here:
//   NodeQ  32 B  two aligned dwordx4 loads per node: {bmin, link} {bmax, skip | count << 28}
//   FInst  48 B  one record per TLAS leaf slot, in leaf order.  For the common case the
//                reference's own default scene uses (identity transform, one sphere, one-node
//                BLAS) it carries the BLAS root box and the sphere inline: the leaf is
//                ONE hop from the TLAS node.  Other instances keep (blasRoot, blasEnd, scale)
//                and an "identity" bit that skips TransformRay (x*1 + y*0 + z*0 + 0 only
//                changes the sign of zeros, which no comparison, division guard or store of
//                this path can observe -- DESIGN.md "exactness").
//   FTri   48 B  one record per BLAS leaf slot: the three vertices, triangle index, material
//                index and two flags resolved at upload (needs texture/alpha path, two-sided),
//                in leaf order, i.e. spatially sorted by the builder.
// Traversal is "while-while": every lane first walks inner nodes until it stands on a leaf
// (cheap box tests, all lanes busy), then all lanes run the expensive leaf body together,
// instead of idling lanes at inner nodes while one neighbour intersects primitives.
// Only (t, leaf slot, primitive slot) of the best hit stay live; normal, albedo, texture
// lookups are pure functions of the winner and are evaluated once, after the walk.
#pragma once
#include "hrt_device.hpp"

namespace hrt {

constexpr int kEnd = 0x0FFFFFFF;            // packed form of the -1 link

struct NodeQ { float4 lo, hi; };            // lo.w = bits(left | first), hi.w = bits(skip | count << 28)
struct FInst { float4 a, b, c; };           // a.w = bits(flags), b.w = bits(index); see hrt_runtime.hip pack_scene()
struct FTri  { float4 v0, v1, v2; };        // v0.w = bits(triIndex), v1.w = bits(matIndex), v2.w = bits(flags)

enum { FI_FAST_SPHERE = 1, FI_IDENTITY = 2, FI_SPHERESET = 4 };
enum { FT_TEXTURED = 1, FT_TWOSIDED = 2 };  // FT_TEXTURED: usable diffuse or alpha map, or AlphaCutoff > 1 (rejects alpha = 1)

// Numberings of the second tree (DPacked::tlasXO): `axes` (bit a: axis a) says which signs of a ray's direction select one; the copy
// index packs those signs, lowest selected axis first.
HRT_HD int ord_copy(int axes, float dx, float dy, float dz)
{
    int k = 0, b = 0;
    if (axes & 1) { k |= (dx > 0.f ? 1 : 0) << b; b++; }
    if (axes & 2) { k |= (dy > 0.f ? 1 : 0) << b; b++; }
    if (axes & 4) { k |= (dz > 0.f ? 1 : 0) << b; b++; }
    return k;
}
HRT_HD int ord_copies(int axes) { return 1 << ((axes & 1) + ((axes >> 1) & 1) + ((axes >> 2) & 1)); }

struct DPacked {
    const NodeQ* tlas;
    const FInst* finst;      // indexed like tlasInstanceIndices
    const NodeQ* blas;       // indexed like blasNodes
    const FTri* ftri;        // indexed like triPrimIdx
    int nTlas;               // records in tlas (>= 1)
    const NodeQ* tlasX;      // sphere-instance scenes (FEAT 0): the TLAS with every leaf followed by one record per instance
    int nTlasX;              //   (box of the instance's one-node BLAS, count field 15, link = leaf slot, skip = next record): nullptr if not built
    int leafTris;            // triangle records the walker fetches per leaf step (2 or 3: hrt_walker.hpp)
    const int* slotMap;      // second tree over the same instances (hrt_walker.hpp, ALT): leaf slot of the uploaded tree -> leaf slot here; else nullptr
    const NodeQ* tlasXO;     // second tree only: tlasX two or four times over, one numbering per combination of the signs of a ray's direction along
    int xStride;             //   the axes xAxes (ord_copy), with the child that is nearer along such a ray first; copy o occupies
    int xAxes;               //   [o * xStride, (o + 1) * xStride) and its links are indices into the whole array.  Leaf slots (finst, slotMap) are
                             //   shared by all copies.  nullptr: not built
    const NodeQ* tlasO;      // the same numberings of the plain node array `tlas` (launch 1 walks that one: closest_raw), copy o at [o * oStride, ...)
    int oStride;
};

HRT_D bool hit_box(const Ray& r, float4 lo, float4 hi, float tMin, float tMax)   // SceneDeviceViews.cs:496-514
{
    float t1 = (lo.x - r.o.x) * r.inv.x;
    float t2 = (hi.x - r.o.x) * r.inv.x;
    float tmin = hrt_fmin(t1, t2);
    float tmax = hrt_fmax(t1, t2);
    t1 = (lo.y - r.o.y) * r.inv.y;
    t2 = (hi.y - r.o.y) * r.inv.y;
    tmin = hrt_fmax(tmin, hrt_fmin(t1, t2));
    tmax = hrt_fmin(tmax, hrt_fmax(t1, t2));
    t1 = (lo.z - r.o.z) * r.inv.z;
    t2 = (hi.z - r.o.z) * r.inv.z;
    tmin = hrt_fmax(tmin, hrt_fmin(t1, t2));
    tmax = hrt_fmin(tmax, hrt_fmax(t1, t2));
    return tmax >= hrt_fmax(tmin, tMin) && tmin <= tMax;
}
// tmin of the test above (the slab entry distance as IntersectAABB computes it)
HRT_D float box_entry(const Ray& r, float4 lo, float4 hi)
{
    float t1 = (lo.x - r.o.x) * r.inv.x;
    float t2 = (hi.x - r.o.x) * r.inv.x;
    float tmin = hrt_fmin(t1, t2);
    t1 = (lo.y - r.o.y) * r.inv.y;
    t2 = (hi.y - r.o.y) * r.inv.y;
    tmin = hrt_fmax(tmin, hrt_fmin(t1, t2));
    t1 = (lo.z - r.o.z) * r.inv.z;
    t2 = (hi.z - r.o.z) * r.inv.z;
    return hrt_fmax(tmin, hrt_fmin(t1, t2));
}
// Closest-hit walks over the SECOND tree of a fast-sphere scene (DPacked::slotMap; hrt_walker.hpp kTies, TracerSecond below).
// What the reference computes is, with one exception, the accepted hit of least t over the instances whose own box test (limit
// 1e30) passes, ties going to the first in its walk order:
//   (R) an instance w whose own slab entry is <= its hit distance (tmin_w <= t_w) is never pruned by the reference while it
//       could still win: every box above it contains its own, so their entries are <= tmin_w <= t_w <= the closest t so far.
// The exception is a winner whose computed entry EXCEEDS its computed hit distance (rounding: the float box can be a fraction of
// an ulp smaller than the sphere, and t of a grazing hit carries an absolute error of ~7e-4 of the distance), which a node test
// with a closest t between the two can skip or not, depending on the order.  So:
//   (1) the second tree's node boxes are inflated (hrt_bvh.hip, inflate_box) and its node tests take closest * kSecondLimit, so
//       that no node holding an instance with a valid hit t_i < closest is pruned -- the walk finds the least t over ALL candidates;
//   (2) a candidate at exactly the closest t so far (a tie), or a winner with tmin_w > t_w, sends the ray to the uploaded tree.
// Why (1) holds.  The point o + t_i d of a valid hit -- a true one up to the rounding of t (at grazing incidence the square root
// amplifies the discriminant's rounding to ~7e-4 of the distance), or the closest approach of a ray that misses by rounding --
// lies within r (1 + 2^-8) + 2^-9 t_i of the centre, hence inside the instance's box grown by the cap of the sphere that its
// rounded corners cut off (chord 2 sqrt(2 r beta), beta = half an ulp of |c| + r), by 2^-7 r, and by a few ulps of the
// coordinates (a ray that skims a face within rounding of it computes that slab's entry with an error of ulp / |d_axis|, which
// is unbounded for the tight box and harmless once the face has moved away by more than the ulp).  inflate_box grows every
// node by at least that for anything inside it, so the node's computed entry is <= t_i (1 + 2^-8) < closest * (1 + 2^-7).
// The instance's OWN test keeps the exact box and the limit 1e30: it is the reference's.  The computed entry of the TIGHT box
// can exceed t_i by any amount in the skimming case, which is what (2) catches for the winner.
// tests/test_second_tree_bound.py samples adversarial (ray, sphere) pairs -- pole and silhouette hits, axis-parallel and skimming
// rays, coordinates from 1e-2 to 1e3, radii from 1e-3 of that up -- against "the grown box's computed entry <= t_i (1 + 2^-7)"
// (27 million candidates in a one-off run, no exception).
// The ORDER in which such a walk meets the children of a node is free as well (nothing above uses it): the tree exists in several
// numberings (DPacked::tlasXO, one per combination of direction signs, near child first) and a closest-hit walk takes its ray's.
constexpr float kSecondLimit = 1.f + 0x1p-7f;

// The world ray is dead weight while a general instance's BLAS is walked with the object-space
// ray, but it must survive for the rest of the TLAS walk.  hipcc can only spill to scratch
// (global memory, hundreds of cycles inside a latency-bound loop); parking the 9 floats in LDS
// instead costs one ds_write/ds_read pair per general leaf and keeps the walk at 64 VGPRs.
// Layout [component][thread]: conflict-free, 9 KiB per 256-thread workgroup.
struct RayPark {
    float (*sh)[256];
    HRT_D void put(const Ray& r) const
    {
        const int t = threadIdx.x;
        sh[0][t] = r.o.x; sh[1][t] = r.o.y; sh[2][t] = r.o.z; sh[3][t] = r.d.x; sh[4][t] = r.d.y; sh[5][t] = r.d.z;
        sh[6][t] = r.inv.x; sh[7][t] = r.inv.y; sh[8][t] = r.inv.z;
    }
    HRT_D Ray get() const
    {
        const int t = threadIdx.x;
        Ray r;
        r.o = mk3(sh[0][t], sh[1][t], sh[2][t]); r.d = mk3(sh[3][t], sh[4][t], sh[5][t]); r.inv = mk3(sh[6][t], sh[7][t], sh[8][t]);
        return r;
    }
};
HRT_D F3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }
// TransformPoint / TransformVector (SceneDeviceViews.cs:483-493) with the identity matrix, evaluated literally:
// ((1*x + 0*y) + 0*z) (+ 0) is x for x != 0 but decides the SIGN of a zero result.  The walks skip the transform of identity
// instances (no comparison, division guard or t of the walk can see a zero's sign); the winner's shading, whose normal is
// stored, applies it, so every output stays bit-for-bit what the reference's statement order produces.
HRT_D F3 ident_point(F3 p)
{
    return mk3(((1.f * p.x + 0.f * p.y) + 0.f * p.z) + 0.f, ((0.f * p.x + 1.f * p.y) + 0.f * p.z) + 0.f, ((0.f * p.x + 0.f * p.y) + 1.f * p.z) + 0.f);
}
HRT_D F3 ident_vector(F3 v)
{
    return mk3((1.f * v.x + 0.f * v.y) + 0.f * v.z, (0.f * v.x + 1.f * v.y) + 0.f * v.z, (0.f * v.x + 0.f * v.y) + 1.f * v.z);
}
HRT_D int wbits(float4 v) { return __float_as_int(v.w); }
HRT_D bool finite_ray(const Ray& r)
{
    return hrt_isfinite(r.inv.x) && hrt_isfinite(r.inv.y) && hrt_isfinite(r.inv.z) &&
           hrt_isfinite(r.o.x) && hrt_isfinite(r.o.y) && hrt_isfinite(r.o.z);
}

// FEAT bit 0: the scene has leaf slots that are not fast spheres (general walkers compiled in)
// FEAT bit 1: some triangle needs the in-walk alpha / texture path (FT_TEXTURED)
// hrt_scene_upload picks the smallest variant that covers the committed scene: code that cannot
// run is not compiled in, which keeps the sphere-only walk within 64 VGPRs (8 waves/SIMD).
template <int FEAT>
struct TracerPackedT {
    static constexpr bool kGeneral = (FEAT & 1) != 0;
    static constexpr bool kAlpha = (FEAT & 2) != 0;
    DPacked P;
    DScene S;     // original arrays: winners' shading data, general-instance transforms, textures

    // object-space ray of a leaf slot (TransformRay, SceneDeviceViews.cs:475-481)
    HRT_D Ray object_ray(const Ray& w, int flags, int instIdx) const
    {
        if (flags & FI_IDENTITY) return w;
        const hrt_instance* inst = &S.instances[instIdx];
        Ray r;
        r.o = xform_point(inst->worldToObject, w.o);
        r.d = xform_vector(inst->worldToObject, w.d);
        r.inv = inv_dir(r.d);
        return r;
    }

    // the object-space ray the winner is shaded with: as object_ray, but an identity transform is evaluated literally
    // (origin and direction only: shading never reads 1/d)
    static HRT_D Ray ident_ray(const Ray& w) { Ray r; r.o = ident_point(w.o); r.d = ident_vector(w.d); r.inv = w.inv; return r; }
    HRT_D Ray shading_ray(const Ray& w, int flags, int instIdx) const
    {
        if (flags & FI_IDENTITY) return ident_ray(w);
        return object_ray(w, flags, instIdx);
    }

    // UV + texture path of one triangle hit (SceneDeviceViews.cs:201-218 / :297-315)
    HRT_D void tri_uv(int ti, float bu, float bv, float& uu, float& vv) const
    {
        hrt_mesh_tri_uv tuv = S.meshTriUVs[ti];
        hrt_float2 t0 = S.meshTexcoords[tuv.t0], t1 = S.meshTexcoords[tuv.t1], t2 = S.meshTexcoords[tuv.t2];
        float w = 1.f - bu - bv;
        uu = t0.X * w + t1.X * bu + t2.X * bv;
        vv = t0.Y * w + t1.Y * bu + t2.Y * bv;
    }

    // ---------------- closest hit inside one BLAS, triangles (TraverseBLAS_Tri_Textured :173-237)
    template <bool COUNT>
    HRT_D void blas_tris_closest(const Ray& iray, int blasStart, int blasEnd, float& tObj, int& slot, Cnt<COUNT>& C) const
    {
        Tex tex(S);
        int cur = blasStart;
        for (;;)
        {
            int lfirst = 0, lcount = 0, lskip = kEnd;
            while (cur < blasEnd)                       // kEnd >= blasEnd: also the "cur != -1" test
            {
                NodeQ n = P.blas[cur];
                C.inc(C_NODE_VISITS);
                int sk = wbits(n.hi);
                int cnt = (int)((unsigned)sk >> 28);
                sk &= kEnd;
                if (!hit_box(iray, n.lo, n.hi, 0.001f, tObj)) { cur = sk; continue; }
                if (cnt > 0) { lfirst = wbits(n.lo); lcount = cnt; lskip = sk; break; }
                cur = wbits(n.lo) & kEnd;
            }
            if (lcount == 0) break;
            for (int j = lfirst; j < lfirst + lcount; j++)
            {
                FTri tr = P.ftri[j];
                C.inc(C_TRI_TESTS);
                float t, bu, bv;
                if (hit_tri_t(iray, xyz(tr.v0), xyz(tr.v1), xyz(tr.v2), t, bu, bv))
                {
                    C.inc(C_TRI_MT_HITS);
                    if (t > 0.001f && t < tObj)
                    {
                        C.inc(C_TRI_ACCEPTED);
                        bool accept = true;
                        if (kAlpha && (wbits(tr.v2) & FT_TEXTURED))
                        {   // `if (alpha < mat.AlphaCutoff) continue;` (:209-218) can reject: evaluate it now.
                            // (flag is also set for map-less materials whose cutoff exceeds alpha = 1)
                            const hrt_material* mat = &S.materials[wbits(tr.v1)];
                            int ati = mat->AlphaTexIndex;
                            float alpha = 1.f;
                            if (mat->HasAlphaMap != 0 && ati >= 0 && ati < S.n_texInfos)
                            {
                                float uu, vv;
                                tri_uv(wbits(tr.v0), bu, bv, uu, vv);
                                alpha = tex.mask_linear(S.texInfos[ati], uu, vv);
                            }
                            accept = !(alpha < mat->AlphaCutoff);
                        }
                        if (accept) { tObj = t; slot = j; }
                    }
                }
            }
            cur = lskip;
        }
    }

    // ---------------- closest hit inside one BLAS, spheres (TraverseBLAS_Sphere :124-170)
    template <bool COUNT>
    HRT_D void blas_spheres_closest(const Ray& iray, int blasStart, int blasEnd, float& tObj, int& sphereIdx, Cnt<COUNT>& C) const
    {
        int cur = blasStart;
        for (;;)
        {
            int lfirst = 0, lcount = 0, lskip = kEnd;
            while (cur < blasEnd)
            {
                NodeQ n = P.blas[cur];
                C.inc(C_NODE_VISITS);
                int sk = wbits(n.hi);
                int cnt = (int)((unsigned)sk >> 28);
                sk &= kEnd;
                if (!hit_box(iray, n.lo, n.hi, 0.001f, tObj)) { cur = sk; continue; }
                if (cnt > 0) { lfirst = wbits(n.lo); lcount = cnt; lskip = sk; break; }
                cur = wbits(n.lo) & kEnd;
            }
            if (lcount == 0) break;
            for (int j = lfirst; j < lfirst + lcount; j++)
            {
                int p = S.spherePrimIdx[j];
                const hrt_sphere* sp = &S.spheres[p];
                C.inc(C_SPHERE_TESTS);
                float t;
                if (hit_sphere_t(iray, cv3(sp->center), sp->radius, t) && t > 0.001f && t < tObj) { tObj = t; sphereIdx = p; }
            }
            cur = lskip;
        }
    }

    // ---------------- TraceClosest (:30-86)
    template <bool COUNT>
    HRT_D bool closest(const Ray& wray_in, Hit& best, Cnt<COUNT>& C) const
    {
        float bestT, bestTObj; int bestSlot, bestPrim;
        closest_raw<COUNT>(wray_in, bestT, bestTObj, bestSlot, bestPrim, C);
        return finish_hit(wray_in, bestT, bestTObj, bestSlot, bestPrim, best);
    }
    // the walk alone: raw winner (world t, object-space t, TLAS leaf slot, primitive), t = 1e30 on a miss
    // EXT: the caller lends its own LDS ray park (kernels that already have one must not pay for a second)
    // TIES: *tie is set when a fast-sphere candidate lies at exactly the current best distance (TracerSecond below)
    template <bool COUNT, bool EXT = false, bool TIES = false>
    HRT_D void closest_raw(const Ray& wray_in, float& bestT, float& bestTObj, int& bestSlot, int& bestPrim, Cnt<COUNT>& C, float (*ext)[256] = nullptr,
                           bool* tie = nullptr) const
    {
        RayPark park;
        if constexpr (EXT) park.sh = ext;
        else { __shared__ float park_mem[kGeneral ? 9 : 1][256]; park.sh = park_mem; }
        Ray wray = wray_in;
        C.inc(C_RAYS_CLOSEST);
        bestT = 1e30f;              // closestT (world)
        bestTObj = 0.f;             // object-space t of the winner (== bestT * scale)
        bestSlot = -1;              // TLAS leaf slot (index into finst / tlasInstanceIndices)
        bestPrim = -1;              // sphere index, or BLAS leaf slot of the triangle
        bool anom = false;          // TIES: the current winner's own slab entry exceeds its hit distance
        int cur = 0;
        const NodeQ* nodes = P.tlas;
#ifndef HRT_NO_ORDERED_PRIMARY     // A/B
        if (TIES && P.tlasO != nullptr)     // the second tree in the numbering of this ray's direction (DPacked::tlasXO): near child first
        {
            nodes = P.tlasO;
            cur = ord_copy(P.xAxes, wray.d.x, wray.d.y, wray.d.z) * P.oStride;
        }
#endif
        for (;;)
        {
            int lfirst = 0, lcount = 0, lskip = kEnd;
            while (cur != kEnd)
            {
                NodeQ n = nodes[cur];
                C.inc(C_NODE_VISITS);
                int sk = wbits(n.hi);
                int cnt = (int)((unsigned)sk >> 28);
                sk &= kEnd;
                if (!hit_box(wray, n.lo, n.hi, 0.001f, TIES ? bestT * kSecondLimit : bestT)) { cur = sk; continue; }
                if (cnt > 0) { lfirst = wbits(n.lo); lcount = cnt; lskip = sk; break; }
                cur = wbits(n.lo) & kEnd;
            }
            if (lcount == 0) { if (TIES && anom) *tie = true; return; }
            for (int i = lfirst; i < lfirst + lcount; i++)
            {
                FInst f = P.finst[i];
                C.inc(C_LEAF_INST);
                const int flags = wbits(f.a);
                if (flags & FI_FAST_SPHERE)
                {   // identity instance, one sphere, one-node BLAS: root box (tMax 1e30) then the sphere
                    C.inc(C_NODE_VISITS);
                    if (hit_box(wray, f.a, f.b, 0.001f, 1e30f))
                    {
                        C.inc(C_SPHERE_TESTS);
                        float t;
                        if (hit_sphere_t(wray, xyz(f.c), f.c.w, t) && t > 0.001f && t < 1e30f)
                        {
                            // hit = tClosest < 1e29 (:169); tWorld = t / 1 (:67)
                            if (t < 1e29f && t < bestT)
                            {
                                bestT = t; bestTObj = t; bestSlot = i; bestPrim = wbits(f.b);
                                if (TIES) anom = box_entry(wray, f.a, f.b) > t;
                            }
                            else if (TIES && t == bestT) *tie = true;
                        }
                    }
                }
                else if (kGeneral)
                {
                    const int instIdx = wbits(f.b);
                    const int blasStart = __float_as_int(f.c.x), blasEnd = __float_as_int(f.c.y);
                    const float scale = f.c.z;
                    float tObj = 1e30f; int prim = -1;
                    {
                        park.put(wray);                              // world ray rests in LDS during the BLAS walk
                        Ray iray = object_ray(wray, flags, instIdx);
                        if (flags & FI_SPHERESET) blas_spheres_closest<COUNT>(iray, blasStart, blasEnd, tObj, prim, C);
                        else                      blas_tris_closest<COUNT>(iray, blasStart, blasEnd, tObj, prim, C);
                        wray = park.get();
                    }
                    if (tObj < 1e29f)
                    {
                        float tWorld = tObj / scale;
                        if (tWorld < bestT) { bestT = tWorld; bestTObj = tObj; bestSlot = i; bestPrim = prim; }
                    }
                }
            }
            cur = lskip;
        }
    }

    // ---- shade the winner of a walk once (normal :534-535/:556, albedo :146-159/:210-223, world normal :71).
    // Pure function of (ray, winner): the streamed pipeline calls it from a separate kernel.
    HRT_D bool finish_hit(const Ray& wray, float bestT, float bestTObj, int bestSlot, int bestPrim, Hit& best) const
    {
        best.t = bestT; best.n = mk3(0.f, 0.f, 0.f); best.albedo = mk3(1.f, 1.f, 1.f); best.objId = -1; best.shade = 0; best.ior = 1.f;
        if (!(bestT < 1e29f)) return false;

        FInst f = P.finst[bestSlot];
        const int flags = wbits(f.a);
        Tex tex(S);
        F3 nObj;
        if (!kGeneral || (flags & (FI_FAST_SPHERE | FI_SPHERESET)))
        {
            Ray iray = (!kGeneral || (flags & FI_FAST_SPHERE)) ? ident_ray(wray) : shading_ray(wray, flags, wbits(f.b));
            const hrt_sphere* sp = &S.spheres[bestPrim];
            nObj = sphere_normal(iray, cv3(sp->center), bestTObj);
            F3 kd = cv3(sp->material.Kd);
            F3 alb = (kd.x == 0.f && kd.y == 0.f && kd.z == 0.f) ? cv3(sp->albedo) : kd;
            int dti = sp->material.DiffuseTexIndex;
            if (sp->material.HasDiffuseMap != 0 && dti >= 0 && dti < S.n_texInfos)
            {
                float u = 0.5f + hrt_atan2(nObj.z, nObj.x) / (2.f * kPI);
                float v = hrt_acos(hrt_fmin(1.f, hrt_fmax(-1.f, nObj.y))) / kPI;
                alb = tex.linear_rgb(S.texInfos[dti], u, v);
            }
            best.albedo = alb;
            best.shade = sp->shading;
            float sior = sp->ior;
            best.ior = sior > 0.f ? sior : 1.f;
            best.objId = -1;
        }
        else
        {
            Ray iray = shading_ray(wray, flags, wbits(f.b));
            FTri tr = P.ftri[bestPrim];
            F3 v0 = xyz(tr.v0), v1 = xyz(tr.v1), v2 = xyz(tr.v2);
            nObj = normalize(cross(v1 - v0, v2 - v0));
            const int tflags = wbits(tr.v2);
            if ((tflags & FT_TWOSIDED) && dot(nObj, iray.d) > 0.f) nObj = nObj * -1.f;
            const hrt_material* mat = &S.materials[wbits(tr.v1)];
            F3 kd = cv3(mat->Kd);
            if (tflags & FT_TEXTURED)
            {
                int dti = mat->DiffuseTexIndex;
                if (mat->HasDiffuseMap != 0 && dti >= 0 && dti < S.n_texInfos)
                {
                    float t, bu, bv, uu, vv;
                    hit_tri_t(iray, v0, v1, v2, t, bu, bv);          // same inputs -> same (bu, bv) as in the walk
                    tri_uv(wbits(tr.v0), bu, bv, uu, vv);
                    kd = tex.linear_rgb(S.texInfos[dti], uu, vv);
                }
            }
            best.albedo = kd;
            best.objId = wbits(tr.v0);
        }
        if (!kGeneral || (flags & FI_IDENTITY)) best.n = normalize(ident_vector(nObj));   // objectToWorld = I, evaluated literally (zero signs)
        else best.n = normalize(xform_vector(S.instances[wbits(f.b)].objectToWorld, nObj));
        return true;
    }

    // ---------------- any hit inside one BLAS (AnyHit_Tri_Textured :270-327, AnyHit_Sphere :240-267)
    template <bool COUNT>
    HRT_D bool blas_tris_any(const Ray& iray, int blasStart, int blasEnd, float tMaxObj, Cnt<COUNT>& C) const
    {
        Tex tex(S);
        int cur = blasStart;
        for (;;)
        {
            int lfirst = 0, lcount = 0, lskip = kEnd;
            while (cur < blasEnd)
            {
                NodeQ n = P.blas[cur];
                C.inc(C_NODE_VISITS);
                int sk = wbits(n.hi);
                int cnt = (int)((unsigned)sk >> 28);
                sk &= kEnd;
                if (!hit_box(iray, n.lo, n.hi, 0.001f, tMaxObj)) { cur = sk; continue; }
                if (cnt > 0) { lfirst = wbits(n.lo); lcount = cnt; lskip = sk; break; }
                cur = wbits(n.lo) & kEnd;
            }
            if (lcount == 0) return false;
            for (int j = lfirst; j < lfirst + lcount; j++)
            {
                FTri tr = P.ftri[j];
                C.inc(C_TRI_TESTS);
                float t, bu, bv;
                if (hit_tri_t(iray, xyz(tr.v0), xyz(tr.v1), xyz(tr.v2), t, bu, bv))
                {
                    if (t <= 0.001f || t >= tMaxObj) continue;
                    C.inc(C_TRI_MT_HITS);
                    if (kAlpha && (wbits(tr.v2) & FT_TEXTURED))
                    {
                        const hrt_material* mat = &S.materials[wbits(tr.v1)];
                        int ati = mat->AlphaTexIndex;
                        if (mat->HasAlphaMap != 0 && ati >= 0 && ati < S.n_texInfos)
                        {
                            C.inc(C_TRI_ACCEPTED);
                            float uu, vv;
                            tri_uv(wbits(tr.v0), bu, bv, uu, vv);
                            hrt_tex_info ainfo = S.texInfos[ati];
                            float aPoint = tex.mask_point(ainfo, uu, vv);
                            float cutoff = mat->AlphaCutoff;
                            if (aPoint < cutoff - 0.10f) continue;
                            if (aPoint >= cutoff + 0.10f) return true;
                            if (tex.mask_linear(ainfo, uu, vv) < cutoff) continue;
                        }
                    }
                    return true;
                }
            }
            cur = lskip;
        }
    }
    template <bool COUNT>
    HRT_D bool blas_spheres_any(const Ray& iray, int blasStart, int blasEnd, float tMaxObj, Cnt<COUNT>& C) const
    {
        int cur = blasStart;
        for (;;)
        {
            int lfirst = 0, lcount = 0, lskip = kEnd;
            while (cur < blasEnd)
            {
                NodeQ n = P.blas[cur];
                C.inc(C_NODE_VISITS);
                int sk = wbits(n.hi);
                int cnt = (int)((unsigned)sk >> 28);
                sk &= kEnd;
                if (!hit_box(iray, n.lo, n.hi, 0.001f, tMaxObj)) { cur = sk; continue; }
                if (cnt > 0) { lfirst = wbits(n.lo); lcount = cnt; lskip = sk; break; }
                cur = wbits(n.lo) & kEnd;
            }
            if (lcount == 0) return false;
            for (int j = lfirst; j < lfirst + lcount; j++)
            {
                const hrt_sphere* sp = &S.spheres[S.spherePrimIdx[j]];
                C.inc(C_SPHERE_TESTS);
                float t;
                if (hit_sphere_t(iray, cv3(sp->center), sp->radius, t) && t > 0.001f && t < tMaxObj) return true;
            }
            cur = lskip;
        }
    }

    // ---------------- ShadowOcclusion (:89-121)
    template <bool COUNT>
    HRT_D bool occluded(const Ray& wray_in, float tMaxWorld, Cnt<COUNT>& C) const { return occluded_ext<COUNT, false>(wray_in, tMaxWorld, C, nullptr); }
    template <bool COUNT, bool EXT>
    HRT_D bool occluded_ext(const Ray& wray_in, float tMaxWorld, Cnt<COUNT>& C, float (*ext)[256]) const
    {
        RayPark park;
        if constexpr (EXT) park.sh = ext;
        else { __shared__ float park_mem[kGeneral ? 9 : 1][256]; park.sh = park_mem; }
        Ray wray = wray_in;
        C.inc(C_RAYS_SHADOW);
        int cur = 0;
        for (;;)
        {
            int lfirst = 0, lcount = 0, lskip = kEnd;
            while (cur != kEnd)
            {
                NodeQ n = P.tlas[cur];
                C.inc(C_NODE_VISITS);
                int sk = wbits(n.hi);
                int cnt = (int)((unsigned)sk >> 28);
                sk &= kEnd;
                if (!hit_box(wray, n.lo, n.hi, 0.001f, tMaxWorld)) { cur = sk; continue; }
                if (cnt > 0) { lfirst = wbits(n.lo); lcount = cnt; lskip = sk; break; }
                cur = wbits(n.lo) & kEnd;
            }
            if (lcount == 0) return false;
            for (int i = lfirst; i < lfirst + lcount; i++)
            {
                FInst f = P.finst[i];
                C.inc(C_LEAF_INST);
                const int flags = wbits(f.a);
                if (flags & FI_FAST_SPHERE)
                {
                    const float tMaxObj = tMaxWorld * 1.f;          // scale == 1 (:107)
                    C.inc(C_NODE_VISITS);
                    if (hit_box(wray, f.a, f.b, 0.001f, tMaxObj))
                    {
                        C.inc(C_SPHERE_TESTS);
                        float t;
                        if (hit_sphere_t(wray, xyz(f.c), f.c.w, t) && t > 0.001f && t < tMaxObj) return true;
                    }
                }
                else if (kGeneral)
                {
                    const int blasStart = __float_as_int(f.c.x), blasEnd = __float_as_int(f.c.y);
                    const float tMaxObj = tMaxWorld * f.c.z;
                    park.put(wray);
                    Ray iray = object_ray(wray, flags, wbits(f.b));
                    bool blocked = (flags & FI_SPHERESET) ? blas_spheres_any<COUNT>(iray, blasStart, blasEnd, tMaxObj, C)
                                                          : blas_tris_any<COUNT>(iray, blasStart, blasEnd, tMaxObj, C);
                    if (blocked) return true;
                    wray = park.get();
                }
            }
            cur = lskip;
        }
    }
};
using TracerPacked = TracerPackedT<3>;       // everything compiled in

// ---------------------------------------------------------------------------------------
// TracerFlat: scenes whose TLAS has a handful of leaves and only fast-sphere instances (the
// reference's default scene, BASELINE config 2).  The tree walk of such a scene is a few
// divergent dependent loads per ray; here every lane runs ONE wave-uniform loop over the
// TLAS LEAVES in walk order instead: leaf records and instance records are read through
// the scalar cache into SGPRs (uniform addresses), lanes differ only in their exec mask.
//
// Same results as the tree walk, by construction of the reference's own tests:
//  * a leaf is entered iff every box on its root path and its own box pass IntersectAABB
//    (SceneDeviceViews.cs:496-514) with the closest t of that moment; a parent's box contains
//    its children's (bounds are unions, exact in binary32), and in that slab test enlarging
//    a box can only turn a miss into a hit (every operation is monotone in the bound) while
//    the closest t only shrinks along the walk: so a leaf whose own test passes has passed
//    all its ancestors' tests -- inner nodes only accelerate, the leaf's own test decides;
//  * leaves are visited in the same order, so the closest t each leaf test sees, every
//    tie-break and the first any-hit are the same.
// Monotonicity needs finite slab arithmetic: a ray whose origin or 1/d is not finite (d
// component denormal) takes the tree walk.  Work counters are those of the tree walk and are
// not produced here: counting frames use TracerPackedT<0>.
// ---------------------------------------------------------------------------------------
constexpr int kFlatMaxLeaves = 16;
// One-ray-per-lane closest hit of a fast-sphere scene over the SECOND tree (DPacked::slotMap, hrt_walker.hpp ALT): the winner is
// the accepted hit of least t whatever the tree; a ray that met two instances at exactly the same distance, or whose slab
// arithmetic is not finite, walks the uploaded tree instead.  Leaf slots handed to finish_hit are the second tree's.
struct TracerSecond {
    TracerPackedT<0> second, uploaded;
    template <bool COUNT>
    HRT_D bool closest(const Ray& wray, Hit& best, Cnt<COUNT>& C) const
    {
        static_assert(!COUNT, "work counters are defined on the uploaded tree");
        float t = 1e30f, tObj = 0.f; int slot = -1, prim = -1;
        bool tie = !finite_ray(wray);
        if (!tie) second.template closest_raw<COUNT, false, true>(wray, t, tObj, slot, prim, C, nullptr, &tie);
        if (tie)
        {
            uploaded.template closest_raw<COUNT>(wray, t, tObj, slot, prim, C);
            if (slot >= 0) slot = second.P.slotMap[slot];
        }
        return second.finish_hit(wray, t, tObj, slot, prim, best);
    }
};

// Wave-uniform records through the SCALAR cache: a pointer in the constant address space makes the compiler fetch with s_load
// into scalar registers (one request per wave, ~4x shorter latency than the vector path, no vector registers) instead of a
// 64-lane global_load of one address.  Valid here because the scene arrays are never written while a render kernel runs.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4v* CF4;
HRT_D float4 sload4(const float4* base, int i)       // base[i] with a wave-uniform i
{
    const f4v v = ((CF4)base)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}

struct TracerFlat {
    TracerPackedT<0> tree;
    const NodeQ* leaves;         // TLAS leaf records in walk order
    int nLeaves;
    HRT_D NodeQ leaf(int l) const { NodeQ n; n.lo = sload4(&leaves->lo, 2 * l); n.hi = sload4(&leaves->lo, 2 * l + 1); return n; }
    HRT_D FInst inst(int i) const
    {
        FInst f; const float4* b = &tree.P.finst->a;
        f.a = sload4(b, 3 * i); f.b = sload4(b, 3 * i + 1); f.c = sload4(b, 3 * i + 2);
        return f;
    }

    template <bool COUNT>
    HRT_D bool closest(const Ray& wray, Hit& best, Cnt<COUNT>& C) const
    {
        if (COUNT || !finite_ray(wray)) return tree.template closest<COUNT>(wray, best, C);
        float bestT = 1e30f; int bestSlot = -1, bestPrim = -1;
        const float a = dot(wray.d, wray.d);
        for (int l = 0; l < nLeaves; l++)
        {
            const NodeQ n = leaf(l);
            PSTAT(14);
            if (!hit_box(wray, n.lo, n.hi, 0.001f, bestT)) continue;
            const int first = wbits(n.lo), cnt = (int)((unsigned)wbits(n.hi) >> 28);
            for (int i = first; i < first + cnt; i++)
            {
                const FInst f = inst(i);
                PSTAT(7);
                if (!hit_box(wray, f.a, f.b, 0.001f, 1e30f)) continue;
                PSTAT(8);
                float t;
                if (hit_sphere_ta(wray, a, xyz(f.c), f.c.w, t) && t > 0.001f && t < 1e30f && t < 1e29f && t < bestT)
                { bestT = t; bestSlot = i; bestPrim = wbits(f.b); }
            }
        }
        PSTAT(13);
        return tree.finish_hit(wray, bestT, bestT, bestSlot, bestPrim, best);
    }

    template <bool COUNT>
    HRT_D bool occluded(const Ray& wray, float tMaxWorld, Cnt<COUNT>& C) const
    {
        if (COUNT || !finite_ray(wray)) return tree.template occluded<COUNT>(wray, tMaxWorld, C);
        // An any-hit query is an OR over the instances whose sphere test the walk reaches, and by the lemma above the walk reaches
        // an instance iff the instance's OWN box test passes (tMax is fixed here, and a box that contains the instance's can only
        // turn a miss into a hit): the leaf boxes decide nothing, so they are not tested, and the order is free.
        bool hit = false;
        const float a = dot(wray.d, wray.d);
        for (int l = 0; l < nLeaves; l++)
        {
            const NodeQ n = leaf(l);
            const int first = wbits(n.lo), cnt = (int)((unsigned)wbits(n.hi) >> 28);
            for (int i = first; i < first + cnt; i++)
            {
                const FInst f = inst(i);
                PSTAT(11);
                if (hit || !hit_box(wray, f.a, f.b, 0.001f, tMaxWorld)) continue;
                PSTAT(12);
                float t;
                if (hit_sphere_ta(wray, a, xyz(f.c), f.c.w, t) && t > 0.001f && t < tMaxWorld) hit = true;
            }
            if (__builtin_amdgcn_ballot_w64(!hit) == 0) break;      // every live lane is occluded
        }
        return hit;
    }
};

} // namespace hrt
