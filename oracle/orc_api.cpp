/*
 * oracle/orc_api.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C entry points (ctypes) over the CPU restatement in orc_kernels.hpp / orc_scene.hpp:
 * the parity oracle for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * PARITY UNPINNED against the reference binary (see orc_kernels.hpp header).
 * Nothing under ilgpu_raytracing_amd/ may link, load or call this library.
 */
#include <thread>
#include <atomic>
#include <vector>
#include <chrono>
#include <cstdio>
#include "orc_kernels.hpp"
#include "orc_scene.hpp"
#include "orc_post.hpp"
#include "../include/hip_raytrace.h"   // hrt_outputs layout only

using namespace orc;

namespace {

template <class T> ArrayView<const T> view_or_empty(const T* p, int64_t n, const T* zero1)
{
    // Scene.AllocateOrEmpty (Scene.cs:370-377): empty -> one zeroed element, Length 1
    ArrayView<const T> v;
    if (p != nullptr && n > 0) { v.p = p; v.Length = n; }
    else { v.p = zero1; v.Length = 1; }
    return v;
}

struct Zeros {
    hrt_bvh_node node; int32_t i; hrt_instance inst; hrt_sphere sph; hrt_float3 f3; hrt_mesh_tri tri;
    hrt_float2 f2; hrt_mesh_tri_uv tuv; hrt_material mat; hrt_rgba32 px; hrt_tex_info ti;
    Zeros() { std::memset(this, 0, sizeof(*this)); }
};
static const Zeros g_zero;

SceneDeviceViews make_views(const hrt_scene_desc* d, Counters* C)
{
    SceneDeviceViews v;
    v.tlasNodes = view_or_empty(d->tlasNodes, d->n_tlasNodes, &g_zero.node);
    v.tlasInstanceIndices = view_or_empty(d->tlasInstanceIndices, d->n_tlasInstanceIndices, &g_zero.i);
    v.instances = view_or_empty(d->instances, d->n_instances, &g_zero.inst);
    v.blasNodes = view_or_empty(d->blasNodes, d->n_blasNodes, &g_zero.node);
    v.spherePrimIdx = view_or_empty(d->spherePrimIdx, d->n_spherePrimIdx, &g_zero.i);
    v.spheres = view_or_empty(d->spheres, d->n_spheres, &g_zero.sph);
    v.triPrimIdx = view_or_empty(d->triPrimIdx, d->n_triPrimIdx, &g_zero.i);
    v.meshPositions = view_or_empty(d->meshPositions, d->n_meshPositions, &g_zero.f3);
    v.meshTris = view_or_empty(d->meshTris, d->n_meshTris, &g_zero.tri);
    v.meshTexcoords = view_or_empty(d->meshTexcoords, d->n_meshTexcoords, &g_zero.f2);
    v.meshTriUVs = view_or_empty(d->meshTriUVs, d->n_meshTriUVs, &g_zero.tuv);
    v.triMatIndex = view_or_empty(d->triMatIndex, d->n_triMatIndex, &g_zero.i);
    v.materials = view_or_empty(d->materials, d->n_materials, &g_zero.mat);
    v.texels = view_or_empty(d->texels, d->n_texels, &g_zero.px);
    v.texInfos = view_or_empty(d->texInfos, d->n_texInfos, &g_zero.ti);
    v.C = C;
    return v;
}

template <class T> ArrayView<T> av(T* p, int64_t n) { ArrayView<T> v; v.p = p; v.Length = p ? n : 0; return v; }

void add_counters(hrt_kernel_counters& a, const hrt_kernel_counters& b)
{
    uint64_t* pa = reinterpret_cast<uint64_t*>(&a);
    const uint64_t* pb = reinterpret_cast<const uint64_t*>(&b);
    for (size_t i = 0; i < sizeof(hrt_kernel_counters) / sizeof(uint64_t); i++) pa[i] += pb[i];
}

// run fn(index, Counters&) over rows [y0,y1) with nthreads workers, dynamic 128-pixel chunks
template <class F>
void parallel_rows(int width, int y0, int y1, int nthreads, hrt_kernel_counters& total, F fn)
{
    if (nthreads < 1) nthreads = 1;
    const long long first = (long long)y0 * width, last = (long long)y1 * width;
    const int chunk = 128;
    std::atomic<long long> next(first);
    std::vector<Counters> cs(nthreads);
    auto work = [&](int tid) {
        Counters& C = cs[tid];
        for (;;) {
            long long b = next.fetch_add(chunk);
            if (b >= last) break;
            long long e = b + chunk < last ? b + chunk : last;
            for (long long i = b; i < e; i++) fn((int)i, C);
        }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    for (int t = 0; t < nthreads; t++) add_counters(total, cs[t]);
}

} // namespace

extern "C" {

// ------------------------------------------------------------------ frame render
// out: every gb_* array and color/depth/objectId must be non-NULL (P elements);
// radiance, cameraId and res_* optional.  prev->res_* = resPrev (read-only), may be NULL
// when both reuse flags are 0.  run_primary=0 reuses the G-buffer already in `out`.
int orc_render_frame(const hrt_scene_desc* scene, const hrt_frame_params* fp,
                     int row_begin, int row_end, int run_primary, int nthreads,
                     const hrt_outputs* out, const hrt_outputs* prev, hrt_stats* stats)
{
    if (!scene || !fp || !out) return -1;
    if (fp->width <= 0 || fp->height <= 0) return -1;
    if (!out->gb_worldPos || !out->gb_normalWS || !out->gb_baseColor || !out->gb_matId || !out->gb_objId || !out->gb_hitMask) return -1;
    if (!out->color || !out->depth || !out->objectId) return -1;
    if (row_begin == 0 && row_end == 0) row_end = fp->height;
    if (row_begin < 0 || row_end > fp->height || row_begin > row_end) return -1;
    int64_t P = (int64_t)fp->width * fp->height;

    hrt_stats st; std::memset(&st, 0, sizeof(st));
    st.n_devices = 0; st.counters_valid = 1;

    GpuGBuffer gb;
    gb.worldPos = av(out->gb_worldPos, P); gb.normalWS = av(out->gb_normalWS, P); gb.baseColor = av(out->gb_baseColor, P);
    gb.matId = av(out->gb_matId, P); gb.objId = av(out->gb_objId, P); gb.hitMask = av(out->gb_hitMask, P);

    if (run_primary)
    {
        auto t0 = std::chrono::steady_clock::now();
        parallel_rows(fp->width, row_begin, row_end, nthreads, st.k[0], [&](int index, Counters& C) {
            GBufferParams p;
            p.width = fp->width; p.height = fp->height; p.frame = fp->frame; p.cam = fp->cam;
            p.views = make_views(scene, &C); p.gb = gb;
            RTRay::PrimaryVisibilityKernel(index, p);
        });
        st.kernel_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }

    if (run_primary != 2)        // 2 = launch 1 only (tile-exchange tests: the G-buffer is exchanged between the launches)
    {
        GpuFramebuffer fb;
        fb.color = av(out->color, P); fb.depth = av(out->depth, P); fb.objectId = av(out->objectId, P);
        fb.cameraId = av(out->cameraId, 1); fb.radiance = av(out->radiance, P);
        GpuReservoirSoA cur, prv;
        cur.L = av(out->res_L, P); cur.wi = av(out->res_wi, P); cur.pdf = av(out->res_pdf, P); cur.w = av(out->res_w, P);
        cur.wSum = av(out->res_wSum, P); cur.m = av(out->res_m, P); cur.lightId = av(out->res_lightId, P);
        if (prev) {
            prv.L = av(prev->res_L, P); prv.wi = av(prev->res_wi, P); prv.pdf = av(prev->res_pdf, P); prv.w = av(prev->res_w, P);
            prv.wSum = av(prev->res_wSum, P); prv.m = av(prev->res_m, P); prv.lightId = av(prev->res_lightId, P);
        } else {
            prv.L = av<hrt_float3>(nullptr, 0); prv.wi = prv.L; prv.pdf = av<float>(nullptr, 0); prv.w = prv.pdf; prv.wSum = prv.pdf;
            prv.m = av<int32_t>(nullptr, 0); prv.lightId = prv.m;
        }
        auto t0 = std::chrono::steady_clock::now();
        parallel_rows(fp->width, row_begin, row_end, nthreads, st.k[1], [&](int index, Counters& C) {
            IntegratorParams k;
            k.width = fp->width; k.height = fp->height; k.frame = fp->frame;
            k.cam = fp->cam; k.prevCam = fp->prevCam;
            k.views = make_views(scene, &C); k.gb = gb; k.fb = fb;
            k.dirLightDir = fp->dirLightDir; k.dirLightRadiance = fp->dirLightRadiance;
            k.skyTintTop = fp->skyTintTop; k.skyTintBottom = fp->skyTintBottom;
            k.debugCamSeq = fp->debugCamSeq;
            k.resPrev = prv; k.resCur = cur;
            k.enableTemporalReuse = fp->enableTemporalReuse; k.enableSpatialReuse = fp->enableSpatialReuse;
            k.rngLockNoise = fp->rngLockNoise; k.spp = fp->spp;
            RTRay::PathTraceKernel(index, k, fp->maxDepth);
        });
        st.kernel_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (stats) *stats = st;
    return 0;
}

// ------------------------------------------------------------------ single-ray probes (tests)
// mode 0: SceneDeviceViews.TraceClosest; mode 1: brute force over every instance and every
// primitive of it (no BVH, no box tests) with the same intersection routines.
int orc_trace_rays(const hrt_scene_desc* scene, int n, const float* origins, const float* dirs, int mode,
                   float* t_out, float* normal_out, float* albedo_out, int* objid_out, int* shade_out, int* hit_out)
{
    Counters C;
    SceneDeviceViews v = make_views(scene, &C);
    for (int r = 0; r < n; r++)
    {
        Ray ray;
        ray.origin = Float3(origins[3 * r], origins[3 * r + 1], origins[3 * r + 2]);
        ray.dir = Float3(dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2]);
        ray.invDir = InvDir(ray.dir);
        float t = 1e30f; Float3 nrm, alb(1.f, 1.f, 1.f); int obj = -1, shade = 0; float ior = 1.f; bool hit;
        if (mode == 0) hit = v.TraceClosest(ray, t, nrm, alb, obj, shade, ior);
        else
        {
            for (int64_t ii = 0; ii < scene->n_instances; ii++)
            {
                const hrt_instance& inst = scene->instances[ii];
                Ray iray = SceneDeviceViews::TransformRay(inst.worldToObject, ray);
                float scale = inst.uniformScale > 0.f ? inst.uniformScale : 1.f;
                if (inst.type == HRT_BLAS_SPHERESET)
                {
                    for (int j = 0; j < inst.primIndexCount; j++)
                    {
                        const hrt_sphere& s = scene->spheres[scene->spherePrimIdx[inst.primIndexFirst + j]];
                        float tt; Float3 nn;
                        if (SceneDeviceViews::IntersectSphere(iray, s, tt, nn) && tt > 0.001f && tt / scale < t)
                        { t = tt / scale; nrm = Normalize(SceneDeviceViews::TransformVector(inst.objectToWorld, nn)); obj = -1; shade = s.shading; }
                    }
                }
                else
                {
                    for (int j = 0; j < inst.primIndexCount; j++)
                    {
                        int triIndex = scene->triPrimIdx[inst.primIndexFirst + j];
                        hrt_mesh_tri tri = scene->meshTris[triIndex];
                        float tt, bu, bv; Float3 nn;
                        if (SceneDeviceViews::IntersectTriangleMT_Bary(iray, scene->meshPositions[tri.i0], scene->meshPositions[tri.i1], scene->meshPositions[tri.i2], tt, nn, bu, bv)
                            && tt > 0.001f && tt / scale < t)
                        { t = tt / scale; nrm = Normalize(SceneDeviceViews::TransformVector(inst.objectToWorld, nn)); obj = triIndex; shade = 0; }
                    }
                }
            }
            hit = t < 1e29f;
        }
        t_out[r] = t; hit_out[r] = hit ? 1 : 0; objid_out[r] = obj; shade_out[r] = shade;
        normal_out[3 * r] = nrm.X; normal_out[3 * r + 1] = nrm.Y; normal_out[3 * r + 2] = nrm.Z;
        albedo_out[3 * r] = alb.X; albedo_out[3 * r + 1] = alb.Y; albedo_out[3 * r + 2] = alb.Z;
    }
    return 0;
}

// ------------------------------------------------------------------ RNG / math probes
void orc_rng_kat(int px, int py, int frame, uint32_t sample, uint32_t salt, int lockNoise, uint32_t out_seed_and_3[4], float out_f[3])
{
    RNG r = RNG::CreateFromPixel(px, py, frame, sample, salt, lockNoise);
    out_seed_and_3[0] = r.state;
    RNG r2 = r;
    for (int i = 0; i < 3; i++) out_seed_and_3[1 + i] = r.NextUInt();
    for (int i = 0; i < 3; i++) out_f[i] = r2.NextFloat();
}
void orc_rng_stream(uint32_t seed, int n, uint32_t* out) { RNG r = RNG::Create(seed); for (int i = 0; i < n; i++) out[i] = r.NextUInt(); }
uint32_t orc_hash3(uint32_t a, uint32_t b, uint32_t c) { return RTRay::Hash3(a, b, c); }
int orc_pack_rgba8(float r, float g, float b) { return GpuFramebuffer::PackRGBA8(Float3(r, g, b)); }

// fn: 0 sin 1 cos 2 tan 3 atan 4 atan2(x,y) 5 acos 6 asin 7 rsqrt 8 sqrt 9 fmin(x,y) 10 fmax(x,y)
// 11 floor 12 round 13 f2i (as float bits of int) 14 1/x 15 x/y
// IntersectAABB (SceneDeviceViews.cs:496-514) on n (ray, box, tMax) tuples: o/d 3 floats per ray (invDir derived as the
// kernels do), lo/hi 3 floats per box.  Lets the tests check properties of the box test itself.
void orc_hit_box(int n, const float* o, const float* d, const float* lo, const float* hi, const float* tmax, int32_t* out)
{
    for (int i = 0; i < n; i++)
    {
        Ray r; r.origin = Float3(o[3 * i], o[3 * i + 1], o[3 * i + 2]); r.dir = Float3(d[3 * i], d[3 * i + 1], d[3 * i + 2]); r.invDir = InvDir(r.dir);
        out[i] = SceneDeviceViews::IntersectAABB(r, Float3(lo[3 * i], lo[3 * i + 1], lo[3 * i + 2]), Float3(hi[3 * i], hi[3 * i + 1], hi[3 * i + 2]), 0.001f, tmax[i]) ? 1 : 0;
    }
}

void orc_math_eval(int fn, int n, const float* x, const float* y, float* out)
{
    for (int i = 0; i < n; i++)
    {
        float a = x[i], b = y ? y[i] : 0.f, r = 0.f;
        switch (fn) {
        case 0: r = hrt_sin(a); break;   case 1: r = hrt_cos(a); break;   case 2: r = hrt_tan(a); break;
        case 3: r = hrt_atan(a); break;  case 4: r = hrt_atan2(a, b); break; case 5: r = hrt_acos(a); break;
        case 6: r = hrt_asin(a); break;  case 7: r = hrt_rsqrt(a); break; case 8: r = hrt_sqrt(a); break;
        case 9: r = hrt_fmin(a, b); break; case 10: r = hrt_fmax(a, b); break;
        case 11: r = hrt_floor(a); break; case 12: r = hrt_round(a); break;
        case 13: { int v = hrt_f2i(a); std::memcpy(&r, &v, 4); } break;
        case 14: r = 1.0f / a; break;    case 15: r = a / b; break;
        case 17: r = hrt_log(a); break;  case 18: r = hrt_exp(a); break;  case 19: r = hrt_pow(a, b); break;
        }
        out[i] = r;
    }
}
// 1 if this build contracts a*b+c into an FMA (must be 0)
int orc_fma_contracted(void)
{
    volatile float a = 1.0f + 1.0f / 4096.0f, b = 1.0f - 1.0f / 4096.0f, c = -1.0f;
    float r = a * b + c;                 // exact product 1 - 2^-24; rounds to 1.0f - 2^-24 (representable) -> differs only with wider accumulate
    volatile float a2 = 1.0f + 1.0f / 8192.0f, b2 = 1.0f + 1.0f / 8192.0f, c2 = -(1.0f + 1.0f / 4096.0f);
    float r2 = a2 * b2 + c2;             // product = 1 + 2^-12 + 2^-26 -> rounds to 1 + 2^-12; fma keeps 2^-26
    (void)r;
    return r2 != 0.0f;
}

// ------------------------------------------------------------------ sort probe (tests)
void orc_dotnet_sort_by_key(int* idx, int n, const float* key)
{
    struct C { const float* k; int operator()(int a, int b) const { return k[a] < k[b] ? -1 : (k[a] > k[b] ? 1 : 0); } } c = {key};
    ArraySort(idx, 0, n, c);
}

// ------------------------------------------------------------------ camera
void orc_camera_create(int w, int h, float fov, hrt_camera* out) { *out = CameraOps::CreateCamera(w, h, fov); }
void orc_camera_lookat(const float* o, const float* l, const float* u, float vfov, float aspect, float focus, hrt_camera* out)
{
    *out = CameraOps::LookAt(Float3(o[0], o[1], o[2]), Float3(l[0], l[1], l[2]), Float3(u[0], u[1], u[2]), vfov, aspect, focus);
}
void orc_camera_translate(hrt_camera* c, const float* d) { CameraOps::Translate(*c, Float3(d[0], d[1], d[2])); }
void orc_camera_bake(hrt_camera* c, int w, int h) { CameraOps::BakeCameraDerived(*c, w, h); }
void orc_sun_dir(float az, float el, float* out) { Float3 s = CameraOps::SunDir(az, el); out[0] = s.X; out[1] = s.Y; out[2] = s.Z; }

// ------------------------------------------------------------------ scene builder
void* orc_scene_new(void) { return new Scene(); }
void orc_scene_free(void* s) { delete static_cast<Scene*>(s); }
void orc_scene_build_default(void* s) { static_cast<Scene*>(s)->BuildDefaultScene(); }
int orc_scene_add_texture(void* s, int w, int h, const hrt_rgba32* px) { return static_cast<Scene*>(s)->AddTexture(w, h, px); }
int orc_scene_add_sphere(void* s, const hrt_sphere* sp) { return static_cast<Scene*>(s)->AddSphere(*sp); }
int orc_scene_build_sphere_instance(void* s_, const int* ids, int n, const hrt_affine3x4* m)
{
    Scene* s = static_cast<Scene*>(s_);
    if (n <= 0) return -1;
    s->_hInstances.push_back(s->BuildSphereInstance(ids, n, *m));
    return (int)s->_hInstances.size() - 1;
}
int orc_scene_load_mesh_instance(void* s_, const hrt_float3* pos, int nPos, const hrt_mesh_tri* tris, int nTris,
                                 const hrt_float2* tex, int nTex, const hrt_mesh_tri_uv* tuv, const int* triMat, int nTriMat,
                                 const hrt_material* mats, int nMats,
                                 const int* texW, const int* texH, const uint8_t* texBGRA, int nTextures,
                                 const hrt_affine3x4* m)
{
    Scene* s = static_cast<Scene*>(s_);
    MeshHost mh;
    mh.Positions.assign(pos, pos + nPos);
    mh.Triangles.assign(tris, tris + nTris);
    mh.Texcoords.assign(tex, tex + nTex);
    mh.TriUVs.assign(tuv, tuv + nTris);
    if (triMat) mh.TriMaterialIndex.assign(triMat, triMat + nTriMat);
    mh.Materials.assign(mats, mats + nMats);
    size_t off = 0;
    for (int i = 0; i < nTextures; i++)
    {
        TextureSrc t; t.Width = texW[i]; t.Height = texH[i];
        size_t bytes = (size_t)t.Width * t.Height * 4;
        t.BGRA.assign(texBGRA + off, texBGRA + off + bytes);
        off += bytes;
        mh.Textures.push_back(t);
    }
    s->LoadMeshInstance(mh, *m);
    return (int)s->_hInstances.size() - 1;
}
void orc_scene_rebuild_tlas(void* s) { static_cast<Scene*>(s)->RebuildTLAS(); }
int orc_scene_set_instance_transform(void* s, int id, const hrt_affine3x4* m)
{
    Scene* sc = static_cast<Scene*>(s);
    if (!m || id < 0 || id >= (int)sc->_hInstances.size()) return -1;
    sc->SetInstanceTransform(id, *m);
    return 0;
}
void orc_scene_get_desc(void* s, hrt_scene_desc* d) { static_cast<Scene*>(s)->GetDesc(*d); }

// ------------------------------------------------------------------ presentation kernels
// mode 0: RTRenderer's non-TAAU branch (blit when sizes match, bilinear upsample otherwise, RTRenderer.cs:225-231)
// mode 1: RTTaa.ResolveUpsample (history arrays are read and written in place)
int orc_present(int mode, const int32_t* lowColor, const int32_t* lowObjId, int inW, int inH, int32_t* outColor, int outW, int outH,
                int32_t* historyColor, int32_t* historyObjId, int isFirstFrame, float feedback, float sharpness, float clampK)
{
    if (!lowColor || !outColor || inW <= 0 || inH <= 0 || outW <= 0 || outH <= 0) return -1;
    const int total = outW * outH;
    if (mode == 1)
    {
        if (!lowObjId || !historyColor || !historyObjId) return -1;
        TaaParams p;
        p.outColor = outColor; p.inColorLow = lowColor; p.inObjIdLow = lowObjId; p.historyColor = historyColor; p.historyObjId = historyObjId;
        p.outW = outW; p.outH = outH; p.inW = inW; p.inH = inH;
        p.feedback = feedback; p.sharpness = sharpness; p.clampK = clampK; p.isFirstFrame = isFirstFrame;
        p.motionScaleX = 0.f; p.motionScaleY = 0.f;
        for (int i = 0; i < total; i++) RTTaa::TaaResolveKernel(i, p);
    }
    else if (inW == outW && inH == outH) { for (int i = 0; i < total; i++) RTPresent::BlitKernel(i, lowColor, (int64_t)inW * inH, outColor, total); }
    else { for (int i = 0; i < total; i++) RTPresent::BilinearUpsampleKernel(i, lowColor, inW, inH, outColor, outW, outH); }
    return 0;
}

int orc_hardware_threads(void) { unsigned n = std::thread::hardware_concurrency(); return n ? (int)n : 1; }

} // extern "C"
