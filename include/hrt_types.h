/*
 * hrt_types.h -- wire structs of the render hot path (C, blittable, 4-byte aligned).
 *
 * These are the scene / camera / reservoir records the reference passes to its two
 * kernels by value or through ArrayView<T>.  Field order, names and sizes follow the
 * reference so a C# host can hand its own arrays over with
 * [StructLayout(LayoutKind.Sequential)] and no marshalling.
 *
 * Reference definitions (all under ILGPU_Raytracing/Engine/):
 *   Float3          Float3.cs:6-10            Float2/MeshTriUV  MeshLoaderOBJ.cs:33-34
 *   Affine3x4       Affine3x4.cs:3-7          MaterialRecord    MeshLoaderOBJ.cs:44-63
 *   Sphere          Sphere.cs:3-15            TLASNode/BLASNode Scene.cs:705-714,730-739
 *   InstanceRecord  Scene.cs:716-728          MeshTri/RGBA32/TexInfo Scene.cs:741-745
 *   Camera          Camera.cs:5-17            Reservoir         RTRay.cs:171-179
 *   Ray             RTUtils.cs:6-10
 *
 * Shared by: the HIP kernels, the C-ABI (hip_raytrace.h), the host-side scene
 * builder and the CPU oracle.  C and C++ clean.
 */
#ifndef HRT_TYPES_H
#define HRT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
#define HRT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define HRT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

typedef struct hrt_float3 { float X, Y, Z; } hrt_float3;              /* 12 B */
typedef struct hrt_float2 { float X, Y; } hrt_float2;                 /*  8 B */

/* row-major 3x4 affine; Affine3x4.cs:3-7 */
typedef struct hrt_affine3x4 {
    float m00, m01, m02, m03;
    float m10, m11, m12, m13;
    float m20, m21, m22, m23;
} hrt_affine3x4;                                                      /* 48 B */

/* MeshLoaderOBJ.cs:44-63.  44 bytes (the reference's "48B" comment counts running END
 * offsets; there is no trailing pad). */
typedef struct hrt_material {
    hrt_float3 Kd;
    int32_t HasDiffuseMap;
    int32_t DiffuseTexIndex;
    int32_t Shading;
    float   IOR;
    int32_t HasAlphaMap;
    int32_t AlphaTexIndex;
    int32_t TwoSided;
    float   AlphaCutoff;
} hrt_material;                                                       /* 44 B */

enum { HRT_SHADING_LAMBERT = 0, HRT_SHADING_MIRROR = 1, HRT_SHADING_GLASS = 2 };

/* Sphere.cs:3-15 */
typedef struct hrt_sphere {
    hrt_float3   center;
    float        radius;
    hrt_float3   albedo;
    hrt_material material;
    int32_t      shading;
    float        ior;
} hrt_sphere;                                                         /* 80 B */

/* Scene.cs:705-714 (TLASNode) == Scene.cs:730-739 (BLASNode).
 * Leaf iff count > 0; inner nodes have first = -1, count = 0; -1 terminates a walk. */
typedef struct hrt_bvh_node {
    hrt_float3 boundsMin;
    hrt_float3 boundsMax;
    int32_t left;
    int32_t right;
    int32_t first;
    int32_t count;
    int32_t skipIndex;
} hrt_bvh_node;                                                       /* 44 B */

enum { HRT_BLAS_SPHERESET = 1, HRT_BLAS_TRIMESH = 2 };                /* Scene.cs:703 */

/* Scene.cs:716-728 */
typedef struct hrt_instance {
    int32_t type;
    int32_t blasRoot;
    int32_t blasNodeCount;
    int32_t primIndexFirst;
    int32_t primIndexCount;
    hrt_affine3x4 objectToWorld;
    hrt_affine3x4 worldToObject;
    float uniformScale;
    hrt_float3 worldBoundsMin;
    hrt_float3 worldBoundsMax;
} hrt_instance;                                                       /* 144 B */

typedef struct hrt_mesh_tri { int32_t i0, i1, i2; } hrt_mesh_tri;     /* 12 B */
typedef struct hrt_mesh_tri_uv { int32_t t0, t1, t2; } hrt_mesh_tri_uv;
typedef struct hrt_rgba32 { uint8_t R, G, B, A; } hrt_rgba32;         /*  4 B */
typedef struct hrt_tex_info { int32_t Offset, Width, Height; } hrt_tex_info;

/* Camera.cs:5-17 */
typedef struct hrt_camera {
    hrt_float3 origin;
    hrt_float3 lowerLeft;
    hrt_float3 horizontal;
    hrt_float3 vertical;
    hrt_float3 forward;
    hrt_float3 right;
    hrt_float3 up;
    float aspect;
    float fovYRadians;
} hrt_camera;                                                         /* 92 B */

/* RTRay.cs:171-179 (stored SoA on the device, RTRay.cs:23-48) */
typedef struct hrt_reservoir {
    hrt_float3 L, wi;
    float pdf, w, wSum;
    int32_t m, lightId;
} hrt_reservoir;                                                      /* 44 B */

HRT_STATIC_ASSERT(sizeof(hrt_float3) == 12, "Float3");
HRT_STATIC_ASSERT(sizeof(hrt_float2) == 8, "Float2");
HRT_STATIC_ASSERT(sizeof(hrt_affine3x4) == 48, "Affine3x4");
HRT_STATIC_ASSERT(sizeof(hrt_material) == 44, "MaterialRecord");
HRT_STATIC_ASSERT(sizeof(hrt_sphere) == 80, "Sphere");
HRT_STATIC_ASSERT(sizeof(hrt_bvh_node) == 44, "TLASNode/BLASNode");
HRT_STATIC_ASSERT(sizeof(hrt_instance) == 144, "InstanceRecord");
HRT_STATIC_ASSERT(sizeof(hrt_mesh_tri) == 12, "MeshTri");
HRT_STATIC_ASSERT(sizeof(hrt_mesh_tri_uv) == 12, "MeshTriUV");
HRT_STATIC_ASSERT(sizeof(hrt_rgba32) == 4, "RGBA32");
HRT_STATIC_ASSERT(sizeof(hrt_tex_info) == 12, "TexInfo");
HRT_STATIC_ASSERT(sizeof(hrt_camera) == 92, "Camera");
HRT_STATIC_ASSERT(sizeof(hrt_reservoir) == 44, "Reservoir");

/*
 * The 15 scene arrays, in the order of the fields of SceneDeviceViews
 * (SceneDeviceViews.cs:13-27).  Each is (host pointer, element count).  A count of 0
 * is uploaded as ONE zeroed element, as Scene.AllocateOrEmpty does (Scene.cs:370-377),
 * so e.g. texInfos.Length reads 1 on the device when the scene has no textures.
 */
typedef struct hrt_scene_desc {
    const hrt_bvh_node*    tlasNodes;           int64_t n_tlasNodes;
    const int32_t*         tlasInstanceIndices; int64_t n_tlasInstanceIndices;
    const hrt_instance*    instances;           int64_t n_instances;
    const hrt_bvh_node*    blasNodes;           int64_t n_blasNodes;
    const int32_t*         spherePrimIdx;       int64_t n_spherePrimIdx;
    const hrt_sphere*      spheres;             int64_t n_spheres;
    const int32_t*         triPrimIdx;          int64_t n_triPrimIdx;
    const hrt_float3*      meshPositions;       int64_t n_meshPositions;
    const hrt_mesh_tri*    meshTris;            int64_t n_meshTris;
    const hrt_float2*      meshTexcoords;       int64_t n_meshTexcoords;
    const hrt_mesh_tri_uv* meshTriUVs;          int64_t n_meshTriUVs;
    const int32_t*         triMatIndex;         int64_t n_triMatIndex;
    const hrt_material*    materials;           int64_t n_materials;
    const hrt_rgba32*      texels;              int64_t n_texels;
    const hrt_tex_info*    texInfos;            int64_t n_texInfos;
} hrt_scene_desc;

/*
 * Per-frame parameters = the scalar fields of GBufferParams / IntegratorParams
 * (RTRay.cs:112-146) plus MaxDepth (RTRenderer.cs:204-205).
 */
typedef struct hrt_frame_params {
    int32_t width, height, frame;
    hrt_camera cam;
    hrt_camera prevCam;
    hrt_float3 dirLightDir, dirLightRadiance;
    hrt_float3 skyTintTop, skyTintBottom;
    int32_t debugCamSeq;
    int32_t enableTemporalReuse, enableSpatialReuse, rngLockNoise;
    int32_t spp;
    int32_t maxDepth;
} hrt_frame_params;

/*
 * Work counters of one frame.  Deterministic functions of (scene, params): the oracle
 * and the HIP kernels (stats build) must report identical values.  The algorithmic
 * byte count of DESIGN.md is computed from these.
 *   index 0 = primary-visibility launch, 1 = path-trace launch.
 */
typedef struct hrt_kernel_counters {
    uint64_t rays_closest;      /* TraceClosest calls                               */
    uint64_t rays_shadow;       /* ShadowOcclusion calls                            */
    uint64_t node_visits;       /* TLAS + BLAS nodes fetched                        */
    uint64_t leaf_instances;    /* instance records fetched in TLAS leaves          */
    uint64_t sphere_tests;      /* IntersectSphere calls                            */
    uint64_t tri_tests;         /* IntersectTriangleMT_Bary calls                   */
    uint64_t tri_mt_hits;       /* ... that returned true (material fetched)        */
    uint64_t tri_accepted;      /* ... that passed the t-compare (UVs fetched)      */
    uint64_t reuse_imports;     /* ImportFromPrevReservoir calls with a valid index */
    uint64_t diffuse_vertices;  /* ReSTIR_Direct calls                              */
} hrt_kernel_counters;

typedef struct hrt_stats {
    hrt_kernel_counters k[2];
    double kernel_ms[2];        /* hipEvent time of each launch (max over devices)  */
    double d2h_ms;              /* gather of the requested outputs                   */
    int32_t n_devices;
    int32_t counters_valid;     /* 1 when the frame(s) ran with HRT_FLAG_COUNTERS    */
    int32_t frames;             /* frames covered by kernel_ms / d2h_ms (sums)       */
    int32_t reserved;
} hrt_stats;

#endif /* HRT_TYPES_H */
