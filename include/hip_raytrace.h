/*
 * hip_raytrace.h -- C ABI of libhip_raytrace.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of NullandKale/ILGPU_Raytracing: the two ILGPU kernel
 * launches inside RTRenderer.RenderDirectToPbo and the device buffers they touch.
 * Plain pointers and sizes only; no C++ or torch types.  The reference-side binding a
 * maintainer adds (C# [DllImport]) is shown in INTEGRATION.md.
 *
 *   entry point            replaces in the reference (ILGPU_Raytracing/Engine/...)
 *   ---------------------  -----------------------------------------------------------
 *   hrt_create             Context.Create + CreateCudaAccelerator + DefaultStream,
 *                          LoadAutoGroupedStreamKernel x2        RTRenderer.cs:66-68,85-86
 *   hrt_scene_upload       Scene.UploadAll (15 Allocate1D H2D copies, empty -> 1 zeroed
 *                          element)                              Scene.cs:258-279,370-377
 *                          reached through SceneManager.Commit / BvhManager.BuildOrRefit
 *                                                                SceneManager.cs:23, BvhManager.cs:27
 *   hrt_render_frame       GBuffer.EnsureLength / Framebuffer.EnsureLength /
 *                          EnsureLowResBuffers                   RTRenderer.cs:118-120,265-279
 *                          _primaryKernel(Index1D(inLen), gp)    RTRenderer.cs:152-153
 *                          Framebuffer.GetReservoirPair(0,frame) RTRenderer.cs:164, Framebuffer.cs:127-146
 *                          _integratorKernel(Index1D(inLen), ip, SpecializedValue(maxDepth))
 *                                                                RTRenderer.cs:181-205
 *                          _cuda.Synchronize()                   RTRenderer.cs:233
 *                          Framebuffer.DownloadToCpu (the unused read-back hook)
 *                                                                Framebuffer.cs:148-160
 *   hrt_present            pbo.MapCuda + RTTaa.ResolveUpsample | BlitKernel | BilinearUpsampleKernel
 *                                                                RTRenderer.cs:208-231,281-345; RTTaa.cs:34-171
 *   hrt_synchronize        _cuda.Synchronize() when frames were enqueued without it  RTRenderer.cs:233
 *   hrt_device_buffers     GpuFramebuffer / GpuGBuffer views handed to the post kernels
 *                          (TAAU, blit) without leaving the device  RTRenderer.cs:155-161,208-231
 *   hrt_reset_history      Framebuffer.EnsureLength re-allocation on resize (fresh
 *                          reservoirs)                           Framebuffer.cs:60-97
 *   hrt_destroy            RTRenderer.Dispose                    RTRenderer.cs:347-363
 *   hrt_last_error         the exception message of CudaException / Argument*Exception
 *
 * Behavioural contract kept from the reference:
 *   - frame parity picks the reservoir pair: even frame -> prev = B, cur = A
 *     (Framebuffer.cs:132-145); reservoirs are zero-initialised when (re)allocated
 *     (the reference leaves them uninitialised; zero makes frame 0 well defined:
 *      m == 0 rejects every import, RTRay.cs:417).
 *   - per-pixel buffers are re-allocated only when width*height changes.
 *   - RNG and ReSTIR hashing key on the GLOBAL pixel index, so tiling never changes a
 *     pixel's value (RTUtils.cs:108-113, RTRay.cs:488).
 *   - hrt_render_frame blocks until the frame is complete; one call at a time per ctx.
 *
 * Multi-GPU: a ctx created on n devices cuts the image into 8-row strips and deals them
 * round-robin to its devices (device slot i renders strips s with s % n == i: sky rows are
 * cheap, geometry rows expensive, interleaving balances the tiles without knowing the image).
 * The scene is replicated, there is no collective; every device's strips are gathered into the
 * caller's host framebuffer by strided hipMemcpy2DAsync copies issued from one host thread per
 * device.  Page-lock the framebuffer once with hrt_host_register and those copies are
 * asynchronous DMA.  One-process-per-GPU hosts instead create one ctx per process and restrict
 * it to a row range / strip set with hrt_render_opts.  With ReSTIR reuse enabled a multi-device ctx
 * exchanges G-buffer and reservoir tiles between its devices (device-to-device copies); partial tiles of a
 * one-process-per-GPU host are refused while reuse is on.
 *
 * Status codes: 0 = ok, negative = error (message via hrt_last_error).
 */
#ifndef HIP_RAYTRACE_H
#define HIP_RAYTRACE_H

#include "hrt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hrt_ctx hrt_ctx;

enum hrt_status {
    HRT_OK                = 0,
    HRT_ERR_INVALID_ARG   = -1,   /* ArgumentNullException / ArgumentOutOfRangeException */
    HRT_ERR_INVALID_STATE = -2,   /* InvalidOperationException (e.g. render before upload) */
    HRT_ERR_HIP           = -3,   /* CudaException analogue: a HIP runtime call failed     */
    HRT_ERR_NO_DEVICE     = -4,
    HRT_ERR_OUT_OF_MEMORY = -5
};

enum hrt_render_flags {
    HRT_FLAG_COUNTERS     = 1u << 0,  /* run the counting build of both kernels, fill hrt_stats.k[] */
    HRT_FLAG_SKIP_PRIMARY = 1u << 1,  /* bench/profiling only: reuse the resident G-buffer          */
    HRT_FLAG_REFERENCE_LAYOUT = 1u << 2, /* walk the reference's own arrays (TracerRef) instead of the device-private
                                         repack: identical results, A/B baseline and fallback for scenes beyond the
                                         packed encoding (leaf count > 15)                                      */
    HRT_FLAG_MEGAKERNEL   = 1u << 4,  /* force the path-trace launch to run as ONE one-pixel-per-lane kernel      */
    HRT_FLAG_STREAMED     = 1u << 5,  /* force the streamed init/shade/walk/finish/resolve pipeline.  Neither flag:
                                         the library picks by scene size (tiny BVH -> fused).  Identical results.  */
    HRT_FLAG_NO_SYNC      = 1u << 3,  /* enqueue only (outputs must be NULL); collect with hrt_synchronize.
                                         Up to 128 frames may be in flight; the 129th call drains first.      */
    /* One process per GPU with ReSTIR reuse ON: a tile needs the current G-buffer (worldPos, normalWS, objId) and the
     * previous reservoirs of the WHOLE image (RTRay.cs:339-374, 488-515).  The host then renders a frame in two calls
     * with an all-gather after each (ilgpu_raytracing_amd/tiling.py, RCCL through torch.distributed):
     *   1. HRT_FLAG_PRIMARY_ONLY            launch 1 on this tile           -> all-gather the 3 G-buffer arrays
     *   2. HRT_FLAG_SKIP_PRIMARY|EXCHANGED  path-trace launch on this tile  -> all-gather the 7 arrays of resCur
     * through the device pointers of hrt_device_buffers.  EXCHANGED is the caller's statement that both all-gathers of
     * the protocol are done; without it a reuse frame on a partial tile is refused (HRT_ERR_INVALID_STATE).        */
    HRT_FLAG_PRIMARY_ONLY = 1u << 6,
    HRT_FLAG_EXCHANGED    = 1u << 7,
    HRT_FLAG_TREELETS     = 1u << 8   /* streamed production frames of scenes with big triangle meshes (>= 4096 BLAS nodes): walk them with the
                                         LDS-staged, treelet-queued walker (csrc/hrt_walker_tl.hpp) instead of the persistent-wave walker.
                                         Identical results; measured SLOWER on every BASELINE config (profiles/r03_treelet_walker.md),
                                         hence opt-in.  Ignored where no treelets exist (sphere scenes, small meshes, after a vertex update).  */
};

/* Host destinations of one frame; any pointer may be NULL (not copied).  Arrays hold
 * width*height elements in the reference's layout (row 0 = bottom row, RTRay.cs:122).
 * With a row range only rows [row_begin,row_end) of each array are written. */
typedef struct hrt_outputs {
    /* GpuFramebuffer, RTRay.cs:51-56 */
    int32_t*    color;        /* packed 0xFFRRGGBB                              */
    float*      depth;
    int32_t*    objectId;
    int32_t*    cameraId;     /* 1 element                                      */
    /* pre-pack Lout (RTRay.cs:323), 3 floats per pixel: the 1e-4 parity target */
    hrt_float3* radiance;
    /* GpuGBuffer, RTRay.cs:80-87 */
    hrt_float3* gb_worldPos;
    hrt_float3* gb_normalWS;
    hrt_float3* gb_baseColor;
    int32_t*    gb_matId;
    int32_t*    gb_objId;
    int32_t*    gb_hitMask;
    /* resCur of this frame, GpuReservoirSoA RTRay.cs:23-31 */
    hrt_float3* res_L;
    hrt_float3* res_wi;
    float*      res_pdf;
    float*      res_w;
    float*      res_wSum;
    int32_t*    res_m;
    int32_t*    res_lightId;
} hrt_outputs;

typedef struct hrt_render_opts {
    uint32_t flags;           /* hrt_render_flags                                   */
    int32_t  row_begin;       /* rows [row_begin,row_end) of the global image;      */
    int32_t  row_end;         /* 0,0 = all rows                                     */
    int32_t  strip_n;         /* > 1: the range is cut into 8-row strips and this   */
    int32_t  strip_i;         /* call renders strips s with s % strip_n == strip_i  */
                              /* (load-balanced tiling for one-process-per-GPU hosts) */
} hrt_render_opts;

/* Device-resident views of the current frame on device slot `dev` (for on-device
 * consumers such as the TAAU/blit kernels or a torch tensor wrapper).  Pointers stay
 * valid until the next hrt_render_frame with a different size, or hrt_destroy.
 * Every array spans the whole image (global pixel index); only the strips this device
 * owns (rows [row_begin,row_end), strips s % strip_n == strip_i) hold this frame. */
typedef struct hrt_device_views {
    int32_t row_begin, row_end, strip_n, strip_i, width, height, device_id, reserved;
    void *color, *depth, *objectId, *radiance;
    void *gb_worldPos, *gb_normalWS, *gb_baseColor, *gb_matId, *gb_objId, *gb_hitMask;
    void *present_color;                      /* display-size RGBA8 of the last hrt_present (NULL before) */
    int32_t present_width, present_height;
    /* reservoir sets A and B (GpuReservoirSoA, RTRay.cs:23-48): L, wi (float3), pdf, w, wSum (float), m, lightId (int).
     * Frame f writes resCur = A and reads resPrev = B when f is even, the other way round when odd (Framebuffer.cs:132-145) */
    void *res_a[7], *res_b[7];
} hrt_device_views;

/* Presentation step of RenderDirectToPbo after the two launches (RTRenderer.cs:208-231): the last frame
 * rendered at (width,height) = (inW,inH) is resolved to the display size. */
enum hrt_present_mode {
    HRT_PRESENT_RESAMPLE = 0,   /* _enableTAAU == false: BlitKernel when sizes match, else BilinearUpsampleKernel
                                   (RTRenderer.cs:225-231,281-345)                                              */
    HRT_PRESENT_TAAU     = 1    /* RTTaa.ResolveUpsample (RTTaa.cs:49-171): history kept per display size,
                                   first frame after (re)allocation or hrt_reset_history ignores it             */
};
typedef struct hrt_present_params {
    int32_t out_width, out_height;
    int32_t mode;                              /* hrt_present_mode */
    float feedback, sharpness, clampK;         /* TAAU tunables; <= 0 selects the reference's 0.075 / 0.10 / 1.25 (RTTaa.cs:77-79) */
} hrt_present_params;

int  hrt_create(const int* device_ids, int n_dev, hrt_ctx** out);
void hrt_destroy(hrt_ctx* ctx);
const char* hrt_last_error(hrt_ctx* ctx);      /* ctx may be NULL: last error of hrt_create */

int  hrt_scene_upload(hrt_ctx* ctx, const hrt_scene_desc* scene);

/* ---- moving instances of the committed scene: BvhManager.BuildOrRefit(scene, policy) (BvhManager.cs:13-27).
 * The reference declares the policy and ignores it: Commit re-runs RebuildTLAS on the host (Scene.cs:358-368)
 * and re-uploads all fifteen arrays (Scene.cs:258-279).  Here the scene stays on the device:
 *   - objectToWorld of instance instance_ids[k] becomes objectToWorld[k]; worldToObject, uniformScale and the
 *     world bounds are derived from it exactly as Scene.cs does (InvertRigidOrUniform :616-638, TransformAABB
 *     :560-580 of the box of the instance's BLAS root node); instance_ids must be distinct;
 *   - HRT_REBUILD_FORCE_REFIT keeps the TLAS topology and recomputes every box bottom-up;
 *   - HRT_REBUILD_FORCE_REBUILD builds a new TLAS over ALL instances (as RebuildTLAS does) with a Morton-order
 *     LBVH, leaves of <= 2 instances;
 *   - HRT_REBUILD_AUTO rebuilds a tree that was uploaded (once: the device-built tree costs about as much as a refit
 *     and walks faster than the reference's median split), afterwards refits, and rebuilds again if the boxes of the
 *     refitted tree have grown, in the geometric mean
 *     over all nodes, to more than 1.5 x the surface area they had when the tree was last built (uploaded or
 *     rebuilt): a measure neither one far-flung instance nor one huge instance dominates.
 * n may be 0 (re-derive / rebuild only).  Blocking; every device of the context is updated.  The TLAS of a scene
 * updated this way is numbered in walk order; hrt_scene_download_tlas returns it in the reference's layout.
 * A different tree visits the same primitives in another order: results change only where two primitives are
 * hit at bit-equal distance (DESIGN.md "inner nodes only accelerate"). */
enum hrt_rebuild_policy { HRT_REBUILD_AUTO = 0, HRT_REBUILD_FORCE_REFIT = 1, HRT_REBUILD_FORCE_REBUILD = 2,
                          HRT_REBUILD_BLAS = 16 /* flag for hrt_scene_update_positions, see there */ };

typedef struct hrt_bvh_update_stats {
    int32_t action;           /* HRT_REBUILD_FORCE_REFIT or HRT_REBUILD_FORCE_REBUILD: what was done          */
    int32_t tlas_nodes;       /* nodes of the TLAS now in use                                                  */
    int32_t tlas_slots;       /* entries of its instance index list                                            */
    int32_t general_instances;/* 1 if a leaf slot holds an instance that needs the ray transform               */
    float   growth_refit;     /* after the refit: geometric mean over the nodes of area / area at the last build (0: no refit) */
    float   growth_final;     /* the same for the tree now in use (1 after a rebuild)                          */
    float   sah_cost;         /* of the tree now in use: sum(area x (leaf ? count : 1)) / area(root)            */
    float   device_ms;        /* HIP-event time of the device work on device slot 0 (tree in use; the library's second tree
                                 over many one-sphere instances is refitted after it, about as long again)    */
    int32_t blas_action;      /* hrt_scene_update_positions: what happened to the mesh BLASes (0 none, refit, rebuild) */
    float   blas_growth;      /* ... and the growth of their node boxes after the refit (0: not measured)       */
} hrt_bvh_update_stats;

int  hrt_scene_update_instances(hrt_ctx* ctx, const int32_t* instance_ids, int32_t n, const hrt_affine3x4* objectToWorld,
                                int32_t policy, hrt_bvh_update_stats* stats /* may be NULL */);

/* ---- deforming meshes: meshPositions[first_vertex, first_vertex + n) := positions (n may be 0).
 * The reference has no counterpart (its Commit rebuilds every BLAS on the host, Scene.cs:405-467, and re-uploads): here
 * every triangle-mesh BLAS keeps its topology and gets its triangle records and boxes recomputed bottom-up on the device
 * (BoundsOfTriangle over the items of each node, Scene.cs:423-429,597-605), the world bounds of the mesh instances are
 * re-derived from the new root boxes (TransformAABB, Scene.cs:560-580), and the TLAS is refitted / rebuilt per `policy`
 * as in hrt_scene_update_instances; with HRT_REBUILD_AUTO the mesh BLASes are rebuilt too when their boxes have grown, in the
 * geometric mean, to more than 1.5 x their area at the last build.  policy | HRT_REBUILD_BLAS first gives every triangle-mesh BLAS a new topology for the
 * new positions: a Morton-order LBVH over its triangles with leaves of <= 4 like the reference's BLAS, built on the device
 * into the node range and the leaf region of triPrimIdx the mesh already owns (blasNodeCount of its instance shrinks to
 * the new node count).  Another BLAS visits triangles in another order: pixels where two triangles are hit at bit-equal
 * distance (shared edges) may change, exactly as they would under a different host builder.
 * Sphere BLASes are untouched.  Blocking; every device of the context is updated. */
int  hrt_scene_update_positions(hrt_ctx* ctx, int64_t first_vertex, int64_t n, const hrt_float3* positions,
                                int32_t policy, hrt_bvh_update_stats* stats /* may be NULL */);

/* ---- moving / resizing / recolouring spheres: spheres[first, first + n) := spheres (n may be 0).  Every sphere-set BLAS
 * keeps its topology and gets its boxes recomputed bottom-up on the device (centre -+ radius per sphere of a leaf,
 * Scene.cs:331-336,386-390), the world bounds of the sphere-set instances are re-derived from the new root boxes, and the
 * TLAS is refitted / rebuilt per `policy` as in hrt_scene_update_instances.  Single-sphere instances with an identity
 * transform stay on the walkers' fast path (a moved INSTANCE would leave it: its ray transform must then be evaluated).
 * Note: the boxes are the unions of the spheres each node really holds; the reference's own builder indexes its
 * pre-computed bounds by array position (Scene.cs:386-395,413-419), which for more than four spheres per instance
 * yields other (wrong) boxes on a rebuild.  Blocking; every device of the context is updated. */
int  hrt_scene_update_spheres(hrt_ctx* ctx, int64_t first_sphere, int64_t n, const hrt_sphere* spheres,
                              int32_t policy, hrt_bvh_update_stats* stats /* may be NULL */);

/* Copies scene array `array` (0..14, in the order of hrt_scene_desc / SceneDeviceViews.cs:13-27) as it is now on device
 * slot `dev` back to the host: what TracerRef walks after updates.  *count (may be NULL) receives the element count;
 * dst may be NULL to query it; cap = capacity of dst in elements. */
int  hrt_scene_download_array(hrt_ctx* ctx, int dev, int array, void* dst, int64_t cap, int64_t* count);

/* Copies the TLAS in use on device slot `dev`, in the reference's layout, and the instance records to the host
 * (any pointer may be NULL).  counts[3] (may be NULL) receives the element counts {tlasNodes, tlasInstanceIndices,
 * instances}; an array is copied only if its capacity (in elements) is large enough, else HRT_ERR_INVALID_ARG. */
int  hrt_scene_download_tlas(hrt_ctx* ctx, int dev, hrt_bvh_node* tlasNodes, int64_t cap_nodes,
                             int32_t* tlasInstanceIndices, int64_t cap_indices, hrt_instance* instances, int64_t cap_instances,
                             int64_t* counts);

int  hrt_render_frame(hrt_ctx* ctx, const hrt_frame_params* params,
                      const hrt_render_opts* opts,      /* may be NULL */
                      const hrt_outputs* outputs,       /* may be NULL: leave results on device */
                      hrt_stats* stats);                /* may be NULL */

/* Waits for every frame enqueued with HRT_FLAG_NO_SYNC.  stats (may be NULL): kernel_ms[]
 * = per-launch HIP-event time summed over those frames (max over devices), frames = their
 * number.  A blocking hrt_render_frame is enqueue + hrt_synchronize. */
int  hrt_synchronize(hrt_ctx* ctx, hrt_stats* stats);

/* out_color_host: out_width*out_height packed 0xFFRRGGBB, may be NULL (result stays in hrt_device_views.present_color).
 * The last frame must have been a full-image render.  The resolve runs on device slot 0; a multi-device ctx first
 * brings the colour / objectId strips of its other devices there (device-to-device copies). */
int  hrt_present(hrt_ctx* ctx, const hrt_present_params* params, int32_t* out_color_host);

/* Per-frame HIP-event times (ms) of the frames the last hrt_synchronize (or blocking hrt_render_frame) collected on device
 * slot `dev`: launch 0 = primary visibility, 1 = the path-trace stage.  *n (may be NULL) receives their number; at most
 * cap values are written to ms (may be NULL). */
int  hrt_frame_times(hrt_ctx* ctx, int dev, int launch, float* ms, int cap, int* n);

/* Page-locks (hipHostRegister, portable) a host range the caller will pass as hrt_outputs destination, e.g. the C# host's
 * pinned framebuffer arrays (GCHandle.Alloc(..., Pinned)): gathers into it are asynchronous DMA instead of staged copies.
 * The range must stay mapped until hrt_host_unregister / hrt_destroy.  Replaces nothing in the reference, whose frame never
 * leaves the GPU (CudaGlInteropIndexBuffer.cs:37-194). */
int  hrt_host_register(hrt_ctx* ctx, void* ptr, int64_t bytes);
int  hrt_host_unregister(hrt_ctx* ctx, void* ptr);

/* Caps a path-state workspace of the streamed pipeline at max_resident_paths paths (324 bytes each; 0 = the default of
 * 2^25 paths = 10.9 GB): frames whose width*height*spp exceeds it run in several sample batches, with identical results.
 * Two sample batches are in flight at a time, each in a workspace of its own, so up to twice that much memory is held.
 * Counterpart of sizing MemoryBuffer1D allocations in the reference (Framebuffer.cs:60-97). */
int  hrt_set_workspace_limit(hrt_ctx* ctx, int64_t max_resident_paths);

int  hrt_device_buffers(hrt_ctx* ctx, int dev, hrt_device_views* out);
int  hrt_reset_history(hrt_ctx* ctx);           /* zero both reservoir sets */

/* Test hooks (math probes, host-side builders of derived trees) are declared in hrt_test_hooks.h and exist only in
 * libhip_raytrace_test.so, the -DHRT_TEST_HOOKS build of the same sources; the shipped library exports none of them. */
int  hrt_device_count(void);                    /* visible HIP devices, <0 on error */
const char* hrt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HIP_RAYTRACE_H */
