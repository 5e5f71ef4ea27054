// HipRaytrace.cs -- P/Invoke surface of libhip_raytrace.so for the reference's C# host (drop into Engine/).
// Mirrors include/hip_raytrace.h and include/hrt_host.h one to one; the Engine structs (Float3, Affine3x4, Sphere,
// MaterialRecord, TLASNode, BLASNode, InstanceRecord, MeshTri, MeshTriUV, Float2, RGBA32, TexInfo, Camera) are already
// sequential blittable value types with the layouts of include/hrt_types.h, so they cross the boundary unchanged.
// Not compiled in this repository (no .NET toolchain in the build image); the same entry points are exercised through
// the Python ctypes mirror (ilgpu_raytracing_amd/engine.py) by the test-suite.
using System;
using System.Runtime.InteropServices;

namespace ILGPU_Raytracing.Engine
{
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HrtSceneDesc            // hrt_scene_desc: 15 x (pointer, Int64 count), order of SceneDeviceViews.cs:13-27
    {
        public TLASNode* tlasNodes;            public long n_tlasNodes;
        public int* tlasInstanceIndices;       public long n_tlasInstanceIndices;
        public InstanceRecord* instances;      public long n_instances;
        public BLASNode* blasNodes;            public long n_blasNodes;
        public int* spherePrimIdx;             public long n_spherePrimIdx;
        public Sphere* spheres;                public long n_spheres;
        public int* triPrimIdx;                public long n_triPrimIdx;
        public Float3* meshPositions;          public long n_meshPositions;
        public MeshTri* meshTris;              public long n_meshTris;
        public Float2* meshTexcoords;          public long n_meshTexcoords;
        public MeshTriUV* meshTriUVs;          public long n_meshTriUVs;
        public int* triMatIndex;               public long n_triMatIndex;
        public MaterialRecord* materials;      public long n_materials;
        public RGBA32* texels;                 public long n_texels;
        public TexInfo* texInfos;              public long n_texInfos;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct HrtFrameParams                 // hrt_frame_params: scalar fields of GBufferParams / IntegratorParams + maxDepth
    {
        public int width, height, frame;
        public Camera cam, prevCam;
        public Float3 dirLightDir, dirLightRadiance, skyTintTop, skyTintBottom;
        public int debugCamSeq, enableTemporalReuse, enableSpatialReuse, rngLockNoise, spp, maxDepth;
    }

    [Flags]
    public enum HrtFlags : uint
    {
        None = 0, Counters = 1, SkipPrimary = 2, ReferenceLayout = 4, NoSync = 8, Megakernel = 16, Streamed = 32,
        PrimaryOnly = 64, Exchanged = 128,
        Treelets = 256      // opt-in LDS-staged treelet walker: same pictures, slower than the default (DESIGN.md 5.4)
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct HrtRenderOpts { public uint flags; public int row_begin, row_end, strip_n, strip_i; }

    [StructLayout(LayoutKind.Sequential)]
    public struct HrtPresentParams { public int out_width, out_height, mode; public float feedback, sharpness, clampK; }   // mode 0 blit / bilinear, 1 TAAU

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HrtOutputs               // host destinations of one frame, any may be null
    {
        public int* color; public float* depth; public int* objectId; public int* cameraId;
        public Float3* radiance;
        public Float3* gb_worldPos, gb_normalWS, gb_baseColor; public int* gb_matId, gb_objId, gb_hitMask;
        public Float3* res_L, res_wi; public float* res_pdf, res_w, res_wSum; public int* res_m, res_lightId;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct HrtStats
    {
        public fixed ulong counters[20];          // 2 x hrt_kernel_counters (launch 1, launch 2)
        public fixed double kernel_ms[2]; public double d2h_ms;
        public int n_devices, counters_valid, frames, reserved;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct HrtBvhUpdateStats
    {
        public int action, tlas_nodes, tlas_slots, general_instances;
        public float growth_refit, growth_final, sah_cost, device_ms;
        public int blas_action; public float blas_growth;
    }

    internal static unsafe class HipRaytrace
    {
        const string Lib = "hip_raytrace";        // libhip_raytrace.so

        [DllImport(Lib)] public static extern int hrt_create(int* deviceIds, int nDev, out IntPtr ctx);
        [DllImport(Lib)] public static extern void hrt_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr hrt_last_error(IntPtr ctx);
        [DllImport(Lib)] public static extern int hrt_scene_upload(IntPtr ctx, HrtSceneDesc* scene);
        // BvhManager.BuildOrRefit(scene, policy) for moved instances: policy = (int)RebuildPolicy (Auto 0, ForceRefit 1, ForceRebuild 2);
        // hrt_scene_update_positions also takes policy | HRT_REBUILD_BLAS (16): new BLAS topology for every mesh
        public const int HRT_REBUILD_BLAS = 16;
        [DllImport(Lib)] public static extern int hrt_scene_update_instances(IntPtr ctx, int* instanceIds, int n, Affine3x4* objectToWorld, int policy, HrtBvhUpdateStats* stats);
        [DllImport(Lib)] public static extern int hrt_scene_update_positions(IntPtr ctx, long firstVertex, long n, Float3* positions, int policy, HrtBvhUpdateStats* stats);
        [DllImport(Lib)] public static extern int hrt_scene_update_spheres(IntPtr ctx, long firstSphere, long n, Sphere* spheres, int policy, HrtBvhUpdateStats* stats);
        [DllImport(Lib)] public static extern int hrt_scene_download_array(IntPtr ctx, int dev, int array, void* dst, long cap, long* count);
        [DllImport(Lib)] public static extern int hrt_scene_download_tlas(IntPtr ctx, int dev, TLASNode* nodes, long capNodes, int* indices, long capIndices, InstanceRecord* instances, long capInstances, long* counts);
        [DllImport(Lib)] public static extern int hrt_render_frame(IntPtr ctx, HrtFrameParams* p, HrtRenderOpts* opts, HrtOutputs* outputs, HrtStats* stats);
        [DllImport(Lib)] public static extern int hrt_present(IntPtr ctx, HrtPresentParams* p, int* outColorHost);
        [DllImport(Lib)] public static extern int hrt_synchronize(IntPtr ctx, HrtStats* stats);
        [DllImport(Lib)] public static extern int hrt_reset_history(IntPtr ctx);
        // page-lock the managed framebuffer arrays once (GCHandle.Alloc(array, GCHandleType.Pinned).AddrOfPinnedObject()): gathers
        // into them become asynchronous DMA; keep the handles alive until hrt_host_unregister / hrt_destroy
        [DllImport(Lib)] public static extern int hrt_host_register(IntPtr ctx, void* ptr, long bytes);
        [DllImport(Lib)] public static extern int hrt_host_unregister(IntPtr ctx, void* ptr);
        [DllImport(Lib)] public static extern int hrt_frame_times(IntPtr ctx, int dev, int launch, float* ms, int cap, int* n);
        [DllImport(Lib)] public static extern int hrt_set_workspace_limit(IntPtr ctx, long maxResidentPaths);
        [DllImport(Lib)] public static extern int hrt_device_count();

        // native asset loader (optional: a C# host may keep MeshLoaderOBJ)
        [DllImport(Lib)] public static extern IntPtr hrth_scene_new();
        [DllImport(Lib)] public static extern void hrth_scene_free(IntPtr scene);
        [DllImport(Lib)] public static extern int hrth_scene_load_obj_instance(IntPtr scene, [MarshalAs(UnmanagedType.LPUTF8Str)] string objPath, Affine3x4* objectToWorld, float uniformScale);
        [DllImport(Lib)] public static extern void hrth_scene_get_desc(IntPtr scene, HrtSceneDesc* desc);
        [DllImport(Lib)] public static extern IntPtr hrth_last_error();

        /// <summary>Turns a negative hrt_status into the exception type the reference's own classes throw.</summary>
        public static void Check(IntPtr ctx, int rc)
        {
            if (rc >= 0) return;
            string msg = Marshal.PtrToStringAnsi(hrt_last_error(ctx)) ?? "hip_raytrace error";
            throw rc switch
            {
                -1 => new ArgumentException(msg),              // HRT_ERR_INVALID_ARG   (ArgumentNull/OutOfRange in Framebuffer.cs:62,101-102)
                -2 => new InvalidOperationException(msg),      // HRT_ERR_INVALID_STATE
                -5 => new OutOfMemoryException(msg),           // HRT_ERR_OUT_OF_MEMORY
                _ => new Exception(msg),                       // HRT_ERR_HIP / HRT_ERR_NO_DEVICE: CudaException analogue
            };
        }
    }
}
