// HipFrameRenderer.cs -- what RTRenderer's device side becomes when the two ILGPU kernel launches and the presentation
// kernels are replaced by libhip_raytrace.so.  RTRenderer keeps its camera controller, sun animation, TAAU switch and
// render scale and calls Render() where it used to launch _primaryKernel / _integratorKernel / _taa (RTRenderer.cs:105-237);
// Scene.UploadAll calls Upload() with its pinned host lists.  Shipped as source; see INTEGRATION.md.
using System;
using System.Runtime.InteropServices;

namespace ILGPU_Raytracing.Engine
{
    public sealed unsafe class HipFrameRenderer : IDisposable
    {
        private IntPtr _ctx;
        private int[] _display = Array.Empty<int>();      // RGBA8 display image (what the PBO received)

        /// <param name="deviceIds">one id: one GPU; several ids: one frame row-tiled over the GPUs of the node.</param>
        public HipFrameRenderer(params int[] deviceIds)
        {
            if (deviceIds == null || deviceIds.Length == 0) deviceIds = new[] { 0 };
            fixed (int* ids = deviceIds)
                HipRaytrace.Check(IntPtr.Zero, HipRaytrace.hrt_create(ids, deviceIds.Length, out _ctx));
        }

        /// <summary>Scene.UploadAll: the 15 host lists, pinned by the caller for the duration of the call (empty list: null, 0).</summary>
        public void Upload(in HrtSceneDesc scene)
        {
            fixed (HrtSceneDesc* p = &scene)
                HipRaytrace.Check(_ctx, HipRaytrace.hrt_scene_upload(_ctx, p));
        }

        /// <summary>One RenderDirectToPbo: both launches at (inW, inH), then TAAU or blit/bilinear to (outW, outH).
        /// Returns the display image (row 0 = bottom row, 0xAARRGGBB), valid until the next call.</summary>
        public ReadOnlySpan<int> Render(in HrtFrameParams frame, int outW, int outH, bool taau)
        {
            fixed (HrtFrameParams* fp = &frame)
                HipRaytrace.Check(_ctx, HipRaytrace.hrt_render_frame(_ctx, fp, null, null, null));      // blocking, results stay on the device
            if (_display.Length != outW * outH) _display = new int[outW * outH];
            var pp = new HrtPresentParams { out_width = outW, out_height = outH, mode = taau ? 1 : 0 };   // tunables <= 0: the reference's 0.075 / 0.10 / 1.25
            fixed (int* dst = _display)
                HipRaytrace.Check(_ctx, HipRaytrace.hrt_present(_ctx, &pp, dst));
            return _display;
        }

        /// <summary>Framebuffer.EnsureLength / RTTaa.Ensure side effect the host may want explicitly (camera cut).</summary>
        public void ResetHistory() => HipRaytrace.Check(_ctx, HipRaytrace.hrt_reset_history(_ctx));

        public void Dispose()
        {
            if (_ctx != IntPtr.Zero) { HipRaytrace.hrt_destroy(_ctx); _ctx = IntPtr.Zero; }
        }
    }
}
