"""Measurement of the presentation step (hrt_present: TAAU resolve / blit / bilinear upsample) at 4K.
Prints one JSON object; run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations.
   python tools/present_bench.py [--out 3840x2160] [--scale 0.67] [--iters 50]"""
import sys, os, time, json, argparse, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="3840x2160")
ap.add_argument("--scale", type=float, default=0.67)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
ow, oh = [int(v) for v in a.out.split("x")]
r = engine.RTRenderer([0])
s = engine.Scene(); scenes.build(2, s); r.commit(s)
res = {}
for name, scale, taau in (("taau_resolve_upsample", a.scale, True), ("bilinear_upsample", a.scale, False), ("blit", 1.0, False)):
    iw = max(1, int(np.rint(np.float32(ow) * np.float32(scale)))); ih = max(1, int(np.rint(np.float32(oh) * np.float32(scale))))
    cfg = scenes.CONFIGS[2]
    p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, width=iw, height=ih, spp=1)
    r.render_params(p, None)
    pp = T.PresentParams(ow, oh, T.PRESENT_TAAU if taau else T.PRESENT_RESAMPLE, 0.0, 0.0, 0.0)
    L = engine.lib()
    for _ in range(3):
        r._check(L.hrt_present(r._ctx, C.byref(pp), None))
    t0 = time.perf_counter()
    for _ in range(a.iters):
        r._check(L.hrt_present(r._ctx, C.byref(pp), None))              # blocking: kernel + one stream synchronise
    ms = (time.perf_counter() - t0) / a.iters * 1e3
    # algorithmic bytes per call: every low-res colour (and objId for TAAU) read once, display colour written once,
    # TAAU history colour + objId read and written once per display pixel (RTTaa.cs:117-171)
    lo, hi = iw * ih, ow * oh
    bytes_ = (lo * 8 + hi * (4 + 8 + 8)) if taau else (lo * 4 + hi * 4)
    res[name] = {"in": "%dx%d" % (iw, ih), "out": "%dx%d" % (ow, oh), "ms_per_call_incl_sync": round(ms, 4), "algorithmic_bytes": bytes_,
                 "GBps_incl_sync": round(bytes_ / ms / 1e6, 1), "hbm_peak_GBps": 8000}
print(json.dumps(res))
