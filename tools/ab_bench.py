"""A/B timing of the kernels over the BASELINE configs (reduced spp for the big ones).
   python tools/ab_bench.py [--configs 2,3,4,5] [--ref] [--frames 5]"""
import sys, os, time, argparse, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="2,3,4,5")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--ref", action="store_true", help="also time the reference-layout tracer")
ap.add_argument("--spp", default="2:4,3:16,4:4,5:2")
ap.add_argument("--check", action="store_true", help="compare packed vs reference-layout outputs")
ap.add_argument("--modes", default="auto,stream,mega", help="subset of auto,stream,mega")
args = ap.parse_args()
spp = dict((int(a), int(b)) for a, b in (x.split(":") for x in args.spp.split(",")))
r = engine.RTRenderer([0])
for cid in [int(c) for c in args.configs.split(",")]:
    cfg = scenes.CONFIGS[cid]
    t = time.time(); s = engine.Scene(); scenes.build(cid, s); tb = time.time() - t
    t = time.time(); r.commit(s); tu = time.time() - t
    p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=spp.get(cid, cfg.spp))
    st = r.render_params(p, None, flags=T.FLAG_COUNTERS)
    rays = sum(st.k[i].rays_closest + st.k[i].rays_shadow for i in range(2))
    line = "cfg%d %dx%d spp%d rays %.1fM build %.2fs upload %.2fs |" % (cid, p.width, p.height, p.spp, rays / 1e6, tb, tu)
    modes = [m for m in [("auto", 0), ("stream", T.FLAG_STREAMED), ("mega", T.FLAG_MEGAKERNEL)] if m[0] in args.modes.split(",")] + ([("mega+reflayout", T.FLAG_REFERENCE_LAYOUT | T.FLAG_MEGAKERNEL), ("stream+reflayout", T.FLAG_REFERENCE_LAYOUT | T.FLAG_STREAMED)] if args.ref else [])
    for name, fl in modes:
        r.render_params(p, None, flags=fl)
        for _ in range(args.frames):
            r.render_params(p, None, flags=fl | T.FLAG_NO_SYNC)
        s2 = r.synchronize()
        k0, k1 = s2.kernel_ms[0] / s2.frames, s2.kernel_ms[1] / s2.frames
        line += " %s: prim %.3f path %.3f ms = %.1f Mrays/s |" % (name, k0, k1, rays / (k0 + k1) / 1e3)
    print(line, flush=True)
    if args.check:
        names = ["color", "depth", "objectId", "radiance", "gb_worldPos", "gb_normalWS", "gb_baseColor", "gb_matId", "gb_objId", "gb_hitMask", "res_L", "res_m"]
        a, oa = T.alloc_outputs(p.width, p.height, names); b, ob = T.alloc_outputs(p.width, p.height, names)
        r.reset_history(); r.render_params(p, oa); r.reset_history(); r.render_params(p, ob, flags=T.FLAG_REFERENCE_LAYOUT | T.FLAG_MEGAKERNEL)
        bad = {k: int(np.count_nonzero(~((a[k] == b[k]) | ((a[k] != a[k]) & (b[k] != b[k]))))) for k in a}
        print("   stream/packed vs mega/reference-layout mismatches:", {k: v for k, v in bad.items() if v} or "none", flush=True)
