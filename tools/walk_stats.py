"""Phase statistics of the persistent-wave walker (variant build with -DHRT_WALK_STATS):
   make -C ilgpu_raytracing_amd/csrc variant NAME=stats DEFS=-DHRT_WALK_STATS
   HRT_LIB=.../variants/libhip_raytrace_stats.so python tools/walk_stats.py --configs 3,4,5"""
import sys, os, argparse, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilgpu_raytracing_amd import _types as T, scenes, engine

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="3,4,5")
ap.add_argument("--spp", default="2:4,3:16,4:4,5:2")
args = ap.parse_args()
spp = dict((int(a), int(b)) for a, b in (x.split(":") for x in args.spp.split(",")))
L = engine.lib()
L.hrt_debug_walk_stats.argtypes = [C.POINTER(C.c_uint64)]
r = engine.RTRenderer([0])
buf = (C.c_uint64 * 48)()
for cid in [int(c) for c in args.configs.split(",")]:
    cfg = scenes.CONFIGS[cid]
    s = engine.Scene(); scenes.build(cid, s); r.commit(s)
    p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=spp.get(cid, cfg.spp))
    r.render_params(p, None, flags=T.FLAG_STREAMED)
    L.hrt_debug_walk_stats(buf)
    r.render_params(p, None, flags=T.FLAG_STREAMED)
    assert L.hrt_debug_walk_stats(buf) == 0
    for w, name in ((0, "shadow"), (1, "closest")):
        v = [int(buf[w * 24 + i]) for i in range(17)]
        it = max(v[0], 1)
        print("cfg%d %-7s waves %d iterations %d segs %d | per iteration: node steps %.2f (lanes %.1f) TLEAF %.2f (lanes %.1f) BLEAF %.2f (lanes %.1f) retire %.2f (lanes %.1f) idle lanes %.1f"
              % (cid, name, v[11], v[0], v[10], v[1] / it, v[2] / max(v[1], 1), v[3] / it, v[4] / max(v[3], 1), v[5] / it, v[6] / max(v[5], 1),
                 v[7] / it, v[8] / max(v[7], 1), v[9] / it), flush=True)
        tot = max(sum(v[12:17]), 1)
        print("      cycles per iteration %.0f | share: refill %.1f%% node %.1f%% TLEAF %.1f%% BLEAF %.1f%% retire %.1f%% | cycles per node step %.0f, per TLEAF run %.0f, per BLEAF run %.0f, per retire run %.0f"
              % (tot / it, 100 * v[12] / tot, 100 * v[13] / tot, 100 * v[14] / tot, 100 * v[15] / tot, 100 * v[16] / tot,
                 v[13] / max(v[1], 1), v[14] / max(v[3], 1), v[15] / max(v[5], 1), v[16] / max(v[7], 1)), flush=True)
