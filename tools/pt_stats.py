"""Section statistics of the fused path-trace kernel (variant build with -DHRT_PT_STATS):
   make -C ilgpu_raytracing_amd/csrc variant NAME=ptstats DEFS=-DHRT_PT_STATS
   HRT_LIB=.../variants/libhip_raytrace_ptstats.so python tools/pt_stats.py [--config 2]"""
import sys, os, argparse, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilgpu_raytracing_amd import _types as T, scenes, engine

NAMES = {0: "bounce-loop iteration", 1: "mirror vertex", 2: "glass vertex", 3: "diffuse vertex (candidates)", 4: "shadow site", 5: "closest site",
         6: "sample end", 7: "closest: instance box test", 8: "closest: sphere test", 9: "sphere: disc >= 0 (sqrt + root)", 10: "sphere: second root",
         11: "shadow: instance box test", 12: "shadow: sphere test", 13: "finish_hit", 14: "closest: leaf box test", 15: "shadow: leaf box test"}
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
args = ap.parse_args()
L = engine.lib()
L.hrt_debug_pt_stats.argtypes = [C.POINTER(C.c_uint64)]
r = engine.RTRenderer([0])
cfg = scenes.CONFIGS[args.config]
s = engine.Scene(); scenes.build(args.config, s); r.commit(s)
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction)
buf = (C.c_uint64 * 64)()
r.render_params(p, None, flags=T.FLAG_MEGAKERNEL)
L.hrt_debug_pt_stats(buf)
r.render_params(p, None, flags=T.FLAG_MEGAKERNEL)
assert L.hrt_debug_pt_stats(buf) == 0
for i in sorted(NAMES):
    n, lanes = int(buf[2 * i]), int(buf[2 * i + 1])
    if n:
        print("%-34s wave-executions %10d  live lanes %5.1f / 64  (%.0f%%)" % (NAMES[i], n, lanes / n, 100.0 * lanes / n / 64), flush=True)
