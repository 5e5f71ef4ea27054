#!/usr/bin/env python3
"""Moving SPHERES (hrt_scene_update_spheres) in a scene of N one-sphere instances: what a frame costs before the move, after a refit
and after a device rebuild of the tree in use -- i.e. whether the second tree (hrt_runtime.hip, refit_second_tree) follows the scene.
   python tools/sphere_update_bench.py [--counts 10000,100000] [--spp 4]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine

ap = argparse.ArgumentParser()
ap.add_argument("--counts", default="10000,100000")
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--spp", type=int, default=4)
args = ap.parse_args()
r = engine.RTRenderer([0])
cfg = scenes.CONFIGS[3]
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=args.spp)


def frame_ms():
    r.render_params(p, None)
    for _ in range(args.frames):
        r.render_params(p, None, flags=T.FLAG_NO_SYNC)
    st = r.synchronize()
    return (st.kernel_ms[0] + st.kernel_ms[1]) / st.frames


for n in [int(x) for x in args.counts.split(",")]:
    s = engine.Scene(); scenes.build_random_spheres(s, n, extent=20.0 * (n / 10000.0) ** 0.5)
    r.commit(s)
    static = frame_ms()
    sp = s.arrays()["spheres"].copy()
    rng = np.random.default_rng(3)
    out = {}
    for name, policy in (("refit", T.REBUILD_FORCE_REFIT), ("rebuild", T.REBUILD_FORCE_REBUILD)):
        for f in "XZ":
            sp["center"][f][1:] += rng.uniform(-0.2, 0.2, len(sp) - 1).astype(np.float32)
        t = time.perf_counter(); st = r.update_spheres(1, sp[1:], policy); wall = (time.perf_counter() - t) * 1e3
        out[name] = (frame_ms(), wall, st.device_ms)
    print("instances %7d, %d spp: frame %.2f ms as uploaded | after moving every sphere: refit %.2f ms (update %.2f ms wall, %.2f ms tree in use) | device rebuild %.2f ms (update %.2f ms wall)"
          % (n + 1, args.spp, static, out["refit"][0], out["refit"][1], out["refit"][2], out["rebuild"][0], out["rebuild"][1]), flush=True)
