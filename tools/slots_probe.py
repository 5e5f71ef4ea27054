"""Frames of a config through one context over 1, 2, 3, 4 device slots that all name GPU 0 (row strips dealt round-robin over the slots,
each with its own stream and buffers): does one slot's launch tail overlap the others' work?   python tools/slots_probe.py [--config 2] [--frames 60]"""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilgpu_raytracing_amd import _types as T, scenes, engine
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2); ap.add_argument("--frames", type=int, default=60); ap.add_argument("--spp", type=int, default=0)
a = ap.parse_args()
cfg = scenes.CONFIGS[a.config]
for n in (1, 2, 3, 4, 1, 2):
    r = engine.RTRenderer([0] * n)
    s = engine.Scene(); scenes.build(a.config, s); r.commit(s)
    p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, spp=a.spp or cfg.spp)
    st = r.render_params(p, None, flags=T.FLAG_COUNTERS)
    rays = sum(st.k[i].rays_closest + st.k[i].rays_shadow for i in range(2))
    for _ in range(5): r.render_params(p, None, flags=T.FLAG_NO_SYNC)
    r.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.frames): r.render_params(p, None, flags=T.FLAG_NO_SYNC)
    r.synchronize()
    dt = (time.perf_counter() - t0) / a.frames
    print("slots %d: %.3f ms per frame (wall, %d frames enqueued back to back) = %.0f Mrays/s" % (n, dt * 1e3, a.frames, rays / dt / 1e6), flush=True)
    r.close()
