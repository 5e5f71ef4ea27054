"""Kernel time of config 2 when one rank renders every N-th 8-row strip (the N-GPU tiling of bench.py, measured on one GPU):
   N x time(N) / time(1) is the strong-scaling loss of the kernels alone.   python tools/tile_scaling.py [--frames 30]"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilgpu_raytracing_amd import _types as T, scenes, engine
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=30); ap.add_argument("--config", type=int, default=2)
a = ap.parse_args()
r = engine.RTRenderer([0]); cfg = scenes.CONFIGS[a.config]
s = engine.Scene(); scenes.build(a.config, s); r.commit(s)
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction)
base = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for k in range(n):
        r.render_params(p, None, strips=(n, k))
        for _ in range(a.frames):
            r.render_params(p, None, flags=T.FLAG_NO_SYNC, strips=(n, k))
        st = r.synchronize()
        worst = max(worst, (st.kernel_ms[0] + st.kernel_ms[1]) / st.frames)
    base = base or worst
    print("N=%d  slowest rank %.3f ms per frame  -> %.0f %% of linear" % (n, worst, 100.0 * base / (n * worst)), flush=True)
