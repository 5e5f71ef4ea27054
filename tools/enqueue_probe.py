import sys, os, time
sys.path.insert(0, os.getcwd())
from ilgpu_raytracing_amd import _types as T, scenes, engine
r = engine.RTRenderer([0]); s = engine.Scene(); scenes.build(2, s); r.commit(s)
cfg = scenes.CONFIGS[2]
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction)
for n in (1, 2, 4, 8):
    r.render_params(p, None, strips=(n, 0)); 
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(100): r.render_params(p, None, flags=T.FLAG_NO_SYNC, strips=(n, 0))
        t1 = time.perf_counter(); st = r.synchronize(); t2 = time.perf_counter()
    print("strips 1/%d: host enqueue %.1f us/frame, total %.3f ms/frame, kernels %.3f ms/frame" % (n, (t1 - t0) / 100 * 1e6, (t2 - t0) / 100 * 1e3, (st.kernel_ms[0] + st.kernel_ms[1]) / st.frames))
