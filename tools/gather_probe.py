"""Step time of config 2 with the 12 B/pixel gather into (a) pageable, (b) hrt_host_register'ed anonymous, (c) registered /dev/shm arrays."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine, tiling
cfg = scenes.CONFIGS[2]
r = engine.RTRenderer([0])
s = engine.Scene(); scenes.build(2, s); r.commit(s)
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction)
names = ["color", "depth", "objectId"]
def timeit(o, n=20):
    r.render_params(p, o)
    t0 = time.perf_counter()
    for _ in range(n):
        st = r.render_params(p, o)
    return (time.perf_counter() - t0) / n * 1e3, st.d2h_ms
for _ in range(20):
    r.render_params(p, None, flags=T.FLAG_NO_SYNC)
r.synchronize()
print("no gather        %.3f ms" % timeit(None)[0])
a, o = T.alloc_outputs(cfg.width, cfg.height, names=names)
print("pageable         %.3f ms (d2h events %.3f)" % timeit(o))
r.register_host(a)
print("registered anon  %.3f ms (d2h events %.3f)" % timeit(o))
r.unregister_host(a)
fb = tiling.SharedFramebuffer("probe%d" % os.getpid(), cfg.width, cfg.height, names, create=True)
o2 = fb.outputs_struct()
print("shm pageable     %.3f ms (d2h events %.3f)" % timeit(o2))
r.register_host(fb.arrays)
print("shm registered   %.3f ms (d2h events %.3f)" % timeit(o2))
r.unregister_host(fb.arrays)
fb.close()
r.close()
