"""Soak: 1500 scene updates of every kind and policy in a row; device memory must not grow after the first hundred."""
import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ilgpu_raytracing_amd import _types as T, scenes, engine
r = engine.RTRenderer([0])
s = engine.Scene(); scenes.build_random_spheres(s, 3000, extent=10.0); r.commit(s)
cfg = scenes.CONFIGS[3]
p = scenes.frame_params(cfg, engine.camera_look_at, engine.bake_camera_derived, engine.sun_direction, width=320, height=180, spp=1)
n = 3000
ids = np.arange(1, n + 1, dtype=np.int32)
rng = np.random.default_rng(1)
sp = s.arrays()["spheres"].copy()
free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
for it in range(1500):
    xf = np.zeros((n, 12), np.float32); xf[:, 0] = xf[:, 5] = xf[:, 10] = 1.0
    xf[:, [3, 7, 11]] = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    pol = [T.REBUILD_AUTO, T.REBUILD_FORCE_REFIT, T.REBUILD_FORCE_REBUILD][it % 3]
    if it % 2 == 0:
        r.update_instances(ids, xf, pol)
    else:
        sp2 = sp.copy(); sp2["center"]["X"][1:] += xf[:, 3]; r.update_spheres(1, sp2[1:], pol)
    if it % 50 == 0:
        r.render_params(p, None)
    if it == 100:
        free0 = torch.cuda.mem_get_info()[0]
free1 = torch.cuda.mem_get_info()[0]
print("1500 updates in %.1f s, device memory growth after the first hundred: %.1f MB" % (time.time() - t0, (free0 - free1) / 1e6))
assert free0 - free1 < 8e6
