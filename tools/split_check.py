import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from ilgpu_raytracing_amd import _types as T, scenes, engine
from oracle import orc
from tests import helpers as H
r = engine.RTRenderer([0])
for builder, cfg in ((scenes.build_config2, scenes.CONFIGS[2]), (lambda b: b.build_default_scene(), scenes.Config("d", 0, 0, 0, (0.0, 1.4, 4.5), (0.0, 0.5, 0.0)))):
    for spp in (7, 5, 2):
        w, h = 100, 61
        ref, ost, _ = H.oracle_frame(orc, builder, cfg, w, h, spp)
        s = engine.Scene(); builder(s); r.commit(s); r.reset_history()
        p = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)
        arrs, o = T.alloc_outputs(w, h)
        r.render_params(p, o, flags=T.FLAG_MEGAKERNEL)
        H.assert_outputs_equal(ref, arrs)
print("SPLIT_OK", os.environ.get("HRT_SPLIT"))
