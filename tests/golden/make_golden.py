"""Generates the golden vectors under tests/golden/ with the CPU oracle.

The reference ships no fixtures for the render path and cannot be run here (C#/.NET 8 +
CUDA-only ILGPU), so these vectors come from this repo's own restatement:
they freeze its behaviour (regression guard) and let the GPU box compare the HIP path with
committed data.  Each fixture = seeded inputs (scene built by name, frame params) + every
output array of the two kernels + the work counters.

    python -m tests.golden.make_golden        # rewrites tests/golden/*.npz
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ilgpu_raytracing_amd import scenes  # noqa: E402
from tests import helpers as H  # noqa: E402

# name -> (scene builder, Config, width, height, spp, frames with ReSTIR reuse)
CASES = {
    "config1_64x64": (scenes.build_config1, scenes.CONFIGS[1], 64, 64, 1, 0),
    "config2_96x54": (scenes.build_config2, scenes.CONFIGS[2], 96, 54, 2, 0),
    "default_scene_80x48": (lambda b: b.build_default_scene(), scenes.Config("default", 80, 48, 2, (0.0, 1.4, 4.5), (0.0, 0.5, 0.0)), 80, 48, 2, 0),
    "textured_72x56_reuse3": (scenes.build_textured_test_scene, scenes.Config("tex", 72, 56, 2, (0.3, 1.3, 4.2), (0.0, 0.7, 0.0)), 72, 56, 2, 3),
}


# BASELINE.json configs[2..4] at their STATED size and sample count: one 8-row strip each (strip index in units of 8 rows,
# chosen through the geometry), every output array of the strip's rows + the work counters of the strip.
# name -> (config id, first row)
FULL_STRIPS = {
    "full_config3_1080p_16spp_rows304": (3, 304),
    "full_config4_4k_64spp_rows1040": (4, 1040),
    "full_config5_4k_256spp_rows800": (5, 800),
}
FULL_NAMES = ["color", "depth", "objectId", "radiance", "gb_hitMask", "gb_worldPos", "gb_normalWS", "gb_objId",
              "res_L", "res_wi", "res_pdf", "res_w", "res_wSum", "res_m", "res_lightId"]


def render_full_strip(orc, name):
    cfg_id, y0 = FULL_STRIPS[name]
    cfg = scenes.CONFIGS[cfg_id]
    builder = {3: scenes.build_config3, 4: scenes.build_config4, 5: scenes.build_config5}[cfg_id]
    w, h = cfg.width, cfg.height
    arrs, st, _ = H.oracle_frame(orc, builder, cfg, w, h, cfg.spp, rows=(y0, y0 + 8))
    out = {}
    for k in FULL_NAMES:
        a = arrs[k]
        out[k] = np.ascontiguousarray(a.reshape(h, w, *a.shape[1:])[y0:y0 + 8])
    return out, st


def render_case(orc, name):
    builder, cfg, w, h, spp, reuse_frames = CASES[name]
    if reuse_frames == 0:
        arrs, st, _ = H.oracle_frame(orc, builder, cfg, w, h, spp)
        return arrs, st
    # ReSTIR temporal + spatial reuse over frames 0..n-1 with the A/B ping-pong of Framebuffer.GetReservoirPair
    A, B = H.new_reservoirs(w, h), H.new_reservoirs(w, h)
    arrs = st = None
    for f in range(reuse_frames):
        prev, cur = (B, A) if f % 2 == 0 else (A, B)
        arrs, st, _ = H.oracle_frame(orc, builder, cfg, w, h, spp, frame=f, reuse=True, prev=prev, cur=cur)
    return arrs, st


def main():
    from oracle import orc
    orc.build()
    only = sys.argv[1:]
    import time
    for name in FULL_STRIPS:
        if only and name not in only:
            continue
        t0 = time.time()
        out, st = render_full_strip(orc, name)
        out["counters_json"] = np.array(json.dumps([st.k[0].as_dict(), st.k[1].as_dict()]))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes, oracle %.1f s" % (time.time() - t0), flush=True)
    for name in CASES:
        if only and name not in only:
            continue
        arrs, st = render_case(orc, name)
        out = {k: v for k, v in arrs.items()}
        out["counters_json"] = np.array(json.dumps([st.k[0].as_dict(), st.k[1].as_dict()]))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
