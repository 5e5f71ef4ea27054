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
    for name in CASES:
        arrs, st = render_case(orc, name)
        out = {k: v for k, v in arrs.items()}
        out["counters_json"] = np.array(json.dumps([st.k[0].as_dict(), st.k[1].as_dict()]))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
