"""The C++ oracle (oracle/orc_kernels.hpp, what the HIP path is compared with) against a SECOND restatement of the reference's two
kernels written independently from the C# (oracle/orc_indep.py: scalar numpy.float32 in the reference's statement order).  Every
output array of both launches must agree bit for bit.  This does not pin the oracle to the reference binary (nothing can),
but a transcription slip in either restatement fails here."""
import os

import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, scenes
from oracle import orc_indep
from tests import helpers as H


def _math(orc):
    def f(name, x, y=None):
        return orc.math_eval(name, np.array([x], np.float32), None if y is None else np.array([y], np.float32))[0]
    return f


def _mixed_scene(b):
    """Ground + 40 spheres with mirror / glass / Lambert shading, one instance each, plus a sphere inside a glass sphere."""
    scenes.build_random_spheres(b, 40, seed=0xBADC0DE, extent=3.0)


CASES = {
    "config1": (scenes.build_config1, scenes.CONFIGS[1], 24, 24, 1),
    "config2": (scenes.build_config2, scenes.CONFIGS[2], 24, 14, 2),
    "config2_depth5_roulette": (scenes.build_config2, scenes.Config("rr", 0, 0, 0, (0.0, 1.5, 5.5), (0.0, 1.2, 0.0), max_depth=5, extra=scenes.CONFIGS[2].extra), 16, 10, 2),
    "mixed_41_instances": (_mixed_scene, scenes.Config("mix", 0, 0, 0, (0.0, 2.2, 7.5), (0.0, 0.6, 0.0)), 22, 14, 2),
    "locked_noise_frame7": (scenes.build_config2, scenes.CONFIGS[2], 12, 8, 1),
    # textured sphere (atan2 / acos), textured two-sided alpha cut-out triangles (point / linear band), scaled instance, mirror, glass
    "textured_cutout": (scenes.build_textured_test_scene, scenes.Config("t", 0, 0, 0, (0.3, 1.3, 4.2), (0.0, 0.7, 0.0)), 30, 22, 2),
    "textured_cutout_close": (scenes.build_textured_test_scene, scenes.Config("t2", 0, 0, 0, (0.1, 1.1, 1.6), (0.0, 1.0, -1.0), vfov=35.0), 24, 20, 2),
    # rotated + uniformly scaled instances of a triangle grid, a multi-sphere BLAS and a glass sphere
    "rotated_instances": (scenes.build_rotated_instances_scene, scenes.Config("r", 0, 0, 0, (0.4, 1.8, 5.0), (0.0, 0.8, 0.0), max_depth=4), 28, 20, 2),
}


@pytest.mark.parametrize("name", list(CASES))
def test_second_restatement_agrees_with_the_oracle(orc, name):
    builder, cfg, w, h, spp = CASES[name]
    lock, frame = (987654321, 7) if name.startswith("locked") else (0, 0)
    ref, _, p = H.oracle_frame(orc, builder, cfg, w, h, spp, frame=frame, lock=lock, nthreads=1)
    so = orc.OrcScene()
    builder(so)
    got = orc_indep.render(so.arrays(), p, _math(orc))
    bad = _differences(got, ref)
    assert not bad, "the two restatements differ {array: elements}: %s" % bad
    assert int(got["gb_hitMask"].sum()) > 0 and np.any(got["res_m"] > 0)
    if name == "textured_cutout_close":         # the case is there for these branches: make sure it takes them
        assert all(v > 0 for v in got["_cover"].values()), got["_cover"]


def _differences(got, ref):
    bad = {}
    for k, a in got.items():
        if k.startswith("_"):
            continue
        n = int(np.count_nonzero(~H.bits_equal(a.reshape(ref[k].shape), ref[k])))
        if n:
            bad[k] = n
    return bad


@pytest.mark.parametrize("moving", [False, True])
def test_second_restatement_reuse_frames(orc, moving):
    """Temporal + spatial ReSTIR reuse over three frames with ping-pong reservoirs (Framebuffer.cs:132-145), static camera and a camera
    that moves (prevCam != cam: ReprojectToPrevPixel lands on other pixels or outside)."""
    builder = scenes.build_textured_test_scene
    w, h, spp = 26, 18, 2
    so = orc.OrcScene()
    builder(so)
    arrs = so.arrays()
    A, B = H.new_reservoirs(w, h), H.new_reservoirs(w, h)
    prev_cam, imported = None, 0
    for f in range(3):
        origin = (0.3 + 0.2 * f, 1.3 + 0.05 * f, 4.2 - 0.15 * f) if moving else (0.3, 1.3, 4.2)
        cfg = scenes.Config("t", w, h, spp, origin, (0.0, 0.7, 0.0))
        prev, cur = (B, A) if f % 2 == 0 else (A, B)
        mine = {k: a.copy() for k, a in cur.items()}          # resCur keeps stale entries where no sample has a diffuse vertex
        ref, st, p = H.oracle_frame(orc, builder, cfg, w, h, spp, frame=f, reuse=True, prev=prev, cur=cur, prev_cam=prev_cam, nthreads=1)
        got = orc_indep.render(arrs, p, _math(orc), prev=prev, cur=mine)
        bad = _differences(got, ref)
        assert not bad, "frame %d: the two restatements differ {array: elements}: %s" % (f, bad)
        imported += st.k[1].reuse_imports
        prev_cam = T.Camera.from_buffer_copy(p.cam)
    assert imported > w * h                     # reuse really ran


@pytest.mark.parametrize("hostile", [False, True])
@pytest.mark.parametrize("case", range(int(os.environ.get("HRT_INDEP_CASES", "0")) or 14))
def test_second_restatement_on_random_scenes(orc, case, hostile):
    """The scene / frame recipes of the GPU differential fuzz (tests/test_fuzz_gpu.py) at postage-stamp size: both restatements of the
    reference must agree on every array, first frame and (where the recipe has reuse on) the second.  hostile: the recipe with
    non-finite / zero / negative / huge numbers planted in it (NaN boxes, inverted boxes, NaN rays: what each comparison, min and
    max does with them is restated twice as well)."""
    from tests import test_fuzz_gpu as F
    seed = int(os.environ.get("HRT_INDEP_SEED", "0"), 0) or 0x0DD50000
    ops, fr = F._scene_recipe(seed + case)
    if hostile:
        ops, fr = F._poison(ops, fr, seed + case)
    so = orc.OrcScene()
    F._apply(so, ops)
    arrs = so.arrays()
    w, h = 14, 10
    cfg = scenes.Config("fz", w, h, min(fr["spp"], 2), fr["origin"], fr["lookat"], max_depth=min(fr["max_depth"], 4), vfov=fr["vfov"],
                        extra={"sun_azimuth": fr["sun"][0], "sun_elevation": fr["sun"][1]})
    A, B = H.new_reservoirs(w, h), H.new_reservoirs(w, h)
    for f in range(2 if fr["reuse"] else 1):
        frame = fr["frame"] + f
        prev, cur = (B, A) if (frame & 1) == 0 else (A, B)
        mine = {k: a.copy() for k, a in cur.items()}
        ref, _, p = H.oracle_frame(orc, lambda b: F._apply(b, ops), cfg, w, h, cfg.spp, frame=frame, reuse=fr["reuse"], lock=fr["lock"], prev=prev, cur=cur, nthreads=1)
        got = orc_indep.render(arrs, p, _math(orc), prev=prev, cur=mine)
        bad = _differences(got, ref)
        assert not bad, "frame %d: the two restatements differ {array: elements}: %s" % (f, bad)
