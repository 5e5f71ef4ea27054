"""HRT_FLAG_TREELETS: the LDS-staged, treelet-queued walker (csrc/hrt_walker_tl.hpp) must render what the oracle renders.

Small scenes reach the code through the test hook that lowers the limits of the treelet cut (hooks build); the BASELINE
meshes (configs 4 and 5, full size, reduced spp) run on the shipped library with the shipped limits, against the plain
persistent-wave walker of the same library (which tests/test_full_size_gpu.py pins to the oracle)."""
import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, engine, scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = {
    "blob_mesh_64x64_quads": (lambda b: scenes.build_config4(b, 64, 64), scenes.CONFIGS[4], 256, 144, 2, (4096, 7, 64)),
    "terrain_96": (lambda b: scenes.build_config5(b, 96), scenes.CONFIGS[5], 256, 144, 2, (6000, 15, 64)),
    "terrain_96_tiny_treelets": (lambda b: scenes.build_config5(b, 96), scenes.CONFIGS[5], 192, 108, 3, (700, 3, 16)),
    "textured_alpha_scaled": (scenes.build_textured_test_scene, scenes.Config("t", 0, 0, 0, (0.3, 1.3, 4.2), (0.0, 0.7, 0.0)), 192, 144, 3, (1024, 3, 8)),
    "rotated_scaled_instances": (scenes.build_rotated_instances_scene, scenes.Config("rot", 0, 0, 0, (0.4, 1.6, 4.6), (0.0, 0.8, 0.0)), 176, 112, 2, (1024, 3, 8)),
    "depth5_blob": (lambda b: scenes.build_config4(b, 48, 48), scenes.Config("d5", 0, 0, 0, (0.0, 1.6, 3.6), (0.0, 1.05, 0.0), max_depth=5), 160, 90, 2, (2048, 7, 32)),
}


@pytest.fixture()
def low_limits(hooks_lib):
    yield hooks_lib
    hooks_lib.hrt_debug_set_treelet_limits(0, 0, 0)


@pytest.mark.parametrize("name", list(CASES))
def test_treelet_walker_matches_oracle(orc, hooks_renderer, low_limits, name):
    builder, cfg, w, h, spp, limits = CASES[name]
    ref, _, _ = H.oracle_frame(orc, builder, cfg, w, h, spp)
    low_limits.hrt_debug_set_treelet_limits(*limits)
    s = engine.Scene()
    builder(s)
    r = hooks_renderer
    r.commit(s)
    r.reset_history()
    n_tl = low_limits.hrt_debug_treelet_count(r._ctx)
    if name != "textured_alpha_scaled" and name != "rotated_scaled_instances":
        assert n_tl >= 4, "the scene got no treelets: the test would not reach the walker"
    p = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)
    got, o = T.alloc_outputs(w, h)
    st = r.render_params(p, o, flags=T.FLAG_STREAMED | T.FLAG_TREELETS)
    assert st.counters_valid == 0
    H.assert_outputs_equal(ref, got)


def test_treelets_under_reuse_and_after_a_vertex_update(orc, hooks_renderer, low_limits):
    """Two reuse frames through the treelet walker == the plain walker; a vertex update drops the treelets (the reduced tree holds
    copies of the boxes as uploaded) and the flag falls back to the plain walker."""
    builder, cfg, w, h, spp = (lambda b: scenes.build_config4(b, 48, 48)), scenes.CONFIGS[4], 192, 108, 2
    low_limits.hrt_debug_set_treelet_limits(2048, 7, 32)
    r = hooks_renderer
    s = engine.Scene(); builder(s); r.commit(s)
    assert low_limits.hrt_debug_treelet_count(r._ctx) > 0
    frames = {}
    for label, fl in (("plain", T.FLAG_STREAMED), ("treelets", T.FLAG_STREAMED | T.FLAG_TREELETS)):
        r.reset_history()
        outs = []
        for f in range(2):
            p = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp, frame=f, reuse=True)
            a, o = T.alloc_outputs(w, h)
            r.render_params(p, o, flags=fl)
            outs.append(a)
        frames[label] = outs
    for f in range(2):
        H.assert_outputs_equal(frames["plain"][f], frames["treelets"][f])
    pos = s.arrays()["meshPositions"].copy()
    pos["Y"] += np.float32(0.01)
    r.update_positions(0, np.stack([pos["X"], pos["Y"], pos["Z"]], 1).astype(np.float32), T.REBUILD_FORCE_REFIT)
    assert low_limits.hrt_debug_treelet_count(r._ctx) == 0
    r.reset_history()
    p = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)
    a, oa = T.alloc_outputs(w, h); b, ob = T.alloc_outputs(w, h)
    r.render_params(p, oa, flags=T.FLAG_STREAMED)
    r.reset_history()
    r.render_params(p, ob, flags=T.FLAG_STREAMED | T.FLAG_TREELETS)
    H.assert_outputs_equal(a, b)


@pytest.mark.parametrize("cfg_id,spp", [(4, 4), (5, 2)])
def test_baseline_meshes_full_size(renderer, cfg_id, spp):
    """Configs 4 / 5 at 3840x2160 on the shipped library and limits: every output array of the treelet walker's frame is the plain
    walker's, bit for bit (128 / 1024+ treelets, several rounds, the clean-up launch)."""
    cfg = scenes.CONFIGS[cfg_id]
    s = engine.Scene(); scenes.build(cfg_id, s)
    renderer.commit(s)
    p = scenes.frame_params(cfg, *H.host_funcs("hrt"), spp=spp)
    names = ["color", "depth", "objectId", "radiance"] + H.RES_NAMES
    out = {}
    for label, fl in (("plain", 0), ("treelets", T.FLAG_TREELETS)):
        renderer.reset_history()
        a, o = T.alloc_outputs(p.width, p.height, names)
        renderer.render_params(p, o, flags=fl)
        out[label] = a
    H.assert_outputs_equal(out["plain"], out["treelets"])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("hostile", [False, True])
def test_random_mesh_scenes_through_the_treelet_walker(orc, hooks_renderer, low_limits, hostile):
    """The differential fuzz of tests/test_fuzz_gpu.py (random sphere sets, 1-3 textured / alpha-cut / transformed meshes, random camera,
    lights, spp, depth 0..7, reuse over two frames; `hostile`: NaN / infinite / degenerate values planted) with treelets of a few nodes:
    every array of every frame against the oracle."""
    from tests import test_fuzz_gpu as FZ
    low_limits.hrt_debug_set_treelet_limits(400, 3, 7)
    r = hooks_renderer
    failures, with_treelets, case = [], 0, 0
    while with_treelets < 24 and case < 400:
        seed = 0x7E1E7 + case + (0x700000 if hostile else 0)
        case += 1
        ops, fr = FZ._scene_recipe(seed)
        if not any(op[0] == "mesh" for op in ops):
            continue
        if hostile:
            ops, fr = FZ._poison(ops, fr, seed)
        so = orc.OrcScene(); FZ._apply(so, ops)
        s = engine.Scene(); FZ._apply(s, ops)
        r.commit(s)
        if low_limits.hrt_debug_treelet_count(r._ctx) == 0:
            continue
        with_treelets += 1
        cfg = scenes.Config("fz", fr["w"], fr["h"], fr["spp"], fr["origin"], fr["lookat"], max_depth=fr["max_depth"], vfov=fr["vfov"],
                            extra={"sun_azimuth": fr["sun"][0], "sun_elevation": fr["sun"][1]})
        w, h = fr["w"], fr["h"]
        r.reset_history()
        A, B = H.new_reservoirs(w, h), H.new_reservoirs(w, h)
        for f in range(2 if fr["reuse"] else 1):
            frame = fr["frame"] + f
            po_ = scenes.frame_params(cfg, *H.host_funcs("orc", orc), frame=frame, reuse=fr["reuse"], rng_lock_noise=fr["lock"])
            pg_ = scenes.frame_params(cfg, *H.host_funcs("hrt"), frame=frame, reuse=fr["reuse"], rng_lock_noise=fr["lock"])
            prev, cur = (B, A) if (frame & 1) == 0 else (A, B)
            ref, oo = T.alloc_outputs(w, h)
            for k, a in cur.items():
                ref[k] = a; setattr(oo, k, a.ctypes.data)
            po = T.Outputs()
            for k, a in prev.items():
                setattr(po, k, a.ctypes.data)
            orc.render_frame(so.desc(), po_, oo, po)
            got, og = T.alloc_outputs(w, h)
            r.render_params(pg_, og, flags=T.FLAG_STREAMED | T.FLAG_TREELETS)
            bad = {k: int(np.count_nonzero(~H.bits_equal(ref[k], got[k]))) for k in ref}
            bad = {k: v for k, v in bad.items() if v}
            if bad:
                failures.append((seed, f, bad))
                break
    assert with_treelets >= 12, "too few random scenes got treelets (%d)" % with_treelets
    assert not failures, "cases that differ from the oracle (seed, frame, {array: elements}): %s" % failures


def test_treelets_survive_instance_moves(hooks_renderer, low_limits):
    """hrt_scene_update_instances regenerates the leaf-slot records and may rebuild the TLAS; the treelets hang on the BLAS roots and stay
    valid: after a move, a TLAS refit and a TLAS rebuild the treelet walker still renders what the plain walker renders."""
    cfg, w, h, spp = scenes.CONFIGS[4], 192, 108, 2
    low_limits.hrt_debug_set_treelet_limits(2048, 7, 32)
    r = hooks_renderer
    s = engine.Scene(); scenes.build_config4(s, 48, 48); r.commit(s)
    n_tl = low_limits.hrt_debug_treelet_count(r._ctx)
    assert n_tl > 0
    mesh = [i for i, rec in enumerate(s.arrays()["instances"]) if rec["type"] == 2][0]
    p = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)
    for policy, xf in ((T.REBUILD_FORCE_REFIT, scenes.rotation_affine("y", 0.0, 1.0, (0.2, 0.05, -0.1))),
                       (T.REBUILD_FORCE_REBUILD, scenes.rotation_affine("y", 25.0, 0.8, (-0.1, 0.1, 0.2)))):
        r.update_instances([mesh], [xf], policy)
        assert low_limits.hrt_debug_treelet_count(r._ctx) == n_tl
        out = {}
        for label, fl in (("plain", T.FLAG_STREAMED), ("treelets", T.FLAG_STREAMED | T.FLAG_TREELETS)):
            r.reset_history()
            a, o = T.alloc_outputs(w, h)
            r.render_params(p, o, flags=fl)
            out[label] = a
        H.assert_outputs_equal(out["plain"], out["treelets"])


def test_treelets_on_a_context_of_two_device_slots(hooks_lib, low_limits):
    """One context over two device slots (both on the one GPU: every slot keeps its own treelets and queues): the interleaved strips of the
    treelet walker's frame are the plain walker's."""
    cfg, w, h, spp = scenes.CONFIGS[4], 192, 112, 2
    low_limits.hrt_debug_set_treelet_limits(2048, 7, 32)
    r = engine.RTRenderer([0, 0], library=hooks_lib)
    try:
        s = engine.Scene(); scenes.build_config4(s, 48, 48); r.commit(s)
        p = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)
        out = {}
        for label, fl in (("plain", T.FLAG_STREAMED), ("treelets", T.FLAG_STREAMED | T.FLAG_TREELETS)):
            r.reset_history()
            a, o = T.alloc_outputs(w, h)
            r.render_params(p, o, flags=fl)
            out[label] = a
        H.assert_outputs_equal(out["plain"], out["treelets"])
    finally:
        r.close()
