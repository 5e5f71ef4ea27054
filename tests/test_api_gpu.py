"""Round-2 additions to the boundary: per-frame event times, page-locked gather targets, the workspace limit, and the second
(device-built) tree that boolean queries of fast-sphere scenes walk."""
import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, engine, scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _params(cfg, w, h, spp):
    return scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)


def test_frame_times_and_registered_gather_targets(hrt_lib):
    r = engine.RTRenderer([0])
    try:
        s = engine.Scene(); scenes.build_config2(s); r.commit(s)
        cfg = scenes.CONFIGS[2]
        w, h = 320, 184
        p = _params(cfg, w, h, 2)
        for _ in range(5):
            r.render_params(p, None, flags=T.FLAG_NO_SYNC)
        st = r.synchronize()
        t0, t1 = r.frame_times(0), r.frame_times(1)
        assert st.frames == 5 and len(t0) == 5 and len(t1) == 5 and np.all(t1 > 0) and np.all(t0 > 0)
        assert abs(float(t1.sum()) - st.kernel_ms[1]) < 1e-3 * max(1.0, st.kernel_ms[1])
        a, oa = T.alloc_outputs(w, h, ["color", "depth", "objectId", "radiance"])
        r.render_params(p, oa)
        b, ob = T.alloc_outputs(w, h, ["color", "depth", "objectId", "radiance"])
        r.register_host(b)
        with pytest.raises(engine.HrtError):
            r.register_host(b)                                       # the same range twice
        r.render_params(p, ob)
        H.assert_outputs_equal(a, b)
        r.unregister_host(b)
        with pytest.raises(engine.HrtError):
            r.unregister_host(b)                                     # no longer registered
        r.render_params(p, ob)                                       # still a valid (pageable) target
        H.assert_outputs_equal(a, b)
        with pytest.raises(engine.HrtError):
            r.set_workspace_limit(-1)
        assert b"gfx950" in hrt_lib.hrt_version()
    finally:
        r.close()


def test_boolean_queries_on_the_second_tree_change_nothing(orc, hrt_lib):
    """600 fast-sphere instances: the production frame walks the device-built tree for shadow rays and the last bounce
    (hrt_walker.hpp, ALT), the counting frame walks the uploaded tree for everything, the oracle restates the reference: all three
    agree bit for bit; and they still do after a scene update has dropped the second tree."""
    def build(b):
        scenes.build_random_spheres(b, 599, seed=0x1234567, extent=6.0)
    cfg = scenes.Config("any", 0, 0, 0, (0.0, 3.0, 11.0), (0.0, 0.8, 0.0))
    w, h, spp = 256, 144, 3
    ref, ost, _ = H.oracle_frame(orc, build, cfg, w, h, spp)
    r = engine.RTRenderer([0])
    try:
        s = engine.Scene(); build(s); r.commit(s)
        p = _params(cfg, w, h, spp)
        prod, op = T.alloc_outputs(w, h)
        r.render_params(p, op, flags=T.FLAG_STREAMED)
        H.assert_outputs_equal(ref, prod)
        r.reset_history()
        cnt, oc = T.alloc_outputs(w, h)
        st = r.render_params(p, oc, flags=T.FLAG_STREAMED | T.FLAG_COUNTERS)
        H.assert_outputs_equal(ref, cnt)
        assert st.k[1].as_dict() == ost.k[1].as_dict()
        # a refit that moves nothing: the second tree is dropped, the picture stays
        r.update_instances([], [], T.REBUILD_FORCE_REFIT)
        r.reset_history()
        again, oa = T.alloc_outputs(w, h)
        r.render_params(p, oa, flags=T.FLAG_STREAMED)
        H.assert_outputs_equal(ref, again)
    finally:
        r.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("count", [40000, 270000])
def test_second_tree_of_a_big_scene(orc, hrt_lib, count):
    """40 001 instances: four renumbered copies of 3 MB each; 270 001: copies of 26 MB each (they need not fit the L2).  A small
    frame in the organisations that use
    the second tree, a sphere update with a refit, and the frame again, against the oracle."""
    ext = 20.0 * (count / 10000.0) ** 0.5
    cfg = scenes.Config("big", 0, 0, 0, (0.0, 6.0, 26.0), (0.0, 1.0, 0.0))
    w, h, spp = 96, 54, 2
    so = orc.OrcScene(); scenes.build_random_spheres(so, count, extent=ext)
    s = engine.Scene(); scenes.build_random_spheres(s, count, extent=ext)
    r = engine.RTRenderer([0])
    try:
        r.commit(s)
        arrs = so.arrays()
        for rnd in range(2):
            if rnd == 1:
                sp = arrs["spheres"].copy()
                sp["center"]["X"][1:] += np.float32(0.1); sp["radius"][1::7] *= np.float32(1.2)
                r.update_spheres(1, sp[1:], T.REBUILD_FORCE_REFIT)
                arrs["spheres"] = sp
                for k in ("blasNodes",): arrs[k] = r.download_array(k)
                nodes, idx, inst, cnt = r.download_tlas()
                arrs["tlasNodes"] = np.frombuffer(nodes, dtype=T.np_dtype(T.BvhNode), count=cnt[0]).copy()
                arrs["tlasInstanceIndices"] = np.frombuffer(idx, dtype=np.int32, count=cnt[1]).copy()
                arrs["instances"] = np.frombuffer(inst, dtype=T.np_dtype(T.InstanceRecord), count=cnt[2]).copy()
            desc, keep = T.scene_desc_from_arrays(arrs)
            po = scenes.frame_params(cfg, *H.host_funcs("orc", orc), width=w, height=h, spp=spp)
            ref, oo = T.alloc_outputs(w, h)
            orc.render_frame(desc, po, oo, None)
            p = _params(cfg, w, h, spp)
            for flags in (0, T.FLAG_STREAMED):
                r.reset_history()
                got, og = T.alloc_outputs(w, h)
                r.render_params(p, og, flags=flags)
                H.assert_outputs_equal(ref, got)
    finally:
        r.close()


def test_equal_distance_instances_on_the_second_tree(orc, hrt_lib):
    """Closest-hit walks over the second tree must still name the instance the UPLOADED tree meets first when several lie at exactly
    the same distance (hrt_walker.hpp, kTies).  400 fast-sphere instances of which 120 are exact duplicates (same centre and radius)
    of others with another colour or shading -- every ray that hits one ties -- listed before or after their twins."""
    def build(b):
        rng = scenes.XorShift32(0xFACEFEED)
        recs = [((0.0, -1000.0, 0.0), 1000.0, (0.6, 0.6, 0.6), T.SHADING_LAMBERT, 1.0)]
        for i in range(280):
            c = (rng.uniform(-5, 5), rng.uniform(0.1, 2.5), rng.uniform(-5, 5))
            recs.append((c, rng.uniform(0.1, 0.4), (rng.uniform(0.2, 0.9), rng.uniform(0.2, 0.9), rng.uniform(0.2, 0.9)), T.SHADING_LAMBERT, 1.0))
        twins = []
        for j in range(120):
            c, rad, kd, _, _ = recs[1 + 2 * j]
            kind = j % 3
            twins.append((c, rad, (1.0 - kd[0], 1.0 - kd[1], kd[2]), [T.SHADING_LAMBERT, T.SHADING_MIRROR, T.SHADING_GLASS][kind], 1.5))
        order = recs[:1] + twins[:60] + recs[1:] + twins[60:]        # half of the twins come first in the instance list, half last
        ids = [b.add_sphere(scenes.sphere(c, rad, kd, sh, ior)) for c, rad, kd, sh, ior in order]
        for i in ids:
            b.build_sphere_instance([i])
        b.rebuild_tlas()
    cfg = scenes.Config("ties", 0, 0, 0, (0.0, 3.0, 9.0), (0.0, 0.8, 0.0), max_depth=4)
    w, h, spp = 224, 128, 3
    ref, ost, _ = H.oracle_frame(orc, build, cfg, w, h, spp)
    r = engine.RTRenderer([0])
    try:
        s = engine.Scene(); build(s); r.commit(s)
        p = _params(cfg, w, h, spp)
        prod, op = T.alloc_outputs(w, h)
        r.render_params(p, op, flags=T.FLAG_STREAMED)
        H.assert_outputs_equal(ref, prod)
        r.reset_history()
        cnt, oc = T.alloc_outputs(w, h)
        st = r.render_params(p, oc, flags=T.FLAG_STREAMED | T.FLAG_COUNTERS)
        H.assert_outputs_equal(ref, cnt)
        assert st.k[1].as_dict() == ost.k[1].as_dict()
    finally:
        r.close()


def test_second_tree_is_not_built_over_stale_world_bounds(orc, renderer):
    """400 one-sphere instances whose uploaded tree is nested and lists every instance once, but ONE instance's own box (its
    one-node BLAS, over a sphere that sits between the old places of the two instances of its leaf) lies outside its worldBounds.  The
    reference walks the uploaded leaf -- whose box holds the sphere -- and sees it; a second tree built from the worldBounds would not.
    Every organisation must show the oracle's picture (the upload path, where round 2 checked own_in_world only after device updates)."""
    cfg = scenes.Config("stale", 0, 0, 0, (0.0, 2.2, 7.5), (0.0, 0.6, 0.0))
    w, h, spp = 160, 100, 2
    so = orc.OrcScene(); scenes.build_random_spheres(so, 400, seed=0x5EED, extent=3.0)
    arrs = {k: v.copy() for k, v in so.arrays().items()}
    tn, ti, inst, bn, sp = arrs["tlasNodes"], arrs["tlasInstanceIndices"], arrs["instances"], arrs["blasNodes"], arrs["spheres"]
    best = None
    for n in tn:
        if n["count"] != 2: continue
        a, b = int(ti[n["first"]]), int(ti[n["first"] + 1])
        if 0 in (a, b): continue                                     # not the ground sphere
        sa, sb = sp[int(arrs["spherePrimIdx"][bn[inst[a]["blasRoot"]]["first"]])], sp[int(arrs["spherePrimIdx"][bn[inst[b]["blasRoot"]]["first"]])]
        ca = np.array([sa["center"][f] for f in "XYZ"]); cb = np.array([sb["center"][f] for f in "XYZ"])
        gap = np.linalg.norm(ca - cb) - sa["radius"] - sb["radius"]
        if best is None or gap > best[0]: best = (gap, a, b, ca, cb, min(sa["radius"], sb["radius"]))
    gap, a, b, ca, cb, rmin = best
    assert gap > 0.1, "no leaf of two well separated spheres in this scene"
    sid = int(arrs["spherePrimIdx"][bn[inst[a]["blasRoot"]]["first"]])
    mid, rad = ((ca + cb) * 0.5).astype(np.float32), np.float32(min(rmin, gap * 0.4))
    for k, f in enumerate("XYZ"):
        sp["center"][f][sid] = mid[k]
        bn["boundsMin"][f][inst[a]["blasRoot"]] = mid[k] - rad
        bn["boundsMax"][f][inst[a]["blasRoot"]] = mid[k] + rad
    sp["radius"][sid] = rad
    sp["albedo"]["X"][sid], sp["albedo"]["Y"][sid], sp["albedo"]["Z"][sid] = 0.95, 0.1, 0.1
    sp["material"]["Kd"]["X"][sid], sp["material"]["Kd"]["Y"][sid], sp["material"]["Kd"]["Z"][sid] = 0.95, 0.1, 0.1
    desc, keep = T.scene_desc_from_arrays(arrs)
    p = scenes.frame_params(cfg, *H.host_funcs("orc", orc), width=w, height=h, spp=spp)
    ref, o = T.alloc_outputs(w, h)
    orc.render_frame(desc, p, o, None)
    renderer.commit(desc)
    try:
        for flags in (0, T.FLAG_STREAMED, T.FLAG_STREAMED | T.FLAG_COUNTERS, T.FLAG_MEGAKERNEL):
            renderer.reset_history()
            pg = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)
            got, og = T.alloc_outputs(w, h)
            renderer.render_params(pg, og, flags=flags)
            H.assert_outputs_equal(ref, got)
    finally:
        s2 = engine.Scene(); scenes.build_config2(s2); renderer.commit(s2)
