"""One ctx over several device slots (the C# host's multi-GPU mode): row-strip tiling, per-tile gather, and the
device-to-device exchange of G-buffer / reservoir tiles that ReSTIR reuse needs.  The 1-GPU test box lists
device 0 several times: every slot is an independent DeviceState (own stream, buffers, strips), so the tiling,
event ordering and exchange code run exactly as on distinct GPUs (peer copies degenerate to on-device copies)."""
import numpy as np
import pytest

from ilgpu_raytracing_amd import _types as T, engine, scenes
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def multi(hrt_lib):
    rs = {n: engine.RTRenderer([0] * n) for n in (2, 3)}
    yield rs
    for r in rs.values():
        r.close()


@pytest.mark.parametrize("n", [2, 3])
def test_multi_slot_frame_equals_oracle(orc, multi, n):
    r = multi[n]
    builder, cfg, w, h, spp = scenes.build_config3, scenes.CONFIGS[3], 160, 100, 2      # 100 rows: 13 strips, ragged last one
    s = engine.Scene(); builder(s); r.commit(s); r.reset_history()
    p = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)
    got, o = T.alloc_outputs(w, h)
    st = r.render_params(p, o, flags=T.FLAG_COUNTERS)
    ref, ost, _ = H.oracle_frame(orc, builder, cfg, w, h, spp)
    H.assert_outputs_equal(ref, got)
    assert st.n_devices == n
    for i in range(2):
        assert st.k[i].as_dict() == ost.k[i].as_dict()           # counters summed over the slots


@pytest.mark.parametrize("n", [2, 3])
def test_multi_slot_restir_reuse_with_tile_exchange(orc, multi, n):
    """Temporal + spatial reuse over 4 frames with a moving camera: every slot needs the other slots' current
    G-buffer (SpatialCompatible) and previous reservoirs; the exchange makes the result identical to one device."""
    r = multi[n]
    builder, cfg = scenes.build_textured_test_scene, scenes.Config("t", 0, 0, 0, (0.3, 1.3, 4.2), (0.0, 0.7, 0.0))
    w, h, spp = 136, 88, 2
    s = engine.Scene(); builder(s); r.commit(s); r.reset_history()
    so = orc.OrcScene(); builder(so)
    A, B = H.new_reservoirs(w, h), H.new_reservoirs(w, h)
    prev_cam = None
    imports = 0
    for f in range(4):
        c2 = scenes.Config("mv", w, h, spp, (0.3 + 0.1 * f, 1.3, 4.2 - 0.05 * f), cfg.cam_lookat)
        p = scenes.frame_params(c2, *H.host_funcs("hrt"), frame=f, reuse=True, prev_cam=prev_cam)
        prev, cur = (B, A) if f % 2 == 0 else (A, B)
        ref, oo = T.alloc_outputs(w, h)
        for k, a in cur.items():
            ref[k] = a; setattr(oo, k, a.ctypes.data)
        po = T.Outputs()
        for k, a in prev.items():
            setattr(po, k, a.ctypes.data)
        ost = orc.render_frame(so.desc(), p, oo, po)
        got, og = T.alloc_outputs(w, h)
        gst = r.render_params(p, og, flags=T.FLAG_COUNTERS)
        H.assert_outputs_equal(ref, got)
        assert gst.k[1].as_dict() == ost.k[1].as_dict()
        imports += gst.k[1].reuse_imports
        prev_cam = engine.copy_camera(p.cam)
    assert imports > w * h
    with pytest.raises(engine.HrtError):                         # partial tiles still cannot do reuse
        r.render_params(p, None, rows=(0, 40))
    with pytest.raises(engine.HrtError):
        r.render_params(p, None, flags=T.FLAG_NO_SYNC)


def test_multi_slot_present(orc, multi):
    r = multi[2]
    s = engine.Scene(); scenes.build_config2(s); r.commit(s); r.reset_history()
    w, h, ow, oh = 120, 72, 180, 108
    p = scenes.frame_params(scenes.CONFIGS[2], *H.host_funcs("hrt"), width=w, height=h, spp=1)
    low, o = T.alloc_outputs(w, h, ["color", "objectId"])
    r.render_params(p, o)
    hist = (np.zeros(ow * oh, np.int32), np.zeros(ow * oh, np.int32))
    assert np.array_equal(r.present(ow, oh, taau=True), orc.present(1, low["color"], low["objectId"], w, h, ow, oh, history=hist, first_frame=True))
    assert np.array_equal(r.present(ow, oh, taau=False), orc.present(0, low["color"], low["objectId"], w, h, ow, oh))


def test_multi_slot_async_frames(multi):
    r = multi[3]
    s = engine.Scene(); scenes.build_config2(s); r.commit(s)
    p = scenes.frame_params(scenes.CONFIGS[2], *H.host_funcs("hrt"), width=256, height=144, spp=2)
    ref, o = T.alloc_outputs(256, 144, ["color", "radiance"])
    r.render_params(p, o)
    for _ in range(4):
        r.render_params(p, None, flags=T.FLAG_NO_SYNC)
    st = r.synchronize()
    assert st.frames == 4 and st.n_devices == 3
    got, o2 = T.alloc_outputs(256, 144, ["color", "radiance"])
    r.render_params(p, o2)
    H.assert_outputs_equal(ref, got)


@pytest.mark.parametrize("policy", [T.REBUILD_FORCE_REFIT, T.REBUILD_FORCE_REBUILD])
def test_moved_instances_on_every_slot(orc, multi, policy):
    """hrt_scene_update_instances updates the tree of every device slot the same way (same kernels, deterministic):
    the slots' downloads are equal and the tiled frame equals the oracle on the downloaded arrays."""
    from tests import test_bvh_update_gpu as B
    r = multi[2]
    builder, cfg, w, h, spp = B.SCENES["sphere_instances"]
    s = engine.Scene(); builder(s); r.commit(s); r.reset_history()
    so = orc.OrcScene(); builder(so)
    ids, xfs = B._moves(len(so.arrays()["instances"]), "rigid")
    r.update_instances(ids, xfs, policy)
    for i, m in zip(ids, xfs):
        so.set_instance_transform(i, m)
    per_slot = []
    for slot in range(2):
        nodes, idx, inst, cnt = r.download_tlas(slot)
        per_slot.append((B._as_np(nodes, cnt[0], T.BvhNode), np.frombuffer(idx, dtype=np.int32, count=cnt[1]).copy(), B._as_np(inst, cnt[2], T.InstanceRecord)))
    for a, b in zip(per_slot[0], per_slot[1]):
        assert a.tobytes() == b.tobytes()
    nodes, idx, inst = per_slot[0]
    assert inst.tobytes() == so.arrays()["instances"].tobytes()
    ref, ost = B._oracle_render(orc, B._desc_with_tlas(so.desc(), nodes, idx, inst), cfg, w, h, spp)
    p = scenes.frame_params(cfg, *H.host_funcs("hrt"), width=w, height=h, spp=spp)
    got, o = T.alloc_outputs(w, h)
    st = r.render_params(p, o, flags=T.FLAG_COUNTERS)
    H.assert_outputs_equal(ref, got)
    for i in range(2):
        assert st.k[i].as_dict() == ost.k[i].as_dict()
